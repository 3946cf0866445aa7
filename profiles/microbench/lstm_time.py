"""Time of one text-encoder call (t2p_encode_text) at 64 and 1,000 queries, D = 256; T2P_LIB selects an ablation build of
csrc/lstm.hip (T2P_LSTM_ABL).  usage (GPU box, repo root): python profiles/microbench/lstm_time.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import text2pos_amd as t2p  # noqa: E402
from text2pos_amd import synthetic as S  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(1234)
m = t2p.CellRetrievalNetwork(S.LABELS + ["pad"], S.COLOR_NAMES, S.known_words(), S.default_args()).to(dev).eval()
for prec in ("f16x3", "fp32"):
    m.language_encoder.precision = prec
    for n in (64, 1000):
        texts = S.make_texts(5, 0, n)
        with torch.no_grad():
            for _ in range(3):
                m.encode_text(texts)
            torch.cuda.synchronize()
            from text2pos_amd.modules import tokenize
            padded, lengths = tokenize(texts, m.language_encoder.known_words)
            tok, ln = torch.from_numpy(padded).to(dev), torch.from_numpy(lengths).to(dev)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                m.language_encoder.encode_tokens(tok, ln, True)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 20
        print(f"{os.environ.get('T2P_LIB', 'default')[-12:]:>12s} {prec:6s} {n:5d} queries, T = {padded.shape[1]}: {dt * 1e3:.3f} ms per call, "
              f"{dt / padded.shape[1] * 1e6:.1f} us per time step")
