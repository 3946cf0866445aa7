"""Small-batch rate of the packed entry point against the number of parts / HIP streams the batch is cut into
(CellRetrievalNetwork.encode_objects_packed(streams=n)): the reference's callers hand over 64 cells per call (training/coarse.py:123-131).
    python profiles/tools/small_batch_streams.py [batch sizes ...]"""
import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch
import weights as W
import text2pos_amd as t2p
from text2pos_amd import synthetic as S

model = t2p.CellRetrievalNetwork(S.LABELS + ["pad"], S.COLOR_NAMES, S.known_words(), S.default_args())
W.fill_state_dict(model, 29)
model = model.cuda().eval()
for B in [int(a) for a in sys.argv[1:]] or [64, 128, 256, 512, 1024]:
    c = S.make_cells(41, B)
    cells = [torch.from_numpy(a).cuda() for a in c[:4]]
    ref = None
    row = []
    for streams in (1, 2, 3, 4):
        with torch.no_grad():
            for _ in range(5):
                out = model.encode_objects_packed(*cells, c[4], streams=streams).cpu()
            ref = out if ref is None else ref
            assert torch.equal(out, ref), "parts change the result"
            ts = []
            for rep in range(3):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(50):
                    model.encode_objects_packed(*cells, c[4], streams=streams).cpu()
                ts.append((time.perf_counter() - t0) / 50)
        row.append(f"streams={streams}: {1e3 * min(ts):.3f} ms ({B / min(ts) / 1e3:.1f} k cells/s)")
    print(f"B={B} ({cells[0].shape[0]} objects): " + "  ".join(row), flush=True)
