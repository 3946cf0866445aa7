"""Host-side mirror of the reference's models/modules.py: `get_mlp` (parameter container with the reference's
state_dict layout) and `LanguageEncoder` (host tokenisation + HIP embedding/biLSTM recurrence).

Reference: models/modules.py:11-36 (get_mlp: every layer, including the last, is Linear -> BatchNorm1d -> ReLU),
models/modules.py:39-92 (LanguageEncoder).
"""
from typing import List

import numpy as np
import torch
import torch.nn as nn

from . import ops, packing


def get_mlp(channels: List[int], add_batchnorm: bool = True) -> nn.Sequential:
    """Parameter container only: on the product path its weights are folded/packed and run by the HIP kernels."""
    blocks = []
    for c_in, c_out in zip(channels[:-1], channels[1:]):
        parts = [nn.Linear(c_in, c_out)]
        if add_batchnorm:
            parts.append(nn.BatchNorm1d(c_out))
        parts.append(nn.ReLU())
        blocks.append(nn.Sequential(*parts))
    return nn.Sequential(*blocks)


def tokenize(descriptions: List[str], known_words: dict):
    """models/modules.py:60-72: strip '.' and ',', lower-case, split on whitespace, unknown -> 0 (the padding row);
    right-pad with 0.  Returns (int32 [B, T_max], int32 [B])."""
    rows = [[known_words.get(w, 0) for w in d.replace(".", "").replace(",", "").lower().split()] for d in descriptions]
    lengths = np.array([len(r) for r in rows], dtype=np.int32)
    if len(rows) == 0 or int(lengths.min()) < 1:
        # torch's pack_padded_sequence, which the reference calls, rejects zero-length sequences the same way
        raise RuntimeError("Length of all samples has to be greater than 0, but found an element that is <= 0")
    padded = np.zeros((len(rows), int(lengths.max())), dtype=np.int32)
    for i, r in enumerate(rows):
        padded[i, : len(r)] = r
    return padded, lengths


class LanguageEncoder(nn.Module):
    def __init__(self, known_words, embedding_dim, bi_dir, num_layers=1):
        super().__init__()
        if not bi_dir or num_layers != 1:
            raise NotImplementedError("only the configuration the reference uses is built: 1 layer, bidirectional")
        self.known_words = {w: i + 1 for i, w in enumerate(known_words)}
        self.known_words["<unk>"] = 0
        self.word_embedding = nn.Embedding(len(self.known_words), embedding_dim, padding_idx=0)
        self.lstm = nn.LSTM(input_size=embedding_dim, hidden_size=embedding_dim, bidirectional=True, num_layers=1)
        self._pack = None

    @property
    def device(self):
        return self.word_embedding.weight.device

    def _weights(self):
        ver = (packing.params_version(self), str(self.device))
        if self._pack is None or self._pack[0] != ver:
            t = packing.pack_text_weights(self, self.device)
            self._pack = (ver, t, ops.make_text_weights(t["embedding"], t["w_ih"], t["w_hh"], t["bias"]))
        return self._pack[2]

    def encode_tokens(self, tokens: torch.Tensor, lengths: torch.Tensor, normalize: bool):
        d = self.word_embedding.embedding_dim
        out, raw = ops.encode_text(tokens, lengths, self._weights(), self.word_embedding.num_embeddings, d, want_raw=True)
        return out if normalize else raw

    def forward(self, descriptions, normalize: bool = False):
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            raise NotImplementedError("the HIP text path is forward-only; call it under torch.no_grad()")
        padded, lengths = tokenize(descriptions, self.known_words)
        dev = self.device
        tok = torch.from_numpy(padded).to(dev, non_blocking=True)
        ln = torch.from_numpy(lengths).to(dev, non_blocking=True)
        return self.encode_tokens(tok, ln, normalize)
