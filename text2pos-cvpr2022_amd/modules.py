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


_LSTM_SIDE = {}


def _lstm_side_stream(dev):
    key = str(dev)
    if key not in _LSTM_SIDE:
        _LSTM_SIDE[key] = ops.concurrent_stream(dev)
    return _LSTM_SIDE[key]


class _LstmTrainFn(torch.autograd.Function):
    """Training-mode text branch (first part): Embedding -> packed biLSTM -> mean of the two final
    hidden states (models/modules.py:77-90) with a backward pass, for `anchor = model.encode_text(...); loss.backward()`
    (training/coarse.py:44-58).  The recurrence runs on the HIP kernels behind t2p_lstm_cell_forward / _backward and
    t2p_gemm, the time loop inside the library (t2p_lstm_train_forward / _backward) when the width allows, else step by step from
    here (the persistent inference kernel keeps no activations); the weight-gradient reductions at the end are plain
    library GEMMs over the stored activations.  Parameters come in nn.LSTM's own layout (weight_* [4D, D], gates i f g o)."""

    @staticmethod
    def forward(ctx, tokens, lengths, emb, w_ih, w_hh, b_ih, b_hh, w_ih_r, w_hh_r, b_ih_r, b_hh_r):
        b, t = tokens.shape
        d = emb.shape[1]
        dev = emb.device
        emb_c = emb.detach().contiguous()
        wih_k = [w_ih.detach().t().contiguous(), w_ih_r.detach().t().contiguous()]      # k-major [D, 4D]
        whh_k = [w_hh.detach().t().contiguous(), w_hh_r.detach().t().contiguous()]
        bias = [(b_ih + b_hh).detach().contiguous(), (b_ih_r + b_hh_r).detach().contiguous()]
        gates = torch.empty((2, t, b, 4 * d), dtype=torch.float32, device=dev)
        cs = torch.zeros((2, t + 1, b, d), dtype=torch.float32, device=dev)
        hs = torch.zeros((2, t + 1, b, d), dtype=torch.float32, device=dev)
        # the time loops run inside the library where the width allows: 1 call per direction instead of 2 T
        in_library = dev.type == "cuda" and ops.lstm_train_loops_supported(d)
        if in_library:
            main, side = torch.cuda.current_stream(dev), _lstm_side_stream(dev)
            side.wait_stream(main)
        for dr in (0, 1):
            if in_library:                                  # the two directions overlap on two HIP streams
                with torch.cuda.stream((main, side)[dr]):
                    table = ops.gemm(emb_c, wih_k[dr], bias[dr])
                    ops.lstm_train_forward(table, whh_k[dr], tokens, lengths, dr == 1, gates[dr], cs[dr], hs[dr])
                continue
            table = ops.gemm(emb_c, wih_k[dr], bias[dr])                                 # [V, 4D] = E W_ih^T + b_ih + b_hh
            for s in range(t):
                pre = ops.gemm(hs[dr, s], whh_k[dr])
                ops.lstm_cell_forward(pre, table, tokens, lengths, s, dr == 1, cs[dr, s], hs[dr, s], gates[dr, s],
                                      cs[dr, s + 1], hs[dr, s + 1])
        if in_library:
            main.wait_stream(side)
        ctx.save_for_backward(tokens, lengths, emb_c, wih_k[0], wih_k[1], whh_k[0], whh_k[1], gates, cs, hs)
        return 0.5 * (hs[0, t] + hs[1, t])

    @staticmethod
    def backward(ctx, dout):
        tokens, lengths, emb_c, wih0, wih1, whh0, whh1, gates, cs, hs = ctx.saved_tensors
        b, t = tokens.shape
        d, v = emb_c.shape[1], emb_c.shape[0]
        dev = emb_c.device
        steps = torch.arange(t, device=dev, dtype=torch.int64)[:, None]                   # [T, 1]
        ln = lengths.to(torch.int64)[None, :]                                             # [1, B]
        active = steps < ln                                                               # [T, B]
        # The two directions are independent until their gradients are added: their (tiny, latency-bound) per-step kernels are
        # issued alternately on two HIP streams and overlap on the GPU (9.2 -> ~5 ms at 64 texts; the forward loop is bound by
        # the host's launch rate and stays on one stream).
        main = torch.cuda.current_stream(dev)
        side = _lstm_side_stream(dev)
        streams = (main, side)
        whh_t = [whh0.t().contiguous(), whh1.t().contiguous()]                            # [4D, D]: d_pre -> dh_{s-1}
        side.wait_stream(main)
        d_pre = [torch.empty((t, b, 4 * d), dtype=torch.float32, device=dev) for _ in range(2)]
        dh_carry, dc, dh_gemm = [None, None], [None, None], [None, None]
        in_library = ops.lstm_train_loops_supported(d)
        for dr in (0, 1):
            with torch.cuda.stream(streams[dr]):
                dh_carry[dr] = (0.5 * dout).contiguous()
                if in_library:
                    ops.lstm_train_backward(dh_carry[dr], whh_t[dr], gates[dr], cs[dr], lengths, d_pre[dr])
                else:
                    dc[dr] = torch.zeros((b, d), dtype=torch.float32, device=dev)
        for s in (range(t - 1, -1, -1) if not in_library else ()):
            for dr in (0, 1):
                with torch.cuda.stream(streams[dr]):
                    dc_new, carry_new = torch.empty_like(dc[dr]), torch.empty_like(dc[dr])
                    ops.lstm_cell_backward(dh_gemm[dr], dh_carry[dr], dc[dr], gates[dr, s], cs[dr, s], cs[dr, s + 1], lengths, s,
                                           d_pre[dr][s], dc_new, carry_new)
                    # (matmul: pads N = D to the GEMM's granule of 8, e.g. D = 300)
                    dh_gemm[dr] = ops.matmul(d_pre[dr][s], whh_t[dr]) if s > 0 else None
                    dc[dr], dh_carry[dr] = dc_new, carry_new
        d_emb_parts, grads = [], []
        for dr, (wih_k, whh_k) in enumerate(((wih0, whh0), (wih1, whh1))):
            with torch.cuda.stream(streams[dr]):
                flat = d_pre[dr].reshape(t * b, 4 * d)
                d_whh_k = ops.gemm_tn(hs[dr, :t].reshape(t * b, d), flat)                 # [D, 4D] = H^T dPre
                # gate-table gradient: rows of d_pre summed per token (one-hot product: deterministic, V is ~40)
                pos = steps if dr == 0 else (ln - 1 - steps).clamp(min=0)                 # token position of (step, sequence)
                tok = torch.gather(tokens.to(torch.int64).t(), 0, pos.expand(t, b))       # [T, B]
                onehot_t = torch.zeros((t * b, v), dtype=torch.float32, device=dev)
                onehot_t.scatter_(1, tok.reshape(t * b, 1), active.reshape(t * b, 1).to(torch.float32))
                d_table = ops.gemm_tn(onehot_t, flat)                                     # [V, 4D]: one-hot^T dPre
                d_bias = d_table.sum(0)
                d_wih_k = ops.gemm_tn(emb_c, d_table)                                     # [D, 4D] = E^T dTable
                d_emb_parts.append(ops.matmul(d_table, wih_k.t().contiguous()))
                grads += [d_wih_k.t().contiguous(), d_whh_k.t().contiguous(), d_bias, d_bias.clone()]
        main.wait_stream(side)
        for tns in grads[4:] + [d_emb_parts[1]] + d_pre:      # produced on the side stream, consumed (and later freed) on the main one
            tns.record_stream(main)
        d_emb = d_emb_parts[0] + d_emb_parts[1]                                           # (fixed order: forward direction first)
        d_emb[0].zero_()                                                                  # nn.Embedding(padding_idx=0)
        return (None, None, d_emb) + tuple(grads)


class PicklableModule(nn.Module):
    """`torch.save(model, path)` is how the reference writes its checkpoints (training/coarse.py:323-324: the whole module,
    pickled).  The product modules cache things a pickle cannot carry - ctypes weight descriptors, HIP streams, pinned staging
    buffers - under the attribute names a class lists in `_TRANSIENT` ({name: value or zero-argument factory}); they leave the
    pickle as that fresh value and are rebuilt on first use after loading."""
    _TRANSIENT = {}

    def __getstate__(self):
        state = self.__dict__.copy()
        for name, fresh in self._TRANSIENT.items():
            if name in state:
                state[name] = fresh() if callable(fresh) else fresh
        return state


class LanguageEncoder(PicklableModule):
    _TRANSIENT = {"_pack": None}

    def __init__(self, known_words, embedding_dim, bi_dir, num_layers=1):
        super().__init__()
        if not bi_dir or num_layers != 1:
            raise NotImplementedError("only the configuration the reference uses is built: 1 layer, bidirectional")
        self.known_words = {w: i + 1 for i, w in enumerate(known_words)}
        self.known_words["<unk>"] = 0
        self.word_embedding = nn.Embedding(len(self.known_words), embedding_dim, padding_idx=0)
        self.lstm = nn.LSTM(input_size=embedding_dim, hidden_size=embedding_dim, bidirectional=True, num_layers=1)
        self.precision = "f16x3"   # arithmetic of the inference recurrence ("fp32": exact fp32 MFMA); the owning model sets it
        self._embedding_dim = int(embedding_dim)
        self._kernel_dim = None    # hidden width the HIP recurrence runs (zero-padded, packing.kernel_embed_dim): resolved on first use,
        self._pack = None          # so that a width only the gradient-mode path supports can still be constructed and trained

    @property
    def kernel_dim(self):
        if self._kernel_dim is None:
            self._kernel_dim = packing.kernel_embed_dim(self._embedding_dim)   # raises NotImplementedError past 384
        return self._kernel_dim

    @kernel_dim.setter
    def kernel_dim(self, value):
        self._kernel_dim = int(value)

    @property
    def device(self):
        return self.word_embedding.weight.device

    def _weights(self):
        ver = (packing.params_version(self), str(self.device), self.precision, self.kernel_dim)
        if self._pack is None or self._pack[0] != ver:
            try:
                t = packing.pack_text_weights(self, self.device, x3=self.precision == "f16x3", pad_to=self.kernel_dim)
            except packing.Fp16RangeError as e:   # a recurrent weight outside fp16's range: the exact path has no such limit
                import warnings
                warnings.warn(f"LanguageEncoder: {e}; the text branch runs its exact fp32 recurrence instead (about 3x "
                              "slower than the f16x3 one)", RuntimeWarning, stacklevel=3)
                t = packing.pack_text_weights(self, self.device, x3=False, pad_to=self.kernel_dim)
            self._pack = (ver, t, ops.make_text_weights(t["embedding"], t["w_ih"], t["w_hh"], t["bias"], t.get("w_hh_x3"),
                                                        t.get("w_hh_scale", 0.0)))
        return self._pack[2]

    def _wants_grad(self):
        return torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())

    def encode_tokens(self, tokens: torch.Tensor, lengths: torch.Tensor, normalize: bool):
        d = self.word_embedding.embedding_dim
        if self._wants_grad():  # training step: the step-wise recurrence that keeps its activations
            m = self.lstm
            raw = _LstmTrainFn.apply(tokens, lengths, self.word_embedding.weight, m.weight_ih_l0, m.weight_hh_l0, m.bias_ih_l0,
                                     m.bias_hh_l0, m.weight_ih_l0_reverse, m.weight_hh_l0_reverse, m.bias_ih_l0_reverse,
                                     m.bias_hh_l0_reverse)
            return nn.functional.normalize(raw, dim=-1) if normalize else raw
        out, raw = ops.encode_text(tokens, lengths, self._weights(), self.word_embedding.num_embeddings, self.kernel_dim, want_raw=True)
        if self.kernel_dim != d:      # zero padding of the hidden width: cut off (it does not change the norm)
            out, raw = out[:, :d].contiguous(), raw[:, :d].contiguous()
        return out if normalize else raw

    def forward(self, descriptions, normalize: bool = False):
        padded, lengths = tokenize(descriptions, self.known_words)
        dev = self.device
        tok = torch.from_numpy(padded).to(dev, non_blocking=True)
        ln = torch.from_numpy(lengths).to(dev, non_blocking=True)
        return self.encode_tokens(tok, ln, normalize)
