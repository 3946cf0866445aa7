"""Drop-in for the reference's coarse model: same constructor, `encode_text`, `encode_objects`, `embed_dim`,
`device` / `get_device()` and state_dict layout as models/cell_retrieval.py::CellRetrievalNetwork, with the forward
arithmetic executed by libt2p_hip.so on an MI355X.

Callers that drop in unchanged: training/coarse.py:111,124 (eval_epoch), evaluation/pipeline.py:73-75,111-113 and the
training step training/coarse.py:31-62.  eval() + torch.no_grad() runs the folded inference kernels (t2p_encode_cells /
t2p_encode_text); train() runs the batch-statistics path of train_cell.py / modules._LstmTrainFn with backward kernels
(SURVEY.md 8(f) #4).  Anything else (gradients through the folded kernels, the stage trace in train mode) raises; nothing
falls back to another implementation.
"""
import warnings
from typing import List

import numpy as np
import torch
import torch.nn as nn

from . import ops, packing
from .data import HostStaging, ObjectMeansCache, pack_cells
from .modules import LanguageEncoder, PicklableModule, get_mlp
from .object_encoder import ObjectEncoder


class DynamicEdgeConv(nn.Module):
    """Holds `nn` under the key torch_geometric.nn.DynamicEdgeConv uses (`graph1.nn.*`)."""

    def __init__(self, nn_, k, aggr="max"):
        super().__init__()
        self.nn, self.k, self.aggr = nn_, k, aggr


class CellRetrievalNetwork(PicklableModule):
    # what `torch.save(model, path)` (training/coarse.py:323-324) must not try to pickle: packed-weight descriptors, the guard word,
    # HIP streams, pinned staging, the per-cell means memo (it would drag the dataset's objects into the checkpoint)
    _TRANSIENT = {"_pack": None, "_overflow": None, "_aux_streams": None, "_aux_beside": None, "_copy_stream": None, "_staging": HostStaging,
                  "object_means_cache": ObjectMeansCache}

    def __init__(self, known_classes: List[str], known_colors: List[str], known_words: List[str], args,
                 add_self_loops: bool = True, precision: str = "f16x3", on_overflow: str = "raise"):
        """add_self_loops=True reproduces torch_geometric's PointConv default, which the reference relies on
        (models/pointcloud/pointnet2.py:23); False gives the plain ball-query neighbourhoods.
        on_overflow: what encode_objects* do when the fp16-range guard of the f16x3 path fires (an activation of this
        checkpoint on this input left fp16's range, include/t2p.h t2p_cell_config.overflow_flag): "raise"
        (FloatingPointError) or "fp32" (warn and recompute the call on the exact fp32 MFMA path)."""
        super().__init__()
        self.embed_dim = args.embed_dim
        # width the kernels run: 128 / 256 as is, anything else up to 384 zero-padded to 384 (training/args.py:19 defaults to
        # 300); the padded channels are exactly 0 and are cut off again before a result leaves this class
        self.kernel_dim = packing.kernel_embed_dim(int(args.embed_dim))
        self.use_features = args.use_features
        self.variation = args.variation
        self.args = args
        self.add_self_loops = add_self_loops
        # "f16x3": the MFMA-heavy layer-2 GEMMs run as three fp16 MFMAs on hi/lo-split operands with fp32 accumulation
        # (split error ~5e-7, below an fp32 fma chain's own rounding); "fp32": exact fp32 MFMA everywhere.
        self.precision = precision
        if on_overflow not in ("raise", "fp32"):
            raise ValueError("on_overflow must be 'raise' or 'fp32'")
        self.on_overflow = on_overflow
        self._overflow = None
        self.cell_streams = None   # default of encode_objects_packed(streams=None): None = 2 parts from 2,048 cells up, else 1
        self.tuning = 0          # t2p_cell_config.tuning: A/B switches between equivalent execution plans (include/t2p.h)
        d = self.embed_dim
        assert args.variation in (0, 1)
        self.graph1 = DynamicEdgeConv(get_mlp([2 * d, d, d], add_batchnorm=True), k=8,
                                      aggr="max" if args.variation == 0 else "mean")
        self.lin = get_mlp([d, d, d])
        self.object_encoder = ObjectEncoder(d, known_classes, known_colors, args)
        self.language_encoder = LanguageEncoder(known_words, d, bi_dir=True)
        self.language_encoder.precision = precision
        self.language_encoder.kernel_dim = self.kernel_dim
        self._pack = None
        self._staging = HostStaging()              # pinned host buffers of encode_objects, kept across calls
        self.object_means_cache = ObjectMeansCache()   # per-cell (centre, mean colour) rows; .clear() after editing objects in place

    # ---- text branch -----------------------------------------------------------------------------------------

    # `precision` also selects the text branch's recurrence: one switch for the whole model, also when it is flipped after
    # construction (bench.py's fp32 pass and the on_overflow="fp32" recomputation do exactly that)
    @property
    def precision(self):
        return self._precision

    @precision.setter
    def precision(self, value):
        if value not in ("f16x3", "fp32"):
            raise ValueError("precision must be 'f16x3' or 'fp32'")
        self._precision = value
        lang = self._modules.get("language_encoder") if "_modules" in self.__dict__ else None
        if lang is not None:
            lang.precision = value
    def encode_text(self, descriptions):
        """List[str] -> [B, D] fp32, L2-normalised (models/cell_retrieval.py:69-75).  With gradients enabled the text
        branch runs its training-mode recurrence and the result carries a grad_fn (training/coarse.py:44)."""
        return self.language_encoder(descriptions, normalize=True)

    # ---- cell branch -----------------------------------------------------------------------------------------
    def _cell_pack(self, precision=None):
        x3 = (precision or self.precision) == "f16x3"
        ver = (packing.params_version(self), str(self.device))
        if self._pack is None or self._pack[0] != ver or (x3 and not self._pack[3]):
            tensors = packing.pack_cell_weights(self, self.device, x3=x3)
            self._pack = (ver, tensors, ops.make_cell_weights(tensors), x3)
        return self._pack[2]

    def _overflow_word(self):
        """The sticky fp16-range guard word of this model's f16x3 calls (int32 [1] on the device)."""
        if self._overflow is None or self._overflow.device != self.device:
            self._overflow = torch.zeros(1, dtype=torch.int32, device=self.device)
        return self._overflow

    def overflow_detected(self) -> int:
        """Reads (synchronises) and clears the guard word: non-zero = some f16x3 call since the last check converted an
        activation outside fp16's range and its result must not be used (bit meanings: include/t2p.h)."""
        if self._overflow is None:
            return 0
        code = int(self._overflow.item())
        if code:
            self._overflow.zero_()
        return code

    def _cell_config(self, n_pts, chunk_objects=0, class_idx=None, color_idx=None, precision=None):
        a = self.args
        if bool(getattr(a, "class_embed", False)) != (class_idx is not None) or \
                bool(getattr(a, "color_embed", False)) != (color_idx is not None):
            raise RuntimeError("args.class_embed / args.color_embed need the per-object class / colour indices "
                               "(encode_objects derives them from the Object3d labels; pass class_idx / color_idx to "
                               "encode_objects_packed)")
        radii = self.object_encoder.pointnet.radii
        return ops.make_cell_config(n_pts=n_pts, embed_dim=self.kernel_dim, pointnet_features=a.pointnet_features,
                                    use_features=tuple(a.use_features), self_loops=self.add_self_loops,
                                    knn_k=self.graph1.k, variation=self.variation, radius=radii,
                                    chunk_objects=chunk_objects, precision=precision or self.precision,
                                    class_idx=class_idx, color_idx=color_idx, tuning=self.tuning,
                                    overflow_flag=self._overflow_word() if (precision or self.precision) == "f16x3" else None)

    _GUARD_BITS = ("bits 0-2 = SA level 1-3 edge inputs, 3 = SA output rows, 4 = GA hidden planes, 5 = GEMM rows past fp16's "
                   "largest value; 6 = NaN among the input points / colours; 7 = a level's activations too SMALL for the fp16 "
                   "pieces (largest magnitude below 2^-7: their low parts would underflow)")

    def _with_guard(self, run):
        """run() -> result of an encode on the CURRENT precision.  On the f16x3 path the sticky guard word is read after the
        launch (one host synchronisation) and acted on as `on_overflow` says: raise, or warn and call run() again with the
        model switched to the exact fp32 path.  The one place this logic lives: every entry point (single stream, several
        streams, pinned-host blocks, the two pipelined halves of encode_objects, the scene path) goes through it."""
        out = run()
        if self.precision != "f16x3":
            return out
        code = self.overflow_detected()
        if not code:
            return out
        msg = f"f16x3 path: an activation left the range its fp16 pieces cover (guard code {code:#x}: {self._GUARD_BITS})"
        if self.on_overflow != "fp32":
            raise FloatingPointError(msg + "; construct the model with precision=\"fp32\" or on_overflow=\"fp32\"")
        warnings.warn(msg + "; recomputing this call on the exact fp32 path", RuntimeWarning)
        saved, self.precision = self.precision, "fp32"
        try:
            return run()
        finally:
            self.precision = saved

    def _trim(self, out):
        """Cuts the zero padding of kernel_dim off an [n, kernel_dim] result (or an (embeddings, trace) pair)."""
        if self.kernel_dim == self.embed_dim:
            return out
        if isinstance(out, tuple):
            emb, tr = out
            if isinstance(tr, dict) and tr.get("obj_emb") is not None:
                tr["obj_emb"] = tr["obj_emb"][:, : self.embed_dim].contiguous()
            return emb[:, : self.embed_dim].contiguous(), tr
        return out[:, : self.embed_dim].contiguous()

    def _check_forward_only(self):
        if self.training:
            raise NotImplementedError("training-mode forward (batch-statistics BatchNorm) is not built for this entry "
                                      "point; call .eval()")
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            raise NotImplementedError("the inference kernels are forward-only (BatchNorm folded): call them under "
                                      "torch.no_grad(), or put the model in train() for the training-mode path")

    def encode_objects_packed(self, xyz, rgb, center, mean_rgb, cell_ptr, cell_ptr_dev=None, want_trace=False,
                              chunk_objects=0, class_idx=None, color_idx=None, check_overflow=True, streams=None):
        """Device-resident packed inputs: xyz/rgb [Nobj, P, 3], center/mean_rgb [Nobj, 3] (fp32, on self.device),
        cell_ptr int32 [B+1] on the host.  Returns [B, D] L2-normalised.  In train() mode (training/coarse.py:32) the
        batch-statistics path of train_cell.py runs instead of the folded inference kernels and the result carries a
        grad_fn.
        check_overflow (f16x3 only): read the fp16-range guard word after the launch (one host synchronisation; the
        reference's callers move the result to the host right away, training/coarse.py:115) and act as `on_overflow` says.
        Pipelined callers pass False and call overflow_detected() once their stream has drained.
        streams = n > 1: the batch is cut into n parts (whole cells, equal object counts) that run on the current stream and
        on n - 1 further HIP streams, each with its own workspace: the latency- and store-bound kernels of one part (FPS / ball
        query, the layer-1 tables) then run under the matrix-bound kernels of another - the chip is power-limited under matrix
        load, so what runs beside an MFMA kernel's tail or on its idle issue slots is nearly free (-2.5 % per step with two).
        Cells are independent: same result, bit for bit.  None (default): 2 from 2,048 cells up, else 1; 1 = everything on
        the current stream (what per-kernel event timings and the rocprofv3 evidence runs need); `self.cell_streams`
        overrides the default for every call of this model."""
        if self.training and not want_trace:
            from .train_cell import encode_objects_train
            return encode_objects_train(self, xyz, rgb, center, mean_rgb, cell_ptr, class_idx, color_idx)
        self._check_forward_only()
        cp = np.ascontiguousarray(np.asarray(cell_ptr), dtype=np.int32)
        if cell_ptr_dev is None:
            cell_ptr_dev = torch.from_numpy(cp).to(self.device)
        if "color" not in self.args.use_features and not getattr(self.args, "class_embed", False):
            rgb = torch.zeros_like(rgb)   # models/object_encoder.py:86-90: the PointNet++ then sees x = 0
        if streams is None:
            streams = self.cell_streams if self.cell_streams else (2 if cp.shape[0] - 1 >= 2048 else 1)

        def run():
            if streams > 1 and not want_trace and cp.shape[0] - 1 >= streams:
                return self._encode_multi_stream(xyz, rgb, center, mean_rgb, cp, cell_ptr_dev, chunk_objects, class_idx, color_idx,
                                                 int(streams))
            cfg = self._cell_config(xyz.shape[1], chunk_objects, class_idx, color_idx)
            return ops.encode_cells(xyz, rgb, center, mean_rgb, cp, cell_ptr_dev, self._cell_pack(), cfg, want_trace)
        return self._trim(self._with_guard(run) if check_overflow and cp.shape[0] > 1 else run())

    def _encode_multi_stream(self, xyz, rgb, center, mean_rgb, cp, cell_ptr_dev, chunk_objects, class_idx, color_idx, n_streams):
        dev = self.device
        n_cells = cp.shape[0] - 1
        # part boundaries: first cell whose start lies past k / n of the objects (whole cells, at least one per part)
        cuts = [0]
        for k in range(1, n_streams):
            c = int(np.searchsorted(cp, cp[-1] * k // n_streams))
            cuts.append(min(max(c, cuts[-1] + 1), n_cells - (n_streams - k)))
        cuts.append(n_cells)
        main = torch.cuda.current_stream(dev)
        aux = getattr(self, "_aux_streams", None) or []
        if aux and getattr(self, "_aux_beside", None) != main.cuda_stream:
            aux = []                       # (picked beside another current stream)
        self._aux_beside = main.cuda_stream
        if len(aux) < n_streams - 1:
            # (streams that really run beside the current one: ops.pick_concurrent_streams - a one-time probe of ~1 ms per candidate)
            aux = aux + ops.pick_concurrent_streams(dev, [main] + aux, n_streams - 1 - len(aux))
        self._aux_streams = aux
        out = torch.empty((n_cells, self.kernel_dim), dtype=torch.float32, device=dev)
        pack = self._cell_pack()
        for st in aux[: n_streams - 1]:
            st.wait_stream(main)
            out.record_stream(st)
        # the side streams' parts are launched first, the current stream's part (part 0) last
        for part in list(range(1, n_streams)) + [0]:
            c0, c1 = cuts[part], cuts[part + 1]
            st = main if part == 0 else aux[part - 1]
            o0, o1 = int(cp[c0]), int(cp[c1])
            with torch.cuda.stream(st):
                sub = [t[o0:o1] for t in (xyz, rgb, center, mean_rgb)]
                ci = None if class_idx is None else class_idx[o0:o1].contiguous()
                co = None if color_idx is None else color_idx[o0:o1].contiguous()
                cfg = self._cell_config(xyz.shape[1], chunk_objects, ci, co)
                cpd = cell_ptr_dev[c0: c1 + 1] - o0 if o0 else cell_ptr_dev[c0: c1 + 1]
                out[c0:c1] = ops.encode_cells(*sub, cp[c0: c1 + 1] - o0, cpd.contiguous(), pack, cfg, False,
                                              ws_tag="encode_cells" if part == 0 else f"encode_cells#{part + 1}")
                if st is not main:
                    for t in (xyz, rgb, center, mean_rgb, cell_ptr_dev):
                        t.record_stream(st)
        for st in aux[: n_streams - 1]:
            main.wait_stream(st)
        return out

    def encode_objects_packed_host(self, xyz, rgb, center, mean_rgb, cell_ptr, cells_per_chunk=2048):
        """The same for HOST tensors (pinned for real overlap): the cells travel in blocks of `cells_per_chunk`; the
        host-to-device copies of block b+1 run on a second HIP stream under the kernels of block b, so the PCIe time
        (6 KB per object) hides behind the encoder instead of preceding it.  Cells are independent, so the result is
        the one of encode_objects_packed on the whole batch."""
        self._check_forward_only()
        dev = self.device
        cp = np.ascontiguousarray(np.asarray(cell_ptr), dtype=np.int32)
        n_cells = cp.shape[0] - 1
        if n_cells <= 0:
            return torch.empty((0, self.embed_dim), dtype=torch.float32, device=dev)
        bounds = list(range(0, n_cells, int(cells_per_chunk))) + [n_cells]
        main = torch.cuda.current_stream(dev)
        if getattr(self, "_copy_stream", None) is None:
            self._copy_stream = ops.concurrent_stream(dev, getattr(self, "_aux_streams", None) or [])
        copy = self._copy_stream

        def stage(b):
            lo, hi = int(cp[bounds[b]]), int(cp[bounds[b + 1]])
            with torch.cuda.stream(copy):
                d = [t[lo:hi].to(dev, non_blocking=True) for t in (xyz, rgb, center, mean_rgb)]
                d.append(torch.from_numpy(cp[bounds[b]: bounds[b + 1] + 1] - lo).to(dev, non_blocking=True))
                ev = torch.cuda.Event()
                ev.record(copy)
            return d, ev, cp[bounds[b]: bounds[b + 1] + 1] - lo

        def run():
            copy.wait_stream(main)
            outs = []
            nxt = stage(0)
            for b in range(len(bounds) - 1):
                d, ev, cpb = nxt
                if b + 2 < len(bounds):
                    nxt = stage(b + 1)
                main.wait_event(ev)
                for t in d:
                    t.record_stream(main)
                outs.append(self.encode_objects_packed(d[0], d[1], d[2], d[3], cpb, d[4], check_overflow=False))
            return outs[0] if len(outs) == 1 else torch.cat(outs)
        return self._with_guard(run)     # one check for all blocks (keeps the copies overlapped)

    def encode_objects(self, objects, object_points):
        """objects: List[List[Object3d]], object_points: List[Batch] (one PyG-style batch per cell)
        -> [B, D] fp32, L2-normalised (models/cell_retrieval.py:77-107).  train() mode: see encode_objects_packed."""
        n_pts = int(getattr(self.args, "pointnet_numpoints", 256))
        n_cells = len(objects)
        if n_cells >= 256 and not self.training and len(object_points) == n_cells:
            # two halves: the kernels of the first half run while the host packs the second (packing a 512-cell batch - two
            # 25 MB concatenations into pinned memory plus the per-cell bookkeeping - takes about as long as encoding it)
            half = n_cells // 2
            return self._with_guard(lambda: torch.cat([
                self._encode_objects_once(objects[a:b], object_points[a:b], n_pts, check_overflow=False)
                for a, b in ((0, half), (half, n_cells))]))
        return self._encode_objects_once(objects, object_points, n_pts)

    def _encode_objects_once(self, objects, object_points, n_pts, check_overflow=True):
        dev = self.device
        # models/object_encoder.py:86-90: without the "color" feature the PointNet++ sees x = 0 - the colours then never travel
        skip_rgb = "color" not in self.args.use_features
        # (the means memo serves evaluation, where the same cell lists come back; training loaders hand over fresh copies)
        xyz, rgb, center, mean_rgb, cell_ptr = pack_cells(objects, object_points, n_pts, staging=self._staging,
                                                          means_cache=None if self.training else self.object_means_cache,
                                                          skip_rgb=skip_rgb, device=dev)
        if rgb is None:
            rgb = torch.zeros_like(xyz)
        to = lambda t: t.to(dev, non_blocking=True)
        # ground-truth embedding ablations (models/object_encoder.py:74-84)
        oe, class_idx, color_idx = self.object_encoder, None, None
        if getattr(self.args, "class_embed", False):
            class_idx = to(torch.tensor([oe.known_classes.get(o.label, 0) for objs in objects for o in objs],
                                        dtype=torch.int32))
        if getattr(self.args, "color_embed", False):
            color_idx = to(torch.tensor([oe.known_colors[o.get_color_text()] for objs in objects for o in objs],
                                        dtype=torch.int32))
        return self.encode_objects_packed(xyz, rgb, center, mean_rgb, cell_ptr, class_idx=class_idx, color_idx=color_idx,
                                          check_overflow=check_overflow)

    def encode_raw_objects(self, objects, generator: np.random.Generator, rotate_degrees: float = None):
        """objects: List[List[Object3d]] with RAW point sets.  The dataloader's per-object transform chain
        (T.FixedPoints(256) [-> T.RandomRotate(rotate_degrees, axis=2), the training transform] -> T.NormalizeScale,
        dataloading/kitti360pose/utils.py:99-109, training/coarse.py:192-199) and the per-object means run on the GPU
        (csrc/small_kernels.hip::k_pack_objects); only the random draws stay on the host."""
        from .data import draw_rotations, flatten_raw_objects
        n_pts = int(getattr(self.args, "pointnet_numpoints", 256))
        raw_xyz, raw_rgb, obj_ptr, sample_idx, cell_ptr = flatten_raw_objects(objects, n_pts, generator)
        dev = self.device
        to = lambda a: torch.from_numpy(a).to(dev, non_blocking=True)
        rot = None if rotate_degrees is None else to(draw_rotations(len(obj_ptr) - 1, rotate_degrees, generator))
        xyz, rgb, center, mean_rgb = ops.pack_objects(to(raw_xyz), to(raw_rgb), to(obj_ptr), to(sample_idx), rot)
        if "color" not in self.args.use_features:
            rgb.zero_()
        return self.encode_objects_packed(xyz, rgb, center, mean_rgb, cell_ptr)

    def encode_scene_cells(self, scene, transform, lo: int = 0, hi: int = None, cell_offset: int = 0, cells_per_call: int = 8192,
                           want_inputs: bool = False):
        """Cells [lo, hi) of a scene.DeviceScene (raw objects resident in HBM) -> [hi - lo, D] fp32, L2-normalised: the
        dataloader chain of evaluation/pipeline.py:303-308 (per object T.FixedPoints -> T.NormalizeScale, the per-object means)
        runs on the GPU (t2p_pack_scene_objects) and feeds t2p_encode_cells without touching the host.  transform:
        pipeline.PerCellTransform (its counter-based draw of global cell `cell_offset + i` is what `transform.for_cell` draws on
        the host: same packed arrays, bit for bit).  want_inputs: also return the packed (xyz, rgb, center, mean_rgb, cell_ptr)
        of the LAST block (parity tests feed them to the oracle)."""
        self._check_forward_only()
        hi = scene.n_cells if hi is None else hi
        if hi <= lo:
            return torch.empty((0, self.embed_dim), dtype=torch.float32, device=self.device)
        want_rgb = "color" in self.args.use_features or bool(getattr(self.args, "class_embed", False))
        class_all, color_all = scene.feature_indices(self)
        kept = []

        def run():
            outs = []
            for a in range(lo, hi, int(cells_per_call)):
                b = min(a + int(cells_per_call), hi)
                (xyz, rgb, center, mean_rgb), cp, ids = scene.pack_cells(transform, a, b, cell_offset, want_rgb=want_rgb)
                if rgb is None:
                    rgb = torch.zeros_like(xyz)      # models/object_encoder.py:86-90: the PointNet++ then sees x = 0
                o0, o1 = int(ids[0]), int(ids[-1]) + 1
                ci = None if class_all is None else class_all[o0:o1].contiguous()
                co = None if color_all is None else color_all[o0:o1].contiguous()
                outs.append(self.encode_objects_packed(xyz, rgb, center, mean_rgb, cp, class_idx=ci, color_idx=co,
                                                       check_overflow=False))
                kept[:] = [(xyz, rgb, center, mean_rgb, cp)]
            return outs[0] if len(outs) == 1 else torch.cat(outs)
        out = self._with_guard(run)
        return (out, kept[0]) if want_inputs else out

    encode_cells = encode_objects  # the name BASELINE.json uses for the same method

    def forward(self):
        raise Exception("Not implemented.")  # as the reference (models/cell_retrieval.py:109-110)

    @property
    def device(self):
        return next(self.lin.parameters()).device

    def get_device(self):
        return self.device
