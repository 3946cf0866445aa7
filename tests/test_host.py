"""CPU tests of the host logic and of the C-ABI library's loadability (no compute calls without a GPU)."""
import os
import sys
import re

import numpy as np
import pytest
import torch

import text2pos_amd as t2p
from text2pos_amd import _lib, data as D, distributed as TD, modules, ops, packing, synthetic as S

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_loads_and_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "t2p.h")).read()
    declared = set(re.findall(r"\b(t2p_[a-z0-9_]+)\s*\(", header))
    assert len(declared) >= 14
    handle = _lib.lib()
    for name in declared:
        assert hasattr(handle, name), f"{name} declared in include/t2p.h but not exported"
    assert declared == set(_lib.SYMBOLS), "ctypes binding and header disagree"
    assert handle.t2p_abi_version() == _lib.ABI_VERSION


def test_workspace_size_queries_need_no_gpu():
    cfg = ops.make_cell_config()
    a = _lib.lib().t2p_encode_cells_workspace_bytes(100, 10, cfg)
    b = _lib.lib().t2p_encode_cells_workspace_bytes(20000, 1000, cfg)
    assert 0 < a < b < 16 << 30
    assert _lib.lib().t2p_encode_text_workspace_bytes(64, 43, 256) > 2 * 43 * 1024 * 4


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libt2p_hip.so")
    with pytest.raises(_lib.T2PError, match="build the HIP extension"):
        _lib.lib()


def test_no_cpu_path():
    with pytest.raises(RuntimeError, match="no CPU path"):
        ops.gemm(torch.zeros(4, 4), torch.zeros(4, 8))
    with pytest.raises(RuntimeError, match="no CPU path"):
        ops.sim_topk(torch.zeros(2, 256), torch.zeros(3, 256), 1)


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "text2pos-cvpr2022_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), f
                assert "/root/reference" not in src, f


# ---- tokeniser / text weights -----------------------------------------------------------------------------------
def test_tokenizer_matches_reference_rules():
    from oracle.model import tokenize as otok
    words = S.known_words()
    vocab = {w: i + 1 for i, w in enumerate(words)}
    texts = ["The pose is north of a gray road.", "Pose, WEST. of a unknownword wall", "a"]
    padded, lens = modules.tokenize(texts, vocab)
    want = otok(texts, vocab)
    assert lens.tolist() == [len(w) for w in want] == [8, 6, 1]
    for i, w in enumerate(want):
        assert padded[i, : len(w)].tolist() == w and (padded[i, len(w):] == 0).all()
    assert padded[1, 4] == 0                                     # unknown word -> 0
    with pytest.raises(RuntimeError):
        modules.tokenize(["", "a"], vocab)


def test_text_weight_packing_layout():
    enc = t2p.LanguageEncoder(S.known_words(), 256, bi_dir=True)
    p = packing.pack_text_weights(enc, "cpu")
    assert p["w_ih"].shape == (2, 256, 1024) and p["w_hh"].shape == (2, 256, 1024) and p["bias"].shape == (2, 1024)
    assert torch.equal(p["w_hh"][1], enc.lstm.weight_hh_l0_reverse.detach().t())
    assert torch.allclose(p["bias"][0], (enc.lstm.bias_ih_l0 + enc.lstm.bias_hh_l0).detach())
    assert p["embedding"].shape == (len(S.known_words()) + 1, 256) and (p["embedding"][0] == 0).all()


# ---- BatchNorm folding / cell weight packing ------------------------------------------------------------------------
def _model():
    import weights as W
    m = t2p.CellRetrievalNetwork(S.LABELS + ["pad"], S.COLOR_NAMES, S.known_words(), S.default_args()).eval()
    W.fill_state_dict(m, 3)
    return m


def test_bn_fold_equals_eval_mode_block():
    m = _model()
    blk = m.object_encoder.pointnet.sa2.point_conv.local_nn[0]
    w, b = packing.fold_linear_bn(blk)
    x = torch.randn(50, 67)
    with torch.no_grad():
        want = blk(x)
    got = torch.relu(x.double() @ w.t() + b).float()
    assert torch.allclose(got, want, atol=1e-5)


def test_cell_weight_pack_shapes_and_edge_split():
    m = _model()
    p = packing.pack_cell_weights(m, "cpu")
    assert [tuple(t.shape) for t in p["sa_w1"]] == [(6, 32), (96, 128), (160, 256)]
    assert [tuple(t.shape) for t in p["sa_w2"]] == [(32, 64), (128, 128), (256, 256)]
    assert tuple(p["ga_w1"].shape) == (288, 512) and tuple(p["ga_w2"].shape) == (512, 1024)
    assert (p["sa_w1"][1][67:] == 0).all() and (p["ga_w1"][259:] == 0).all()      # zero K padding
    assert tuple(p["merge_w"].shape) == (768, 256) and tuple(p["lin1_w"].shape) == (1024, 512)
    # layer 1 of an SA block on [x_j | pos_j - pos_i] == A_j - B_i with the packed tables
    blk = m.object_encoder.pointnet.sa2.point_conv.local_nn[0]
    xj, pj, pi = torch.randn(9, 64), torch.randn(9, 3), torch.randn(9, 3)
    with torch.no_grad():
        want = blk(torch.cat([xj, pj - pi], 1))
    w1, b1 = p["sa_w1"][1], p["sa_b1"][1]
    a = torch.cat([xj, pj, torch.zeros(9, 29)], 1) @ w1 + b1
    bt = pi @ w1[64:67]
    assert torch.allclose(torch.relu(a - bt), want, atol=1e-5)
    # DynamicEdgeConv layer 1 on [x_i | x_j - x_i] == P_i + Q_j
    g1 = m.graph1.nn[0]
    xi, xj = torch.randn(7, 256), torch.randn(7, 256)
    with torch.no_grad():
        want = g1(torch.cat([xi, xj - xi], 1))
    got = torch.relu(xi @ p["g_wp"] + p["g_bp"] + xj @ p["g_wq"])
    assert torch.allclose(got, want, atol=1e-5)


def test_f16x3_split_reconstructs_fp32_weights():
    """w = hi + lo/2048 with hi, lo fp16: the packed planes must reproduce w to ~2^-22 relative, and the product of two
    split operands (dropping lo.lo) must match the fp64 product to ~1e-6 relative -- the basis of the f16x3 MFMA path."""
    g = torch.Generator().manual_seed(0)
    w = (torch.randn(256, 128, generator=g) / 16).float()
    packed = packing.pack_f16x3(w)
    assert packed.shape == (2, 4, 16, 2, 32, 8) and packed.dtype == torch.int16
    v = packed.view(torch.float16).float()
    # undo the register layout: [plane][tile][step][half][lane][e] -> [k = half*128 + 8*step + e][n = 32*tile + lane]
    planes = v.permute(0, 3, 2, 5, 1, 4).reshape(2, 256, 128)
    rec = planes[0].double() + planes[1].double() / 2048.0
    rel = ((rec - w.double()).abs() / w.double().abs().clamp_min(1e-3)).max().item()
    assert rel < 2.0 ** -20
    h = torch.relu(torch.randn(64, 256, generator=g))
    hh = h.half()                                            # hi (round-to-nearest here; the kernel rounds toward zero)
    hl = ((h - hh.float()) * 2048).half()
    approx = (hh.double() @ planes[0].double()) + (hh.double() @ planes[1].double() + hl.double() @ planes[0].double()) / 2048
    exact = h.double() @ w.double()
    assert (approx - exact).abs().max().item() < 1e-6 * exact.abs().max().item() + 1e-7


def test_state_dict_layout_matches_reference_keys(oracle_model):
    m = _model()
    sd = m.state_dict()
    expect = {
        "language_encoder.word_embedding.weight": (43, 256),
        "language_encoder.lstm.weight_ih_l0": (1024, 256),
        "language_encoder.lstm.weight_hh_l0_reverse": (1024, 256),
        "language_encoder.lstm.bias_hh_l0": (1024,),
        "object_encoder.pointnet.sa1.point_conv.local_nn.0.0.weight": (32, 6),
        "object_encoder.pointnet.sa2.point_conv.local_nn.0.0.weight": (128, 67),
        "object_encoder.pointnet.sa3.point_conv.local_nn.1.0.weight": (256, 256),
        "object_encoder.pointnet.sa3.point_conv.local_nn.1.1.running_var": (256,),
        "object_encoder.pointnet.ga.mlp.0.0.weight": (512, 259),
        "object_encoder.pointnet.ga.mlp.1.0.weight": (1024, 512),
        "object_encoder.pointnet.lin1.weight": (512, 1024),
        "object_encoder.pointnet.lin2.weight": (256, 512),
        "object_encoder.pointnet.class_classifier.weight": (22, 256),
        "object_encoder.pointnet.color_classifier.weight": (8, 256),
        "object_encoder.mlp_pointnet.0.0.weight": (256, 256),
        "object_encoder.mlp_merge.0.0.weight": (256, 768),
        "object_encoder.pos_encoder.0.0.weight": (64, 3),
        "object_encoder.color_encoder.1.0.weight": (256, 64),
        "object_encoder.class_embedding.weight": (23, 256),
        "object_encoder.color_embedding.weight": (8, 256),
        "graph1.nn.0.0.weight": (256, 512),
        "graph1.nn.1.1.num_batches_tracked": (),
        "lin.1.0.weight": (256, 256),
    }
    for k, shape in expect.items():
        assert k in sd and tuple(sd[k].shape) == shape, k
    assert set(sd) == set(oracle_model.state_dict())


def test_forward_raises_like_the_reference_and_train_mode_is_refused():
    m = _model()
    with pytest.raises(Exception, match="Not implemented"):
        m.forward()
    m.train()
    with pytest.raises(RuntimeError, match="no CPU path"):   # both branches train on the GPU only
        m.encode_text(["north"])
    objs, pts, _ = _cell(2)
    with pytest.raises(RuntimeError, match="no CPU path"):
        m.encode_objects([objs], [pts])


# ---- input packing -------------------------------------------------------------------------------------------------
def _cell(n, n_pts=256, seed=0):
    xyz, rgb, center, mean_rgb, _ = S.make_cells(seed, n, fixed_n=1)
    objs = [D.Object3d(i, i, np.tile(center[i].astype(np.float64), (3, 1)), np.tile(mean_rgb[i].astype(np.float64), (3, 1)),
                       "box") for i in range(n)]
    pts = D.Batch(x=torch.from_numpy(rgb.reshape(-1, 3)), pos=torch.from_numpy(xyz.reshape(-1, 3)),
                  batch=torch.arange(n).repeat_interleave(n_pts))
    return objs, pts, (xyz, rgb, center, mean_rgb)


def test_pack_cells_roundtrip_and_validation():
    o1, p1, raw1 = _cell(3, seed=1)
    o2, p2, raw2 = _cell(2, seed=2)
    xyz, rgb, center, mean_rgb, ptr = D.pack_cells([o1, o2], [p1, p2], 256)
    assert ptr.tolist() == [0, 3, 5] and xyz.shape == (5, 256, 3)
    assert np.array_equal(xyz[:3].numpy(), raw1[0]) and np.array_equal(rgb[3:].numpy(), raw2[1])
    assert np.allclose(center[:3].numpy(), raw1[2]) and np.allclose(mean_rgb[3:].numpy(), raw2[3])
    z = D.pack_cells([o1], [p1], 256, zero_color=True)
    assert (z[1] == 0).all()
    with pytest.raises(RuntimeError, match="resampled"):
        D.pack_cells([o1[:2]], [p1], 256)
    bad = D.Batch(x=p1.x, pos=p1.pos, batch=p1.batch.flip(0))
    with pytest.raises(RuntimeError, match="contiguous groups"):
        D.pack_cells([o1], [bad], 256)
    with pytest.raises(RuntimeError):
        D.pack_cells([o1, o2], [p1], 256)
    # persistent staging buffers + the per-cell means cache give the same arrays, call after call
    st, cache = D.HostStaging(), D.ObjectMeansCache()
    for _ in range(3):
        got = D.pack_cells([o1, o2], [p1, p2], 256, staging=st, means_cache=cache)
        assert all(torch.equal(a, b) for a, b in zip(got[:4], (xyz, rgb, center, mean_rgb))) and got[4].tolist() == [0, 3, 5]
    assert cache.get(o1) is not None and cache.get(list(o1)) is None            # keyed by the list's identity
    o1[0], keep = o2[0], o1[0]
    assert cache.get(o1) is None                                                    # the list holds another object now
    o1[0] = keep
    assert D.pack_cells([o1], [p1], 256, skip_rgb=True)[1] is None


def test_object_means_equal_the_reference_accessors_bit_for_bit():
    """models/object_encoder.py:121-131: torch.tensor([obj.get_center() ...], dtype=torch.float) - a float64 np.mean per object,
    rounded to fp32.  data.object_means sums all objects of a cell with one reduceat (another summation order) and keeps a row
    only where no order could round differently; the others fall back to np.mean.  Must be bit-identical, also for centred
    clouds (mean ~ 0: cancellation), large world coordinates, single-point objects and fp32 inputs."""
    rng = np.random.default_rng(3)
    n_unsafe = 0
    for trial in range(200):
        objs = []
        for j in range(int(rng.integers(1, 27))):
            m = int(rng.integers(1, 3000))
            kind = trial % 4
            xyz = rng.standard_normal((m, 3)) * (10.0 ** rng.integers(-3, 3))
            if kind == 1:
                xyz += rng.random(3) * 5000.0          # world coordinates of a KITTI-360 scene
            if kind == 2:
                xyz -= xyz.mean(axis=0)                 # mean ~ 1e-17
            rgb = rng.random((m, 3))
            if kind == 3 and j % 3 == 0:
                xyz, rgb = xyz.astype(np.float32), rgb.astype(np.float32)
            objs.append(D.Object3d(j, j, xyz, rgb, "box"))
        center, color = D.object_means(objs)
        want_c = torch.tensor(np.stack([o.get_center() for o in objs]), dtype=torch.float).numpy()
        want_r = torch.tensor(np.stack([o.get_color_rgb() for o in objs]), dtype=torch.float).numpy()
        assert center.dtype == np.float32 and np.array_equal(center, want_c) and np.array_equal(color, want_r), trial


def test_object_means_many_through_the_c_helper_bit_for_bit():
    """data.object_means_many: a whole call's cells in one pass of the CPython helper _t2p_host (csrc/host_ext.c; four
    interleaved partial sums per column, a few threads).  Same contract as above - bit-identical to the reference's per-object
    np.mean rounded to fp32 - including the routes around the helper: float32 arrays, a tuple instead of a list, a subclass
    that overrides an accessor, and the cache it fills."""
    assert D.host_ext() is not None, "the host helper is built by __graft_entry__.build() / build.py"
    rng = np.random.default_rng(11)

    class Shifted(D.Object3d):
        def get_center(self):
            return np.mean(self.xyz, axis=0) + 1.0

    def want(objs):
        return (torch.tensor(np.stack([o.get_center() for o in objs]), dtype=torch.float).numpy(),
                torch.tensor(np.stack([o.get_color_rgb() for o in objs]), dtype=torch.float).numpy())

    cells = []
    for c in range(120):
        objs = []
        for j in range(int(rng.integers(1, 27))):
            m = int(rng.integers(1, 6000))
            xyz = rng.standard_normal((m, 3)) * (10.0 ** rng.integers(-3, 3))
            if c % 4 == 1:
                xyz += rng.random(3) * 5000.0
            if c % 4 == 2:
                xyz -= xyz.mean(axis=0)
            objs.append(D.Object3d(j, j, xyz, rng.random((m, 3)), "box"))
        cells.append(objs)
    cells[7][0].xyz = cells[7][0].xyz.astype(np.float32)              # not float64: the NumPy route takes the call
    cells[9] = tuple(cells[9])                                         # not a list
    cells[11][0] = Shifted(0, 0, cells[11][0].xyz, cells[11][0].rgb, "box")
    cells[13][0].xyz = np.asfortranarray(cells[13][0].xyz)            # not C-contiguous
    for threads in (1, 4):
        for part in (cells[:7], cells[:8], cells[8:10], cells[10:12], cells[12:14], cells[14:]):
            cache = D.ObjectMeansCache()
            got = D.object_means_many(list(part), cache, threads=threads)
            assert len(got) == len(part)
            for objs, (center, color) in zip(part, got):
                wc, wr = want(objs)
                assert center.dtype == np.float32 and np.array_equal(center, wc) and np.array_equal(color, wr)
                hit = cache.get(objs)
                assert hit is not None and hit[0] is not None and np.array_equal(hit[0], wc)
    assert D.object_means_many([], None) == []


def test_batch_object_points_mirrors_reference_pipeline():
    rng = np.random.default_rng(0)
    objs = [D.Object3d(i, i, rng.random((40 + i, 3)) * 5, rng.random((40 + i, 3)), "box") for i in range(4)]
    tf = D.Compose([D.FixedPoints(256, np.random.default_rng(1)), D.NormalizeScale()])
    b = D.batch_object_points(objs, tf)
    assert b.pos.shape == (4 * 256, 3) and b.x.shape == (4 * 256, 3) and b.batch.tolist() == sum([[i] * 256 for i in range(4)], [])
    for i in range(4):
        p = b.pos[i * 256:(i + 1) * 256]
        assert abs(float(p.abs().max()) - 0.999999) < 1e-6 and float(p.mean(0).abs().max()) < 1e-6
        assert len(np.unique(p.numpy(), axis=0)) <= 40 + i            # sampled with replacement


# ---- synthetic generator / sharding ------------------------------------------------------------------------------------
def test_synthetic_is_a_pure_function_of_seed_and_index():
    full = S.make_cells(9, 10)
    part = S.make_cells(9, 10, 4, 7)
    lo, hi = full[4][4], full[4][7]
    for a, b in zip(full[:4], part[:4]):
        assert np.array_equal(a[lo:hi], b)
    assert np.array_equal(part[4], full[4][4:8] - lo)
    sizes = S.cell_sizes(9, 5000)
    assert sizes.min() == 6 and sizes.max() == 26 and abs(sizes.mean() - 16) < 0.3
    assert S.make_texts(9, 5, 6) == S.make_texts(9, 0, 8)[5:6]
    assert all(48 <= len(t.replace(".", "").split()) <= 54 for t in S.make_texts(9, 0, 50))


def test_shard_range_partitions_exactly():
    for n in (0, 1, 7, 8, 12000, 100003):
        for w in (1, 2, 3, 8):
            r = [TD.shard_range(n, k, w) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[i][1] == r[i + 1][0] for i in range(w - 1))
            assert max(b - a for a, b in r) - min(b - a for a, b in r) <= 1


# ---- evaluation metrics ------------------------------------------------------------------------------
def test_evaluation_metrics_match_reference_fixture(golden_dir):
    """hit@k / close-by@k (training/coarse.py:142-163) and recall@k within thresholds (evaluation/utils.py:31-54):
    fixture produced by the reference's own calc_sample_accuracies on reference Pose / Cell objects."""
    from text2pos_amd import evaluation as E
    z = np.load(os.path.join(golden_dir, "eval_metrics.npz"))
    top_k, threshs = [1, 5, 10], [5, 10, 15]
    cells = [D.Cell(i, str(z["cell_scene"][i]), [], 30.0, z["cell_bbox"][i]) for i in range(len(z["cell_bbox"]))]
    cells_dict = {c.id: c for c in cells}
    db_ids = np.array([c.id for c in cells])
    poses = [D.Pose(None, z["pose_w"][q], str(z["pose_cell"][q]), str(z["pose_cell"][q]).split("_")[0]) for q in range(len(z["pose_w"]))]
    centers = np.array([c.get_center()[0:2] for c in cells])
    acc, close, top = E.retrieval_accuracies(z["top_idx"], db_ids, z["pose_cell"], z["pose_w"][:, 0:2], centers, 30.0, top_k)
    assert np.allclose([acc[k] for k in top_k], z["hit"]) and np.allclose([close[k] for k in top_k], z["close"])
    assert list(top[3]) == list(db_ids[z["top_idx"][3]])
    rec = E.localisation_accuracies(poses, [db_ids[r] for r in z["top_idx"]], cells_dict, top_k, threshs)
    assert np.allclose([[rec[k][t] for t in threshs] for k in top_k], z["recall"])
    one = E.calc_sample_accuracies(poses[0], [cells_dict[c] for c in db_ids[z["top_idx"][0]]], 0.5 * np.ones((10, 2)), top_k, threshs)
    assert set(one) == set(top_k) and all(isinstance(v, bool) for d in one.values() for v in d.values())
    assert 0.0 < z["hit"][2] < 1.0 and z["recall"][2, 2] >= z["recall"][0, 0]      # the fixture is not degenerate


def test_fine_state_dict_layout_matches_reference_keys(vocab):
    """SuperGlueMatch keeps the reference's parameter names and shapes (models/superglue_matcher.py:64-83,
    models/superglue.py:53-64,97-129,203-221), so whole checkpoints interchange."""
    import text2pos_amd as t2p
    from oracle import model as OM
    m = t2p.SuperGlueMatch(vocab["classes"], vocab["colors"], vocab["words"],
                           OM.default_args(embed_dim=128, num_layers=6, sinkhorn_iters=50))
    sd = m.state_dict()
    want = {"superglue.bin_score": (), "superglue.final_proj.weight": (128, 128, 1), "superglue.final_proj.bias": (128,),
            "superglue.gnn.layers.11.attn.merge.weight": (128, 128, 1), "superglue.gnn.layers.0.attn.proj.2.bias": (128,),
            "superglue.gnn.layers.5.mlp.0.weight": (256, 256, 1), "superglue.gnn.layers.5.mlp.1.running_var": (256,),
            "superglue.gnn.layers.5.mlp.3.weight": (128, 256, 1), "superglue.kenc.encoder.0.weight": (32, 3, 1),
            "superglue.kenc.encoder.12.bias": (128,), "mlp_offsets.0.weight": (64, 128), "mlp_offsets.2.weight": (2, 64),
            "object_encoder.mlp_merge.0.0.weight": (128, 384), "language_encoder.lstm.weight_hh_l0_reverse": (512, 128)}
    for k, shape in want.items():
        assert k in sd and tuple(sd[k].shape) == shape, k
    assert len([k for k in sd if k.startswith("superglue.gnn.layers.")]) == 12 * (8 + 2 + 5 + 2)
    assert not any(".mlp.2." in k for k in sd if "gnn" in k)  # mlp.2 is the ReLU


def test_run_fine_bookkeeping_matches_per_sample_loop():
    """evaluation.run_fine (batched) == the reference's per-query loop (evaluation/pipeline.py:205-270) written out with
    get_pos_in_cell + calc_sample_accuracies, for a stub model with fixed matches / offsets."""
    from types import SimpleNamespace as NS
    from text2pos_amd import data as D, evaluation as E
    from text2pos_amd.superglue_matcher import get_pos_in_cell
    rng = np.random.default_rng(3)
    cells = {}
    for i in range(12):
        objs = [D.Object3d(j, j, rng.random((20, 3)), rng.random((20, 3)), "box") for j in range(int(rng.integers(3, 20)))]
        x, y = float(rng.integers(0, 5)) * 30.0, float(rng.integers(0, 5)) * 30.0
        c = D.Cell(i, "2013_05_28_drive_0003_sync" if i % 4 else "2013_05_28_drive_0010_sync", objs, 30.0,
                   np.array([x, y, 0.0, x + 30.0, y + 30.0, 10.0]))
        cells[c.id] = c
    ids = list(cells)
    desc = [NS(direction="north", object_color_text="gray", object_label="road")] * 6
    poses = [D.Pose(rng.random(3), rng.random(3) * 150.0, ids[int(rng.integers(0, 12))], "s", desc) for _ in range(7)]
    for p in poses:
        p.cell_id = p.cell_id  # "<scene>_<idx>"
    retrievals = [[ids[int(k)] for k in rng.choice(12, 3, replace=False)] for _ in poses]
    calls = []

    def model(objects, hints, points):
        b = len(objects)
        assert all(len(o) == 16 for o in objects) and all(len(h) == 6 for h in hints) and len(points) == b
        g = np.random.default_rng(100 + len(calls))
        m0 = g.integers(-1, 6, size=(b, 16))
        calls.append(b)
        return NS(matches0=m0, offsets=g.standard_normal((b, 6, 2)) * 0.1)

    tf = D.Compose([D.FixedPoints(32, generator=np.random.default_rng(0)), D.NormalizeScale()])
    top_k, threshs = [1, 3], [5, 10, 15]
    got = E.run_fine(model, poses, cells, retrievals, tf, 16, top_k, threshs, queries_per_call=4)
    assert calls == [12, 9]
    # replay the same stub outputs through the per-sample formulation
    calls.clear()
    outs = [model([[None] * 16] * 12, [[None] * 6] * 12, [None] * 12), model([[None] * 16] * 9, [[None] * 6] * 9, [None] * 9)]
    m0 = np.concatenate([o.matches0 for o in outs]).reshape(7, 3, 16)
    off = np.concatenate([o.offsets for o in outs]).reshape(7, 3, 6, 2)
    acc = [{k: {t: [] for t in threshs} for k in top_k} for _ in range(2)]
    acc_conf = {1: {t: [] for t in threshs}}
    for q, pose in enumerate(poses):
        top_cells = [cells[c] for c in retrievals[q]]
        padded = [list(c.objects)[:16] + [D.Object3d.create_padding()] * max(0, 16 - len(c.objects)) for c in top_cells]
        pm = np.array([get_pos_in_cell(padded[c], m0[q, c], np.zeros((6, 2))) for c in range(3)])
        po = np.array([get_pos_in_cell(padded[c], m0[q, c], off[q, c]) for c in range(3)])
        for a, pos in zip(acc, (pm, po)):
            r = E.calc_sample_accuracies(pose, top_cells, pos, top_k, threshs)
            for k in top_k:
                for t in threshs:
                    a[k][t].append(r[k][t])
        ci = int(np.argmax(np.sum(m0[q] >= 0, axis=1)))
        r = E.calc_sample_accuracies(pose, top_cells[ci: ci + 1], pm[ci: ci + 1], [1], threshs)
        for t in threshs:
            acc_conf[1][t].append(r[1][t])
    for k in top_k:
        for t in threshs:
            assert abs(got[0][k][t] - np.mean(acc[0][k][t])) < 1e-12 and abs(got[1][k][t] - np.mean(acc[1][k][t])) < 1e-12
    for t in threshs:
        assert abs(got[2][1][t] - np.mean(acc_conf[1][t])) < 1e-12


@pytest.mark.parametrize("scene", ["toy", "toy_legacy"])
def test_scene_pickles_written_by_the_reference_load_without_it(golden_dir, scene):
    """tests/golden/scene/*: pickles dumped by the reference's own Cell / Pose / Object3d / Description* classes, under the
    current and the legacy module path (dataloading/__init__.py:8-10); io.load_scenes restores them onto data.py."""
    from text2pos_amd import data as D, io as IO
    assert "datapreparation" not in sys.modules
    g = np.load(os.path.join(golden_dir, "scene_expect.npz"))
    sc = IO.load_scenes(os.path.join(golden_dir, "scene"), [scene])
    assert [c.id for c in sc.all_cells] == list(g["cell_ids"]) and [len(c.objects) for c in sc.all_cells] == list(g["n_objects"])
    assert all(isinstance(c, D.Cell) and isinstance(c.objects[0], D.Object3d) for c in sc.all_cells)
    assert np.allclose(np.array([c.get_center() for c in sc.all_cells]), g["centers"])
    o = sc.all_cells[0].objects[0]
    assert np.allclose(o.get_center(), g["obj0_center"]) and o.get_color_text() == str(g["obj0_color"])
    assert np.allclose(np.array([p.pose_w for p in sc.all_poses]), g["pose_w"]) and [p.cell_id for p in sc.all_poses] == list(g["pose_cell"])
    assert sc.hint_descriptions[0][0] == str(g["hint0"]) and len(sc.hint_descriptions[0]) == 6
    assert np.array_equal(np.array([[d.is_matched for d in p.descriptions] for p in sc.all_poses]), g["matched"])
    assert "pose" in sc.get_known_words() and "pad" in sc.get_known_classes()
    assert sc.texts[0].count("The pose is") == 6


def test_whole_module_checkpoint_converts_without_its_classes(tmp_path):
    """A `torch.save(model)` pickle whose classes are not importable (the reference's models.*, torch_geometric.*) still
    yields its state_dict (evaluation/pipeline.py:313-314 loads such files)."""
    import types
    from text2pos_amd import io as IO
    code = """
import torch.nn as nn
class PointConv(nn.Module):
    def __init__(self, local_nn):
        super().__init__(); self.local_nn = local_nn
class Net(nn.Module):
    def __init__(self):
        super().__init__()
        self.conv = PointConv(nn.Sequential(nn.Sequential(nn.Linear(6, 8), nn.BatchNorm1d(8), nn.ReLU())))
        self.lin = nn.Linear(8, 4)
        self.args = {'embed_dim': 4}
"""
    for name in ("models", "models.cell_retrieval"):
        sys.modules[name] = types.ModuleType(name)
    try:
        exec(code, sys.modules["models.cell_retrieval"].__dict__)
        Net = sys.modules["models.cell_retrieval"].Net
        Net.__module__ = "models.cell_retrieval"
        sys.modules["models.cell_retrieval"].PointConv.__module__ = "models.cell_retrieval"
        torch.manual_seed(3)
        net = Net()
        want = {k: v.clone() for k, v in net.state_dict().items()}
        torch.save(net, tmp_path / "whole.pth")
        torch.save(net.state_dict(), tmp_path / "sd.pth")
    finally:
        del sys.modules["models.cell_retrieval"], sys.modules["models"]
    for f in ("whole.pth", "sd.pth"):
        got = IO.load_reference_checkpoint(str(tmp_path / f))
        assert list(got.keys()) == list(want.keys())
        assert all(torch.equal(got[k], want[k]) for k in want)


def test_product_modules_pickle_the_reference_way(tmp_path):
    """training/coarse.py:323-324 saves checkpoints with `torch.save(model, path)`: the product modules must survive that (their
    caches of ctypes descriptors / streams / pinned staging stay out of the pickle), and the file must read back both as a
    module and - through the loader a reference checkpoint goes through - as a state_dict with the training arguments."""
    import text2pos_amd as t2p
    from text2pos_amd import io as IO, synthetic as S
    from text2pos_amd.data import HostStaging, ObjectMeansCache
    torch.manual_seed(5)
    m = t2p.CellRetrievalNetwork(S.LABELS + ["pad"], S.COLOR_NAMES, S.known_words(), S.default_args(variation=1))
    m._pack = ("stale", object(), lambda: None, True)          # what a used model holds: not picklable
    m.language_encoder._pack = ("stale", {}, lambda: None)
    m._staging.buffer(0, "xyz", 1000)
    m.object_means_cache.put([1], np.zeros((1, 3)), np.zeros((1, 3)))
    torch.save(m, tmp_path / "coarse.pth")
    assert m._pack[0] == "stale" and len(m.object_means_cache.d) == 1      # saving leaves the live model alone
    back = torch.load(tmp_path / "coarse.pth", weights_only=False)
    assert back._pack is None and back.language_encoder._pack is None and back._overflow is None
    assert isinstance(back._staging, HostStaging) and not back._staging.sets[0]
    assert isinstance(back.object_means_cache, ObjectMeansCache) and not back.object_means_cache.d
    sd, args = IO.load_reference_checkpoint(str(tmp_path / "coarse.pth"), return_args=True)
    assert args["variation"] == 1 and args["embed_dim"] == 256 and args["use_features"] == ["class", "color", "position"]
    want = m.state_dict()
    assert list(sd.keys()) == list(want.keys()) and all(torch.equal(sd[k], want[k]) for k in want)
    fine = t2p.SuperGlueMatch(S.LABELS + ["pad"], S.COLOR_NAMES, S.known_words(),
                              S.default_args(embed_dim=128, num_layers=1, sinkhorn_iters=5))
    fine._opack = ("stale", {}, lambda: None)
    torch.save(fine, tmp_path / "fine.pth")
    assert torch.load(tmp_path / "fine.pth", weights_only=False)._opack is None
    assert set(IO.load_reference_checkpoint(str(tmp_path / "fine.pth"))) == set(fine.state_dict())


def test_paired_texts_describe_their_cells():
    """synthetic.make_paired_texts (the training pairs of train_checkpoint.py): text i names six DIFFERENT objects of cell i by the
    attributes make_objects draws for them; a pure function of (seed, global cell index), so any block equals the same rows of
    the whole; every word is in the vocabulary; the benchmark's random queries (make_texts) are untouched."""
    from text2pos_amd import synthetic as S
    from text2pos_amd.data import COLORS
    from text2pos_amd.modules import tokenize
    seed, n = 31, 40
    texts = S.make_paired_texts(seed, n)
    assert texts[7:19] == S.make_paired_texts(seed, n, 7, 19) and len(texts) == n
    sizes = S.cell_sizes(seed, n)
    ptr = np.concatenate([[0], np.cumsum(sizes)])
    shape, color_id, center = S.object_attributes(seed, 0, int(ptr[-1]))
    _, _, got_center, got_mean = S.make_objects(seed, 0, int(ptr[-1]))
    assert np.array_equal(got_center, center.astype(np.float32))
    assert np.abs(got_mean - COLORS[color_id]).max() < 0.05          # (clipped N(0, 0.05^2) noise around the named colour)
    for c, t in enumerate(texts):
        hints = [h.strip() + "." for h in t.split(".") if h.strip()]
        assert len(hints) == 6 and len(set(hints)) >= 1
        own = {S.describe_object(shape[i], color_id[i], center[i]) for i in range(ptr[c], ptr[c + 1])}
        assert set(hints) <= own, (c, set(hints) - own)
    tok, lens = tokenize(texts, {w: i + 1 for i, w in enumerate(S.known_words())})
    assert (tok[np.arange(tok.shape[1])[None, :] < lens[:, None]] > 0).all(), "a word outside the known vocabulary"
    assert set(w for g in S.LABEL_GROUPS for w in g) == set(S.LABELS)
    # a cell with fewer objects than hints repeats them in order
    short = S.make_paired_texts(seed, 4, fixed_n=2)
    assert all(len(set(h.strip() for h in t.split(".") if h.strip())) <= 2 for t in short)


def test_train_checkpoint_collate_is_the_reference_batch_shape():
    """train_checkpoint.collate: what Kitti360CoarseDataset.collate_fn yields (dataloading/kitti360pose/cells.py:98-110) for
    synthetic pairs - and the Object3d accessors return the generator's centre / mean colour (models/object_encoder.py:121-131)."""
    import train_checkpoint as TC
    from text2pos_amd import synthetic as S
    b = TC.collate(77, 20, 3, 8)
    xyz, rgb, center, mean_rgb, ptr = S.make_cells(77, 20, 3, 8)
    assert set(b) == {"texts", "objects", "object_points"} and len(b["texts"]) == len(b["objects"]) == len(b["object_points"]) == 5
    assert b["texts"] == S.make_paired_texts(77, 20, 3, 8)
    for c in range(5):
        a, e = int(ptr[c]), int(ptr[c + 1])
        assert len(b["objects"][c]) == e - a
        assert np.array_equal(b["object_points"][c].pos.numpy().reshape(e - a, 256, 3), xyz[a:e])
        assert np.array_equal(b["object_points"][c].x.numpy().reshape(e - a, 256, 3), rgb[a:e])
        assert np.allclose(b["objects"][c][0].get_center(), center[a]) and np.allclose(b["objects"][c][-1].get_color_rgb(), mean_rgb[e - 1])


def _stack_global_payload(module, name, args=()):
    """A protocol-4 pickle that resolves `module`.`name` with STACK_GLOBAL and calls it (REDUCE) with `args`."""
    import pickle
    import pickletools  # noqa: F401  (documentation of the opcodes used below)
    def short_unicode(s_):
        b = s_.encode()
        return b"\x8c" + bytes([len(b)]) + b
    body = b"\x80\x04" + short_unicode(module) + short_unicode(name) + b"\x93"      # PROTO 4, two strings, STACK_GLOBAL
    body += pickle.dumps(tuple(args), protocol=4)[2:-1]                                # the argument tuple (no PROTO / STOP)
    return body + b"R."                                                               # REDUCE, STOP


def test_unpicklers_refuse_dotted_names_and_foreign_callables(tmp_path):
    """under pickle protocol 4 `find_class` resolves dotted names attribute by attribute, so an allowlist keyed
    on the MODULE alone ("torch.nn.modules.*") reaches any callable ("torch.os.getcwd" inside torch.nn.modules.module).
    Neither loader may execute such a payload; names that merely live in an allowed module's namespace are refused too."""
    import io as _io
    import pickle
    from text2pos_amd import io as IO
    hits = []
    import os as _os
    real = _os.getcwd
    _os.getcwd = lambda: hits.append("called") or real()
    try:
        for module, name in (("torch.nn.modules.module", "torch.os.getcwd"), ("torch._utils", "torch.os.getcwd"),
                             ("torch.nn.modules.module", "OrderedDict.fromkeys")):
            blob = _stack_global_payload(module, name)
            out = IO._CheckpointUnpickler(_io.BytesIO(blob)).load()       # resolves to an inert shell, never the callable
            assert not hits and isinstance(out, (IO._Shell, IO._DictShell)), (module, name, out)
            with pytest.raises(pickle.UnpicklingError):
                IO._SceneUnpickler(_io.BytesIO(blob)).load()
            assert not hits
        # a plain name that an allowed module only IMPORTS (not an nn.Module class defined there) is refused outright
        for module, name in (("torch.nn.modules.module", "warnings"), ("torch.nn.modules.module", "OrderedDict")):
            with pytest.raises(pickle.UnpicklingError):
                IO._CheckpointUnpickler(_io.BytesIO(_stack_global_payload(module, name))).load()
        # ... and torch._utils only yields its tensor / parameter rebuild functions
        out = IO._CheckpointUnpickler(_io.BytesIO(_stack_global_payload("torch._utils", "_rebuild_not_a_thing"))).load()
        assert isinstance(out, IO._Shell)
        assert not IO._allowed_global("torch._utils", "_get_device_index", True)
        assert IO._allowed_global("torch._utils", "_rebuild_tensor_v2", True) and IO._allowed_global("torch._utils", "_rebuild_parameter", True)
    finally:
        _os.getcwd = real


def test_draw_rotations_and_oracle_rotate_z():
    """Host draw of T.RandomRotate(120, axis=2) + its CPU restatement: unit (cos, sin) rows inside the range, z and the
    xy norms untouched, and the matrix convention pos @ [[c, s, 0], [-s, c, 0], [0, 0, 1]]."""
    from oracle import pyg_restated as P
    from text2pos_amd import data as D
    rot = D.draw_rotations(500, 120.0, np.random.default_rng(3))
    assert rot.dtype == np.float32 and rot.shape == (500, 2)
    assert np.abs((rot ** 2).sum(1) - 1).max() < 1e-6 and rot[:, 0].min() >= -0.5 - 1e-6
    assert (rot[:, 1] > 0).any() and (rot[:, 1] < 0).any()
    pos = torch.randn(64, 3)
    out = P.RotateZ(float(rot[0, 0]), float(rot[0, 1]))(P.Data(pos=pos.clone())).pos
    assert torch.equal(out[:, 2], pos[:, 2])
    assert (out[:, :2].norm(dim=1) - pos[:, :2].norm(dim=1)).abs().max() < 1e-5
    q = P.RotateZ(0.0, 1.0)(P.Data(pos=torch.tensor([[1.0, 0.0, 0.0]]))).pos   # +90 degrees: x axis -> (0, 1, 0)
    assert torch.allclose(q, torch.tensor([[0.0, 1.0, 0.0]]))


# ---- on-device input path: host statements (the GPU side is tests/test_gpu_parity.py::test_pack_scene_*) -----------------
def _aten_mean_order(x: np.ndarray) -> np.ndarray:
    """Column sums of an fp32 [P, 3] array in the order csrc/small_kernels.hip::aten_column_sums uses (its comment cites
    ATen's SumKernel.cpp): four interleaved partial sums per column, each folding into a second level every 16 items, the
    P % 4 tail rows on partial sum 0, then ((p0 + p1) + p2) + p3."""
    f = np.float32
    n = x.shape[0]
    size_ilp = n // 4
    out = np.zeros(3, f)
    for c in range(3):
        part = []
        for k in range(4):
            acc = [f(0)] * 4
            i = 0
            while i + 16 <= size_ilp:
                for _ in range(16):
                    acc[0] = f(acc[0] + x[4 * i + k, c])
                    i += 1
                for j in range(1, 4):
                    acc[j] = f(acc[j] + acc[j - 1])
                    acc[j - 1] = f(0)
                    if i & (0xF << (4 * j)):
                        break
            while i < size_ilp:
                acc[0] = f(acc[0] + x[4 * i + k, c])
                i += 1
            for j in range(1, 4):
                acc[0] = f(acc[0] + acc[j])
            if k == 0:
                for r in range(size_ilp * 4, n):
                    acc[0] = f(acc[0] + x[r, c])
            part.append(acc[0])
        out[c] = f(f(f(part[0] + part[1]) + part[2]) + part[3])
    return out


@pytest.mark.parametrize("n_pts", [256, 64, 100, 1024, 1023])
def test_pack_kernel_summation_order_is_this_torch_builds_cpu_mean(n_pts):
    """The pack kernels reproduce T.NormalizeScale bit for bit by summing `pos.mean(dim=-2)` in ATen's CPU order.  That order
    is a property of the torch build: this test pins it where it can run (CPU), so that a torch upgrade that changes it fails
    HERE with a clear message instead of in the GPU bit-equality tests."""
    rng = np.random.default_rng(n_pts)
    for scale in (1.0, 30.0, 1e-3):
        x = (rng.standard_normal((n_pts, 3)) * scale + rng.random(3) * scale).astype(np.float32)
        want = torch.from_numpy(x).mean(dim=-2).numpy()
        got = _aten_mean_order(x) / np.float32(n_pts)
        assert np.array_equal(got, want), "torch's CPU sum order changed: update aten_column_sums in csrc/small_kernels.hip"


def test_counter_based_fixed_points_draw():
    """pipeline.PerCellTransform: for_cell(i)'s host chain draws exactly data.keyed_draws(keys(i, slot), m) - the statement
    the pack kernel executes - and the draw of a cell depends on (seed, cell, slot) only."""
    from text2pos_amd import pipeline as PL
    rng = np.random.default_rng(3)
    tf = PL.PerCellTransform(256, 5)
    objs = [D.Object3d(i, i, rng.random((m, 3)) * 7, rng.random((m, 3)), "box") for i, m in enumerate((1, 25, 300, 4000, 77))]
    sizes = np.array([len(o.xyz) for o in objs])
    idx = D.keyed_draws(tf.keys(41, np.arange(5)), sizes, 256)
    assert idx.shape == (5, 256) and (idx >= 0).all() and (idx < sizes[:, None]).all() and (idx[0] == 0).all()
    b = D.batch_object_points(objs, tf.for_cell(41))
    for i, o in enumerate(objs):
        want = D.NormalizeScale()(D.Data(x=torch.tensor(o.rgb, dtype=torch.float)[idx[i]], pos=torch.tensor(o.xyz, dtype=torch.float)[idx[i]]))
        assert torch.equal(b.pos[i * 256:(i + 1) * 256], want.pos) and torch.equal(b.x[i * 256:(i + 1) * 256], want.x)
    again = D.batch_object_points(objs, tf.for_cell(41))
    assert torch.equal(again.pos, b.pos)
    assert not torch.equal(D.batch_object_points(objs, tf.for_cell(42)).pos, b.pos)
    assert not torch.equal(D.batch_object_points(objs, PL.PerCellTransform(256, 6).for_cell(41)).pos, b.pos)
    # uniformity of the multiply-shift map: every point of a 1000-point object is drawn about 256 * n / 1000 times
    hits = np.bincount(D.keyed_draws(tf.keys(np.arange(4000), 0), np.full(4000, 1000), 256).ravel(), minlength=1000)
    assert abs(hits.mean() - 1024.0) < 1e-9 and hits.std() < 1.25 * np.sqrt(1024.0)


def test_object_sums_helper_one_walk_and_float32_image():
    """_t2p_host.object_sums: .xyz and .rgb of a whole scene in one walk, the float32 upload image beside the sums, on the
    persistent thread pool (used repeatedly, with changing thread counts)."""
    ext = D.host_ext()
    assert ext is not None
    rng = np.random.default_rng(21)
    cells = [[D.Object3d(j, j, rng.standard_normal((m, 3)) * 20, rng.random((m, 3)), "box")
              for j, m in enumerate(rng.integers(1, 3000, size=int(rng.integers(1, 20))))] for _ in range(300)]
    flat = [o for c in cells for o in c]
    n = len(flat)
    rows = np.empty(n, dtype=np.int64)
    assert ext.point_rows(cells, rows) == n and np.array_equal(rows, [len(o.xyz) for o in flat])
    total = int(rows.sum())
    for threads in (1, 3, 8, 32, 2):
        sums, asums, rows2 = np.empty((2, n, 3)), np.empty((2, n, 3)), np.empty((2, n), dtype=np.int64)
        xyz, rgb = np.zeros((total, 3), np.float32), np.zeros((total, 3), np.float32)
        assert ext.object_sums(cells, sums, asums, rows2, threads, xyz, rgb) == n
        assert np.array_equal(rows2[0], rows) and np.array_equal(rows2[1], rows)
        assert np.array_equal(xyz, np.concatenate([o.xyz for o in flat]).astype(np.float32))
        assert np.array_equal(rgb, np.concatenate([o.rgb for o in flat]).astype(np.float32))
        want = np.stack([o.xyz.sum(0) for o in flat])
        assert np.allclose(sums[0], want, rtol=0, atol=1e-9 * np.abs(want).max())
        assert np.allclose(asums[1], np.stack([np.abs(o.rgb).sum(0) for o in flat]), rtol=1e-12)
    flat[5].rgb = flat[5].rgb[:-1] if len(flat[5].rgb) > 1 else np.zeros((2, 3))      # colours and points no longer pair up
    sums, asums, rows2 = np.empty((2, n, 3)), np.empty((2, n, 3)), np.empty((2, n), dtype=np.int64)
    assert ext.object_sums(cells, sums, asums, rows2, 4, np.zeros((total + 8, 3), np.float32), np.zeros((total + 8, 3), np.float32)) == -6
    assert ext.object_sums(cells, sums, asums, rows2, 4) == n          # without the image the two may differ
    with pytest.raises(ValueError):
        ext.object_sums(cells, sums, asums, rows2, 4, np.zeros((3, 3), np.float32), np.zeros((3, 3), np.float32))


def test_device_scene_host_side():
    """scene.DeviceScene on the CPU device (no kernels): the upload image, the exact per-object means, the padded object ids
    of the fine stage and the sampling keys of pack_cells agree with the host chain's statements."""
    from text2pos_amd import pipeline as PL
    from text2pos_amd.scene import DeviceScene
    rng = np.random.default_rng(8)
    cells = []
    for i in range(40):
        objs = [D.Object3d(j, j, rng.standard_normal((m, 3)) * 3 + rng.random(3) * 30, rng.random((m, 3)), "box")
                for j, m in enumerate(rng.integers(25, 600, size=int(rng.integers(1, 22))))]
        cells.append(D.Cell(i, "s", objs, 30.0, np.arange(6.0)))
    np.random.seed(3)
    sc = DeviceScene(cells, "cpu", n_pad=16)
    # the padding objects come from `pad_seed`, not from the process-global np.random (every rank of a process group
    # builds its own scene): same seed, same pads whatever np.random did in between; another seed, other pads
    np.random.seed(99)
    np.random.rand(5)
    again, other = DeviceScene(cells, "cpu", n_pad=16), DeviceScene(cells, "cpu", n_pad=16, pad_seed=1)
    assert torch.equal(again.raw_xyz, sc.raw_xyz) and np.array_equal(again.center64, sc.center64)
    assert not torch.equal(other.raw_xyz[-8 * 16:], sc.raw_xyz[-8 * 16:]) and torch.equal(other.raw_xyz[:-8 * 16], sc.raw_xyz[:-8 * 16])
    flat = [o for c in cells for o in c.objects]
    assert sc.n_cells == 40 and sc.n_objects == len(flat) + 16 and sc.cell_ids[3] == cells[3].id and sc.row_of[cells[3].id] == 3
    assert np.array_equal(sc.raw_xyz.numpy()[: sc.obj_ptr[len(flat)]], np.concatenate([o.xyz for o in flat]).astype(np.float32))
    want_c = torch.tensor(np.stack([o.get_center() for o in flat]), dtype=torch.float).numpy()
    want_k = torch.tensor(np.stack([o.get_color_rgb() for o in flat]), dtype=torch.float).numpy()
    assert np.array_equal(sc.center.numpy()[: len(flat)], want_c) and np.array_equal(sc.color.numpy()[: len(flat)], want_k)
    assert np.abs(sc.center64[: len(flat)] - np.stack([o.get_center() for o in flat])).max() < 1e-12
    assert (sc.rows[len(flat):] == 8).all() and float(sc.raw_xyz[-8 * 16:].abs().max()) <= 0.001       # padding objects
    ids = sc.padded_object_ids(16)
    for i in (0, 7, 39):
        n = len(cells[i].objects)
        lo = int(sc.cell_ptr[i])
        assert ids[i, : min(n, 16)].tolist() == list(range(lo, lo + min(n, 16)))
        assert ids[i, min(n, 16):].tolist() == sc.pad_ids[min(n, 16): 16].tolist()
    with pytest.raises(RuntimeError, match="n_pad"):
        sc.padded_object_ids(17)
    # keys of pack_cells = the keys the host chain of that (global) cell uses, slot by slot
    tf = PL.PerCellTransform(256, 9)
    seen = {}
    sc.pack = lambda obj_ids, keys, n_pts, **kw: seen.update(ids=obj_ids, keys=keys) or (None, None, None, None)
    _, cp, _ = sc.pack_cells(tf, 5, 9, cell_offset=100)
    want = np.concatenate([tf.keys(100 + c, np.arange(len(cells[c].objects))) for c in range(5, 9)])
    assert np.array_equal(seen["keys"], want) and cp[0] == 0 and cp[-1] == len(want)
    assert seen["ids"].tolist() == list(range(int(sc.cell_ptr[5]), int(sc.cell_ptr[9])))


def test_positions_in_cell_is_get_pos_in_cell_for_a_batch():
    from text2pos_amd import evaluation as E
    from text2pos_amd.superglue_matcher import get_pos_in_cell
    rng = np.random.default_rng(1)
    objs = [[D.Object3d(j, j, rng.random((30, 3)), rng.random((30, 3)), "box") for j in range(16)] for _ in range(12)]
    m0 = rng.integers(-1, 6, size=(12, 16))
    m0[3] = -1
    off = rng.standard_normal((12, 6, 2)).astype(np.float32)
    cxy = np.array([[o.get_center()[0:2] for o in row] for row in objs])
    got = E.positions_in_cell(cxy, m0, off)
    for b in range(12):
        assert np.abs(got[b] - get_pos_in_cell(objs[b], m0[b], off[b])).max() < 1e-12
    assert got[3].tolist() == [0.5, 0.5]


def test_every_cells_batch_vector_is_checked():
    """data.pack_cells verifies EVERY cell's batch vector (a sampled check let a permuted vector in any other
    cell through)."""
    rng = np.random.default_rng(0)
    objects, points = [], []
    for c in range(40):
        objs = [D.Object3d(i, i, rng.random((30, 3)), rng.random((30, 3)), "box") for i in range(3 + c % 4)]
        objects.append(objs)
        points.append(D.batch_object_points(objs, D.Compose([D.FixedPoints(256, rng), D.NormalizeScale()])))
    D.pack_cells(objects, points, 256)
    for bad_cell in (1, 17, 38):
        saved = points[bad_cell].batch
        b = saved.clone()
        b[256], b[255] = b[255].item(), b[256].item()          # two entries of neighbouring objects swapped
        points[bad_cell].batch = b
        with pytest.raises(RuntimeError, match=f"cell {bad_cell}: batch vector"):
            D.pack_cells(objects, points, 256)
        points[bad_cell].batch = saved
    points[5].batch = points[5].batch[:-1]
    with pytest.raises(RuntimeError, match="cell 5: batch vector"):
        D.pack_cells(objects, points, 256)


# ---- bench.py launch form ----------------------------------------------------------------------------------------------
def test_bench_self_launch_argv_and_conditions():
    """`python bench.py --gpus N` (the driver's command form) starts its own ranks: the re-exec argv / environment, and when it
    happens (N > 1 or --self-launch, and no launcher environment).  The launched ranks then see RANK / WORLD_SIZE and run."""
    sys.path.insert(0, ROOT)
    import bench
    cmd, env = bench.self_launch_command(8, ["--gpus", "8", "--steps", "20", "--warmup", "3", "--self-launch"], port=29555)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29555"
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "8", "--steps", "20", "--warmup", "3"]          # same arguments, minus the hook
    assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and env["T2P_BENCH_LAUNCHED"] == "self"
    port = int(bench.self_launch_command(2, [])[0][bench.self_launch_command(2, [])[0].index("--master-port") + 1])
    assert 1024 < port < 65536
    assert bench.needs_self_launch(8, environ={}) and bench.needs_self_launch(2, environ={"WORLD_SIZE": "1"})
    assert not bench.needs_self_launch(1, environ={})
    assert bench.needs_self_launch(1, force=True, environ={})
    assert not bench.needs_self_launch(8, environ={"RANK": "3", "WORLD_SIZE": "8"})       # already under a launcher
    assert not bench.needs_self_launch(1, force=True, environ={"RANK": "0", "WORLD_SIZE": "1"})


def test_bench_default_sizes_are_baseline_configs():
    """`python bench.py` = configs[1] (12,000 cells + 1,000 queries on one GPU); `python bench.py --gpus 8` (the driver's scaling
    command) = configs[2] (100,000-cell database + 10,000 queries: 12,500 + 1,250 per rank); explicit --cells / --queries win, so
    `--gpus 8 --cells 12000` keeps the per-GPU share of the 1-GPU line.  The blocks every rank derives from the totals tile them."""
    sys.path.insert(0, ROOT)
    import bench
    from text2pos_amd import distributed as TD
    assert bench.default_sizes(1) == (12000, 1000)
    for n in (2, 4, 8):
        assert bench.default_sizes(n) == (12500, 1250)
    assert 8 * bench.default_sizes(8)[0] == 100000 and 8 * bench.default_sizes(8)[1] == 10000
    assert bench.default_sizes(8, cells=12000) == (12000, 1250) and bench.default_sizes(1, queries=7) == (12000, 7)
    blocks = [TD.shard_range(100000, r, 8) for r in range(8)]
    assert blocks[0] == (0, 12500) and blocks[-1] == (87500, 100000) and all(b[1] == c[0] for b, c in zip(blocks, blocks[1:]))
    uneven = [TD.shard_range(1001, r, 2) for r in range(2)]          # --cells-total 1001 --gpus 2: the padded-shard branch
    assert uneven == [(0, 501), (501, 1001)]


def test_bench_refuses_counter_evidence_taken_on_other_sources(tmp_path, monkeypatch):
    """roofline.traffic / mfma_busy come from committed rocprofv3 --pmc passes (profiles/*_evidence.json) that carry the content
    hashes of the kernel sources they were taken on: unchanged sources pass, a changed csrc/sa3.hip is named as stale."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "profiles"))
    import bench
    import evidence
    doc = {"source_sha256_16": evidence.source_hashes()}
    assert "csrc/sa3.hip" in doc["source_sha256_16"] and "csrc/t2p_common.h" in doc["source_sha256_16"]
    assert bench.stale_sources(doc, None) == [] and bench.stale_sources(doc, ("csrc/sa3.hip",)) == []
    doc["source_sha256_16"]["csrc/sa3.hip"] = "0" * 16
    assert bench.stale_sources(doc, ("csrc/sa3.hip", "csrc/t2p_common.h")) == ["csrc/sa3.hip"]
    assert bench.stale_sources(doc, ("csrc/ga2.hip",)) == []
    name, got = bench.committed_evidence()
    assert (name is None) == (got is None)
    if got is not None:
        assert "k_sa3" in got["kernels"] and got["kernels"]["k_sa3"]["hbm_bytes_per_launch"] > 1e9


def test_bench_plain_multi_gpu_command_launches_ranks_and_propagates_failure():
    """End to end on this GPU-less container: `python bench.py --gpus 2` re-executes under torch.distributed.run, the ranks
    start (rank 0 logs its generation step), fail at the first GPU call, and the plain command exits non-zero with no JSON line."""
    import subprocess
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR", "TORCHELASTIC_RUN_ID"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--cells", "4",
                        "--queries", "4"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        assert r.returncode == 0
        return
    assert r.returncode != 0
    assert "self-launch:" in r.stderr and "--nproc-per-node=2" in r.stderr
    assert "generated " in r.stderr, r.stderr[-2000:]           # the ranks came up and did their host-side work (rank 0 logs it)
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_bench_oracle_full_check_machinery(oracle_model):
    """bench.py --oracle-cells: spawned oracle workers over blocks of the workload, neighbour lists as global
    object rows, per-path summary.  Fed with the oracle's OWN embeddings as the 'HIP path' it must report no difference at all;
    with one cell's embedding nudged and one neighbour list edited, exactly those."""
    sys.path.insert(0, ROOT)
    import bench
    from text2pos_amd import synthetic as S
    seed, n_total, n = 99, 40, 12
    sd = {k: v.clone() for k, v in oracle_model.state_dict().items()}
    xyz, rgb, center, mean_rgb, cp = S.make_cells(seed, n_total, 0, n)
    import tempfile
    with tempfile.NamedTemporaryFile(suffix=".pt", delete=False) as f:
        path = f.name
    torch.save(sd, path)
    try:
        lo, hi, cells, knn = bench._oracle_chunk((path, seed, n_total, 0, n, 4))
    finally:
        os.unlink(path)
    assert (lo, hi) == (0, n) and cells.shape == (n, 256) and knn.shape == (int(cp[-1]), 8)
    cell_of = np.repeat(np.arange(n), np.diff(cp))
    assert (cell_of[np.maximum(knn, 0)] == cell_of[:, None])[knn >= 0].all()          # global rows, inside their own cell
    same = bench.oracle_full_check(seed, n_total, n, sd, {"self": (torch.from_numpy(cells), knn)}, cp, processes=2)
    assert same["oracle_processes"] == 1 and same["self"]["cells_with_a_knn_graph_difference"] == 0      # (12 cells: one block of 64)
    assert same["self"]["cells_beyond_1e-4"] == 0 and same["self"]["max_abs_all_cells"] < 1e-6      # (thread count changes the rounding)
    cells2, knn2 = cells.copy(), knn.copy()
    cells2[3, 0] += 1e-3
    o = int(cp[7])
    knn2[o, 0], knn2[o, 1] = knn2[o, 1], knn2[o, 0]            # same set, other order: not a difference
    knn2[int(cp[9]), 7] = -1 if knn2[int(cp[9]), 7] >= 0 else int(cp[9])
    got = bench.oracle_full_check(seed, n_total, n, sd, {"hip": (torch.from_numpy(cells2), knn2)}, cp, processes=1)["hip"]
    assert got["cells_with_a_knn_graph_difference"] == 1 and got["cells_beyond_1e-4"] == 1
    assert got["cells_beyond_1e-4_without_a_graph_difference"] == 1 and abs(got["max_abs_all_cells"] - 1e-3) < 1e-6
    # chunk-local neighbour rows -> global rows
    g = bench.global_knn(np.array([[0, 1], [1, -1], [0, 1], [1, 0]]), np.array([0, 2, 4]), chunk_objects=2)
    assert g.tolist() == [[0, 1], [1, -1], [2, 3], [3, 2]]


def test_sa_rows_inline_asm_mfmas_keep_their_distance_from_valu_writes():
    """k_sa_rows issues its MFMAs as inline asm (their weight operands live in AGPRs), which hipcc's hazard recogniser does not see as
    MFMAs: gfx950 needs two wait states between a VALU write of a VGPR and an MFMA that reads it as A / B operand, and only the
    source's structure provides them.  Compile the file for gfx950 and check the generated code: no VALU instruction writes a register
    that an MFMA reads within the next two issue slots (an s_nop in between counts for its wait states)."""
    import re
    import subprocess
    import tempfile
    src = os.path.join(ROOT, "text2pos-cvpr2022_amd", "csrc", "sa_rows.hip")
    hipcc = "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "sa_rows.s")
        subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "--cuda-device-only", "-S", src, "-o", out],
                       check=True, capture_output=True)
        text = open(out).read().split("\n")
    st = [i for i, l in enumerate(text) if re.match(r"^_Z\w+:", l) and "k_sa_rows" in l][0]
    fe = next(i for i in range(st, len(text)) if text[i].startswith(".Lfunc_end"))
    ops = [l.strip() for l in text[st + 1: fe] if l.strip() and not l.strip().startswith((";", ".")) and not l.strip().endswith(":")]

    def regs(tok):
        tok = tok.strip()
        m = re.match(r"v\[(\d+):(\d+)\]", tok)
        if m:
            return set(range(int(m.group(1)), int(m.group(2)) + 1))
        m = re.match(r"v(\d+)$", tok)
        return {int(m.group(1))} if m else set()
    n_mfma, close = 0, []
    for i, o in enumerate(ops):
        if not o.startswith("v_mfma"):
            continue
        n_mfma += 1
        parts = [p.strip() for p in o.split(None, 1)[1].split(",")]
        src_regs = regs(parts[1]) | regs(parts[2])
        states = 0
        for j in range(i - 1, max(i - 4, -1), -1):
            p = ops[j]
            if p.startswith("s_nop"):
                states += int(p.split()[1]) + 1
                continue
            if states >= 2:
                break
            if p.startswith("v_") and not p.startswith(("v_mfma", "v_cmp")) and regs(p.split(None, 1)[1].split(",")[0]) & src_regs:
                close.append((states, p, o))
                break
            states += 1
    assert n_mfma == 96, n_mfma
    assert not close, close[:3]


def test_concurrent_stream_selection_on_a_model_of_the_hardware_queues(monkeypatch):
    """ops.concurrent_stream's selection logic on a stand-in for the runtime: four hardware queues handed out round-robin by first use
    (what MI355X / ROCm 7.2 does; the measurement itself runs in the GPU suite).  The stream handed out never shares a queue with the
    current stream or the ones named, avoids streams handed out before while a free queue exists, and the set stays bounded."""
    class FakeStream:
        made = 0

        def __init__(self, device=None):
            self.device = torch.device("cuda:0")
            self.cuda_stream = 1000 + FakeStream.made
            FakeStream.made += 1

    queue = lambda s: 0 if s.cuda_stream == 0 else (s.cuda_stream - 1000 + skew[0]) % 4
    skew = [1]
    null = FakeStream()
    null.cuda_stream = 0
    FakeStream.made = 0
    current = [null]
    monkeypatch.setattr(torch.cuda, "Stream", FakeStream)
    monkeypatch.setattr(torch.cuda, "current_stream", lambda device=None: current[0])
    monkeypatch.setattr(torch.cuda, "is_current_stream_capturing", lambda: False)
    monkeypatch.setattr(torch.cuda, "_sleep", lambda cycles: None, raising=False)
    monkeypatch.setattr(ops, "_measure_overlap", lambda a, b: queue(a) != queue(b))
    for name in ("_overlap_memo", "_stream_sets", "_handed_out"):
        monkeypatch.setattr(ops, name, {})
    monkeypatch.delenv("T2P_NO_STREAM_PROBE", raising=False)
    for skew[0] in (0, 1, 2, 3):                  # (whatever the process used before: the first candidate's queue varies)
        for name in ("_overlap_memo", "_stream_sets", "_handed_out"):
            getattr(ops, name).clear()
        FakeStream.made = 0
        text = ops.concurrent_stream("cuda:0")
        aux = ops.concurrent_stream("cuda:0")
        copy = ops.concurrent_stream("cuda:0", [aux])
        assert len({queue(s) for s in (null, text, aux, copy)}) == 4, skew
        fifth = ops.concurrent_stream("cuda:0", [aux])          # four queues are taken: the named partners still get their own
        assert queue(fifth) not in (queue(null), queue(aux))
        current[0] = text                                       # a caller running under the text stream
        beside_text = ops.concurrent_stream("cuda:0")
        assert queue(beside_text) != queue(text)
        current[0] = null
        for _ in range(40):                                     # a long-lived process: the set of streams stays bounded
            s = ops.concurrent_stream("cuda:0")
            assert queue(s) != queue(null)
        assert FakeStream.made <= ops._MAX_KEPT_STREAMS
