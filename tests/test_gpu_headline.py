"""GPU parity gates (-m gpu) at BASELINE.json's headline size and over the option matrix of the folded inference kernels.

  * the bench's 12,000 cells (seed 20220002) on both arithmetic paths (f16x3 / exact fp32 MFMA): full-size agreement,
    a drawn sample + the extreme cell sizes against the CPU oracle, and the top-10 of 1,000 ENCODED queries over the
    12,000 ENCODED cells bit-exact against the reference's float64 NumPy ranking (training/coarse.py:100-140);
  * the fp16-range guard of the f16x3 path (a checkpoint whose BatchNorm pushes an SA activation past 65504 must raise or
    be recomputed in fp32, never saturate silently);
  * eval-mode options: pointnet_features x use_features (models/object_encoder.py:53-58, :86-90, :137-140);
  * evaluation.pipeline coarse + fine on a synthetic scene against the same pipeline driven by the oracle
    (evaluation/pipeline.py:38-137, :172-279, evaluation/utils.py:31-54).
"""
import os
import warnings

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
TOL = 1e-4
SEED = 20220002          # bench.py: 20220000 + config id


def _dev():
    return torch.device("cuda:0")


def _to_dev(*arrs):
    return [torch.from_numpy(np.ascontiguousarray(a)).to(_dev()) for a in arrs]


@pytest.fixture(autouse=True)
def _restore_torch_threads():
    """_oracle_threads() below changes torch's intra-op pool for the oracle's sake; other test modules (the training-gradient
    comparisons, whose float32 reduction order depends on it) must see the pool they always saw."""
    n = torch.get_num_threads()
    yield
    torch.set_num_threads(n)


def _oracle_threads():
    # the oracle's per-cell eager graph is made of tiny ops: 8-16 intra-op threads are fastest, the box's 256 pathological
    torch.set_num_threads(min(16, os.cpu_count() or 1))


def _sub_batch(arrs, cell_ptr, cells):
    """Objects of the chosen cells, concatenated, with their own CSR pointer."""
    idx = np.concatenate([np.arange(cell_ptr[c], cell_ptr[c + 1]) for c in cells])
    ptr = np.zeros(len(cells) + 1, dtype=np.int32)
    ptr[1:] = np.cumsum([cell_ptr[c + 1] - cell_ptr[c] for c in cells])
    return [a[idx] for a in arrs], ptr


def _calibrate_batchnorm_packed(om, xyz, rgb, center, mean_rgb, cell_ptr):
    """Random weights give nearly identical cell embeddings (pairwise cosine > 0.99999: rankings are near-ties and a 1e-4
    bar on the unit-norm output says little about the layers in front).  Like a trained checkpoint, the test model gets
    BatchNorm running statistics that match its data: one train-mode pass of the oracle over a sample with cumulative
    averaging (momentum=None; the PointNet++ layers average their per-cell statistics), then eval()."""
    bns = [m for m in om.modules() if isinstance(m, torch.nn.BatchNorm1d)]
    for m in bns:
        m.reset_running_stats()
        m.momentum = None
    om.train()
    with torch.no_grad():
        om.encode_objects_packed_grad(xyz, rgb, center, mean_rgb, cell_ptr)
    om.eval()
    for m in bns:
        m.momentum = 0.1


def _global_knn(knn, cell_ptr):
    """t2p_cell_trace.knn_idx rows are local to the library's internal chunk (whole cells, at most DEFAULT_CHUNK_OBJECTS
    objects): add each object's chunk start, keep -1."""
    from text2pos_amd.ops import DEFAULT_CHUNK_OBJECTS
    knn = np.asarray(knn).astype(np.int64)
    chunk0 = np.zeros(knn.shape[0], dtype=np.int64)
    lo = 0
    for c in range(len(cell_ptr) - 1):
        if cell_ptr[c + 1] - lo > DEFAULT_CHUNK_OBJECTS:
            lo = cell_ptr[c]
        chunk0[cell_ptr[c]: cell_ptr[c + 1]] = lo
    return np.where(knn >= 0, knn + chunk0[:, None], -1)


def _knn_flips(knn_a, knn_b, embn64, cell_ptr):
    """Objects whose DynamicEdgeConv neighbour lists differ between two runs, and the evidence that every such difference
    is a near-tie: the squared distances (float64, from one run's normalised object embeddings) of the neighbours that
    appear in only one of the two lists differ by less than 1e-4.  Returns (cells containing such an object, worst gap)."""
    diff = np.flatnonzero((knn_a != knn_b).any(axis=1))
    cell_of = np.repeat(np.arange(len(cell_ptr) - 1), np.diff(cell_ptr))
    worst = 0.0
    for i in diff:
        only = sorted((set(knn_a[i].tolist()) ^ set(knn_b[i].tolist())) - {-1})
        if not only:      # same set, different order: distances tied to the last bit
            continue
        d2 = ((embn64[only] - embn64[i]) ** 2).sum(axis=1)
        worst = max(worst, float(d2.max() - d2.min()))
    return np.unique(cell_of[diff]), worst


@pytest.mark.parametrize("checkpoint", ["golden", "calibrated", "trained"])
def test_headline_config_full_size_parity(oracle_model, vocab, checkpoint, request):
    """BASELINE configs[1] under a parity gate: 12,000 cells + 1,000 queries, embed_dim 256, top-10.  Twice: with the
    golden random weights, and with the same weights after a BatchNorm calibration pass over 48 of the cells (embeddings
    spread out like a trained model's: pairwise cosine well below 1, so the 1e-4 bar bites).  And a third time (round 6) with
    a TRAINED checkpoint: 320 Adam steps of the reference's training loop on this repo's HIP training path, written with
    `torch.save(model)` and read back through io.load_reference_checkpoint (conftest.trained_checkpoint) - weights, BatchNorm
    running estimates and activation ranges of a model that has learnt to retrieve (hit@k on held-out pairs is asserted).  The
    f16x3 call runs with its fp16-range guard armed (`on_overflow="raise"`): the test passing means the guard stayed clear.

    The path is continuous up to the object embeddings and then takes a DISCRETE step: DynamicEdgeConv's kNN graph
    (models/cell_retrieval.py:46-48).  Two evaluations whose object embeddings differ by 1e-5 pick a different 8th
    neighbour wherever the 8th and 9th distances nearly tie (measured: 8 of 191,749 objects between the f16x3 and fp32
    paths), and such a cell's embedding then moves by up to ~5e-2 - in the reference as much as here.  So the gate is:
    object embeddings within 1e-4 everywhere; every neighbour-list difference a proven near-tie; cell embeddings within
    1e-4 for every cell without such a difference; such cells rare."""
    import copy
    import ctypes as C
    import text2pos_amd as t2p
    from oracle import lib as oracle_lib
    from oracle.model import retrieve_topk_f64
    from text2pos_amd import synthetic as S
    n_cells, n_q = 12000, 1000
    xyz, rgb, center, mean_rgb, cell_ptr = S.make_cells(SEED, n_cells)
    assert xyz.shape[0] == int(cell_ptr[-1]) and 180_000 < xyz.shape[0] < 200_000
    _oracle_threads()
    if checkpoint == "calibrated":
        oracle_model = copy.deepcopy(oracle_model)
        n0 = int(cell_ptr[48])
        _calibrate_batchnorm_packed(oracle_model, xyz[:n0], rgb[:n0], center[:n0], mean_rgb[:n0], cell_ptr[:49])
    if checkpoint == "trained":
        from text2pos_amd import io as IO
        path, info = request.getfixturevalue("trained_checkpoint")
        oracle_model = copy.deepcopy(oracle_model)
        oracle_model.load_state_dict(IO.load_reference_checkpoint(path), strict=True)
        hits = info["hit_at_k_held_out_2048_cells"]
        print(f"[headline gate, trained] epoch losses {info.get('epoch_losses')}, held-out hit@k {hits} "
              f"(before training: {info.get('hit_at_k_before_training')})")
        # the training moved the model: the loss fell, and held-out retrieval is far above chance (10 / 2,048 = 0.5 %)
        # (observed over the round's runs: loss 41 -> 15-16, hit@10 6.6-8.0 %; the bars leave room for the run-to-run spread of a
        # training path whose scatter-backward kernels add in arrival order)
        assert info["epoch_losses"][-1] < 0.6 * info["epoch_losses"][0], info["epoch_losses"]
        assert hits[10] > 0.03, hits
    dargs = _to_dev(xyz, rgb, center, mean_rgb)
    models = {}
    for precision in ("f16x3", "fp32"):
        m = t2p.CellRetrievalNetwork(vocab["classes"], vocab["colors"], vocab["words"], S.default_args(), precision=precision)
        m.load_state_dict(oracle_model.state_dict(), strict=True)
        models[precision] = m.to(_dev()).eval()
    hip_model, fp32_model = models["f16x3"], models["fp32"]
    light = ("obj_emb", "knn_idx")                                       # (the full stage trace would be 24 GB here)
    with torch.no_grad():
        x3, tr3 = hip_model.encode_objects_packed(*dargs, cell_ptr, want_trace=light)   # check_overflow: guard stays clear
        f32, tr32 = fp32_model.encode_objects_packed(*dargs, cell_ptr, want_trace=light)
    assert x3.shape == (n_cells, 256) and bool(torch.isfinite(x3).all())
    if checkpoint in ("calibrated", "trained"):
        sample = x3[:512]
        cos = (sample @ sample.T)[torch.triu(torch.ones(512, 512, dtype=torch.bool, device=x3.device), 1)]
        assert cos.max().item() < 0.999 and cos.mean().item() < 0.9, "calibrated embeddings should be spread out"
    # (a) the two arithmetic paths at full size: continuous part everywhere, cell embeddings wherever the graphs agree
    d_obj = (tr3["obj_emb"] - tr32["obj_emb"]).abs().max().item()
    assert d_obj < TOL, f"object embeddings, f16x3 vs fp32 over {xyz.shape[0]} objects: {d_obj:.3e}"
    embn64 = torch.nn.functional.normalize(tr32["obj_emb"].double(), dim=-1).cpu().numpy()
    ka, kb = _global_knn(tr3["knn_idx"].cpu().numpy(), cell_ptr), _global_knn(tr32["knn_idx"].cpu().numpy(), cell_ptr)
    cell_of = np.repeat(np.arange(n_cells), np.diff(cell_ptr))
    assert (cell_of[np.maximum(ka, 0)] == cell_of[:, None])[ka >= 0].all()       # neighbours stay inside their cell
    flip_cells, worst = _knn_flips(ka, kb, embn64, cell_ptr)
    assert worst < 1e-4, f"a neighbour-list difference that is not a near-tie (distance gap {worst:.2e})"
    # observed with the round-to-nearest split: 6 of 12,000 (round 3's toward-zero split: 13); the bar is ~3 x that
    assert len(flip_cells) <= 20, f"{len(flip_cells)} cells with kNN tie flips between the two arithmetic paths"
    print(f"[headline gate, {checkpoint}] f16x3 vs fp32 over 12,000 cells: object embeddings {d_obj:.2e}, {len(flip_cells)} cells with "
          f"a kNN near-tie flip")
    same = np.ones(n_cells, dtype=bool)
    same[flip_cells] = False
    per_cell = (x3 - f32).abs().max(dim=1).values.cpu().numpy()
    d = float(per_cell[same].max())
    assert d < TOL, f"f16x3 vs fp32 over the {int(same.sum())} cells with identical graphs: max|delta| = {d:.3e}"
    # (b) both against the oracle on drawn cells (4,096 with the calibrated checkpoint, 128 with the golden one) + the extreme
    #     sizes the generator produces (n = 6 and n = 26), stage by stage: the sub-batch is its own call (cells do not depend
    #     on their neighbours in a batch: bit-identical)
    sizes = cell_ptr[1:] - cell_ptr[:-1]
    assert sizes.min() == 6 and sizes.max() == 26
    rng = np.random.default_rng(7)
    pick = set(rng.choice(n_cells, {"calibrated": 4096, "trained": 1024}.get(checkpoint, 128), replace=False).tolist())
    pick |= set(np.flatnonzero(sizes == 6)[:8].tolist()) | set(np.flatnonzero(sizes == 26)[:8].tolist())
    pick = sorted(pick)
    sel = torch.tensor(pick)
    sub, sub_ptr = _sub_batch((xyz, rgb, center, mean_rgb), cell_ptr, pick)
    want_parts, emb_parts = [], []
    for lo in range(0, len(pick), 256):          # (the oracle's stage trace of 256 cells is ~0.4 GB: kept per slice only)
        part, part_ptr = _sub_batch((xyz, rgb, center, mean_rgb), cell_ptr, pick[lo: lo + 256])
        otr = []
        want_parts.append(oracle_model.encode_objects_packed(*part, part_ptr, trace=otr))
        emb_parts.append([t for t in otr if "object_embeddings" in t][0]["object_embeddings"])
        del otr
    want, want_emb = torch.cat(want_parts), torch.cat(emb_parts)
    want_embn = np.ascontiguousarray(torch.nn.functional.normalize(want_emb, dim=-1).numpy())
    want_knn = np.zeros((want_embn.shape[0], 8), np.int32)
    fp = lambda a_, t_: a_.ctypes.data_as(C.POINTER(t_))
    oracle_lib().t2p_oracle_knn(fp(want_embn, C.c_float), fp(sub_ptr, C.c_int32), C.c_int32(len(pick)), C.c_int32(256),
                                C.c_int32(8), fp(want_knn, C.c_int32))
    for name, model, full in (("f16x3", hip_model, x3), ("fp32", fp32_model, f32)):
        with torch.no_grad():
            got, gtr = model.encode_objects_packed(*_to_dev(*sub), sub_ptr, want_trace=light)
        assert torch.equal(got, full[sel.to(full.device)]), f"{name}: a cell's embedding depends on its batch"
        e_obj = (gtr["obj_emb"].cpu() - want_emb).abs().max().item()
        assert e_obj < TOL, f"{name} object embeddings vs oracle: {e_obj:.3e}"
        flips, worst = _knn_flips(_global_knn(gtr["knn_idx"].cpu().numpy(), sub_ptr), want_knn.astype(np.int64),
                                  want_embn.astype(np.float64), sub_ptr)
        # observed against the oracle: ~1 cell in 1,000 (round-to-nearest split); the bar is 3 x that
        assert worst < 1e-4 and len(flips) <= max(2, 3 * len(pick) // 1000), (name, worst, len(flips))
        ok = np.ones(len(pick), dtype=bool)
        ok[flips] = False
        err = (got.cpu() - want).abs().max(dim=1).values.numpy()[ok].max()
        assert err < TOL, f"{name} vs oracle on {int(ok.sum())} of the 12,000 cells: {err:.3e}"
        print(f"[headline gate, {checkpoint}] {name} vs oracle on {len(pick)} cells: object embeddings {e_obj:.2e}, "
              f"{len(flips)} cells with a kNN near-tie flip (worst distance gap {worst:.1e}), other cells {err:.2e}")
    # (c) retrieval of the ENCODED queries over the ENCODED cells: bit-exact indices against the reference's float64 NumPy
    texts = S.make_texts(SEED, 0, n_q)
    with torch.no_grad():
        q = hip_model.encode_text(texts)
    idx, score = t2p.retrieve_topk(x3, q, 10)
    widx, wscore = retrieve_topk_f64(x3.cpu().numpy(), q.cpu().numpy(), 10)
    assert np.array_equal(idx.cpu().numpy(), widx)
    assert np.abs(score.cpu().numpy() - wscore).max() < 1e-12
    # the ranking is also stable across the two arithmetic paths: eps = the largest score change any (query, cell) pair
    # sees between them (cells with a tie flip aside); a query whose top-11 gaps all exceed 2 eps, with no flipped cell
    # among either run's top-11, must keep its top-10 exactly
    flipped = torch.from_numpy(~same).to(x3.device)
    eps = (q @ (x3 - f32)[~flipped].T).abs().max().item()
    idx32, _ = t2p.retrieve_topk(f32, q, 11)
    idx11, score11 = t2p.retrieve_topk(x3, q, 11)
    gap = (score11[:, :-1] - score11[:, 1:]).min(dim=1).values
    stable = (gap > 2.02 * eps) & ~flipped[idx11].any(dim=1) & ~flipped[idx32].any(dim=1)
    assert torch.equal(idx11[stable][:, :10], idx32[stable][:, :10])
    if checkpoint != "golden":              # (the golden weights' collapsed embeddings leave few queries with a clear gap)
        assert int(stable.sum()) > n_q // 4, (int(stable.sum()), eps)
    # queries against the oracle too (drawn sample)
    qs = rng.choice(n_q, 64, replace=False)
    want_q = oracle_model.encode_text([texts[i] for i in qs])
    assert (q.cpu()[torch.from_numpy(qs)] - want_q).abs().max().item() < TOL


def test_trained_checkpoint_activation_census(oracle_model, vocab, trained_checkpoint):
    """VERDICT r5 'what's weak' 4: does a TRAINED model's BatchNorm-folded arithmetic stay inside what the fp16 pieces of f16x3
    cover?  (a) The guard word after encoding 2,048 cells of the headline workload with the trained checkpoint: clear.  (b) The
    census behind it (tests/tools/activation_census.py, oracle hooks on every BatchNorm output over 48 cells): every layer's
    largest |activation| below 65504 and every layer's largest positive activation above 2^-7, with the margins written to
    gpurun_out/r06_trained_census.json beside the same table for the random-init + calibrated weights the earlier rounds used."""
    import copy
    import json
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools"))
    import activation_census as AC
    import text2pos_amd as t2p
    from text2pos_amd import io as IO, synthetic as S
    path, info = trained_checkpoint
    sd = IO.load_reference_checkpoint(path)
    xyz, rgb, center, mean_rgb, cell_ptr = S.make_cells(SEED, 12000, 0, 2048)
    m = t2p.CellRetrievalNetwork(vocab["classes"], vocab["colors"], vocab["words"], S.default_args(), on_overflow="raise")
    m.load_state_dict(sd, strict=True)
    m = m.to(_dev()).eval()
    with torch.no_grad():
        out = m.encode_objects_packed(*_to_dev(xyz, rgb, center, mean_rgb), cell_ptr, check_overflow=False)
    code = m.overflow_detected()
    assert code == 0, f"fp16-range guard fired on the trained checkpoint: {code:#x} ({m._GUARD_BITS})"
    assert bool(torch.isfinite(out).all())
    _oracle_threads()
    n0 = int(cell_ptr[48])
    sample = (xyz[:n0], rgb[:n0], center[:n0], mean_rgb[:n0], cell_ptr[:49])
    trained = copy.deepcopy(oracle_model)
    trained.load_state_dict(sd, strict=True)
    calibrated = copy.deepcopy(oracle_model)
    _calibrate_batchnorm_packed(calibrated, *sample)
    report = {"cells_sampled": 48, "guard_word_after_2048_cells": code, "training": info}
    for name, om in (("trained", trained), ("random_init_calibrated", calibrated)):
        rows = AC.census(om.eval(), *sample)
        report[name] = {"summary": AC.summary(rows), "layers": rows}
        for layer, r in rows.items():
            assert r["max_abs"] < AC.FP16_MAX, (name, layer, r)
        print(f"[census, {name}] {json.dumps(AC.summary(rows))}")
    # the low edge is a per-LEVEL test in the library (an SA level whose largest activation is below 2^-7); hold the trained model to it
    for layer, r in report["trained"]["layers"].items():
        if ".sa" in layer or ".ga." in layer:
            assert r["max_pos"] > AC.LOW_EDGE, (layer, r)
    os.makedirs(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out"), exist_ok=True)
    with open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "r06_trained_census.json"), "w") as f:
        json.dump(report, f, indent=1)


def test_oracle_encoded_database_retrieves_the_same_cells(oracle_model, vocab):
    """training/coarse.py:100-140 end to end against an ORACLE-ENCODED database: 2,048 cells + 256 queries of the headline
    workload are encoded on the host by the oracle (BatchNorm-calibrated weights) and ranked by the reference's float64 NumPy
    statements; the same inputs go through the HIP path (encoders + sim_topk).  The two embedding sets agree to 1e-4 except
    for cells with a DynamicEdgeConv near-tie flip (rare, see test_headline_config_full_size_parity); every query whose
    oracle top-11 score gaps all exceed 2e-4 - and whose top-11 holds no such cell on either side - must retrieve exactly
    the oracle's ten cells in the oracle's order.  (test_headline_config_full_size_parity (c) ranks HIP embeddings twice;
    this one compares two independently ENCODED databases.)"""
    import copy
    import text2pos_amd as t2p
    from oracle.model import retrieve_topk_f64
    from text2pos_amd import synthetic as S
    n_cells, n_q = 2048, 256
    xyz, rgb, center, mean_rgb, cell_ptr = S.make_cells(SEED, 12000, 0, n_cells)
    texts = S.make_texts(SEED, 0, n_q)
    _oracle_threads()
    om = copy.deepcopy(oracle_model)
    n0 = int(cell_ptr[48])
    _calibrate_batchnorm_packed(om, xyz[:n0], rgb[:n0], center[:n0], mean_rgb[:n0], cell_ptr[:49])
    want_c = []
    for lo in range(0, n_cells, 64):                      # batch_size 64 cells per call, as eval_epoch does
        hi = min(lo + 64, n_cells)
        a, b = cell_ptr[lo], cell_ptr[hi]
        want_c.append(om.encode_objects_packed(xyz[a:b], rgb[a:b], center[a:b], mean_rgb[a:b], cell_ptr[lo: hi + 1] - a))
    want_c = torch.cat(want_c).numpy()
    want_q = om.encode_text(texts).numpy()
    widx, wscore = retrieve_topk_f64(want_c, want_q, 11)
    m = t2p.CellRetrievalNetwork(vocab["classes"], vocab["colors"], vocab["words"], S.default_args())
    m.load_state_dict(om.state_dict(), strict=True)
    m = m.to(_dev()).eval()
    with torch.no_grad():
        got_c = m.encode_objects_packed(*_to_dev(xyz, rgb, center, mean_rgb), cell_ptr)
        got_q = m.encode_text(texts)
    idx, score = t2p.retrieve_topk(got_c, got_q, 11)
    idx, score = idx.cpu().numpy(), score.cpu().numpy()
    per_cell = np.abs(got_c.cpu().numpy() - want_c).max(axis=1)
    flipped = per_cell >= TOL                              # a discrete kNN step went the other way (near-tie): rare
    assert flipped.sum() <= n_cells // 100, f"{int(flipped.sum())} of {n_cells} cells differ from the oracle by >= 1e-4"
    assert np.abs(got_q.cpu().numpy() - want_q).max() < TOL
    cos = want_c[:512] @ want_c[:512].T
    assert np.triu(cos, 1).max() < 0.9995 and cos[np.triu_indices(512, 1)].mean() < 0.9, "calibrated embeddings should be spread out"
    gap = (wscore[:, :-1] - wscore[:, 1:]).min(axis=1)
    clear = (gap > 2e-4) & ~flipped[widx].any(axis=1) & ~flipped[idx].any(axis=1)
    assert np.array_equal(idx[clear][:, :10], widx[clear][:, :10]), "a query with clear score gaps retrieved other cells"
    assert np.abs(score[clear][:, :10] - wscore[clear][:, :10]).max() < 2e-4
    assert int(clear.sum()) >= n_q // 4, f"only {int(clear.sum())} of {n_q} queries have top-11 gaps above 2e-4"
    print(f"[oracle-encoded DB] {int(clear.sum())} / {n_q} queries with top-11 gaps > 2e-4: identical top-10; "
          f"{int(flipped.sum())} / {n_cells} cells with a kNN near-tie flip; max|cell emb - oracle| over the rest "
          f"{per_cell[~flipped].max():.2e}")


def _hot_checkpoint(oracle_model, factor):
    """The golden weights with the BatchNorm behind SA3's first Linear scaled: its activations grow by `factor`."""
    import copy
    sd = copy.deepcopy(oracle_model.state_dict())
    key = "object_encoder.pointnet.sa3.point_conv.local_nn.0.1."
    sd[key + "weight"] = sd[key + "weight"] * factor
    sd[key + "bias"] = sd[key + "bias"] * factor
    return sd


def test_f16x3_activation_overflow_is_caught(oracle_model, vocab):
    """A checkpoint with a hot channel: relu(A_j - B_i) of SA level 3 exceeds fp16's 65504.  The f16x3 path must not return
    saturated numbers: on_overflow="raise" raises FloatingPointError, on_overflow="fp32" recomputes the call on the exact
    fp32 path (== the fp32 model, bit for bit); a moderately hot checkpoint (activations ~1e3) passes the guard and still
    meets the oracle."""
    import text2pos_amd as t2p
    from oracle import model as OM
    from text2pos_amd import synthetic as S
    xyz, rgb, center, mean_rgb, cell_ptr = S.make_cells(91, 4)
    dargs = _to_dev(xyz, rgb, center, mean_rgb)

    def build(sd, **kw):
        m = t2p.CellRetrievalNetwork(vocab["classes"], vocab["colors"], vocab["words"], S.default_args(), **kw)
        m.load_state_dict(sd, strict=True)
        return m.to(_dev()).eval()

    hot = _hot_checkpoint(oracle_model, 3.0e5)
    om = OM.OracleCellRetrieval(vocab["classes"], vocab["colors"], vocab["words"], OM.default_args()).eval()
    om.load_state_dict(hot, strict=True)
    tr = []
    want = om.encode_objects_packed(xyz, rgb, center, mean_rgb, cell_ptr, trace=tr)
    sa3_in_scale = max(float(d["out"].abs().max()) for t in tr if "sa" in t for d in t["sa"])
    assert np.isfinite(want.numpy()).all() and sa3_in_scale > 1.0      # the oracle itself is fine in fp32
    with torch.no_grad():
        with pytest.raises(FloatingPointError, match="fp16"):
            build(hot).encode_objects_packed(*dargs, cell_ptr)
        exact = build(hot, precision="fp32").encode_objects_packed(*dargs, cell_ptr)
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            redo = build(hot, on_overflow="fp32").encode_objects_packed(*dargs, cell_ptr)
        assert any("fp32" in str(x.message) for x in w)
    assert torch.equal(redo, exact)
    assert (exact.cpu() - want).abs().max().item() < 5e-4      # fp32 at activations ~1e6: the bar scales with them
    # deferred form used by pipelined callers: no synchronisation inside the call, one check afterwards
    m = build(hot)
    with torch.no_grad():
        m.encode_objects_packed(*dargs, cell_ptr, check_overflow=False)
    assert m.overflow_detected() & 4 and m.overflow_detected() == 0     # bit 2 = SA level 3; reading clears
    # warm, not hot: two orders of magnitude below the limit -> guard silent, parity holds
    warm = _hot_checkpoint(oracle_model, 3.0e2)
    om.load_state_dict(warm, strict=True)
    want_w = om.encode_objects_packed(xyz, rgb, center, mean_rgb, cell_ptr)
    with torch.no_grad():
        got_w = build(warm).encode_objects_packed(*dargs, cell_ptr)
    assert (got_w.cpu() - want_w).abs().max().item() < TOL


def test_overflow_guard_fires_through_every_entry_point(oracle_model, vocab):
    """The overflow -> raise / fp32-recompute logic lives in ONE helper (CellRetrievalNetwork._with_guard); every entry point
    goes through it.  With the hot checkpoint: the multi-stream path (2,048 cells: the product default from that size up), the
    pinned-host block pipeline, encode_objects' two pipelined halves (256 cells of Python objects) and the scene path each
    raise under on_overflow="raise" and return exactly the fp32 model's result under on_overflow="fp32"."""
    import text2pos_amd as t2p
    from text2pos_amd import data as D, pipeline as PL, synthetic as S
    from text2pos_amd.scene import DeviceScene

    def build(**kw):
        m = t2p.CellRetrievalNetwork(vocab["classes"], vocab["colors"], vocab["words"], S.default_args(), **kw)
        m.load_state_dict(_hot_checkpoint(oracle_model, 3.0e5), strict=True)
        return m.to(_dev()).eval()
    raising, redoing, exact = build(), build(on_overflow="fp32"), build(precision="fp32")
    xyz, rgb, center, mean_rgb, cell_ptr = S.make_cells(93, 2048)
    dargs = _to_dev(xyz, rgb, center, mean_rgb)
    n256 = int(cell_ptr[256])
    objects, points = [], []
    for c in range(256):
        lo, hi = int(cell_ptr[c]), int(cell_ptr[c + 1])
        objs = [D.Object3d(i, i, xyz[i].astype(np.float64), rgb[i].astype(np.float64), "box") for i in range(lo, hi)]
        objects.append(objs)
        points.append(D.Batch.from_data_list([D.Data(x=torch.from_numpy(rgb[i]), pos=torch.from_numpy(xyz[i])) for i in range(lo, hi)]))
    cells = [D.Cell(c, "s", objs, 30.0, np.arange(6.0)) for c, objs in enumerate(objects)]
    tf = PL.PerCellTransform(256, 1)
    host = [torch.from_numpy(a[:n256]).pin_memory() for a in (xyz, rgb, center, mean_rgb)]
    entries = {
        "multi-stream (2,048 cells)": lambda m: m.encode_objects_packed(*dargs, cell_ptr),
        "single stream": lambda m: m.encode_objects_packed(*[t[:n256] for t in dargs], cell_ptr[:257], streams=1),
        "pinned-host blocks": lambda m: m.encode_objects_packed_host(*host, cell_ptr[:257], cells_per_chunk=100),
        "encode_objects, two halves": lambda m: m.encode_objects(objects, points),
        "scene path": lambda m: m.encode_scene_cells(DeviceScene(cells, _dev()), tf, cells_per_call=100),
    }
    with torch.no_grad():
        for name, call in entries.items():
            with pytest.raises(FloatingPointError, match="0x"):
                call(raising)
            assert raising.overflow_detected() == 0, name                      # the check consumed the sticky word
            with warnings.catch_warnings(record=True) as w:
                warnings.simplefilter("always")
                redo = call(redoing)
            assert any("recomputing" in str(x.message) for x in w), name
            assert redoing.precision == "f16x3", name                          # switched back after the recompute
            assert torch.equal(redo, call(exact)), name


def _cold_checkpoint(oracle_model, s):
    """The golden weights with SA3's hidden layer scaled DOWN by s and the scale taken back behind it: BatchNorm 1's weight and
    bias x s (the hidden activations relu(.) shrink by s), Linear 2's bias and BatchNorm 2's running mean x s, BatchNorm 2's weight
    / s.  Mathematically the same network; its SA3 hidden activations are s times smaller."""
    import copy
    sd = copy.deepcopy(oracle_model.state_dict())
    nn_ = "object_encoder.pointnet.sa3.point_conv.local_nn."
    for k in (nn_ + "0.1.weight", nn_ + "0.1.bias", nn_ + "1.0.bias", nn_ + "1.1.running_mean"):
        sd[k] = sd[k] * s
    sd[nn_ + "1.1.weight"] = sd[nn_ + "1.1.weight"] / s
    return sd


def test_f16x3_cold_activations_are_caught(oracle_model, vocab):
    """LOW side of the fp16-range guard.  A checkpoint whose SA3 hidden activations sit 1e-5 below their usual scale (and whose
    next BatchNorm takes the factor back): the fp16 pieces of such activations fall into fp16's subnormals - the hi piece keeps
    a few bits, the lo piece underflows - and the following rescale would expose the loss.  The guard's bit 7 (an SA level's
    largest hidden activation / output below 2^-7) must fire: raise, or recompute on the exact fp32 path, which meets the oracle.
    A merely cool checkpoint (x 1/16) passes the guard and still meets the oracle at 1e-4 on the f16x3 path."""
    import text2pos_amd as t2p
    from oracle import model as OM
    from text2pos_amd import synthetic as S
    xyz, rgb, center, mean_rgb, cell_ptr = S.make_cells(91, 6)
    dargs = _to_dev(xyz, rgb, center, mean_rgb)

    def build(sd, **kw):
        m = t2p.CellRetrievalNetwork(vocab["classes"], vocab["colors"], vocab["words"], S.default_args(), **kw)
        m.load_state_dict(sd, strict=True)
        return m.to(_dev()).eval()
    om = OM.OracleCellRetrieval(vocab["classes"], vocab["colors"], vocab["words"], OM.default_args()).eval()
    cold = _cold_checkpoint(oracle_model, 1.0e-5)
    om.load_state_dict(cold, strict=True)
    want = om.encode_objects_packed(xyz, rgb, center, mean_rgb, cell_ptr)
    base = oracle_model.encode_objects_packed(xyz, rgb, center, mean_rgb, cell_ptr)
    assert (want - base).abs().max().item() < 1e-5            # the same network as the golden one, in fp32
    with torch.no_grad():
        with pytest.raises(FloatingPointError, match="0x80"):
            build(cold).encode_objects_packed(*dargs, cell_ptr)
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            redo = build(cold, on_overflow="fp32").encode_objects_packed(*dargs, cell_ptr)
        assert any("fp32" in str(x.message) for x in w)
        assert torch.equal(redo, build(cold, precision="fp32").encode_objects_packed(*dargs, cell_ptr))
        assert (redo.cpu() - want).abs().max().item() < TOL
        cool = _cold_checkpoint(oracle_model, 1.0 / 16.0)
        om.load_state_dict(cool, strict=True)
        got = build(cool).encode_objects_packed(*dargs, cell_ptr)            # guard silent
        assert (got.cpu() - om.encode_objects_packed(xyz, rgb, center, mean_rgb, cell_ptr)).abs().max().item() < TOL


def test_f16x3_weight_out_of_range_is_refused(oracle_model, vocab):
    """A folded weight past fp16's range cannot enter the f16x3 images: packing raises and names precision="fp32"."""
    import copy
    import text2pos_amd as t2p
    from text2pos_amd import packing, synthetic as S
    sd = copy.deepcopy(oracle_model.state_dict())
    sd["object_encoder.pointnet.ga.mlp.1.0.weight"][3, 5] = 1.0e9
    m = t2p.CellRetrievalNetwork(vocab["classes"], vocab["colors"], vocab["words"], S.default_args())
    m.load_state_dict(sd, strict=True)
    m = m.to(_dev()).eval()
    args = _to_dev(*S.make_objects(5, 0, 6))
    with torch.no_grad(), pytest.raises(packing.Fp16RangeError, match="fp32"):
        m.encode_objects_packed(*args, np.array([0, 6], dtype=np.int32))


@pytest.mark.parametrize("where", ["xyz", "rgb"])
def test_f16x3_nan_input_is_caught(hip_model, where):
    """The f16x3 SA kernels aggregate with a float max (ds_max_f32 from -inf), which DROPS NaN operands where the reference's
    ReLU + scatter-max propagate them: a NaN among the input points / colours would come out as a finite embedding.  The
    guard word carries the inputs' largest |bit pattern| (a NaN is the largest of all): the call must raise."""
    from text2pos_amd import synthetic as S
    xyz, rgb, center, mean_rgb = S.make_objects(5, 0, 12)
    bad = {"xyz": xyz, "rgb": rgb}[where].copy()
    bad[7, 100, 1] = np.float32(np.nan) if where == "xyz" else -np.float32(np.nan)
    args = _to_dev(bad if where == "xyz" else xyz, bad if where == "rgb" else rgb, center, mean_rgb)
    ptr = np.array([0, 5, 12], dtype=np.int32)
    with torch.no_grad(), pytest.raises(FloatingPointError, match="0x4"):
        hip_model.encode_objects_packed(*args, ptr)
    with torch.no_grad():                                     # the sticky word was cleared: the next (clean) call is fine
        out = hip_model.encode_objects_packed(*_to_dev(xyz, rgb, center, mean_rgb), ptr)
    assert bool(torch.isfinite(out).all())


_FEATURE_SETS = {"all": ["class", "color", "position"], "class+position": ["class", "position"], "color": ["color"],
                 "class": ["class"]}


@pytest.mark.parametrize("use", list(_FEATURE_SETS))
@pytest.mark.parametrize("pointnet_features", [0, 1, 2])
def test_eval_option_matrix_vs_oracle(vocab, pointnet_features, use):
    """The folded inference kernels under every --pointnet_features x --use_features combination: PointNet++ feature
    vectors (features0 [1024] / features1 [512] / features2 [256]), ObjectEncoder output and cell embeddings against the
    oracle.  Without "color" the PointNet++ sees zeroed colours (models/object_encoder.py:86-90); with one feature the
    merge MLP is skipped (:137-140); without "class" the PointNet++ output is not used at all.
    (The reference reads `.features2` whatever the flag (models/object_encoder.py:93), so its own forward only runs with
    pointnet_features = 2; 0 / 1 follow the constructor's widths, :53-58, as the oracle does.)"""
    import weights as W
    import text2pos_amd as t2p
    from oracle import model as OM
    from text2pos_amd import synthetic as S
    kw = dict(use_features=_FEATURE_SETS[use], pointnet_features=pointnet_features)
    om = OM.OracleCellRetrieval(vocab["classes"], vocab["colors"], vocab["words"], OM.default_args(**kw)).eval()
    W.fill_state_dict(om, 17)
    hm = t2p.CellRetrievalNetwork(vocab["classes"], vocab["colors"], vocab["words"], S.default_args(**kw))
    hm.load_state_dict(om.state_dict(), strict=True)
    hm = hm.to(_dev()).eval()
    xyz, rgb, center, mean_rgb, cell_ptr = S.make_cells(300 + pointnet_features, 3)
    tr = []
    want = om.encode_objects_packed(xyz, rgb.copy(), center, mean_rgb, cell_ptr, trace=tr)
    with torch.no_grad():
        got, gtr = hm.encode_objects_packed(*_to_dev(xyz, rgb, center, mean_rgb), cell_ptr, want_trace=True)
    assert (got.cpu() - want).abs().max().item() < TOL
    emb = [d for d in tr if "object_embeddings" in d][0]["object_embeddings"]
    assert (gtr["obj_emb"].cpu() - emb).abs().max().item() < TOL
    if "class" in _FEATURE_SETS[use]:
        pn = [d for d in tr if "sa" in d]
        for name in ("features0", "features1", "features2"):
            ref = torch.cat([d[name] for d in pn])
            assert (gtr[name].cpu() - ref).abs().max().item() < TOL, name
    # the reference-signature entry point zeroes the colours itself when "color" is not a feature
    if "color" not in _FEATURE_SETS[use]:
        from text2pos_amd import data as D
        objects, points = [], []
        for c in range(len(cell_ptr) - 1):
            lo, hi = int(cell_ptr[c]), int(cell_ptr[c + 1])
            objects.append([D.Object3d(i, i, np.tile(center[i].astype(np.float64), (2, 1)),
                                       np.tile(mean_rgb[i].astype(np.float64), (2, 1)), "box") for i in range(lo, hi)])
            points.append(D.Batch(x=torch.from_numpy(rgb[lo:hi].reshape(-1, 3).copy()),
                                  pos=torch.from_numpy(xyz[lo:hi].reshape(-1, 3).copy()),
                                  batch=torch.arange(hi - lo).repeat_interleave(256)))
        with torch.no_grad():
            assert torch.equal(hm.encode_objects(objects, points), got)


# ---- end-to-end pipeline against the oracle --------------------------------------------------------------------------
def _toy_scene(n_cells=24, n_poses=24, seed=11):
    from text2pos_amd import data as D, synthetic as S
    rng = np.random.default_rng(seed)
    cells, poses = [], []
    dirs = ["north", "south", "east", "west", "on-top"]
    for i in range(n_cells):
        objs = []
        for j in range(int(rng.integers(6, 20))):
            c = rng.random(3) * np.array([1.0, 1.0, 0.3])
            n = int(rng.integers(30, 400))
            col = np.clip(rng.random(3), 0, 1)
            objs.append(D.Object3d(j, 1000 * i + j, c + 0.05 * rng.standard_normal((n, 3)),
                                   np.clip(col + 0.05 * rng.standard_normal((n, 3)), 0, 1),
                                   S.LABELS[int(rng.integers(0, len(S.LABELS)))]))
        x, y = 30.0 * (i % 6), 30.0 * (i // 6)
        cells.append(D.Cell(i, "toy1", objs, 30.0, np.array([x, y, 0.0, x + 30.0, y + 30.0, 10.0])))
    for q in range(n_poses):
        c = cells[int(rng.integers(0, n_cells))]
        descs = [D.DescriptionBestCell(dirs[int(rng.integers(0, 5))], o.get_color_text(), o.label, o.id, True)
                 for o in [c.objects[int(k)] for k in rng.choice(len(c.objects), 6, replace=False)]]   # distinct objects:
        # a repeated hint sentence would tie two columns of the matching matrix exactly
        poses.append(D.Pose(rng.random(3), c.bbox_w[0:3] + rng.random(3) * 30.0, c.id, "toy1", descs))
    return cells, poses


class _OracleCoarse:
    """The oracle behind the reference's model interface (encode_objects / encode_text), host tensors."""

    def __init__(self, om):
        self.om = om

    def encode_objects(self, objects, object_points):
        from text2pos_amd.data import pack_cells
        xyz, rgb, center, mean_rgb, cell_ptr = pack_cells(objects, object_points, 256)
        return self.om.encode_objects_packed(xyz.numpy(), rgb.numpy(), center.numpy(), mean_rgb.numpy(), cell_ptr)

    def encode_text(self, texts):
        return self.om.encode_text(texts)


class _OracleFine:
    def __init__(self, orc):
        self.orc = orc

    def __call__(self, objects, hints, object_points):
        from text2pos_amd.data import pack_cells
        from text2pos_amd.superglue_matcher import MatchOutputs
        xyz, rgb, center, mean_rgb, cell_ptr = pack_cells(objects, object_points, 256)
        out = self.orc.forward_packed(xyz.numpy(), rgb.numpy(), center.numpy(), mean_rgb.numpy(), cell_ptr, hints)
        return MatchOutputs(**out)


def _calibrate_batchnorm(om, cells, seed):
    """_calibrate_batchnorm_packed over the cells of a scene (resampled with their own seeded T.FixedPoints draw)."""
    from text2pos_amd import data as D, pipeline as PL
    from text2pos_amd.data import pack_cells
    tf = PL.default_transform(256, seed)
    objs = [list(c.objects) for c in cells]
    xyz, rgb, center, mean_rgb, cell_ptr = pack_cells(objs, [D.batch_object_points(o, tf) for o in objs], 256)
    _calibrate_batchnorm_packed(om, xyz.numpy(), rgb.numpy(), center.numpy(), mean_rgb.numpy(), cell_ptr)


def _calibrated_fine_pair(vocab, cells, pad):
    """(product SuperGlueMatch on the GPU, oracle) with the object encoder's BatchNorm calibrated on the scene (distinct
    object encodings) and final_proj scaled up so that the matcher's scores peak: with the plain random weights every score
    sits at the uniform 1/16 and nothing is ever matched - a comparison of all -1 lists would prove little."""
    from conftest import make_fine_pair
    from text2pos_amd import data as D, pipeline as PL
    from text2pos_amd.data import pack_cells
    prod, orc = make_fine_pair(vocab, 2, 14)
    bns = [m for m in orc.object_encoder.modules() if isinstance(m, torch.nn.BatchNorm1d)]
    for m in bns:
        m.reset_running_stats()
        m.momentum = None
    orc.object_encoder.train()
    objs = [(list(c.objects)[:pad] + [D.Object3d.create_padding()] * pad)[:pad] for c in cells]
    tf = PL.default_transform(256, 98)
    xyz, rgb, center, mean_rgb, cell_ptr = pack_cells(objs, [D.batch_object_points(o, tf) for o in objs], 256)
    orc.forward_packed(xyz.numpy(), rgb.numpy(), center.numpy(), mean_rgb.numpy(), cell_ptr,
                       [["The pose is north of a gray road."] * 6] * len(objs))       # (no_grad; train-mode statistics only)
    orc.eval()
    for m in bns:
        m.momentum = 0.1
    with torch.no_grad():
        for t in (orc.superglue.final_proj.weight, orc.superglue.final_proj.bias,
                  prod.superglue.final_proj.weight, prod.superglue.final_proj.bias):
            t.mul_(4.0)
    prod.object_encoder.load_state_dict(orc.object_encoder.state_dict(), strict=True)
    return prod.to(_dev()).eval(), orc


def test_pipeline_end_to_end_vs_oracle(tmp_path, oracle_model, vocab):
    """pipeline.run_coarse + evaluation.run_fine (io -> coarse retrieval -> fine localisation -> accuracy tables) on the
    HIP path against the same pipeline driven by the oracle: identical retrieval lists, bit-exact matches, poses within
    1e-4, equal accuracy tables.  The T.FixedPoints draws are seeded per stage, so both runs see the same resampled
    objects.  Queries whose ORACLE ranking has a float64 score gap below 1e-3 inside the top-(k+1) are left out of the
    scene (a gap below the 1e-4 embedding tolerance would make "identical lists" a coin toss, not a parity statement)."""
    import copy
    import text2pos_amd as t2p
    from oracle.model import retrieve_topk_f64
    from text2pos_amd import data as D, evaluation as E, io as IO, pipeline as PL, synthetic as S
    top_k, threshs, pad, n_keep = (1, 2, 3), (5, 10, 15), 16, 24
    np.random.seed(7)      # Object3d.create_padding draws from np.random: the calibration data must not depend on test order
    cells, poses = _toy_scene(n_poses=60)
    _oracle_threads()
    prod_fine, orc_fine = _calibrated_fine_pair(vocab, cells, pad)
    om = copy.deepcopy(oracle_model)
    _calibrate_batchnorm(om, cells, seed=99)
    hip = t2p.CellRetrievalNetwork(vocab["classes"], vocab["colors"], vocab["words"], S.default_args())
    hip.load_state_dict(om.state_dict(), strict=True)
    hip = hip.to(_dev()).eval()

    # ---- oracle coarse stage (evaluation/pipeline.py:60-137): encode, NumPy float64 ranking; pick the scene's queries
    IO.save_scene(str(tmp_path / "all"), "toy1", cells, poses)
    sc_all = IO.load_scenes(str(tmp_path / "all"), ["toy1"])
    oc = _OracleCoarse(om)
    tf = PL.default_transform(256, 1)
    enc = []
    for lo in range(0, len(sc_all.all_cells), 64):
        objs = [list(c.objects) for c in sc_all.all_cells[lo: lo + 64]]
        enc.append(oc.encode_objects(objs, [D.batch_object_points(o, tf) for o in objs]))
    cell_enc, text_enc = torch.cat(enc).numpy(), oc.encode_text(sc_all.texts).numpy()
    cos = (cell_enc @ cell_enc.T)[np.triu_indices(len(cells), 1)]
    assert cos.max() < 0.9, "calibrated embeddings should be well separated"
    widx, wscore = retrieve_topk_f64(cell_enc, text_enc, max(top_k) + 1)
    clear = np.flatnonzero((wscore[:, :-1] - wscore[:, 1:]).min(axis=1) > 1e-3)
    assert len(clear) >= n_keep, f"only {len(clear)} of {len(poses)} queries have unambiguous rankings"
    keep = clear[:n_keep]
    widx = widx[keep]
    IO.save_scene(str(tmp_path / "kept"), "toy1", cells, [poses[i] for i in keep])
    sc = IO.load_scenes(str(tmp_path / "kept"), ["toy1"])
    assert len(sc.all_poses) == n_keep

    # ---- HIP path: the product's own drivers, stage by stage so that each stage gets its own seeded draw
    retr, acc = PL.run_coarse(hip, sc, PL.default_transform(256, 1), top_k, threshs)
    seen = {}

    class Spy:                                  # records what the fine model returned per (query, candidate)
        def __call__(self, objects, hints, points):
            out = prod_fine(objects, hints, points)
            seen.setdefault("m0", []).append(out.matches0.cpu().numpy())
            seen.setdefault("off", []).append(out.offsets.cpu().numpy())
            seen.setdefault("P", []).append(out.P.cpu().numpy())
            return out
    np.random.seed(2022)   # Object3d.create_padding draws its 8 points from np.random (imports.py:81): same pads in both runs
    fine_tables = E.run_fine(Spy(), sc.all_poses, sc.cells_dict, retr, PL.default_transform(256, 2), pad, list(top_k),
                             list(threshs), queries_per_call=8)

    # ---- comparison
    db_ids = [c.id for c in sc.all_cells]
    want_retr = [[db_ids[j] for j in row[:max(top_k)]] for row in widx]
    assert retr == want_retr                                            # identical retrieval lists
    centers = np.array([c.get_center()[0:2] for c in sc.all_cells])
    w_hit, w_close, _ = E.retrieval_accuracies(widx[:, :max(top_k)], db_ids, [p.cell_id for p in sc.all_poses],
                                               np.array([p.pose_w for p in sc.all_poses]), centers, 30.0, list(top_k))
    assert acc["hit"] == w_hit and acc["close"] == w_close
    assert acc["localisation"] == E.localisation_accuracies(sc.all_poses, want_retr, sc.cells_dict, list(top_k), list(threshs))
    oseen = {}

    class OSpy:
        def __call__(self, objects, hints, points):
            out = _OracleFine(orc_fine)(objects, hints, points)
            oseen.setdefault("m0", []).append(out.matches0.numpy())
            oseen.setdefault("off", []).append(out.offsets.numpy())
            oseen.setdefault("P", []).append(out.P.numpy())
            return out
    np.random.seed(2022)
    want_tables = E.run_fine(OSpy(), sc.all_poses, sc.cells_dict, want_retr, PL.default_transform(256, 2), pad, list(top_k),
                             list(threshs), queries_per_call=8)
    m0, om0 = np.concatenate(seen["m0"]), np.concatenate(oseen["m0"])
    if not np.array_equal(m0, om0):
        bad = np.argwhere(m0 != om0)
        P_h, P_o = np.concatenate(seen["P"]), np.concatenate(oseen["P"])
        print("DIFF", len(bad), "max|dP|", np.abs(P_h - P_o).max())
        for b, o in bad[:10]:
            print(b, o, "hip", m0[b, o], "oracle", om0[b, o], "P row hip", np.round(P_h[b, o], 4), "P row oracle", np.round(P_o[b, o], 4))
    assert np.array_equal(m0, om0) and (m0 >= 0).any() and (m0 < 0).any()       # matches bit-exact, both kinds present
    assert np.abs(np.concatenate(seen["off"]) - np.concatenate(oseen["off"])).max() < TOL
    assert np.abs(np.concatenate(seen["P"]) - np.concatenate(oseen["P"])).max() < TOL
    # poses in the cell from matches + offsets (models/superglue_matcher.py:139-161), per (query, candidate)
    from text2pos_amd.superglue_matcher import get_pos_in_cell
    k = max(top_k)
    pad_objs = lambda cid: (list(sc.cells_dict[cid].objects)[:pad] + [D.Object3d.create_padding()] * pad)[:pad]
    # (a padding object's centre is ~0.0005: which of them stands in here moves the compared positions by < 1e-3 * 1e-3)
    off, ooff = np.concatenate(seen["off"]), np.concatenate(oseen["off"])
    for i in range(m0.shape[0]):
        objs = pad_objs(retr[i // k][i % k])
        assert np.abs(get_pos_in_cell(objs, m0[i], off[i]) - get_pos_in_cell(objs, om0[i], ooff[i])).max() < TOL
    assert fine_tables == want_tables                                   # the three accuracy tables (mean / offset / conf)


# ---- equivalent execution plans ----------------------------------------------------------------------------------------
def test_pipeline_config4_scale_vs_oracle(tmp_path, oracle_model, vocab):
    """BASELINE configs[4] at the size one GPU and a CPU oracle can carry: `pipeline.evaluate` (coarse retrieval + fine
    localisation + accuracy tables, evaluation/pipeline.py:282-342) over a synthetic scene of 2,048 cells and 1,024 poses.
      * coarse: the database is ALSO encoded by the oracle on the host (same per-cell T.FixedPoints draws) and ranked by the
        reference's float64 NumPy statements; cell embeddings agree to 1e-4 except for cells with a DynamicEdgeConv near-tie
        flip (rare), and every query whose oracle top-(k+1) score gaps exceed 2e-4 - with no such cell in either list - gets
        exactly the oracle's retrieval list from the pipeline (>= 512 such queries);
      * fine: what `evaluate` fed the fine model for its first 64 poses (x top-5 candidates = 320 samples) goes through the
        oracle's SuperGlueMatch as well: P and offsets within 1e-4, matches identical wherever the oracle's decision has a
        margin above 1e-3 (and on > 99 % of the entries overall);
      * the accuracy tables are recomputed from the retrieval lists / per-sample outputs with the metric functions the
        reference fixture pins.
    Prints the pipeline's wall time."""
    import copy
    import time
    import text2pos_amd as t2p
    from oracle.model import retrieve_topk_f64
    from text2pos_amd import data as D, evaluation as E, io as IO, pipeline as PL, synthetic as S
    n_cells, n_poses, top_k, threshs, pad = 2048, 1024, (1, 3, 5), (5, 10, 15), 16
    kmax = max(top_k)
    np.random.seed(7)
    cells, poses = _toy_scene(n_cells=n_cells, n_poses=n_poses, seed=12)
    _oracle_threads()
    prod_fine, orc_fine = _calibrated_fine_pair(vocab, cells[:64], pad)
    om = copy.deepcopy(oracle_model)
    _calibrate_batchnorm(om, cells[:48], seed=99)
    hip = t2p.CellRetrievalNetwork(vocab["classes"], vocab["colors"], vocab["words"], S.default_args())
    hip.load_state_dict(om.state_dict(), strict=True)
    hip = hip.to(_dev()).eval()
    IO.save_scene(str(tmp_path / "big"), "toy1", cells, poses)
    sc = IO.load_scenes(str(tmp_path / "big"), ["toy1"])
    assert len(sc.all_cells) == n_cells and len(sc.all_poses) == n_poses
    tf = PL.PerCellTransform(256, 5)

    # ---- the product pipeline (input side on the GPU: scene.DeviceScene), with a spy on the fine model's first call
    seen = {}
    n_spy = 64 * kmax

    class Spy(torch.nn.Module):
        """Stands in for SuperGlueMatch on the packed entry point the on-device pipeline calls; keeps what the first call was
        fed (its first 64 queries x kmax candidates) and everything it returned."""
        device = _dev()
        args, object_encoder, encode_hints = prod_fine.args, prod_fine.object_encoder, prod_fine.encode_hints

        def forward_packed(self, xyz, rgb, center, mean_rgb, cell_ptr, hints, class_idx=None, color_idx=None):
            out = prod_fine.forward_packed(xyz, rgb, center, mean_rgb, cell_ptr, hints, class_idx, color_idx)
            if "in" not in seen:
                seen["in"] = [t[: n_spy * pad].cpu().numpy() for t in (xyz, rgb, center, mean_rgb)] + [np.asarray(cell_ptr)[: n_spy + 1]]
                seen["out"] = {k: out[k][:n_spy].cpu().numpy() for k in ("matches0", "offsets", "P")}
            seen.setdefault("m0", []).append(out.matches0.cpu().numpy())
            seen.setdefault("off", []).append(out.offsets.cpu().numpy())
            return out
    np.random.seed(2022)
    torch.cuda.synchronize()
    timings = {}
    t0 = time.perf_counter()
    out = PL.evaluate(hip, Spy(), sc, tf, top_k, threshs, pad, queries_per_call=64, timings=timings)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    t0 = time.perf_counter()
    out2 = PL.run_coarse(hip, sc, tf, top_k, threshs)        # the coarse stage alone (uploads the scene again)
    wall_coarse2 = time.perf_counter() - t0
    assert out2[0] == out["retrievals"]
    np.random.seed(2022)
    t0 = time.perf_counter()
    out3 = PL.evaluate(hip, Spy(), sc, tf, top_k, threshs, pad, queries_per_call=64)
    torch.cuda.synchronize()
    wall_again = time.perf_counter() - t0
    assert out3["retrievals"] == out["retrievals"] and out3["fine_offset"] == out["fine_offset"]
    assert "in" in seen, "the pipeline did not take the on-device input path"

    # ---- oracle coarse stage on the same draws
    oc = _OracleCoarse(om)
    enc, hip_enc = [], []
    for lo in range(0, n_cells, 64):
        objs = [c.objects for c in sc.all_cells[lo: lo + 64]]
        pts = [D.batch_object_points(o, tf.for_cell(lo + i)) for i, o in enumerate(objs)]
        enc.append(oc.encode_objects(objs, pts))
        with torch.no_grad():
            hip_enc.append(hip.encode_objects(objs, pts).cpu())
    cell_enc, text_enc = torch.cat(enc).numpy(), oc.encode_text(sc.texts).numpy()
    # the on-device input path produced the very embeddings of the host chain (same draws, same packed bits)
    from text2pos_amd.scene import DeviceScene
    with torch.no_grad():
        assert torch.equal(hip.encode_scene_cells(DeviceScene(sc.all_cells, _dev()), tf).cpu(), torch.cat(hip_enc))
    per_cell = np.abs(torch.cat(hip_enc).numpy() - cell_enc).max(axis=1)
    flipped = per_cell >= TOL
    assert flipped.sum() <= n_cells // 200, f"{int(flipped.sum())} of {n_cells} cells differ from the oracle by >= 1e-4"
    widx, wscore = retrieve_topk_f64(cell_enc, text_enc, kmax + 1)
    db_ids = [c.id for c in sc.all_cells]
    row_of = {cid: i for i, cid in enumerate(db_ids)}
    got_idx = np.array([[row_of[cid] for cid in r] for r in out["retrievals"]])
    gap = (wscore[:, :-1] - wscore[:, 1:]).min(axis=1)
    clear = (gap > 2e-4) & ~flipped[widx].any(axis=1) & ~flipped[got_idx].any(axis=1)
    assert int(clear.sum()) >= 512, f"only {int(clear.sum())} of {n_poses} queries have unambiguous oracle rankings"
    assert np.array_equal(got_idx[clear], widx[clear][:, :kmax]), "a query with clear score gaps retrieved other cells"
    # accuracy tables from the lists the pipeline returned, through the metric functions the reference fixture pins
    centers = np.array([c.get_center()[0:2] for c in sc.all_cells])
    w_hit, w_close, _ = E.retrieval_accuracies(got_idx, db_ids, [p.cell_id for p in sc.all_poses],
                                               np.array([p.pose_w for p in sc.all_poses]), centers, 30.0, list(top_k))
    assert out["hit"] == w_hit and out["close"] == w_close
    assert out["localisation"] == E.localisation_accuracies(sc.all_poses, out["retrievals"], sc.cells_dict, list(top_k), list(threshs))

    # ---- fine stage: what the first call packed on the GPU for its first 64 queries, through the oracle
    from text2pos_amd.superglue_matcher import MatchOutputs
    fx, fr, fc, fm, fcp = seen["in"]
    assert fx.shape == (n_spy * pad, 256, 3) and fcp[-1] == n_spy * pad
    hints = [E.create_hint_description(sc.all_poses[b // kmax]) for b in range(n_spy)]
    want = MatchOutputs(**orc_fine.forward_packed(fx, fr, fc, fm, fcp, hints))
    # ... and those packed samples are the host chain's: sample q * kmax + c = cell retrievals[q][c], cut / padded to 16
    for b in (0, 7, n_spy - 1):
        cell = sc.cells_dict[out["retrievals"][b // kmax][b % kmax]]
        objs = list(cell.objects)[:pad]
        hp = D.batch_object_points(objs, tf.for_cell(b))
        assert np.array_equal(fx[b * pad: b * pad + len(objs)].reshape(-1, 3), hp.pos.numpy())
        assert np.array_equal(fr[b * pad: b * pad + len(objs)].reshape(-1, 3), hp.x.numpy())
        assert np.array_equal(fc[b * pad: b * pad + len(objs)], np.stack([o.get_center() for o in objs]).astype(np.float32))
    wP, woff, wm0 = want.P.numpy(), want.offsets.numpy(), want.matches0.numpy()
    assert np.abs(seen["out"]["P"] - wP).max() < TOL and np.abs(seen["out"]["offsets"] - woff).max() < TOL
    m0 = seen["out"]["matches0"]
    diff = np.argwhere(m0 != wm0)
    assert len(diff) <= 0.01 * m0.size, f"{len(diff)} of {m0.size} match entries differ"
    for b, o in diff:      # a differing entry must be a decision the oracle itself takes with a margin below 1e-3
        row = wP[b, o, :-1] if wP.shape[2] == woff.shape[1] + 1 else wP[b, o]
        top2 = np.sort(row)[-2:]
        j = int(np.argmax(row))
        col = np.sort(wP[b, :wm0.shape[1], j])[-2:]
        margin = min(top2[1] - top2[0], col[1] - col[0], abs(top2[1] - 0.2))
        assert margin < 1e-3, f"sample {b}, object {o}: matches differ although the oracle's margin is {margin:.2e}"
    assert (m0 >= 0).any() and (m0 < 0).any()
    # the fine tables are the metric functions applied to the per-sample outputs the spy saw (first evaluate call)
    k_all, o_all = np.concatenate(seen["m0"])[: n_poses * kmax], np.concatenate(seen["off"])[: n_poses * kmax]
    assert k_all.shape[0] == n_poses * kmax
    twin = DeviceScene(sc.all_cells, "cpu", n_pad=pad, pad_seed=tf.seed)    # (host side only: same padding objects as evaluate's scene)
    rows = np.array([[twin.row_of[c] for c in r] for r in out["retrievals"]]).reshape(-1)
    cxy = twin.center64[twin.padded_object_ids(pad)[rows]][:, :, 0:2]
    for name, offs in (("fine_offset", o_all), ("fine_mean", np.zeros_like(o_all))):
        pos = E.positions_in_cell(cxy, k_all, offs).reshape(n_poses, kmax, 2)
        assert out[name] == E.localisation_accuracies(sc.all_poses, out["retrievals"], sc.cells_dict, list(top_k), list(threshs), pos)
    print(f"[configs[4] at 2,048 cells / 1,024 poses] pipeline.evaluate (upload + coarse + fine + metrics) {wall:.2f} s wall "
          f"first pass (scene upload {timings['scene_s']:.2f}, coarse {timings['coarse_s']:.2f}, fine {timings['fine_s']:.2f}), "
          f"{wall_again:.2f} s again; run_coarse alone (with its own upload) {wall_coarse2:.2f} s; "
          f"{int(clear.sum())} / {n_poses} queries with clear oracle rankings: identical lists; "
          f"{int(flipped.sum())} / {n_cells} cells with a kNN near-tie flip; fine stage: {len(diff)} / {m0.size} match entries "
          f"at a sub-1e-3 margin differ; hit@k {out['hit']}")
    assert wall_again < 3.0, "the end-to-end pipeline should be GPU-bound (about half a second), not host-bound"


def test_bench_exchange_runs_through_rccl_at_world_size_one():
    """BASELINE configs[2]'s exchange step on the one GPU there is: bench.py under torch.distributed.run with ONE rank and
    --force-exchange initialises the RCCL ("nccl") process group, runs the step's all_gather_into_tensor on device tensors
    through it and fills the `exchange` block of the JSON line - the code an 8-GPU run executes (bench.py, distributed.py)."""
    import json
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
           "--cells", "768", "--queries", "128", "--force-exchange", "--no-cpu-baseline", "--no-extras", "--no-dropin",
           "--no-fp32-pass", "--no-two-stream"]
    r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    ex = line["exchange"]
    assert ex["backend"] == "nccl" and ex["world_size"] == 1 and ex["forced_at_world_1"] and ex["events_recorded"] == 2
    assert ex["bytes_per_rank"] == 768 * 256 * 4 == ex["bytes_gathered"]
    assert len(ex["all_gather_ms_per_rank"]) == 1 and 0.0 < ex["all_gather_ms_per_rank"][0] < 50.0
    assert line["n_gpus"] == 1 and line["value"] > 0


def test_bench_starts_its_own_ranks_from_the_plain_command():
    """`python bench.py --gpus N` - the form the driver uses - with no launcher around it: bench.py re-executes itself under
    torch.distributed.run (one rank per GPU), rank 0 prints the ONE JSON line last, the exit code is the launcher's.  On the one
    GPU there is, --self-launch forces that route at N = 1; the line carries the RCCL exchange block and says how it started."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR", "TORCHELASTIC_RUN_ID"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--self-launch", "--steps", "2", "--warmup", "1", "--cells", "768",
           "--queries", "128", "--no-cpu-baseline", "--no-extras", "--no-dropin", "--no-fp32-pass", "--no-pipeline", "--no-trained"]
    r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "self-launch:" in r.stderr and "torch.distributed.run" in r.stderr
    last = [l for l in r.stdout.splitlines() if l.strip()][-1]
    line = json.loads(last)                                    # the JSON line is the LAST line of stdout
    assert line["launched"].startswith("self") and line["n_gpus"] == 1 and line["value"] > 0
    assert line["exchange"]["backend"] == "nccl" and line["exchange"]["world_size"] == 1
    # a rank that dies makes the plain command exit non-zero (here: an argument the ranks reject)
    bad = subprocess.run(cmd[:2] + ["--gpus", "1", "--self-launch", "--cells", "0", "--steps", "1"], cwd=root, env=env, capture_output=True,
                         text=True, timeout=600)
    assert bad.returncode != 0


def test_bench_two_ranks_on_one_gpu_retrieve_what_one_rank_retrieves():
    """The N > 1 branches of bench.py on the one GPU there is: `python bench.py --gpus 2 --share-gpu` starts two ranks (self-launch),
    both on cuda:0, process group gloo (RCCL refuses two ranks on one device): contiguous cell / query blocks per rank, the one
    all-gather of the cell embeddings, each rank ranking its query block against the full database, the per-rank exchange
    bookkeeping.  Same total database and query set as a one-rank run => rank 0's first queries retrieve the same global rows."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR", "TORCHELASTIC_RUN_ID"):
        env.pop(k, None)
    common = ["--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-extras", "--no-dropin", "--no-fp32-pass", "--no-pipeline",
              "--no-trained"]

    def run(extra):
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + extra + common, cwd=root, env=env, capture_output=True,
                           text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-3000:]
        return json.loads([l for l in r.stdout.splitlines() if l.strip()][-1])
    two = run(["--gpus", "2", "--share-gpu", "--cells", "384", "--queries", "64"])
    one = run(["--gpus", "1", "--cells", "768", "--queries", "128"])
    assert two["n_gpus"] == 2 and two["launched"].startswith("self")
    assert two["config"]["cells_total"] == 768 == one["config"]["cells_total"] and two["config"]["queries_total"] == 128
    ex = two["exchange"]
    assert ex["world_size"] == 2 and ex["backend"] == "gloo" and len(ex["all_gather_ms_per_rank"]) == 2
    assert ex["bytes_per_rank"] == 384 * 256 * 4 and ex["bytes_gathered"] == 768 * 256 * 4 and ex["events_recorded"] == 2
    assert all(v > 0.0 for v in ex["all_gather_ms_per_rank"])
    assert two["top_k_of_first_queries"] == one["top_k_of_first_queries"]
    rows = np.array(two["top_k_of_first_queries"])
    assert rows.shape == (16, 10) and rows.max() < 768 and (rows >= 384).any()      # rows of the OTHER rank's block are retrieved too
    assert len(ex["encode_ms_per_rank"]) == len(ex["ranking_ms_per_rank"]) == 2 and min(ex["encode_ms_per_rank"] + ex["ranking_ms_per_rank"]) > 0.0
    # an UNEVEN split (1,001 cells / 129 queries over two ranks: 501 + 500, 65 + 64): the padded-shard branch of all_gather_rows
    # (distributed.py: shards padded to the largest block, the padding rows cut out of the gathered matrix) on real kernels
    odd2 = run(["--gpus", "2", "--share-gpu", "--cells-total", "1001", "--queries-total", "129"])
    odd1 = run(["--gpus", "1", "--cells", "1001", "--queries", "129"])
    ex = odd2["exchange"]
    assert ex["padded_shards"] and ex["cells_per_rank"] == [501, 500] and ex["bytes_per_rank"] == 501 * 256 * 4
    assert ex["bytes_gathered"] == 1001 * 256 * 4 and odd2["config"]["cells_total"] == 1001 == odd1["config"]["cells_total"]
    assert odd2["top_k_of_first_queries"] == odd1["top_k_of_first_queries"]
    rows = np.array(odd2["top_k_of_first_queries"])
    assert rows.max() < 1001 and (rows >= 501).any()


def test_bench_runs_on_a_checkpoint_file(tmp_path):
    """`bench.py --weights <file>`: a checkpoint in the reference's format (whole pickled module, written by train_checkpoint.py after a
    short training run) becomes the benchmarked model - loaded through io.load_reference_checkpoint, no BatchNorm calibration pass, the
    guard clear, `config.weights` saying what ran."""
    import json
    import subprocess
    import sys
    import train_checkpoint as TC
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = str(tmp_path / "short.pth")
    model, info = TC.trained_model(path, steps=12, batch=16, train_cells=128)
    del model
    assert os.path.getsize(path) > 10_000_000 and len(info["epoch_losses"]) >= 1
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR", "TORCHELASTIC_RUN_ID"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--weights", path, "--cells", "256", "--queries", "32", "--steps", "1",
                        "--warmup", "1", "--no-cpu-baseline", "--no-extras", "--no-dropin", "--no-fp32-pass", "--no-pipeline"],
                       cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.strip()][-1])
    w = line["config"]["weights"]
    assert w["kind"] == "checkpoint file" and w["file"] == "short.pth" and "hit_at_k_held_out_2048_cells" in w
    assert line["fp16_range_guard"] == "clear" and "trained_weights" not in line and line["config"]["cells_total"] == 256


def test_sharded_pipeline_two_ranks_on_one_gpu(tmp_path):
    """`pipeline.evaluate` with the input side on the GPU under a process group of TWO ranks (both on cuda:0, gloo): every rank
    uploads the scene, encodes its block of the cells from it (global cell indices key the draws), the embeddings are gathered,
    each rank ranks and then matches its block of the queries, the estimates are gathered.  Retrieval lists and all accuracy
    tables equal the single-process run exactly (tests/tools/sharded_pipeline_worker.py)."""
    import json
    import socket
    import subprocess
    import sys
    from text2pos_amd import io as IO
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    np.random.seed(3)
    cells, poses = _toy_scene(n_cells=50, n_poses=37, seed=21)
    IO.save_scene(str(tmp_path / "sc"), "toy1", cells, poses)
    worker = os.path.join(root, "tests", "tools", "sharded_pipeline_worker.py")
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR", "TORCHELASTIC_RUN_ID"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, worker, str(tmp_path / "sc"), str(tmp_path / "one.json")], env=env, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), worker, str(tmp_path / "sc"), str(tmp_path / "two.json")], env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    one, two = json.load(open(tmp_path / "one.json")), json.load(open(tmp_path / "two.json"))
    assert one["retrievals"] == two["retrievals"] and len(one["retrievals"]) == 37
    for k in ("hit", "close", "localisation", "fine_mean", "fine_offset", "fine_mean_conf"):
        assert one[k] == two[k], k


def test_all_gather_rows_on_device_tensors_through_rccl():
    """distributed.all_gather_rows / sharded_retrieval with the "nccl" backend (RCCL) in a one-rank group, in this process:
    device tensors in, device tensors out, the forced collective returns the rows unchanged and t2p_sim_topk ranks them."""
    import socket
    import torch.distributed as dist
    import text2pos_amd as t2p
    from text2pos_amd import distributed as TD
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=_dev())
    try:
        g = torch.Generator().manual_seed(3)
        cells = torch.nn.functional.normalize(torch.randn(1000, 256, generator=g), dim=-1).to(_dev())
        queries = torch.nn.functional.normalize(torch.randn(37, 256, generator=g), dim=-1).to(_dev())
        got = TD.all_gather_rows(cells, 1000, force=True)
        import ctypes
        ctypes.CDLL(None).fflush(None)      # RCCL's version banner (C-level stdout) leaves libc's buffer here, not behind pytest's summary
        assert got.is_cuda and got.data_ptr() != cells.data_ptr() and torch.equal(got, cells)
        marks = []
        idx, sc = TD.sharded_retrieval(lambda lo, hi: cells[lo:hi], lambda lo, hi: queries[lo:hi],
                                       lambda q, c, k: t2p.retrieve_topk(c, q, k), 1000, 37, 10, around_exchange=marks.append,
                                       force_exchange=True)
        widx, wsc = t2p.retrieve_topk(cells, queries, 10)
        assert marks == ["begin", "end"] and torch.equal(idx, widx) and torch.equal(sc, wsc)
    finally:
        dist.destroy_process_group()


def _row_lists(nbr, cnt):
    """The compact level-1 edge-row lists k_sample_group writes, rebuilt from the neighbour tables: per centroid its hits
    ((c << 8) | point, ascending) and then its self loop (((c | 0x80) << 8) | c); four 0xFFFF behind the last row."""
    n_obj, nc, _ = nbr.shape
    rows = np.full((n_obj, nc * 33), 0xFFFF, dtype=np.uint16)
    n_rows = np.zeros(n_obj, dtype=np.uint16)
    for o in range(n_obj):
        out = []
        for c in range(nc):
            out += [(c << 8) | int(j) for j in nbr[o, c, : cnt[o, c]]]
            out.append(((c | 0x80) << 8) | c)
        rows[o, : len(out)] = out
        n_rows[o] = len(out)
    return rows, n_rows


def test_dedup_rows_drops_only_repeats():
    """t2p_dedup_rows against its contract: the kept list is the original list minus rows whose source point repeats an
    EARLIER point bit for bit (xyz and rgb); self-loop rows stay; order, terminator and counts are right; detection may
    miss a repeat (hash collision) but never drops anything else."""
    from text2pos_amd import ops, synthetic as S
    xyz, rgb, _, _ = S.make_objects(123, 0, 300)
    xyz[5] = xyz[5, :1]                      # one point 256 times
    rgb[5] = rgb[5, :1]
    rgb[6, 100:] = rgb[6, :156]              # same coordinates would be needed too: these are NOT repeats
    xyz[7, 200:] = xyz[7, :56]               # coordinates repeat, colours differ: not repeats either
    d_xyz, d_rgb = _to_dev(xyz, rgb)
    gt = ops.sample_group(d_xyz)
    rows, n_rows = _row_lists(gt["nbr"][0].cpu().numpy(), gt["cnt"][0].cpu().numpy())
    d_rows = torch.from_numpy(rows.view(np.int16)).to(_dev())
    d_n = torch.from_numpy(n_rows.view(np.int16)).to(_dev())
    ops.dedup_rows(d_xyz, d_rgb, d_rows, d_n)
    got = d_rows.cpu().numpy().view(np.uint16)
    got_n = d_n.cpu().numpy().view(np.uint16)
    dropped = kept_repeats = 0
    for o in range(xyz.shape[0]):
        pts = np.concatenate([xyz[o], rgb[o]], axis=1).view(np.uint32)
        first = {}
        repeat = np.zeros(256, dtype=bool)
        for j in range(256):
            key = pts[j].tobytes()
            repeat[j] = key in first
            first.setdefault(key, j)
        orig = rows[o, : n_rows[o]]
        is_loop = (orig & 0x8000) != 0
        src = orig & 0xFF
        must_keep = is_loop | ~repeat[src]
        kept = got[o, : got_n[o]]
        # subsequence of the original that contains every row that must stay
        it = iter(orig.tolist())
        assert all(any(v == w for w in it) for v in kept.tolist()), f"object {o}: not a subsequence"
        want_min = orig[must_keep]
        assert len(kept) >= len(want_min)
        km = np.zeros(len(orig), dtype=bool)
        pos = 0
        for i, w in enumerate(orig.tolist()):                      # which original rows survived (greedy match)
            if pos < len(kept) and kept[pos] == w:
                km[i] = True
                pos += 1
        assert pos == len(kept) and (km | ~must_keep).all(), f"object {o}: a row that is no repeat was dropped"
        assert (got[o, got_n[o]: got_n[o] + 4] == 0xFFFF).all()
        dropped += int((~km).sum())
        kept_repeats += int((km & ~must_keep).sum())
    total_repeats = dropped + kept_repeats
    assert dropped > 0 and kept_repeats <= 0.08 * total_repeats, (dropped, kept_repeats)   # (hash collisions)
    assert got_n[5] == 128 * 2                                      # all-equal object: one hit + one self loop per centroid


def test_execution_plans_are_bit_identical(hip_model):
    """Equivalent execution plans must not change a single bit of any output: t2p_cell_config.tuning bit 0 (the edge rows of
    repeated points kept / dropped at SA level 1), and the cell batch cut into 1, 2 or 3 parts on as many HIP streams (own
    workspaces; 2 is the default from 2,048 cells up).  Bits outside T2P_TUNING_MASK are refused."""
    from text2pos_amd import synthetic as S
    from text2pos_amd._lib import T2PError
    xyz, rgb, center, mean_rgb, cell_ptr = S.make_cells(77, 40)
    args = _to_dev(xyz, rgb, center, mean_rgb)
    outs = {}
    try:
        for tuning in (0, 1):
            hip_model.tuning = tuning
            with torch.no_grad():
                outs[tuning] = hip_model.encode_objects_packed(*args, cell_ptr, want_trace=("sa_out", "obj_emb"))
        hip_model.tuning = 2
        with torch.no_grad(), pytest.raises(T2PError, match="tuning"):
            hip_model.encode_objects_packed(*args, cell_ptr)
    finally:
        hip_model.tuning = 0
    (out0, tr0), (out1, tr1) = outs[0], outs[1]
    assert torch.equal(out0, out1) and torch.equal(tr0["obj_emb"], tr1["obj_emb"])
    for l in range(3):
        assert torch.equal(tr0["sa_out"][l], tr1["sa_out"][l]), f"SA{l + 1} output depends on the dedup switch"
    with torch.no_grad():
        for n in (1, 2, 3):
            got = hip_model.encode_objects_packed(*args, cell_ptr, streams=n)
            again = hip_model.encode_objects_packed(*args, cell_ptr, streams=n)
            assert torch.equal(got, out0) and torch.equal(again, got), f"{n} streams"


def test_streams_handed_out_run_beside_each_other(hip_model):
    """HIP maps streams onto a few hardware queues by first use; two streams in one queue serialise and the two-stream encode loses its
    gain (88.0 -> 90.2 ms per step when the process had used three streams before, profiles/r06_t_stream_queues_ab.txt).
    ops.concurrent_stream probes for it: whatever the process did before (here: five streams created and used), the stream it hands
    out overlaps the current stream and the ones it is told to stand beside; the cell encoder's second stream comes from it and is
    re-picked when the caller's current stream changes."""
    from text2pos_amd import ops, synthetic as S
    DEV = _dev()
    used = [torch.cuda.Stream(device=DEV) for _ in range(5)]
    for st in used:
        with torch.cuda.stream(st):
            torch.zeros(8, device=DEV).add_(1)
    torch.cuda.synchronize()
    main = torch.cuda.current_stream(DEV)
    a = ops.concurrent_stream(DEV)
    b = ops.concurrent_stream(DEV, [a])
    assert len({main.cuda_stream, a.cuda_stream, b.cuda_stream}) == 3
    assert ops.streams_overlap(main, a) and ops.streams_overlap(main, b) and ops.streams_overlap(a, b)
    assert not ops._measure_overlap(a, a)                      # (the probe itself: one queue -> two spin times)
    args = _to_dev(*S.make_cells(78, 24)[:4])
    cell_ptr = S.make_cells(78, 24)[4]
    with torch.no_grad():
        one = hip_model.encode_objects_packed(*args, cell_ptr, streams=1)
        two = hip_model.encode_objects_packed(*args, cell_ptr, streams=2)
        aux = hip_model._aux_streams[0]
        assert ops.streams_overlap(main, aux)
        with torch.cuda.stream(a):                             # another current stream: the second stream must stand beside THAT one
            two_a = hip_model.encode_objects_packed(*args, cell_ptr, streams=2)
            assert hip_model._aux_streams[0].cuda_stream != a.cuda_stream and ops.streams_overlap(a, hip_model._aux_streams[0])
        torch.cuda.synchronize()
    assert torch.equal(one, two) and torch.equal(one, two_a)


def test_cold_cache_runs_are_bit_identical(hip_model):
    """The SA kernels of the default plan fetch their rows by LDS-DMA behind COUNTED `s_waitcnt vmcnt(n)` waits: a count that
    is one too high only shows when a fetch is slow.  Same cells with L2 / MALL flushed in front of the launch (1 GiB
    rewritten) and warm: every SA output and the embeddings must not change in a single bit.  (Found a wrong count in a
    round-3 lab kernel that 1 run in 12 exposed.)"""
    from text2pos_amd import synthetic as S
    xyz, rgb, center, mean_rgb, cell_ptr = S.make_cells(5001, 1500)
    args = _to_dev(xyz, rgb, center, mean_rgb)
    thrash = torch.empty(1 << 28, dtype=torch.float32, device=_dev())

    def run(tuning, cold):
        hip_model.tuning = tuning
        if cold:
            thrash.add_(1.0)
            torch.cuda.synchronize()
        with torch.no_grad():
            out, tr = hip_model.encode_objects_packed(*args, cell_ptr, want_trace=("sa_out",))
        return [x[:, :c].clone() for x, c in zip(tr["sa_out"], (64, 128, 256))] + [out.clone()]

    try:
        for tuning in (0, 1):
            ref = run(tuning, cold=False)
            for rep in range(4):
                got = run(tuning, cold=(rep % 2 == 0))
                for i, (a, b) in enumerate(zip(got, ref)):
                    assert torch.equal(a, b), f"tuning {tuning}, run {rep}: output {i} differs in {(a != b).any(1).sum().item()} rows"
            del ref, got
    finally:
        hip_model.tuning = 0


@pytest.mark.parametrize("precision", ["fp32", "f16x3"])
def test_result_does_not_depend_on_workspace_contents(hip_model, precision):
    """The scratch buffer is the caller's and arrives with arbitrary bytes.  Some columns of it are never written (the pad
    behind [features | xyz 0] of the SA output rows; their readers mask them, WsParams::k_live): fill every cached
    workspace with NaN bit patterns and with zeros, the cell embeddings must be the same bits."""
    from text2pos_amd import ops, synthetic as S
    xyz, rgb, center, mean_rgb, cell_ptr = S.make_cells(78, 24)
    args = _to_dev(xyz, rgb, center, mean_rgb)
    outs, saved = [], hip_model.precision
    try:
        hip_model.precision = precision
        with torch.no_grad():
            hip_model.encode_objects_packed(*args, cell_ptr)   # (allocates the workspace)
            for fill in (0xFF, 0x00, 0x7F):
                for ws in ops._workspaces.values():
                    if ws is not None:
                        ws.fill_(fill)
                outs.append(hip_model.encode_objects_packed(*args, cell_ptr))
    finally:
        hip_model.precision = saved
    assert torch.isfinite(outs[0]).all()
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])


@pytest.mark.parametrize("variation", [0, 1])
def test_coarse_model_at_embed_dim_128(vocab, variation):
    """--embed_dim 128 (training/args.py:19 makes it a free parameter): cell branch, text branch and retrieval against the
    oracle; both aggregation variants of the cell graph."""
    import weights as W
    import text2pos_amd as t2p
    from oracle import model as OM
    from oracle.model import retrieve_topk_f64
    from text2pos_amd import synthetic as S
    kw = dict(embed_dim=128, variation=variation)
    om = OM.OracleCellRetrieval(vocab["classes"], vocab["colors"], vocab["words"], OM.default_args(**kw)).eval()
    W.fill_state_dict(om, 31)
    for precision in ("f16x3", "fp32"):
        hm = t2p.CellRetrievalNetwork(vocab["classes"], vocab["colors"], vocab["words"], S.default_args(**kw), precision=precision)
        hm.load_state_dict(om.state_dict(), strict=True)
        hm = hm.to(_dev()).eval()
        xyz, rgb, center, mean_rgb, cell_ptr = S.make_cells(401, 7)
        assert int(cell_ptr[-1]) == 123
        # cells of 3 and 2 objects (fewer than k = 8 neighbours) and one of 70 (beyond the kNN kernel's LDS-staged size)
        cell_ptr = np.concatenate([[0], np.cumsum([9, 3, 12, 20, 2, 7, 70])]).astype(np.int32)
        texts = S.make_texts(401, 0, 9)
        want_c = om.encode_objects_packed(xyz, rgb, center, mean_rgb, cell_ptr)
        want_q = om.encode_text(texts)
        with torch.no_grad():
            got_c = hm.encode_objects_packed(*_to_dev(xyz, rgb, center, mean_rgb), cell_ptr)
            got_q = hm.encode_text(texts)
        assert got_c.shape == (7, 128) and (got_c.cpu() - want_c).abs().max().item() < TOL, precision
        assert (got_q.cpu() - want_q).abs().max().item() < TOL
        idx, score = t2p.retrieve_topk(got_c, got_q, 5)
        widx, wscore = retrieve_topk_f64(got_c.cpu().numpy(), got_q.cpu().numpy(), 5)
        assert np.array_equal(idx.cpu().numpy(), widx) and np.abs(score.cpu().numpy() - wscore).max() < 1e-12
