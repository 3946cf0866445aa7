"""A second opinion on the oracle's UNPINNED integer stages (CPU; SURVEY.md 8(c): torch_geometric / torch_cluster are absent, so
oracle/primitives.c restates fps / radius / knn from their published algorithms and nothing of the reference pins them).

These tests do not pin PyG either.  They check the C restatement against implementations that share no code with it and come
with this image: scipy.spatial.cKDTree (ball query), sklearn.neighbors + torch.cdist (kNN), and NumPy float32 array arithmetic
written from the papers' definitions (FPS: Qi et al., PointNet++, sec. 3.2 - "iterative farthest point sampling"; ball query:
all points within a radius, capped).  Inputs are the bench generator's objects (text2pos_amd.synthetic, the distribution every
GPU parity test draws from), including the duplicate-heavy ones (m = 25 base points resampled to 256).

Reference call sites: models/pointcloud/pointnet2.py:26-30 (fps, radius), models/cell_retrieval.py:46-48,97 (DynamicEdgeConv knn).
"""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import lib
from text2pos_amd import synthetic as S

SEED = 20220002


def _fp(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def _objects(n=96):
    """Generator objects + the most duplicate-heavy ones of a larger draw (fewest distinct points)."""
    xyz = S.make_objects(SEED, 0, 2048)[0]
    uniq = np.array([len(np.unique(o, axis=0)) for o in xyz])
    heavy = np.argsort(uniq, kind="stable")[:16]
    assert uniq[heavy].max() <= 30, "the generator should produce objects with ~25 distinct points"
    pick = np.concatenate([np.arange(n - 16), heavy])
    return np.ascontiguousarray(xyz[pick]), uniq[pick]


def _oracle_fps(xyz, n_samples):
    out = np.zeros((xyz.shape[0], n_samples), np.int32)
    lib().t2p_oracle_fps(_fp(xyz, C.c_float), C.c_int64(xyz.shape[0]), C.c_int32(xyz.shape[1]), C.c_int32(n_samples),
                         _fp(out, C.c_int32))
    return out


def _oracle_ball(xyz, cent, r, cap=32):
    n, nc = cent.shape
    nbr = np.zeros((n, nc, cap), np.int32)
    cnt = np.zeros((n, nc), np.int32)
    lib().t2p_oracle_ball_query(_fp(xyz, C.c_float), _fp(np.ascontiguousarray(cent), C.c_int32), C.c_int64(n),
                                C.c_int32(xyz.shape[1]), C.c_int32(nc), C.c_float(r), C.c_int32(cap), _fp(nbr, C.c_int32),
                                _fp(cnt, C.c_int32))
    return nbr, cnt


def _d2_f32(p, q):
    """Squared distances of the rows of p [n, 3] to the point q [3] in IEEE float32 array arithmetic: (dx dx + dy dy) + dz dz."""
    d = p - q[None, :]
    sq = d * d
    assert sq.dtype == np.float32
    return (sq[:, 0] + sq[:, 1]) + sq[:, 2]


def _numpy_fps(p, n_samples):
    """Farthest point sampling as the paper states it: start from the first point; repeatedly take the point farthest from
    the chosen set (running minimum of squared distances; np.argmax = first maximum)."""
    chosen = [0]
    dist = _d2_f32(p, p[0])
    for _ in range(1, n_samples):
        nxt = int(np.argmax(dist))
        chosen.append(nxt)
        dist = np.minimum(dist, _d2_f32(p, p[nxt]))
    return chosen


def test_fps_equals_a_numpy_restatement_of_the_paper():
    xyz, uniq = _objects()
    got = _oracle_fps(xyz, 128)
    for o in range(xyz.shape[0]):
        assert got[o].tolist() == _numpy_fps(xyz[o], 128), f"object {o} ({uniq[o]} distinct points)"
    # the defining property, independent of any tie rule: every pick is AT the maximum of the running min-distance (float64)
    for o in (0, 17, xyz.shape[0] - 1):
        p = xyz[o].astype(np.float64)
        dist = ((p - p[0]) ** 2).sum(1)
        for s in range(1, 128):
            j = got[o, s]
            assert dist[j] >= dist.max() - 1e-6 * max(1.0, dist.max())
            dist = np.minimum(dist, ((p - p[j]) ** 2).sum(1))
    # levels 2 and 3 sample the previous level's subset (models/pointcloud/pointnet2.py:26 on pos[idx])
    sub = np.ascontiguousarray(np.take_along_axis(xyz, got[:, :, None].astype(np.int64), axis=1))
    got2 = _oracle_fps(sub, 64)
    for o in range(0, xyz.shape[0], 7):
        assert got2[o].tolist() == _numpy_fps(sub[o], 64)


@pytest.mark.parametrize("r,n_cent", [(0.2, 128), (0.4, 64)])
def test_ball_query_vs_ckdtree_and_numpy(r, n_cent):
    """scipy's k-d tree finds the geometric neighbourhood in float64; the oracle decides `d2 < r2` in float32.  The two can
    only disagree inside a rounding band around the sphere, so:  tree(r (1 - 1e-6)) is a subset of the oracle's hits, which are a
    subset of tree(r (1 + 1e-6));  inside the band the decision is re-derived with NumPy float32 arithmetic;  the kept list is the first 32
    hits in ascending index (torch_cluster's radius: loop over x ascending, stop at max_num_neighbors)."""
    from scipy.spatial import cKDTree
    xyz, uniq = _objects()
    cent = _oracle_fps(xyz, n_cent)
    nbr, cnt = _oracle_ball(xyz, cent, r)
    r32 = np.float32(r)
    r2 = r32 * r32
    band_total = 0
    for o in range(xyz.shape[0]):
        p = xyz[o]
        tree = cKDTree(p.astype(np.float64))
        centres = p[cent[o]].astype(np.float64)
        inner = tree.query_ball_point(centres, float(r32) * (1.0 - 1e-6))
        outer = tree.query_ball_point(centres, float(r32) * (1.0 + 1e-6))
        for c in range(n_cent):
            lo, hi = set(inner[c]), set(outer[c])
            assert lo <= hi
            hits_np = np.flatnonzero(_d2_f32(p, p[cent[o, c]]) < r2)          # NumPy's float32 decision, all points
            hits = set(hits_np.tolist())
            assert lo <= hits <= hi, f"object {o} centroid {c}: float32 decision outside the float64 band"
            band_total += len(hi - lo)
            want = hits_np[:32]                                               # ascending index, capped
            assert cnt[o, c] == len(want) and nbr[o, c, : len(want)].tolist() == want.tolist()
            assert (nbr[o, c, len(want):] == -1).all()
            assert cent[o, c] in hits                                         # a centroid lies in its own ball (d2 = 0 < r2)
    assert (cnt == 32).any() and (cnt < 32).any()                             # both the capped and the uncapped case occurred
    print(f"[ball query r={r}] {band_total} points inside the rounding band over {xyz.shape[0] * n_cent} balls")


def test_knn_vs_sklearn_and_torch_cdist():
    """DynamicEdgeConv's graph: k = 8 nearest rows of the same cell (self included) among unit-norm 256-d embeddings.
    sklearn (float64 brute force) and torch.cdist (float32, another formula) are the independent searches.  Distances that
    differ by less than 1e-6 may legitimately order differently across arithmetic, so the check is: (a) the oracle's list
    satisfies the kNN property in float64 (nothing outside the list is closer than its farthest member, beyond rounding);
    (b) wherever the k-th / (k+1)-th float64 distances are separated by more than 1e-6 the three lists are the same SET, and
    where all gaps inside the list are clear too, the same SEQUENCE; (c) exact ties resolve to the lower index."""
    from sklearn.neighbors import NearestNeighbors
    rng = np.random.default_rng(5)
    sizes = [1, 2, 7, 8, 9, 26, 40, 13]
    seg = np.zeros(len(sizes) + 1, np.int32)
    seg[1:] = np.cumsum(sizes)
    x = rng.standard_normal((seg[-1], 256)).astype(np.float32)
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    x[seg[5] + 3] = x[seg[5] + 1]                                             # an exact duplicate inside the 26-row cell
    x[seg[6] + 10] = x[seg[6] + 2]
    k = 8
    got = np.zeros((seg[-1], k), np.int32)
    lib().t2p_oracle_knn(_fp(x, C.c_float), _fp(seg, C.c_int32), C.c_int32(len(sizes)), C.c_int32(256), C.c_int32(k),
                         _fp(got, C.c_int32))
    n_seq = n_set = 0
    for s in range(len(sizes)):
        lo, hi = int(seg[s]), int(seg[s + 1])
        m = hi - lo
        kk = min(k, m)
        xs = x[lo:hi].astype(np.float64)
        d64 = ((xs[:, None, :] - xs[None, :, :]) ** 2).sum(2)
        sk = NearestNeighbors(n_neighbors=kk, algorithm="brute").fit(xs).kneighbors(xs, return_distance=False)
        td = torch.cdist(torch.from_numpy(x[lo:hi]), torch.from_numpy(x[lo:hi]))
        tc = torch.argsort(td, dim=1, stable=True)[:, :kk].numpy()
        for i in range(m):
            mine = got[lo + i]
            assert (mine[kk:] == -1).all() and (mine[:kk] >= lo).all() and (mine[:kk] < hi).all()
            loc = mine[:kk] - lo
            assert len(set(loc.tolist())) == kk
            dl = d64[i, loc]
            assert (np.diff(dl) >= -1e-6).all()                                # ascending distance
            outside = np.setdiff1d(np.arange(m), loc)
            if len(outside):
                assert d64[i, outside].min() >= dl.max() - 1e-6               # (a) the kNN property
            order = np.argsort(d64[i], kind="stable")
            ds = d64[i, order]
            if kk == m or ds[kk] - ds[kk - 1] > 1e-6:                          # (b) clear boundary: same set
                assert set(loc.tolist()) == set(sk[i].tolist()) == set(tc[i].tolist())
                n_set += 1
                if (np.diff(ds[: kk]) > 1e-6).all():                           # all gaps clear: same sequence
                    assert loc.tolist() == sk[i].tolist() == tc[i].tolist() == order[:kk].tolist()
                    n_seq += 1
            assert loc[0] == min(i, int(np.flatnonzero(d64[i] == 0.0)[0]))     # (c) self first, or its lower-index duplicate
    dup = seg[5] + 3
    assert got[dup, 0] == seg[5] + 1 and got[dup, 1] == dup                    # exact tie at distance 0 -> lower index first
    assert n_set >= seg[-1] - 4 and n_seq >= seg[-1] // 2
