"""Fine-stage benchmark (BASELINE.json configs[3]: "superglue_matcher object<->hint attention + offset_regression
forward on top-10 candidates, 1 GPU").  Not the driver's headline bench (that is bench.py); same conventions:
synthetic KITTI360Pose-shaped inputs resident in HBM before the timed region, random-init weights of the reference
architecture (embed_dim 128, 6 x [self, cross] GNN layers, 50 Sinkhorn iterations, pad_size 16, 6 hints), one JSON line.

    python bench_fine.py [--queries 1000] [--topk 10] [--steps 3] [--warmup 1]

A step = SuperGlueMatch.forward on queries x topk (query, candidate cell) pairs: ObjectEncoder over 16 objects per
pair, LanguageEncoder over 6 hint sentences per pair, matcher + offsets.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import bench as B  # noqa: E402  (shares the input generator)


def run(queries=1000, topk=10, steps=3, warmup=1, precision="f16x3"):
    """One fine-stage measurement; returns the JSON-ready dict (bench.py --fine / its default report call this too)."""
    from types import SimpleNamespace
    args = SimpleNamespace(queries=queries, topk=topk, steps=steps, warmup=warmup, precision=precision)
    import torch
    import text2pos_amd as t2p
    from text2pos_amd import ops, synthetic as S
    from text2pos_amd.modules import tokenize

    n_pairs = args.queries * args.topk
    workers = max(1, min(64, os.cpu_count() or 1))
    xyz, rgb, center, mean_rgb, cell_ptr = B.generate_cells(S, B.SEED + 3, n_pairs, 0, n_pairs, workers, fixed_n=16)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    torch.manual_seed(1234)
    margs = S.default_args()
    margs.embed_dim, margs.num_layers, margs.sinkhorn_iters = 128, 6, 50
    model = t2p.SuperGlueMatch(S.LABELS + ["pad"], S.COLOR_NAMES, S.known_words(), margs, precision=args.precision)
    g = torch.Generator().manual_seed(4321)
    for m in model.modules():
        if isinstance(m, torch.nn.BatchNorm1d):
            m.running_mean.copy_(torch.randn(m.num_features, generator=g) * 0.1)
            m.running_var.copy_(torch.rand(m.num_features, generator=g) + 0.5)
    model = model.to(dev).eval()
    d_in = [torch.from_numpy(a).to(dev) for a in (xyz, rgb, center, mean_rgb)]
    # every query's 6 single-hint sentences, repeated for its topk candidates
    sent = S.make_texts(B.SEED + 3, 0, args.queries * 6, n_hints=1)
    flat = [sent[q * 6 + h] for q in range(args.queries) for _ in range(args.topk) for h in range(6)]
    tok, lens = tokenize(flat, model.language_encoder.known_words)
    hints = (torch.from_numpy(tok).to(dev), torch.from_numpy(lens).to(dev))

    def step():
        with torch.no_grad():
            return model.forward_packed(*d_in, cell_ptr, hints, check_overflow=False)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    ops.profile_enable(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    ops.profile_enable(False)
    prof = ops.profile_report()
    phases = {k: round(v[1] / args.steps, 3) for k, v in sorted(prof.items(), key=lambda kv: -kv[1][1])}
    matcher_ms = sum(v for k, v in phases.items() if k.startswith("match_"))
    assert bool((out.P >= 0).all()) and bool((out.matches0 >= -1).all()) and bool((out.matches0 < 6).all())
    assert model.overflow_detected() == 0, "fp16-range guard fired: the f16x3 numbers are invalid"
    return ({
        "metric": "fine stage: (query, candidate cell) pairs matched per second (16 objects x 6 hints, embed_dim 128)",
        "value": n_pairs / (elapsed / args.steps), "unit": "pairs/s", "n_gpus": 1, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
        "dtype": "f16x3 / f32 MFMA for the encoders, f32 for the matcher", "data": "synthetic",
        "config": {"workload": f"{args.queries} queries x top-{args.topk} cells = {n_pairs} pairs, {n_pairs * 16} objects x 256 "
                               f"pts, {n_pairs * 6} hint sentences; GNN 6 x [self, cross], 50 Sinkhorn iterations"},
        "queries_per_s": args.queries / (elapsed / args.steps),
        "matched_fraction": float((out.matches1 >= 0).float().mean()),
        "kernel_ms_per_step": phases, "matcher_kernels_ms_per_step": round(matcher_ms, 3),
    })


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--queries", type=int, default=1000)
    ap.add_argument("--topk", type=int, default=10)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--precision", choices=["f16x3", "fp32"], default="f16x3")
    a = ap.parse_args()
    print(json.dumps(run(a.queries, a.topk, a.steps, a.warmup, a.precision)), flush=True)


if __name__ == "__main__":
    main()
