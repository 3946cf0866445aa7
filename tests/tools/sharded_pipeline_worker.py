"""Worker of tests/test_gpu_headline.py::test_sharded_pipeline_two_ranks_on_one_gpu: one rank of `pipeline.evaluate` on cuda:0
behind a gloo process group (RCCL refuses two ranks on one device; gloo's all_gather_into_tensor is staged through the host).

    python tests/tools/sharded_pipeline_worker.py <scene_dir> <out.json>      (RANK / WORLD_SIZE / MASTER_* from the environment)
"""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def build_models(dev):
    import text2pos_amd as t2p
    from text2pos_amd import pipeline as PL, synthetic as S
    torch.manual_seed(77)
    coarse = t2p.CellRetrievalNetwork(S.LABELS + ["pad"], S.COLOR_NAMES, S.known_words(), S.default_args())
    g = torch.Generator().manual_seed(5)
    for m in coarse.modules():           # spread the BatchNorm statistics (random-init statistics collapse the embeddings)
        if isinstance(m, torch.nn.BatchNorm1d):
            m.running_mean.copy_(torch.randn(m.num_features, generator=g) * 0.1)
            m.running_var.copy_(torch.rand(m.num_features, generator=g) + 0.5)
    fine = t2p.SuperGlueMatch(S.LABELS + ["pad"], S.COLOR_NAMES, S.known_words(), PL._model_args(128, num_layers=2, sinkhorn_iters=20))
    return coarse.to(dev).eval(), fine.to(dev).eval()


def main():
    scene_dir, out_path = sys.argv[1], sys.argv[2]
    from text2pos_amd import io as IO, pipeline as PL
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
        agit = dist.all_gather_into_tensor

        def staged(out, inp, group=None):
            if not inp.is_cuda:
                return agit(out, inp, group=group)
            o = torch.empty(out.shape, dtype=out.dtype)
            agit(o, inp.cpu(), group=group)
            out.copy_(o)
        dist.all_gather_into_tensor = staged
    np.random.seed(2022)                 # the scene's padding objects (Object3d.create_padding) are the same on every rank
    scenes = IO.load_scenes(scene_dir, ["toy1"])
    coarse, fine = build_models(dev)
    out = PL.evaluate(coarse, fine, scenes, PL.PerCellTransform(256, 3), (1, 3, 5), (5, 10, 15), 16, queries_per_call=16)
    if rank == 0:
        json.dump({k: (v if k == "retrievals" else {str(a): {str(b): c for b, c in d.items()} if isinstance(d, dict) else d for a, d in v.items()})
                   for k, v in out.items()}, open(out_path, "w"))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
