"""Per-kernel time of ONE small call of the packed entry point (the reference's callers hand over 64 cells per call,
training/coarse.py:123-131): the library's own profile scopes (hipEvents around every launch, single stream).
    python profiles/tools/small_batch_kernels.py [cells per call, default 64]"""
import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch
import weights as W
import text2pos_amd as t2p
from text2pos_amd import ops, synthetic as S

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
model = t2p.CellRetrievalNetwork(S.LABELS + ["pad"], S.COLOR_NAMES, S.known_words(), S.default_args())
W.fill_state_dict(model, 29)
model = model.cuda().eval()
c = S.make_cells(41, B)
cells = [torch.from_numpy(a).cuda() for a in c[:4]]
reps = 50
with torch.no_grad():
    for _ in range(5):
        model.encode_objects_packed(*cells, c[4], streams=1).cpu()
    ts = []
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps):
            model.encode_objects_packed(*cells, c[4], streams=1).cpu()
        ts.append((time.perf_counter() - t0) / reps)
    ops.profile_enable(True)
    for _ in range(reps):
        model.encode_objects_packed(*cells, c[4], streams=1).cpu()
    torch.cuda.synchronize()
    ops.profile_enable(False)
rep = ops.profile_report()
print(f"B={B} ({cells[0].shape[0]} objects): {1e3 * min(ts):.3f} ms per call with .cpu() ({B / min(ts) / 1e3:.1f} k cells/s)")
total = 0.0
for name, v in sorted(rep.items(), key=lambda kv: -kv[1][1]):
    print(f"  {name:28s} {v[0] / reps:5.1f} launches  {1e3 * v[1] / reps:8.1f} us per call")
    total += v[1] / reps
print(f"  sum of kernels {1e3 * total:.1f} us per call")
