"""Upper bound of what two-stream chunk pipelining can give: the two halves of the 12k-cell batch encoded (a) back to back on
one stream, (b) concurrently on two streams with separate workspaces.  Any overlap of kernels from different chunks
shows up as (b) < (a)."""
import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np, torch
import text2pos_amd as t2p
from text2pos_amd import ops, synthetic as S
import bench as B
dev = torch.device("cuda:0")
torch.manual_seed(1234)
model = t2p.CellRetrievalNetwork(S.LABELS + ["pad"], S.COLOR_NAMES, S.known_words(), S.default_args()).to(dev).eval()
xyz, rgb, center, mean_rgb, cell_ptr = B.generate_cells(S, B.SEED, 12000, 0, 12000, 32)
half = 6000
o = int(cell_ptr[half])
parts = []
for lo, hi, a, b in ((0, half, 0, o), (half, 12000, o, int(cell_ptr[-1]))):
    d = [torch.from_numpy(x[a:b]).to(dev) for x in (xyz, rgb, center, mean_rgb)]
    cp = (cell_ptr[lo:hi + 1] - a).astype(np.int32)
    parts.append((d, cp, torch.from_numpy(cp).to(dev)))
orig_ws = ops.workspace
def ws_per_stream(device, nbytes, tag):
    return orig_ws(device, nbytes, tag + str(torch.cuda.current_stream(device).cuda_stream))
ops.workspace = ws_per_stream
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def run(concurrent, chunk=0):
    with torch.no_grad():
        outs = []
        for (d, cp, cpd), st in zip(parts, (s1, s2 if concurrent else s1)):
            with torch.cuda.stream(st):
                outs.append(model.encode_objects_packed(*d, cp, cpd, check_overflow=False, chunk_objects=chunk))
    return outs
for chunk in (0, 8192):
    for conc in (False, True, False, True):
        run(conc, chunk); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5): run(conc, chunk)
        torch.cuda.synchronize()
        print(f"chunk_objects={chunk or 32768} concurrent={conc}: {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms per 12k cells")
