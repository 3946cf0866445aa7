# Timing + torch.profiler table of one coarse training step on the HIP path (docs/notebook.md 4.8): python profiles/train_step_time.py [batch]
import sys, time, torch, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests/golden')
import weights as W
import text2pos_amd as t2p
from text2pos_amd import synthetic as S
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
model = t2p.CellRetrievalNetwork(S.LABELS + ["pad"], S.COLOR_NAMES, S.known_words(), S.default_args()); W.fill_state_dict(model, 29); model = model.cuda()
c = S.make_cells(41, B); cells = [torch.from_numpy(a).cuda() for a in c[:4]]; cell_ptr = c[4]
texts = S.make_texts(41, 0, B)
opt = torch.optim.Adam(model.parameters(), lr=1e-3); crit = t2p.PairwiseRankingLoss(0.35); model.train()
for it in range(4):
    torch.cuda.synchronize(); t0 = time.time()
    opt.zero_grad(); a = model.encode_text(texts); torch.cuda.synchronize(); t1 = time.time()
    p = model.encode_objects_packed(*cells, cell_ptr); torch.cuda.synchronize(); t2 = time.time()
    loss = crit(a, p); loss.backward(); torch.cuda.synchronize(); t3 = time.time(); opt.step(); torch.cuda.synchronize(); t4 = time.time()
    print(f"step {it}: loss {loss.item():.4f}  text fwd {1e3*(t1-t0):.1f} ms  cells fwd {1e3*(t2-t1):.1f} ms  loss+bwd {1e3*(t3-t2):.1f} ms  adam {1e3*(t4-t3):.1f} ms  mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB")
# whole steps back to back (no synchronisation inside a step), text branch on the main stream / on its own stream as training.train_epoch runs it
def run_steps(n, overlap):
    side = torch.cuda.Stream() if overlap else None
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(n):
        opt.zero_grad()
        if side is not None:
            main = torch.cuda.current_stream(); side.wait_stream(main)
            with torch.cuda.stream(side):
                a = model.encode_text(texts)
            p = model.encode_objects_packed(*cells, cell_ptr); main.wait_stream(side); a.record_stream(main)
        else:
            a = model.encode_text(texts); p = model.encode_objects_packed(*cells, cell_ptr)
        loss = crit(a, p); loss.backward(); opt.step()
    torch.cuda.synchronize(); return 1e3 * (time.time() - t0) / n
for overlap in (False, True, False, True):
    print(f"8 steps back to back, text branch on {'its own stream' if overlap else 'the main stream'}: {run_steps(8, overlap):.2f} ms per step")
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
    opt.zero_grad(); loss = crit(model.encode_text(texts), model.encode_objects_packed(*cells, cell_ptr)); loss.backward(); torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=18, max_name_column_width=60))
print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=25, max_name_column_width=60))
