"""Which of a process's HIP streams share a hardware queue (and so serialise)?  Nine streams, each used once, then a single-thread spin
kernel on a pair of them behind a common start event: side by side the pair takes one spin time, in one queue two.
    python profiles/tools/probe_stream_queues.py            (GPU_MAX_HW_QUEUES=8 python ... for the 8-queue runtime setting)
Measured on MI355X / ROCm 7.2 (round 6): the default stream shares with the 7th used stream, stream 1 with stream 6; no pair
shares with GPU_MAX_HW_QUEUES=8.  ops.concurrent_stream is the product-side use of the same measurement."""
import sys, time
sys.path.insert(0, "/root/repo")
import torch
import text2pos_amd
from text2pos_amd import ops
dev = torch.device("cuda:0")
main = torch.cuda.current_stream(dev)
print("spin cycles", ops._spin_cycles(dev))
ss = []
for i in range(9):
    s = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(s):
        torch.zeros(8, device=dev).add_(1)
    ss.append(s)
torch.cuda.synchronize()
def pair_time(a, b):
    cycles = ops._spin_cycles(dev)
    torch.cuda.synchronize(dev)
    start, ea, eb = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    start.record(a); b.wait_event(start)
    with torch.cuda.stream(a):
        torch.cuda._sleep(cycles); ea.record(a)
    with torch.cuda.stream(b):
        torch.cuda._sleep(cycles); eb.record(b)
    ea.synchronize(); eb.synchronize()
    return start.elapsed_time(ea), start.elapsed_time(eb)
for i, s in enumerate(ss):
    print("main vs stream", i + 1, ["%.3f" % t for t in pair_time(main, s)], ops.streams_overlap(main, s))
for i in range(1, 9):
    print("stream 1 vs stream", i + 1, ["%.3f" % t for t in pair_time(ss[0], ss[i])])
