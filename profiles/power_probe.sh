#!/bin/bash
# Is the chip power-limited under this path's kernels?  Samples rocm-smi (socket power, sclk) every 0.25 s while
#   (1) nothing runs, (2) bench.py's step loops, (3) the bare f16 MFMA loop of microbench/mix2 (random operands) runs.
# usage (GPU box, repo root): bash profiles/power_probe.sh <outdir>
out=${1:-gpurun_out/power}; mkdir -p $out; cd /root/repo; export TMPDIR=/tmp
sample() {  # $1 = tag, $2 = seconds
  for i in $(seq 1 $(( $2 * 4 ))); do
    echo "== $(date +%s.%N)"; /opt/rocm/bin/rocm-smi -P -c 2>/dev/null | grep -E "Power|sclk|mclk|fclk"; sleep 0.25
  done > $out/smi_$1.txt
}
/opt/rocm/bin/rocm-smi --showsclkrange --showmclkrange 2>/dev/null | grep -vE "^=|^$" > $out/ranges.txt
/opt/rocm/bin/rocm-smi -M 2>/dev/null | grep -iE "power|cap" >> $out/ranges.txt
sample idle 2
python bench.py --steps 150 --warmup 5 --no-cpu-baseline --no-extras --no-fp32-pass --no-dropin --no-force-exchange > $out/bench_loop.json 2> $out/bench_loop.err &
sleep 9; sample bench 8; wait
hipcc --offload-arch=gfx950 -O3 -DMIX2_LONG=1 -o /tmp/mix2_long profiles/microbench/mix2.hip 2>/dev/null
/tmp/mix2_long > $out/mix2_long.txt 2>&1 &
sleep 1; sample mfma 6; wait
python - "$out" <<'PY'
import re, sys, statistics as st
out = sys.argv[1]
for tag in ("idle", "bench", "mfma"):
    txt = open(f"{out}/smi_{tag}.txt").read()
    pw = [float(x) for x in re.findall(r"Power \(W\):\s*([0-9.]+)", txt)] or [float(x) for x in re.findall(r"Power[^:]*:\s*([0-9.]+)", txt)]
    sc = [float(x) for x in re.findall(r"sclk[^(]*\(([0-9.]+)Mhz\)", txt)]
    print(tag, "samples", len(pw), "power W mean/max", round(st.mean(pw), 1) if pw else None, max(pw) if pw else None,
          "sclk MHz mean/min", round(st.mean(sc), 1) if sc else None, min(sc) if sc else None)
print(open(f"{out}/ranges.txt").read())
PY
