#!/bin/bash
# Interleaved A/B of library builds on one box (same process order every round, so box drift hits all arms alike):
#   bash profiles/ab.sh <tag> <reps> <lib or "default"> [<lib> ...]      libs are paths relative to text2pos-cvpr2022_amd/
# Prints per run: the step (two streams), the single-stream step and the per-kernel milliseconds; appends to gpurun_out/<tag>/ab.txt.
# Variant libraries come from `python text2pos-cvpr2022_amd/build.py --variant NAME DEFINE...` (-> libt2p_hip_NAME.so).
tag=$1; reps=$2; shift 2
cd /root/repo; mkdir -p gpurun_out/$tag
common="--steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-dropin --no-fp32-pass --no-pipeline --no-trained --prof-steps 5 $AB_ARGS"
for rep in $(seq $reps); do
  for lib in "$@"; do
    if [ "$lib" = default ]; then L=""; else L=$PWD/text2pos-cvpr2022_amd/$lib; fi
    T2P_LIB=$L timeout 600 python bench.py $common 2>/dev/null | tail -1 | python -c "
import json, sys
d = json.loads(sys.stdin.read()); k = d['kernel_ms_per_step']
short = lambda n: n.replace('ws_edge_sa_', 'sa_').replace('ws_groupmax_', 'ga2_').replace('ws_dense_', 'tab_')
print('$lib step %.2f single %.2f | ' % (d['ms_per_step'], (d.get('single_stream') or {}).get('ms_per_step', 0.0)) + ' '.join('%s %.3f' % (short(n), v) for n, v in list(k.items())[:14]))
" | tee -a gpurun_out/$tag/ab.txt
  done
done
