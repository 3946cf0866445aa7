#!/bin/bash
# Evidence run behind profiles/<tag>_*: GPU tests, bench lines, rocprofv3 kernel stats, FETCH / WRITE PMC passes.
# usage (on the GPU box, from the repo root): bash profiles/collect.sh r01_k   -> gpurun_out/<tag>/ (copy the summaries into profiles/)
tag=${1:-r01_j}
cd /root/repo; export TMPDIR=/tmp
mkdir -p gpurun_out/$tag
# SKIP_PYTEST=1: the GPU suite of the same tree ran in its own call (its tail is copied to profiles/<tag>_pytest.txt by hand)
[ -n "$SKIP_PYTEST" ] || timeout 2400 python -m pytest tests -q -m gpu --durations=8 2>&1 | tail -16 > gpurun_out/$tag/pytest.txt
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag/stats -o s -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fp32-pass --cell-streams 1 --no-extras --no-dropin --no-force-exchange --no-pipeline --no-trained > gpurun_out/$tag/stats.log 2>&1
trace=$(find gpurun_out/$tag/stats -name "s_kernel_trace.csv" | head -1)
python profiles/summarize.py $trace gpurun_out/$tag/kernel_stats.md "$tag: python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fp32-pass --cell-streams 1 --no-extras --no-dropin --no-force-exchange --no-pipeline --no-trained (12k cells + 1k queries, 1 x MI355X)" > /dev/null
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/$tag/fetch -o p -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-fp32-pass --cell-streams 1 --no-extras --no-dropin --no-force-exchange --no-pipeline --no-trained > gpurun_out/$tag/fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/$tag/write -o p -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-fp32-pass --cell-streams 1 --no-extras --no-dropin --no-force-exchange --no-pipeline --no-trained > gpurun_out/$tag/write.log 2>&1
python profiles/pmc_traffic.py gpurun_out/$tag/fetch gpurun_out/$tag/write gpurun_out/$tag/pmc_traffic.json > /dev/null
# matrix-pipe utilisation of the kernels (own pass: counters are never combined with the trace domains gpurun refuses)
timeout 600 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d gpurun_out/$tag/sq -o p -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-fp32-pass --cell-streams 1 --no-extras --no-dropin --no-force-exchange --no-pipeline --no-trained > gpurun_out/$tag/sq.log 2>&1
python profiles/pmc_summary.py $(dirname $(find gpurun_out/$tag/sq -name "p_counter_collection.csv" | head -1)) p 10 > gpurun_out/$tag/pmc_sq.txt 2>&1
# effective shader clock per kernel (is a kernel held by the 1,400 W package power cap?) + the power probe itself
timeout 600 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/$tag/grbm -o p -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-fp32-pass --cell-streams 1 --no-extras --no-dropin --no-force-exchange --no-pipeline --no-trained > gpurun_out/$tag/grbm.log 2>&1
python profiles/pmc_clock.py $(dirname $(find gpurun_out/$tag/grbm -name "p_counter_collection.csv" | head -1)) p 14 > gpurun_out/$tag/pmc_clock.txt 2>&1
bash profiles/power_probe.sh gpurun_out/$tag/power > gpurun_out/$tag/power_probe.txt 2>&1
# the bench line quotes the HBM traffic and the MFMA-busy fraction of the dominant kernel from the newest profiles/*_evidence.json
# (stamped with the kernel sources' hashes; bench.py refuses figures whose kernel source changed since): this run's
cp gpurun_out/$tag/pmc_traffic.json profiles/${tag}_pmc_traffic.json
python profiles/evidence.py gpurun_out/$tag/pmc_traffic.json gpurun_out/$tag/sq gpurun_out/$tag/grbm gpurun_out/$tag/evidence.json $tag > gpurun_out/$tag/evidence.txt 2>&1
cp gpurun_out/$tag/evidence.json profiles/${tag}_evidence.json
timeout 900 python bench.py --steps 20 --warmup 5 2> gpurun_out/$tag/bench.err | tail -1 > gpurun_out/$tag/bench.json
# the driver's own form (no flags): how long the whole default run takes, wall clock
t0=$SECONDS; timeout 900 python bench.py 2> gpurun_out/$tag/bench_default.err | tail -1 > gpurun_out/$tag/bench_default.json; echo "python bench.py (no flags): $((SECONDS - t0)) s wall" > gpurun_out/$tag/bench_default_wall.txt
timeout 600 python bench_fine.py 2> gpurun_out/$tag/bench_fine.err | tail -1 > gpurun_out/$tag/bench_fine.json
# keep the merge small: drop the raw traces
cp $trace gpurun_out/$tag/kernel_trace.csv 2>/dev/null   # (kept for re-summarising; not committed)
find gpurun_out/$tag -name "*.db" -delete; find gpurun_out/$tag -mindepth 2 -name "*.csv" -delete
cat gpurun_out/$tag/pytest.txt gpurun_out/$tag/bench_default_wall.txt 2>/dev/null; cut -c1-400 gpurun_out/$tag/bench.json; head -12 gpurun_out/$tag/kernel_stats.md
