#!/bin/bash
# cfg3 (BASELINE configs[2]: N = 32768, M = 256, P = 12, float32) under the profiler: timeline, kernel statistics alone and
# overlapped, matrix-core busy alone, HBM traffic alone.  Output under gpurun_out/r03cfg3/, summaries to profiles/r03/cfg3_*.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03cfg3
mkdir -p $O
B="--no-cpu --no-f32-line --no-sgpr-lines --M 256 --partials 5 --float-type f32"
timeout -k 10 300 python3 bench.py $B --steps 30 --warmup 5 > $O/bench.json 2> $O/bench.err || exit 1
timeout -k 10 300 python3 bench.py $B --overlap 0 --steps 30 --warmup 5 > $O/bench_overlap0.json 2> $O/bench_overlap0.err || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python3 bench.py $B --steps 5 --warmup 2 > $O/stats.log 2>&1 || exit 1
python3 tools/timeline.py $O/stats/s_kernel_trace.csv 20 > $O/timeline.txt 2>&1 || true
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats0 -o s -- python3 bench.py $B --overlap 0 --steps 5 --warmup 2 > $O/stats0.log 2>&1 || exit 1
timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_mfma -o p -- python3 bench.py $B --overlap 0 --steps 2 --warmup 1 > $O/pmc_mfma.log 2>&1 || exit 1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc0_$c -o p -- python3 bench.py $B --overlap 0 --steps 2 --warmup 1 > $O/pmc0_$c.log 2>&1 || exit 1
done
du -sh $O
