#!/bin/bash
# Runs on the GPU box (gpurun): the round's bench lines, rocprofv3 kernel statistics and the PMC passes the roofline
# figures are checked against.  Everything lands under gpurun_out/r02prof/; the summaries are copied to profiles/r02/.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r02prof
mkdir -p $O
B="--no-cpu --no-f32-line"
echo "== bench default" && timeout -k 10 300 python3 bench.py --no-cpu --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err || exit 1
echo "== bench overlap0" && timeout -k 10 300 python3 bench.py $B --overlap 0 --steps 20 --warmup 5 > $O/bench_overlap0.json 2> $O/bench_overlap0.err || exit 1
echo "== bench f32" && timeout -k 10 300 python3 bench.py $B --float-type f32 --steps 20 --warmup 5 > $O/bench_f32.json 2> $O/bench_f32.err || exit 1
echo "== stats default" && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_default -o s -- python3 bench.py $B --steps 5 --warmup 2 > $O/stats_default.log 2>&1 || exit 1
echo "== stats overlap0" && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_overlap0 -o s -- python3 bench.py $B --overlap 0 --steps 5 --warmup 2 > $O/stats_overlap0.log 2>&1 || exit 1
echo "== stats f32" && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_f32 -o s -- python3 bench.py $B --float-type f32 --steps 5 --warmup 2 > $O/stats_f32.log 2>&1 || exit 1
for c in FETCH_SIZE WRITE_SIZE; do
  echo "== pmc $c" && timeout -k 10 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -o p -- python3 bench.py $B --steps 2 --warmup 1 > $O/pmc_$c.log 2>&1 || exit 1
  echo "== pmc $c overlap0" && timeout -k 10 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc0_$c -o p -- python3 bench.py $B --overlap 0 --steps 2 --warmup 1 > $O/pmc0_$c.log 2>&1 || exit 1
done
echo "== pmc mfma (overlap 0: kernels alone)" && timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_mfma -o p -- python3 bench.py $B --overlap 0 --steps 2 --warmup 1 > $O/pmc_mfma.log 2>&1 || exit 1
find $O -name "*.csv" | head -40
du -sh $O
