#!/bin/bash
# Runs on the GPU box (gpurun): round 4's bench lines, rocprofv3 kernel statistics, the PMC passes the roofline figures are
# checked against and the exact-integer probe of gemm_wave.hip.  Everything lands under gpurun_out/r04prof/ (the script then
# copies the summaries — no trace files — to profiles/r04/ inside the box's copy; gpurun merges gpurun_out/ back).
# Part selection: PARTS="bench stats pmc mfma cfg5 chol probe" (default: all)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04prof
mkdir -p $O
B="--no-cpu --no-f32-line --no-sgpr-lines"
PARTS=${PARTS:-"bench stats pmc mfma cfg5 chol probe"}
has() { [[ " $PARTS " == *" $1 "* ]]; }
if has bench; then
echo "== bench default (the driver's command + steps; whole-step CPU check included)" && timeout -k 10 900 python3 bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err || exit 1
echo "== bench overlap0" && timeout -k 10 300 python3 bench.py $B --overlap 0 --steps 20 --warmup 5 > $O/bench_overlap0.json 2> $O/bench_overlap0.err || exit 1
echo "== bench round-3 strip kernels (strip_wave=0)" && GPITCH_AMD_SWITCHES=strip_wave=0 timeout -k 10 300 python3 bench.py $B --steps 20 --warmup 5 > $O/bench_r03_kernels.json 2> $O/bench_r03_kernels.err || exit 1
echo "== bench round-3 strip kernels overlap0" && GPITCH_AMD_SWITCHES=strip_wave=0 timeout -k 10 300 python3 bench.py $B --overlap 0 --steps 20 --warmup 5 > $O/bench_r03_kernels_overlap0.json 2> $O/bench_r03_kernels_overlap0.err || exit 1
echo "== bench f32" && timeout -k 10 300 python3 bench.py $B --float-type f32 --steps 20 --warmup 5 > $O/bench_f32.json 2> $O/bench_f32.err || exit 1
fi
if has stats; then
echo "== stats default" && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_default -o s -- python3 bench.py $B --steps 5 --warmup 2 > $O/stats_default.log 2>&1 || exit 1
echo "== stats overlap0" && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_overlap0 -o s -- python3 bench.py $B --overlap 0 --steps 5 --warmup 2 > $O/stats_overlap0.log 2>&1 || exit 1
python3 tools/timeline.py $O/stats_default/s_kernel_trace.csv 150 > $O/bench_timeline.txt 2>&1 || true
fi
if has pmc; then
for c in FETCH_SIZE WRITE_SIZE; do
  echo "== pmc $c" && timeout -k 10 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -o p -- python3 bench.py $B --steps 2 --warmup 1 > $O/pmc_$c.log 2>&1 || exit 1
  echo "== pmc $c overlap0" && timeout -k 10 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc0_$c -o p -- python3 bench.py $B --overlap 0 --steps 2 --warmup 1 > $O/pmc0_$c.log 2>&1 || exit 1
done
fi
if has mfma; then
echo "== pmc mfma busy (kernels alone)" && timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_mfma -o p -- python3 bench.py $B --overlap 0 --steps 2 --warmup 1 > $O/pmc_mfma.log 2>&1 || exit 1
echo "== pmc inst mix (kernels alone)" && timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES --output-format csv -d $O/pmc_inst -o p -- python3 bench.py $B --overlap 0 --steps 2 --warmup 1 > $O/pmc_inst.log 2>&1 || exit 1
fi
if has cfg5; then
echo "== cfg5 stats" && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_cfg5 -o s -- python3 tools/time_sgpr.py --only f64 > $O/cfg5_time_profiled.log 2>&1 || exit 1
python3 tools/sgpr_timeline.py $O/stats_cfg5/s_kernel_trace.csv > $O/cfg5_timeline.txt 2>&1 || true
echo "== cfg5 timers" && timeout -k 10 300 python3 tools/time_sgpr.py > $O/cfg5_time.log 2> $O/cfg5_time.err || exit 1
grep float $O/cfg5_time.log > $O/cfg5_time.txt
echo "== cfg3 stats" && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_cfg3 -o s -- python3 bench.py $B --M 256 --partials 5 --float-type f32 --steps 5 --warmup 2 > $O/stats_cfg3.log 2>&1 || exit 1
python3 tools/timeline.py $O/stats_cfg3/s_kernel_trace.csv 0 > $O/cfg3_timeline.txt 2>&1 || true
echo "== cfg3 bench" && timeout -k 10 300 python3 bench.py $B --M 256 --partials 5 --float-type f32 --steps 30 --warmup 5 > $O/cfg3_bench.json 2> $O/cfg3_bench.err || exit 1
echo "== cfg5 from the graph, cluster on / off" && : > $O/cfg5_graph.txt
for sw in chol_cluster=1 chol_cluster=0; do echo "GPITCH_AMD_SWITCHES=$sw" >> $O/cfg5_graph.txt; GPITCH_AMD_SWITCHES=$sw timeout -k 10 300 python3 tools/bench_cfg5.py 30 2>/dev/null | grep -E "^f(64|32)" >> $O/cfg5_graph.txt || exit 1; done
fi
if has chol; then
echo "== cluster factorisation: kernel durations" && : > $O/chol_cluster.txt
for sw in chol_cluster=1 chol_cluster=0; do for m in 512 256 128; do
  GPITCH_AMD_SWITCHES=$sw TIME_M=$m timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/cc_$m -o s -- python3 tools/time_chol.py > $O/cc.log 2>&1 || exit 1
  grep "M=" $O/cc.log >> $O/chol_cluster.txt
  python3 -c "
import csv
for r in list(csv.DictReader(open('$O/cc_$m/s_kernel_stats.csv')))[:3]:
    if 'chol' in r['Name'] or 'tri_inv' in r['Name']: print('   %-70s calls %4s avg %9.1f us' % (r['Name'][:70], r['Calls'], float(r['AverageNs']) / 1e3))" >> $O/chol_cluster.txt
  rm -rf $O/cc_$m
done; done
if [ -f tools/ab/lib_ccstamps.so ]; then
  echo "== cluster factorisation: stamps of the chain" && GPITCH_AMD_LIB=tools/ab/lib_ccstamps.so timeout -k 10 100 python3 tools/chol_cluster_stamps.py 512 1 > $O/chol_cluster_stamps.txt 2>&1 || exit 1
fi
echo "== small Pdgp models, cluster on / off" && : > $O/bench_small_P.txt
for sw in chol_cluster=1 chol_cluster=0; do for P in 1 2; do echo "GPITCH_AMD_SWITCHES=$sw P=$P" >> $O/bench_small_P.txt; GPITCH_AMD_SWITCHES=$sw timeout -k 10 200 python3 bench.py $B --P $P --steps 10 --warmup 3 2>/dev/null | python3 tools/bench_line.py >> $O/bench_small_P.txt || exit 1; done; done
fi
if has probe; then
echo "== exact-integer probe of the wave kernels" && : > $O/probe_wave.txt
for args in "1 256 32768 8 40" "3 256 32768 8 40" "1 512 32768 12 20" "3 512 32768 12 20"; do timeout -k 10 200 ./tools/probe_wave $args 2>&1 | tail -1 >> $O/probe_wave.txt || exit 1; done
fi
# summaries only -> profiles/r04 (inside gpurun_out so that they travel back)
P=$O/summary; mkdir -p $P
cp $O/bench.json $O/bench_overlap0.json $O/bench_r03_kernels.json $O/bench_r03_kernels_overlap0.json $O/bench_f32.json $P/ 2>/dev/null
cp $O/stats_default/s_kernel_stats.csv $P/bench_kernel_stats.csv 2>/dev/null
cp $O/stats_overlap0/s_kernel_stats.csv $P/bench_overlap0_kernel_stats.csv 2>/dev/null
cp $O/stats_cfg5/s_kernel_stats.csv $P/cfg5_kernel_stats.csv 2>/dev/null
cp $O/stats_cfg3/s_kernel_stats.csv $P/cfg3_kernel_stats.csv 2>/dev/null
cp $O/bench_timeline.txt $O/cfg5_timeline.txt $O/cfg5_time.txt $O/probe_wave.txt $O/cfg3_timeline.txt $O/cfg3_bench.json $O/cfg5_graph.txt $O/chol_cluster.txt $O/chol_cluster_stamps.txt $O/bench_small_P.txt $P/ 2>/dev/null
python3 tools/make_traffic_json.py $O/pmc_FETCH_SIZE/p_counter_collection.csv $O/pmc_WRITE_SIZE/p_counter_collection.csv $P/hbm_traffic.json 2>/dev/null
python3 tools/make_traffic_json.py $O/pmc0_FETCH_SIZE/p_counter_collection.csv $O/pmc0_WRITE_SIZE/p_counter_collection.csv $P/hbm_traffic_overlap0.json 2>/dev/null
python3 tools/pmc_summary.py $O/pmc_mfma/p_counter_collection.csv > $P/overlap0_sq_counters.txt 2>/dev/null
python3 tools/pmc_summary.py $O/pmc_inst/p_counter_collection.csv >> $P/overlap0_sq_counters.txt 2>/dev/null
# the trace CSVs are large: keep the summaries, drop the rest from what travels back
rm -rf $O/stats_default $O/stats_overlap0 $O/stats_cfg5 $O/stats_cfg3 $O/pmc_* $O/pmc0_*
du -sh $O; ls $P
