#!/bin/bash
# Runs on the GPU box (gpurun): round 3's bench lines, rocprofv3 kernel statistics, the PMC passes the roofline figures are
# checked against, and the probes behind DESIGN.md 3.0 (float64 MFMA vs vector ALU).  Everything lands under
# gpurun_out/r03prof/; tools/collect_profiles_r03_copy.sh copies the summaries to profiles/r03/.
# Part selection: PARTS="bench stats pmc valu probes cfg5" (default: all; probes needs the tools/ab variant libraries)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03prof
mkdir -p $O
B="--no-cpu --no-f32-line --no-sgpr-lines"
PARTS=${PARTS:-"bench stats pmc valu probes cfg5"}
has() { [[ " $PARTS " == *" $1 "* ]]; }
if has bench; then
echo "== bench default (the driver's command + steps)" && timeout -k 10 600 python3 bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err || exit 1
echo "== bench overlap0" && timeout -k 10 300 python3 bench.py $B --overlap 0 --steps 20 --warmup 5 > $O/bench_overlap0.json 2> $O/bench_overlap0.err || exit 1
echo "== bench f32" && timeout -k 10 300 python3 bench.py $B --float-type f32 --steps 20 --warmup 5 > $O/bench_f32.json 2> $O/bench_f32.err || exit 1
echo "== bench round-2 strip kernels (GP_STRIP_LEAN=0 GP_KUF_DIRECT=0)" && GP_STRIP_LEAN=0 GP_KUF_DIRECT=0 timeout -k 10 300 python3 bench.py $B --steps 20 --warmup 5 > $O/bench_r02_kernels.json 2> $O/bench_r02_kernels.err || exit 1
fi
if has stats; then
echo "== stats default" && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_default -o s -- python3 bench.py $B --steps 5 --warmup 2 > $O/stats_default.log 2>&1 || exit 1
echo "== stats overlap0" && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_overlap0 -o s -- python3 bench.py $B --overlap 0 --steps 5 --warmup 2 > $O/stats_overlap0.log 2>&1 || exit 1
echo "== stats f32" && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_f32 -o s -- python3 bench.py $B --float-type f32 --steps 5 --warmup 2 > $O/stats_f32.log 2>&1 || exit 1
python3 tools/timeline.py $O/stats_default/s_kernel_trace.csv 150 > $O/timeline.txt 2>&1 || true
fi
if has pmc; then
for c in FETCH_SIZE WRITE_SIZE; do
  echo "== pmc $c" && timeout -k 10 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -o p -- python3 bench.py $B --steps 2 --warmup 1 > $O/pmc_$c.log 2>&1 || exit 1
  echo "== pmc $c overlap0" && timeout -k 10 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc0_$c -o p -- python3 bench.py $B --overlap 0 --steps 2 --warmup 1 > $O/pmc0_$c.log 2>&1 || exit 1
done
fi
if has valu; then
# matrix-core busy and instruction mix, kernels alone: this round's lean strip kernels against round 2's (same box)
for lean in 1 0; do
  echo "== pmc mfma busy lean=$lean" && GP_STRIP_LEAN=$lean timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_mfma_lean$lean -o p -- python3 bench.py $B --overlap 0 --steps 2 --warmup 1 > $O/pmc_mfma_lean$lean.log 2>&1 || exit 1
  echo "== pmc inst mix lean=$lean" && GP_STRIP_LEAN=$lean timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES --output-format csv -d $O/pmc_inst_lean$lean -o p -- python3 bench.py $B --overlap 0 --steps 2 --warmup 1 > $O/pmc_inst_lean$lean.log 2>&1 || exit 1
done
fi
if has probes; then
echo "== valu probe" && : > $O/valu_probe.txt
timeout -k 10 200 python3 tools/valu_probe.py >> $O/valu_probe.txt 2>> $O/valu_probe.err || exit 1
for n in 48 96; do
  GPITCH_AMD_LIB=tools/ab/lib_valu$n.so timeout -k 10 200 python3 tools/valu_probe.py >> $O/valu_probe.txt 2>> $O/valu_probe.err || exit 1
done
echo "== strip stamps" && GPITCH_AMD_LIB=tools/ab/lib_stamps.so timeout -k 10 200 python3 tools/strip_stamps.py > $O/strip_stamps.txt 2> $O/strip_stamps.err || exit 1
echo "== chol stamps" && GPITCH_AMD_LIB=tools/ab/lib_chstamps.so timeout -k 10 200 python3 tools/chol_stamps.py 512 > $O/chol_stamps.txt 2> $O/chol_stamps.err || exit 1
echo "== kuf A/B" && : > $O/kuf_ab.txt
for d in 3 0; do echo "GP_KUF_DIRECT=$d" >> $O/kuf_ab.txt; GP_KUF_DIRECT=$d timeout -k 10 200 python3 tools/bench_kuf.py >> $O/kuf_ab.txt 2>> $O/kuf_ab.err || exit 1; done
fi
if has cfg5; then
# BASELINE configs[4] (sgpr_ss, N = 65536, M = 512, 5 kernels): kernel statistics and one evaluation's launch list (eager
# launches under the profiler), the per-class timers without it; the hipGraph-replayed figure is bench_default.json's cfg5_sgpr
echo "== cfg5 stats" && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_cfg5 -o s -- python3 tools/time_sgpr.py --only f64 > $O/cfg5_time_profiled.log 2>&1 || exit 1
python3 tools/sgpr_timeline.py $O/stats_cfg5/s_kernel_trace.csv > $O/cfg5_timeline.txt 2>&1 || true
echo "== cfg5 timers" && timeout -k 10 300 python3 tools/time_sgpr.py > $O/cfg5_time.log 2> $O/cfg5_time.err || exit 1
grep float $O/cfg5_time.log > $O/cfg5_time.txt
fi
du -sh $O
