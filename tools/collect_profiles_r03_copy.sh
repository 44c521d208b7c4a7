#!/bin/bash
# After tools/collect_profiles_r03.sh came back through gpurun: summaries -> profiles/r03/
set -e
cd "$(dirname "$0")/.."
O=gpurun_out/r03prof; P=profiles/r03
mkdir -p $P
[ -s $O/bench_default.json ] && cp $O/bench_default.json $P/bench.json
for v in overlap0 f32 r02_kernels; do [ -s $O/bench_$v.json ] && cp $O/bench_$v.json $P/bench_$v.json; done
[ -f $O/stats_default/s_kernel_stats.csv ] && cp $O/stats_default/s_kernel_stats.csv $P/bench_kernel_stats.csv
[ -f $O/stats_overlap0/s_kernel_stats.csv ] && cp $O/stats_overlap0/s_kernel_stats.csv $P/bench_overlap0_kernel_stats.csv
[ -f $O/stats_f32/s_kernel_stats.csv ] && cp $O/stats_f32/s_kernel_stats.csv $P/bench_f32_kernel_stats.csv
[ -s $O/timeline.txt ] && cp $O/timeline.txt $P/bench_timeline.txt
if [ -f $O/pmc_FETCH_SIZE/p_counter_collection.csv ]; then
  python3 tools/make_traffic_json.py $O/pmc_FETCH_SIZE/p_counter_collection.csv $O/pmc_WRITE_SIZE/p_counter_collection.csv $P/hbm_traffic.json
  python3 tools/make_traffic_json.py $O/pmc0_FETCH_SIZE/p_counter_collection.csv $O/pmc0_WRITE_SIZE/p_counter_collection.csv $P/hbm_traffic_overlap0.json
fi
for lean in 1 0; do
  if [ -f $O/pmc_mfma_lean$lean/p_counter_collection.csv ]; then
    python3 tools/pmc_summary.py $O/pmc_mfma_lean$lean/p_counter_collection.csv $O/pmc_inst_lean$lean/p_counter_collection.csv > $P/overlap0_sq_counters_lean$lean.txt
  fi
done
[ -f $O/stats_cfg5/s_kernel_stats.csv ] && cp $O/stats_cfg5/s_kernel_stats.csv $P/cfg5_kernel_stats.csv
for f in cfg5_timeline cfg5_time; do [ -s $O/$f.txt ] && cp $O/$f.txt $P/$f.txt; done
for f in valu_probe strip_stamps chol_stamps kuf_ab; do [ -s $O/$f.txt ] && cp $O/$f.txt $P/$f.txt; done
ls -la $P
