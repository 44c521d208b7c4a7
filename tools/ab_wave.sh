#!/bin/bash
# same-box A/B of the strip-product forms: kernels alone (tools/valu_probe.py), then the bench step
mkdir -p gpurun_out
for cfg in "GPITCH_AMD_SWITCHES=strip_wave=0" "GPITCH_AMD_SWITCHES=strip_wave=1"; do
  echo "== $cfg" >> gpurun_out/ab_wave.txt
  env $cfg python tools/valu_probe.py >> gpurun_out/ab_wave.txt 2>&1 || exit 1
done
for cfg in "GPITCH_AMD_SWITCHES=strip_wave=0" "GPITCH_AMD_SWITCHES=strip_wave=1"; do
  echo "== bench $cfg" >> gpurun_out/ab_wave.txt
  env $cfg python bench.py --steps 10 --warmup 3 --no-cpu --no-f32-line --no-sgpr-lines > gpurun_out/ab_bench.json 2>gpurun_out/ab_bench.err || exit 1
  python tools/show_bench.py gpurun_out/ab_bench.json >> gpurun_out/ab_wave.txt 2>&1 || exit 1
done
cat gpurun_out/ab_wave.txt
