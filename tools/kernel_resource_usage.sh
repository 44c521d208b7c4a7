#!/bin/bash
# per-kernel register / scratch / LDS table of every source of the library (profiles/rNN/kernel_resource_usage.txt)
# usage: tools/kernel_resource_usage.sh > profiles/r04/kernel_resource_usage.txt
cd "$(dirname "$0")/../gpitch_amd/csrc" || exit 1
echo "# kernel resource usage of libgpitch_hip.so (hipcc -Rpass-analysis=kernel-resource-usage, gfx950)"
echo "# file | VGPRs | AGPRs | scratch bytes/lane | occupancy waves/SIMD | spilled VGPRs | LDS bytes/block (static) | kernel"
echo "#"
for f in $(sed -n 's/^SRCS *= *//p' Makefile); do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=on -Rpass-analysis=kernel-resource-usage -c "$f" -o /tmp/kru.o 2>&1 |
  python3 -c '
import re, sys, subprocess
fn = sys.argv[1]; cur = None; rows = []
for line in sys.stdin:
    m = re.search(r"remark:\s+(.*?)\s*\[-Rpass", line)
    if not m: continue
    t = m.group(1).strip()
    k = re.match(r"Function Name: (\S+)", t)
    if k: cur = {"name": k.group(1)}; rows.append(cur); continue
    if cur is None: continue
    for key, pat in (("v", r"^VGPRs: (\d+)"), ("a", r"^AGPRs: (\d+)"), ("s", r"^ScratchSize \[bytes/lane\]: (\d+)"), ("o", r"^Occupancy \[waves/SIMD\]: (\d+)"),
                     ("sp", r"^VGPRs Spill: (\d+)"), ("l", r"^LDS Size \[bytes/block\]: (\d+)")):
        mm = re.match(pat, t)
        if mm: cur[key] = int(mm.group(1))
names = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in rows), capture_output=True, text=True).stdout.split("\n")
for r, n in zip(rows, names):
    n = re.sub(r"\(.*$", "", n)
    print("%-20s %5d %4d %5d %2d %4d %6d  %s" % (fn, r.get("v", -1), r.get("a", 0), r.get("s", 0), r.get("o", 0), r.get("sp", 0), r.get("l", 0), n))
' "$f"
done
rm -f /tmp/kru.o
