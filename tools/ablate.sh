#!/bin/bash
# usage: tools/ablate.sh ENVVAR v1 v2 ...   -> prints GEMM kernel ms/step for each value
var=$1; shift
for s in "$@"; do
  export $var=$s
  timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu 2>/dev/null | tail -1 > /tmp/abl.json
  python - "$var=$s" <<'PY'
import sys, json
d = json.load(open('/tmp/abl.json')); k = d["kernel_ms_per_step"]
print(sys.argv[1], "step %.2f" % d["ms_per_step"], {x: round(k[x], 2) for x in ("cond_A", "cond_LTA", "nt_gemm", "kuf_bar", "hyper", "chol", "kuf_build", "kuf_build_sm")})
PY
done
