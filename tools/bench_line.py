"""print value / ms_per_step / kernel classes of a bench.py JSON line read from stdin"""
import json, sys
lines = [l for l in sys.stdin.read().strip().splitlines() if l.startswith("{")]
d = json.loads(lines[-1])
print("steps/s %.3f  ms/step %.3f  %s" % (d["value"], d["ms_per_step"], {k: round(v, 3) for k, v in d.get("kernel_ms_per_step", {}).items()}))
