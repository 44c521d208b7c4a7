"""print the headline fields of a bench.py JSON line"""
import json, sys
d = json.load(open(sys.argv[1]))
print("steps/s %.3f  ms/step %.3f  roofline %.3f (%s)" % (d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["kernel"]))
print({k: round(v, 3) for k, v in d["kernel_ms_per_step"].items()})
