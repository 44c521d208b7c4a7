"""bench.py's cfg5 line on its own (SGPRSS bound + gradient, N = 65536, M = 512, 5 kernels; replayed from a hipGraph):
    python tools/bench_cfg5.py [steps]"""
import json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import bench


class A(object):
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20


r = bench.bench_cfg5(A(), False)
for k in ("f64", "f32"):
    print(k, "%.3f ms per evaluation (graph replay), %.3f eager;" % (r[k]["ms_per_evaluation"], r[k]["ms_per_evaluation_eager_launches"]),
          json.dumps({a: round(b, 3) for a, b in r[k]["kernel_ms_per_evaluation"].items()}))
