import sys, time, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import numpy as np, torch
from bench_windows import make_windows, build_model
from gpitch_amd import _lib
wins = make_windows(1, 2001, 64, 3, 10)
h = _lib.default_handle()
m = build_model(wins[0][3], h)
x = wins[0][0]
for name, fn in [("build_likelihood", lambda: m.build_likelihood()), ("predict_f", lambda: m.predict_f(x)), ("predict_s", lambda: m.predict_s(x))]:
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): fn()
    torch.cuda.synchronize()
    print(name, "%.2f ms" % ((time.perf_counter() - t0) / 5 * 1e3))
h.check(h.lib.gp_timers_enable(h.h, 1)); h.check(h.lib.gp_timers_reset(h.h))
m.predict_s(x); torch.cuda.synchronize()
print({k: v for k, v in h.timers().items() if v[1]})
