"""Diagnostic: where a strip-GEMM workgroup's time goes (build gemm_strip.hip with -DGS_STAMPS into tools/ab/lib_stamps.so).
    GPITCH_AMD_LIB=tools/ab/lib_stamps.so python tools/strip_stamps.py
Per role and row-block: median cycles of prologue (entry -> first K-tile staged), K loop, epilogue (stores drained)."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from gpitch_amd import _lib
    from gpitch_amd.synth import make_problem, pdgp_from_problem
    model = pdgp_from_problem(make_problem(32768, 512, 12, num_partials=20, seed=0))
    model.za.fixed = True; model.zc.fixed = True
    model._pack()
    h = model._handle
    h.check(h.lib.gp_pdgp_set_overlap(model._plan, 0))
    lib = h.lib
    fn = lib.gp_debug_strip_stamps
    fn.restype = C.c_int
    buf = (C.c_ulonglong * (6 * 65536))()
    for _ in range(2):
        model._elbo(True, sync=False)
    torch.cuda.synchronize()
    fn(buf, 65536)              # drop warm-up records
    model._elbo(True, sync=False)
    torch.cuda.synchronize()
    n = fn(buf, 65536)
    a = np.frombuffer(buf, dtype=np.uint64).reshape(-1, 6)[:n].astype(np.int64)
    print("records", n)
    for key in sorted(set(a[:, 4])):
        r = a[a[:, 4] == key]
        pro, loop, epi = r[:, 1] - r[:, 0], r[:, 2] - r[:, 1], r[:, 3] - r[:, 2]
        nkt = r[0, 5]
        print("role %d row-block %d: %5d workgroups, K-tiles %2d | prologue %6.0f  loop %7.0f (%.0f per K-tile)  epilogue %6.0f  | total %7.0f cycles (s_memtime ticks, median)"
              % (key // 100, key % 100, len(r), nkt, np.median(pro), np.median(loop), np.median(loop) / nkt, np.median(epi),
                 np.median(r[:, 3] - r[:, 0])))


if __name__ == "__main__":
    main()
