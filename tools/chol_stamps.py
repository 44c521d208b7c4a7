"""Diagnostic: where the one-workgroup Kuu factorisation's time goes (build chol.hip with -DCH_STAMPS:
tools/build_variant.sh chstamps chol.hip -DCH_STAMPS).  Per 32-column panel of matrix 0: cycles of the panel product,
the trailing update and the look-ahead diagonal block (s_memtime ticks, 100 MHz -> x 10 ns).
    GPITCH_AMD_LIB=tools/ab/lib_chstamps.so python tools/chol_stamps.py [M]"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from gpitch_amd import _lib
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    h = _lib.default_handle()
    rng = np.random.RandomState(0)
    z = np.sort(rng.rand(M)) * 0.5
    K = np.exp(-np.abs(z[:, None] - z[None, :]) / 0.1) + 1e-6 * np.eye(M)
    fn = h.lib.gp_debug_chol_stamps
    fn.restype = C.c_int
    buf = (C.c_ulonglong * (5 * 64 + 64))()
    for rep in range(3):
        A = h.to_device(K.copy())
        torch.cuda.synchronize()
        t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
        t0.record(torch.cuda.current_stream())
        h.check(h.lib.gp_cholesky_inplace(h.h, A.data_ptr(), M, M))
        t1.record(torch.cuda.current_stream())
        torch.cuda.synchronize()
    L = np.tril(A.cpu().numpy())
    print("M=%d  max |L L^T - K| = %.3e   launch %.1f us" % (M, np.abs(L @ L.T - K).max(), t0.elapsed_time(t1) * 1e3))
    fn(buf)
    raw = np.frombuffer(buf, dtype=np.uint64).astype(np.int64)
    a = raw[:320].reshape(64, 5)
    ph = raw[320:].reshape(8, 8)
    T = M // 32 - 1
    tot = a[T - 1, 2] - a[0, 0]
    print("panel  product  update  (diag block inside the update)   [ticks of 10 ns]")
    for k in range(T):
        print("%3d   %6d  %6d   %6d" % (k, a[k, 1] - a[k, 0], a[k, 2] - a[k, 1], a[k, 4] - a[k, 3]))
    print("panel 0 update loop per wavefront: tiles | cycles per tile: operand reads + next draw / C loads issued | matrix products (drained) | stores (drained)")
    for w in range(8):
        n = max(ph[w, 3], 1)
        print("  wave %d: %3d | %6.0f | %6.0f | %6.0f" % (w, ph[w, 3], ph[w, 0] / n, ph[w, 1] / n, ph[w, 2] / n))
    print("sum: product %d  update %d  diag %d  | first panel start -> last update end %d ticks = %.1f us"
          % ((a[:T, 1] - a[:T, 0]).sum(), (a[:T, 2] - a[:T, 1]).sum(), (a[:T, 4] - a[:T, 3]).sum(), tot, tot * 0.01))


if __name__ == "__main__":
    main()
