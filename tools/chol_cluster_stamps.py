"""Diagnostic: where the workgroup-cluster factorisation's time goes (build chol_cluster.hip with -DCC_STAMPS:
tools/build_variant.sh ccstamps chol_cluster.hip -DCC_STAMPS).  Per period of the chain wavefront (s_memtime ticks):
diagonal block, publish, wait for the handed-over tiles, load them, wait for L(s+1, s-1) + two products, solve + publish,
last update; and when each worker finished each of its steps relative to the chain.
    GPITCH_AMD_LIB=tools/ab/lib_ccstamps.so python tools/chol_cluster_stamps.py [M] [inverse 0|1]"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from gpitch_amd import _lib
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    inv = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    h = _lib.default_handle()
    rng = np.random.RandomState(0)
    z = np.sort(rng.rand(M)) * 0.5
    K = np.exp(-np.abs(z[:, None] - z[None, :]) / 0.1) + 1e-6 * np.eye(M)
    fn = h.lib.gp_debug_chol_cluster_stamps
    fn.restype = C.c_int
    buf = (C.c_ulonglong * (16 * 8 + 64 * 16))()
    A = h.to_device(K.copy())
    W = h.empty(M, M)
    for rep in range(3):
        A.copy_(h.torch.as_tensor(K))
        if inv:
            th = h.to_device(np.array([1.3, 0.3])); d = _lib.KernelDesc(_lib.KERN_MATERN32, 0, th.data_ptr())
            dz = h.to_device(np.linspace(0, 2.0, M)); ws = h.workspace(h.lib.gp_chol_workspace_bytes(M))
            h.check(h.lib.gp_kuu_cholesky(h.h, C.byref(d), dz.data_ptr(), M, 1e-6, A.data_ptr(), W.data_ptr(), ws.data_ptr(), ws.numel()))
        else:
            st = h.lib.gp_cholesky_inplace(h.h, A.data_ptr(), M, M)
            if not os.environ.get("CC_PROBE_IGNORE_STATUS"):      # (timing probes with deliberately wrong arithmetic)
                h.check(st)
    fn(buf)
    raw = np.frombuffer(buf, dtype=np.uint64).astype(np.int64)
    T = M // 32
    a = raw[:128].reshape(16, 8)
    wk = raw[128:].reshape(64, 16)
    t0 = a[0, 0]
    print("M=%d inverse=%d  chain, ticks per period:" % (M, inv))
    print("  s   diag   wait+read Q   solve + X to LDS   last update | period   start")
    for s in range(T):
        r = a[s]
        if s + 1 < T:
            print("%3d %6d %10d %14d %16d | %6d %8d" % (s, r[1] - r[0], r[2] - r[1], r[3] - r[2], r[4] - r[3], r[4] - r[0], r[0] - t0))
        else:
            print("%3d %6d   (last)                                      | %6d %8d" % (s, r[1] - r[0], r[1] - r[0], r[0] - t0))
    print("chain total %d ticks" % (a[T - 1, 1] - t0))


if __name__ == "__main__":
    main()
