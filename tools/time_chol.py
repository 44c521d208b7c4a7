"""20 x gp_cholesky_inplace and 20 x gp_kuu_cholesky (factor + inverse) of one M x M matrix (TIME_M, default 512), for
`rocprofv3 --kernel-trace --stats -- python3 tools/time_chol.py` (the entry points synchronise, so the kernel durations
are read from the trace).  GPITCH_AMD_SWITCHES=chol_cluster=0 selects the one-workgroup kernels."""
import os, sys, ctypes as C
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from gpitch_amd._lib import Handle
import gpitch_amd._lib as _lib

M = int(os.environ.get("TIME_M", "512"))
h = Handle(0)
z = np.linspace(0, 2.0, M).reshape(-1, 1)
th = h.to_device(np.array([1.3, 0.3]))
d = _lib.KernelDesc(_lib.KERN_MATERN32, 0, th.data_ptr())
dz = h.to_device(z)
r = np.abs(z - z.T) / 0.3
K = 1.3 * (1 + np.sqrt(3) * r) * np.exp(-np.sqrt(3) * r) + 1e-6 * np.eye(M)
dK = h.to_device(K)
dA = h.empty(M, M)
L, W = h.empty(M, M), h.empty(M, M)
ws = h.workspace(h.lib.gp_chol_workspace_bytes(M))
for _ in range(20):
    dA.copy_(dK)
    h.check(h.lib.gp_cholesky_inplace(h.h, dA.data_ptr(), M, M))
for _ in range(20):
    h.check(h.lib.gp_kuu_cholesky(h.h, C.byref(d), dz.data_ptr(), M, 1e-6, L.data_ptr(), W.data_ptr(), ws.data_ptr(), ws.numel()))
Lg, Wg, La = L.cpu().numpy(), W.cpu().numpy(), dA.cpu().numpy()
print("M=%d switches=%r  |L L^T - K| %.2e (in place %.2e)  |W L - I| %.2e" % (
    M, os.environ.get("GPITCH_AMD_SWITCHES", ""), np.abs(Lg @ Lg.T - K).max(), np.abs(La @ La.T - K).max(), np.abs(Wg @ Lg - np.eye(M)).max()))
