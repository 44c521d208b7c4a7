"""ctypes binding of libgpitch_hip.so (include/gpitch_abi.h).

The library is the product: there is NO CPU fallback.  Importing this module loads the shared
library (failing loudly if it has not been built); creating a handle fails loudly when no gfx950
device is present.  torch is used only to own device memory and to hand out raw pointers.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# GPITCH_AMD_LIB: another build of the same library (same-box A/B measurements of kernel variants)
LIB_PATH = os.environ.get("GPITCH_AMD_LIB") or os.path.join(_HERE, "libgpitch_hip.so")

GP_OK, GP_ERR_BAD_ARG, GP_ERR_NOT_PD, GP_ERR_HIP, GP_ERR_NO_DEVICE, GP_ERR_WORKSPACE, GP_ERR_UNSUPPORTED = range(7)

(KERN_MATERN12, KERN_MATERN32, KERN_MATERN52, KERN_RBF, KERN_MERCER_MATERN12SM, KERN_MATERN12SM, KERN_MATERN32SM,
 KERN_MERCER_MATERN52SM) = range(8)
NLIN_LOGISTIC, NLIN_SOFTPLUS, NLIN_GAUSS = range(3)
(TIMER_KUF_BUILD, TIMER_COND_A, TIMER_COND_LTA, TIMER_NT_GEMM, TIMER_KUF_BAR, TIMER_CHOL, TIMER_LIK, TIMER_SMALL_GEMM,
 TIMER_HYPER, TIMER_KUF_BUILD_SM) = range(10)
TIMER_NAMES = ["kuf_build", "cond_A", "cond_LTA", "nt_gemm", "kuf_bar", "chol", "lik", "small_gemm", "hyper",
               "kuf_build_sm"]

# every symbol include/gpitch_abi.h declares
ABI_SYMBOLS = [
    "gp_create", "gp_destroy", "gp_sync", "gp_last_error", "gp_abi_version", "gp_last_not_pd_index",
    "gp_kernel_build", "gp_kernel_build_f32", "gp_kernel_diag", "gp_chol_workspace_bytes", "gp_kuu_cholesky", "gp_cholesky_inplace",
    "gp_conditional_workspace_bytes", "gp_conditional_diag", "gp_conditional_diag_f32", "gp_conditional_diag_f32w", "gp_conditional_full_workspace_bytes", "gp_conditional_full", "gp_gauss_kl_workspace_bytes", "gp_gauss_kl", "gp_gauss_kl_matrix", "gp_mpd_varexp",
    "gp_pdgp_create", "gp_pdgp_destroy", "gp_pdgp_num_params", "gp_pdgp_layout", "gp_pdgp_workspace_bytes",
    "gp_pdgp_set_workspace", "gp_pdgp_set_precision", "gp_pdgp_set_gp_precision", "gp_pdgp_set_grad_needs", "gp_pdgp_set_overlap", "gp_pdgp_elbo", "gp_pdgp_elbo_begin", "gp_pdgp_elbo_end", "gp_pdgp_create_subset", "gp_pdgp_cond_begin", "gp_pdgp_cond_end", "gp_pdgp_predict", "gp_pdgp_predict_reuse",
    "gp_overlap_merge", "gp_transform_register_logistic", "gp_transform_forward", "gp_transform_backward", "gp_poll_not_pd", "gp_check_not_pd", "gp_take_not_pd", "gp_adam_step",
    "gp_sgpr_create", "gp_sgpr_destroy", "gp_sgpr_num_params", "gp_sgpr_workspace_bytes", "gp_sgpr_set_workspace", "gp_sgpr_set_precision",
    "gp_sgpr_bound", "gp_sgpr_bound_grad", "gp_sgpr_residual_grad", "gp_sgpr_exchange_doubles", "gp_sgpr_bound_begin", "gp_sgpr_bound_end", "gp_sgpr_set_graphs", "gp_sgpr_eval_counts", "gp_sgpr_predict_f", "gp_sgpr_predict_f_full", "gp_sgpr_predict_source_full", "gp_sgpr_predict_source_workspace_bytes", "gp_sgpr_predict_source",
    "gp_sgprb_create", "gp_sgprb_destroy", "gp_sgprb_num_params", "gp_sgprb_num_windows", "gp_sgprb_workspace_bytes",
    "gp_sgprb_set_workspace", "gp_sgprb_bound_grad", "gp_sgprb_set_graphs", "gp_sgprb_eval_counts",
    "gp_sgprb_predict_f", "gp_sgprb_predict_source_workspace_bytes", "gp_sgprb_predict_source",
    "gp_timers_enable", "gp_timers_reset", "gp_timers_read",
    "gp_comm_unique_id", "gp_comm_create", "gp_comm_destroy", "gp_comm_world", "gp_comm_rank", "gp_comm_allreduce_sum",
    "gp_pdgp_elbo_pitch_sharded", "gp_pdgp_elbo_gp_sharded", "gp_sgpr_bound_grad_sharded",
]


class GpitchError(RuntimeError):
    def __init__(self, status, msg):
        RuntimeError.__init__(self, "libgpitch_hip status %d: %s" % (status, msg))
        self.status = status


class NotPositiveDefiniteError(GpitchError):
    """Cholesky failure (the reference surfaces a TF InvalidArgumentError here)."""


class KernelDesc(C.Structure):
    _fields_ = [("type", C.c_int32), ("num_partials", C.c_int32), ("theta", C.c_void_p)]


class PdgpConfig(C.Structure):
    _fields_ = [("num_sources", C.c_int32), ("whiten", C.c_int32), ("nlin", C.c_int32), ("max_batch", C.c_int32),
                ("M_act", C.POINTER(C.c_int32)), ("M_com", C.POINTER(C.c_int32)),
                ("kern_type_act", C.POINTER(C.c_int32)), ("kern_type_com", C.POINTER(C.c_int32)),
                ("partials_act", C.POINTER(C.c_int32)), ("partials_com", C.POINTER(C.c_int32)),
                ("jitter", C.c_double)]


class AdamArgs(C.Structure):
    """gp_adam_args: gp_adam_step's arguments for the sharded one-call steps"""
    _fields_ = [("free_state", C.c_void_p), ("tcode", C.c_void_p), ("m", C.c_void_p), ("v", C.c_void_p),
                ("nparams", C.c_int64), ("t", C.c_int64), ("lr", C.c_double), ("beta1", C.c_double), ("beta2", C.c_double),
                ("eps", C.c_double)]


class SgprConfig(C.Structure):
    _fields_ = [("num_kernels", C.c_int32), ("max_N", C.c_int32), ("M", C.c_int32),
                ("kern_type", C.POINTER(C.c_int32)), ("partials", C.POINTER(C.c_int32)),
                ("jitter", C.c_double), ("reg", C.c_int32)]


_lib = None


def precision_bits(float_type):
    """64 or 32 from what a caller passes as the reference's `float_type` setting (pdgp.py:13): None / np.float64 /
    'float64' / 64 -> 64; np.float32 / 'float32' / 32 -> 32."""
    if float_type is None:
        return 64
    if float_type in (32, 64):
        return int(float_type)
    bits = np.dtype(float_type).itemsize * 8
    if bits not in (32, 64):
        raise ValueError("float_type must be float64 or float32")
    return bits


def load_library():
    """dlopen the C-ABI library.  Raises (never falls back) when it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    # One HIP runtime per process: torch's wheel bundles its own libamdhip64 / libhsa-runtime64, and a
    # second copy (the system one this .so was linked against) cannot see the device once torch's is
    # live.  Importing torch first makes the dynamic loader bind our DT_NEEDED libamdhip64.so.7 to the
    # copy that is already loaded, so device pointers, streams and events are shared with torch.
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise ImportError("gpitch_amd: %s not found — build it with `python -c 'import __graft_entry__ as g; "
                          "g.build()'` (hipcc --offload-arch=gfx950); there is no CPU fallback" % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    vp, i32, i64, dbl, sz = C.c_void_p, C.c_int32, C.c_int64, C.c_double, C.c_size_t
    KD = C.POINTER(KernelDesc)
    sig = {
        "gp_create": (i32, [i32, vp, C.POINTER(vp)]),
        "gp_destroy": (i32, [vp]),
        "gp_sync": (i32, [vp]),
        "gp_last_error": (C.c_char_p, [vp]),
        "gp_abi_version": (i32, []),
        "gp_last_not_pd_index": (i32, [vp]),
        "gp_kernel_build": (i32, [vp, KD, vp, i32, vp, i32, vp, i64, i32]),
        "gp_kernel_build_f32": (i32, [vp, KD, vp, i32, vp, i32, vp, i64, i32]),
        "gp_kernel_diag": (i32, [vp, KD, i32, vp, i32]),
        "gp_chol_workspace_bytes": (sz, [i32]),
        "gp_kuu_cholesky": (i32, [vp, KD, vp, i32, dbl, vp, vp, vp, sz]),
        "gp_cholesky_inplace": (i32, [vp, vp, i32, i64]),
        "gp_conditional_workspace_bytes": (sz, [i32, i32]),
        "gp_conditional_diag": (i32, [vp, KD, vp, i32, vp, i32, vp, vp, i32, dbl, vp, vp, vp, sz]),
        "gp_conditional_full_workspace_bytes": (sz, [i32, i32]),
        "gp_conditional_full": (i32, [vp, KD, vp, i32, vp, i32, vp, vp, i32, dbl, vp, vp, vp, sz]),
        "gp_conditional_diag_f32": (i32, [vp, KD, vp, i32, vp, i32, vp, vp, dbl, vp, vp, vp, sz]),
        "gp_conditional_diag_f32w": (i32, [vp, KD, vp, i32, vp, i32, vp, vp, i32, dbl, vp, vp, vp, sz]),
        "gp_gauss_kl_workspace_bytes": (sz, [i32, i32]),
        "gp_gauss_kl": (i32, [vp, vp, vp, i32, KD, vp, dbl, C.POINTER(dbl), vp, sz]),
        "gp_gauss_kl_matrix": (i32, [vp, vp, vp, i32, vp, C.POINTER(dbl), vp, sz]),
        "gp_mpd_varexp": (i32, [vp, vp, vp, vp, i32, i32, i32, vp, vp, C.POINTER(dbl)]),
        "gp_pdgp_create": (i32, [vp, C.POINTER(PdgpConfig), C.POINTER(vp)]),
        "gp_pdgp_destroy": (i32, [vp]),
        "gp_pdgp_num_params": (i64, [vp]),
        "gp_pdgp_layout": (i32, [vp, i32, C.POINTER(i64), C.POINTER(i64), C.POINTER(i64), C.POINTER(i64)]),
        "gp_pdgp_workspace_bytes": (sz, [vp]),
        "gp_pdgp_set_workspace": (i32, [vp, vp, sz]),
        "gp_pdgp_set_precision": (i32, [vp, i32]),
        "gp_pdgp_set_gp_precision": (i32, [vp, vp, i32]),
        "gp_pdgp_set_grad_needs": (i32, [vp, i32, i32, i32]),
        "gp_pdgp_set_overlap": (i32, [vp, i32]),
        "gp_pdgp_elbo": (i32, [vp, vp, vp, vp, i32, dbl, vp, C.POINTER(dbl), vp]),
        "gp_pdgp_elbo_begin": (i32, [vp, vp, vp, vp, i32, vp, vp]),
        "gp_pdgp_elbo_end": (i32, [vp, vp, vp, vp, i32, dbl, vp, vp, C.POINTER(dbl), vp]),
        "gp_pdgp_create_subset": (i32, [vp, C.POINTER(PdgpConfig), C.POINTER(i32), i32, C.POINTER(vp)]),
        "gp_pdgp_cond_begin": (i32, [vp, vp, vp, i32, vp, vp, vp, vp]),
        "gp_pdgp_cond_end": (i32, [vp, vp, vp, vp, i32, dbl, vp, vp, vp, vp, C.POINTER(dbl), vp]),
        "gp_pdgp_predict": (i32, [vp, vp, vp, i32, vp, vp, vp]),
        "gp_pdgp_predict_reuse": (i32, [vp, vp, vp, i32, vp, vp, vp]),
        "gp_overlap_merge": (i32, [vp, vp, i32, i32, i64, i32, i32, vp]),
        "gp_transform_register_logistic": (i32, [vp, dbl, dbl, C.POINTER(C.c_uint8)]),
        "gp_transform_forward": (i32, [vp, vp, vp, i64, vp]),
        "gp_transform_backward": (i32, [vp, vp, vp, i64, vp]),
        "gp_poll_not_pd": (i32, [vp, C.POINTER(i32)]),
        "gp_check_not_pd": (i32, [vp]),
        "gp_take_not_pd": (i32, [vp, vp]),
        "gp_adam_step": (i32, [vp, vp, vp, vp, vp, vp, vp, i64, i64, dbl, dbl, dbl, dbl]),
        "gp_sgpr_create": (i32, [vp, C.POINTER(SgprConfig), C.POINTER(vp)]),
        "gp_sgpr_destroy": (i32, [vp]),
        "gp_sgpr_num_params": (i64, [vp]),
        "gp_sgpr_workspace_bytes": (sz, [vp]),
        "gp_sgpr_set_workspace": (i32, [vp, vp, sz]),
        "gp_sgpr_set_precision": (i32, [vp, i32]),
        "gp_sgpr_bound": (i32, [vp, vp, vp, vp, i32, vp, vp, C.POINTER(dbl)]),
        "gp_sgpr_bound_grad": (i32, [vp, vp, vp, vp, i32, vp, vp, C.POINTER(dbl), vp]),
        "gp_sgpr_residual_grad": (i32, [vp, vp, vp, i32, vp]),
        "gp_sgpr_exchange_doubles": (i64, [vp]),
        "gp_sgpr_bound_begin": (i32, [vp, vp, vp, vp, i32, vp, vp]),
        "gp_sgpr_bound_end": (i32, [vp, vp, vp, vp, i32, i64, vp, vp, vp, C.POINTER(dbl), vp, i32]),
        "gp_sgpr_set_graphs": (i32, [vp, i32]),
        "gp_sgpr_eval_counts": (i32, [vp, C.POINTER(i64), C.POINTER(i64), C.POINTER(i64)]),
        "gp_sgpr_predict_f": (i32, [vp, vp, vp, vp, i32, vp, vp, i32, vp, vp]),
        "gp_sgpr_predict_f_full": (i32, [vp, vp, vp, vp, i32, vp, vp, i32, vp, vp, vp]),
        "gp_sgpr_predict_source_full": (i32, [vp, vp, vp, vp, i32, vp, i32, vp, vp, vp, vp, sz]),
        "gp_sgpr_predict_source_workspace_bytes": (sz, [i32, i32]),
        "gp_sgpr_predict_source": (i32, [vp, vp, vp, vp, i32, vp, i32, vp, vp, vp, sz]),
        "gp_sgprb_create": (i32, [vp, C.POINTER(SgprConfig), i32, C.POINTER(vp)]),
        "gp_sgprb_destroy": (i32, [vp]),
        "gp_sgprb_num_params": (i64, [vp]),
        "gp_sgprb_num_windows": (i32, [vp]),
        "gp_sgprb_workspace_bytes": (sz, [vp]),
        "gp_sgprb_set_workspace": (i32, [vp, vp, sz]),
        "gp_sgprb_bound_grad": (i32, [vp, vp, vp, vp, vp, i32, vp, vp]),
        "gp_sgprb_set_graphs": (i32, [vp, i32]),
        "gp_sgprb_eval_counts": (i32, [vp, C.POINTER(i64), C.POINTER(i64), C.POINTER(i64)]),
        "gp_sgprb_predict_f": (i32, [vp, vp, vp, vp, vp, vp, i32, i32, vp, vp]),
        "gp_sgprb_predict_source_workspace_bytes": (sz, [vp, i32, i32]),
        "gp_sgprb_predict_source": (i32, [vp, vp, vp, vp, vp, i32, i32, vp, vp, vp, sz]),
        "gp_timers_enable": (i32, [vp, i32]),
        "gp_timers_reset": (i32, [vp]),
        "gp_timers_read": (i32, [vp, i32, C.POINTER(dbl), C.POINTER(i64)]),
        "gp_comm_unique_id": (i32, [vp]),
        "gp_comm_create": (i32, [vp, vp, i32, i32, C.POINTER(vp)]),
        "gp_comm_destroy": (i32, [vp]),
        "gp_comm_world": (i32, [vp]),
        "gp_comm_rank": (i32, [vp]),
        "gp_comm_allreduce_sum": (i32, [vp, vp, i64]),
        "gp_pdgp_elbo_pitch_sharded": (i32, [vp, vp, vp, vp, vp, i32, dbl, vp, vp, C.POINTER(dbl), vp, C.POINTER(AdamArgs)]),
        "gp_pdgp_elbo_gp_sharded": (i32, [vp, vp, vp, vp, vp, i32, dbl, i32, i32, vp, vp, vp, vp, C.POINTER(dbl), vp,
                                        C.POINTER(AdamArgs)]),
        "gp_sgpr_bound_grad_sharded": (i32, [vp, vp, vp, vp, vp, i32, i64, vp, vp, vp, C.POINTER(dbl), vp]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def _ptr(t):
    """raw device pointer of a torch tensor (or None)"""
    if t is None:
        return None
    return C.c_void_p(t.data_ptr())


class Handle(object):
    """One (process, GPU) library handle — replaces gpitch.init_settings / the global TF session
    (gpitch/methods.py:155-180)."""

    def __init__(self, device_id=0, use_torch_stream=True, stream=None):
        """stream: a torch.cuda.Stream the library launches on (the caller keeps torch's current stream on it, e.g.
        `with torch.cuda.stream(s):`, so that copies and kernels stay ordered); default: torch's current stream."""
        import torch
        self.lib = load_library()
        if not torch.cuda.is_available():
            raise GpitchError(GP_ERR_NO_DEVICE, "no HIP device visible: gpitch_amd needs an MI355X (gfx950); "
                                                "there is no CPU fallback")
        self.torch = torch
        self.device = torch.device("cuda", device_id)
        torch.cuda.set_device(self.device)
        self.stream = stream
        if stream is not None:
            stream = stream.cuda_stream
        else:
            stream = torch.cuda.current_stream(self.device).cuda_stream if use_torch_stream else None
        h = C.c_void_p()
        st = self.lib.gp_create(device_id, C.c_void_p(stream) if stream else None, C.byref(h))
        if st != GP_OK:
            raise GpitchError(st, "gp_create failed (is this an MI355X / gfx950 box?)")
        self.h = h

    def check(self, st):
        if st == GP_OK:
            return
        msg = self.lib.gp_last_error(self.h)
        msg = msg.decode() if msg else ""
        if st == GP_ERR_NOT_PD:
            raise NotPositiveDefiniteError(st, msg)
        raise GpitchError(st, msg)

    def sync(self):
        self.check(self.lib.gp_sync(self.h))

    def empty(self, *shape):
        return self.torch.empty(*shape, dtype=self.torch.float64, device=self.device)

    def zeros(self, *shape):
        return self.torch.zeros(*shape, dtype=self.torch.float64, device=self.device)

    def to_device(self, a):
        return self.torch.as_tensor(np.ascontiguousarray(np.asarray(a, dtype=np.float64)), device=self.device)

    def workspace(self, nbytes):
        # torch allocations are >= 512-byte aligned
        return self.torch.empty(int(nbytes) + 256, dtype=self.torch.uint8, device=self.device)

    def comm(self, world=None, rank=None):
        """The RCCL communicator of this handle over the ranks of the initialised torch.distributed group (or a one-rank
        communicator without one): rank 0 draws the 128-byte unique id (gp_comm_unique_id) and the group broadcasts it;
        every rank then calls gp_comm_create.  None when the group's backend is not nccl (the gloo rehearsals keep the
        torch.distributed exchange) or the library finds no RCCL.  Cached per handle."""
        if getattr(self, "_comm_done", False):
            return self._comm
        self._comm_done, self._comm = True, None
        import torch.distributed as dist
        grouped = dist.is_available() and dist.is_initialized()
        if grouped and dist.get_backend() != "nccl":
            return None
        w = dist.get_world_size() if grouped else 1
        r = dist.get_rank() if grouped else 0
        if (world is not None and world != w) or (rank is not None and rank != r):
            return None
        ident = (C.c_uint8 * 128)()
        if r == 0 and self.lib.gp_comm_unique_id(ident) != GP_OK:
            ident = None
        if grouped:
            t = self.torch.zeros(129, dtype=self.torch.uint8, device=self.device)
            if r == 0 and ident is not None:
                t[:128] = self.torch.frombuffer(bytearray(bytes(ident)), dtype=self.torch.uint8).to(self.device)
                t[128] = 1
            dist.broadcast(t, 0)
            host = t.cpu().numpy()
            if host[128] != 1:
                return None
            ident = (C.c_uint8 * 128)(*[int(v) for v in host[:128]])
        elif ident is None:
            return None
        c = C.c_void_p()
        st = self.lib.gp_comm_create(self.h, ident, r, w, C.byref(c))
        if st == GP_ERR_UNSUPPORTED:
            return None
        self.check(st)
        self._comm = c
        return c

    def timers(self):
        out = {}
        for i, name in enumerate(TIMER_NAMES):
            ms = C.c_double()
            n = C.c_int64()
            self.check(self.lib.gp_timers_read(self.h, i, C.byref(ms), C.byref(n)))
            out[name] = (ms.value, n.value)
        return out

    def close(self):
        if getattr(self, "_comm", None):
            self.lib.gp_comm_destroy(self._comm)
            self._comm = None
        if getattr(self, "h", None):
            self.lib.gp_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_default_handle = None


def default_handle():
    global _default_handle
    if _default_handle is None:
        _default_handle = Handle(int(os.environ.get("LOCAL_RANK", "0")))
    return _default_handle
