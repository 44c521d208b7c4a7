"""SGPRSS — sparse Gaussian process regression for source separation (gpitch/sgpr_ss.py:10-114), on the
HIP engine (csrc/sgpr.hip) through gp_sgpr_*.  Same constructor and method names as the reference:

    m = SGPRSS(X, Y, kern, Z, reg=False)       # kern: sum of pitch kernels (np.sum(list) -> Add, .kern_list)
    m.build_likelihood()                        # collapsed bound (sgpr_ss.py:29-71)
    mean, var = m.predict_f(Xnew)               # GPflow SGPR.build_predict (separation.py:306)
    smean, svar = m.predict_s(Xnew)             # per-source exact posterior (sgpr_ss.py:73-114)
"""
import ctypes as C

import numpy as np

from . import _lib
from .kernels import Add
from .param import DataHolder, Param, ParamList, Parameterized, transforms


class _Gaussian(Parameterized):
    """gpflow.likelihoods.Gaussian: variance Param (positive), initial value 1.0"""

    def __init__(self):
        self.variance = Param(1.0, transforms.positive)


class SGPRSS(Parameterized):
    def __init__(self, X, Y, kern, Z, mean_function=None, reg=False, handle=None, shard=None, float_type=None):
        """shard=(rank, world): ONE window spread over `world` GPUs by frames (one process each).  Every rank is
        built with the full X, Y; it uploads only its contiguous slice, and each bound / gradient evaluation
        exchanges one all-reduce of M^2 + M + 2 doubles (plus one of the small gradient vector): see
        include/gpitch_abi.h gp_sgpr_bound_begin / _end.  Predictions need the whole window on one GPU."""
        # mean_function (sgpr_ss.py:14,25): subtracted from Y in the bound (:40) and in the exact posterior (:90), added back
        # to predicted means (:95); gpitch_amd.mean_functions.Constant / Linear carry trainable Params as GPflow's do
        # (optimize() trains them unless .fixed), any other callable is a fixed function of the inputs
        if mean_function is not None and not callable(mean_function):
            raise TypeError("mean_function must be callable on an (n, 1) array (gpitch_amd.mean_functions)")
        object.__setattr__(self, "mean_function", mean_function)
        if not isinstance(kern, Add):
            kern = Add([kern])
        if reg:
            kern.var_vector = ParamList([k.variance for k in kern.kern_list])   # sgpr_ss.py:17-22
        Y = np.asarray(Y, dtype=np.float64)
        if Y.ndim != 2 or Y.shape[1] < 1:
            raise ValueError("Y must be N x D")
        self.X = DataHolder(np.asarray(X, dtype=np.float64).reshape(-1, 1))
        self.Y = DataHolder(Y)
        self.Z = DataHolder(np.asarray(Z, dtype=np.float64).reshape(-1, 1), on_shape_change='pass')   # :26
        self.kern = kern
        self.reg = reg
        self.likelihood = _Gaussian()
        # D = Y.shape[1] outputs share X, Z, the kernel and the noise (sgpr_ss.py:38 output_dim): the bound is the sum of
        # the D one-column bounds (every term of :55-62 is linear in the columns or carries the factor output_dim), so
        # D > 1 is D evaluations of the one-column launch sequence, summed on the device (_bound).  The reference's
        # callers only ever pass one column.
        self.num_latent = Y.shape[1]
        self._handle = handle
        self._plan = None
        self._plan_key = None
        if shard is not None:
            rank, world = int(shard[0]), int(shard[1])
            if not (0 <= rank < world):
                raise ValueError("shard=(rank, world) needs 0 <= rank < world")
            shard = (rank, world)
        object.__setattr__(self, "_shard", shard)
        # float_type: the reference's dtype setting (sgpr_ss.py:7 float_type = settings.dtypes.float_type); float32 puts
        # the M x N strips and the strip products on the float32 matrix path (gp_sgpr_set_precision)
        object.__setattr__(self, "_bits", _lib.precision_bits(float_type))

    def _frames(self):
        """frame indices this rank holds (all of them when unsharded): contiguous, sizes differ by at most one"""
        N = self.X.shape[0]
        if not self._shard:
            return slice(0, N)
        rank, world = self._shard
        base, extra = divmod(N, world)
        lo = rank * base + min(rank, extra)
        return slice(lo, lo + base + (1 if rank < extra else 0))

    # data can be swapped between windows (transcription.py:253-263: model.X = x; model.Y = 20*y; model.Z = z)
    def __setattr__(self, name, value):
        cur = self.__dict__.get(name)
        if isinstance(cur, DataHolder) and not isinstance(value, DataHolder):
            value = np.asarray(value, dtype=np.float64)
            if name == "Y" and value.ndim == 2:
                object.__setattr__(self, name, DataHolder(value))
                object.__setattr__(self, "num_latent", value.shape[1])
            else:
                object.__setattr__(self, name, DataHolder(value.reshape(-1, 1)))
        else:
            Parameterized.__setattr__(self, name, value)

    # ------------------------------------------------------------------------------------------
    def _compile(self, n_pred=0):
        h = self._handle = self._handle or _lib.default_handle()
        kl = self.kern.kern_list
        fr = self._frames()
        N, M = fr.stop - fr.start, self.Z.shape[0]
        if N < 1:
            raise ValueError("more ranks than frames")
        maxN = max(N, n_pred)
        key = (maxN, M, tuple((k.type_code, int(k.num_partials)) for k in kl), bool(self.reg))
        if self._plan is not None and self._plan_key == key:
            return
        self._destroy()
        P = len(kl)
        i32 = C.c_int32 * P
        self._keep = (i32(*[k.type_code for k in kl]), i32(*[int(k.num_partials) for k in kl]))
        cfg = _lib.SgprConfig(P, maxN, M, self._keep[0], self._keep[1], 1e-6, int(bool(self.reg)))
        plan = C.c_void_p()
        h.check(h.lib.gp_sgpr_create(h.h, C.byref(cfg), C.byref(plan)))
        self._plan, self._plan_key = plan, key
        if self._bits == 32:
            h.check(h.lib.gp_sgpr_set_precision(plan, 32))
        self._ws = h.workspace(h.lib.gp_sgpr_workspace_bytes(plan))
        h.check(h.lib.gp_sgpr_set_workspace(plan, self._ws.data_ptr(), self._ws.numel()))
        self._nparams = int(h.lib.gp_sgpr_num_params(plan))
        self._bound_dev = h.zeros(1)

    def _dev(self, name, host):
        """device copy of `host` in a buffer that is reused while the size stays the same: stable pointers let the
        engine keep its descriptors and replay the recorded evaluation graph across evaluations and windows"""
        h = self._handle
        host = np.ascontiguousarray(np.asarray(host, dtype=np.float64).reshape(-1))
        cur = self.__dict__.get(name)
        if cur is None or cur.numel() != host.size or cur.device != h.device:
            cur = h.empty(host.size)
            object.__setattr__(self, name, cur)
        cur.copy_(h.torch.as_tensor(host))
        return cur

    def _pack(self):
        vec = [self.likelihood.variance.value.reshape(-1)]
        for k in self.kern.kern_list:
            vec.append(k.theta())
        host = np.concatenate(vec)
        assert host.size == self._nparams
        fr = self._frames()
        self._dev("_params", host)
        self._dev("_Xd", self.X._array[fr])
        self._upload_err()
        self._dev("_Zd", self.Z._array)
        object.__setattr__(self, "_n_local", fr.stop - fr.start)
        object.__setattr__(self, "_obj_state", None)      # Param values / .fixed flags may have changed

    def _err_host(self):
        """err = Y - mean_function(X) on this rank's frames (sgpr_ss.py:40); a mean function of one column is
        subtracted from every column"""
        fr = self._frames()
        yv = self.Y._array[fr]
        if self.mean_function is not None:
            mv = np.asarray(self.mean_function(self.X._array[fr]), dtype=np.float64).reshape(yv.shape[0], -1)
            yv = yv - mv
        return yv

    def _upload_err(self):
        """the residual columns on the device: `_Yd` is the column the library reads (a stable pointer: the recorded
        launch sequence stays valid); with D > 1 `_Yall` holds all of them, column-major, and _bound copies each in turn"""
        err = self._err_host()
        if self.num_latent == 1:
            self._dev("_Yd", err)
            object.__setattr__(self, "_Yall", None)
            return
        self._dev("_Yall", np.ascontiguousarray(err.T))
        self._dev("_Yd", err[:, 0])

    def _columns(self):
        """iterate over the output columns with `_Yd` holding the current one"""
        D = self.num_latent
        if D == 1:
            yield 0
            return
        n = self._Yd.numel()
        for d in range(D):
            self._Yd.copy_(self._Yall[d * n:(d + 1) * n])
            yield d

    def _mean_params(self):
        """trainable Params of the mean function (gpflow.mean_functions.Constant.c, Linear.A / .b), [] for a plain callable"""
        mf = self.mean_function
        return list(mf.params()) if (mf is not None and hasattr(mf, "params") and hasattr(mf, "grad_from_residual")) else []

    def _bound(self, grad=None, per_column=None):
        """the bound (and its gradient into the device vector `grad`): the sum over the D output columns of the
        one-column evaluation (per_column(d) is called after each, while the plan still holds that column's state)"""
        D = self.num_latent
        if D == 1:
            v = self._bound_column(grad)
            if per_column is not None:
                per_column(0)
            return v
        h = self._handle
        total, gsum = 0.0, (None if grad is None else h.zeros(grad.numel()))
        for d in self._columns():
            total += self._bound_column(grad)
            if grad is not None:
                gsum += grad
            if per_column is not None:
                per_column(d)
        if self.reg:
            # every column's bound carries the whole L1 term -beta sum |variance_k| (sgpr_ss.py:64-68): keep one
            beta, off = 1000., 1
            corr = np.zeros(self._nparams)
            for k in self.kern.kern_list:
                th = k.theta()                    # th[0]: the kernel's variance in the library's layout
                total += (D - 1) * beta * abs(th[0])
                corr[off] = (D - 1) * beta * np.sign(th[0])
                off += th.size
            if grad is not None:
                gsum += h.to_device(corr)
        if grad is not None:
            grad.copy_(gsum)
        return total

    def _bound_column(self, grad=None):
        """one evaluation of the one-column bound (and gradient into the device vector `grad`) on this model's frames, for
        the residual column in `_Yd`; the frame-sharded form runs begin -> all-reduce -> end -> all-reduce(grad)"""
        h = self._handle
        out = C.c_double()
        n = self._n_local
        args = (self._plan, self._params.data_ptr(), self._Xd.data_ptr(), self._Yd.data_ptr(), n)
        if not self._shard:
            if grad is None:
                h.check(h.lib.gp_sgpr_bound(*args, self._Zd.data_ptr(), self._bound_dev.data_ptr(), C.byref(out)))
            else:
                h.check(h.lib.gp_sgpr_bound_grad(*args, self._Zd.data_ptr(), self._bound_dev.data_ptr(), C.byref(out),
                                                 grad.data_ptr()))
            return out.value
        from .dist import allreduce_sum_
        xchg = self.__dict__.get("_xchg")
        need = int(h.lib.gp_sgpr_exchange_doubles(self._plan))
        if xchg is None or xchg.numel() != need:
            xchg = h.empty(need)
            object.__setattr__(self, "_xchg", xchg)
        comm = self.__dict__.get("_comm_cache", False)
        if comm is False:
            import torch.distributed as dist
            world = self._shard[1]
            grouped = dist.is_available() and dist.is_initialized()
            ok = (grouped and dist.get_backend() == "nccl" and dist.get_world_size() == world) or (not grouped and world == 1)
            comm = h.comm() if ok else None
            object.__setattr__(self, "_comm_cache", comm)
        if comm is not None:
            # an RCCL group: begin -> all-reduce -> end -> all-reduce(grad) inside ONE library call (gp_sgpr_bound_grad_sharded)
            h.check(h.lib.gp_sgpr_bound_grad_sharded(args[0], comm, *args[1:], self.X.shape[0], self._Zd.data_ptr(), xchg.data_ptr(),
                                                     self._bound_dev.data_ptr(), C.byref(out),
                                                     None if grad is None else grad.data_ptr()))
            return out.value
        h.check(h.lib.gp_sgpr_bound_begin(*args, self._Zd.data_ptr(), xchg.data_ptr()))
        allreduce_sum_(xchg)
        h.check(h.lib.gp_sgpr_bound_end(*args, self.X.shape[0], self._Zd.data_ptr(), xchg.data_ptr(),
                                        self._bound_dev.data_ptr(), C.byref(out),
                                        None if grad is None else grad.data_ptr(), int(self._shard[0] == 0)))
        if grad is not None:
            allreduce_sum_(grad)
        return out.value

    def build_likelihood(self):
        """the bound on the marginal likelihood (sgpr_ss.py:29-71)"""
        self._compile()
        self._pack()
        return self._bound()

    def compute_log_likelihood(self):
        return self.build_likelihood()

    def predict_f(self, Xnew, full_cov=False):
        """mean and variance of the mixture at Xnew (GPflow 0.5 SGPR.build_predict; full_cov=True — `predict_f_full_cov`
        in GPflow — returns the n x n x 1 covariance)"""
        if self._shard:
            raise NotImplementedError("predictions of a frame-sharded window: build the model unsharded on one GPU")
        Xnew = np.asarray(Xnew, dtype=np.float64).reshape(-1)
        n = Xnew.size
        self._compile(n_pred=n)
        self._pack()
        h = self._handle
        xs = h.to_device(Xnew)
        mean, var = h.empty(n), h.empty(n)
        cov = h.empty(n, n) if full_cov else None
        D = self.num_latent
        mu = np.empty((n, D))
        for d in self._columns():                   # the mean is linear in the column, the variance does not see it
            if full_cov:
                h.check(h.lib.gp_sgpr_predict_f_full(self._plan, self._params.data_ptr(), self._Xd.data_ptr(),
                                                     self._Yd.data_ptr(), self.X.shape[0], self._Zd.data_ptr(), xs.data_ptr(),
                                                     n, mean.data_ptr(), var.data_ptr(), cov.data_ptr()))
            else:
                h.check(h.lib.gp_sgpr_predict_f(self._plan, self._params.data_ptr(), self._Xd.data_ptr(), self._Yd.data_ptr(),
                                                self.X.shape[0], self._Zd.data_ptr(), xs.data_ptr(), n, mean.data_ptr(),
                                                var.data_ptr()))
            mu[:, d] = mean.cpu().numpy()
        if self.mean_function is not None:          # SGPR.build_predict: + mean_function(Xnew)
            mu = mu + np.asarray(self.mean_function(Xnew.reshape(-1, 1)), dtype=np.float64).reshape(n, -1)
        if full_cov:                                # GPflow tiles the covariance over the D outputs
            return mu, np.tile(cov.cpu().numpy().reshape(n, n, 1), (1, 1, D))
        return mu, np.tile(var.cpu().numpy().reshape(-1, 1), (1, D))

    def predict_f_full_cov(self, Xnew):
        """GPflow's AutoFlow'd name for predict_f(full_cov=True)"""
        return self.predict_f(Xnew, full_cov=True)

    def build_predict_source(self, Xnew, full_cov=False):
        """p(source* | Y): exact GP on the N training frames, one posterior per kernel in kern_list
        (sgpr_ss.py:73-106; the variance uses the SUM kernel's Kdiag as the reference does)."""
        if self._shard:
            raise NotImplementedError("predictions of a frame-sharded window: build the model unsharded on one GPU")
        Xnew = np.asarray(Xnew, dtype=np.float64).reshape(-1)
        n, N, P = Xnew.size, self.X.shape[0], len(self.kern.kern_list)
        self._compile()
        self._pack()
        h = self._handle
        xs = h.to_device(Xnew)
        mean, var = h.empty(P, n), h.empty(P, n)
        ws = h.workspace(h.lib.gp_sgpr_predict_source_workspace_bytes(N, n))
        cov = h.empty(P, n, n) if full_cov else None
        D = self.num_latent
        m = np.empty((P, n, D))
        for d in self._columns():                   # one exact posterior per column (the reference's callers have one)
            if full_cov:
                h.check(h.lib.gp_sgpr_predict_source_full(self._plan, self._params.data_ptr(), self._Xd.data_ptr(),
                                                          self._Yd.data_ptr(), N, xs.data_ptr(), n, mean.data_ptr(),
                                                          var.data_ptr(), cov.data_ptr(), ws.data_ptr(), ws.numel()))
            else:
                h.check(h.lib.gp_sgpr_predict_source(self._plan, self._params.data_ptr(), self._Xd.data_ptr(),
                                                     self._Yd.data_ptr(), N, xs.data_ptr(), n, mean.data_ptr(), var.data_ptr(),
                                                     ws.data_ptr(), ws.numel()))
            m[:, :, d] = mean.cpu().numpy()
        v = var.cpu().numpy()
        if self.mean_function is not None:          # sgpr_ss.py:95 adds it to EVERY source's mean
            m = m + np.asarray(self.mean_function(Xnew.reshape(-1, 1)), dtype=np.float64).reshape(1, n, -1)
        if full_cov:                                # sgpr_ss.py:95-99: tiled to n x n x D
            c = cov.cpu().numpy()
            return [m[i] for i in range(P)], [np.tile(c[i].reshape(n, n, 1), (1, 1, D)) for i in range(P)]
        return [m[i] for i in range(P)], [np.tile(v[i].reshape(-1, 1), (1, D)) for i in range(P)]   # :101-102

    def predict_s(self, Xnew):
        """sgpr_ss.py:108-114"""
        return self.build_predict_source(Xnew)

    # ---- training: GPflow Model.optimize -> scipy L-BFGS-B on the free state (transcription.py:283) ----
    def _param_list(self):
        ps = [self.likelihood.variance]
        for k in self.kern.kern_list:
            ps.extend(k.theta_params())
        return ps + self._mean_params()     # (after the engine's parameter vector: they live on the host)

    def _objective_setup(self):
        """vectorised view of the free state for _objective (rebuilt by optimize(): .fixed flags, transforms and the
        values of the fixed Params are read once per optimisation, not once per evaluation)"""
        from .param import Identity, Log1pe, Logistic
        ps = self._param_list()
        free_idx = np.array([i for i, p in enumerate(ps) if not p.fixed], dtype=np.int64)
        kind = np.zeros(free_idx.size, dtype=np.int64)      # 0 identity, 1 positive, 2 logistic, 3 anything else
        lo = np.zeros(free_idx.size)
        span = np.ones(free_idx.size)
        other = {}
        for j, i in enumerate(free_idx):
            tr = ps[i].transform
            if isinstance(tr, Identity):
                kind[j] = 0
            elif isinstance(tr, Log1pe):
                kind[j] = 1
                lo[j] = tr._lower
            elif isinstance(tr, Logistic):
                kind[j] = 2
                lo[j], span[j] = tr.a, tr.b - tr.a
            else:
                kind[j] = 3
                other[j] = tr
        st = {"ps": ps, "free_idx": free_idx, "kind": kind, "lo": lo, "span": span, "other": other,
              "vals0": np.array([p.value[0] for p in ps])}
        object.__setattr__(self, "_obj_state", st)
        return st

    @staticmethod
    def _free_to_params(st, x):
        """forward transforms and their derivatives for the whole free vector at once"""
        x = np.asarray(x, dtype=np.float64)
        kind, lo, span = st["kind"], st["lo"], st["span"]
        sig = 1. / (1. + np.exp(-x))
        y = np.where(kind == 1, np.logaddexp(0., x) + lo, np.where(kind == 2, lo + span * sig, x))
        dy = np.where(kind == 1, sig, np.where(kind == 2, span * sig * (1. - sig), 1.))
        for j, tr in st["other"].items():       # (x may be one free vector or a (windows, n) stack of them)
            col = np.atleast_1d(x[..., j])
            y[..., j] = np.array([tr.forward(np.array([v]))[0] for v in col]).reshape(np.shape(x[..., j]))
            dy[..., j] = np.array([tr.dforward(np.array([v]))[0] for v in col]).reshape(np.shape(x[..., j]))
        return y, dy

    def _objective(self, x_free):
        """(-(bound), -d bound / d free-state) — GPflow Model._objective"""
        h = self._handle
        st = self.__dict__.get("_obj_state") or self._objective_setup()
        y, dy = self._free_to_params(st, x_free)
        vals = st["vals0"].copy()
        vals[st["free_idx"]] = y
        nd = self._nparams                        # the engine's parameters; the mean function's follow them
        mps = self._mean_params()
        train_mean = bool(mps) and bool(np.any(st["free_idx"] >= nd))
        if train_mean:                            # a new mean function: new residuals (same device buffer: the recorded
            for p_, v in zip(mps, vals[nd:]):     # launch sequence stays valid)
                p_._array = np.array([v], dtype=np.float64)
            self._upload_err()
        self._dev("_params", vals[:nd])
        grad = self.__dict__.get("_grad_dev")
        if grad is None or grad.numel() != nd:
            grad = h.empty(nd)
            object.__setattr__(self, "_grad_dev", grad)
        gm = np.zeros(len(mps))

        def mean_grad(d):
            # d bound / d err of the column just evaluated, chained through the mean function (shared by the columns)
            r = self.__dict__.get("_resid_dev")
            if r is None or r.numel() != self._n_local:
                r = h.empty(self._n_local)
                object.__setattr__(self, "_resid_dev", r)
            h.check(h.lib.gp_sgpr_residual_grad(self._plan, self._params.data_ptr(), self._Yd.data_ptr(), self._n_local,
                                                r.data_ptr()))
            gm[:] += np.concatenate(self.mean_function.grad_from_residual(self.X._array[self._frames()], r.cpu().numpy()))

        value = self._bound(grad, per_column=mean_grad if train_mean else None)
        g = grad.cpu().numpy()
        if mps:
            if train_mean:
                if self._shard:                   # a frame-sharded window: every rank holds its slice's share of the sums
                    from .dist import allreduce_sum_
                    gm = allreduce_sum_(h.torch.as_tensor(gm)).numpy()
            g = np.concatenate([g, gm])
        return -value, -(g[st["free_idx"]] * dy)

    def optimize(self, method='L-BFGS-B', tol=None, callback=None, maxiter=1000, disp=False, **kw):
        from scipy.optimize import minimize
        self._compile()
        self._pack()
        st = self._objective_setup()
        ps, free_idx = st["ps"], st["free_idx"]
        x0 = np.array([ps[i].transform.backward(ps[i].value)[0] for i in free_idx])
        res = minimize(self._objective, x0, jac=True, method=method, tol=tol, callback=callback,
                       options=dict(maxiter=maxiter, disp=disp))
        y, _ = self._free_to_params(st, res.x)
        for j, i in enumerate(free_idx):
            ps[i].value = y[j:j + 1]
        object.__setattr__(self, "_obj_state", None)
        return res

    def _destroy(self):
        if self._plan is not None and self._handle is not None and self._handle.h:
            self._handle.sync()
            self._handle.lib.gp_sgpr_destroy(self._plan)
        self._plan = None

    def __del__(self):
        try:
            self._destroy()
        except Exception:
            pass
