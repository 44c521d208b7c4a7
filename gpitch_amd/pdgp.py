"""Pdgp — pitch detection using Gaussian processes: same constructor, attributes and methods as
gpitch/pdgp.py:48-208, executed by the HIP engine (csrc/pdgp.hip, bwd.hip) through the C-ABI.

    m = Pdgp(x, y, z, kern=[[kact...], [kcom...]], whiten=True, minibatch_size=None, nlinfun=logistic_tf)
    m.za.fixed = True; m.zc.fixed = True
    m.optimize(method=AdamOptimizer(0.005), maxiter=1000)
    mean_a, var_a, mean_c, var_c, mean_src = m.predict_act_n_com(xtest)
"""
import ctypes as C

import numpy as np

from . import _lib
from .likelihoods import MpdLik
from .methods import logistic, logistic_tf, nlin_code
from .param import MinibatchData, Param, ParamList, Parameterized, param_version, sorted_params
from .train import AdamOptimizer, OptimizeResult

jitter = 1e-6   # gpflow settings.numerics.jitter_level (pdgp.py:14)


def predict_windowed(model, xnew, ws=1600):
    """gpitch/pdgp.py:17-44 (Python-2 integer division at :25 made explicit; the trailing remainder
    is dropped exactly as there).  NOTE the reference applies the *default* logistic for the source
    mean here (:35,:42) regardless of model.nlinfun; kept."""
    n = xnew.size
    P = model.num_sources
    m_a_l = [[] for _ in range(P)]
    v_a_l = [[] for _ in range(P)]
    m_c_l = [[] for _ in range(P)]
    v_c_l = [[] for _ in range(P)]
    m_s_l = [[] for _ in range(P)]
    for i in range(n // ws):
        x = xnew[i * ws:(i + 1) * ws].copy()
        m_a, v_a = model.predict_act(x)
        m_c, v_c = model.predict_com(x)
        for j in range(P):
            m_a_l[j].append(m_a[j].copy())
            v_a_l[j].append(v_a[j].copy())
            m_c_l[j].append(m_c[j].copy())
            v_c_l[j].append(v_c[j].copy())
    for j in range(P):
        m_a_l[j] = np.asarray(m_a_l[j]).reshape(-1, 1)
        v_a_l[j] = np.asarray(v_a_l[j]).reshape(-1, 1)
        m_c_l[j] = np.asarray(m_c_l[j]).reshape(-1, 1)
        v_c_l[j] = np.asarray(v_c_l[j]).reshape(-1, 1)
        m_s_l[j] = logistic(m_a_l[j]) * m_c_l[j]
    return m_a_l, v_a_l, m_c_l, v_c_l, m_s_l


def pitch_assignment(num_sources, world_size, rank):
    """pitches held by `rank` when one model is spread over `world_size` GPUs (pitch p -> rank p mod world)"""
    return list(range(rank, num_sources, world_size))


class Pdgp(Parameterized):
    def __init__(self, x, y, z, kern, whiten=True, minibatch_size=None, nlinfun=logistic_tf, handle=None,
                 max_predict_batch=None, shard=None, float_type=None):
        """Pitch detection using Gaussian process (pdgp.py:49-111).
        x, y: (N,1) arrays; z = [[za_0..], [zc_0..]]; kern = [[kern_act...], [kern_com...]].

        shard=(rank, world) spreads ONE model over `world` GPUs (one process each): this rank's engine plan holds
        the pitches {p : p mod world == rank} (both GPs of a pitch), the likelihood noise is replicated, and each
        ELBO evaluation exchanges one all-reduce of 3n+1 doubles (include/gpitch_abi.h: gp_pdgp_elbo_begin/_end).
        shard=("gp", rank, world) deals the 2P LATENT GPs instead (row g of [g_0..g_{P-1}, f_0..f_{P-1}] -> rank g mod
        world; 24 GPs on 8 GPUs = 3 each, where whole pitches give 2,2,2,2,1,1,1,1): one all-gather of (fmean, fvar)
        per step, the likelihood recomputed on every rank, backward local (gp_pdgp_cond_begin/_end).
        Every rank constructs the model with the same arguments.

        float_type: the reference's `settings.dtypes.float_type` (pdgp.py:13), np.float64 (default) or np.float32.
        With float32 the M x N strips (Kuf, Lm^-1 Kuf, Kuf_bar) and the four O(M^2 N) products are float32 on the
        float32 matrix cores; parameters, Kuu, its Cholesky factor, all reductions, the likelihood and the
        gradients stay float64 (include/gpitch_abi.h: gp_pdgp_set_precision); `whiten` and `float_type` are independent, as in
        the reference (pdgp.py:13,49,122-129).
        A pair `(float_type_act, float_type_com)` sets the strips' precision per group of latent GPs — activation GPs,
        component GPs — and `(np.float64, np.float32)` is the useful one: the activation GPs (Matern-3/2 on a 16-kHz
        grid, cond(Kuu) ~ 1e9) keep float64 strips, the component GPs run in float32 (gp_pdgp_set_gp_precision; float64
        latent GPs must precede float32 ones in the engine's order, so (float32, float64) is refused)."""
        if isinstance(float_type, (tuple, list)):
            if len(float_type) != 2:
                raise ValueError("float_type: one type or a pair (activation GPs, component GPs)")
            self._bits = tuple(_lib.precision_bits(f) for f in float_type)
            if self._bits[0] == self._bits[1]:
                self._bits = self._bits[0]
        else:
            self._bits = _lib.precision_bits(float_type)
        x = np.asarray(x, dtype=np.float64).reshape(-1, 1)
        y = np.asarray(y, dtype=np.float64).reshape(-1, 1)
        if minibatch_size is None:
            minibatch_size = x.shape[0]
        self.minibatch_size = int(minibatch_size)
        self.num_data = x.shape[0]
        self.num_sources = len(kern[0])
        self._gp_shard = None          # ("gp", rank, world): the latent GPs this rank holds, rows of [g_0.., f_0..]
        if shard is None:
            self._shard = None
            self._local = list(range(self.num_sources))
        elif len(shard) == 3 and shard[0] == "gp":
            rank, world = int(shard[1]), int(shard[2])
            if not (0 <= rank < world) or world > 2 * self.num_sources:
                raise ValueError('shard=("gp", rank, world) needs 0 <= rank < world <= 2 x number of sources')
            from .dist import gp_assignment
            self._shard = (rank, world)
            self._gp_shard = gp_assignment(2 * self.num_sources, world, rank)
            self._local = []           # no whole pitch lives here
        else:
            if len(shard) == 3 and shard[0] == "pitch":
                shard = shard[1:]
            rank, world = int(shard[0]), int(shard[1])
            if not (0 <= rank < world) or world > self.num_sources:
                raise ValueError("shard=(rank, world) needs 0 <= rank < world <= number of sources")
            self._shard = (rank, world)
            self._local = pitch_assignment(self.num_sources, world, rank)
        self.whiten = whiten
        self.nlinfun = nlinfun
        self.likelihood = MpdLik(nlinfun=self.nlinfun, num_sources=self.num_sources)
        self.x = MinibatchData(x, self.minibatch_size, np.random.RandomState(0))   # pdgp.py:76-77: same seed
        self.y = MinibatchData(y, self.minibatch_size, np.random.RandomState(0))
        self.kern_act = ParamList(kern[0])
        self.kern_com = ParamList(kern[1])
        self.num_inducing_a, self.num_inducing_c = [], []
        za_l, zc_l, q_mu_com_l, q_mu_act_l, q_sqrt_com_l, q_sqrt_act_l = [], [], [], [], [], []
        for i in range(self.num_sources):
            Ma, Mc = np.asarray(z[0][i]).size, np.asarray(z[1][i]).size
            self.num_inducing_a.append(Ma)
            self.num_inducing_c.append(Mc)
            za_l.append(Param(np.asarray(z[0][i], dtype=np.float64).reshape(-1, 1).copy()))
            zc_l.append(Param(np.asarray(z[1][i], dtype=np.float64).reshape(-1, 1).copy()))
            q_mu_act_l.append(Param(np.zeros((Ma, 1))))
            q_mu_com_l.append(Param(np.zeros((Mc, 1))))
            q_sqrt_act_l.append(Param(np.eye(Ma)[:, :, None].copy()))     # M x M x 1 (pdgp.py:102-103)
            q_sqrt_com_l.append(Param(np.eye(Mc)[:, :, None].copy()))
        self.za = ParamList(za_l)
        self.zc = ParamList(zc_l)
        self.q_mu_com = ParamList(q_mu_com_l)
        self.q_mu_act = ParamList(q_mu_act_l)
        self.q_sqrt_com = ParamList(q_sqrt_com_l)
        self.q_sqrt_act = ParamList(q_sqrt_act_l)
        # ---- engine state (created lazily; needs the GPU) ----
        self._handle = handle
        self._plan = None
        self._max_predict_batch = max_predict_batch
        self._adam_t = 0

    # ------------------------------------------------------------------------------------------
    # engine plumbing
    def _gps(self):
        """GP order of the engine: act then com of the pitches this rank holds (all of them when unsharded)"""
        out = []
        if self._gp_shard is not None:
            P = self.num_sources
            for g in self._gp_shard:
                i = g if g < P else g - P
                out.append((self.kern_act[i], self.za[i], self.q_mu_act[i], self.q_sqrt_act[i]) if g < P else
                           (self.kern_com[i], self.zc[i], self.q_mu_com[i], self.q_sqrt_com[i]))
            return out
        for i in self._local:
            out.append((self.kern_act[i], self.za[i], self.q_mu_act[i], self.q_sqrt_act[i]))
        for i in self._local:
            out.append((self.kern_com[i], self.zc[i], self.q_mu_com[i], self.q_sqrt_com[i]))
        return out

    def _compile(self):
        if self._plan is not None:
            return
        h = self._handle = self._handle or _lib.default_handle()
        loc = self._local if self._gp_shard is None else list(range(self.num_sources))   # (GP-sharded: cfg = whole model)
        P = len(loc)
        i32 = C.c_int32 * P
        self._cfg_keep = dict(
            M_act=i32(*[self.num_inducing_a[i] for i in loc]), M_com=i32(*[self.num_inducing_c[i] for i in loc]),
            kt_act=i32(*[self.kern_act[i].type_code for i in loc]),
            kt_com=i32(*[self.kern_com[i].type_code for i in loc]),
            np_act=i32(*[int(self.kern_act[i].num_partials) for i in loc]),
            np_com=i32(*[int(self.kern_com[i].num_partials) for i in loc]))
        k = self._cfg_keep
        self._max_batch = max(self.minibatch_size, self._max_predict_batch or min(self.num_data, 32768))
        cfg = _lib.PdgpConfig(P, int(bool(self.whiten)), nlin_code(self.nlinfun), self._max_batch,
                              k["M_act"], k["M_com"], k["kt_act"], k["kt_com"], k["np_act"], k["np_com"], jitter)
        plan = C.c_void_p()
        if self._gp_shard is not None:
            idx = (C.c_int32 * len(self._gp_shard))(*self._gp_shard)
            h.check(h.lib.gp_pdgp_create_subset(h.h, C.byref(cfg), idx, len(self._gp_shard), C.byref(plan)))
        else:
            h.check(h.lib.gp_pdgp_create(h.h, C.byref(cfg), C.byref(plan)))
        self._plan = plan
        if self._bits == 32:
            h.check(h.lib.gp_pdgp_set_precision(plan, 32))
        elif isinstance(self._bits, tuple):
            rows = list(range(2 * P)) if self._gp_shard is None else list(self._gp_shard)
            bits = [self._bits[0] if r < P else self._bits[1] for r in rows]
            h.check(h.lib.gp_pdgp_set_gp_precision(plan, (C.c_int32 * len(bits))(*bits), len(bits)))
        n = self._nparams = int(h.lib.gp_pdgp_num_params(plan))
        self._layout = []
        for g in range(2 * P if self._gp_shard is None else len(self._gp_shard)):
            o = [C.c_int64() for _ in range(4)]
            h.check(h.lib.gp_pdgp_layout(plan, g, *[C.byref(v) for v in o]))
            self._layout.append(tuple(v.value for v in o))
        t = h.torch
        self._params = h.zeros(n)
        self._free = h.zeros(n)
        self._grad = h.zeros(n)
        self._adam_m = h.zeros(n)
        self._adam_v = h.zeros(n)
        self._tcode = t.zeros(n, dtype=t.uint8, device=h.device)
        self._elbo_dev = h.zeros(2)     # [ELBO, sum of KL terms]
        self._ws = h.workspace(h.lib.gp_pdgp_workspace_bytes(plan))
        h.check(h.lib.gp_pdgp_set_workspace(plan, self._ws.data_ptr(), self._ws.numel()))
        self._x_dev = h.to_device(self.x._array.reshape(-1))
        self._y_dev = h.to_device(self.y._array.reshape(-1))
        self._xchg = h.zeros(3 * self._max_batch + 1) if (self._shard and self._gp_shard is None) else None
        if self._gp_shard is not None:
            from .dist import gp_exchange_layout
            _, blk = gp_exchange_layout(2 * self.num_sources, self._shard[1], self._max_batch)
            self._gp_send = h.zeros(blk)
            self._gp_recv = h.zeros(blk * self._shard[1])

    def _segments(self):
        """[(offset, Param)] of every Param in the flat vector"""
        segs = [(0, self.likelihood.variance)]
        for g, (kern, z, q_mu, q_sqrt) in enumerate(self._gps()):
            o_th, o_z, o_mu, o_sq = self._layout[g]
            for j, p in enumerate(kern.theta_params()):
                segs.append((o_th + j, p))
            segs += [(o_z, z), (o_mu, q_mu), (o_sq, q_sqrt)]
        return segs

    def _pack(self):
        """host Param values -> device parameter vector, free state and transform codes"""
        self._compile()
        h = self._handle
        host = np.zeros(self._nparams)
        tc = np.full(self._nparams, 2, dtype=np.uint8)   # padding slots: fixed
        for off, p in self._segments():
            v = p.value.reshape(-1)
            host[off:off + v.size] = v
            tc[off:off + v.size] = 2 if p.fixed else p.transform.device_code(h)
        self._params.copy_(h.torch.as_tensor(host))
        self._tcode.copy_(h.torch.as_tensor(tc))
        # `.fixed` Params drop out of the backward pass (GPflow removes them from the free state)
        for g, (kern, z, q_mu, q_sqrt) in enumerate(self._gps()):
            need_theta = any(not p.fixed for p in kern.theta_params())
            h.check(h.lib.gp_pdgp_set_grad_needs(self._plan, g, int(need_theta), int(not z.fixed)))
        h.check(h.lib.gp_transform_backward(h.h, self._params.data_ptr(), self._tcode.data_ptr(), self._nparams,
                                            self._free.data_ptr()))
        self._packed_key = self._host_key()

    def _host_key(self):
        """what the packed device copy depends on: every Param value (param_version) and the `.fixed` flags"""
        return (param_version(), tuple(p.fixed for _, p in self._segments()))

    def _free_index(self):
        """engine offsets of the entries of GPflow's free-state vector, in GPflow's order (param.sorted_params: Params by
        attribute name, `.fixed` ones absent) — the `x` / `jac` layout of optimize(), its callback and _objective()"""
        key = tuple(p.fixed for _, p in self._segments())
        memo = self.__dict__.get("_free_index_memo")
        if memo is None or memo[0] != key:
            off = {id(p): o for o, p in self._segments()}
            idx = [np.arange(off[id(p)], off[id(p)] + p.size) for p in sorted_params(self)
                   if not p.fixed and id(p) in off]
            idx = np.concatenate(idx) if idx else np.zeros(0, dtype=np.int64)
            memo = (key, idx, self._handle.torch.as_tensor(idx, device=self._handle.device))
            self._free_index_memo = memo
        return memo[1], memo[2]

    def _unpack(self):
        host = self._params.cpu().numpy()
        for off, p in self._segments():
            p.value = host[off:off + p.size]
        self._packed_key = self._host_key()

    def _batch(self):
        """fresh minibatch (x and y generators are seeded identically so rows stay paired: pdgp.py:76-77)"""
        # The batch is a SET of frames (the ELBO sums over it): handing it to the engine in time order changes nothing but
        # the order of that sum, and lets the covariance kernels factorise the envelope away from the diagonal band
        # (cov.hip: separable envelope) — with a shuffled batch every 64-column tile straddles the whole signal.
        n_all = int(self._x_dev.shape[0])
        if self.x.minibatch_size >= n_all and self.y.minibatch_size >= n_all:
            # minibatch_size = N (the benchmark configurations): GPflow draws rng.permutation(N)[:N] — every draw is the
            # whole data set, whatever the order; in time order that is the data as it lies in memory.  No draw, no sort,
            # no index upload, no gather: at 4-5 ms per step the two 32768-element permutations and the sort were 1.7 ms
            # of host time per step, as much as the rest of the step's launches (tools/host_profile.py).
            return self._x_dev, self._y_dev, n_all
        idx = self.x.next_indices()
        # the y generator is seeded like x's and drawn in lockstep (pdgp.py:76-77): it is advanced by copying the state
        # instead of drawing the same indices a second time
        if self.y.rng is not self.x.rng:
            if hasattr(self.x.rng, "get_state") and hasattr(self.y.rng, "set_state"):
                self.y.rng.set_state(self.x.rng.get_state())
            else:
                self.y.next_indices()        # a generator object without state access: draw the pair, as before
        idx = np.sort(idx, kind="stable")
        ti = self._upload_indices(idx)
        return self._x_dev.index_select(0, ti).contiguous(), self._y_dev.index_select(0, ti).contiguous(), idx.size

    def _upload_indices(self, idx):
        """The step's index vector goes up through a small ring of pinned host buffers with a non-blocking copy: a
        plain `as_tensor(idx, device=...)` is a synchronous copy from pageable memory, i.e. one host-device sync per
        step, after which the device sits idle until the host has issued the next step's first launches."""
        h = self._handle
        torch = h.torch
        if h.device.type != "cuda":
            return torch.as_tensor(idx, device=h.device)
        n = int(idx.size)
        ring = getattr(self, "_idx_ring", None)
        if ring is None or ring[0][0].numel() < n:
            cap = max(n, int(self.minibatch_size))
            ring = [(torch.empty(cap, dtype=torch.int64).pin_memory(),
                     torch.empty(cap, dtype=torch.int64, device=h.device), torch.cuda.Event()) for _ in range(3)]
            self._idx_ring, self._idx_pos = ring, 0
        pinned, dev, ev = ring[self._idx_pos % len(ring)]
        self._idx_pos += 1
        ev.synchronize()                     # the copy that last read this pinned buffer has run
        pinned[:n].numpy()[...] = idx
        dev[:n].copy_(pinned[:n], non_blocking=True)
        ev.record()
        return dev[:n]

    def _sharded_comm(self):
        """the handle's RCCL communicator when this sharded model can use the one-call forms (gp_pdgp_elbo_pitch_sharded /
        gp_pdgp_elbo_gp_sharded: begin -> exchange -> end [-> Adam] inside the library): an nccl process group of the
        model's world size (or no group at all for a one-rank model).  None: the torch.distributed exchange between the two
        stages (gloo rehearsals, emulated ranks)."""
        if "_comm_cache" not in self.__dict__:
            import torch.distributed as dist
            world = self._shard[1]
            grouped = dist.is_available() and dist.is_initialized()
            ok = (grouped and dist.get_backend() == "nccl" and dist.get_world_size() == world) or (not grouped and world == 1)
            object.__setattr__(self, "_comm_cache", self._handle.comm() if ok else None)
        return self._comm_cache

    def _elbo_one_call(self, comm, want_grad, sync, adam):
        """a sharded evaluation (and, with `adam`, the optimiser step behind it) as ONE library call"""
        h = self._handle
        xb, yb, n = self._batch()
        self._last_batch = (xb, yb)
        out = C.c_double()
        grad = self._grad.data_ptr() if want_grad else None
        aa = C.byref(adam) if adam is not None else None
        if self._gp_shard is not None:
            from .dist import gp_exchange_layout
            G, world = 2 * self.num_sources, self._shard[1]
            per, blk = gp_exchange_layout(G, world, n)
            full = self.__dict__.get("_gp_full")
            if full is None or full.numel() < 2 * G * n + 8:
                full = h.empty(2 * G * self._max_batch + 8)
                object.__setattr__(self, "_gp_full", full)
            h.check(h.lib.gp_pdgp_elbo_gp_sharded(self._plan, comm, self._params.data_ptr(), xb.data_ptr(), yb.data_ptr(), n,
                                                  float(self.num_data), G, len(self._gp_shard), self._gp_send.data_ptr(),
                                                  self._gp_recv.data_ptr(), full.data_ptr(), self._elbo_dev.data_ptr(),
                                                  C.byref(out) if sync else None, grad, aa))
        else:
            h.check(h.lib.gp_pdgp_elbo_pitch_sharded(self._plan, comm, self._params.data_ptr(), xb.data_ptr(), yb.data_ptr(), n,
                                                     float(self.num_data), self._xchg.data_ptr(), self._elbo_dev.data_ptr(),
                                                     C.byref(out) if sync else None, grad, aa))
        return out.value if sync else None

    def _elbo(self, want_grad, sync=True, adam=None):
        h = self._handle
        self._pred_state = None      # the engine drops its prediction factorisation on every ELBO evaluation
        if self._shard:
            comm = self._sharded_comm()
            if comm is not None:
                return self._elbo_one_call(comm, want_grad, sync, adam)
        assert adam is None
        if self._gp_shard is not None:
            from .dist import allgather_
            send = self._gp_begin(want_grad)
            recv = self._gp_recv[:send.numel() * self._shard[1]]
            allgather_(recv, send)
            return self._gp_end(want_grad, recv, sync)
        if self._shard:
            xchg = self._elbo_begin(want_grad)
            from .dist import allreduce_sum_
            allreduce_sum_(xchg)
            return self._elbo_end(want_grad, sync)
        xb, yb, n = self._batch()
        self._last_batch = (xb, yb)   # keep alive while the stream uses them
        out = C.c_double()
        h.check(h.lib.gp_pdgp_elbo(self._plan, self._params.data_ptr(), xb.data_ptr(), yb.data_ptr(), n,
                                   float(self.num_data), self._elbo_dev.data_ptr(), C.byref(out) if sync else None,
                                   self._grad.data_ptr() if want_grad else None))
        return out.value if sync else None

    def _elbo_begin(self, want_grad):
        """pitch-sharded stage 1: returns the exchange tensor [A | B | D | sum KL] (3n+1) to be summed over ranks"""
        h = self._handle
        self._pred_state = None
        xb, yb, n = self._batch()
        self._last_batch = (xb, yb)
        xchg = self._xchg[:3 * n + 1]
        h.check(h.lib.gp_pdgp_elbo_begin(self._plan, self._params.data_ptr(), xb.data_ptr(), yb.data_ptr(), n,
                                         self._grad.data_ptr() if want_grad else None, xchg.data_ptr()))
        return xchg

    def _gp_begin(self, want_grad):
        """GP-sharded stage 1: conditionals of this rank's latent GPs, written straight into the all-gather's send block
        [fmean rows | fvar rows | sum of the local KL terms] (dist.gp_exchange_layout)"""
        from .dist import gp_exchange_layout
        h = self._handle
        self._pred_state = None
        xb, yb, n = self._batch()
        self._last_batch = (xb, yb)
        per, blk = gp_exchange_layout(2 * self.num_sources, self._shard[1], n)
        send = self._gp_send[:blk]
        if len(self._gp_shard) < per:
            send.zero_()                  # this rank's last row slot is padding
        h.check(h.lib.gp_pdgp_cond_begin(self._plan, self._params.data_ptr(), xb.data_ptr(), n,
                                         self._grad.data_ptr() if want_grad else None, send.data_ptr(),
                                         send[per * n:].data_ptr(), send[2 * per * n:].data_ptr()))
        return send

    def _gp_end(self, want_grad, gathered, sync=True):
        """GP-sharded stage 2 on the all-gather's output (world x block): the whole model's likelihood, local backward"""
        from .dist import gp_assemble
        h = self._handle
        xb, yb = self._last_batch
        n = xb.numel()
        fm, fv, kl = gp_assemble(gathered, 2 * self.num_sources, self._shard[1], n)
        self._gp_keep = (fm, fv, kl)      # alive while the stream uses them
        out = C.c_double()
        h.check(h.lib.gp_pdgp_cond_end(self._plan, self._params.data_ptr(), xb.data_ptr(), yb.data_ptr(), n,
                                       float(self.num_data), fm.data_ptr(), fv.data_ptr(), kl.data_ptr(),
                                       self._elbo_dev.data_ptr(), C.byref(out) if sync else None,
                                       self._grad.data_ptr() if want_grad else None))
        return out.value if sync else None

    def _elbo_end(self, want_grad, sync=True):
        """pitch-sharded stage 2 on the rank-summed exchange tensor"""
        h = self._handle
        xb, yb = self._last_batch
        n = xb.numel()
        out = C.c_double()
        h.check(h.lib.gp_pdgp_elbo_end(self._plan, self._params.data_ptr(), xb.data_ptr(), yb.data_ptr(), n,
                                       float(self.num_data), self._xchg.data_ptr(), self._elbo_dev.data_ptr(),
                                       C.byref(out) if sync else None, self._grad.data_ptr() if want_grad else None))
        return out.value if sync else None

    # ------------------------------------------------------------------------------------------
    # reference API
    def build_prior_kl(self):
        """compute KL divergences (pdgp.py:113-131)"""
        from .conditionals import gauss_kl
        if not self.whiten or self._shard:
            # K = Kuu + jitter I (pdgp.py:126-129): evaluated by the engine next to the conditionals
            self._pack()
            self._elbo(False)
            return float(self._elbo_dev[1].item())
        kl = 0.
        for i in range(self.num_sources):
            kl += gauss_kl(self.q_mu_act[i].value, self.q_sqrt_act[i].value)
            kl += gauss_kl(self.q_mu_com[i].value, self.q_sqrt_com[i].value)
        return kl

    def build_likelihood(self):
        """Compute the objective function (pdgp.py:133-170): the ELBO on a fresh minibatch."""
        self._pack()
        return self._elbo(False)

    def compute_log_likelihood(self):
        return self.build_likelihood()

    def _objective(self, x_free):
        """GPflow Model._objective: (-(ELBO), -grad wrt the free state) in float64"""
        h = self._handle
        if self._plan is None or self.__dict__.get("_packed_key") != self._host_key():
            self._pack()       # (inside optimize()'s loop the device copy is ahead of the host Params and is left alone)
        idx, idx_dev = self._free_index()
        xf = np.zeros(self._nparams)
        xf[idx] = np.asarray(x_free, dtype=np.float64).reshape(-1)
        self._free.index_copy_(0, idx_dev, h.torch.as_tensor(xf[idx]).to(h.device))
        h.check(h.lib.gp_transform_forward(h.h, self._free.data_ptr(), self._tcode.data_ptr(), self._nparams,
                                           self._params.data_ptr()))
        f = self._elbo(True)
        g = self._grad.cpu().numpy()
        # chain rule through each Param's transform (fixed Params are not part of the free state)
        scale = np.zeros_like(g)
        for off, p in self._segments():
            if not p.fixed:
                scale[off:off + p.size] = p.transform.dforward(xf[off:off + p.size])
        g = g * scale
        return -f, -g[idx]

    def get_free_state(self):
        """GPflow Model.get_free_state: the non-fixed Params' unconstrained values, ordered by name"""
        self._pack()
        return self._free.index_select(0, self._free_index()[1]).cpu().numpy()

    def optimize(self, method='L-BFGS-B', tol=None, callback=None, maxiter=1000, disp=False, **kw):
        """GPflow Model.optimize.  `method` is an AdamOptimizer token (demo-modgp.py:44-45) or a
        scipy.optimize.minimize method name."""
        self._pack()
        h = self._handle
        if isinstance(method, AdamOptimizer):
            flag = C.c_int32(0)
            one_call = bool(self._shard) and self._sharded_comm() is not None
            for it in range(maxiter):
                self._adam_t += 1
                if one_call:
                    # a sharded model on an RCCL group: conditionals -> exchange -> likelihood + backward -> Adam, one call
                    aa = _lib.AdamArgs(self._free.data_ptr(), self._tcode.data_ptr(), self._adam_m.data_ptr(),
                                       self._adam_v.data_ptr(), self._nparams, self._adam_t, method.learning_rate,
                                       method.beta1, method.beta2, method.epsilon)
                    self._elbo(True, sync=False, adam=aa)
                else:
                    self._elbo(True, sync=False)
                    h.check(h.lib.gp_adam_step(h.h, self._free.data_ptr(), self._params.data_ptr(), self._grad.data_ptr(),
                                               self._tcode.data_ptr(), self._adam_m.data_ptr(), self._adam_v.data_ptr(),
                                               self._nparams, self._adam_t, method.learning_rate, method.beta1,
                                               method.beta2, method.epsilon))
                # a failed Cholesky freezes the optimiser state on the device (gp_adam_step); the loop itself stops at
                # the next poll that has seen the flag — no host synchronisation per step
                h.check(h.lib.gp_poll_not_pd(h.h, C.byref(flag)))
                if flag.value:
                    h.check(h.lib.gp_check_not_pd(h.h))     # raises NotPositiveDefiniteError with the pivot index
                if callback is not None:
                    callback(self._free.index_select(0, self._free_index()[1]).cpu().numpy())
            h.check(h.lib.gp_check_not_pd(h.h))             # a failure in the last few steps, not polled yet
            # GPflow evaluates the returned `fun`/`jac` on a fresh minibatch (demo_modgp.ipynb:140-146)
            x_final = self._free.index_select(0, self._free_index()[1]).cpu().numpy()
            f, g = self._objective(x_final)
            self._unpack()
            if self._shard:
                self.sync_params()
            return OptimizeResult(fun=f, jac=g, x=x_final, message='Finished iterations.', status='Finished iterations.',
                                  success=True)
        if self._shard:
            # each rank owns a different slice of the free state: only element-wise update rules (Adam) stay
            # consistent across ranks without exchanging it
            raise NotImplementedError("a pitch-sharded Pdgp is trained with AdamOptimizer")
        from scipy.optimize import minimize
        idx_dev = self._free_index()[1]
        x0 = self._free.index_select(0, idx_dev).cpu().numpy()
        res = minimize(self._objective, x0, jac=True, method=method, tol=tol, callback=callback,
                       options=dict(maxiter=maxiter, disp=disp))
        self._free.index_copy_(0, idx_dev, h.torch.as_tensor(res.x).to(h.device))
        h.check(h.lib.gp_transform_forward(h.h, self._free.data_ptr(), self._tcode.data_ptr(), self._nparams,
                                           self._params.data_ptr()))
        self._unpack()
        return res

    def _predict(self, xnew, want_source):
        h = self._handle
        xnew = np.asarray(xnew, dtype=np.float64).reshape(-1)
        # (i) predict_act followed by predict_com at the same inputs (pdgp.py:17-44 does exactly that per window)
        #     is one engine evaluation; (ii) while no Param changed, Kuu / its Cholesky factor / inverse of the
        #     previous prediction are reused instead of rebuilt for every call.  `.fixed` flags do not matter here.
        state = (param_version(), self._adam_t)
        memo = getattr(self, "_pred_memo", None)
        if memo is not None and memo[0] == state and memo[1].shape == xnew.shape and np.array_equal(memo[1], xnew):
            return memo[2]
        reuse = self._plan is not None and getattr(self, "_pred_state", None) == state
        if not reuse:
            self._pack()
            h = self._handle
        predict = h.lib.gp_pdgp_predict_reuse if reuse else h.lib.gp_pdgp_predict
        P, n = self.num_sources, xnew.size
        loc = self._local
        gp_mode = self._gp_shard is not None
        Pl, Gl = len(loc), (len(self._gp_shard) if gp_mode else 2 * len(loc))
        fmean = np.zeros((2 * P, n))
        fvar = np.zeros((2 * P, n))
        src = np.zeros((P, n))
        # engine row -> model row [g_0..g_{P-1}, f_0..f_{P-1}]
        rows = np.array(self._gp_shard if gp_mode else loc + [P + i for i in loc])
        step = self._max_batch
        for s in range(0, n, step):
            xs = h.to_device(xnew[s:s + step])
            c = xs.numel()
            fm, fv, ms = h.empty(Gl, c), h.empty(Gl, c), (None if gp_mode else h.empty(Pl, c))
            h.check(predict(self._plan, self._params.data_ptr(), xs.data_ptr(), c, fm.data_ptr(), fv.data_ptr(),
                            None if gp_mode else ms.data_ptr()))
            predict = h.lib.gp_pdgp_predict_reuse     # further chunks of the same call share the factorisation
            fmean[rows, s:s + c] = fm.cpu().numpy()
            fvar[rows, s:s + c] = fv.cpu().numpy()
            if not gp_mode:
                src[loc, s:s + c] = ms.cpu().numpy()
        self._pred_state = state
        if self._shard:
            # rows of other ranks are zero here: a sum over ranks assembles the full prediction (emulated ranks — a test
            # driving several shards in one process — pass allow_local=True through _pred_allow_local and get their rows only)
            from .dist import allreduce_sum_, require_group
            if not self.__dict__.get("_pred_allow_local", False):
                require_group(self._shard[1], "Pdgp.predict")
            t = h.torch
            for a in ((fmean, fvar) if gp_mode else (fmean, fvar, src)):
                a[...] = allreduce_sum_(t.as_tensor(a)).numpy()
        if gp_mode:
            # an activation GP and its component GP may live on different ranks: the source mean (pdgp.py:207) is formed
            # from the assembled rows — complete only once the sum over ranks has run (a rank on its own sees its rows)
            src = self.nlinfun(fmean[:P]) * fmean[P:]
        self._pred_memo = (state, xnew.copy(), (fmean, fvar, src))
        return fmean, fvar, src

    def sync_params(self):
        """pitch-sharded model: after training, give every rank the Param values of every pitch"""
        import torch.distributed as dist
        if not (self._shard and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
            return
        P = self.num_sources
        owned = self._gp_shard if self._gp_shard is not None else self._local + [P + i for i in self._local]
        mine = {}
        for g in owned:
            i, act = (g, True) if g < P else (g - P, False)
            kname = "kern_act" if act else "kern_com"
            mine[(kname, i)] = [q.value.copy() for q in getattr(self, kname)[i].theta_params()]
            for name in (("za", "q_mu_act", "q_sqrt_act") if act else ("zc", "q_mu_com", "q_sqrt_com")):
                mine[(name, i)] = getattr(self, name)[i].value.copy()
        everyone = [None] * dist.get_world_size()
        dist.all_gather_object(everyone, mine)
        for part in everyone:
            for (name, i), val in part.items():
                if (name, i) in mine:
                    continue
                if name.startswith("kern"):
                    for q, v in zip(getattr(self, name)[i].theta_params(), val):
                        q.value = v
                else:
                    getattr(self, name)[i].value = val

    def predict_act(self, xnew):
        """pdgp.py:172-179"""
        P = self.num_sources
        fm, fv, _ = self._predict(xnew, False)
        return [fm[i].reshape(-1, 1).copy() for i in range(P)], [fv[i].reshape(-1, 1).copy() for i in range(P)]

    def predict_com(self, xnew):
        """pdgp.py:181-188"""
        P = self.num_sources
        fm, fv, _ = self._predict(xnew, False)
        return [fm[P + i].reshape(-1, 1).copy() for i in range(P)], [fv[P + i].reshape(-1, 1).copy() for i in range(P)]

    def predict_act_n_com(self, xnew):
        """pdgp.py:190-208"""
        P = self.num_sources
        fm, fv, src = self._predict(xnew, True)
        col = lambda a, i: a[i].reshape(-1, 1).copy()
        return ([col(fm, i) for i in range(P)], [col(fv, i) for i in range(P)],
                [col(fm, P + i) for i in range(P)], [col(fv, P + i) for i in range(P)],
                [col(src, i) for i in range(P)])

    def __del__(self):
        try:
            if self._plan is not None and self._handle is not None and self._handle.h:
                self._handle.sync()
                self._handle.lib.gp_pdgp_destroy(self._plan)
        except Exception:
            pass
