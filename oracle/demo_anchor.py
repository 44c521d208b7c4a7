"""The reference's one printed, fully deterministic ELBO — restated through the oracle (TEST INFRASTRUCTURE ONLY).

/root/reference/demos/notebooks/demo_modgp-real-audio.ipynb, cells 4-9 (execution counts 5-10: one clean top-to-bottom
run under gpflow 0.5 / tensorflow 1.2.1, cell 2):

    x, y, fs = gpitch.readaudio("../data/011PFNOF_M60_train.wav")            cell 4   (methods.py:36-54)
    z, u = gpitch.init_liv(x=x, y=y, win_size=31, thres=0.033, dec=9)        cell 5   -> "109" inducing variables
    kact = gpflow.kernels.Matern32(input_dim=1, lengthscales=1.0, variance=1.0)
    kcom = MercerMatern12sm(input_dim=1, energy=ones(5), frequency=f0*[1..5])   cell 6 (variance = lengthscales = 1)
    m = gpitch.pdgp.Pdgp(x, y, z, kern=[[kact],[kcom]], minibatch_size=100); m.za.fixed = m.zc.fixed = True   cell 7
    logger: every 10th callback calls m._objective(x) (which draws a minibatch)                              cell 8
    m.optimize(method=tf.train.AdamOptimizer(learning_rate=0.0025), maxiter=10000, callback=logger)          cell 9

and the printed OptimizeResult:  fun = -69632.62624963776, the first and last three entries of `jac` and of `x`.

GPflow 0.5 semantics this run depends on beyond the ELBO itself (restated from that release):
  * Model._optimize_tf: per iteration ONE update_feed_dict (= one index draw per MinibatchData) + one session.run of
    the Adam step; then callback(free_state).  After the loop `fun, jac = self._objective(final_x)` on a FRESH draw.
  * Model._objective(x): update_feed_dict (one more draw) and returns (-f, -g) at x, without touching the Adam slots.
  * MinibatchData draws rng.randint(N, size=mb) when mb/N < 0.5; x and y own identically seeded RandomState(0)
    (pdgp.py:76-77), so one shared stream models both.
  * the free-state vector orders Params by attribute NAME (Parameterized.sorted_params), ParamLists in list order:
      kern_act[0].{lengthscales, variance}, kern_com[0].{energy[0..4], frequency[0..4], lengthscales, variance},
      likelihood.variance, q_mu_act[0], q_mu_com[0], q_sqrt_act[0], q_sqrt_com[0]     (za, zc fixed: absent)
    => printed x[0:3] = free(lengthscales_act), free(variance_act), free(energy_0); x[-3:] = q_sqrt_com[0][108,106:109,0].
  * tf.train.AdamOptimizer (TF 1.2): oracle.gpflow05.adam_step.
"""
import numpy as np

from . import gpflow05 as orc
from . import host

PRINTED = dict(
    fun=-69632.62624963776,
    jac_head=np.array([-5.90866639e+02, 4.25027187e+03, 6.57101668e+02]),
    jac_tail=np.array([4.10200613e-02, 2.65353723e-01, -7.03963363e-01]),
    x_head=np.array([0.93966396, 1.04036735, 3.99887154]),
    x_tail=np.array([0.07770003, 0.31399524, 0.6851173]),
    num_inducing=109, f0=261.6255653005986)

# free-state order of the run (see the module docstring); (name, transform) with "+" = transforms.positive
ORDER = ([("act.lengthscales", "+"), ("act.variance", "+")] +
         [("com.energy%d" % i, "+") for i in range(5)] + [("com.frequency%d" % i, "+") for i in range(5)] +
         [("com.lengthscales", "+"), ("com.variance", "+"), ("noise", "+"),
          ("q_mu_act", ""), ("q_mu_com", ""), ("q_sqrt_act", ""), ("q_sqrt_com", "")])


def demo_inputs(samples, fs):
    """cells 4-6: time axis as readaudio builds it (methods.py:53), inducing inputs from init_liv, kernel values."""
    y = np.asarray(samples, dtype=np.float64).reshape(-1, 1)          # soundfile.read -> float64
    n = y.size
    x = np.linspace(0., (n - 1.) / fs, n).reshape(-1, 1)
    z, u = host.init_liv(x=x, y=y, win_size=31, thres=0.033, dec=9)
    f0 = 2. ** ((60 - 69.) / 12.) * 440.                               # methods.py:266-267 midi2freq(60)
    return x, y, z, f0


def initial_state(z, f0):
    """constrained initial values by name (cells 6-7; pdgp.py:92-103; likelihoods.py:283)"""
    M = z[0][0].shape[0]
    st = {"act.lengthscales": np.array(1.0), "act.variance": np.array(1.0),
          "com.lengthscales": np.array(1.0), "com.variance": np.array(1.0), "noise": np.array(1.0),
          "q_mu_act": np.zeros((M, 1)), "q_mu_com": np.zeros((M, 1)),
          "q_sqrt_act": np.eye(M)[:, :, None].copy(), "q_sqrt_com": np.eye(M)[:, :, None].copy()}
    for i in range(5):
        st["com.energy%d" % i] = np.array(1.0)
        st["com.frequency%d" % i] = np.array(f0 * (i + 1))
    return st


def to_free(st):
    return {k: (orc.positive_backward(st[k]) if t == "+" else np.array(st[k], dtype=np.float64)) for k, t in ORDER}


def flatten(d):
    """name -> array dict to the flat vector in the run's free-state order"""
    return np.concatenate([np.asarray(d[k], dtype=np.float64).reshape(-1) for k, _ in ORDER])


class DemoObjective(object):
    """-(ELBO) and its free-state gradient on a given index set, by torch-CPU autograd through oracle.gpflow05
    (mirrors tf.gradients of  build_likelihood + build_prior  w.r.t. the free-state Variable)."""

    def __init__(self, x, y, z):
        import torch
        from .backend import TorchBackend
        self.torch, self.tb = torch, TorchBackend()
        self.x, self.y = torch.as_tensor(x), torch.as_tensor(y)
        self.za, self.zc = torch.as_tensor(np.asarray(z[0][0])), torch.as_tensor(np.asarray(z[1][0]))
        self.N = x.shape[0]

    def __call__(self, free, idx):
        torch = self.torch
        leaves = {k: torch.tensor(np.asarray(free[k], dtype=np.float64), requires_grad=True) for k, _ in ORDER}
        # Log1pe.tf_forward: softplus(x) + 1e-6
        c = {k: (torch.nn.functional.softplus(leaves[k]) + 1e-6 if t == "+" else leaves[k]) for k, t in ORDER}
        kact = {"type": "matern32", "variance": c["act.variance"], "lengthscales": c["act.lengthscales"],
                "energy": [], "frequency": []}
        kcom = {"type": "mercer_matern12sm", "variance": c["com.variance"], "lengthscales": c["com.lengthscales"],
                "energy": [c["com.energy%d" % i] for i in range(5)],
                "frequency": [c["com.frequency%d" % i] for i in range(5)]}
        ti = torch.as_tensor(np.asarray(idx))
        elbo = orc.pdgp_elbo(self.x[ti], self.y[ti], [self.za], [self.zc], [kact], [kcom],
                             [c["q_mu_act"]], [c["q_sqrt_act"]], [c["q_mu_com"]], [c["q_sqrt_com"]], c["noise"],
                             num_data=self.N, whiten=True, nlin_code=orc.NLIN_LOGISTIC, xp=self.tb)
        elbo.backward()
        g = {k: -(leaves[k].grad.numpy().copy() if leaves[k].grad is not None else np.zeros(leaves[k].shape))
             for k, _ in ORDER}
        return -float(elbo.detach()), g


def run_demo(samples, fs, maxiter=10000, lr=0.0025, mb=100, log_every=10, progress=None, record_at=()):
    """cells 4-9 through the oracle.  Returns a dict with the final `fun`, flat `jac` and `x` (free state, the run's
    order), the logger's trace, the constrained final state by name and the index stream's draw count."""
    x, y, z, f0 = demo_inputs(samples, fs)
    obj = DemoObjective(x, y, z)
    rng = np.random.RandomState(0)                     # pdgp.py:76-77 (x and y streams are identical)
    free = to_free(initial_state(z, f0))
    mom = {k: (np.zeros_like(free[k]), np.zeros_like(free[k])) for k, _ in ORDER}
    logf, draws, snaps = [], 0, {}
    logger_i = 1                                        # cell 8: logger.i = 1
    for it in range(1, maxiter + 1):
        idx = orc.minibatch_indices(rng, obj.N, mb); draws += 1
        _, g = obj(free, idx)
        for k, _ in ORDER:
            free[k], m_, v_ = orc.adam_step(free[k], g[k], mom[k][0], mom[k][1], it, lr)
            mom[k] = (m_, v_)
        if (logger_i % log_every) == 0:                 # the callback's own _objective call draws a minibatch
            idx = orc.minibatch_indices(rng, obj.N, mb); draws += 1
            logf.append(obj(free, idx)[0])
        logger_i += 1
        if it in record_at:
            snaps[it] = flatten(free).copy()
        if progress and it % 1000 == 0:
            progress(it, logf[-1] if logf else float("nan"))
    idx_final = orc.minibatch_indices(rng, obj.N, mb); draws += 1
    fun, jac = obj(free, idx_final)
    return dict(fun=fun, jac=flatten(jac), x=flatten(free), logf=np.array(logf), draws=draws, free=free,
                idx_final=idx_final, num_inducing=z[0][0].shape[0], f0=f0, z=z[0][0], snaps=snaps,
                rng_state=rng.get_state())
