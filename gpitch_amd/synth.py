"""Synthetic 16 kHz audio problems of the (N frames x M inducing x P pitches) shape.

Generalises the reference's demo generator (demos/scripts/demo-modgp.py:10-26: harmonic component
times a two-bump Gaussian envelope plus 1e-6-variance noise) to P pitches, with the hyper-parameter
initialisation the reference uses for transcription models: activation kernel Matern32(l=1, v=3.5)
(gpitch/init_kernels.py:12), component kernel MercerMatern12sm(v=1, l=0.1, e_k = 1/m, f_k = k f0)
(gpitch/init_kernels.py:29-37), uniform inducing grid (gpitch/init_models.py:54-71), sigma^2 = 1
(gpitch/likelihoods.py:283).  Host-side numpy only.
"""
import numpy as np


def midi2freq(midi):
    """gpitch/methods.py:266-267"""
    return 2. ** ((midi - 69.) / 12.) * 440.


def per_fun(xin, npartials, freq):
    """demos/scripts/demo-modgp.py:10-15"""
    f = np.zeros(xin.shape)
    for i in range(npartials):
        f += np.sin(2 * np.pi * xin * (i + 1) * freq)
    return f / np.max(np.abs(f))


def uniform_inducing(x, M):
    """init_iv-style uniform decimation with the last sample appended (init_models.py:63-69),
    truncated to exactly M points."""
    N = x.shape[0]
    dec = max(N // M, 1)
    z = np.vstack([x[::dec].copy(), x[-1:].copy()])
    if z.shape[0] < M:
        raise ValueError("cannot place %d inducing points on %d frames" % (M, N))
    return z[:M].copy()


def make_problem(N, M, P, num_partials=5, seed=0, fs=16000, noise_var=1.0, base_midi=60, trivial_q=False):
    """Returns a dict with x, y (N,1); z lists; kernel dicts (oracle format); q_mu / q_sqrt lists."""
    rng = np.random.RandomState(seed)
    x = np.linspace(0., (N - 1.) / fs, N).reshape(-1, 1)
    T = N / float(fs)
    y = np.zeros_like(x)
    kern_act, kern_com = [], []
    for p in range(P):
        f0 = midi2freq(base_midi + p)
        comp = per_fun(x, num_partials, f0)
        c1 = T * (0.2 + 0.5 * p / max(P, 1))
        c2 = T * (0.55 + 0.4 * p / max(P, 1))
        env = np.exp(-25. / T ** 2 * (x - c1) ** 2) + np.exp(-75. / T ** 2 * (x - c2) ** 2)
        env /= np.max(np.abs(env))
        y += env * comp
        kern_act.append({"type": "matern32", "variance": 3.5, "lengthscales": 1.0, "energy": [], "frequency": []})
        kern_com.append({"type": "mercer_matern12sm", "variance": 1.0, "lengthscales": 0.1,
                         "energy": [1. / num_partials] * num_partials,
                         "frequency": [(k + 1) * f0 for k in range(num_partials)]})
    y += np.sqrt(1e-6) * rng.randn(N, 1)
    y /= np.max(np.abs(y))
    z = uniform_inducing(x, M)
    za = [z.copy() for _ in range(P)]
    zc = [z.copy() for _ in range(P)]
    rq = np.random.RandomState(seed + 1)
    q_mu_act, q_mu_com, q_sqrt_act, q_sqrt_com = [], [], [], []
    for p in range(P):
        for mus, sqs in ((q_mu_act, q_sqrt_act), (q_mu_com, q_sqrt_com)):
            if trivial_q:
                mus.append(np.zeros((M, 1)))
                sqs.append(np.eye(M)[:, :, None].copy())
            else:
                mus.append(0.3 * rq.randn(M, 1))
                sqs.append(np.tril(np.eye(M) + 0.05 * rq.randn(M, M))[:, :, None].copy())
    return dict(x=x, y=y, za=za, zc=zc, kern_act=kern_act, kern_com=kern_com,
                q_mu_act=q_mu_act, q_sqrt_act=q_sqrt_act, q_mu_com=q_mu_com, q_sqrt_com=q_sqrt_com,
                noise_var=float(noise_var), N=N, M=M, P=P, fs=fs, num_partials=num_partials)


def kernels_from_problem(prob):
    """product kernel objects for the oracle-format kernel dicts of gpitch_amd.synth.make_problem"""
    import gpitch_amd
    from gpitch_amd.kernels import Matern12, Matern32, Matern52, RBF
    from gpitch_amd.matern12_spectral_mixture import MercerMatern12sm, Matern12sm
    cls = {"matern12": Matern12, "matern32": Matern32, "matern52": Matern52, "rbf": RBF}

    def mk(d):
        if d["type"] == "mercer_matern12sm":
            return MercerMatern12sm(1, energy=np.array(d["energy"]), frequency=np.array(d["frequency"]),
                                    variance=d["variance"], lengthscales=d["lengthscales"])
        if d["type"] == "matern32sm":
            from gpitch_amd.kernels import Matern32sm
            return Matern32sm(1, len(d["frequency"]), lengthscales=d["lengthscales"], variances=np.array(d["energy"]),
                              frequencies=np.array(d["frequency"]))
        if d["type"] == "mercer_matern52sm":
            from gpitch_amd.kernels import Matern52, MercerCosMix
            a = Matern52(1, lengthscales=d["lengthscales"], variance=1.0)
            a.variance.fixed = True
            b = MercerCosMix(1, energy=np.array(d["energy"]), frequency=np.array(d["frequency"]), variance=d["variance"],
                             features_as_params=True)
            b.variance.fixed = True
            return a * b
        if d["type"] == "matern12sm":
            return Matern12sm(1, variance=d["variance"], lengthscales=d["lengthscales"],
                              energy=np.array(d["energy"]), frequency=np.array(d["frequency"]))
        return cls[d["type"]](1, variance=d["variance"], lengthscales=d["lengthscales"])
    return [[mk(d) for d in prob["kern_act"]], [mk(d) for d in prob["kern_com"]]]


def pdgp_from_problem(prob, whiten=True, minibatch_size=None, nlinfun=None, handle=None, shard=None, float_type=None):
    import gpitch_amd
    from gpitch_amd.pdgp import Pdgp
    kern = kernels_from_problem(prob)
    m = Pdgp(prob["x"], prob["y"], [prob["za"], prob["zc"]], kern, whiten=whiten, minibatch_size=minibatch_size,
             nlinfun=nlinfun or gpitch_amd.logistic_tf, handle=handle, shard=shard, float_type=float_type)
    for i in range(prob["P"]):
        m.q_mu_act[i].value = prob["q_mu_act"][i]
        m.q_mu_com[i].value = prob["q_mu_com"][i]
        m.q_sqrt_act[i].value = prob["q_sqrt_act"][i]
        m.q_sqrt_com[i].value = prob["q_sqrt_com"][i]
    m.likelihood.variance = prob["noise_var"]
    return m
