"""CPU oracle for the gpitch pdgp / sgpr_ss ELBO path.

TEST INFRASTRUCTURE ONLY.  This package is a CPU restatement of the reference's
algorithm (gpitch + the GPflow-0.5 functions it calls).  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it,
and only as the checker / the reported CPU baseline.  The product package
``gpitch_amd`` never imports it and has no CPU fallback.

PARITY PIN.  The reference ships no tests, fixtures or golden vectors and cannot be
imported in this container (Python 2 + GPflow 0.5 + TensorFlow 1.2.1: ordinary
ModuleNotFoundError).  What it does hold is ONE printed result of a deterministic
ELBO job — demos/notebooks/demo_modgp-real-audio.ipynb cells 4-9: the shipped
recording, init_liv -> 109 inducing points, Matern32 + MercerMatern12sm(5 partials),
minibatches of 100 from RandomState(0), tf.train.AdamOptimizer(0.0025) x 10000 with a
logging callback — printing ``fun: -69632.62624963776`` and the first / last three
entries of ``jac`` and ``x``.  oracle/demo_anchor.py runs that job through this
restatement (tests/test_demo_anchor.py, fixture tests/golden/demo_real_audio_anchor.npz):
    fun   -69632.58690621857   (5.7e-7 relative to the printed value)
    x     [0.93924 1.04038 3.99886 ... 0.07771 0.31397 0.68514]   (printed 0.93966 1.04037 3.99887 ... 0.07770 0.31400 0.68512)
    jac   within 2.5e-4 relative of the six printed entries
after 10000 float64 Adam steps on another BLAS.  This pins, end to end, the Pdgp ELBO
and its gradient (MercerMatern12sm + Matern32 kernels, whitened conditional, gauss_kl,
MpdLik + Gauss-Hermite, the shifted logistic), the Log1pe transform, MinibatchData's
index stream, TF-1.2 Adam, GPflow's free-state order and init_liv.  NOT covered by
that run, and therefore pinned only by (a) analytic known-answer identities
(tests/test_oracle_kat.py) and (b) independent 50-digit mpmath evaluations
(oracle/mp_elbo.py -> tests/golden/*.npz): the SGPRSS bound and its predictions,
whiten=False, P > 1 (the likelihood's cross terms), softplus / gaussfun, Matern12sm,
Matern32sm, Matern52 x MercerCosMix, the Logistic transform.
"""
