"""CPU oracle for the gpitch pdgp / sgpr_ss ELBO path.

TEST INFRASTRUCTURE ONLY.  This package is a CPU restatement of the reference's
algorithm (gpitch + the GPflow-0.5 functions it calls).  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it,
and only as the checker / the reported CPU baseline.  The product package
``gpitch_amd`` never imports it and has no CPU fallback.

PARITY UNPINNED: the reference ships no tests, fixtures or golden vectors, and it
cannot be imported in this container (Python 2 + GPflow 0.5 + TensorFlow 1.2.1,
ordinary ModuleNotFoundError).  The oracle is therefore pinned only by
 (a) analytic known-answer identities (tests/test_oracle_kat.py), and
 (b) an independent 50-digit mpmath evaluation of the same formulas
     (oracle/mp_elbo.py -> tests/golden/*.npz).
"""
