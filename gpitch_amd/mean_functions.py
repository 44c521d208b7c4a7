"""GPflow-0.5 mean functions as they can be handed to SGPRSS(mean_function=...) (gpitch/sgpr_ss.py:14,25,40,90-99).

The reference passes `mean_function` through to gpflow.sgpr.SGPR and subtracts it from the data in the bound
(`err = Y - mean_function(X)`, :40) and in the exact per-source posterior (:90), adding it back to the predicted
means (:95).  None of its callers sets one (transcription.py:245, separation.py:257 use the default Zero), so what is
mirrored here is the part that changes results: a FIXED function of the inputs.  GPflow's Constant / Linear carry
trainable Params; here their values are held fixed during optimize() (documented deviation: no gradient flows to
them).  Anything callable on an (n, 1) array that returns (n, 1) works."""
import numpy as np


class Zero(object):
    def __call__(self, X):
        return np.zeros((np.asarray(X).reshape(-1, 1).shape[0], 1))


class Constant(object):
    def __init__(self, c=0.0):
        self.c = float(np.asarray(c).reshape(-1)[0])

    def __call__(self, X):
        return np.full((np.asarray(X).reshape(-1, 1).shape[0], 1), self.c)


class Linear(object):
    def __init__(self, A=1.0, b=0.0):
        self.A = float(np.asarray(A).reshape(-1)[0])
        self.b = float(np.asarray(b).reshape(-1)[0])

    def __call__(self, X):
        return np.asarray(X, dtype=np.float64).reshape(-1, 1) * self.A + self.b
