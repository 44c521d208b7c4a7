"""Optimiser tokens.  `AdamOptimizer` plays the role of tf.train.AdamOptimizer as passed to
Model.optimize(method=...) in demos/scripts/demo-modgp.py:44-45; the update itself runs on the GPU
(gp_adam_step, csrc/opt.hip)."""


class AdamOptimizer(object):
    def __init__(self, learning_rate=0.001, beta1=0.9, beta2=0.999, epsilon=1e-8):
        self.learning_rate = float(learning_rate)
        self.beta1 = float(beta1)
        self.beta2 = float(beta2)
        self.epsilon = float(epsilon)


class OptimizeResult(dict):
    """scipy-style result as GPflow returns it (demo_modgp.ipynb:140-146)."""
    __getattr__ = dict.get
