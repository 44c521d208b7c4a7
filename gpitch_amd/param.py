"""Thin re-creation of the GPflow-0.5 parameter containers the reference uses
(gpflow.param.Param / ParamList / DataHolder, gpflow.minibatch.MinibatchData, transforms.positive;
call sites gpitch/pdgp.py:76-111, gpitch/matern12_spectral_mixture.py:26-32,86-94,
gpitch/sgpr_ss.py:23-26).  Host-side bookkeeping only; values are packed into one flat device
vector by the models."""
import numpy as np


class Identity(object):
    code = 0

    def forward(self, x):
        return np.asarray(x, dtype=np.float64)

    def backward(self, y):
        return np.asarray(y, dtype=np.float64)

    def dforward(self, x):
        """d forward(x) / dx (the factor GPflow's free-state gradient carries)"""
        return np.ones_like(np.asarray(x, dtype=np.float64))

    def device_code(self, handle):
        """transform code of the C-ABI (include/gpitch_abi.h gp_transform_forward)"""
        return self.code


class Log1pe(object):
    """GPflow transforms.positive: y = log(1 + exp(x)) + 1e-6."""
    code = 1

    def __init__(self, lower=1e-6):
        self._lower = lower

    def forward(self, x):
        return np.logaddexp(0., np.asarray(x, dtype=np.float64)) + self._lower

    def backward(self, y):
        y = np.asarray(y, dtype=np.float64) - self._lower
        return y + np.log(-np.expm1(-y))

    def dforward(self, x):
        return 1. / (1. + np.exp(-np.asarray(x, dtype=np.float64)))

    def device_code(self, handle):
        return self.code


class Logistic(object):
    """gpflow.transforms.Logistic(a, b): y = a + (b - a) / (1 + exp(-x)) (kernels.py:219-223,333;
    init_models.py:189).  On the device the (a, b) pair is registered per handle."""

    def __init__(self, a=0., b=1.):
        if not b > a:
            raise ValueError("Logistic(a, b) needs b > a")
        self.a, self.b = float(a), float(b)

    def forward(self, x):
        ex = np.exp(-np.asarray(x, dtype=np.float64))
        return self.a + (self.b - self.a) / (1. + ex)

    def backward(self, y):
        return -np.log((self.b - self.a) / (np.asarray(y, dtype=np.float64) - self.a) - 1.)

    def dforward(self, x):
        s = 1. / (1. + np.exp(-np.asarray(x, dtype=np.float64)))
        return (self.b - self.a) * s * (1. - s)

    def device_code(self, handle):
        import ctypes as C
        code = C.c_uint8()
        handle.check(handle.lib.gp_transform_register_logistic(handle.h, self.a, self.b, C.byref(code)))
        return int(code.value)


class transforms(object):
    Identity = Identity
    Log1pe = Log1pe
    Logistic = Logistic
    positive = Log1pe()


_version = [0]


def param_version():
    """bumped by every Param assignment anywhere: lets models tell whether their packed device copy is stale"""
    return _version[0]


class Param(object):
    def __init__(self, array, transform=None):
        self._array = np.atleast_1d(np.array(array, dtype=np.float64))
        self.transform = transform if transform is not None else Identity()
        self.fixed = False

    @property
    def value(self):
        return self._array.copy()

    @value.setter
    def value(self, v):
        self._array = np.asarray(v, dtype=np.float64).reshape(self._array.shape).copy()
        _version[0] += 1

    def assign(self, v):
        self.value = np.broadcast_to(np.asarray(v, dtype=np.float64), self._array.shape)

    @property
    def shape(self):
        return self._array.shape

    @property
    def size(self):
        return self._array.size

    def __repr__(self):
        return "Param(%s, fixed=%s)" % (np.array2string(self._array, threshold=6), self.fixed)


class ParamList(object):
    """gpflow.param.ParamList: a list of Params (or Parameterized objects such as kernels)."""

    def __init__(self, items):
        self._list = list(items)

    def __getitem__(self, i):
        return self._list[i]

    def __len__(self):
        return len(self._list)

    def __iter__(self):
        return iter(self._list)

    @property
    def fixed(self):
        return all(getattr(p, "fixed", False) for p in self._list)

    @fixed.setter
    def fixed(self, v):
        for p in self._list:
            p.fixed = v


class DataHolder(object):
    def __init__(self, array, on_shape_change="raise"):
        self._array = np.asarray(array, dtype=np.float64).copy()

    @property
    def value(self):
        return self._array.copy()

    @property
    def shape(self):
        return self._array.shape


class MinibatchData(DataHolder):
    """gpflow.minibatch.MinibatchData(array, minibatch_size, rng): a fresh index set on every draw;
    with replacement when mb/N < 0.5, otherwise a permutation prefix (GPflow 0.5)."""

    def __init__(self, array, minibatch_size, rng=None):
        DataHolder.__init__(self, array)
        self.minibatch_size = int(minibatch_size)
        self.rng = rng if rng is not None else np.random.RandomState(0)

    def next_indices(self):
        N = self._array.shape[0]
        if float(self.minibatch_size) / float(N) < 0.5:
            return self.rng.randint(N, size=self.minibatch_size)
        return self.rng.permutation(N)[:self.minibatch_size]


class Parameterized(object):
    """Anything with Params.  `.fixed = True` fixes every Param underneath (GPflow semantics used at
    init_models.py:97-98)."""

    def _params(self):
        out = []
        for v in self.__dict__.values():
            if isinstance(v, Param):
                out.append(v)
            elif isinstance(v, (ParamList, Parameterized)):
                out.extend(_collect(v))
        return out

    @property
    def fixed(self):
        ps = self._params()
        return bool(ps) and all(p.fixed for p in ps)

    @fixed.setter
    def fixed(self, v):
        for p in self._params():
            p.fixed = v

    def __setattr__(self, name, value):
        # GPflow lets `kern.lengthscales = 0.5` assign into the existing Param (init_models.py:101-105)
        cur = self.__dict__.get(name)
        if isinstance(cur, Param) and not isinstance(value, Param):
            cur.assign(value)
        else:
            object.__setattr__(self, name, value)


def sorted_params(obj):
    """Every Param underneath `obj` in the order of GPflow 0.5's free-state vector (Parameterized.sorted_params /
    get_free_state): children by attribute NAME, the items of a ParamList in list order.  This is the order of the `x`
    and `jac` that Model.optimize returns and hands to callbacks (demos/notebooks/demo_modgp-real-audio.ipynb cell 9
    prints them)."""
    if isinstance(obj, Param):
        return [obj]
    out = []
    if isinstance(obj, ParamList):
        for it in obj:
            out.extend(sorted_params(it))
    elif isinstance(obj, Parameterized):
        for name in sorted(k for k, v in obj.__dict__.items() if isinstance(v, (Param, ParamList, Parameterized))):
            out.extend(sorted_params(obj.__dict__[name]))
    return out


def _collect(obj):
    if isinstance(obj, Param):
        return [obj]
    if isinstance(obj, ParamList):
        out = []
        for it in obj:
            out.extend(_collect(it))
        return out
    if isinstance(obj, Parameterized):
        return obj._params()
    return []
