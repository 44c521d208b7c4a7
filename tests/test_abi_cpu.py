"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol that
include/gpitch_abi.h declares; host logic of the Python mirror.  No compute calls (no GPU here)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
    src = open(os.path.join(ROOT, "include", "gpitch_abi.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(gp_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from gpitch_amd import _lib
    lib = _lib.load_library()
    declared = _header_functions()
    assert len(declared) >= 30
    assert sorted(_lib.ABI_SYMBOLS) == declared
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.gp_abi_version() == 1


def test_no_cpu_fallback_without_device():
    import torch
    from gpitch_amd import _lib
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    lib = _lib.load_library()
    h = C.c_void_p()
    assert lib.gp_create(0, None, C.byref(h)) == _lib.GP_ERR_NO_DEVICE
    with pytest.raises(_lib.GpitchError):
        _lib.Handle(0)
    import gpitch_amd
    k = gpitch_amd.kernels.Matern32(1)
    with pytest.raises(_lib.GpitchError):
        k.K(np.zeros((3, 1)))          # product path fails loudly; it never routes to the oracle


def test_product_never_imports_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "gpitch_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), os.path.join(dirpath, f)


def test_param_transforms_and_containers():
    import gpitch_amd
    from gpitch_amd.param import Param, ParamList, transforms, MinibatchData
    p = Param(2.0, transforms.positive)
    x = p.transform.backward(p.value)
    np.testing.assert_allclose(p.transform.forward(x), [2.0], rtol=1e-12)
    assert transforms.positive.code == 1
    k = gpitch_amd.matern12_spectral_mixture.MercerMatern12sm(1, energy=np.array([0.5, 0.5]), frequency=np.array([100., 200.]),
                                                              lengthscales=0.1, len_fixed=True)
    assert k.lengthscales.fixed and not k.variance.fixed and k.num_partials == 2
    np.testing.assert_allclose(k.theta(), [1.0, 0.1, 0.5, 0.5, 100., 200.])
    k.lengthscales = 0.25            # GPflow-style assignment into the Param (init_models.py:101-105)
    assert isinstance(k.lengthscales, Param) and k.lengthscales.value[0] == 0.25
    k2 = gpitch_amd.matern12_spectral_mixture.Matern12sm(1, variance=1., lengthscales=0.2, energy=[1., 2.], frequency=[3., 4.])
    assert all(e.fixed for e in k2.energy) and all(f.fixed for f in k2.frequency)   # m12sm.py:34
    add = np.sum([gpitch_amd.kernels.Matern32(1), k])
    assert len(add.kern_list) == 2
    pl = ParamList([Param(1.0), Param(2.0)])
    pl.fixed = True
    assert pl.fixed
    a = np.arange(20.).reshape(-1, 1)
    mx, my = MinibatchData(a, 5, np.random.RandomState(0)), MinibatchData(2 * a, 5, np.random.RandomState(0))
    assert np.array_equal(mx.next_indices(), my.next_indices())     # rows stay paired (pdgp.py:76-77)
    assert sorted(MinibatchData(a, 20).next_indices()) == list(range(20))


def test_init_liv_reproduces_reference_quirk():
    import gpitch_amd
    fs, n = 16000, 4000
    x = np.linspace(0, (n - 1.) / fs, n).reshape(-1, 1)
    y = (np.sin(2 * np.pi * 200 * x) * np.exp(-((x - 0.12) / 0.04) ** 2))
    z, u = gpitch_amd.init_liv(x, y, win_size=31, thres=0.05, dec=1)
    za = z[0][0]
    # extrema above threshold = k; the tuple-argsort bug returns the FIRST k extrema (init_models.py:38-43)
    assert za.ndim == 2 and za.shape[1] == 1 and za.shape[0] == u.shape[0] > 5
    assert np.all(np.diff(za[:, 0]) > 0)
    ys = y.reshape(-1)
    z2 = gpitch_amd.init_iv(x, 2, 100, 200, fs)
    assert z2[0][0].shape[0] == len(x[::160]) + 1 and z2[1][1].shape[0] == len(x[::80]) + 1
    ka, kc = gpitch_amd.init_kern(2, [np.array(0.1)] * 2, [np.array([1., 1.])] * 2, [np.array([100., 200.])] * 2)
    assert ka[0].variance.value[0] == 3.5 and kc[1].lengthscales.fixed


def test_nonlinearity_tokens():
    import gpitch_amd
    assert gpitch_amd.logistic_tf.code == 0 and gpitch_amd.softplus_tf.code == 1 and gpitch_amd.gaussfun_tf.code == 2
    np.testing.assert_allclose(gpitch_amd.logistic(np.pi), 0.5)
    np.testing.assert_allclose(gpitch_amd.logistic_tf(np.array([np.pi])), [0.5])
    with pytest.raises(TypeError):
        gpitch_amd.methods.nlin_code(np.tanh)


def test_windowing_and_overlap_add():
    import gpitch_amd
    from gpitch_amd import window_overlap as wo
    n, ws = 10001, 2001
    x = np.linspace(0, 1, n).reshape(-1, 1)
    y = np.sin(40 * x)
    xw, yw = wo.windowed(x, y, ws)
    l = (ws - 1) // 2
    assert len(xw) == (n - ws) // l + 1 and all(w.shape == (ws, 1) for w in xw)
    assert np.array_equal(xw[1][:l + 1], xw[0][l:])                       # 50 % overlap
    # Hann windows at 50 % overlap sum to one: overlap-add of the windows themselves returns the signal
    merged = wo.merged_mean(yw, ws, n)
    np.testing.assert_allclose(merged, y, atol=1e-12)
    np.testing.assert_allclose(wo.merged_x(xw, ws), x, atol=1e-12)
    v = wo.merged_variance([np.ones((ws, 1)) for _ in yw], ws, n)
    assert v.min() >= 0.5 - 1e-12 and v.max() <= 1 + 1e-12
    xs, ys = gpitch_amd.segmented(x, y, window_size=3000)
    assert len(xs) == 3 and ys[2].shape == (3000, 1)
    xa, ya = wo.segmented(x, y, window_size=3000, aug=True)
    assert ya[0].shape == (3000 + 3200, 1) and ya[0][:1600].max() == 0


def test_init_cparam_and_kernel_initialisers():
    """host-side initialisers of SURVEY 8f rank 1 / 4 (methods.py:91-153, init_models.py:74-121,154-198): spectral
    peaks of a synthetic note become the Matern32sm frequencies / variances; the reference's quirks are kept."""
    import gpitch_amd
    from gpitch_amd import init_models
    from gpitch_amd.kernels import Matern32sm, Prod
    from gpitch_amd.methods import find_ideal_f0, init_cparam, midi2freq, peak_indexes
    from gpitch_amd.param import Logistic
    assert find_ideal_f0(['011PFNOF_M60_train.wav']) == [261.6255653005986]      # demo_modgp-real-audio.ipynb:66
    # peak picker: plateau -> its middle; weaker neighbours within min_dist are suppressed
    y = np.array([0., 1., 0., 2., 2., 2., 0., .5, 3., .5, 0.])
    assert list(peak_indexes(y, thres=0.1, min_dist=1)) == [1, 4, 8]
    assert list(peak_indexes(y, thres=0.1, min_dist=3)) == [4, 8]      # 4 (height 2) shadows 1 (height 1)
    assert list(peak_indexes(y, thres=0.1, min_dist=4)) == [1, 8]      # 8 (height 3) shadows 4, which then spares 1
    assert list(peak_indexes(y, thres=0.8, min_dist=1)) == [8]
    fs, N = 16000, 16000
    x = np.arange(N) / float(fs)
    f0 = midi2freq(60)
    amps = {1: 1.0, 2: 0.5, 3: 0.25}
    y = sum(a * np.sin(2 * np.pi * k * f0 * x) for k, a in amps.items()) + 1e-3 * np.random.RandomState(0).randn(N)
    freq, var, F, S, thres = init_cparam(y, fs, maxh=3, ideal_f0=f0)
    assert freq.size == 3 and np.all(np.abs(freq - f0 * np.array([1, 2, 3])) < 1.5)
    assert abs(var.sum() - 1.) < 1e-12 and var[0] > var[1] > var[2]
    kern, iparam = init_models.init_kernel_training([y], ['note_M60_x.wav'], fs, maxh=3)
    kc = kern[1][0]
    assert isinstance(kc, Matern32sm) and kc.num_partials == 3
    assert all(v.fixed for v in kc.variance) and not any(f.fixed for f in kc.frequency)     # vars_n_freqs_fixed()
    assert isinstance(kc.lengthscales.transform, Logistic) and kc.lengthscales.transform.b == 2.
    np.testing.assert_allclose([f.value[0] for f in kc.frequency], iparam[0][0])

    class _M(object):           # a "trained per-pitch model" as init_kernel_with_trained_models reads it
        kern_act, kern_com = [kern[0][0]], [kc]
    kern[0][0].lengthscales = 0.3
    k2 = init_models.init_kernel_with_trained_models([_M()])
    assert k2[0][0].lengthscales.value[0] == 0.3 and not k2[0][0].lengthscales.fixed
    assert not k2[1][0].lengthscales.fixed and all(v.fixed for v in k2[1][0].variance)
    np.testing.assert_allclose(k2[1][0].theta(), kc.theta())
    k3 = init_models.init_kern(2, [np.array([.6, .4])] * 2, [np.array([100., 200.])] * 2)
    p = k3[1][0]
    assert isinstance(p, Prod) and p.type_code == 7
    np.testing.assert_allclose(p.theta(), [0.25, 0.25, .6, .4, 100., 200.])
    assert [q.fixed for q in p.theta_params()] == [True, False, True, True, True, True]
    f, e = init_models.get_features(np.arange(10.), np.arange(10.) + 1., None, 1, False, 3)
    np.testing.assert_allclose(f, [9., 8., 7.])
    np.testing.assert_allclose(e, np.array([10., 9., 8.]) / 27.)
    # Logistic transform round trip and derivative (host mirror of the device kernels)
    t = Logistic(0., 0.5)
    xs = np.linspace(-4, 4, 9)
    np.testing.assert_allclose(t.backward(t.forward(xs)), xs, atol=1e-12)
    np.testing.assert_allclose(t.dforward(xs), (t.forward(xs + 1e-6) - t.forward(xs - 1e-6)) / 2e-6, rtol=1e-6)
