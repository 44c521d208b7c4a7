"""Host-side initialisers and windowing: the package's own vectorised implementations
against (a) the one anchor the reference prints — 109 inducing points on its demo recording — and (b) the oracle's
literal restatement (oracle/host.py) on seeded random signals, quirks included.  CPU only."""
import os

import numpy as np
import pytest

from oracle import host as ref

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _noisy_notes(seed, n=8000, fs=16000):
    rng = np.random.RandomState(seed)
    x = np.linspace(0., (n - 1.) / fs, n).reshape(-1, 1)
    env = np.exp(-40. * (x - 0.12) ** 2 / 0.02) + 0.6 * np.exp(-40. * (x - 0.36) ** 2 / 0.01)
    y = env * (np.sin(2 * np.pi * 220. * x) + 0.5 * np.sin(2 * np.pi * 440. * x + 0.3)) + 1e-3 * rng.randn(n, 1)
    return x, y, fs


def test_init_liv_real_audio_gives_the_109_points_the_reference_prints():
    """demos/notebooks/demo_modgp-real-audio.ipynb:88,116 (fixture: tests/golden/make_init_fixtures.py)"""
    import gpitch_amd
    g = np.load(os.path.join(GOLD, "init_liv_real_audio.npz"))
    y = g["y"].astype(np.float64).reshape(-1, 1)
    fs = int(g["fs"])
    x = np.linspace(0., (y.size - 1.) / fs, y.size).reshape(-1, 1)          # gpitch/methods.py:53
    kw = dict(win_size=int(g["win_size"]), thres=float(g["thres"]), dec=int(g["dec"]))
    z, u = gpitch_amd.init_liv(x=x, y=y, **kw)
    assert u.size == int(g["expected_num_inducing"]) == 109
    assert z[0][0].shape == (109, 1) and z[1][0].shape == (109, 1)
    # the oracle's statement-by-statement restatement agrees point for point
    z_ref, u_ref = ref.init_liv(x, y, **kw)
    assert u_ref.size == 109
    np.testing.assert_array_equal(z[0][0], z_ref[0][0])
    np.testing.assert_array_equal(u, u_ref)
    from gpitch_amd.methods import find_ideal_f0
    assert find_ideal_f0([str(g["fname"])]) == [float(g["expected_ideal_f0"])]  # notebook :66


@pytest.mark.parametrize("seed,win,thres,dec,ns", [(0, 9, 0.0025, 1, 1), (1, 31, 0.05, 1, 2), (2, 15, 0.2, 3, 3),
                                                   (3, 5, 0.6, 2, 1)])
def test_init_liv_matches_restatement(seed, win, thres, dec, ns):
    import gpitch_amd
    x, y, _ = _noisy_notes(seed)
    z, u = gpitch_amd.init_liv(x, y, num_sources=ns, win_size=win, thres=thres, dec=dec)
    zr, ur = ref.init_liv(x, y, num_sources=ns, win_size=win, thres=thres, dec=dec)
    assert len(z) == 2 and len(z[0]) == len(z[1]) == ns
    for role in (0, 1):
        for i in range(ns):
            np.testing.assert_array_equal(z[role][i], zr[role][i])
    np.testing.assert_array_equal(u, ur)
    # independent copies per source and role (the reference appends .copy() each time)
    z[0][0][0, 0] = -1.
    assert z[1][0][0, 0] != -1.


def test_init_liv_quirk_first_k_extrema_not_the_loud_ones():
    """init_models.py:38-43: the threshold fixes the COUNT, the points are the first k turning points in time"""
    import gpitch_amd
    n, fs = 6000, 16000
    x = np.linspace(0., (n - 1.) / fs, n).reshape(-1, 1)
    y = np.sin(2 * np.pi * 200. * x) * np.where(x > x[n // 2], 1.0, 0.02)     # quiet first half, loud second half
    z, u = gpitch_amd.init_liv(x, y, win_size=9, thres=0.5, dec=1)
    k = u.size
    assert 0 < k
    # all returned points lie at the START of the signal although only the second half is above the threshold
    assert z[0][0].max() < x[n // 2]


def test_init_iv_matches_restatement():
    import gpitch_amd
    x, _, fs = _noisy_notes(0, n=5000)
    z = gpitch_amd.init_iv(x, 2, nivps_a=33, nivps_c=200, fs=fs)
    zr = ref.init_iv(x, 2, 33, 200, fs)
    for role in (0, 1):
        for i in range(2):
            np.testing.assert_array_equal(z[role][i], zr[role][i])
    assert z[0][0].shape == (len(x[::fs // 33]) + 1, 1) and z[0][0][-1, 0] == x[-1, 0]


def test_peak_indexes_matches_sequential_restatement():
    from gpitch_amd.methods import peak_indexes
    rng = np.random.RandomState(5)
    for trial in range(40):
        n = rng.randint(5, 200)
        y = np.round(rng.randn(n) * 2.) / 2.                     # coarse grid: many plateaus and ties in slope
        if trial % 4 == 0:
            y[: rng.randint(1, 4)] = y[0]                        # leading flat
            y[-rng.randint(1, 4):] = y[-1]                       # trailing flat
        thres, md = rng.rand() * 0.8, rng.randint(1, 12)
        np.testing.assert_array_equal(peak_indexes(y, thres, md), ref.peak_pick(y, thres, md))
    assert peak_indexes(np.ones(7)).size == 0
    y = np.array([0., 1., 0., 0., 2., 0., 1., 1., 3., 3., 3., 0.])
    assert list(peak_indexes(y, thres=0.1, min_dist=1)) == [1, 4, 9]      # flat top 8..10 -> its middle
    assert list(peak_indexes(y, thres=0.1, min_dist=3)) == [4, 9]         # 4 (height 2) shadows 1 (height 1)
    assert list(peak_indexes(y, thres=0.8, min_dist=1)) == [9]


def test_init_cparam_matches_restatement_and_keeps_its_quirks():
    from gpitch_amd.methods import init_cparam, peak_indexes
    fs, n, f0 = 16000, 8000, 250.
    t = np.arange(n) / float(fs)
    amps = [1.0, 0.5, 0.25, 0.7, 0.1]
    y = sum(a * np.sin(2 * np.pi * (k + 1) * f0 * t) for k, a in enumerate(amps)) + 0.3 * np.sin(2 * np.pi * 60. * t)
    y = y + 1e-4 * np.random.RandomState(0).randn(n)
    for maxh in (3, 25):
        freq, var, F, S, thres = init_cparam(y, fs, maxh=maxh, ideal_f0=f0)
        logS = np.log(S) + abs(np.log(S).min())                       # methods.py:110-112
        logS = logS / logS.max()
        idx = peak_indexes(logS, thres=thres, min_dist=0.8 * int(np.argmin(np.abs(F - f0))))
        fr, vr = ref.cparam_select(F, S, idx, maxh, f0)
        np.testing.assert_array_equal(freq, fr)
        np.testing.assert_array_equal(var, vr)
        assert np.all(np.diff(freq) > 0) and abs(var.sum() - 1.) < 1e-12 and freq.size <= maxh
    # quirk: a strong LOW peak (60 Hz < 0.75 f0) is NOT removed, because only the highest-frequency peak is examined
    freq, var, F, S, _ = init_cparam(y, fs, maxh=25, ideal_f0=f0, min_dis=0.1)
    assert freq[0] < 0.75 * f0
    # unscaled variances are the raw spectral magnitudes
    _, var_raw, _, S, _ = init_cparam(y, fs, maxh=3, ideal_f0=f0, scaled=False)
    assert np.all(np.isin(var_raw, S))
    with pytest.raises(ValueError):
        init_cparam(np.r_[1., np.zeros(63)], fs, maxh=3, ideal_f0=f0)   # an impulse: flat spectrum, no peak


@pytest.mark.parametrize("n,ws", [(10000, 2001), (8192, 1025), (5000, 2001), (4003, 2001), (12000, 2000)])
def test_windowing_matches_restatement(n, ws):
    from gpitch_amd import window_overlap as wo
    rng = np.random.RandomState(n)
    x = np.linspace(0., 1., n).reshape(-1, 1)
    y = rng.randn(n, 1)
    xw, yw = wo.windowed(x, y, ws)
    xr, yr = ref.windowed(x, y, ws)
    assert len(xw) == len(xr) == (n - ws) // ((ws - 1) // 2) + 1
    for a, b in zip(xw + yw, xr + yr):
        assert a.shape == (ws, 1)
        np.testing.assert_array_equal(a, b)
    np.testing.assert_array_equal(wo.merged_x(xw, ws), ref.merged_x(xr, ws))
    if ws % 2 == 1:   # odd windows tile the axis exactly: merged_x gives back the covered part of x
        m = wo.merged_x(xw, ws)
        np.testing.assert_array_equal(m, x[:m.size])


def test_windowed_short_signal_and_segmented():
    from gpitch_amd import window_overlap as wo
    x = np.arange(100.).reshape(-1, 1)
    assert wo.windowed(x, x, 201) == ([], [])
    n = 10000
    x = np.linspace(0., 1., n).reshape(-1, 1)
    y = np.random.RandomState(1).randn(n, 1)
    for aug in (False, True):
        xs, ys = wo.segmented(x, y, window_size=3000, aug=aug)
        xr, yr = ref.segmented(x, y, window_size=3000, aug=aug)
        assert len(xs) == len(xr) == 3
        for a, b in zip(xs + ys, xr + yr):
            assert a.shape == b.shape
            np.testing.assert_array_equal(a, b)
    xa, ya = wo.augmentate(x, y, augment_size=160)
    assert ya.shape == (n + 320, 1) and np.all(ya[:160] == 0) and np.all(ya[-160:] == 0)
    assert abs(xa[0, 0] - (x[0, 0] - 0.01)) < 1e-15 and xa.shape == (n + 320, 1)


def test_merge_sources_is_the_reference_overlap_add_of_every_source():
    """windows.merge_sources = SoSp.predict_s's merging (separation.py:340-368) of per-window source posteriors"""
    from gpitch_amd import window_overlap
    from gpitch_amd.windows import merge_sources
    rng = np.random.RandomState(0)
    ws, n = 21, 101
    x = np.linspace(0, 1, n).reshape(-1, 1)
    xw, _ = window_overlap.windowed(x, x, ws)
    res = [{"smean": [rng.randn(ws, 1) for _ in range(3)], "svar": [rng.rand(ws, 1) for _ in range(3)]} for _ in xw]
    out = merge_sources(res, ws, n)
    assert len(out) == 3
    for k in range(3):
        np.testing.assert_array_equal(out[k][0], window_overlap.merged_mean([r["smean"][k] for r in res], ws, n))
        np.testing.assert_array_equal(out[k][1], window_overlap.merged_variance([r["svar"][k] for r in res], ws, n))
        assert out[k][0].shape == (n, 1)


def test_readaudio_matches_the_reference_reader_on_its_demo_recording(tmp_path):
    """methods.py:36-54 through a float32 and a 16-bit file: samples as soundfile.read hands them out (float64; PCM
    divided by full scale), the time axis of :53, mono mix-down, `scaled` and `aug`."""
    from scipy.io import wavfile
    import gpitch_amd
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "init_liv_real_audio.npz"))
    y32, fs = d["y"], int(d["fs"])
    f = str(tmp_path / "a.wav")
    wavfile.write(f, fs, y32)
    x, y, fs2 = gpitch_amd.readaudio(f)
    assert fs2 == fs and y.dtype == np.float64 and y.shape == (y32.size, 1) == x.shape
    np.testing.assert_array_equal(y[:, 0], y32.astype(np.float64))
    np.testing.assert_array_equal(x[:, 0], np.linspace(0., (y32.size - 1.) / fs, y32.size))
    pcm = np.round(y32[:4000] * 32767).astype(np.int16)
    wavfile.write(f, fs, np.stack([pcm, pcm // 2], 1))                       # stereo PCM16
    x, y, _ = gpitch_amd.readaudio(f, frames=1000, start=100, aug=True, scaled=True)
    mono = np.mean(np.stack([pcm, pcm // 2], 1)[100:1100].astype(np.float64) / 32768., 1)
    assert y.shape == (2000, 1) and np.all(y[:1000] == 0)
    np.testing.assert_allclose(y[1000:, 0], mono / np.max(np.abs(mono)), rtol=0, atol=1e-15)
    assert x[-1, 0] == (2000 - 1.) / fs
