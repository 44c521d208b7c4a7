"""Host-side helpers mirroring gpitch/methods.py (nonlinearities :197-233, midi2freq :266-270)."""
import numpy as np

from . import _lib


def norm(x):
    """methods.py:193-195"""
    return x / np.max(np.abs(x))


def logistic(x):
    """methods.py:197-199 — NOTE shifted and scaled: 1/(1+exp(-2(x-pi)))"""
    return 1. / (1. + np.exp(-2. * (x - np.pi)))


def ilogistic(x):
    """methods.py:201-203"""
    return - np.log(1. / x - 1.)


def softplus(x):
    """methods.py:205-207"""
    return np.log(np.exp(x) + 1.)


def isoftplus(x):
    """methods.py:209-211"""
    return np.log(np.exp(x) - 1.)


def gaussfun(x):
    """methods.py:213-214"""
    return np.exp(-2. * (x - np.pi) ** 2)


class _Nonlinearity(object):
    """Stand-in for the reference's `*_tf` graph functions (methods.py:216-233): callable on host
    arrays, and carries the code the HIP likelihood kernel switches on."""

    def __init__(self, name, code, fn):
        self.__name__ = name
        self.code = code
        self._fn = fn

    def __call__(self, x):
        return self._fn(np.asarray(x, dtype=np.float64))

    def __repr__(self):
        return "<gpitch_amd nonlinearity %s>" % self.__name__


logistic_tf = _Nonlinearity("logistic_tf", _lib.NLIN_LOGISTIC, logistic)
softplus_tf = _Nonlinearity("softplus_tf", _lib.NLIN_SOFTPLUS, softplus)
gaussfun_tf = _Nonlinearity("gaussfun_tf", _lib.NLIN_GAUSS, gaussfun)


def nlin_code(nlinfun):
    code = getattr(nlinfun, "code", None)
    if code is None:
        raise TypeError("nlinfun must be one of gpitch_amd.logistic_tf / softplus_tf / gaussfun_tf "
                        "(arbitrary Python callables cannot run inside the HIP likelihood kernel)")
    return code


def midi2freq(midi):
    """methods.py:266-267"""
    return 2. ** ((midi - 69.) / 12.) * 440.


def freq2midi(freq):
    """methods.py:269-270"""
    return int(69. + 12. * np.log2(freq / 440.))


def find_ideal_f0(string):
    """Ideal fundamental of every file name carrying an 'M<midi>' tag (MAPS naming), 21 <= midi <= 108
    (methods.py:26-33).  Plain substring search, as in the reference: 'M21' also matches inside 'M210', and a name
    with several tags contributes several entries."""
    return [midi2freq(midi) for name in string for midi in range(21, 109) if ("M%d" % midi) in name]


def readaudio(fname, frames=-1, start=0, aug=False, scaled=False):
    """methods.py:36-54: wav file -> (x, y, fs) with y (n,1) float64, x = linspace(0, (n-1)/fs, n) (n,1).
    The reference reads through `soundfile.read` (float64 output: integer PCM divided by its full scale, float
    files as stored); here the RIFF container is parsed by scipy.io.wavfile and scaled the same way.  Stereo is
    averaged to mono (:40-41), `scaled` divides by max|y| (:43-47), `aug` prepends 1000 zeros (:48-50)."""
    from scipy.io import wavfile
    fs, data = wavfile.read(fname)
    if data.dtype.kind == "i":
        y = data.astype(np.float64) / float(2 ** (8 * data.dtype.itemsize - 1))
    elif data.dtype.kind == "u":                      # 8-bit PCM is unsigned, centred on 128
        y = (data.astype(np.float64) - 128.) / 128.
    else:
        y = data.astype(np.float64)
    y = y[start:] if frames < 0 else y[start:start + frames]
    if len(y.shape) == 1:
        y = y.reshape(-1, 1)
    if y.shape[1] == 2:  # convert to mono
        y = np.mean(y, 1)
    y = y.reshape(-1, 1)
    if scaled:
        beta = np.max(np.abs(y))
        if beta == 0.:
            beta = 1.
        y /= beta
    if aug:
        y = np.append(np.zeros((1000, 1)), y).reshape(-1, 1)
    n = y.size
    x = np.linspace(0., (n - 1.) / fs, n).reshape(-1, 1)
    return x, y, int(fs)


def _fill_flat_steps(step):
    """First differences with every run of zeros replaced by a neighbouring non-zero slope: a leading run takes the
    slope that follows it, a trailing run the slope before it, an interior run (a flat top or bottom) the left slope
    for its first half and the right slope from its middle on — so a plateau yields one sign change, at its centre."""
    n = step.size
    pos = np.arange(n)
    live = step != 0
    before = np.maximum.accumulate(np.where(live, pos, -1))              # nearest non-zero step at or before i
    after = np.minimum.accumulate(np.where(live, pos, n)[::-1])[::-1]    # nearest at or after i
    flat = ~live
    centre = 0.5 * ((before + 1) + (after - 1))                          # median index of the zero run i lies in
    from_left = flat & (before >= 0) & ((after >= n) | (pos < centre))
    from_right = flat & ~from_left
    out = step.copy()
    out[from_left] = step[before[from_left]]
    out[from_right] = step[np.minimum(after[from_right], n - 1)]
    return out


def peak_indexes(y, thres=0.3, min_dist=1):
    """Indices of the local maxima of `y` that rise above `thres` (a fraction of the data range) and are at least
    `min_dist` samples from any higher one.  Behaves as `peakutils.indexes(y, thres, min_dist)` of PeakUtils 1.x
    (MIT licence; third-party, called at methods.py:115, not part of the reference tree), written from its
    documented algorithm: slope sign change with flat tops resolved to their middle, then greedy suppression from
    the highest peak down."""
    y = np.asarray(y, dtype=np.float64)
    level = thres * (y.max() - y.min()) + y.min()
    step = np.diff(y)
    if not np.any(step):
        return np.array([], dtype=int)
    step = _fill_flat_steps(step)
    rising_in = np.concatenate([[False], step > 0.])
    falling_out = np.concatenate([step < 0., [False]])
    peaks = np.flatnonzero(rising_in & falling_out & (y > level))
    reach = int(min_dist)
    if peaks.size > 1 and reach > 1:
        kept = np.zeros(y.size, dtype=bool)
        kept[peaks] = True
        for p in peaks[np.argsort(y[peaks])][::-1]:          # highest first; a surviving peak clears its neighbourhood
            if kept[p]:
                kept[max(0, p - reach):p + reach + 1] = False
                kept[p] = True
        peaks = np.flatnonzero(kept)
    return peaks


def init_cparam(y, fs, maxh, ideal_f0, scaled=True, win_size=10, thres=0.1, min_dis=0.8):
    """Partial frequencies and (normalised) variances for a component kernel from the magnitude spectrum of one
    training note: peaks of the log-spectrum at least `min_dis` x (bin of ideal_f0) apart, the `maxh` strongest,
    returned in order of frequency as [frequencies, variances, F, S, threshold].  (methods.py:91-153)

    Quirks of the reference kept on purpose: its low-frequency filter (:124-130) re-derives its result from the
    unfiltered arrays on every pass, so in effect only the HIGHEST-frequency peak is examined (and dropped when it
    lies below 0.75 f0); the log-spectrum is shifted by |min| rather than by -min; `win_size` is accepted and, as
    there (the smoothed spectrum of :107-108 is never used), has no effect.  With no peak at all the reference
    dies on an unbound name; here that is a ValueError."""
    y = np.asarray(y, dtype=np.float64).reshape(-1)
    half = y.size // 2
    S = (2. / y.size) * np.abs(np.fft.fft(y)[:half])
    F = np.linspace(0, fs / 2., half)
    logS = np.log(S)
    logS = logS + np.abs(logS.min())
    logS = logS / logS.max()
    thres = thres * logS.max()
    f0_bin = int(np.argmin(np.abs(F - ideal_f0)))
    at = peak_indexes(logS, thres=thres, min_dist=min_dis * f0_bin)
    if at.size == 0:
        raise ValueError("init_cparam: no spectral peak above the threshold")
    by_freq = np.argsort(F[at])
    pk_f, pk_s = F[at][by_freq], S[at][by_freq]
    if pk_f[-1] < 0.75 * ideal_f0:
        pk_f, pk_s = pk_f[:-1], pk_s[:-1]
    strongest = np.argsort(pk_s)[::-1][:maxh]
    var = pk_s[strongest]
    if scaled:
        var = var * (1. / np.sum(var))
    order = np.argsort(pk_f[strongest])
    return [pk_f[strongest][order], var[order], F, S, thres]
