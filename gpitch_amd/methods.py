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
    """methods.py:26-33: ideal f0 of every file name that carries an 'M<midi>' tag (MAPS naming).
    ('M21' also matches inside 'M210'...: the reference's substring test is kept as is.)"""
    ideal_f0 = []
    for j in range(len(string)):
        for i in range(21, 109):
            if string[j].find('M' + str(i)) != -1:
                ideal_f0.append(midi2freq(i))
    return ideal_f0


def peak_indexes(y, thres=0.3, min_dist=1):
    """Peak picker used by init_cparam (methods.py:115 calls `peakutils.indexes`, a third-party package that is
    not vendored in the reference; restated here from its published algorithm, peakutils 1.x): first-difference
    sign change, amplitude above `thres` of the data range, flat tops resolved towards their middle, then greedy
    suppression of weaker peaks within `min_dist` samples of a stronger one."""
    y = np.asarray(y, dtype=np.float64)
    thres = thres * (np.max(y) - np.min(y)) + np.min(y)
    min_dist = int(min_dist)
    dy = np.diff(y)
    zeros, = np.where(dy == 0)
    if len(zeros) == len(y) - 1:
        return np.array([], dtype=int)
    if len(zeros):
        zeros_diff = np.diff(zeros)
        zeros_diff_not_one, = np.add(np.where(zeros_diff != 1), 1)
        zero_plateaus = np.split(zeros, zeros_diff_not_one)
        if zero_plateaus[0][0] == 0:
            dy[zero_plateaus[0]] = dy[zero_plateaus[0][-1] + 1]
            zero_plateaus.pop(0)
        if len(zero_plateaus) and zero_plateaus[-1][-1] == len(dy) - 1:
            dy[zero_plateaus[-1]] = dy[zero_plateaus[-1][0] - 1]
            zero_plateaus.pop(-1)
        for plateau in zero_plateaus:
            median = np.median(plateau)
            dy[plateau[plateau < median]] = dy[plateau[0] - 1]
            dy[plateau[plateau >= median]] = dy[plateau[-1] + 1]
    peaks = np.where((np.hstack([dy, 0.]) < 0.) & (np.hstack([0., dy]) > 0.) & (np.greater(y, thres)))[0]
    if peaks.size > 1 and min_dist > 1:
        highest = peaks[np.argsort(y[peaks])][::-1]
        rem = np.ones(y.size, dtype=bool)
        rem[peaks] = False
        for peak in highest:
            if not rem[peak]:
                sl = slice(max(0, peak - min_dist), peak + min_dist + 1)
                rem[sl] = True
                rem[peak] = False
        peaks = np.arange(y.size)[~rem]
    return peaks


def init_cparam(y, fs, maxh, ideal_f0, scaled=True, win_size=10, thres=0.1, min_dis=0.8):
    """Component-kernel parameters from the spectrum of a training note (methods.py:91-153): frequencies and
    (normalised) variances of up to `maxh` spectral peaks.  Quirks kept: the low-frequency filter loop
    (:124-130) overwrites its result on every pass, so only the LAST peak is ever dropped (when it lies below
    0.75 f0); the smoothed spectrum `Ss` (:108) is computed and never used."""
    from scipy.fftpack import fft
    y = np.asarray(y, dtype=np.float64)
    N = y.size
    Y = fft(y.reshape(-1,))
    S = 2. / N * np.abs(Y[0:N // 2])
    F = np.linspace(0, fs / 2., N // 2)
    Sslog = np.log(S)
    Sslog = Sslog + np.abs(np.min(Sslog))
    Sslog /= np.max(Sslog)
    thres = thres * np.max(Sslog)
    min_dist = min_dis * np.argmin(np.abs(F - ideal_f0))
    idx = peak_indexes(Sslog, thres=thres, min_dist=min_dist)
    F_star, S_star = F[idx], S[idx]
    idx_sorted = np.argsort(F_star.copy())
    S_star = S_star[idx_sorted]
    F_star = np.sort(F_star)
    F_star2, S_star2 = F_star.copy(), S_star.copy()
    for index in range(F_star.size):
        if F_star[index] < 0.75 * ideal_f0:
            F_star2 = np.delete(F_star, [index])
            S_star2 = np.delete(S_star, [index])
        else:
            F_star2 = F_star.copy()
            S_star2 = S_star.copy()
    aux1 = np.flip(np.sort(S_star2), 0)
    aux2 = np.flip(np.argsort(S_star2), 0)
    if aux1.size > maxh:
        vvec = aux1[0:maxh]
        idxf = aux2[0:maxh]
    else:
        vvec = aux1
        idxf = aux2
    if scaled:
        vvec = vvec * (1. / np.sum(vvec))
    freq_final = F_star2[idxf]
    var_final = vvec
    idx_sorted = np.argsort(freq_final.copy())
    var_final = var_final[idx_sorted]
    freq_final = np.sort(freq_final)
    return [freq_final, var_final, F, S, thres]
