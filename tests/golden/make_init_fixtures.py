"""Generates tests/golden/init_liv_real_audio.npz (build container only: it reads the reference's DATA file).

The reference holds one deterministic anchor for its host-side initialisers: on its shipped recording
demos/data/011PFNOF_M60_train.wav (32000 samples, 16 kHz, mono float32),
    z, u = gpitch.init_liv(x=x, y=y, win_size=31, thres=0.033, dec=9)
prints "number of inducing variables 109" (demos/notebooks/demo_modgp-real-audio.ipynb:88,116), and
find_ideal_f0 of the file name prints [261.6255653005986] (:66).  The fixture stores the samples (data, not code),
the time axis parameters as gpitch.readaudio builds them (gpitch/methods.py:36-54: x = linspace(0, (n-1)/fs, n)),
and those two printed values.

    python tests/golden/make_init_fixtures.py
"""
import os

import numpy as np
from scipy.io import wavfile

HERE = os.path.dirname(os.path.abspath(__file__))
WAV = "/root/reference/demos/data/011PFNOF_M60_train.wav"


def main():
    fs, y = wavfile.read(WAV)
    assert y.ndim == 1 and y.dtype == np.float32, (y.shape, y.dtype)
    np.savez_compressed(os.path.join(HERE, "init_liv_real_audio.npz"),
                        y=y, fs=np.int64(fs), fname="011PFNOF_M60_train.wav",
                        win_size=31, thres=0.033, dec=9, expected_num_inducing=109,
                        expected_ideal_f0=261.6255653005986)
    print("wrote init_liv_real_audio.npz: %d samples at %d Hz" % (y.size, fs))


if __name__ == "__main__":
    main()
