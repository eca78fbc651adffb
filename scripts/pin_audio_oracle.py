#!/usr/bin/env python
# coding: utf-8
"""Pin oracle/audio_oracle.py's restatement of the lws framing to the REAL package.

Run once on any box where `pip install lws` (and optionally `librosa`) works -- neither can be installed in the build
image (no network):

    python scripts/pin_audio_oracle.py            # writes tests/golden/audio_lws.npz

The file holds inputs and the package's own outputs for the reference's call sites (audio.py:31-35,46-55,74-76):
`lws.lws(1024, 256, mode="speech")`: its analysis window `awin`, `stft(x)` and `istft(S)` for four signal lengths, and
`librosa.filters.mel` for the preset's mel parameters.  tests/test_audio.py::test_oracle_lws_framing_against_pinned_vectors
then holds the restatement (window amplitude included: AudioConfig(window_scale=...)) to those vectors on every box, and
SURVEY rows a16 / f2 stop being "parity unpinned" for the framing.  Nothing of the package's source is copied: the file
is data (inputs + outputs)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    try:
        import lws
    except ImportError:
        sys.exit("the `lws` package is not importable here: run this on a box where it is installed")
    proc = lws.lws(1024, 256, mode="speech")
    rng = np.random.RandomState(1234)
    out = {"awin": np.asarray(proc.awin, dtype=np.float64)}
    sigs = [rng.randn(L) * 0.1 for L in (2560, 256 * 37, 5000, 1000)]
    out["n"] = np.int64(len(sigs))
    for i, x in enumerate(sigs):
        S = proc.stft(x)
        out["x%d" % i], out["S%d" % i], out["y%d" % i] = x, S, proc.istft(S)
    try:
        import librosa
        out["mel"] = librosa.filters.mel(sr=22050, n_fft=1024, n_mels=80, fmin=125, fmax=7600)
    except ImportError:
        print("librosa not importable: the mel basis stays pinned to the independent construction only")
    path = os.path.join(ROOT, "tests", "golden", "audio_lws.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, "| awin[512]^2 =", float(out["awin"][512] ** 2),
          "(0.5 = hop-normalised, 1.0 = plain sqrt-Hann)")


if __name__ == "__main__":
    main()
