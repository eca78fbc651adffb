# coding: utf-8
"""Audio inverse: audio.inv_spectrogram (audio.py:37-43) on the device.

Phase reconstruction is PARITY UNPINNED against the reference (third-party lws, oracle/audio_oracle.py
header); what can be pinned is: the reference's own amp/db helpers (tests/test_audio.py:15-20 of the
reference), the HIP STFT / iSTFT against torch's FFTs, and the HIP Griffin-Lim against the independent
CPU restatement on identical inputs."""
import numpy as np
import pytest
import torch

from oracle import audio_oracle as A


def test_oracle_amp_db_and_denormalize_known_answers():
    # audio.py:84-93 with hparams.py:42-43 (min_level_db=-100, ref_level_db=20)
    assert np.allclose(A.denormalize(np.array([0.0, 0.5, 1.0, 1.7, -3.0])), [-100, -50, 0, 0, -100])
    assert np.allclose(A.db_to_amp(np.array([0.0, 20.0, -20.0])), [1.0, 10.0, 0.1])
    # reference tests/test_audio.py:15-20: _db_to_amp(_amp_to_db(x)) == x
    x = np.random.RandomState(0).rand(100) + 1e-3
    assert np.allclose(A.db_to_amp(20 * np.log10(x)), x)
    assert np.allclose(A.magnitudes(np.array([1.0])), 10.0 ** 1.4)


def test_oracle_helpers_match_reference_golden():
    """oracle/audio_oracle.py against the reference's own audio._amp_to_db / _db_to_amp / _normalize /
    _denormalize run unmodified under the ljspeech preset (tests/golden/audio_helpers.npz, written by
    oracle/make_golden.py:gen_audio_helpers): bit exact, including values below min_level and outside
    the clipping range, and the two composite chains inv_spectrogram / spectrogram apply around lws."""
    from tests.util import load_golden
    fx = load_golden("audio_helpers")
    mn, ref, power = float(fx["hp/min_level_db"]), float(fx["hp/ref_level_db"]), float(fx["hp/power"])
    assert (mn, ref, power, float(fx["hp/preemphasis"])) == (-100.0, 20.0, 1.4, 0.97)   # the defaults the oracle carries
    assert np.array_equal(A.amp_to_db(fx["in/amp"], mn), fx["out/amp_to_db"])
    assert np.array_equal(A.amp_to_db(fx["in/amp32"], mn), fx["out/amp_to_db32"])
    assert np.array_equal(A.db_to_amp(fx["in/db"]), fx["out/db_to_amp"])
    assert np.array_equal(A.normalize(fx["in/db"], mn), fx["out/normalize"])
    d = A.denormalize(fx["in/norm"], mn)
    assert d.dtype == fx["out/denormalize"].dtype and np.array_equal(d, fx["out/denormalize"])
    assert np.array_equal(A.magnitudes(fx["in/norm"], mn, ref, power), fx["out/inv_mag"])
    assert np.array_equal(A.normalize(A.amp_to_db(fx["in/amp"], mn) - ref, mn), fx["out/spec_norm"])


def test_oracle_stft_istft_roundtrip_and_griffin_lim_converges():
    rng = np.random.RandomState(1)
    T, hop = 40, 256
    y = torch.from_numpy(rng.randn(2, hop * (T - 1))).double()
    Z = A.stft(y, hop)
    assert Z.shape == (2, T, 513)
    # iSTFT inverts the STFT away from the reflect-padded borders
    y2 = A.istft(Z, hop)
    assert float((y2 - y).abs().max()) < 1e-9
    mag = Z.abs()
    sc = [A.spectral_convergence(A.griffin_lim(mag, n, hop), mag, hop) for n in (0, 5, 30)]
    assert sc[2] < sc[1] < sc[0]


def test_oracle_inv_preemphasis_inverts_preemphasis():
    x = np.random.RandomState(2).randn(3, 1000)
    pre = np.concatenate([x[:, :1], x[:, 1:] - 0.97 * x[:, :-1]], axis=1)     # nnmnkwii.preemphasis
    assert np.allclose(A.inv_preemphasis(pre, 0.97), x)


# ------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def _rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / np.abs(b).max())


@pytest.mark.gpu
@pytest.mark.parametrize("B,T,hop", [(1, 9, 256), (3, 37, 256), (2, 20, 200)])
def test_hip_stft_istft_match_torch_fft(dev, B, T, hop):
    from deepvoice3_pytorch_amd import audio
    rng = np.random.RandomState(B * 100 + T)
    y = torch.from_numpy(rng.randn(B, hop * (T - 1)).astype(np.float32))
    ph, sp = audio.stft(y.to(dev), T, hop, want_phasor=True, want_spec=True)
    Z = A.stft(y.double(), hop)
    got = torch.view_as_complex(sp.cpu().double().contiguous())
    assert _rel(torch.view_as_real(got).numpy(), torch.view_as_real(Z).numpy()) < 2e-6
    gph = torch.view_as_complex(ph.cpu().double().contiguous())
    big = Z.abs() > 1e-3 * Z.abs().max()
    assert float(((gph - Z / Z.abs())[big]).abs().max()) < 1e-3
    # inverse: arbitrary (non-Hermitian-consistent) magnitudes + phases, like a Griffin-Lim step
    mag = torch.from_numpy(rng.rand(B, T, 513).astype(np.float32))
    phz = torch.from_numpy(rng.uniform(-np.pi, np.pi, (B, T, 513)).astype(np.float32))
    phasor = torch.stack([torch.cos(phz), torch.sin(phz)], dim=-1)
    yg = audio.istft(mag.to(dev), phasor.to(dev), hop)
    want = A.istft(mag.double() * torch.view_as_complex(phasor.double().contiguous()), hop)
    assert _rel(yg.cpu().numpy(), want.numpy()) < 2e-5
    yz = audio.istft(mag.to(dev), None, hop)                       # zero phase
    assert _rel(yz.cpu().numpy(), A.istft(mag.double().to(torch.complex128), hop).numpy()) < 2e-5


@pytest.mark.gpu
def test_hip_griffin_lim_matches_oracle_and_converges(dev):
    from deepvoice3_pytorch_amd import audio
    rng = np.random.RandomState(5)
    B, T, hop = 2, 48, 256
    # a plausible normalised spectrogram: smooth random field in [0, 1]
    lin = torch.from_numpy(np.clip(0.55 + 0.25 * rng.randn(B, T, 513), -0.2, 1.2).astype(np.float32))
    cfg = audio.AudioConfig(griffin_lim_iters=4)
    mag = audio.magnitudes(lin.to(dev), cfg)
    assert _rel(mag.cpu().numpy(), A.magnitudes(lin.numpy())) < 2e-5
    phz = torch.from_numpy(rng.uniform(-np.pi, np.pi, (B, T, 513)).astype(np.float32))
    phasor = torch.stack([torch.cos(phz), torch.sin(phz)], dim=-1)
    init = torch.view_as_complex(phasor.double().contiguous())
    for n_iter in (1, 4):
        got = audio.griffin_lim(mag, hop, n_iter, phasor.to(dev))
        want = A.griffin_lim(mag.cpu().double(), n_iter, hop, init)
        assert _rel(got.cpu().numpy(), want.numpy()) < 5e-4, n_iter
    # more iterations: compare through the size-independent property (spectral convergence decreases)
    sc = [A.spectral_convergence(audio.griffin_lim(mag, hop, n, phasor.to(dev)).cpu().double(),
                                 mag.cpu().double(), hop) for n in (0, 10, 40)]
    assert sc[2] < sc[1] < sc[0]
    # a CONSISTENT magnitude (that of a real signal) must be approached much more closely
    sig = torch.from_numpy(np.cumsum(rng.randn(B, hop * (T - 1)), axis=1).astype(np.float32) * 0.05)
    cmag = A.stft(sig.double(), hop).abs().float().contiguous().to(dev)
    sc = [A.spectral_convergence(audio.griffin_lim(cmag, hop, n, phasor.to(dev)).cpu().double(),
                                 cmag.cpu().double(), hop) for n in (0, 60)]
    assert sc[1] < 0.5 * sc[0]


@pytest.mark.gpu
def test_hip_deemphasis_and_inv_spectrogram(dev):
    from deepvoice3_pytorch_amd import audio
    rng = np.random.RandomState(6)
    y = torch.from_numpy(rng.randn(3, 7000).astype(np.float32))
    got = audio.inv_preemphasis_(y.clone().to(dev), 0.97).cpu().numpy()
    assert _rel(got, A.inv_preemphasis(y.numpy(), 0.97)) < 2e-5
    # the chunked parallel form over many chunks (rows 16-byte aligned or not), a constant offset (the filter's DC gain
    # of 33 must survive the chunk restarts), and a coefficient whose memory outlasts the warm-up (serial form)
    for L, coef, dc in ((50000, 0.97, 0.0), (20481, 0.97, 0.5), (9000, 0.9995, 0.0), (4097, 0.5, 1.0)):
        y = torch.from_numpy((rng.randn(2, L) + dc).astype(np.float32))
        got = audio.inv_preemphasis_(y.to(dev), coef).cpu().numpy()
        # (a pole at 0.9995 amplifies fp32 rounding ~2000 x: the serial form is held to 1e-4 there)
        assert _rel(got, A.inv_preemphasis(y.numpy(), coef)) < (1e-4 if coef > 0.99 else 2e-5), (L, coef)
    # end to end, reference calling convention: (513, T) numpy in, waveform numpy out
    spec = np.clip(0.5 + 0.2 * rng.randn(513, 30), 0, 1).astype(np.float32)
    wav = audio.inv_spectrogram(spec, audio.AudioConfig(griffin_lim_iters=3))
    mag = A.magnitudes(spec.T[None])
    want = A.inv_preemphasis(A.griffin_lim(torch.from_numpy(mag), 3).numpy(), 0.97)[0]
    assert wav.shape == (256 * 29,) and np.isfinite(wav).all()
    assert _rel(wav, want) < 1e-3


def test_mel_basis_matches_independent_construction():
    from deepvoice3_pytorch_amd import audio
    W = audio.mel_basis()
    assert W.shape == (80, 513) and W.dtype == np.float32
    assert np.abs(W - A.slaney_mel_basis()).max() < 1e-6 * A.slaney_mel_basis().max()
    assert (W.sum(axis=1) > 0).all()          # no empty filter at 22.05 kHz / 1024 / 80 mels


@pytest.mark.gpu
def test_hip_spectrogram_and_melspectrogram(dev):
    """forward analysis (audio.py:31-35,46-51) on the device against the numpy/torch-FFT restatement"""
    from deepvoice3_pytorch_amd import audio
    rng = np.random.RandomState(8)
    B, T, hop = 2, 33, 256
    t = np.arange(hop * (T - 1)) / 22050.0
    wav = (0.3 * np.sin(2 * np.pi * 440 * t)[None] + 0.05 * rng.randn(B, t.size)).astype(np.float32)
    S = audio.spectrogram_batch(torch.from_numpy(wav).to(dev))
    want = A.spectrogram(wav.astype(np.float64))
    assert S.shape == (B, 513, T)
    assert np.abs(S.cpu().numpy() - want).max() < 2e-5            # values live in [0, 1]
    for mode in ("f16x3", "bf16x3", "f32"):
        from deepvoice3_pytorch_amd import ops
        prev = ops.set_gemm_precision(mode)
        M = audio.melspectrogram_batch(torch.from_numpy(wav).to(dev))
        ops.set_gemm_precision(prev)
        wantm = A.melspectrogram(wav.astype(np.float64))
        assert M.shape == (B, 80, T)
        assert np.abs(M.cpu().numpy() - wantm).max() < 2e-5, mode
