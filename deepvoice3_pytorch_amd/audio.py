# coding: utf-8
"""Spectrogram -> waveform on the GPU: the device-side counterpart of the reference's
audio.inv_spectrogram (audio.py:37-43) and its helpers (audio.py:26-28,84-93).

The reference hands phase reconstruction to the third-party `lws` package on the host
(audio.py:40-42); here it is Griffin-Lim on hand-written HIP FFT kernels (csrc/audio.hip), batched
over utterances, so synthesis never leaves the device (synthesis.py:64-71 copies the
spectrogram to the CPU first).  STFT conventions: see include/dv3hip.h ("Audio inverse").
"""
import numpy as np
import torch

from . import _lib
from .ops import _stream, _chk, _c

N_FFT = 1024
N_BIN = N_FFT // 2 + 1


class AudioConfig(object):
    """The hparams audio.py reads (hparams.py:38-43,124)."""

    def __init__(self, fft_size=1024, hop_size=256, sample_rate=22050, preemphasis=0.97,
                 min_level_db=-100, ref_level_db=20, power=1.4, griffin_lim_iters=60):
        if fft_size != N_FFT:
            raise ValueError("the HIP FFT kernels are built for fft_size=1024 (every reference preset)")
        self.fft_size, self.hop_size, self.sample_rate = fft_size, hop_size, sample_rate
        self.preemphasis, self.min_level_db, self.ref_level_db = preemphasis, min_level_db, ref_level_db
        self.power, self.griffin_lim_iters = power, griffin_lim_iters


def magnitudes(linear_outputs, cfg):
    """(B, T, 513) normalised spectrogram (the model's linear_outputs) -> magnitudes ** power."""
    x = _c(_chk(linear_outputs, "linear_outputs"))
    mag = torch.empty_like(x)
    _lib.call("dv3_gl_prepare_f32", x.data_ptr(), mag.data_ptr(), x.numel(), float(cfg.min_level_db),
              float(cfg.ref_level_db), float(cfg.power), _stream())
    return mag


def istft(mag, phasor, hop):
    """mag (B,T,513), phasor (B,T,513,2) or None -> y (B, hop*(T-1))."""
    B, T, F = mag.shape
    assert F == N_BIN
    frames = torch.empty((B, T, N_FFT), dtype=torch.float32, device=mag.device)
    _lib.call("dv3_istft_frames_f32", mag.data_ptr(), phasor.data_ptr() if phasor is not None else None,
              frames.data_ptr(), B, T, _stream())
    y = torch.empty((B, hop * (T - 1)), dtype=torch.float32, device=mag.device)
    _lib.call("dv3_overlap_add_f32", frames.data_ptr(), y.data_ptr(), B, T, hop, _stream())
    return y


def stft(y, T, hop, want_phasor=True, want_spec=False):
    """y (B, hop*(T-1)) -> unit phasors and/or the complex STFT, each (B,T,513,2)."""
    y = _c(_chk(y, "y"))
    B = y.shape[0]
    assert y.shape[1] == hop * (T - 1)
    ph = torch.empty((B, T, N_BIN, 2), dtype=torch.float32, device=y.device) if want_phasor else None
    sp = torch.empty((B, T, N_BIN, 2), dtype=torch.float32, device=y.device) if want_spec else None
    _lib.call("dv3_stft_phase_f32", y.data_ptr(), ph.data_ptr() if ph is not None else None,
              sp.data_ptr() if sp is not None else None, None, B, T, hop, _stream())
    return ph, sp


# ---------------------------------------------------------------------------------------------
# forward analysis: audio.spectrogram / audio.melspectrogram (audio.py:31-35,46-51)
# ---------------------------------------------------------------------------------------------
def mel_basis(sample_rate=22050, n_fft=1024, n_mels=80, fmin=125.0, fmax=7600.0):
    """The Slaney-style triangular filterbank librosa.filters.mel builds by default (htk=False,
    norm='slaney'), restated: audio.py:70-76 with hparams.py fmin/fmax.  (num_mels, 513) float32."""
    def hz_to_mel(f):
        f = np.asarray(f, dtype=np.float64)
        mel = f / (200.0 / 3)
        lin_end = 1000.0 / (200.0 / 3)
        logstep = np.log(6.4) / 27.0
        return np.where(f >= 1000.0, lin_end + np.log(np.maximum(f, 1e-30) / 1000.0) / logstep, mel)

    def mel_to_hz(m):
        m = np.asarray(m, dtype=np.float64)
        lin_end = 1000.0 / (200.0 / 3)
        logstep = np.log(6.4) / 27.0
        return np.where(m >= lin_end, 1000.0 * np.exp(logstep * (m - lin_end)), m * (200.0 / 3))
    fftfreqs = np.linspace(0, sample_rate / 2.0, n_fft // 2 + 1)
    mel_f = mel_to_hz(np.linspace(hz_to_mel(fmin), hz_to_mel(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fftfreqs[None, :]
    lower = -ramps[:-2] / fdiff[:-1, None]
    upper = ramps[2:] / fdiff[1:, None]
    w = np.maximum(0, np.minimum(lower, upper))
    w *= (2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels]))[:, None]
    return w.astype(np.float32)


def _analysis_mag(wav, cfg):
    """(B, L) waveform -> |STFT(preemphasis(wav))| as (B, 513, T); L must be hop*(T-1)."""
    wav = _c(_chk(wav, "wav"))
    B, L = wav.shape
    hop = cfg.hop_size
    if L % hop:
        raise ValueError("waveform length must be a multiple of hop_size (%d)" % hop)
    T = L // hop + 1
    pre = torch.empty_like(wav)
    _lib.call("dv3_preemphasis_f32", wav.data_ptr(), pre.data_ptr(), B, L, float(cfg.preemphasis), _stream())
    mag = torch.empty((B, N_BIN, T), dtype=torch.float32, device=wav.device)
    _lib.call("dv3_stft_phase_f32", pre.data_ptr(), None, None, mag.data_ptr(), B, T, hop, _stream())
    return mag


def _db_norm(x, cfg):
    out = torch.empty_like(x)
    _lib.call("dv3_amp_to_db_norm_f32", x.data_ptr(), out.data_ptr(), x.numel(), float(cfg.min_level_db),
              float(cfg.ref_level_db), _stream())
    return out


def spectrogram_batch(wav, cfg=None):
    """audio.spectrogram (audio.py:31-35) for a (B, L) device batch -> (B, 513, T) in [0, 1]."""
    cfg = cfg or AudioConfig()
    return _db_norm(_analysis_mag(wav, cfg), cfg)


def melspectrogram_batch(wav, cfg=None, num_mels=80, fmin=125.0, fmax=7600.0):
    """audio.melspectrogram (audio.py:46-51) for a (B, L) device batch -> (B, num_mels, T) in [0, 1]:
    the filterbank product runs on the tap-GEMM kernel as a 1x1 convolution over the 513 bins."""
    from . import ops
    cfg = cfg or AudioConfig()
    mag = _analysis_mag(wav, cfg)
    B, F, T = mag.shape
    basis = torch.from_numpy(mel_basis(cfg.sample_rate, cfg.fft_size, num_mels, fmin, fmax)).to(wav.device)
    pk = ops.pack_weights(basis.unsqueeze(-1).contiguous(), None, need_bwd=False)
    mel = ops.conv_gemm(mag, pk.fwd, pk.lda, 0, B=B, Cin=F, Tin=T, M=num_mels, Tout=T, a_split=pk.fwd_s)
    return _db_norm(mel, cfg)


def griffin_lim(mag, hop, n_iter, init_phasor=None):
    """Griffin & Lim: alternate projections between the given magnitudes and consistent STFTs."""
    y = istft(mag, init_phasor, hop)
    B, T, _ = mag.shape
    if n_iter > 0:
        frames = torch.empty((B, T, N_FFT), dtype=torch.float32, device=mag.device)
        y2 = torch.empty_like(y)
        for _ in range(n_iter):
            # stft -> unit phase -> x magnitude -> inverse FFT -> window in one launch (the phasors never reach HBM)
            _lib.call("dv3_gl_project_f32", y.data_ptr(), mag.data_ptr(), frames.data_ptr(), B, T, hop, _stream())
            _lib.call("dv3_overlap_add_f32", frames.data_ptr(), y2.data_ptr(), B, T, hop, _stream())
            y, y2 = y2, y
    return y


def inv_preemphasis_(y, coef):
    """de-emphasis filter (audio.py:26-28) of (B, L) waveforms -> a new tensor"""
    out = torch.empty_like(y)
    _lib.call("dv3_deemphasis_f32", y.data_ptr(), out.data_ptr(), y.shape[0], y.shape[1], float(coef), _stream())
    return out


def inv_spectrogram_batch(linear_outputs, cfg=None, init_phasor=None):
    """(B, T, 513) device tensor (model linear_outputs) -> (B, hop*(T-1)) waveforms on the device."""
    cfg = cfg or AudioConfig()
    mag = magnitudes(linear_outputs, cfg)
    y = griffin_lim(mag, cfg.hop_size, cfg.griffin_lim_iters, init_phasor)
    return inv_preemphasis_(y, cfg.preemphasis)


def inv_spectrogram(spectrogram, cfg=None, device="cuda:0"):
    """Drop-in for audio.inv_spectrogram (audio.py:37-43): (513, T) numpy -> waveform numpy."""
    s = torch.as_tensor(np.ascontiguousarray(np.asarray(spectrogram, dtype=np.float32).T)).unsqueeze(0)
    y = inv_spectrogram_batch(s.to(device), cfg)
    return y[0].cpu().numpy()
