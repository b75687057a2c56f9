"""Mel-spectrogram L1 metric with the reference's definition — numpy only, TEST INFRASTRUCTURE.

reference mel_processing.py:95-142 (mel_spectrogram_torch): n_fft 2048, hop 512, win 2048, periodic hann, reflect
padding (n_fft-hop)/2 on both sides, center=False, magnitude sqrt(re^2+im^2+1e-6), 128 mel bins (fmin 0, fmax sr/2)
from librosa.filters.mel, then log(clamp(x, 1e-5)).  librosa is absent here, so the Slaney-scale, Slaney-normalised
filterbank (librosa's defaults htk=False, norm="slaney") is rebuilt below.
"""
import numpy as np


def _hz_to_mel(f):
    f = np.asarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-10) / min_log_hz) / logstep, mels)


def _mel_to_hz(m):
    m = np.asarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)


def mel_filterbank(sr=44100, n_fft=2048, n_mels=128, fmin=0.0, fmax=None):
    fmax = sr / 2 if fmax is None else fmax
    fftfreqs = np.linspace(0, sr / 2, 1 + n_fft // 2)
    mel_f = _mel_to_hz(np.linspace(_hz_to_mel(fmin), _hz_to_mel(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fftfreqs[None, :]
    lower = -ramps[:-2] / fdiff[:-1, None]
    upper = ramps[2:] / fdiff[1:, None]
    w = np.maximum(0, np.minimum(lower, upper))
    w *= (2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels]))[:, None]
    return w


_FB = {}


def log_mel(y, sr=44100, n_fft=2048, hop=512, n_mels=128):
    """y: [S] float -> [n_mels, frames]."""
    y = np.asarray(y, dtype=np.float64)
    pad = (n_fft - hop) // 2
    if len(y) <= pad:
        y = np.pad(y, (0, pad + 1 - len(y)))
    y = np.pad(y, (pad, pad), mode="reflect")
    n = 1 + (len(y) - n_fft) // hop
    win = 0.5 - 0.5 * np.cos(2 * np.pi * np.arange(n_fft) / n_fft)          # torch.hann_window (periodic)
    idx = np.arange(n_fft)[None, :] + hop * np.arange(n)[:, None]
    spec = np.fft.rfft(y[idx] * win[None, :], axis=1)
    mag = np.sqrt(spec.real ** 2 + spec.imag ** 2 + 1e-6).T
    key = (sr, n_fft, n_mels)
    if key not in _FB:
        _FB[key] = mel_filterbank(sr, n_fft, n_mels)
    return np.log(np.clip(_FB[key] @ mag, 1e-5, None))


def mel_l1(a, b, valid_samples, **kw):
    """mean |logmel(a) - logmel(b)| over the batch, each utterance cut to its valid samples."""
    tot, cnt = 0.0, 0
    for ai, bi, n in zip(a, b, valid_samples):
        n = int(n)
        ma, mb = log_mel(ai[:n], **kw), log_mel(bi[:n], **kw)
        tot += np.abs(ma - mb).sum()
        cnt += ma.size
    return tot / max(cnt, 1)
