"""Log-mel front-end with the reference's interface (scripts/featureExtractor.py:8-33): `mfsc`, `normalize`,
`extractFeatures` — computed with torch on the device (pre-emphasis, framed rFFT through rocFFT, Slaney mel filterbank,
log, mean normalisation), so `getEmbeddingExample.py`'s audio -> features -> embedding flow stays on the GPU.

The reference delegates the arithmetic to librosa 0.7.2 (`stft(center=False)`, `filters.mel(htk=False, norm=None)`),
which is absent from this image: the algorithm is restated from its published definition (oracle/feature_oracle.py is the
float64 restatement the tests compare against; parity with librosa itself is unpinned — DESIGN.md §7).
"""
import math

import numpy as np
import torch

__all__ = ["mfsc", "normalize", "extractFeatures", "mel_filterbank", "features_on_device"]

_F_SP = 200.0 / 3.0
_MIN_LOG_HZ = 1000.0
_MIN_LOG_MEL = _MIN_LOG_HZ / _F_SP
_LOGSTEP = math.log(6.4) / 27.0
_cache = {}


def _hz_to_mel(f):
    return _MIN_LOG_MEL + math.log(f / _MIN_LOG_HZ) / _LOGSTEP if f >= _MIN_LOG_HZ else f / _F_SP


def mel_filterbank(sr, n_fft, n_mels, fmin, fmax, device="cpu"):
    """[n_mels, 1 + n_fft//2] triangular Slaney-scale filters with unit peak (librosa.filters.mel, htk=False, norm=None)."""
    key = (float(sr), n_fft, n_mels, float(fmin), float(fmax), str(device))
    if key not in _cache:
        mels = torch.linspace(_hz_to_mel(fmin), _hz_to_mel(fmax), n_mels + 2, dtype=torch.float64)
        edges = torch.where(mels >= _MIN_LOG_MEL, _MIN_LOG_HZ * torch.exp(_LOGSTEP * (mels - _MIN_LOG_MEL)), _F_SP * mels)
        freqs = torch.linspace(0.0, sr / 2.0, 1 + n_fft // 2, dtype=torch.float64)
        ramps = edges[:, None] - freqs[None, :]
        width = edges[1:] - edges[:-1]
        rising = -ramps[:-2] / width[:-1, None]
        falling = ramps[2:] / width[1:, None]
        _cache[key] = torch.clamp(torch.minimum(rising, falling), min=0.0).float().to(device)
    return _cache[key]


def mfsc(y, sfr, window_size=0.025, window_stride=0.010, window='hamming', n_mels=80, preemCoef=0.97, device=None):
    """Log-mel spectrogram [n_mels, frames] of the samples `y` (float, +-1 full scale), as featureExtractor.py:8-23:
    x32768, pre-emphasis, 512-point STFT (25 ms periodic Hamming window, 10 ms hop, no centring), magnitude, 80 Slaney
    mel bands up to sfr/2, log(max(1, .)).  Computed on `device` (default: the GPU when there is one).  Like the
    reference it returns a numpy array for numpy samples; tensor samples give a tensor that stays on the device."""
    out = _mfsc(y, sfr, window_size, window_stride, window, n_mels, preemCoef, device)
    return out if torch.is_tensor(y) else out.cpu().numpy()


def _mfsc(y, sfr, window_size, window_stride, window, n_mels, preemCoef, device):
    if window != 'hamming':
        raise ValueError("only the reference's 'hamming' window is implemented")
    if device is None:
        device = torch.device("cuda") if torch.cuda.is_available() else torch.device("cpu")
    y = torch.as_tensor(np.asarray(y), dtype=torch.float64).to(device=device, dtype=torch.float32) if not torch.is_tensor(y) \
        else y.to(device=device, dtype=torch.float32)
    win_length, hop_length, n_fft = int(sfr * window_size), int(sfr * window_stride), 512
    if y.numel() < n_fft:
        raise ValueError("need at least %d samples, got %d" % (n_fft, y.numel()))
    y = y * 32768.0
    z = torch.empty_like(y)
    z[1:] = y[1:] - preemCoef * y[:-1]
    z[0] = y[0] * (1.0 - preemCoef)
    win = torch.hamming_window(win_length, periodic=True, dtype=torch.float32, device=device)
    S = torch.stft(z, n_fft=n_fft, hop_length=hop_length, win_length=win_length, window=win, center=False, return_complex=True)
    mel = mel_filterbank(sfr, n_fft, n_mels, 0.0, sfr / 2.0, device) @ S.abs()
    return torch.log(torch.clamp(mel, min=1.0))


def normalize(features):
    """Mean normalisation over time of [frames, n_mels] features (featureExtractor.py:25-26)."""
    if torch.is_tensor(features):
        return features - features.mean(dim=0, keepdim=True)
    return features - np.mean(features, axis=0)


def features_on_device(samples, sfr, device=None):
    """[frames, 80] mean-normalised log-mel features as a tensor on `device`: what `getEmbedding(x.unsqueeze(0))` takes,
    without the host round trip of the reference's extractFeatures -> prepareInput (getEmbeddingExample.py:7-19)."""
    return normalize(_mfsc(samples, sfr, 0.025, 0.010, 'hamming', 80, 0.97, device).transpose(0, 1))


def extractFeatures(audioPath, device=None):
    """[frames, 80] mean-normalised log-mel features of a sound file as a numpy array (featureExtractor.py:29-33;
    getEmbeddingExample.py:10 wraps the result in torch.FloatTensor).  Reading the file needs the `soundfile` package,
    exactly as in the reference."""
    try:
        import soundfile as sf
    except ImportError as e:   # same dependency as the reference; not shipped in this image
        raise ImportError("extractFeatures needs the 'soundfile' package to read %s (featureExtractor.py:31); "
                          "call mfsc(samples, rate) directly when the samples are already in memory" % audioPath) from e
    y, sfreq = sf.read(audioPath)
    return features_on_device(y, sfreq, device).cpu().numpy()
