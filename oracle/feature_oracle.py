"""TEST INFRASTRUCTURE ONLY — float64 numpy restatement of the reference's log-mel front-end.

Follows scripts/featureExtractor.py:8-33 of the reference (`mfsc`, `normalize`, `extractFeatures`).  The arithmetic that
file delegates to lives in a third-party dependency that is ABSENT from this image and from /root/reference:
librosa 0.7.2 (`librosa.stft`, `librosa.feature.melspectrogram`, `librosa.filters.mel`).  Its published algorithm is
restated here:
  * stft(center=False): frames of n_fft samples every hop_length, multiplied by the periodic ('fftbins') Hamming window of
    win_length samples zero-padded symmetrically to n_fft, rfft of each frame (1 + n_fft/2 bins);
  * filters.mel(htk=False, norm=None): Slaney mel scale (linear below 1 kHz at 200/3 Hz per mel, logarithmic above with
    step ln(6.4)/27), n_mels + 2 band edges equally spaced in mel, triangular weights max(0, min(rising, falling)) on the
    rfft bin frequencies, no area normalisation;
  * melspectrogram(S=|stft|): mel_basis @ S (the magnitude spectrogram is used as given).
PARITY STATUS: UNPINNED — neither librosa nor any fixture of the reference pins these numbers; the tests compare the
product's torch implementation against this restatement only.
"""
import numpy as np


def hz_to_mel_slaney(f):
    f = np.asarray(f, dtype=np.float64)
    f_sp = 200.0 / 3.0
    mel = f / f_sp
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-300) / min_log_hz) / logstep, mel)


def mel_to_hz_slaney(m):
    m = np.asarray(m, dtype=np.float64)
    f_sp = 200.0 / 3.0
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)


def mel_filterbank(sr, n_fft, n_mels, fmin, fmax):
    """[n_mels, 1 + n_fft//2] triangular filters, peak 1 (norm=None)."""
    fftfreqs = np.linspace(0.0, sr / 2.0, 1 + n_fft // 2)
    mel_f = mel_to_hz_slaney(np.linspace(hz_to_mel_slaney(fmin), hz_to_mel_slaney(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fftfreqs[None, :]
    w = np.zeros((n_mels, fftfreqs.size))
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        w[i] = np.maximum(0.0, np.minimum(lower, upper))
    return w


def hamming_periodic(n):
    return 0.54 - 0.46 * np.cos(2.0 * np.pi * np.arange(n) / n)


def mfsc(y, sfr, window_size=0.025, window_stride=0.010, n_mels=80, preemCoef=0.97):
    """featureExtractor.py:8-23 — returns [n_mels, frames] like the reference (time last)."""
    y = np.array(y, dtype=np.float64) * 32768.0
    win_length, hop = int(sfr * window_size), int(sfr * window_stride)
    n_fft = 512
    z = y.copy()
    z[1:] = y[1:] - preemCoef * y[:-1]          # featureExtractor.py:16 (right-hand side uses the original samples)
    z[0] = y[0] * (1.0 - preemCoef)             # featureExtractor.py:17
    win = np.zeros(n_fft)
    lpad = (n_fft - win_length) // 2
    win[lpad:lpad + win_length] = hamming_periodic(win_length)
    n_frames = 1 + (len(z) - n_fft) // hop
    frames = np.stack([z[i * hop:i * hop + n_fft] * win for i in range(n_frames)], axis=1)   # [n_fft, frames]
    D = np.abs(np.fft.rfft(frames, axis=0))
    mel = mel_filterbank(sfr, n_fft, n_mels, 0.0, sfr / 2.0) @ D
    return np.log(np.maximum(1.0, mel))


def normalize(features):
    """featureExtractor.py:25-26 — cepstral mean normalisation over time ([frames, n_mels] in)."""
    return features - np.mean(features, axis=0)


def extract_features(y, sfr):
    """featureExtractor.py:29-33 without the file read: [frames, n_mels], mean-normalised."""
    return normalize(np.transpose(mfsc(y, sfr)))
