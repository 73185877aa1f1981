"""TEST INFRASTRUCTURE ONLY -- CPU restatement (numpy float64) of the reference's mel <-> waveform DSP
(SURVEY §8f row 4): preprocess/tacotron/utils.py:27-155 with preprocess/tacotron/hyperparams.py:20-34.

PARITY UNPINNED.  The reference implements this path with `librosa` (stft / istft / filters.mel /
effects.trim; no version pinned -- the repository has no requirements file; the code targets the 0.6-0.7 API:
positional `librosa.filters.mel(sr, n_fft, n_mels)`, `librosa.stft(y, n_fft, hop, win_length=...)`), and librosa
is not installed here (no network), so the reference cannot be run on this path and it ships no test or golden
vector for it.  The functions below restate librosa's PUBLISHED algorithms for exactly the call sites the
reference uses; each cites the reference line it stands for.  What they are checked against instead
(tests/test_dsp.py): scipy.signal.stft / istft on the same window and framing (an independent implementation of the
same transform), closed-form properties (a pure tone lands in its bin, istft(stft(y)) == y, the mel filters'
Slaney area normalisation), and scipy.signal.lfilter for the de-emphasis.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import numpy as np
from scipy import signal


class Hyperparams:
    """preprocess/tacotron/hyperparams.py:20-34 (signal processing block)."""
    top_db = 15
    sr = 24000
    n_fft = 2048
    frame_shift = 0.0125
    frame_length = 0.05
    hop_length = int(sr * frame_shift)      # 300
    win_length = int(sr * frame_length)     # 1200
    n_mels = 512
    n_iter = 100
    preemphasis = .97
    max_db = 100
    ref_db = 20


def small_hyperparams(n_fft=64, hop=10, win=40, n_mels=8, sr=8000, n_iter=4):
    hp = type("SmallHyperparams", (Hyperparams,), {})
    hp.n_fft, hp.hop_length, hp.win_length, hp.n_mels, hp.sr, hp.n_iter = n_fft, hop, win, n_mels, sr, n_iter
    return hp


# ---- librosa pieces the reference calls --------------------------------------------------------
def hann_periodic(n):
    """scipy.signal.get_window('hann', n, fftbins=True): what librosa.stft / istft build from window='hann'."""
    return 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n) / n)


def pad_center(w, size):
    lpad = (size - len(w)) // 2
    return np.pad(w, (lpad, size - len(w) - lpad))


def stft(y, n_fft, hop_length, win_length):
    """librosa.stft(y, n_fft, hop_length, win_length) with the defaults of the reference's call sites
    (utils.py:63-66,141): hann window padded to n_fft, center=True with reflect padding, rfft of every frame.
    Returns [1 + n_fft/2, 1 + len(y) // hop] complex."""
    y = np.asarray(y, dtype=np.float64)
    w = pad_center(hann_periodic(win_length), n_fft)
    yp = np.pad(y, n_fft // 2, mode="reflect")
    n_frames = 1 + (len(yp) - n_fft) // hop_length
    idx = np.arange(n_fft)[:, None] + hop_length * np.arange(n_frames)[None, :]
    return np.fft.rfft(w[:, None] * yp[idx], axis=0)


def istft(S, hop_length, win_length):
    """librosa.istft(S, hop_length, win_length=win_length, window='hann') (utils.py:150-154): windowed irfft of every
    frame, overlap-add, division by the window's sum of squares where it is not tiny, n_fft/2 trimmed at both ends."""
    n_fft = 2 * (S.shape[0] - 1)
    T = S.shape[1]
    w = pad_center(hann_periodic(win_length), n_fft)
    n = n_fft + hop_length * (T - 1)
    y = np.zeros(n)
    wss = np.zeros(n)
    frames = w[:, None] * np.fft.irfft(S, n=n_fft, axis=0)
    for t in range(T):
        y[t * hop_length:t * hop_length + n_fft] += frames[:, t]
        wss[t * hop_length:t * hop_length + n_fft] += w * w
    nz = wss > np.finfo(np.float32).tiny
    y[nz] /= wss[nz]
    return y[n_fft // 2:-(n_fft // 2)]


def mel_filter(sr, n_fft, n_mels):
    """librosa.filters.mel(sr, n_fft, n_mels) (utils.py:28,72): fmin 0, fmax sr/2, Slaney mel scale, area-normalised triangles."""
    f_sp = 200.0 / 3
    min_log_hz, min_log_mel, logstep = 1000.0, 1000.0 / f_sp, np.log(6.4) / 27.0

    def hz_to_mel(f):
        f = np.asarray(f, dtype=np.float64)
        return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-30) / min_log_hz) / logstep, f / f_sp)

    def mel_to_hz(m):
        m = np.asarray(m, dtype=np.float64)
        return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)

    fftfreqs = np.linspace(0, sr / 2.0, 1 + n_fft // 2)
    mel_f = mel_to_hz(np.linspace(hz_to_mel(0.0), hz_to_mel(sr / 2.0), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fftfreqs[None, :]
    weights = np.zeros((n_mels, 1 + n_fft // 2))
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        weights[i] = np.maximum(0, np.minimum(lower, upper))
    enorm = 2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels])
    return weights * enorm[:, None]


def trim(y, top_db=60, frame_length=2048, hop_length=512):
    """librosa.effects.trim(y, top_db) (utils.py:57,107,130): frames whose RMS power is within top_db of the loudest."""
    yp = np.pad(np.asarray(y, dtype=np.float64), frame_length // 2, mode="reflect")
    n_frames = 1 + (len(yp) - frame_length) // hop_length
    idx = np.arange(frame_length)[:, None] + hop_length * np.arange(n_frames)[None, :]
    mse = np.mean(yp[idx] ** 2, axis=0)
    db = 10.0 * np.log10(np.maximum(1e-10, mse)) - 10.0 * np.log10(np.maximum(1e-10, mse.max()))
    nz = np.flatnonzero(db > -top_db)
    if nz.size == 0:
        return y[0:0], (0, 0)
    start, end = int(nz[0] * hop_length), min(len(y), int((nz[-1] + 1) * hop_length))
    return y[start:end], (start, end)


# ---- the reference's functions -------------------------------------------------------------------
def mel_to_linear_matrix(sr, n_fft, n_mels):
    """utils.py:27-32 `_mel_to_linear_matrix`."""
    m = mel_filter(sr, n_fft, n_mels)
    m_t = m.T
    p = m @ m_t
    d = np.array([1.0 / x if np.abs(x) > 1.0e-8 else x for x in np.sum(p, axis=0)])
    return m_t @ np.diag(d)


def get_spectrograms(y, hp=Hyperparams, do_trim=True):
    """utils.py:34-87 from the loaded waveform on (librosa.load's decoding / resampling is file I/O)."""
    y = np.asarray(y, dtype=np.float64)
    if do_trim:
        y, _ = trim(y, top_db=hp.top_db)                                  # :57
    y = np.append(y[0], y[1:] - hp.preemphasis * y[:-1])                  # :60
    linear = stft(y, hp.n_fft, hp.hop_length, hp.win_length)              # :63-66
    mag = np.abs(linear)                                                  # :69
    mel = mel_filter(hp.sr, hp.n_fft, hp.n_mels) @ mag                    # :72-73
    mel = 20 * np.log10(np.maximum(1e-5, mel))                            # :76-77
    mag = 20 * np.log10(np.maximum(1e-5, mag))
    mel = np.clip((mel - hp.ref_db + hp.max_db) / hp.max_db, 1e-8, 1)     # :80-81
    mag = np.clip((mag - hp.ref_db + hp.max_db) / hp.max_db, 1e-8, 1)
    return mel.T.astype(np.float32), mag.T.astype(np.float32)             # :84-87


def invert_spectrogram(S, hp=Hyperparams):
    """utils.py:150-154."""
    return istft(S, hp.hop_length, hp.win_length)


def griffin_lim(spectrogram, hp=Hyperparams, n_iter=None):
    """utils.py:136-147."""
    X_best = spectrogram.astype(np.complex128)
    for _ in range(hp.n_iter if n_iter is None else n_iter):
        X_t = invert_spectrogram(X_best, hp)
        est = stft(X_t, hp.n_fft, hp.hop_length, hp.win_length)
        phase = est / np.maximum(1e-8, np.abs(est))
        X_best = spectrogram * phase
    return np.real(invert_spectrogram(X_best, hp))


def _finish(wav, hp, do_trim):
    wav = signal.lfilter([1], [1, -hp.preemphasis], wav)                   # :104,127 de-preemphasis
    if do_trim:
        wav, _ = trim(wav)                                                 # :107,130
    return wav.astype(np.float32)


def melspectrogram2wav(mel, hp=Hyperparams, do_trim=True, n_iter=None):
    """utils.py:89-109.  mel: [T, n_mels] normalised."""
    mel = np.asarray(mel, dtype=np.float64).T
    mel = (np.clip(mel, 0, 1) * hp.max_db) - hp.max_db + hp.ref_db        # :95
    mel = np.power(10.0, mel * 0.05)                                       # :98
    mag = mel_to_linear_matrix(hp.sr, hp.n_fft, hp.n_mels) @ mel          # :99-100
    return _finish(griffin_lim(mag, hp, n_iter), hp, do_trim)


def spectrogram2wav(mag, hp=Hyperparams, do_trim=True, n_iter=None):
    """utils.py:111-132.  mag: [T, 1 + n_fft/2] normalised."""
    mag = np.asarray(mag, dtype=np.float64).T
    mag = (np.clip(mag, 0, 1) * hp.max_db) - hp.max_db + hp.ref_db
    mag = np.power(10.0, mag * 0.05)
    return _finish(griffin_lim(mag, hp, n_iter), hp, do_trim)
