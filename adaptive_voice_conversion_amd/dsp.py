"""Mel <-> waveform DSP on the GPU: the audio front and back end of a conversion.

Host-side mirror of the reference's ``preprocess/tacotron/utils.py`` -- same function names, argument meaning and
hyper-parameters (``preprocess/tacotron/hyperparams.py:20-34``) -- over the ``avc_dsp_*`` entry points of
``libavc_hip.so``: the STFT / inverse STFT are GEMMs against windowed DFT bases on the fp32 MFMA, Griffin-Lim's
100 iterations stay on the device, only the trimmed waveform comes back.  No CPU fallback: without the library this
module raises like the rest of the package.

Differences from the reference, all at the file-I/O edge: ``get_spectrograms`` also accepts an already decoded waveform;
a wav file is decoded with ``scipy.io.wavfile`` and resampled with ``scipy.signal.resample_poly`` when its rate is not
``hp.sr`` (the reference calls ``librosa.load``, whose resampler is a different filter: resample offline for bit-level
agreement with features made by the reference's pipeline).
"""
import ctypes

import numpy as np
import torch

from . import _lib


class Hyperparams:
    """preprocess/tacotron/hyperparams.py:20-34."""
    top_db = 15
    sr = 24000
    n_fft = 2048
    frame_shift = 0.0125
    frame_length = 0.05
    hop_length = int(sr * frame_shift)
    win_length = int(sr * frame_length)
    n_mels = 512
    n_iter = 100
    preemphasis = .97
    max_db = 100
    ref_db = 20


def _P(t):
    return ctypes.c_void_p(0 if t is None else t.data_ptr())


def mel_filter_bank(sr, n_fft, n_mels):
    """The matrix ``librosa.filters.mel(sr, n_fft, n_mels)`` builds (utils.py:28,72): fmin 0, fmax sr/2, Slaney mel
    scale (linear below 1 kHz, log above), triangles normalised to unit area.  [n_mels, 1 + n_fft/2] float64."""
    f_sp, min_log_hz = 200.0 / 3, 1000.0
    min_log_mel, logstep = min_log_hz / f_sp, np.log(6.4) / 27.0
    top = min_log_mel + np.log((sr / 2.0) / min_log_hz) / logstep if sr / 2.0 >= min_log_hz else (sr / 2.0) / f_sp
    mels = np.linspace(0.0, top, n_mels + 2)
    hz = np.where(mels >= min_log_mel, min_log_hz * np.exp(logstep * (mels - min_log_mel)), f_sp * mels)
    bins = np.linspace(0, sr / 2.0, 1 + n_fft // 2)
    up = (bins[None, :] - hz[:-2, None]) / np.diff(hz)[:-1, None]
    down = (hz[2:, None] - bins[None, :]) / np.diff(hz)[1:, None]
    w = np.maximum(0.0, np.minimum(up, down))
    return w * (2.0 / (hz[2:] - hz[:-2]))[:, None]


def mel_to_linear_matrix(sr, n_fft, n_mels):
    """utils.py:27-32 ``_mel_to_linear_matrix``: m^T diag(1 / colsum(m m^T))."""
    m = mel_filter_bank(sr, n_fft, n_mels)
    colsum = (m @ m.T).sum(axis=0)
    d = np.where(np.abs(colsum) > 1.0e-8, 1.0 / np.where(colsum == 0, 1.0, colsum), colsum)
    return m.T * d[None, :]


class MelDSP:
    """One instance per (hyper-parameters, device): holds the DFT / mel bases in device memory."""

    def __init__(self, hp=Hyperparams, device=None, lib=None):
        self.hp = hp
        self.lib = lib if lib is not None else _lib.load()
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.F = hp.n_fft // 2 + 1
        lib_, dev = self.lib, self.device
        scratch = torch.empty(lib_.avc_dsp_basis_scratch_floats(hp.n_fft, hp.win_length), device=dev)
        self.basis_fwd = torch.zeros(lib_.avc_dsp_basis_floats(hp.n_fft, hp.win_length, 0), device=dev)
        self.basis_inv = torch.zeros(lib_.avc_dsp_basis_floats(hp.n_fft, hp.win_length, 1), device=dev)
        with self._dev():
            for inv, dst in ((0, self.basis_fwd), (1, self.basis_inv)):
                self._ok(lib_.avc_dsp_make_basis(hp.n_fft, hp.hop_length, hp.win_length, inv, _P(scratch), _P(dst), self._stream()))
            self._sync()   # (scratch is released below)
        self.mel_w = self._pack(torch.from_numpy(mel_filter_bank(hp.sr, hp.n_fft, hp.n_mels).astype(np.float32)))
        self.mel_inv_w = self._pack(torch.from_numpy(mel_to_linear_matrix(hp.sr, hp.n_fft, hp.n_mels).astype(np.float32)))

    # ---- plumbing
    def _dev(self):
        import contextlib
        return torch.cuda.device(self.device) if self.device.type == "cuda" else contextlib.nullcontext()

    def _stream(self):
        return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream) if self.device.type == "cuda" else None

    def _sync(self):
        if self.device.type == "cuda":
            torch.cuda.current_stream(self.device).synchronize()

    def _ok(self, rc):
        if rc != 0:
            if rc == -6:
                raise ValueError("signal too short for the reflect padding of the STFT (needs more than n_fft/2 samples)")
            raise RuntimeError(f"avc_dsp call failed: {rc}")

    def _pack(self, w):
        """[Cout, Cin] matrix -> the LDS-image order the 1x1 GEMM reads."""
        w = w.to(self.device).contiguous()
        Cout, Cin = w.shape
        dst = torch.zeros(self.lib.avc_packed_weight_floats(Cout, Cin, 1, 0), device=self.device)
        arr = (ctypes.c_void_p * 1)(w.data_ptr())
        with self._dev():
            self._ok(self.lib.avc_pack_weight(arr, 1, Cout, Cout, Cin, 1, 0, _P(dst), self._stream()))
            self._sync()
        return (dst, Cout, Cin)

    def _matmul(self, packed, x):
        """packed [Cout, Cin] x  x [Cin, T] -> [Cout, T] on the fp32 MFMA."""
        wp, Cout, Cin = packed
        T = x.shape[1]
        out = torch.empty(Cout, T, device=self.device)
        with self._dev():
            self._ok(self.lib.avc_conv1d_fwd(_P(x), 0, T, 1, 1, Cin, T, _P(wp), None, Cout, 1, 1, 0, _P(out), 0, T, 1, 1, None, 0, 0, 0,
                                             0, 0, None, 0, self._stream()))
        return out

    def _wave(self, y):
        y = torch.as_tensor(np.asarray(y, dtype=np.float32) if not torch.is_tensor(y) else y, dtype=torch.float32)
        return y.to(self.device).contiguous().view(-1)

    # ---- librosa pieces
    def stft(self, y):
        """librosa.stft(y, n_fft, hop_length, win_length) -> [2F, T] fp32 (Re / Im rows interleaved)."""
        hp, y = self.hp, self._wave(y)
        T = self.lib.avc_dsp_num_frames(y.numel(), hp.hop_length)
        frames = torch.empty(hp.win_length * T, device=self.device)
        spec = torch.empty(2 * self.F, T, device=self.device)
        with self._dev():
            self._ok(self.lib.avc_dsp_stft(_P(y), y.numel(), hp.n_fft, hp.hop_length, hp.win_length, _P(self.basis_fwd), _P(frames),
                                           _P(spec), self._stream()))
        return spec

    def istft(self, spec):
        """librosa.istft(spec, hop_length, win_length=win_length, window="hann") -> hop * (T - 1) samples."""
        hp = self.hp
        spec = spec.to(self.device).contiguous()
        T = spec.shape[1]
        tf = torch.empty(hp.win_length * T, device=self.device)
        y = torch.empty(hp.hop_length * (T - 1), device=self.device)
        with self._dev():
            self._ok(self.lib.avc_dsp_istft(_P(spec), T, hp.n_fft, hp.hop_length, hp.win_length, _P(self.basis_inv), _P(tf), _P(y),
                                            self._stream()))
        return y

    def trim(self, y, top_db=60, frame_length=2048, hop_length=512):
        """librosa.effects.trim(y, top_db): the frame powers come from the device, the two indices are picked here."""
        y = self._wave(y)
        nf = self.lib.avc_dsp_num_frames(y.numel(), hop_length)
        mse = torch.empty(nf, device=self.device)
        with self._dev():
            self._ok(self.lib.avc_dsp_frame_power(_P(y), y.numel(), frame_length, hop_length, _P(mse), self._stream()))
        mse = mse.double().cpu().numpy()
        db = 10.0 * np.log10(np.maximum(1e-10, mse)) - 10.0 * np.log10(np.maximum(1e-10, mse.max()))
        nz = np.flatnonzero(db > -top_db)
        if nz.size == 0:
            return y[0:0], (0, 0)
        start, end = int(nz[0] * hop_length), min(y.numel(), int((nz[-1] + 1) * hop_length))
        return y[start:end], (start, end)

    # ---- the reference's functions
    def get_spectrograms(self, wav, do_trim=True):
        """utils.py:34-87.  `wav`: path of a wav file, or a decoded mono waveform at hp.sr.
        Returns (mel [T, n_mels], mag [T, 1 + n_fft/2]) float32 numpy, normalised to (0, 1]."""
        mel, mag = self.get_spectrograms_device(wav, do_trim)
        return mel.cpu().numpy(), mag.cpu().numpy()

    def get_spectrograms_device(self, wav, do_trim=True):
        hp = self.hp
        y = self._wave(load_wav(wav, hp.sr) if isinstance(wav, str) else wav)
        if do_trim:
            y, _ = self.trim(y, top_db=hp.top_db)                                          # :57
            y = y.contiguous()
        pre = torch.empty_like(y)
        with self._dev():
            self._ok(self.lib.avc_dsp_preemphasis(_P(y), y.numel(), hp.preemphasis, _P(pre), self._stream()))   # :60
        spec = self.stft(pre)                                                              # :63-66
        T = spec.shape[1]
        mag = torch.empty(self.F, T, device=self.device)
        with self._dev():
            self._ok(self.lib.avc_dsp_magnitude(_P(spec), hp.n_fft, T, _P(mag), self._stream()))                # :69
        mel = self._matmul(self.mel_w, mag)                                                # :72-73
        mel_n = torch.empty(T, hp.n_mels, device=self.device)
        mag_n = torch.empty(T, self.F, device=self.device)
        with self._dev():                                                                  # :76-85
            self._ok(self.lib.avc_dsp_db_normalize(_P(mel), hp.n_mels, T, hp.ref_db, hp.max_db, _P(mel_n), self._stream()))
            self._ok(self.lib.avc_dsp_db_normalize(_P(mag), self.F, T, hp.ref_db, hp.max_db, _P(mag_n), self._stream()))
        return mel_n, mag_n

    def griffin_lim(self, spectrogram, n_iter=None):
        """utils.py:136-147.  spectrogram: [F, T] magnitudes (device tensor or array).  Returns the device waveform."""
        hp = self.hp
        S = torch.as_tensor(spectrogram, dtype=torch.float32).to(self.device).contiguous()
        T = S.shape[1]
        ws = torch.empty(self.lib.avc_dsp_griffin_lim_ws_floats(T, hp.n_fft, hp.hop_length, hp.win_length), device=self.device)
        y = torch.empty(hp.hop_length * (T - 1), device=self.device)
        with self._dev():
            self._ok(self.lib.avc_dsp_griffin_lim(_P(S), T, hp.n_fft, hp.hop_length, hp.win_length, hp.n_iter if n_iter is None else n_iter,
                                                  _P(self.basis_fwd), _P(self.basis_inv), _P(ws), _P(y), self._stream()))
        return y

    def griffin_lim_batch(self, spectrograms, n_iter=None):
        """B equally long utterances in ONE set of launches: spectrograms [B, F, T] (or a list of [F, T]).  Returns [B, hop (T-1)]
        device waveforms.  A lone 400-frame utterance fills half the chip; eight of them are one efficient GEMM per transform."""
        hp = self.hp
        S = torch.stack([torch.as_tensor(x, dtype=torch.float32) for x in spectrograms]) if not torch.is_tensor(spectrograms) else spectrograms
        S = S.to(self.device).float()
        B, F, T = S.shape
        Scat = S.permute(1, 0, 2).reshape(F, B * T).contiguous()          # [F][B T]: utterance b in columns b T ..
        return self._griffin_lim_cat(Scat, B, T, n_iter)

    def _griffin_lim_cat(self, Scat, B, T, n_iter):
        hp = self.hp
        ws = torch.empty(self.lib.avc_dsp_griffin_lim_ws_floats(B * T, hp.n_fft, hp.hop_length, hp.win_length), device=self.device)
        y = torch.empty(B, hp.hop_length * (T - 1), device=self.device)
        with self._dev():
            self._ok(self.lib.avc_dsp_griffin_lim_batch(_P(Scat), B, T, hp.n_fft, hp.hop_length, hp.win_length,
                                                        hp.n_iter if n_iter is None else n_iter, _P(self.basis_fwd), _P(self.basis_inv),
                                                        _P(ws), _P(y), self._stream()))
        return y

    def melspectrogram2wav_batch(self, mels, do_trim=True, n_iter=None):
        """utils.py:89-109 for B utterances in ONE launch set ([B, T, n_mels] or a list of [T_b, n_mels] of any lengths: their
        frames are the columns of one GEMM per transform; unequal lengths go through avc_dsp_griffin_lim_ragged).
        Returns a list of float32 numpy waveforms."""
        hp = self.hp
        mels = [torch.as_tensor(m, dtype=torch.float32) for m in mels]
        B = len(mels)
        Ts = [int(m.shape[0]) for m in mels]
        if any(hp.hop_length * (T - 1) <= hp.n_fft // 2 for T in Ts):
            raise ValueError("signal too short for the reflect padding of the STFT (needs more than n_fft/2 samples)")
        amp = self._amplitudes(torch.cat(mels, dim=0))                       # [C][sum T]
        mag = self._matmul(self.mel_inv_w, amp)
        if all(T == Ts[0] for T in Ts):
            y = self._griffin_lim_cat(mag, B, Ts[0], n_iter)
            return [self._finish(y[b], do_trim) for b in range(B)]
        toff_host = np.ascontiguousarray(np.concatenate([[0], np.cumsum(Ts)]), dtype=np.int32)   # validated by the library
        toff = torch.from_numpy(toff_host).to(self.device)
        Ttot = int(sum(Ts))
        ws = torch.empty(self.lib.avc_dsp_griffin_lim_ws_floats(Ttot, hp.n_fft, hp.hop_length, hp.win_length), device=self.device)
        y = torch.empty(hp.hop_length * (Ttot - B), device=self.device)
        with self._dev():
            self._ok(self.lib.avc_dsp_griffin_lim_ragged(_P(mag), _P(toff), ctypes.c_void_p(toff_host.ctypes.data), B, Ttot, hp.n_fft, hp.hop_length, hp.win_length,
                                                         hp.n_iter if n_iter is None else n_iter, _P(self.basis_fwd), _P(self.basis_inv),
                                                         _P(ws), _P(y), self._stream()))
        starts = [hp.hop_length * (sum(Ts[:b]) - b) for b in range(B)]
        return [self._finish(y[starts[b]:starts[b] + hp.hop_length * (Ts[b] - 1)].contiguous(), do_trim) for b in range(B)]

    def _finish(self, wav, do_trim):
        hp = self.hp
        out = torch.empty_like(wav)
        with self._dev():
            self._ok(self.lib.avc_dsp_deemphasis(_P(wav), wav.numel(), hp.preemphasis, _P(out), self._stream()))   # :104,127
        if do_trim:
            out, _ = self.trim(out)                                                                              # :107,130
        return out.cpu().numpy().astype(np.float32)

    def _amplitudes(self, feat):
        hp = self.hp
        feat = torch.as_tensor(feat, dtype=torch.float32).to(self.device).contiguous()     # [T, C] normalised
        T, C = feat.shape
        amp = torch.empty(C, T, device=self.device)
        with self._dev():                                                                  # :92-98 / :114-120
            self._ok(self.lib.avc_dsp_denormalize_amp(_P(feat), C, T, hp.ref_db, hp.max_db, _P(amp), self._stream()))
        return amp

    def melspectrogram2wav(self, mel, do_trim=True, n_iter=None):
        """utils.py:89-109.  mel: [T, n_mels] normalised (what AE.inference returns after denormalisation)."""
        mag = self._matmul(self.mel_inv_w, self._amplitudes(mel))                          # :99-100
        return self._finish(self.griffin_lim(mag, n_iter), do_trim)

    def spectrogram2wav(self, mag, do_trim=True, n_iter=None):
        """utils.py:111-132.  mag: [T, 1 + n_fft/2] normalised."""
        return self._finish(self.griffin_lim(self._amplitudes(mag), n_iter), do_trim)


def load_wav(path, sr):
    """Decode a wav file to mono float32 in [-1, 1] at `sr` (the reference: librosa.load(path, sr=hp.sr))."""
    from scipy.io import wavfile
    rate, data = wavfile.read(path)
    if data.dtype.kind == "i":
        data = data.astype(np.float32) / float(np.iinfo(data.dtype).max + 1)
    elif data.dtype.kind == "u":
        data = (data.astype(np.float32) - 128.0) / 128.0
    data = data.astype(np.float32)
    if data.ndim > 1:
        data = data.mean(axis=1)
    if rate != sr:
        from math import gcd
        from scipy.signal import resample_poly
        g = gcd(int(sr), int(rate))
        data = resample_poly(data, int(sr) // g, int(rate) // g).astype(np.float32)
    return data
