"""Mel <-> waveform DSP: the avc_dsp_* entry points / adaptive_voice_conversion_amd.dsp against the
numpy restatement of preprocess/tacotron/utils.py (oracle/dsp_oracle.py).

The oracle itself is UNPINNED against the reference (librosa is not installed; see its header), so the first block pins
it to an independent implementation (scipy.signal) and to closed-form properties; the second block compares the HIP
path with it through the C ABI.  Tolerances: fp32 MFMA DFT of 40..1200 taps against float64 -> 2e-5 of the spectrum's
scale; normalised dB features 1e-4 absolute (they live in (0, 1]) for bins within 80 dB of the peak; waveforms 1e-4 of the peak after a few Griffin-Lim
iterations (the projection is not a contraction: fp32 noise in near-empty bins grows with the iteration count, so the
100-iteration run is compared through its spectral convergence instead).
"""
import numpy as np
import pytest
import torch
from scipy import signal

from adaptive_voice_conversion_amd import dsp as P
from oracle import dsp_oracle as D
from tests.emu_util import KINDS, backend

GPU = pytest.mark.gpu


def speechlike(n, sr, seed=0):
    """A few drifting harmonics under an envelope with pauses, plus a little noise."""
    r = np.random.RandomState(seed)
    t = np.arange(n) / sr
    f0 = 120 + 30 * np.sin(2 * np.pi * 0.7 * t)
    ph = 2 * np.pi * np.cumsum(f0) / sr
    y = sum(np.sin(k * ph) / k for k in range(1, 12))
    env = np.clip(np.sin(2 * np.pi * 1.3 * t), 0, None) ** 2
    return (0.3 * env * y + 0.002 * r.randn(n)).astype(np.float32)


def hp_for(kind):
    return D.Hyperparams if kind == "gpu" else D.small_hyperparams()


def product_hp(hp):
    return type("HP", (P.Hyperparams,), {k: getattr(hp, k) for k in ("sr", "n_fft", "hop_length", "win_length", "n_mels", "n_iter", "top_db")})


# ---------------------------------------------------------------- the oracle against independent implementations
def test_oracle_stft_equals_scipy_stft_and_inverts():
    hp = D.Hyperparams
    y = speechlike(12000, hp.sr).astype(np.float64)
    S = D.stft(y, hp.n_fft, hp.hop_length, hp.win_length)
    assert S.shape == (1 + hp.n_fft // 2, 1 + len(y) // hp.hop_length)
    w = D.pad_center(D.hann_periodic(hp.win_length), hp.n_fft)
    np.testing.assert_allclose(D.hann_periodic(hp.win_length), signal.get_window("hann", hp.win_length, fftbins=True), atol=1e-15)
    _, _, Z = signal.stft(np.pad(y, hp.n_fft // 2, mode="reflect"), window=w, nperseg=hp.n_fft, noverlap=hp.n_fft - hp.hop_length,
                          boundary=None, padded=False)
    np.testing.assert_allclose(S[:, :Z.shape[1]], Z * w.sum(), atol=1e-10)
    back = D.istft(S, hp.hop_length, hp.win_length)
    assert len(back) == hp.hop_length * (S.shape[1] - 1)
    np.testing.assert_allclose(back, y[:len(back)], atol=1e-12)       # hop = win/4 Hann: exact reconstruction


def test_oracle_tone_mel_filters_and_deemphasis():
    hp = D.Hyperparams
    k = 100
    y = np.cos(2 * np.pi * k * np.arange(9000) / hp.n_fft)
    mag = np.abs(D.stft(y, hp.n_fft, hp.hop_length, hp.win_length))[:, 5:20]
    assert (mag.argmax(axis=0) == k).all()
    np.testing.assert_allclose(mag[k], D.hann_periodic(hp.win_length).sum() / 2, rtol=1e-6)
    m = D.mel_filter(hp.sr, hp.n_fft, 40)
    hz = np.linspace(0, hp.sr / 2, 1 + hp.n_fft // 2)
    area = np.trapezoid(m, hz, axis=1)
    np.testing.assert_allclose(area, 1.0, rtol=2e-2)                   # Slaney normalisation: unit-area triangles
    assert (np.diff(m.argmax(axis=1)) > 0).all()
    x = np.random.RandomState(1).randn(1000)
    pre = np.append(x[0], x[1:] - hp.preemphasis * x[:-1])
    np.testing.assert_allclose(signal.lfilter([1], [1, -hp.preemphasis], pre), x, atol=1e-9)   # :60 and :104 are inverses


def test_product_mel_matrices_equal_the_oracles():
    for sr, n_fft, n_mels in ((24000, 2048, 512), (8000, 64, 8), (16000, 512, 80)):
        np.testing.assert_allclose(P.mel_filter_bank(sr, n_fft, n_mels), D.mel_filter(sr, n_fft, n_mels), atol=1e-12)
        np.testing.assert_allclose(P.mel_to_linear_matrix(sr, n_fft, n_mels), D.mel_to_linear_matrix(sr, n_fft, n_mels), atol=1e-9)


# ---------------------------------------------------------------- the HIP path against the oracle
def make(kind):
    lib, dev = backend(kind)
    hp = hp_for(kind)
    return P.MelDSP(product_hp(hp), device=dev, lib=lib), hp, dev


def to_complex(spec):
    s = spec.double().cpu().numpy()
    return s[0::2] + 1j * s[1::2]


@pytest.mark.parametrize("kind", KINDS)
def test_stft_istft_match_oracle(kind):
    dsp, hp, dev = make(kind)
    n = 48000 if kind == "gpu" else 437      # (odd length: the last partial hop is dropped like librosa does)
    y = speechlike(n, hp.sr)
    S = to_complex(dsp.stft(y))
    ref = D.stft(y, hp.n_fft, hp.hop_length, hp.win_length)
    assert S.shape == ref.shape
    scale = np.abs(ref).max()
    assert np.abs(S - ref).max() <= 2e-5 * scale, np.abs(S - ref).max() / scale
    # inverse of the ORACLE's spectrum (independent of the forward kernel), then the round trip
    spec = torch.from_numpy(np.stack([ref.real, ref.imag], axis=1).reshape(2 * ref.shape[0], -1).astype(np.float32))
    back = dsp.istft(spec).cpu().numpy()
    want = D.istft(ref, hp.hop_length, hp.win_length)
    assert back.shape == want.shape
    np.testing.assert_allclose(back, want, atol=2e-5 * np.abs(want).max())
    rt = dsp.istft(dsp.stft(y)).cpu().numpy()
    np.testing.assert_allclose(rt, y[:len(rt)], atol=3e-5 * np.abs(y).max())


@pytest.mark.parametrize("kind", KINDS)
def test_stft_rejects_signals_shorter_than_the_padding(kind):
    dsp, hp, dev = make(kind)
    with pytest.raises(ValueError, match="reflect padding"):
        dsp.stft(np.zeros(hp.n_fft // 2, dtype=np.float32))


@pytest.mark.parametrize("kind", KINDS)
def test_get_spectrograms_matches_oracle(kind):
    dsp, hp, dev = make(kind)
    n = 60000 if kind == "gpu" else 6000
    y = np.concatenate([np.zeros(3000 if kind == "gpu" else 2100, np.float32), speechlike(n, hp.sr, 3), np.zeros(2500, np.float32)])
    mel, mag = dsp.get_spectrograms(y)
    rmel, rmag = D.get_spectrograms(y, hp)
    assert mel.shape == rmel.shape and mag.shape == rmag.shape and mel.dtype == np.float32
    _, (s0, s1) = D.trim(y, top_db=hp.top_db)
    assert 0 < s0 < s1 < len(y)                                        # the trim really cut both ends
    # normalised dB: d(feature) = 20 / (ln 10 * max_db) * d|X| / |X| -- the fp32 DFT's absolute error (2e-5 of the spectrum's
    # peak, previous test) is a large RELATIVE error in a near-empty bin, so the 1e-4 bar applies to bins within 80 dB of the
    # peak and a loose one to the rest
    for got, want in ((mag, rmag), (mel, rmel), (dsp.get_spectrograms(y, do_trim=False)[0], D.get_spectrograms(y, hp, do_trim=False)[0])):
        loud = want > want.max() - 0.8
        assert loud.mean() > 0.2
        np.testing.assert_allclose(got[loud], want[loud], atol=1e-4)
        np.testing.assert_allclose(got, want, atol=2e-2)


@pytest.mark.parametrize("kind", KINDS)
def test_deemphasis_and_trim_match_scipy_and_oracle(kind):
    dsp, hp, dev = make(kind)
    lib = dsp.lib
    n = 100001 if kind == "gpu" else 5003
    x = torch.from_numpy(speechlike(n, hp.sr, 5))
    xd = x.to(dev)
    out = torch.empty_like(xd)
    assert lib.avc_dsp_deemphasis(P._P(xd), n, 0.97, P._P(out), None) == 0
    ref = signal.lfilter([1], [1, -0.97], x.double().numpy())
    np.testing.assert_allclose(out.cpu().numpy(), ref, atol=2e-5 * np.abs(ref).max())
    y = np.concatenate([np.zeros(4000, np.float32), x.numpy(), np.zeros(3000, np.float32)])
    got, idx = dsp.trim(y, top_db=20)
    want, widx = D.trim(y, top_db=20)
    assert idx == widx and got.numel() == len(want)


@pytest.mark.parametrize("kind", KINDS)
def test_griffin_lim_and_melspectrogram2wav_match_oracle(kind):
    dsp, hp, dev = make(kind)
    n = 36000 if kind == "gpu" else 3000
    y = speechlike(n, hp.sr, 7)
    mel, mag = D.get_spectrograms(y, hp, do_trim=False)
    amp = np.power(10.0, ((np.clip(mag.T.astype(np.float64), 0, 1) * hp.max_db) - hp.max_db + hp.ref_db) * 0.05)
    for it in (0, 3):
        got = dsp.griffin_lim(amp.astype(np.float32), n_iter=it).cpu().numpy()
        want = D.griffin_lim(amp, hp, n_iter=it)
        assert got.shape == want.shape
        np.testing.assert_allclose(got, want, atol=(2e-5 if it == 0 else 2e-4) * np.abs(want).max())
    wav = dsp.melspectrogram2wav(mel, do_trim=False, n_iter=2)
    ref = D.melspectrogram2wav(mel, hp, do_trim=False, n_iter=2)
    assert wav.dtype == np.float32 and wav.shape == ref.shape
    np.testing.assert_allclose(wav, ref, atol=5e-4 * np.abs(ref).max())
    wav2 = dsp.spectrogram2wav(mag, do_trim=False, n_iter=2)
    np.testing.assert_allclose(wav2, D.spectrogram2wav(mag, hp, do_trim=False, n_iter=2), atol=5e-4 * np.abs(ref).max())


@pytest.mark.parametrize("kind", KINDS)
def test_batched_griffin_lim_equals_one_utterance_at_a_time(kind):
    """avc_dsp_griffin_lim_batch: the frames of B utterances are columns of one GEMM per transform; every column's dot
    products are the ones of the single-utterance call, so the waveforms must agree to the last bit."""
    dsp, hp, dev = make(kind)
    n = 12000 if kind == "gpu" else 1500
    mels = [D.get_spectrograms(speechlike(n, hp.sr, 20 + i), hp, do_trim=False)[0] for i in range(3)]
    one = [dsp.melspectrogram2wav(m, do_trim=False, n_iter=2) for m in mels]
    many = dsp.melspectrogram2wav_batch(mels, do_trim=False, n_iter=2)
    assert len(many) == 3
    for a, b in zip(one, many):
        assert a.shape == b.shape and np.array_equal(a, b)
    # ... and utterances of DIFFERENT lengths (avc_dsp_griffin_lim_ragged): same columns, same dot products
    ragged = [mels[0], mels[1][:-7], mels[2][:-13]]
    one_r = [dsp.melspectrogram2wav(m, do_trim=False, n_iter=2) for m in ragged]
    many_r = dsp.melspectrogram2wav_batch(ragged, do_trim=False, n_iter=2)
    for a, b in zip(one_r, many_r):
        assert a.shape == b.shape and np.array_equal(a, b)
    with pytest.raises(ValueError, match="reflect padding"):
        dsp.melspectrogram2wav_batch([mels[0], mels[1][:2]])


@GPU
def test_gpu_hundred_griffin_lim_iterations_converge_like_the_oracle():
    """utils.py's n_iter = 100: compare what the iteration is FOR -- the spectral convergence
    || |stft(x)| - S || / ||S|| of the result -- between the device run and the float64 oracle."""
    dsp, hp, dev = make("gpu")
    y = speechlike(24000, hp.sr, 9)
    S = np.abs(D.stft(y.astype(np.float64), hp.n_fft, hp.hop_length, hp.win_length))

    def sc(x):
        X = np.abs(D.stft(np.asarray(x, dtype=np.float64), hp.n_fft, hp.hop_length, hp.win_length))
        return np.linalg.norm(X - S) / np.linalg.norm(S)
    c0 = sc(dsp.griffin_lim(S.astype(np.float32), n_iter=0).cpu().numpy())
    c100 = sc(dsp.griffin_lim(S.astype(np.float32), n_iter=100).cpu().numpy())
    o100 = sc(D.griffin_lim(S, hp, n_iter=100))
    print(f"[gpu] Griffin-Lim spectral convergence: 0 iterations {c0:.4f}, 100 iterations {c100:.4f} (float64 oracle {o100:.4f})")
    assert c100 < 0.5 * c0
    assert abs(c100 - o100) < 0.1 * o100 + 1e-3


@pytest.mark.parametrize("kind", KINDS)
def test_inference_from_path_wav_to_wav(kind, tmp_path):
    """inference.py:86-93 end to end: two wav files in, the converted wav out -- features, model and Griffin-Lim on the
    device -- against the same chain of the two oracles (dsp_oracle -> avc_oracle.ae_inference -> dsp_oracle)."""
    import pickle
    import types
    from scipy.io import wavfile
    from adaptive_voice_conversion_amd.inference import Inferencer
    from oracle import avc_oracle as O
    lib, dev = backend(kind)
    hp = D.small_hyperparams(n_fft=64, hop=10, win=40, n_mels=16, sr=8000, n_iter=3)
    cfg = O.tiny_config()
    sd = O.make_state_dict(cfg, 7)
    torch.save(sd, tmp_path / "m.ckpt")
    attr = {"mean": np.linspace(0.2, 0.6, 16).astype(np.float32), "std": np.linspace(0.1, 0.3, 16).astype(np.float32)}
    with open(tmp_path / "attr.pkl", "wb") as f:
        pickle.dump(attr, f)
    src, tgt = speechlike(2600, hp.sr, 11), speechlike(1900, hp.sr, 12)
    for name, y in (("src.wav", src), ("tgt.wav", tgt)):
        wavfile.write(tmp_path / name, hp.sr, y)          # float32 wav: decoded without quantisation
    args = types.SimpleNamespace(model=str(tmp_path / "m.ckpt"), attr=str(tmp_path / "attr.pkl"), source=str(tmp_path / "src.wav"),
                                 target=str(tmp_path / "tgt.wav"), output=str(tmp_path / "out.wav"), sample_rate=hp.sr)
    inf = Inferencer(cfg, args, lib=lib if kind == "emu" else None, dsp_hp=product_hp(hp))
    wav, mel = inf.inference_from_path()
    rate, written = wavfile.read(tmp_path / "out.wav")
    assert rate == hp.sr and written.dtype == np.float32 and np.array_equal(written, wav)
    # the oracles' chain
    smel = (D.get_spectrograms(src, hp)[0] - attr["mean"]) / attr["std"]
    tmel = (D.get_spectrograms(tgt, hp)[0] - attr["mean"]) / attr["std"]
    pad = lambda m: m            # frame_size 1: utt_make_frames is a transpose
    dec = O.ae_inference(torch.from_numpy(pad(smel)).t()[None].float(), torch.from_numpy(pad(tmel)).t()[None].float(), sd, cfg)[0].t().numpy()
    dec = dec * attr["std"] + attr["mean"]
    np.testing.assert_allclose(mel, dec, rtol=1e-3, atol=2e-4)
    # the batched front: three pairs, two of them equally long -> one batched Griffin-Lim for those two
    pairs = [(torch.from_numpy(smel).float(), torch.from_numpy(tmel).float())] * 2 + [(torch.from_numpy(smel[:-8]).float(), torch.from_numpy(tmel).float())]
    wavs, mels = inf.convert_batch_to_wav(pairs)
    assert len(wavs) == 3 and np.array_equal(wavs[0], wavs[1]) and wavs[2].shape != wavs[0].shape
    np.testing.assert_allclose(mels[0], mel, rtol=1e-4, atol=1e-4)
    # (the batch goes through the ragged plan, whose InstanceNorm rows reduce in another order than the uniform plan's: the mels agree
    # to 1e-6, three Griffin-Lim iterations later the waveforms to ~1e-5 of the peak)
    assert wavs[0].shape == wav.shape and np.allclose(wavs[0], wav, atol=1e-4 * np.abs(wav).max())
    ref = D.melspectrogram2wav(dec, hp)
    assert abs(len(wav) - len(ref)) <= 512                # (trim picks whole 512-sample hops: a borderline frame may flip)
    n = min(len(wav), len(ref))
    if len(wav) == len(ref):
        assert np.linalg.norm(wav[:n] - ref[:n]) <= 2e-2 * np.linalg.norm(ref[:n])


def test_load_wav_decodes_pcm_stereo_and_resamples(tmp_path):
    """dsp.load_wav (the file-I/O edge of get_spectrograms: the reference calls librosa.load(path, sr=hp.sr)): int16 PCM is
    scaled to [-1, 1), channels are averaged, a different rate is resampled to the requested one."""
    from scipy.io import wavfile
    sr = 8000
    t = np.arange(4000) / sr
    tone = 0.5 * np.sin(2 * np.pi * 440 * t)
    wavfile.write(tmp_path / "mono16.wav", sr, (tone * 32767).astype(np.int16))
    y = P.load_wav(str(tmp_path / "mono16.wav"), sr)
    assert y.dtype == np.float32 and y.shape == (4000,)
    np.testing.assert_allclose(y, tone, atol=2.0 / 32768)
    wavfile.write(tmp_path / "stereo.wav", sr, np.stack([tone, -tone + 0.2], axis=1).astype(np.float32))
    np.testing.assert_allclose(P.load_wav(str(tmp_path / "stereo.wav"), sr), np.full(4000, 0.1), atol=1e-6)
    wavfile.write(tmp_path / "slow.wav", sr // 2, tone[::2].astype(np.float32))
    up = P.load_wav(str(tmp_path / "slow.wav"), sr)
    assert abs(len(up) - 4000) <= 2
    k = np.argmax(np.abs(np.fft.rfft(up[:4000 - 2])))
    assert abs(k * sr / (4000 - 2) - 440) < 3.0                      # still a 440 Hz tone at the new rate
