// C-ABI entry points of the mel <-> waveform DSP (declared in include/avc_hip.h; kernels in dsp.hip, the two transforms
// run as 1x1 "convolutions" on conv_gemm.hip's fp32-MFMA kernel).  Reference: preprocess/tacotron/utils.py:27-155.
#include <hip/hip_runtime.h>

#include "avc_common.h"
#include "avc_hip.h"
#include "avc_internal.h"

extern "C" {

int avc_dsp_num_frames(long L, int hop_length) { return (int)(1 + L / hop_length); }   // librosa.stft, center=True

static bool dsp_geom_ok(int n_fft, int hop, int win) { return n_fft >= 4 && (n_fft & 1) == 0 && win >= 2 && win <= n_fft && hop >= 1 && hop <= win; }

long avc_dsp_basis_floats(int n_fft, int win_length, int inverse) {
    const int F2 = n_fft + 2;
    return inverse ? avc_packed_weight_floats(win_length, F2, 1, 0) : avc_packed_weight_floats(F2, win_length, 1, 0);
}
long avc_dsp_basis_scratch_floats(int n_fft, int win_length) { return (long)(n_fft + 2) * win_length; }

int avc_dsp_make_basis(int n_fft, int hop_length, int win_length, int inverse, float* dense_scratch, float* packed, void* stream) {
    if (!dsp_geom_ok(n_fft, hop_length, win_length) || !dense_scratch || !packed) return -1;
    int rc = avc_launch_dsp_basis(inverse ? 1 : 0, n_fft, win_length, dense_scratch, (hipStream_t)stream);
    if (rc) return rc;
    const float* src = dense_scratch;
    const int F2 = n_fft + 2;
    return inverse ? avc_pack_weight(&src, 1, win_length, win_length, F2, 1, 0, packed, stream)
                   : avc_pack_weight(&src, 1, F2, F2, win_length, 1, 0, packed, stream);
}

// spec[2F][B T] = Wf[2F][win] x frames[win][B T]   (B signals of L samples: their frames are columns of one GEMM)
int avc_dsp_stft_batch(const float* y, long L, int B, int n_fft, int hop_length, int win_length, const float* basis_fwd, float* frames_ws,
                       float* spec, void* stream) {
    if (!dsp_geom_ok(n_fft, hop_length, win_length) || !y || !basis_fwd || !frames_ws || !spec || B < 1) return -1;
    if (L <= n_fft / 2) return -6;   // numpy reflect padding needs pad < len (the reference's librosa call raises there)
    const int T = avc_dsp_num_frames(L, hop_length), F2 = n_fft + 2;
    const long BT = (long)B * T;
    if (BT > 0x7fffffffL) return -1;
    int rc = avc_launch_dsp_frames(y, L, B, T, hop_length, n_fft, win_length, frames_ws, (hipStream_t)stream);
    if (rc) return rc;
    return avc_conv1d_fwd(frames_ws, 0, BT, 1, 1, win_length, (int)BT, basis_fwd, nullptr, F2, 1, 1, 0, spec, 0, BT, 1, 1, nullptr, 0, 0, 0,
                          0, 0, nullptr, 0, stream);
}
int avc_dsp_stft(const float* y, long L, int n_fft, int hop_length, int win_length, const float* basis_fwd, float* frames_ws,
                 float* spec, void* stream) {
    return avc_dsp_stft_batch(y, L, 1, n_fft, hop_length, win_length, basis_fwd, frames_ws, spec, stream);
}

// y[hop (T-1)] = overlap-add of tf[win][T] = Wi[win][2F] x spec[2F][T], over the window's sum of squares
int avc_dsp_istft_batch(const float* spec, int B, int T, int n_fft, int hop_length, int win_length, const float* basis_inv, float* tf_ws,
                        float* y, void* stream) {
    if (!dsp_geom_ok(n_fft, hop_length, win_length) || !spec || !basis_inv || !tf_ws || !y || T < 2 || B < 1) return -1;
    const int F2 = n_fft + 2;
    const long BT = (long)B * T;
    if (BT > 0x7fffffffL) return -1;
    int rc = avc_conv1d_fwd(spec, 0, BT, 1, 1, F2, (int)BT, basis_inv, nullptr, win_length, 1, 1, 0, tf_ws, 0, BT, 1, 1, nullptr, 0, 0, 0, 0,
                            0, nullptr, 0, stream);
    if (rc) return rc;
    return avc_launch_dsp_ola(tf_ws, B, T, hop_length, n_fft, win_length, y, (hipStream_t)stream);
}
int avc_dsp_istft(const float* spec, int T, int n_fft, int hop_length, int win_length, const float* basis_inv, float* tf_ws, float* y,
                  void* stream) {
    return avc_dsp_istft_batch(spec, 1, T, n_fft, hop_length, win_length, basis_inv, tf_ws, y, stream);
}

long avc_dsp_griffin_lim_ws_floats(int T, int n_fft, int hop_length, int win_length) {   // T = frames of ALL utterances of a batch
    const long F2 = n_fft + 2;
    auto up = [](long n) { return (n + 63) / 64 * 64; };
    return 2 * up(F2 * T) + up((long)win_length * T) + up((long)hop_length * T);
}

// utils.py:136-147: X = S; repeat n_iter: x = istft(X); est = stft(x); X = S * est / max(1e-8, |est|); return istft(X)
// B utterances of T frames each: S is [F][B T] (utterance b in columns b T .. b T + T - 1), y is [B][hop (T - 1)]
int avc_dsp_griffin_lim_batch(const float* S, int B, int T, int n_fft, int hop_length, int win_length, int n_iter, const float* basis_fwd,
                              const float* basis_inv, float* ws, float* y, void* stream) {
    if (!dsp_geom_ok(n_fft, hop_length, win_length) || !S || !basis_fwd || !basis_inv || !ws || !y || T < 2 || B < 1 || n_iter < 0) return -1;
    const long Ly = (long)hop_length * (T - 1), BT = (long)B * T;
    if (Ly <= n_fft / 2) return -6;
    if (BT > 0x7fffffffL) return -1;
    const int F = n_fft / 2 + 1;
    const long F2 = n_fft + 2;
    auto up = [](long n) { return (n + 63) / 64 * 64; };
    float* xbest = ws;
    float* est = xbest + up(F2 * BT);
    float* fr = est + up(F2 * BT);                 // frames of the STFT / time frames of the iSTFT (never live together)
    float* xt = fr + up((long)win_length * BT);
    hipStream_t s = (hipStream_t)stream;
    int rc = avc_launch_dsp_phase(nullptr, S, F, (int)BT, xbest, s);
    for (int i = 0; i < n_iter && !rc; ++i) {
        rc = avc_dsp_istft_batch(xbest, B, T, n_fft, hop_length, win_length, basis_inv, fr, xt, stream);
        if (!rc) rc = avc_dsp_stft_batch(xt, Ly, B, n_fft, hop_length, win_length, basis_fwd, fr, est, stream);   // 1 + Ly / hop == T frames
        if (!rc) rc = avc_launch_dsp_phase(est, S, F, (int)BT, xbest, s);
    }
    if (!rc) rc = avc_dsp_istft_batch(xbest, B, T, n_fft, hop_length, win_length, basis_inv, fr, y, stream);
    return rc;
}
int avc_dsp_griffin_lim(const float* S, int T, int n_fft, int hop_length, int win_length, int n_iter, const float* basis_fwd,
                        const float* basis_inv, float* ws, float* y, void* stream) {
    return avc_dsp_griffin_lim_batch(S, 1, T, n_fft, hop_length, win_length, n_iter, basis_fwd, basis_inv, ws, y, stream);
}

// Utterances of DIFFERENT lengths: S is [F][Ttot] with utterance b in columns toff[b] .. toff[b+1]-1 (toff: B + 1 device ints,
// toff[0] = 0, toff[B] = Ttot), y receives the waveforms back to back: utterance b at sample hop (toff[b] - b).  toff_host is the
// HOST copy of the same B + 1 offsets: the kernels index frames and samples through the device array without bounds checks, so the
// library validates the host copy (monotone, toff[0] = 0, toff[B] = Ttot, hop (T_b - 1) > n_fft / 2 for the reflect padding of every
// utterance's STFT: -6, as the equal-length entry point) and the caller keeps the two identical.
int avc_dsp_griffin_lim_ragged(const float* S, const int* toff, const int* toff_host, int B, int Ttot, int n_fft, int hop_length, int win_length,
                               int n_iter, const float* basis_fwd, const float* basis_inv, float* ws, float* y, void* stream) {
    if (!dsp_geom_ok(n_fft, hop_length, win_length) || !S || !toff || !toff_host || !basis_fwd || !basis_inv || !ws || !y || B < 1 || Ttot < 2 * B ||
        n_iter < 0)
        return -1;
    if (toff_host[0] != 0 || toff_host[B] != Ttot) return -1;
    for (int b = 0; b < B; ++b) {
        const int Tb = toff_host[b + 1] - toff_host[b];
        if (Tb < 2) return -1;
        if ((long)hop_length * (Tb - 1) <= n_fft / 2) return -6;
    }
    const int F = n_fft / 2 + 1, F2 = n_fft + 2;
    auto up = [](long n) { return (n + 63) / 64 * 64; };
    float* xbest = ws;
    float* est = xbest + up((long)F2 * Ttot);
    float* fr = est + up((long)F2 * Ttot);
    float* xt = fr + up((long)win_length * Ttot);
    hipStream_t s = (hipStream_t)stream;
    auto istft = [&](const float* spec, float* out) {
        int rc = avc_conv1d_fwd(spec, 0, Ttot, 1, 1, F2, Ttot, basis_inv, nullptr, win_length, 1, 1, 0, fr, 0, Ttot, 1, 1, nullptr, 0, 0, 0, 0, 0,
                                nullptr, 0, stream);
        return rc ? rc : avc_launch_dsp_ola_ragged(fr, toff, B, Ttot, hop_length, n_fft, win_length, out, s);
    };
    int rc = avc_launch_dsp_phase(nullptr, S, F, Ttot, xbest, s);
    for (int i = 0; i < n_iter && !rc; ++i) {
        rc = istft(xbest, xt);
        if (!rc) rc = avc_launch_dsp_frames_ragged(xt, toff, B, Ttot, hop_length, n_fft, win_length, fr, s);
        if (!rc) rc = avc_conv1d_fwd(fr, 0, Ttot, 1, 1, win_length, Ttot, basis_fwd, nullptr, F2, 1, 1, 0, est, 0, Ttot, 1, 1, nullptr, 0, 0, 0, 0, 0,
                                     nullptr, 0, stream);
        if (!rc) rc = avc_launch_dsp_phase(est, S, F, Ttot, xbest, s);
    }
    if (!rc) rc = istft(xbest, y);
    return rc;
}

int avc_dsp_magnitude(const float* spec, int n_fft, int T, float* mag, void* stream) {
    if (!spec || !mag || T < 1) return -1;
    return avc_launch_dsp_mag(spec, n_fft / 2 + 1, T, mag, (hipStream_t)stream);
}
int avc_dsp_db_normalize(const float* in, int C, int T, float ref_db, float max_db, float* out, void* stream) {
    if (!in || !out || C < 1 || T < 1) return -1;
    return avc_launch_dsp_db_norm(in, C, T, ref_db, max_db, out, (hipStream_t)stream);
}
int avc_dsp_denormalize_amp(const float* in, int C, int T, float ref_db, float max_db, float* out, void* stream) {
    if (!in || !out || C < 1 || T < 1) return -1;
    return avc_launch_dsp_denorm_amp(in, C, T, ref_db, max_db, out, (hipStream_t)stream);
}
int avc_dsp_preemphasis(const float* y, long L, float a, float* out, void* stream) {
    if (!y || !out || L < 1 || y == out) return -1;
    return avc_launch_dsp_preemph(y, L, a, out, (hipStream_t)stream);
}
int avc_dsp_deemphasis(const float* x, long L, float a, float* out, void* stream) {
    if (!x || !out || L < 1) return -1;
    return avc_launch_dsp_deemph(x, L, a, out, (hipStream_t)stream);
}
int avc_dsp_frame_power(const float* y, long L, int frame_length, int hop_length, float* out, void* stream) {
    if (!y || !out || frame_length < 2 || hop_length < 1) return -1;
    if (L <= frame_length / 2) return -6;
    return avc_launch_dsp_frame_power(y, L, frame_length, hop_length, avc_dsp_num_frames(L, hop_length), out, (hipStream_t)stream);
}
}
