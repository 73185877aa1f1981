// Mel <-> waveform DSP on the device (reference: preprocess/tacotron/utils.py:27-155 with
// hyperparams.py:20-34 -- there it is librosa on the CPU, and the 100-iteration Griffin-Lim dominates the
// end-to-end latency of a conversion).
//
// The two transforms are GEMMs against DFT bases that carry the (centre-padded) Hann window, run on the same
// exact-fp32 MFMA kernel as the model's 1x1 convolutions (conv_gemm.hip):
//
//   STFT   (utils.py:63-66,141)  spec[2f+ri, t] = sum_k Wf[2f+ri, k] * frames[k, t],  frames[k, t] = y_reflect[t*hop + n0 + k - n_fft/2]
//                                Wf[2f, k] = w[k] cos(2 pi f (k+n0) / n_fft),  Wf[2f+1, k] = -w[k] sin(...)      (k < win_length)
//   iSTFT  (utils.py:150-154)    tf[k, t] = sum_{f,ri} Wi[k, 2f+ri] * spec[2f+ri, t]   (the windowed irfft of frame t on the
//                                window's support), then overlap-add / window-sum-of-squares in one gather kernel.
//
// Only the win_length taps under the window are multiplied (K = 1200 of n_fft = 2048).  Everything else here is the
// row / elementwise glue: framing with librosa's reflect padding, |.|, the Griffin-Lim phase step, dB + normalise
// (+ transpose to the [T, C] layout of the pickles), pre- / de-emphasis (the latter a first-order IIR as a blocked scan),
// frame powers for effects.trim.  A complex spectrogram is stored as [2F][T] fp32 rows (Re, Im interleaved), T contiguous.
#include <hip/hip_runtime.h>
#include <math.h>

#include "avc_common.h"
#include "avc_internal.h"

static inline int dsp_blocks(long n) {
    long b = (n + AVC_THREADS - 1) / AVC_THREADS;
    return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

// numpy.pad(mode='reflect') index for a pad shorter than the signal
static __device__ __forceinline__ long dsp_reflect(long v, long L) {
    if (v < 0) v = -v;
    if (v >= L) v = 2 * (L - 1) - v;
    return v;
}

// which = 0: forward basis in state-dict layout [2F][win] (Cout = 2F, Cin = win);  which = 1: inverse basis [win][2F]
__global__ void __launch_bounds__(AVC_THREADS) dsp_basis_kernel(int which, int n_fft, int win, float* W) {
    const int F = n_fft / 2 + 1, n0 = (n_fft - win) / 2;
    const long total = (long)2 * F * win;
    for (long e = (long)blockIdx.x * AVC_THREADS + threadIdx.x; e < total; e += (long)gridDim.x * AVC_THREADS) {
        int row, k;
        if (which == 0) { row = (int)(e / win); k = (int)(e - (long)row * win); }
        else { k = (int)(e / (2 * F)); row = (int)(e - (long)k * 2 * F); }
        const int f = row >> 1, ri = row & 1;
        const double w = 0.5 - 0.5 * cospi(2.0 * (double)k / (double)win);          // periodic Hann (fftbins=True)
        const long m = ((long)f * (k + n0)) % n_fft;                                 // exact angle reduction
        double s, c;
        sincospi(2.0 * (double)m / (double)n_fft, &s, &c);
        double v = ri ? -s : c;
        if (which == 1) {   // irfft: x[n] = (1/N) (X0 + (-1)^n X_{N/2} + 2 sum_f Re(X_f e^{+i..})); Im of DC / Nyquist is ignored
            const bool edge = (f == 0) || (f == n_fft / 2);
            v = ri ? (edge ? 0.0 : -2.0 * s) : (edge ? c : 2.0 * c);
            v /= (double)n_fft;
        }
        W[e] = (float)(w * v);
    }
}

// B signals of L samples each (y[b * L + i]) -> frames[k][b * T + t]: the utterances of a batch are columns of ONE GEMM
__global__ void __launch_bounds__(AVC_THREADS) dsp_frames_kernel(const float* y, long L, int B, int T, int hop, int n_fft, int win, float* frames) {
    const int n0 = (n_fft - win) / 2;
    const long BT = (long)B * T, total = (long)win * BT;
    for (long e = (long)blockIdx.x * AVC_THREADS + threadIdx.x; e < total; e += (long)gridDim.x * AVC_THREADS) {
        const int k = (int)(e / BT);
        const long c = e - (long)k * BT;
        const int b = (int)(c / T), t = (int)(c - (long)b * T);
        frames[e] = y[(long)b * L + dsp_reflect((long)t * hop + n0 + k - n_fft / 2, L)];
    }
}

// y[s] = sum_t tf[q - t*hop - n0, t] / sum_t w^2[q - t*hop - n0],  q = s + n_fft/2   (tf already carries the window)
__global__ void __launch_bounds__(AVC_THREADS) dsp_ola_kernel(const float* tf, int B, int T, int hop, int n_fft, int win, float* y, long Ly) {
    const int n0 = (n_fft - win) / 2;
    const long BT = (long)B * T;
    for (long e = (long)blockIdx.x * AVC_THREADS + threadIdx.x; e < (long)B * Ly; e += (long)gridDim.x * AVC_THREADS) {
        const int b = (int)(e / Ly);
        const long s = e - (long)b * Ly;
        const long q = s + n_fft / 2 - n0;          // k = q - t*hop must lie in [0, win)
        long t1 = q / hop;
        long t0 = (q - win + hop) / hop;            // ceil((q - win + 1) / hop) for q - win + 1 > 0
        if (q - win + 1 <= 0) t0 = 0;
        if (t1 > T - 1) t1 = T - 1;
        float acc = 0.f, wss = 0.f;
        for (long t = t0; t <= t1; ++t) {
            const int k = (int)(q - t * hop);
            const float w = 0.5f - 0.5f * cospif(2.0f * (float)k / (float)win);
            acc += tf[(long)k * BT + (long)b * T + t];
            wss += w * w;
        }
        y[e] = wss > 1.17549435e-38f ? acc / wss : acc;
    }
}

// ---- utterances of DIFFERENT lengths in one launch set: toff[b] = first frame (column) of utterance b, toff[B] = all frames;
// utterance b has T_b = toff[b+1] - toff[b] frames and hop (T_b - 1) samples, stored back to back from sample hop (toff[b] - b)
static __device__ __forceinline__ int dsp_find(const int* toff, int B, long key, int hop, bool samples) {
    int lo = 0, hi = B - 1;   // largest b with start(b) <= key
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        const long start = samples ? (long)hop * (toff[mid] - mid) : (long)toff[mid];
        if (start <= key) lo = mid; else hi = mid - 1;
    }
    return lo;
}
__global__ void __launch_bounds__(AVC_THREADS) dsp_frames_ragged_kernel(const float* y, const int* toff, int B, int hop, int n_fft, int win, float* frames) {
    const int n0 = (n_fft - win) / 2;
    const long TT = toff[B], total = (long)win * TT;
    for (long e = (long)blockIdx.x * AVC_THREADS + threadIdx.x; e < total; e += (long)gridDim.x * AVC_THREADS) {
        const int k = (int)(e / TT);
        const long c = e - (long)k * TT;
        const int b = dsp_find(toff, B, c, hop, false);
        const int t = (int)(c - toff[b]);
        const long L = (long)hop * (toff[b + 1] - toff[b] - 1);
        frames[e] = y[(long)hop * (toff[b] - b) + dsp_reflect((long)t * hop + n0 + k - n_fft / 2, L)];
    }
}
__global__ void __launch_bounds__(AVC_THREADS) dsp_ola_ragged_kernel(const float* tf, const int* toff, int B, int hop, int n_fft, int win, float* y) {
    const int n0 = (n_fft - win) / 2;
    const long TT = toff[B], Ly = (long)hop * (TT - B);
    for (long e = (long)blockIdx.x * AVC_THREADS + threadIdx.x; e < Ly; e += (long)gridDim.x * AVC_THREADS) {
        const int b = dsp_find(toff, B, e, hop, true);
        const long s = e - (long)hop * (toff[b] - b);
        const int T = toff[b + 1] - toff[b];
        const long q = s + n_fft / 2 - n0;
        long t1 = q / hop;
        long t0 = (q - win + hop) / hop;
        if (q - win + 1 <= 0) t0 = 0;
        if (t1 > T - 1) t1 = T - 1;
        float acc = 0.f, wss = 0.f;
        for (long t = t0; t <= t1; ++t) {
            const int k = (int)(q - t * hop);
            const float w = 0.5f - 0.5f * cospif(2.0f * (float)k / (float)win);
            acc += tf[(long)k * TT + toff[b] + t];
            wss += w * w;
        }
        y[e] = wss > 1.17549435e-38f ? acc / wss : acc;
    }
}

// Griffin-Lim projection (utils.py:142-143): X_best = spectrogram * est / max(1e-8, |est|)
__global__ void __launch_bounds__(AVC_THREADS) dsp_phase_kernel(const float* est, const float* S, int F, int T, float* out) {
    const long total = (long)F * T;
    for (long e = (long)blockIdx.x * AVC_THREADS + threadIdx.x; e < total; e += (long)gridDim.x * AVC_THREADS) {
        const int f = (int)(e / T), t = (int)(e - (long)f * T);
        const float re = est[(long)(2 * f) * T + t], im = est[(long)(2 * f + 1) * T + t];
        const float sc = S[e] / fmaxf(1e-8f, sqrtf(re * re + im * im));
        out[(long)(2 * f) * T + t] = re * sc;
        out[(long)(2 * f + 1) * T + t] = im * sc;
    }
}
// first iteration: X_best = spectrogram (real)
__global__ void __launch_bounds__(AVC_THREADS) dsp_real_to_complex_kernel(const float* S, int F, int T, float* out) {
    const long total = (long)F * T;
    for (long e = (long)blockIdx.x * AVC_THREADS + threadIdx.x; e < total; e += (long)gridDim.x * AVC_THREADS) {
        const int f = (int)(e / T), t = (int)(e - (long)f * T);
        out[(long)(2 * f) * T + t] = S[e];
        out[(long)(2 * f + 1) * T + t] = 0.f;
    }
}
__global__ void __launch_bounds__(AVC_THREADS) dsp_mag_kernel(const float* spec, int F, int T, float* mag) {
    const long total = (long)F * T;
    for (long e = (long)blockIdx.x * AVC_THREADS + threadIdx.x; e < total; e += (long)gridDim.x * AVC_THREADS) {
        const int f = (int)(e / T), t = (int)(e - (long)f * T);
        const float re = spec[(long)(2 * f) * T + t], im = spec[(long)(2 * f + 1) * T + t];
        mag[e] = sqrtf(re * re + im * im);
    }
}
// utils.py:76-87: out[t][c] = clip((20 log10(max(1e-5, in[c][t])) - ref_db + max_db) / max_db, 1e-8, 1)
__global__ void __launch_bounds__(AVC_THREADS) dsp_db_norm_kernel(const float* in, int C, int T, float ref_db, float max_db, float* out) {
    const long total = (long)C * T;
    for (long e = (long)blockIdx.x * AVC_THREADS + threadIdx.x; e < total; e += (long)gridDim.x * AVC_THREADS) {
        const int t = (int)(e / C), c = (int)(e - (long)t * C);
        const float db = 20.0f * log10f(fmaxf(1e-5f, in[(long)c * T + t]));
        out[e] = fminf(fmaxf((db - ref_db + max_db) / max_db, 1e-8f), 1.0f);
    }
}
// utils.py:95-98: out[c][t] = 10 ^ (0.05 * (clip(in[t][c], 0, 1) * max_db - max_db + ref_db))
__global__ void __launch_bounds__(AVC_THREADS) dsp_denorm_amp_kernel(const float* in, int C, int T, float ref_db, float max_db, float* out) {
    const long total = (long)C * T;
    for (long e = (long)blockIdx.x * AVC_THREADS + threadIdx.x; e < total; e += (long)gridDim.x * AVC_THREADS) {
        const int c = (int)(e / T), t = (int)(e - (long)c * T);
        const float v = fminf(fmaxf(in[(long)t * C + c], 0.f), 1.f) * max_db - max_db + ref_db;
        out[e] = exp10f(v * 0.05f);
    }
}
// utils.py:60: out[0] = y[0], out[i] = y[i] - a y[i-1]
__global__ void __launch_bounds__(AVC_THREADS) dsp_preemph_kernel(const float* y, long L, float a, float* out) {
    for (long i = (long)blockIdx.x * AVC_THREADS + threadIdx.x; i < L; i += (long)gridDim.x * AVC_THREADS)
        out[i] = i == 0 ? y[0] : y[i] - a * y[i - 1];
}
// utils.py:104: scipy.signal.lfilter([1], [1, -a], x): out[i] = x[i] + a out[i-1].  ONE workgroup of 1024 lanes, each a
// contiguous run: local recurrence, then the carries in sequence (1024 steps), then out += a^(j+1) * carry_in.
__global__ void __launch_bounds__(1024) dsp_deemph_kernel(const float* x, long L, float a, float* out) {
    __shared__ float last[1024], cin[1024], apow[1024];
    const int tid = threadIdx.x;
    const long run = (L + 1023) / 1024;
    const long i0 = (long)tid * run, i1 = (i0 + run < L) ? i0 + run : L;
    float v = 0.f, p = 1.f;
    for (long i = i0; i < i1; ++i) {
        v = x[i] + a * v;
        out[i] = v;
        p *= a;
    }
    last[tid] = v;
    apow[tid] = p;
    __syncthreads();
    if (tid == 0) {
        float c = 0.f;
        for (int j = 0; j < 1024; ++j) {
            cin[j] = c;
            c = last[j] + apow[j] * c;
        }
    }
    __syncthreads();
    const float c = cin[tid];
    if (c != 0.f) {
        float q = a;
        for (long i = i0; i < i1; ++i) {
            out[i] += q * c;
            q *= a;
        }
    }
}
// mean square of the frames librosa.feature.rms(frame_length, hop_length, center=True) cuts (effects.trim, utils.py:57,107)
__global__ void __launch_bounds__(AVC_THREADS) dsp_frame_power_kernel(const float* y, long L, int frame_length, int hop, float* out) {
    __shared__ float red[AVC_THREADS];
    const int fr = blockIdx.x;
    float s = 0.f;
    for (int k = threadIdx.x; k < frame_length; k += AVC_THREADS) {
        const float v = y[dsp_reflect((long)fr * hop + k - frame_length / 2, L)];
        s += v * v;
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = AVC_THREADS / 2; o >= 1; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[fr] = red[0] / (float)frame_length;
}

// --------------------------------------------------------------------------
int avc_launch_dsp_basis(int which, int n_fft, int win, float* W, hipStream_t s) {
    hipLaunchKernelGGL(dsp_basis_kernel, dim3(dsp_blocks((long)(n_fft + 2) * win)), dim3(AVC_THREADS), 0, s, which, n_fft, win, W);
    return (int)hipGetLastError();
}
int avc_launch_dsp_frames(const float* y, long L, int B, int T, int hop, int n_fft, int win, float* frames, hipStream_t s) {
    ProfScope ps(AVC_K_MISC, 0.0, 8.0 * (double)win * B * T, s);
    hipLaunchKernelGGL(dsp_frames_kernel, dim3(dsp_blocks((long)win * B * T)), dim3(AVC_THREADS), 0, s, y, L, B, T, hop, n_fft, win, frames);
    return (int)hipGetLastError();
}
int avc_launch_dsp_ola(const float* tf, int B, int T, int hop, int n_fft, int win, float* y, hipStream_t s) {
    const long Ly = (long)hop * (T - 1);
    if (Ly < 1) return -1;
    ProfScope ps(AVC_K_MISC, 0.0, 4.0 * B * ((double)win * T + Ly), s);
    hipLaunchKernelGGL(dsp_ola_kernel, dim3(dsp_blocks((long)B * Ly)), dim3(AVC_THREADS), 0, s, tf, B, T, hop, n_fft, win, y, Ly);
    return (int)hipGetLastError();
}
int avc_launch_dsp_frames_ragged(const float* y, const int* toff, int B, int Ttot, int hop, int n_fft, int win, float* frames, hipStream_t s) {
    hipLaunchKernelGGL(dsp_frames_ragged_kernel, dim3(dsp_blocks((long)win * Ttot)), dim3(AVC_THREADS), 0, s, y, toff, B, hop, n_fft, win, frames);
    return (int)hipGetLastError();
}
int avc_launch_dsp_ola_ragged(const float* tf, const int* toff, int B, int Ttot, int hop, int n_fft, int win, float* y, hipStream_t s) {
    hipLaunchKernelGGL(dsp_ola_ragged_kernel, dim3(dsp_blocks((long)hop * (Ttot - B))), dim3(AVC_THREADS), 0, s, tf, toff, B, hop, n_fft, win, y);
    return (int)hipGetLastError();
}
int avc_launch_dsp_phase(const float* est, const float* S, int F, int T, float* out, hipStream_t s) {
    ProfScope ps(AVC_K_MISC, 0.0, 20.0 * (double)F * T, s);
    if (est) hipLaunchKernelGGL(dsp_phase_kernel, dim3(dsp_blocks((long)F * T)), dim3(AVC_THREADS), 0, s, est, S, F, T, out);
    else hipLaunchKernelGGL(dsp_real_to_complex_kernel, dim3(dsp_blocks((long)F * T)), dim3(AVC_THREADS), 0, s, S, F, T, out);
    return (int)hipGetLastError();
}
int avc_launch_dsp_mag(const float* spec, int F, int T, float* mag, hipStream_t s) {
    ProfScope ps(AVC_K_MISC, 0.0, 12.0 * (double)F * T, s);
    hipLaunchKernelGGL(dsp_mag_kernel, dim3(dsp_blocks((long)F * T)), dim3(AVC_THREADS), 0, s, spec, F, T, mag);
    return (int)hipGetLastError();
}
int avc_launch_dsp_db_norm(const float* in, int C, int T, float ref_db, float max_db, float* out, hipStream_t s) {
    hipLaunchKernelGGL(dsp_db_norm_kernel, dim3(dsp_blocks((long)C * T)), dim3(AVC_THREADS), 0, s, in, C, T, ref_db, max_db, out);
    return (int)hipGetLastError();
}
int avc_launch_dsp_denorm_amp(const float* in, int C, int T, float ref_db, float max_db, float* out, hipStream_t s) {
    hipLaunchKernelGGL(dsp_denorm_amp_kernel, dim3(dsp_blocks((long)C * T)), dim3(AVC_THREADS), 0, s, in, C, T, ref_db, max_db, out);
    return (int)hipGetLastError();
}
int avc_launch_dsp_preemph(const float* y, long L, float a, float* out, hipStream_t s) {
    hipLaunchKernelGGL(dsp_preemph_kernel, dim3(dsp_blocks(L)), dim3(AVC_THREADS), 0, s, y, L, a, out);
    return (int)hipGetLastError();
}
int avc_launch_dsp_deemph(const float* x, long L, float a, float* out, hipStream_t s) {
    hipLaunchKernelGGL(dsp_deemph_kernel, dim3(1), dim3(1024), 0, s, x, L, a, out);
    return (int)hipGetLastError();
}
int avc_launch_dsp_frame_power(const float* y, long L, int frame_length, int hop, int n_frames, float* out, hipStream_t s) {
    hipLaunchKernelGGL(dsp_frame_power_kernel, dim3(n_frames), dim3(AVC_THREADS), 0, s, y, L, frame_length, hop, out);
    return (int)hipGetLastError();
}
