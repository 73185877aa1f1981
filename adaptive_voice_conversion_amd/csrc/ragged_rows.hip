// Row / glue kernels of the RAGGED forward pass (AE.inference over utterances of different lengths in one launch set;
// reference: inference.py:54-70, model.py:387-391 -- the reference converts one utterance per call).
//
// Packed buffers: sample b owns a contiguous [channels][T_b] block starting at element channels * off[b] (off = prefix sums
// of the per-sample lengths at that level).  InstanceNorm statistics, reflect padding and ceil-mode pooling all depend on the
// TRUE length of a sample, so nothing is padded: every row is processed at its own length.
#include <hip/hip_runtime.h>

#include "avc_common.h"
#include "avc_internal.h"

static __device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// nn.InstanceNorm1d(affine=False) (model.py:296,341) [+ append_cond, model.py:77-83] + ReLU [+ residual join]: one wavefront
// per row (b, c), three cached passes over the row's own T_b frames -- the arithmetic (two-pass statistics, in_xhat / in_preact)
// is instnorm_fwd_generic_kernel's of rowops.hip.
__global__ void __launch_bounds__(AVC_THREADS) rag_instnorm_fwd_kernel(const RagINArgs a) {
    const int tid = threadIdx.x, l = tid & 63;
    const int row = blockIdx.x * 4 + (tid >> 6);
    if (row >= a.B * a.C) return;   // (whole wavefronts leave together: a row is one wavefront)
    const int b = row / a.C, c = row - b * a.C;
    const int T = a.T[b];
    const long base = (long)a.C * a.off[b] + (long)c * T;
    const float* yrow = a.y + base;
    float s = 0.f;
    for (int t = l; t < T; t += 64) s += yrow[t];
    const float invT = 1.0f / (float)T;
    const float mean = wave_sum(s) * invT;
    float ss = 0.f;
    for (int t = l; t < T; t += 64) {
        float d = yrow[t] - mean;
        ss += d * d;
    }
    const float rstd = 1.0f / sqrtf(wave_sum(ss) * invT + AVC_IN_EPS);   // biased variance
    float gamma = 1.f, beta = 0.f;
    if (a.cond) {
        const float* cr = a.cond + (long)b * a.cond_sb + a.cond_off;
        beta = cr[c];
        gamma = cr[a.C + c];
    }
    float* orow = a.out + base;
    int Tres = 0;
    const float* rrow = nullptr;
    if (a.res) {
        Tres = a.Tres[b];
        rrow = a.res + (long)a.C * a.offres[b] + (long)c * Tres;
    }
    for (int t = l; t < T; t += 64) {
        float w = avc_act(in_preact(in_xhat(yrow[t], mean, rstd), gamma, beta), a.slope);
        if (rrow) {
            if (a.res_mode == AVC_RES_IDENTITY)
                w += rrow[t];
            else if (a.res_mode == AVC_RES_UP2)          // nearest x2 (model.py:61-63)
                w += rrow[t >> 1];
            else if (a.res_mode == AVC_RES_AVGPOOL2) {   // avg_pool1d(2, ceil_mode): the clipped last window divides by 1 (model.py:319)
                const int i0 = 2 * t, i1 = i0 + 1;
                w += (i1 < Tres) ? (rrow[i0] + rrow[i1]) * 0.5f : rrow[i0];
            }
        }
        orow[t] = w;
    }
}

static __device__ __forceinline__ int rag_find(const int* off, int B, int f) {   // sample whose frames [off[b], off[b+1]) hold frame f
    int lo = 0, hi = B - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (off[mid] <= f) lo = mid;
        else hi = mid - 1;
    }
    return lo;
}

// raw input last in the concat buffer (model.py:90): dst[b, c0 + m, t] = x[b, m, t]; x is the packed input with explicit
// element strides (xsc, xst) inside a sample's block of M * T_b elements
__global__ void __launch_bounds__(AVC_THREADS)
rag_copy_rows_kernel(const float* x, long xsc, long xst, const int* T, const int* off, int B, int M, int sumT, float* dst, int CC, int c0) {
    const long n = (long)M * sumT;
    for (long e = (long)blockIdx.x * AVC_THREADS + threadIdx.x; e < n; e += (long)gridDim.x * AVC_THREADS) {
        // e = M * off[b] + m * T_b + t  (the order of the destination rows: coalesced stores)
        const int f = (int)(e / M);                 // a frame index inside sample b's block
        const int b = rag_find(off, B, f);
        const long r = e - (long)M * off[b];
        const int Tb = T[b];
        const int m = (int)(r / Tb), t = (int)(r - (long)m * Tb);
        dst[(long)CC * off[b] + (long)(c0 + m) * Tb + t] = x[(long)M * off[b] + (long)m * xsc + (long)t * xst];
    }
}

// AdaptiveAvgPool1d(1) (model.py:273) over each sample's own length -> out[c * B + b] (channel-major for the dense stack)
__global__ void __launch_bounds__(AVC_THREADS) rag_timepool_fwd_kernel(const float* in, const int* T, const int* off, int B, int C, float* out) {
    const int r = blockIdx.x * AVC_THREADS + threadIdx.x;
    if (r >= B * C) return;
    const int b = r / C, c = r - b * C;
    const int Tb = T[b];
    const float* p = in + (long)C * off[b] + (long)c * Tb;
    float s = 0.f;
    for (int t = 0; t < Tb; ++t) s += p[t];
    out[(long)c * B + b] = s / (float)Tb;
}

int avc_launch_rag_in_fwd(const RagINArgs& a, hipStream_t s) {
    ProfScope ps(AVC_K_IN_FWD, 0.0, 0.0, s);
    hipLaunchKernelGGL(rag_instnorm_fwd_kernel, dim3(avc_cdiv(a.B * a.C, 4)), dim3(AVC_THREADS), 0, s, a);
    return (int)hipGetLastError();
}
int avc_launch_rag_copy_rows(const float* x, long xsc, long xst, const int* T, const int* off, int B, int M, int sumT, float* dst, int CC, int c0,
                             hipStream_t s) {
    long n = (long)M * sumT;
    long blocks = (n + AVC_THREADS - 1) / AVC_THREADS;
    if (blocks > 2048) blocks = 2048;
    ProfScope ps(AVC_K_MISC, 0.0, 0.0, s);
    hipLaunchKernelGGL(rag_copy_rows_kernel, dim3((int)blocks), dim3(AVC_THREADS), 0, s, x, xsc, xst, T, off, B, M, sumT, dst, CC, c0);
    return (int)hipGetLastError();
}
int avc_launch_rag_timepool_fwd(const float* in, const int* T, const int* off, int B, int C, float* out, hipStream_t s) {
    ProfScope ps(AVC_K_MISC, 0.0, 0.0, s);
    hipLaunchKernelGGL(rag_timepool_fwd_kernel, dim3(avc_cdiv(B * C, AVC_THREADS)), dim3(AVC_THREADS), 0, s, in, T, off, B, C, out);
    return (int)hipGetLastError();
}
