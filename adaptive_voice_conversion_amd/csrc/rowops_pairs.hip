// HBM-bound row kernels on bf16 PAIR tensors (bf16_pairs.h; compute_dtype "bf16", BASELINE configs[2]).
//
// InstanceNorm1d(affine=False) + AdaIN affine + ReLU + residual join, forward and backward -- the arithmetic of rowops.hip
// (reference: nn.InstanceNorm1d model.py:296,341; append_cond :77-83; block bodies :309-320 / :353-369 and their autograd) with
// HALF the bytes per element: a dword row of T frames carries the rows of channels 2p and 2p + 1, statistics and all
// arithmetic are fp32, values are rounded to bf16 once, on store.  LPR lanes share a dword row, each lane keeps NV x 4 frames
// x 2 channels in registers between the statistics pass and the normalise pass: forward = 1 read + 1 write of the row.
//
// "Planar" rows: a pixel-shuffling conv (model.py:52-59, upsample 2) stores conv-output pairs (rows 2c, 2c + 1 at frame t), which
// IS the natural [B][C][2 T] bf16 layout of the shuffled tensor: channel c's row is 2 T contiguous bf16.  The InstanceNorm after
// such a conv reads two planar rows per dword row of its output, and its backward writes dy planar -- exactly the pair layout the
// conv's input-gradient and weight-gradient launches read.
#include <hip/hip_runtime.h>
#include <stdlib.h>

#include "avc_common.h"
#include "avc_internal.h"
#include "bf16_pairs.h"

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

template <int LPR>
static __device__ __forceinline__ float group_sum(float v) { return avc_group_sum<LPR>(v); }   // (avc_common.h: no ds_bpermute)

// frames 4 i4 .. 4 i4 + 3 of the dword row (b, p): lo[k] / hi[k] = channel 2p / 2p + 1 at frame 4 i4 + k
static __device__ __forceinline__ void ld_pairs4(const unsigned* rows, long row, int T, int i4, float (&lo)[4], float (&hi)[4]) {
    const u32x4 v = *(const u32x4*)(rows + row * T + 4 * i4);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        lo[k] = bh_lo(v[k]);
        hi[k] = bh_hi(v[k]);
    }
}
// the same frames from planar rows: channel c's row = T bf16 = T / 2 dwords at dword offset (b C + c) T / 2
static __device__ __forceinline__ void ld_planar4(const unsigned* rows, long row, int T, int i4, float (&lo)[4], float (&hi)[4]) {
    const u32x2 a = *(const u32x2*)(rows + (2 * row) * (T >> 1) + 2 * i4);
    const u32x2 b = *(const u32x2*)(rows + (2 * row + 1) * (T >> 1) + 2 * i4);
    lo[0] = bh_lo(a[0]); lo[1] = bh_hi(a[0]); lo[2] = bh_lo(a[1]); lo[3] = bh_hi(a[1]);
    hi[0] = bh_lo(b[0]); hi[1] = bh_hi(b[0]); hi[2] = bh_lo(b[1]); hi[3] = bh_hi(b[1]);
}
static __device__ __forceinline__ void st_pairs4(unsigned* rows, long row, int T, int i4, const float (&lo)[4], const float (&hi)[4]) {
    u32x4 v;
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = bh_pack(lo[k], hi[k]);
    *(u32x4*)(rows + row * T + 4 * i4) = v;
}
static __device__ __forceinline__ void st_planar4(unsigned* rows, long row, int T, int i4, const float (&lo)[4], const float (&hi)[4]) {
    u32x2 a, b;
    a[0] = bh_pack(lo[0], lo[1]); a[1] = bh_pack(lo[2], lo[3]);
    b[0] = bh_pack(hi[0], hi[1]); b[1] = bh_pack(hi[2], hi[3]);
    *(u32x2*)(rows + (2 * row) * (T >> 1) + 2 * i4) = a;
    *(u32x2*)(rows + (2 * row + 1) * (T >> 1) + 2 * i4) = b;
}

// residual for output frames 4 i4 .. + 3 of dword row `row` (pair rows of Tres dwords)
static __device__ __forceinline__ void res_pairs4(const unsigned* res, long row, int mode, int i4, int Tres, float (&lo)[4], float (&hi)[4]) {
    const unsigned* rrow = res + row * Tres;
    if (mode == AVC_RES_IDENTITY) {
        const u32x4 v = *(const u32x4*)(rrow + 4 * i4);
#pragma unroll
        for (int k = 0; k < 4; ++k) { lo[k] = bh_lo(v[k]); hi[k] = bh_hi(v[k]); }
    } else if (mode == AVC_RES_UP2) {          // nearest x2 (model.py:61-63): Tres = T / 2
        const u32x2 v = *(const u32x2*)(rrow + 2 * i4);
        lo[0] = lo[1] = bh_lo(v[0]); hi[0] = hi[1] = bh_hi(v[0]);
        lo[2] = lo[3] = bh_lo(v[1]); hi[2] = hi[3] = bh_hi(v[1]);
    } else if (mode == AVC_RES_AVGPOOL2) {     // avg_pool1d(2, ceil_mode) with Tres = 2 T (model.py:319)
        const u32x4 p = *(const u32x4*)(rrow + 8 * i4), q = *(const u32x4*)(rrow + 8 * i4 + 4);
        lo[0] = (bh_lo(p[0]) + bh_lo(p[1])) * 0.5f; hi[0] = (bh_hi(p[0]) + bh_hi(p[1])) * 0.5f;
        lo[1] = (bh_lo(p[2]) + bh_lo(p[3])) * 0.5f; hi[1] = (bh_hi(p[2]) + bh_hi(p[3])) * 0.5f;
        lo[2] = (bh_lo(q[0]) + bh_lo(q[1])) * 0.5f; hi[2] = (bh_hi(q[0]) + bh_hi(q[1])) * 0.5f;
        lo[3] = (bh_lo(q[2]) + bh_lo(q[3])) * 0.5f; hi[3] = (bh_hi(q[2]) + bh_hi(q[3])) * 0.5f;
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) lo[k] = hi[k] = 0.f;
    }
}

// a.R = B * C / 2 dword rows, a.C = channels, a.T = frames (T % 4 == 0); mean / rstd are per channel row (b C + c), as in rowops.hip
template <int LPR, int NV>
__global__ void __launch_bounds__(AVC_THREADS) instnorm_fwd_pairs_kernel(const INFwdArgs a) {
    constexpr int RPB = AVC_THREADS / LPR;
    const int tid = threadIdx.x;
    const int row = blockIdx.x * RPB + tid / LPR;
    const int l = tid % LPR;
    const bool rvalid = row < a.R;
    const long rr = rvalid ? row : 0;
    const int n4 = a.T >> 2;
    const unsigned* y = (const unsigned*)a.y;
    float vl[NV][4], vh[NV][4], rl[NV][4], rh[NV][4];
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int i4 = k * LPR + l;
        if (i4 < n4) {
            if (a.planar) ld_planar4(y, rr, a.T, i4, vl[k], vh[k]);
            else ld_pairs4(y, rr, a.T, i4, vl[k], vh[k]);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) vl[k][e] = vh[k][e] = 0.f;
        }
    }
    // the AdaIN scale / shift and the residual are requested NOW, behind the row itself (rowops.hip instnorm_fwd_kernel: one memory
    // round trip per launch instead of three)
    const int C2 = a.C >> 1;
    const int b = (int)(rr / C2), p = (int)(rr - (long)b * C2);
    const int c0 = 2 * p;
    float g0 = 1.f, g1 = 1.f, be0 = 0.f, be1 = 0.f;
    if (a.cond) {
        const float* cr = a.cond + (long)b * a.cond_sb + a.cond_off;
        be0 = cr[c0]; be1 = cr[c0 + 1];               // first half = shift (model.py:81)
        g0 = cr[a.C + c0]; g1 = cr[a.C + c0 + 1];     // second half = scale
    }
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int i4 = k * LPR + l;
        if (a.res && i4 < n4) {
            res_pairs4((const unsigned*)a.res, rr, a.res_mode, i4, a.Tres, rl[k], rh[k]);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) rl[k][e] = rh[k][e] = 0.f;
        }
    }
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        s0 += (vl[k][0] + vl[k][1]) + (vl[k][2] + vl[k][3]);
        s1 += (vh[k][0] + vh[k][1]) + (vh[k][2] + vh[k][3]);
    }
    const float invT = 1.0f / (float)a.T;
    const float mean0 = group_sum<LPR>(s0) * invT, mean1 = group_sum<LPR>(s1) * invT;
    float q0 = 0.f, q1 = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int i4 = k * LPR + l;
        if (i4 < n4) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float d0 = vl[k][e] - mean0, d1 = vh[k][e] - mean1;
                q0 += d0 * d0;
                q1 += d1 * d1;
            }
        }
    }
    const float rstd0 = 1.0f / sqrtf(group_sum<LPR>(q0) * invT + AVC_IN_EPS);   // biased variance
    const float rstd1 = 1.0f / sqrtf(group_sum<LPR>(q1) * invT + AVC_IN_EPS);
    if (rvalid && l == 0) {
        const long sr = (long)b * a.C + c0;
        a.mean[sr] = mean0; a.mean[sr + 1] = mean1;
        a.rstd[sr] = rstd0; a.rstd[sr + 1] = rstd1;
    }
    if (!rvalid) return;
    unsigned* out = (unsigned*)a.out;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int i4 = k * LPR + l;
        if (i4 < n4) {
            float o0[4], o1[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float w0 = in_preact(in_xhat(vl[k][e], mean0, rstd0), g0, be0);
                const float w1 = in_preact(in_xhat(vh[k][e], mean1, rstd1), g1, be1);
                o0[e] = a.relu ? avc_act(w0, a.slope) : w0;
                o1[e] = a.relu ? avc_act(w1, a.slope) : w1;
            }
            if (a.res) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { o0[e] += rl[k][e]; o1[e] += rh[k][e]; }
            }
            st_pairs4(out, rr, a.T, i4, o0, o1);
        }
    }
}

// backward (formulas: rowops.hip instnorm_bwd_kernel): g and the recomputed activation are pair rows, y / dy pair or planar rows
template <int LPR, int NV>
__global__ void __launch_bounds__(AVC_THREADS) instnorm_bwd_pairs_kernel(const INBwdArgs a) {
    constexpr int RPB = AVC_THREADS / LPR;
    const int tid = threadIdx.x;
    const int row = blockIdx.x * RPB + tid / LPR;
    const int l = tid % LPR;
    const bool rvalid = row < a.R;
    const long rr = rvalid ? row : 0;
    const int n4 = a.T >> 2;
    const int C2 = a.C >> 1;
    const int b = (int)(rr / C2), p = (int)(rr - (long)b * C2);
    const int c0 = 2 * p;
    const long sr = (long)b * a.C + c0;
    const float mean0 = a.mean[sr], mean1 = a.mean[sr + 1], rstd0 = a.rstd[sr], rstd1 = a.rstd[sr + 1];
    float g0 = 1.f, g1 = 1.f, be0 = 0.f, be1 = 0.f;
    if (a.cond) {
        const float* cr = a.cond + (long)b * a.cond_sb + a.cond_off;
        be0 = cr[c0]; be1 = cr[c0 + 1];
        g0 = cr[a.C + c0]; g1 = cr[a.C + c0 + 1];
    }
    const unsigned* y = (const unsigned*)a.y;
    const unsigned* gin = (const unsigned*)a.g;
    float xh0[NV][4], xh1[NV][4], gm0[NV][4], gm1[NV][4];
    float s10 = 0.f, s20 = 0.f, s11 = 0.f, s21 = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int i4 = k * LPR + l;
        if (i4 < n4) {
            float y0[4], y1[4], gg0[4], gg1[4];
            if (a.planar) ld_planar4(y, rr, a.T, i4, y0, y1);
            else ld_pairs4(y, rr, a.T, i4, y0, y1);
            ld_pairs4(gin, rr, a.T, i4, gg0, gg1);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float h0 = in_xhat(y0[e], mean0, rstd0), h1 = in_xhat(y1[e], mean1, rstd1);
                const float w0 = in_preact(h0, g0, be0), w1 = in_preact(h1, g1, be1);
                const float m0 = avc_act_grad(gg0[e], !a.relu || w0 > 0.f, a.slope);
                const float m1 = avc_act_grad(gg1[e], !a.relu || w1 > 0.f, a.slope);
                xh0[k][e] = h0; xh1[k][e] = h1;
                gm0[k][e] = m0; gm1[k][e] = m1;
                s10 += m0; s20 += m0 * h0;
                s11 += m1; s21 += m1 * h1;
            }
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) xh0[k][e] = xh1[k][e] = gm0[k][e] = gm1[k][e] = 0.f;
        }
    }
    s10 = group_sum<LPR>(s10); s20 = group_sum<LPR>(s20);   // dbeta, dgamma of channel 2p
    s11 = group_sum<LPR>(s11); s21 = group_sum<LPR>(s21);   // ... 2p + 1
    if (!rvalid) return;
    if (a.dcond && l == 0) {
        float* dc = a.dcond + (long)b * a.dcond_sb + a.dcond_off;
        dc[c0] = s10; dc[c0 + 1] = s11;
        dc[a.C + c0] = s20; dc[a.C + c0 + 1] = s21;
    }
    const float invT = 1.0f / (float)a.T;
    const float m10 = g0 * s10 * invT, m20 = g0 * s20 * invT, m11 = g1 * s11 * invT, m21 = g1 * s21 * invT;
    unsigned* dy = (unsigned*)a.dy;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int i4 = k * LPR + l;
        if (i4 < n4) {
            float o0[4], o1[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                o0[e] = rstd0 * (gm0[k][e] * g0 - m10 - xh0[k][e] * m20);
                o1[e] = rstd1 * (gm1[k][e] * g1 - m11 - xh1[k][e] * m21);
            }
            if (a.planar) st_planar4(dy, rr, a.T, i4, o0, o1);
            else st_pairs4(dy, rr, a.T, i4, o0, o1);
        }
    }
}

// --------------------------------------------------------------------------
// glue kernels at the fp32 <-> pair seams
// --------------------------------------------------------------------------
// dst[b, p, t] = (bf16 x[b, 2p, t], bf16 x[b, 2p + 1, t]); x fp32 with explicit element strides (the transposed [B, T, M] view of
// data_utils.py:14-16 included), dst dword rows with strides (db, dc)
__global__ void __launch_bounds__(AVC_THREADS)
to_pairs_kernel(const float* x, long sxb, long sxc, long sxt, int B, int C2, int T, unsigned* dst, long db, long dc) {
    const long n = (long)B * C2 * T;
    for (long e = (long)blockIdx.x * AVC_THREADS + threadIdx.x; e < n; e += (long)gridDim.x * AVC_THREADS) {
        const int t = (int)(e % T);
        const long r = e / T;
        const int p = (int)(r % C2), b = (int)(r / C2);
        const float* s = x + (long)b * sxb + (long)(2 * p) * sxc + (long)t * sxt;
        dst[(long)b * db + (long)p * dc + t] = bh_pack(s[0], s[sxc]);
    }
}

// z = mu + exp(log_sigma / 2) * eps (model.py:383-384): muls fp32 [B, 2C, Tb] -> z pairs [B][C/2][Tb]
__global__ void __launch_bounds__(AVC_THREADS) reparam_fwd_pairs_kernel(const float* muls, const float* eps, int B, int C, int Tb, unsigned* z) {
    const int C2 = C >> 1;
    const long n = (long)B * C2 * Tb, per = (long)C * Tb;
    for (long e = (long)blockIdx.x * AVC_THREADS + threadIdx.x; e < n; e += (long)gridDim.x * AVC_THREADS) {
        const int t = (int)(e % Tb);
        const long r = e / Tb;
        const int p = (int)(r % C2), b = (int)(r / C2);
        float v[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const long i = (long)(2 * p + u) * Tb + t;
            const float mu = muls[(long)b * 2 * per + i], ls = muls[(long)b * 2 * per + per + i];
            v[u] = eps ? mu + expf(ls * 0.5f) * eps[(long)b * per + i] : mu;
        }
        z[e] = bh_pack(v[0], v[1]);
    }
}

// AdaptiveAvgPool1d(1) (model.py:273) of pair rows -> out[c * B + b] fp32 (channel-major for the dense stack)
__global__ void __launch_bounds__(AVC_THREADS) timepool_fwd_pairs_kernel(const unsigned* in, int B, int C, int T, float* out) {
    const int r = blockIdx.x * AVC_THREADS + threadIdx.x;
    const int C2 = C >> 1;
    if (r >= B * C2) return;
    const unsigned* pr = in + (long)r * T;
    float s0 = 0.f, s1 = 0.f;
    for (int t = 0; t < T; ++t) {
        const unsigned d = pr[t];
        s0 += bh_lo(d);
        s1 += bh_hi(d);
    }
    const int b = r / C2, p = r - b * C2;
    out[(long)(2 * p) * B + b] = s0 / (float)T;
    out[(long)(2 * p + 1) * B + b] = s1 / (float)T;
}

// backward of the pooling + ReLU mask of the producing block: G = dP / T ; dy = G * (a > 0), both pair rows
__global__ void __launch_bounds__(AVC_THREADS)
timepool_bwd_pairs_kernel(const float* dP, const unsigned* amask, int B, int C, int T, unsigned* G, unsigned* dy, float slope) {
    const int C2 = C >> 1;
    const long n = (long)B * C2 * T;
    for (long e = (long)blockIdx.x * AVC_THREADS + threadIdx.x; e < n; e += (long)gridDim.x * AVC_THREADS) {
        const long r = e / T;
        const int b = (int)(r / C2), p = (int)(r - (long)b * C2);
        const float g0 = dP[(long)(2 * p) * B + b] / (float)T, g1 = dP[(long)(2 * p + 1) * B + b] / (float)T;
        if (G) G[e] = bh_pack(g0, g1);
        if (dy) {
            const unsigned m = amask[e];
            dy[e] = bh_pack(avc_act_grad(g0, bh_lo(m) > 0.f, slope), avc_act_grad(g1, bh_hi(m) > 0.f, slope));
        }
    }
}

// --------------------------------------------------------------------------
// launchers
// --------------------------------------------------------------------------
template <int LPR, int NV>
static void launch_fwd(const INFwdArgs& a, hipStream_t s) {
    hipLaunchKernelGGL((instnorm_fwd_pairs_kernel<LPR, NV>), dim3(avc_cdiv(a.R, AVC_THREADS / LPR)), dim3(AVC_THREADS), 0, s, a);
}
template <int LPR, int NV>
static void launch_bwd(const INBwdArgs& a, hipStream_t s) {
    hipLaunchKernelGGL((instnorm_bwd_pairs_kernel<LPR, NV>), dim3(avc_cdiv(a.R, AVC_THREADS / LPR)), dim3(AVC_THREADS), 0, s, a);
}

// a.R = B * C / 2.  -2: shape outside the pair kernels (T % 4, odd C, rows longer than 2048 frames, ceil-mode pooling of an odd row)
int avc_launch_in_fwd_pairs(const INFwdArgs& a, hipStream_t s) {
    const int n4 = a.T >> 2;
    if ((a.T & 3) || (a.C & 1) || n4 > 512) return -2;
    if (a.res && ((a.res_mode == AVC_RES_AVGPOOL2 && a.Tres != 2 * a.T) || (a.res_mode == AVC_RES_UP2 && 2 * a.Tres != a.T) ||
                  (a.res_mode == AVC_RES_IDENTITY && a.Tres != a.T)))
        return -2;
    ProfScope ps(AVC_K_IN_FWD, 0.0, 2.0 * 4.0 * (double)a.R * a.T, s);   // 1 read + 1 write of 2 bytes x 2 channels per dword
    // lanes per row x 16-byte vectors per lane: nv (tuning in_pairs_nv) vectors per lane while the row still fills >= 4 lanes
    int nv = a.nv_hint == 2 || a.nv_hint == 4 ? a.nv_hint : 1;
    while (nv > 1 && (n4 / nv < 4 || n4 % nv)) nv >>= 1;
    const int lpr = n4 / nv;
#define AVC_INP(L_, N_) launch_fwd<L_, N_>(a, s)
    if (nv == 1 || (lpr & (lpr - 1)) || lpr > 64) {
        if (n4 <= 4) AVC_INP(4, 1);
        else if (n4 <= 8) AVC_INP(8, 1);
        else if (n4 <= 16) AVC_INP(16, 1);
        else if (n4 <= 32) AVC_INP(32, 1);
        else if (n4 <= 64) AVC_INP(64, 1);
        else if (n4 <= 128) AVC_INP(64, 2);
        else if (n4 <= 256) AVC_INP(64, 4);
        else AVC_INP(64, 8);
    } else if (nv == 2) {
        if (lpr == 4) AVC_INP(4, 2); else if (lpr == 8) AVC_INP(8, 2); else if (lpr == 16) AVC_INP(16, 2); else if (lpr == 32) AVC_INP(32, 2); else AVC_INP(64, 2);
    } else {
        if (lpr == 4) AVC_INP(4, 4); else if (lpr == 8) AVC_INP(8, 4); else if (lpr == 16) AVC_INP(16, 4); else if (lpr == 32) AVC_INP(32, 4); else AVC_INP(64, 4);
    }
#undef AVC_INP
    return (int)hipGetLastError();
}
int avc_launch_in_bwd_pairs(const INBwdArgs& a, hipStream_t s) {
    const int n4 = a.T >> 2;
    if ((a.T & 3) || (a.C & 1) || n4 > 512) return -2;
    ProfScope ps(AVC_K_IN_BWD, 0.0, 3.0 * 4.0 * (double)a.R * a.T, s);
    // lanes per row x 16-byte vectors per lane: nv (tuning in_pairs_nv) vectors per lane while the row still fills >= 4 lanes
    int nv = a.nv_hint == 2 || a.nv_hint == 4 ? a.nv_hint : 1;
    while (nv > 1 && (n4 / nv < 4 || n4 % nv)) nv >>= 1;
    const int lpr = n4 / nv;
#define AVC_INP(L_, N_) launch_bwd<L_, N_>(a, s)
    if (nv == 1 || (lpr & (lpr - 1)) || lpr > 64) {
        if (n4 <= 4) AVC_INP(4, 1);
        else if (n4 <= 8) AVC_INP(8, 1);
        else if (n4 <= 16) AVC_INP(16, 1);
        else if (n4 <= 32) AVC_INP(32, 1);
        else if (n4 <= 64) AVC_INP(64, 1);
        else if (n4 <= 128) AVC_INP(64, 2);
        else if (n4 <= 256) AVC_INP(64, 4);
        else AVC_INP(64, 8);
    } else if (nv == 2) {
        if (lpr == 4) AVC_INP(4, 2); else if (lpr == 8) AVC_INP(8, 2); else if (lpr == 16) AVC_INP(16, 2); else if (lpr == 32) AVC_INP(32, 2); else AVC_INP(64, 2);
    } else {
        if (lpr == 4) AVC_INP(4, 4); else if (lpr == 8) AVC_INP(8, 4); else if (lpr == 16) AVC_INP(16, 4); else if (lpr == 32) AVC_INP(32, 4); else AVC_INP(64, 4);
    }
#undef AVC_INP
    return (int)hipGetLastError();
}

static int ew_blocks(long n) {
    long b = (n + AVC_THREADS - 1) / AVC_THREADS;
    if (b < 1) b = 1;
    if (b > 2048) b = 2048;
    return (int)b;
}
int avc_launch_to_pairs(const float* x, long sxb, long sxc, long sxt, int B, int C, int T, float* dst, long db, long dc, hipStream_t s) {
    if (C & 1) return -2;
    ProfScope ps(AVC_K_MISC, 0.0, 0.0, s);
    hipLaunchKernelGGL(to_pairs_kernel, dim3(ew_blocks((long)B * (C / 2) * T)), dim3(AVC_THREADS), 0, s, x, sxb, sxc, sxt, B, C / 2, T, (unsigned*)dst, db, dc);
    return (int)hipGetLastError();
}
int avc_launch_reparam_fwd_pairs(const float* muls, const float* eps, int B, int C, int Tb, float* z, hipStream_t s) {
    if (C & 1) return -2;
    ProfScope ps(AVC_K_MISC, 0.0, 0.0, s);
    hipLaunchKernelGGL(reparam_fwd_pairs_kernel, dim3(ew_blocks((long)B * (C / 2) * Tb)), dim3(AVC_THREADS), 0, s, muls, eps, B, C, Tb, (unsigned*)z);
    return (int)hipGetLastError();
}
int avc_launch_timepool_fwd_pairs(const float* in, int B, int C, int T, float* out, hipStream_t s) {
    if (C & 1) return -2;
    ProfScope ps(AVC_K_MISC, 0.0, 0.0, s);
    hipLaunchKernelGGL(timepool_fwd_pairs_kernel, dim3(avc_cdiv(B * (C / 2), AVC_THREADS)), dim3(AVC_THREADS), 0, s, (const unsigned*)in, B, C, T, out);
    return (int)hipGetLastError();
}
int avc_launch_timepool_bwd_pairs(const float* dP, const float* amask, int B, int C, int T, float* G, float* dy, float slope, hipStream_t s) {
    if (C & 1) return -2;
    ProfScope ps(AVC_K_MISC, 0.0, 0.0, s);
    hipLaunchKernelGGL(timepool_bwd_pairs_kernel, dim3(ew_blocks((long)B * (C / 2) * T)), dim3(AVC_THREADS), 0, s, dP, (const unsigned*)amask, B, C, T,
                       (unsigned*)G, (unsigned*)dy, slope);
    return (int)hipGetLastError();
}
