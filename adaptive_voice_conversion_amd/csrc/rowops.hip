// HBM-bound row kernels of the AdaIN-VC path (gfx950).
//
// InstanceNorm1d(affine=False) + AdaIN affine + ReLU + residual join, forward
// and backward, as single-pass register-resident row kernels:
//   reference: nn.InstanceNorm1d (model.py:296,341), append_cond (model.py:77-83),
//   the block bodies model.py:309-320 / :353-369 and their autograd.
// A row is one (b, c) instance of T contiguous fp32 values.  LPR lanes of a
// wavefront share a row (LPR = 4..64 so that T=16 bottleneck rows still fill
// the wave), each lane holds NV float4 in registers between the statistics
// pass and the normalise pass: forward = 1 read + 1 write of the row, backward
// = 2 reads + 1 write (SURVEY.md App. B traffic model).  Reductions are
// xor-shuffles inside the LPR group.
#include <hip/hip_runtime.h>
#include <stdlib.h>

#include "avc_common.h"
#include "avc_internal.h"

// (in_xhat / in_preact / AVC_IN_EPS: avc_common.h, shared with the fused conv epilogue)

template <int LPR>
static __device__ __forceinline__ float group_sum(float v) {
#pragma unroll
    for (int o = LPR / 2; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

static __device__ __forceinline__ float4 ld4(const float* p, bool aligned) {
    if (aligned) return *(const float4*)p;
    return make_float4(p[0], p[1], p[2], p[3]);
}

// residual value for output positions t..t+3 of a row (t % 4 == 0, row bases 16-B aligned)
static __device__ __forceinline__ float4 res4(const float* rrow, int mode, int t, int Tres) {
    float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
    if (mode == AVC_RES_IDENTITY) {
        r = *(const float4*)(rrow + t);  // Tres == T, T % 4 == 0
    } else if (mode == AVC_RES_UP2) {    // nearest x2 (model.py:61-63); Tres = T/2 is even
        float2 ab = *(const float2*)(rrow + (t >> 1));
        r = make_float4(ab.x, ab.x, ab.y, ab.y);
    } else if (mode == AVC_RES_AVGPOOL2) {  // ceil-mode avg-pool (model.py:319)
        if ((Tres & 3) == 0) {              // Tres == 2T: two aligned vector loads
            float4 p = *(const float4*)(rrow + 2 * t), q = *(const float4*)(rrow + 2 * t + 4);
            r = make_float4((p.x + p.y) * 0.5f, (p.z + p.w) * 0.5f, (q.x + q.y) * 0.5f, (q.z + q.w) * 0.5f);
        } else {
            float o[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                int i0 = 2 * (t + k), i1 = i0 + 1;
                float v0 = rrow[i0];
                o[k] = (i1 < Tres) ? (v0 + rrow[i1]) * 0.5f : v0;
            }
            r = make_float4(o[0], o[1], o[2], o[3]);
        }
    }
    return r;
}

template <int LPR, int NV>
__global__ void __launch_bounds__(AVC_THREADS) instnorm_fwd_kernel(const INFwdArgs a) {
    constexpr int RPB = AVC_THREADS / LPR;
    const int tid = threadIdx.x;
    const int row = blockIdx.x * RPB + tid / LPR;
    const int l = tid % LPR;
    const bool rvalid = row < a.R;
    const int rr = rvalid ? row : 0;
    const int n4 = a.T >> 2;
    const float* yrow = a.y + (long)rr * a.T;
    float4 v[NV], rv[NV];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        int i4 = k * LPR + l;
        v[k] = (i4 < n4) ? *(const float4*)(yrow + 4 * i4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    // every other load of the row -- the AdaIN scale / shift and the residual -- goes out NOW, behind the row itself: written where they
    // are used (after the two reductions) they are a second and a third dependent memory round trip of a 5-8 us kernel
    float gamma = 1.f, beta = 0.f;
    const int b = rr / a.C, c = rr - b * a.C;
    if (a.cond) {
        const float* cr = a.cond + (long)b * a.cond_sb + a.cond_off;
        beta = cr[c];          // first half = shift   (model.py:81)
        gamma = cr[a.C + c];   // second half = scale
    }
    const float* rrow = a.res ? a.res + (long)rr * a.Tres : nullptr;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        int i4 = k * LPR + l;
        rv[k] = (rrow && i4 < n4) ? res4(rrow, a.res_mode, 4 * i4, a.Tres) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int k = 0; k < NV; ++k) s += (v[k].x + v[k].y) + (v[k].z + v[k].w);
    const float invT = 1.0f / (float)a.T;
    const float mean = group_sum<LPR>(s) * invT;
    float ss = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        int i4 = k * LPR + l;
        if (i4 < n4) {
            float dx = v[k].x - mean, dy = v[k].y - mean, dz = v[k].z - mean, dw = v[k].w - mean;
            ss += (dx * dx + dy * dy) + (dz * dz + dw * dw);
        }
    }
    const float var = group_sum<LPR>(ss) * invT;  // biased variance
    const float rstd = 1.0f / sqrtf(var + AVC_IN_EPS);
    if (rvalid && l == 0) {
        a.mean[row] = mean;
        a.rstd[row] = rstd;
    }
    if (!rvalid) return;
    float* orow = a.out + (long)row * a.T;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        int i4 = k * LPR + l;
        if (i4 < n4) {
            float o[4] = {v[k].x, v[k].y, v[k].z, v[k].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float w = in_preact(in_xhat(o[e], mean, rstd), gamma, beta);
                o[e] = a.relu ? avc_act(w, a.slope) : w;
            }
            if (rrow) {
                const float4 r = rv[k];
                o[0] += r.x; o[1] += r.y; o[2] += r.z; o[3] += r.w;
            }
            *(float4*)(orow + 4 * i4) = make_float4(o[0], o[1], o[2], o[3]);
        }
    }
}

// any T (odd inference lengths): one wavefront per row, three cached passes
__global__ void __launch_bounds__(AVC_THREADS) instnorm_fwd_generic_kernel(const INFwdArgs a) {
    const int tid = threadIdx.x, l = tid & 63;
    const int row = blockIdx.x * 4 + (tid >> 6);
    const bool rvalid = row < a.R;
    const int rr = rvalid ? row : 0;
    const float* yrow = a.y + (long)rr * a.T;
    float s = 0.f;
    for (int t = l; t < a.T; t += 64) s += yrow[t];
    const float invT = 1.0f / (float)a.T;
    const float mean = group_sum<64>(s) * invT;
    float ss = 0.f;
    for (int t = l; t < a.T; t += 64) {
        float d = yrow[t] - mean;
        ss += d * d;
    }
    const float var = group_sum<64>(ss) * invT;
    const float rstd = 1.0f / sqrtf(var + AVC_IN_EPS);
    if (!rvalid) return;
    float gamma = 1.f, beta = 0.f;
    const int b = rr / a.C, c = rr - b * a.C;
    if (a.cond) {
        const float* cr = a.cond + (long)b * a.cond_sb + a.cond_off;
        beta = cr[c];
        gamma = cr[a.C + c];
    }
    if (l == 0) {
        a.mean[row] = mean;
        a.rstd[row] = rstd;
    }
    float* orow = a.out + (long)row * a.T;
    const float* rrow = a.res ? a.res + (long)row * a.Tres : nullptr;
    for (int t = l; t < a.T; t += 64) {
        float w = in_preact(in_xhat(yrow[t], mean, rstd), gamma, beta);
        w = a.relu ? avc_act(w, a.slope) : w;
        if (rrow) {
            if (a.res_mode == AVC_RES_IDENTITY)
                w += rrow[t];
            else if (a.res_mode == AVC_RES_UP2)
                w += rrow[t >> 1];
            else if (a.res_mode == AVC_RES_AVGPOOL2) {
                int i0 = 2 * t, i1 = i0 + 1;
                w += (i1 < a.Tres) ? (rrow[i0] + rrow[i1]) * 0.5f : rrow[i0];
            }
        }
        orow[t] = w;
    }
}

// backward:  g = dL/d(out);  out = relu(xh*gamma+beta) [+ res]
//   gm = g * 1[xh*gamma+beta > 0];  dbeta = sum gm;  dgamma = sum gm*xh
//   dy = rstd * (gm*gamma - mean_T(gm*gamma) - xh * mean_T(gm*gamma*xh))
template <int LPR, int NV>
__global__ void __launch_bounds__(AVC_THREADS) instnorm_bwd_kernel(const INBwdArgs a) {
    constexpr int RPB = AVC_THREADS / LPR;
    const int tid = threadIdx.x;
    const int row = blockIdx.x * RPB + tid / LPR;
    const int l = tid % LPR;
    const bool rvalid = row < a.R;
    const int rr = rvalid ? row : 0;
    const int n4 = a.T >> 2;
    const float* yrow = a.y + (long)rr * a.T;
    const float* grow = a.g + (long)rr * a.T;
    const float mean = a.mean[rr], rstd = a.rstd[rr];
    float gamma = 1.f, beta = 0.f;
    const int b = rr / a.C, c = rr - b * a.C;
    if (a.cond) {
        const float* cr = a.cond + (long)b * a.cond_sb + a.cond_off;
        beta = cr[c];
        gamma = cr[a.C + c];
    }
    float4 xh[NV], gm[NV];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        int i4 = k * LPR + l;
        if (i4 < n4) {
            float4 yv = *(const float4*)(yrow + 4 * i4);
            float4 gv = *(const float4*)(grow + 4 * i4);
            float xx[4] = {yv.x, yv.y, yv.z, yv.w}, gg[4] = {gv.x, gv.y, gv.z, gv.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float h = in_xhat(xx[e], mean, rstd);
                float w = in_preact(h, gamma, beta);
                float gme = avc_act_grad(gg[e], !a.relu || w > 0.f, a.slope);
                xx[e] = h;
                gg[e] = gme;
                s1 += gme;
                s2 += gme * h;
            }
            xh[k] = make_float4(xx[0], xx[1], xx[2], xx[3]);
            gm[k] = make_float4(gg[0], gg[1], gg[2], gg[3]);
        } else {
            xh[k] = make_float4(0.f, 0.f, 0.f, 0.f);
            gm[k] = xh[k];
        }
    }
    s1 = group_sum<LPR>(s1);  // dbeta
    s2 = group_sum<LPR>(s2);  // dgamma
    if (!rvalid) return;
    if (a.dcond && l == 0) {
        float* dc = a.dcond + (long)b * a.dcond_sb + a.dcond_off;
        dc[c] = s1;
        dc[a.C + c] = s2;
    }
    const float invT = 1.0f / (float)a.T;
    const float m1 = gamma * s1 * invT, m2 = gamma * s2 * invT;
    float* drow = a.dy + (long)row * a.T;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        int i4 = k * LPR + l;
        if (i4 < n4) {
            float4 o;
            o.x = rstd * (gm[k].x * gamma - m1 - xh[k].x * m2);
            o.y = rstd * (gm[k].y * gamma - m1 - xh[k].y * m2);
            o.z = rstd * (gm[k].z * gamma - m1 - xh[k].z * m2);
            o.w = rstd * (gm[k].w * gamma - m1 - xh[k].w * m2);
            *(float4*)(drow + 4 * i4) = o;
        }
    }
}

__global__ void __launch_bounds__(AVC_THREADS) instnorm_bwd_generic_kernel(const INBwdArgs a) {
    const int tid = threadIdx.x, l = tid & 63;
    const int row = blockIdx.x * 4 + (tid >> 6);
    const bool rvalid = row < a.R;
    const int rr = rvalid ? row : 0;
    const float* yrow = a.y + (long)rr * a.T;
    const float* grow = a.g + (long)rr * a.T;
    const float mean = a.mean[rr], rstd = a.rstd[rr];
    float gamma = 1.f, beta = 0.f;
    const int b = rr / a.C, c = rr - b * a.C;
    if (a.cond) {
        const float* cr = a.cond + (long)b * a.cond_sb + a.cond_off;
        beta = cr[c];
        gamma = cr[a.C + c];
    }
    float s1 = 0.f, s2 = 0.f;
    for (int t = l; t < a.T; t += 64) {
        float h = in_xhat(yrow[t], mean, rstd);
        float w = in_preact(h, gamma, beta);
        float gme = avc_act_grad(grow[t], !a.relu || w > 0.f, a.slope);
        s1 += gme;
        s2 += gme * h;
    }
    s1 = group_sum<64>(s1);
    s2 = group_sum<64>(s2);
    if (!rvalid) return;
    if (a.dcond && l == 0) {
        float* dc = a.dcond + (long)b * a.dcond_sb + a.dcond_off;
        dc[c] = s1;
        dc[a.C + c] = s2;
    }
    const float invT = 1.0f / (float)a.T;
    const float m1 = gamma * s1 * invT, m2 = gamma * s2 * invT;
    float* drow = a.dy + (long)row * a.T;
    for (int t = l; t < a.T; t += 64) {
        float h = in_xhat(yrow[t], mean, rstd);
        float w = in_preact(h, gamma, beta);
        float gme = avc_act_grad(grow[t], !a.relu || w > 0.f, a.slope);
        drow[t] = rstd * (gme * gamma - m1 - h * m2);
    }
}

// --------------------------------------------------------------------------
// small glue kernels
// --------------------------------------------------------------------------
// dst[b, coff + m, t] = x[b, m, t]   (x may be the transposed view of data_utils.py:14-16)
// VEC: rows of T floats with T % 4 == 0, unit time stride and 16-byte aligned bases / strides on both sides: one float4 per thread
// and 32-bit index math (the scalar loop with its 64-bit divisions ran at 0.75 TB/s: 28 us for the 80 x 128 x 256 input of an encoder)
template <bool VEC>
__global__ void __launch_bounds__(AVC_THREADS)
copy_rows_kernel(const float* x, long sxb, long sxc, int sxt, int B, int M, int T, float* dst, long db, long dc) {
    if constexpr (VEC) {
        const int T4 = T >> 2;
        const int n4 = B * M * T4;
        for (int e = blockIdx.x * AVC_THREADS + threadIdx.x; e < n4; e += gridDim.x * AVC_THREADS) {
            const int t4 = e % T4, r = e / T4;
            const int m = r % M, b = r / M;
            *(float4*)(dst + (long)b * db + (long)m * dc + 4 * t4) = *(const float4*)(x + (long)b * sxb + (long)m * sxc + 4 * t4);
        }
        return;
    }
    long n = (long)B * M * T;
    for (long e = (long)blockIdx.x * AVC_THREADS + threadIdx.x; e < n; e += (long)gridDim.x * AVC_THREADS) {
        int t = (int)(e % T);
        long r = e / T;
        int m = (int)(r % M);
        int b = (int)(r / M);
        dst[(long)b * db + (long)m * dc + t] = x[(long)b * sxb + (long)m * sxc + (long)t * sxt];
    }
}

// Device-side data feed (reference: PickleDataset.__getitem__ + CollateFn, data_utils.py:10-22,
// 51-54): segment b = rows [start[b], start[b]+T) of the HBM-resident corpus [sum_T][M] (mel bins contiguous),
// emitted as out[b][m][t] with t contiguous -- the layout every first-layer loader reads with unit stride.
// One workgroup = one sample x 32 frames x (up to) 128 mel bins: coalesced row reads -> LDS tile (odd row
// pitch) -> coalesced 128-byte time runs per mel bin.
#define AVC_GATHER_MB 128
__global__ void __launch_bounds__(AVC_THREADS)
gather_segments_kernel(const float* corpus, long n_rows, int M, const long* starts, int T, float* out) {
    __shared__ float tile[32 * (AVC_GATHER_MB + 1)];
    const int b = blockIdx.y, t0 = blockIdx.x * 32, m0 = blockIdx.z * AVC_GATHER_MB;
    const int mb = (M - m0) < AVC_GATHER_MB ? (M - m0) : AVC_GATHER_MB;
    constexpr int pitch = AVC_GATHER_MB + 1;
    const long r0 = starts[b] + t0;
    const int nt = (T - t0) < 32 ? (T - t0) : 32;
    for (int e = threadIdx.x; e < 32 * mb; e += AVC_THREADS) {
        const int row = e / mb, col = e - row * mb;
        long r = r0 + row;
        r = r < n_rows ? r : n_rows - 1;  // (rows past the segment / corpus end are never stored)
        tile[row * pitch + col] = (row < nt) ? corpus[r * M + m0 + col] : 0.f;
    }
    __syncthreads();
    float* ob = out + ((long)b * M + m0) * T + t0;
    for (int e = threadIdx.x; e < 32 * mb; e += AVC_THREADS) {
        const int m = e >> 5, t = e & 31;
        if (t < nt) ob[(long)m * T + t] = tile[t * pitch + m];
    }
}

// dst[c][b] += src[b][c]   (upstream d(emb) of the autograd seam joins the channel-major gradient)
__global__ void __launch_bounds__(AVC_THREADS) add_transposed_kernel(float* dst, const float* src, int B, int C) {
    int e = blockIdx.x * AVC_THREADS + threadIdx.x;
    if (e >= B * C) return;
    int c = e / B, b = e - c * B;
    dst[e] += src[(long)b * C + c];
}

// AdaptiveAvgPool1d(1) (model.py:273): in [B,C,T] -> out[c*B + b]  (channel-major for the dense stack)
__global__ void __launch_bounds__(AVC_THREADS) timepool_fwd_kernel(const float* in, int B, int C, int T, float* out) {
    int r = blockIdx.x * AVC_THREADS + threadIdx.x;
    if (r >= B * C) return;
    const float* p = in + (long)r * T;
    float s = 0.f;
    for (int t = 0; t < T; ++t) s += p[t];
    int b = r / C, c = r - b * C;
    out[(long)c * B + b] = s / (float)T;
}

// backward of the pooling + ReLU mask of the producing block: G = dP/T ; dy = G * (a > 0)
__global__ void __launch_bounds__(AVC_THREADS)
timepool_bwd_kernel(const float* dP, const float* amask, int B, int C, int T, float* G, float* dy, float slope) {
    long n = (long)B * C * T;
    for (long e = (long)blockIdx.x * AVC_THREADS + threadIdx.x; e < n; e += (long)gridDim.x * AVC_THREADS) {
        long r = e / T;
        int b = (int)(r / C), c = (int)(r - (long)b * C);
        float gv = dP[(long)c * B + b] / (float)T;
        if (G) G[e] = gv;
        if (dy) dy[e] = avc_act_grad(gv, amask[e] > 0.f, slope);
    }
}

// z = mu + exp(log_sigma/2) * eps   (model.py:383-384); muls = [B, 2C, Tb] (mu | log_sigma)
__global__ void __launch_bounds__(AVC_THREADS)
reparam_fwd_kernel(const float* muls, const float* eps, int B, int C, int Tb, float* z) {
    long n = (long)B * C * Tb;
    for (long e = (long)blockIdx.x * AVC_THREADS + threadIdx.x; e < n; e += (long)gridDim.x * AVC_THREADS) {
        long per = (long)C * Tb;
        int b = (int)(e / per);
        long i = e - (long)b * per;
        float mu = muls[(long)b * 2 * per + i], ls = muls[(long)b * 2 * per + per + i];
        z[e] = eps ? mu + expf(ls * 0.5f) * eps[e] : mu;
    }
}

// d(muls) from the KL term (solver.py:86) and from z (model.py:384):
//   dmu = lkl*mu/N + dz ;  dls = lkl*0.5*(exp(ls)-1)/N + dz*eps*0.5*exp(ls/2)
__global__ void __launch_bounds__(AVC_THREADS)
latent_bwd_kernel(const float* muls, const float* eps, const float* dz, const float* dmuls_up, int B, int C, int Tb,
                  float lambda_kl_over_n, float* dmuls) {
    long n = (long)B * C * Tb;
    for (long e = (long)blockIdx.x * AVC_THREADS + threadIdx.x; e < n; e += (long)gridDim.x * AVC_THREADS) {
        long per = (long)C * Tb;
        int b = (int)(e / per);
        long i = e - (long)b * per;
        long imu = (long)b * 2 * per + i, ils = imu + per;
        float mu = muls[imu], ls = muls[ils];
        float dzv = dz ? dz[e] : 0.f;
        float ev = eps ? eps[e] : 0.f;
        float dmu = lambda_kl_over_n * mu + dzv;
        float dls = lambda_kl_over_n * 0.5f * (expf(ls) - 1.0f) + dzv * ev * 0.5f * expf(ls * 0.5f);
        if (dmuls_up) {
            dmu += dmuls_up[imu];
            dls += dmuls_up[ils];
        }
        dmuls[imu] = dmu;
        dmuls[ils] = dls;
    }
}

static __device__ __forceinline__ float block_sum(float v, float* red) {
    v = group_sum<64>(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

// L1 + KL partial sums (solver.py:84-86) and d(dec) = scale * sign(dec - x)
// VEC: x has unit time stride, T % 4 == 0 and 16-byte aligned rows on every side: float4 per thread, 32-bit index math
template <bool VEC>
__global__ void __launch_bounds__(AVC_THREADS)
loss_partial_kernel(const float* dec, const float* x, long sxb, long sxc, int sxt, int B, int M, int T,
                    const float* muls, int C, int Tb, float ddec_scale, float* ddec, float* partial) {
    __shared__ float red[4];
    long n = (long)B * M * T;
    float s = 0.f;
    if constexpr (VEC) {
        const int T4 = T >> 2;
        const int n4 = (int)(n >> 2);
        for (int e = blockIdx.x * AVC_THREADS + threadIdx.x; e < n4; e += gridDim.x * AVC_THREADS) {
            const int t4 = e % T4, r = e / T4;
            const int m = r % M, b = r / M;
            const float4 dv = *(const float4*)(dec + 4L * e);
            const float4 xv = *(const float4*)(x + (long)b * sxb + (long)m * sxc + 4 * t4);
            const float d0 = dv.x - xv.x, d1 = dv.y - xv.y, d2 = dv.z - xv.z, d3 = dv.w - xv.w;
            s += (fabsf(d0) + fabsf(d1)) + (fabsf(d2) + fabsf(d3));
            if (ddec) {
                float4 g;
                g.x = (d0 > 0.f) ? ddec_scale : ((d0 < 0.f) ? -ddec_scale : 0.f);
                g.y = (d1 > 0.f) ? ddec_scale : ((d1 < 0.f) ? -ddec_scale : 0.f);
                g.z = (d2 > 0.f) ? ddec_scale : ((d2 < 0.f) ? -ddec_scale : 0.f);
                g.w = (d3 > 0.f) ? ddec_scale : ((d3 < 0.f) ? -ddec_scale : 0.f);
                *(float4*)(ddec + 4L * e) = g;
            }
        }
    } else {
        for (long e = (long)blockIdx.x * AVC_THREADS + threadIdx.x; e < n; e += (long)gridDim.x * AVC_THREADS) {
            int t = (int)(e % T);
            long r = e / T;
            int m = (int)(r % M);
            int b = (int)(r / M);
            float d = dec[e] - x[(long)b * sxb + (long)m * sxc + (long)t * sxt];
            s += fabsf(d);
            if (ddec) ddec[e] = (d > 0.f) ? ddec_scale : ((d < 0.f) ? -ddec_scale : 0.f);
        }
    }
    float k = 0.f;
    long nk = (long)B * C * Tb, per = (long)C * Tb;
    for (long e = (long)blockIdx.x * AVC_THREADS + threadIdx.x; e < nk; e += (long)gridDim.x * AVC_THREADS) {
        int b = (int)(e / per);
        long i = e - (long)b * per;
        float mu = muls[(long)b * 2 * per + i], ls = muls[(long)b * 2 * per + per + i];
        k += expf(ls) + mu * mu - 1.0f - ls;
    }
    float bs = block_sum(s, red);
    float bk = block_sum(k, red);
    if (threadIdx.x == 0) {
        partial[2 * blockIdx.x] = bs;
        partial[2 * blockIdx.x + 1] = bk;
    }
}

__global__ void __launch_bounds__(AVC_THREADS)
loss_final_kernel(const float* partial, int nblocks, float inv_n_rec, float half_inv_n_kl, float* losses) {
    __shared__ float red[4];
    float s = 0.f, k = 0.f;
    for (int i = threadIdx.x; i < nblocks; i += AVC_THREADS) {
        s += partial[2 * i];
        k += partial[2 * i + 1];
    }
    float ts = block_sum(s, red);
    float tk = block_sum(k, red);
    if (threadIdx.x == 0) {
        losses[0] = ts * inv_n_rec;        // loss_rec = mean |dec - x|
        losses[1] = tk * half_inv_n_kl;    // loss_kl = 0.5 * mean(...)
    }
}

// --------------------------------------------------------------------------
// optimizer: global-norm clip (solver.py:91-92) + Adam/amsgrad with coupled L2
// weight decay (solver.py:75-77, torch.optim.Adam) over the flat buffers
// --------------------------------------------------------------------------
__global__ void __launch_bounds__(AVC_THREADS) sumsq_partial_kernel(const float* g, long n, float* partial) {
    __shared__ float red[4];
    float s = 0.f;
    for (long e = (long)blockIdx.x * AVC_THREADS + threadIdx.x; e < n; e += (long)gridDim.x * AVC_THREADS) {
        float v = g[e];
        s += v * v;
    }
    float bs = block_sum(s, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = bs;
}

__global__ void __launch_bounds__(AVC_THREADS) clip_adam_kernel(const AdamArgs a) {
    __shared__ float red[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < a.npartial; i += AVC_THREADS) s += a.partial[i];
    float total = block_sum(s, red);
    float gnorm = sqrtf(total) * a.grad_prescale;   // grads are (sum over ranks) * grad_prescale
    float coef = a.max_norm / (gnorm + 1e-6f);      // clip_grad_norm_: min(1, max_norm/(norm+1e-6))
    coef = fminf(coef, 1.0f);
    if (a.max_norm <= 0.f) coef = 1.0f;
    coef *= a.grad_prescale;
    if (blockIdx.x == 0 && threadIdx.x == 0 && a.gnorm_out) a.gnorm_out[0] = gnorm;
    // 16-byte path (the flat buffers of a plan: every tensor padded to four floats, 256-byte aligned allocations): the scalar loop below
    // is one dependent memory round trip per ELEMENT and thread (19 of them at 4.9 M parameters), the same arithmetic on four elements
    // per trip runs at the HBM rate
    const bool vec = ((a.n & 3) == 0) && ((((size_t)a.p | (size_t)a.g | (size_t)a.m | (size_t)a.v | (size_t)a.vmax) & 15) == 0);
    if (vec) {
        const long n4 = a.n >> 2;
        for (long e = (long)blockIdx.x * AVC_THREADS + threadIdx.x; e < n4; e += (long)gridDim.x * AVC_THREADS) {
            const float4 p4 = ((const float4*)a.p)[e], g4 = ((const float4*)a.g)[e], m4 = ((const float4*)a.m)[e], v4 = ((const float4*)a.v)[e];
            const float4 x4 = a.amsgrad ? ((const float4*)a.vmax)[e] : make_float4(0.f, 0.f, 0.f, 0.f);
            float pp[4] = {p4.x, p4.y, p4.z, p4.w}, gg[4] = {g4.x, g4.y, g4.z, g4.w}, mm[4] = {m4.x, m4.y, m4.z, m4.w};
            float vv[4] = {v4.x, v4.y, v4.z, v4.w}, xx[4] = {x4.x, x4.y, x4.z, x4.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float g = gg[k] * coef;
                gg[k] = g;
                g = g + a.weight_decay * pp[k];
                const float m = a.beta1 * mm[k] + (1.0f - a.beta1) * g;
                const float v = a.beta2 * vv[k] + (1.0f - a.beta2) * g * g;
                float denom;
                if (a.amsgrad) {
                    const float vm = fmaxf(xx[k], v);
                    xx[k] = vm;
                    denom = sqrtf(vm) / a.sqrt_bc2 + a.eps;
                } else {
                    denom = sqrtf(v) / a.sqrt_bc2 + a.eps;
                }
                mm[k] = m;
                vv[k] = v;
                pp[k] = pp[k] - a.step_size * (m / denom);
            }
            if (a.write_clipped) ((float4*)a.g)[e] = make_float4(gg[0], gg[1], gg[2], gg[3]);
            if (a.amsgrad) ((float4*)a.vmax)[e] = make_float4(xx[0], xx[1], xx[2], xx[3]);
            ((float4*)a.m)[e] = make_float4(mm[0], mm[1], mm[2], mm[3]);
            ((float4*)a.v)[e] = make_float4(vv[0], vv[1], vv[2], vv[3]);
            ((float4*)a.p)[e] = make_float4(pp[0], pp[1], pp[2], pp[3]);
        }
        return;
    }
    for (long e = (long)blockIdx.x * AVC_THREADS + threadIdx.x; e < a.n; e += (long)gridDim.x * AVC_THREADS) {
        float p = a.p[e];
        float g = a.g[e] * coef;
        if (a.write_clipped) a.g[e] = g;
        g = g + a.weight_decay * p;                 // coupled L2 (not AdamW)
        float m = a.beta1 * a.m[e] + (1.0f - a.beta1) * g;
        float v = a.beta2 * a.v[e] + (1.0f - a.beta2) * g * g;
        float denom;
        if (a.amsgrad) {
            float vm = fmaxf(a.vmax[e], v);
            a.vmax[e] = vm;
            denom = sqrtf(vm) / a.sqrt_bc2 + a.eps;
        } else {
            denom = sqrtf(v) / a.sqrt_bc2 + a.eps;
        }
        a.m[e] = m;
        a.v[e] = v;
        a.p[e] = p - a.step_size * (m / denom);
    }
}

// --------------------------------------------------------------------------
// launchers
// --------------------------------------------------------------------------
template <int LPR, int NV>
static void launch_in_fwd(const INFwdArgs& a, hipStream_t s) {
    int rpb = AVC_THREADS / LPR;
    hipLaunchKernelGGL((instnorm_fwd_kernel<LPR, NV>), dim3(avc_cdiv(a.R, rpb)), dim3(AVC_THREADS), 0, s, a);
}
template <int LPR, int NV>
static void launch_in_bwd(const INBwdArgs& a, hipStream_t s) {
    int rpb = AVC_THREADS / LPR;
    hipLaunchKernelGGL((instnorm_bwd_kernel<LPR, NV>), dim3(avc_cdiv(a.R, rpb)), dim3(AVC_THREADS), 0, s, a);
}

int avc_launch_in_fwd(const INFwdArgs& a, hipStream_t s) {
    ProfScope ps(AVC_K_IN_FWD, 0.0, 2.0 * 4.0 * (double)a.R * a.T, s);  // 1 read + 1 write
    int n4 = a.T >> 2;
    bool fast = (a.T % 4 == 0) && n4 <= 512 && (((uintptr_t)a.y | (uintptr_t)a.out) % 16 == 0);
    if (!fast) {
        hipLaunchKernelGGL(instnorm_fwd_generic_kernel, dim3(avc_cdiv(a.R, 4)), dim3(AVC_THREADS), 0, s, a);
    } else if (n4 <= 4) launch_in_fwd<4, 1>(a, s);
    else if (n4 <= 8) launch_in_fwd<8, 1>(a, s);
    else if (n4 <= 16) launch_in_fwd<16, 1>(a, s);
    else if (n4 <= 32) launch_in_fwd<32, 1>(a, s);
    else if (n4 <= 64) launch_in_fwd<64, 1>(a, s);
    else if (n4 <= 128) launch_in_fwd<64, 2>(a, s);
    else if (n4 <= 256) launch_in_fwd<64, 4>(a, s);
    else launch_in_fwd<64, 8>(a, s);
    return (int)hipGetLastError();
}

int avc_launch_in_bwd(const INBwdArgs& a, hipStream_t s) {
    ProfScope ps(AVC_K_IN_BWD, 0.0, 3.0 * 4.0 * (double)a.R * a.T, s);  // 2 reads + 1 write
    int n4 = a.T >> 2;
    bool fast = (a.T % 4 == 0) && n4 <= 512 && (((uintptr_t)a.y | (uintptr_t)a.g | (uintptr_t)a.dy) % 16 == 0);
    if (!fast) {
        hipLaunchKernelGGL(instnorm_bwd_generic_kernel, dim3(avc_cdiv(a.R, 4)), dim3(AVC_THREADS), 0, s, a);
    } else if (n4 <= 4) launch_in_bwd<4, 1>(a, s);
    else if (n4 <= 8) launch_in_bwd<8, 1>(a, s);
    else if (n4 <= 16) launch_in_bwd<16, 1>(a, s);
    else if (n4 <= 32) launch_in_bwd<32, 1>(a, s);
    else if (n4 <= 64) launch_in_bwd<64, 1>(a, s);
    else if (n4 <= 128) launch_in_bwd<64, 2>(a, s);
    else if (n4 <= 256) launch_in_bwd<64, 4>(a, s);
    else launch_in_bwd<64, 8>(a, s);
    return (int)hipGetLastError();
}

static int ew_blocks(long n) {
    long b = (n + AVC_THREADS - 1) / AVC_THREADS;
    if (b < 1) b = 1;
    if (b > 2048) b = 2048;
    return (int)b;
}

int avc_launch_copy_rows(const float* x, long sxb, long sxc, int sxt, int B, int M, int T, float* dst, long db, long dc,
                         hipStream_t s) {
    ProfScope ps(AVC_K_MISC, 0.0, 0.0, s);
    const bool vec = sxt == 1 && T % 4 == 0 && sxb % 4 == 0 && sxc % 4 == 0 && db % 4 == 0 && dc % 4 == 0 && (long)B * M * T < (1L << 31) &&
                     ((uintptr_t)x & 15) == 0 && ((uintptr_t)dst & 15) == 0;
    if (vec) hipLaunchKernelGGL(copy_rows_kernel<true>, dim3(ew_blocks((long)B * M * T / 4)), dim3(AVC_THREADS), 0, s, x, sxb, sxc, sxt, B, M, T, dst, db, dc);
    else hipLaunchKernelGGL(copy_rows_kernel<false>, dim3(ew_blocks((long)B * M * T)), dim3(AVC_THREADS), 0, s, x, sxb, sxc, sxt, B, M, T, dst, db, dc);
    return (int)hipGetLastError();
}
int avc_launch_gather_segments(const float* corpus, long n_rows, int M, const long* starts, int B, int T, float* out,
                               hipStream_t s) {
    if (B < 1 || T < 1 || M < 1) return -1;
    ProfScope ps(AVC_K_MISC, 0.0, 8.0 * (double)B * M * T, s);
    hipLaunchKernelGGL(gather_segments_kernel, dim3(avc_cdiv(T, 32), B, avc_cdiv(M, AVC_GATHER_MB)), dim3(AVC_THREADS), 0, s, corpus,
                       n_rows, M, starts, T, out);
    return (int)hipGetLastError();
}
int avc_launch_add_transposed(float* dst, const float* src, int B, int C, hipStream_t s) {
    ProfScope ps(AVC_K_MISC, 0.0, 0.0, s);
    hipLaunchKernelGGL(add_transposed_kernel, dim3(avc_cdiv(B * C, AVC_THREADS)), dim3(AVC_THREADS), 0, s, dst, src, B, C);
    return (int)hipGetLastError();
}
int avc_launch_timepool_fwd(const float* in, int B, int C, int T, float* out, hipStream_t s) {
    ProfScope ps(AVC_K_MISC, 0.0, 0.0, s);
    hipLaunchKernelGGL(timepool_fwd_kernel, dim3(avc_cdiv(B * C, AVC_THREADS)), dim3(AVC_THREADS), 0, s, in, B, C, T, out);
    return (int)hipGetLastError();
}
int avc_launch_timepool_bwd(const float* dP, const float* amask, int B, int C, int T, float* G, float* dy, float slope, hipStream_t s) {
    ProfScope ps(AVC_K_MISC, 0.0, 0.0, s);
    hipLaunchKernelGGL(timepool_bwd_kernel, dim3(ew_blocks((long)B * C * T)), dim3(AVC_THREADS), 0, s, dP, amask, B, C, T,
                       G, dy, slope);
    return (int)hipGetLastError();
}
int avc_launch_reparam_fwd(const float* muls, const float* eps, int B, int C, int Tb, float* z, hipStream_t s) {
    ProfScope ps(AVC_K_MISC, 0.0, 0.0, s);
    hipLaunchKernelGGL(reparam_fwd_kernel, dim3(ew_blocks((long)B * C * Tb)), dim3(AVC_THREADS), 0, s, muls, eps, B, C, Tb, z);
    return (int)hipGetLastError();
}
int avc_launch_latent_bwd(const float* muls, const float* eps, const float* dz, const float* dmuls_up, int B, int C,
                          int Tb, float lambda_kl_over_n, float* dmuls, hipStream_t s) {
    ProfScope ps(AVC_K_MISC, 0.0, 0.0, s);
    hipLaunchKernelGGL(latent_bwd_kernel, dim3(ew_blocks((long)B * C * Tb)), dim3(AVC_THREADS), 0, s, muls, eps, dz,
                       dmuls_up, B, C, Tb, lambda_kl_over_n, dmuls);
    return (int)hipGetLastError();
}
int avc_loss_blocks(long n) {
    int b = ew_blocks(n);
    return b > 512 ? 512 : b;
}
int avc_launch_loss(const float* dec, const float* x, long sxb, long sxc, int sxt, int B, int M, int T, const float* muls,
                    int C, int Tb, float lambda_rec, float* ddec, float* partial, float* losses, hipStream_t s) {
    ProfScope ps(AVC_K_MISC, 0.0, 0.0, s);
    long n = (long)B * M * T, nk = (long)B * C * Tb;
    int blocks = avc_loss_blocks(n);
    const bool vec = sxt == 1 && T % 4 == 0 && sxb % 4 == 0 && sxc % 4 == 0 && n < (1L << 31) && ((uintptr_t)x & 15) == 0 && ((uintptr_t)dec & 15) == 0 &&
                     (!ddec || ((uintptr_t)ddec & 15) == 0);
    if (vec) hipLaunchKernelGGL(loss_partial_kernel<true>, dim3(blocks), dim3(AVC_THREADS), 0, s, dec, x, sxb, sxc, sxt, B, M, T, muls, C, Tb, lambda_rec / (float)n, ddec, partial);
    else hipLaunchKernelGGL(loss_partial_kernel<false>, dim3(blocks), dim3(AVC_THREADS), 0, s, dec, x, sxb, sxc, sxt, B, M, T, muls, C, Tb, lambda_rec / (float)n, ddec, partial);
    hipLaunchKernelGGL(loss_final_kernel, dim3(1), dim3(AVC_THREADS), 0, s, partial, blocks, 1.0f / (float)n,
                       0.5f / (float)nk, losses);
    return (int)hipGetLastError();
}
int avc_adam_blocks(long n) {
    int b = ew_blocks((n + 3) / 4);
    return b > 1024 ? 1024 : b;
}
int avc_launch_sumsq(const float* g, long n, float* partial, hipStream_t s) {
    ProfScope ps(AVC_K_MISC, 0.0, 0.0, s);
    hipLaunchKernelGGL(sumsq_partial_kernel, dim3(avc_adam_blocks(n)), dim3(AVC_THREADS), 0, s, g, n, partial);
    return (int)hipGetLastError();
}
int avc_launch_clip_adam(const AdamArgs& a, hipStream_t s) {
    ProfScope ps(AVC_K_ADAM, 0.0, 9.0 * 4.0 * (double)a.n, s);
    hipLaunchKernelGGL(clip_adam_kernel, dim3(avc_adam_blocks(a.n)), dim3(AVC_THREADS), 0, s, a);
    return (int)hipGetLastError();
}
