// Tile geometry and epilogue pieces shared by the conv kernels (conv_gemm.hip, conv_rs.hip).
#pragma once
#include "avc_common.h"
#include "bf16_pairs.h"

#define AVC_CONV_NJ 6   // source-tile rows of up to 384 positions (conv_x3.hip: 64 positions per LDS-DMA instruction)
#define AVC_CONV_NJ4 10  // conv_gemm.hip: 16 positions x 4 k-steps per LDS-DMA instruction
#define AVC_CONV_MAXROW4 (16 * AVC_CONV_NJ4)

struct ConvGeom {
    int b0, t0, SPT, ncols, SEG, seg_p0, ROWDATA, ROW;
};

static inline __host__ __device__ ConvGeom conv_geom(int mode, int stride, int Tout, int KS, int BN, int tile) {
    ConvGeom q;
    if (Tout >= BN) {
        int tps = avc_cdiv(Tout, BN);
        q.b0 = tile / tps;
        q.t0 = (tile % tps) * BN;
        q.SPT = 1;
        q.ncols = BN;
    } else {
        q.SPT = BN / Tout;
        if (BN == 64) {   // conv_gemm.hip keeps ROW / 16 per-lane source offsets in registers: very short rows (T_l < 8, the
                          // bottleneck of 17..63-frame utterances) take fewer samples per tile instead of a longer LDS row
            const int seg = (mode == 0) ? (Tout - 1) * stride + KS : Tout + 3 * (KS - 1);
            const int fit = (AVC_CONV_MAXROW4 - KS) / seg;
            q.SPT = q.SPT < fit ? q.SPT : (fit < 1 ? 1 : fit);
        }
        q.b0 = tile * q.SPT;
        q.t0 = 0;
        q.ncols = Tout;
    }
    if (mode == 0) {
        q.SEG = (q.ncols - 1) * stride + KS;
        q.seg_p0 = q.t0 * stride;
    } else {
        // main window + both mirror windows of the reflect-padding adjoint (a column
        // within padR of the end may sit in the last-but-one tile: +(KS-1) slack)
        q.SEG = q.ncols + 3 * (KS - 1);
        q.seg_p0 = q.t0;
    }
    q.ROWDATA = q.SPT * q.SEG;
    q.ROW = q.ROWDATA + KS;  // trailing KS zeros: the "null window" of inactive mirror terms / masked columns
    return q;
}

// what the epilogue needs of a launch: the uniform-length form copies ConvArgs; a ragged tile (one sample) substitutes the
// sample's own lengths and buffer bases (conv_gemm.hip)
struct ConvEpi {
    long ob, oc, rb, rc;
    long obase, rbase;   // added to every output / residual index (ragged: the sample's block inside the packed buffers)
    int ot, ops, rt, Tres, Tout, M, act, res_mode, res_to_primary;
    float slope;
    int pairs;           // bf16 pair storage of out / out2 / res / mask (bf16_pairs.h); strides in DWORDS, ot == rt == 1
};
static __device__ __forceinline__ ConvEpi conv_epi(const ConvArgs& a) {
    ConvEpi e;
    e.ob = a.ob; e.oc = a.oc; e.rb = a.rb; e.rc = a.rc; e.obase = 0; e.rbase = 0;
    e.ot = a.ot; e.ops = a.ops; e.rt = a.rt; e.Tres = a.Tres; e.Tout = a.Tout; e.M = a.M; e.act = a.act;
    e.res_mode = a.res_mode; e.res_to_primary = a.res_to_primary;
    e.slope = a.slope;
    e.pairs = a.pairs;
    return e;
}

// residual term of one output element from the (up to two) values conv_res_fetch loaded for it
static __device__ __forceinline__ float conv_res_value(const ConvEpi& a, float v0, float v1, int t) {
    switch (a.res_mode) {
        case AVC_RES_IDENTITY:
            return v0;
        case AVC_RES_AVGPOOL2:
            return (2 * t + 1 < a.Tres) ? (v0 + v1) * 0.5f : v0;  // clipped window of ceil_mode: divisor 1
        case AVC_RES_POOLT: {
            const bool single = (a.Tout & 1) && (t == a.Tout - 1);
            return single ? v0 : v0 * 0.5f;
        }
        case AVC_RES_UPT:
            return v0 + v1;
        default:
            return 0.f;
    }
}
// element offsets (inside a residual row) of the values conv_res_value combines; i1 < 0: one value only.  Always in range.
static __device__ __forceinline__ void conv_res_index(const ConvEpi& a, int t, long& i0, long& i1) {
    i1 = -1;
    switch (a.res_mode) {
        case AVC_RES_IDENTITY: i0 = (long)t * a.rt; break;
        case AVC_RES_AVGPOOL2: {
            const int j1 = 2 * t + 1;
            i0 = (long)(2 * t) * a.rt;
            i1 = (long)(j1 < a.Tres ? j1 : 2 * t) * a.rt;
            break;
        }
        case AVC_RES_POOLT: i0 = (long)(t >> 1) * a.rt; break;
        case AVC_RES_UPT: i0 = (long)(2 * t) * a.rt; i1 = (long)(2 * t + 1) * a.rt; break;
        default: i0 = 0;
    }
}
static inline __device__ float conv_load_res(const ConvEpi& a, const float* res, int b, int m, int t) {
    const float* base = res + a.rbase + (long)b * a.rb + (long)m * a.rc;
    long i0, i1;
    conv_res_index(a, t, i0, i1);
    return conv_res_value(a, base[i0], i1 >= 0 ? base[i1] : 0.f, t);
}


// Epilogue of one 32x32 accumulator fragment (C/D map of the 32x32 MFMAs: column = lane & 31,
// row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)): bias, ReLU, pixel-shuffle store index, residual /
// gradient join, secondary output and ReLU mask of the backward pass.  m_base = first output row of the
// fragment, (b, t) = the lane's column.
// Eight rows at a time in three phases: every global LOAD of the eight rows (bias, residual, mask; row indices clamped instead of
// branched on, only launch-uniform conditions around them), ONE straight-line block that consumes them, then the stores.  Written
// row by row (load bias - use - load residual - use - store - load mask - use - store, inside `m < M` conditionals) the compiler
// drains the vector-memory counter in front of every use -- up to three exposed memory round trips per ROW, 16 rows per fragment:
// invisible where four workgroups share a CU, most of the run time of the small-grid launches (T_l <= 32, B <= 16).
static __device__ __forceinline__ void conv_store_frag(const ConvEpi& a, const ConvGroup& g, const f32x16& acc, int m_base, int h,
                                                       int b, int t) {
    const bool has_res = a.res_mode != AVC_RES_NONE;
    const bool has_mask = g.out2 && g.mask;
    const bool res_two = a.res_mode == AVC_RES_AVGPOOL2 || a.res_mode == AVC_RES_UPT;   // (launch-uniform: two values per residual term)
    long ri0 = 0, ri1 = -1;
    if (has_res) conv_res_index(a, t, ri0, ri1);
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        float bs[8], q0[8], q1[8], mk[8];
        long o[8];
        // ---- loads
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int r = 8 * half + e;
            const int m = m_base + (r & 3) + 8 * (r >> 2) + 4 * h;
            const int mc = m < a.M ? m : a.M - 1;
            if (a.ops == 1)
                o[e] = a.obase + (long)b * a.ob + (long)mc * a.oc + (long)t * a.ot;
            else
                o[e] = a.obase + (long)b * a.ob + (long)(mc / a.ops) * a.oc + (long)(t * a.ops + (mc % a.ops)) * a.ot;
            bs[e] = q0[e] = q1[e] = mk[e] = 0.f;
        }
        if (g.bias) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int r = 8 * half + e;
                const int m = m_base + (r & 3) + 8 * (r >> 2) + 4 * h;
                bs[e] = g.bias[m < a.M ? m : a.M - 1];
            }
        }
        if (has_res) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int r = 8 * half + e;
                const int m = m_base + (r & 3) + 8 * (r >> 2) + 4 * h;
                const float* base = g.res + a.rbase + (long)b * a.rb + (long)(m < a.M ? m : a.M - 1) * a.rc;
                q0[e] = base[ri0];
                if (res_two) q1[e] = base[ri1];
            }
        }
        if (has_mask) {
#pragma unroll
            for (int e = 0; e < 8; ++e) mk[e] = g.mask[o[e]];
        }
        // ---- everything that depends on a load, in one straight-line block
        float v[8], v2[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float x = g.bias ? acc[8 * half + e] + bs[e] : acc[8 * half + e];
            if (a.act == 1) x = avc_act(x, a.slope);
            const float rr = has_res ? conv_res_value(a, q0[e], q1[e], t) : 0.f;
            if (a.res_to_primary) x += rr;
            float y = a.res_to_primary ? x : x + rr;
            if (has_mask) y = avc_act_grad(y, mk[e] > 0.f, a.slope);
            v[e] = x;
            v2[e] = y;
        }
        // ---- stores
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int r = 8 * half + e;
            const int m = m_base + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (m < a.M) {
                if (g.out) g.out[o[e]] = v[e];
                if (g.out2) g.out2[o[e]] = v2[e];
            }
        }
    }
}

// ---- InstanceNorm / AdaIN / activation / residual fused into the epilogue of a 64 x 64 tile whose columns are whole rows of whole samples
// (Tout = 16 | 32 | 64: 4 | 2 | 1 samples per tile).  The accumulator fragments (+ bias) go through LDS once, row-major; then every lane
// owns 16 consecutive frames of one conv row: statistics by xor-shuffles over the 1 / 2 / 4 lanes of a row segment (and the partner row of
// a pixel-shuffling conv, model.py:52-59, whose rows 2c, 2c + 1 ARE frames 2t, 2t + 1 of channel c), the same two-pass arithmetic and the
// same explicitly rounded x_hat / pre-activation sequence as instnorm_fwd_kernel (rowops.hip), 64-byte contiguous stores of y and out.
#define AVC_IN_LDT 68   // floats per tile row in LDS (16-byte aligned rows)
#define AVC_IN_LDS_BYTES (64 * AVC_IN_LDT * 4)
static __device__ __forceinline__ float4 conv_in_res4(const float* rrow, int mode, int t, int Tres) {   // = res4 of rowops.hip
    float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
    if (mode == AVC_RES_IDENTITY) {
        r = *(const float4*)(rrow + t);
    } else if (mode == AVC_RES_UP2) {
        float2 ab = *(const float2*)(rrow + (t >> 1));
        r = make_float4(ab.x, ab.x, ab.y, ab.y);
    } else if (mode == AVC_RES_AVGPOOL2) {   // (fused rows: Tres == 2 T, multiples of 16)
        float4 p = *(const float4*)(rrow + 2 * t), q = *(const float4*)(rrow + 2 * t + 4);
        r = make_float4((p.x + p.y) * 0.5f, (p.z + p.w) * 0.5f, (q.x + q.y) * 0.5f, (q.z + q.w) * 0.5f);
    }
    return r;
}
static __device__ void conv_epilogue_in(const ConvArgs& a, const ConvGroup& g, const f32x16& acc, float* tile, int tid, int wave_m, int wave_n,
                                        int li, int h, int m_tile0, int b0) {
    const ConvINFuse& f = a.in;
    const int col = wave_n * 32 + li;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = wave_m * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        const int m = m_tile0 + row;
        const float bs = g.bias ? g.bias[m < a.M ? m : a.M - 1] : 0.f;
        tile[row * AVC_IN_LDT + col] = acc[r] + bs;
    }
    __syncthreads();
    const int r = tid >> 2, qd = tid & 3;
    const int m = m_tile0 + r;
    const int Tout = a.Tout, lpr = Tout >> 4;             // lanes per row segment of one sample: 1, 2, 4
    const int bl = (16 * qd) / Tout, t0 = 16 * qd - bl * Tout;
    const int b = b0 + bl;
    const bool valid = m < a.M && b < a.B;
    float v[16];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float4 x = *(const float4*)(tile + r * AVC_IN_LDT + 16 * qd + 4 * k);
        v[4 * k] = x.x; v[4 * k + 1] = x.y; v[4 * k + 2] = x.z; v[4 * k + 3] = x.w;
    }
    const int ops = a.ops;                                  // 1, or 2 = pixel shuffle
    const int c = ops == 2 ? m >> 1 : m, j = ops == 2 ? (m & 1) : 0;
    const int Tn = Tout * ops;
    const float invT = 1.0f / (float)Tn;
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) s += (v[4 * k] + v[4 * k + 1]) + (v[4 * k + 2] + v[4 * k + 3]);
    for (int o = 1; o < lpr; o <<= 1) s += __shfl_xor(s, o);
    if (ops == 2) s += __shfl_xor(s, 4);
    const float mean = s * invT;
    float ss = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float dx = v[4 * k] - mean, dy = v[4 * k + 1] - mean, dz = v[4 * k + 2] - mean, dw = v[4 * k + 3] - mean;
        ss += (dx * dx + dy * dy) + (dz * dz + dw * dw);
    }
    for (int o = 1; o < lpr; o <<= 1) ss += __shfl_xor(ss, o);
    if (ops == 2) ss += __shfl_xor(ss, 4);
    const float rstd = 1.0f / sqrtf(ss * invT + AVC_IN_EPS);
    int fbase = t0;                                         // first frame of this lane's 16 values inside the normalised row
    if (ops == 2) {   // the two lanes of a channel exchange halves: each then holds 16 CONSECUTIVE frames
        float w[16];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float rcv = __shfl_xor(j ? v[i] : v[8 + i], 4);
            w[2 * i] = j ? rcv : v[i];
            w[2 * i + 1] = j ? v[8 + i] : rcv;
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = w[i];
        fbase = 2 * t0 + 16 * j;
    }
    if (!valid) return;
    const long rowi = (long)b * f.C + c;
    if (qd % lpr == 0 && j == 0) {
        f.mean[rowi] = mean;
        f.rstd[rowi] = rstd;
    }
    float gamma = 1.f, beta = 0.f;
    if (f.cond) {
        const float* cr = f.cond + (long)b * f.cond_sb + f.cond_off;
        beta = cr[c];
        gamma = cr[f.C + c];
    }
    float4 rv[4];
    const float* rrow = f.res ? f.res + rowi * f.Tres : nullptr;
#pragma unroll
    for (int k = 0; k < 4; ++k) rv[k] = rrow ? conv_in_res4(rrow, f.res_mode, fbase + 4 * k, f.Tres) : make_float4(0.f, 0.f, 0.f, 0.f);
    float* yrow = g.out + rowi * Tn + fbase;
    float* orow = f.out + rowi * Tn + fbase;
#pragma unroll
    for (int k = 0; k < 4; ++k) *(float4*)(yrow + 4 * k) = make_float4(v[4 * k], v[4 * k + 1], v[4 * k + 2], v[4 * k + 3]);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float w = in_preact(in_xhat(v[4 * k + e], mean, rstd), gamma, beta);
            o[e] = f.relu ? avc_act(w, a.slope) : w;
        }
        *(float4*)(orow + 4 * k) = make_float4(o[0] + rv[k].x, o[1] + rv[k].y, o[2] + rv[k].z, o[3] + rv[k].w);
    }
}

// ---- the backward twin: an input-gradient launch whose output rows g are d(loss)/d(out) of an InstanceNorm layer (ConvINBwd).  The
// fragments go through LDS by (sample, frame) -- the stride-2 instances deal their columns to the waves by parity, so the tile column is
// computed from the lane's own (sample, frame) -- then every lane owns 16 consecutive frames of one channel row: the residual-gradient
// join the epilogue would have applied (identity / pool^T / up^T, all "to primary"), g stored if somebody else reads it, and
//   gm = g * 1[xh gamma + beta > 0];  dbeta = sum gm;  dgamma = sum gm xh;  dy = rstd (gm gamma - mean_T(gm gamma) - xh mean_T(gm gamma xh))
// with xh and the activation decision recomputed from the saved y, mean, rstd by the forward pass's own rounding sequence.
static __device__ void conv_epilogue_in_bwd(const ConvArgs& a, const ConvGroup& g, const f32x16& acc, float* tile, int tid, int wave_m, int h,
                                            int m_tile0, int b0, int t0_tile, int bl_lane, int t_lane, bool col_valid) {
    const ConvINBwd& f = a.inb;
    const int Tout = a.Tout, lpr = Tout >> 4;
    if (col_valid) {
        const int col = bl_lane * Tout + (t_lane - t0_tile);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = wave_m * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            tile[row * AVC_IN_LDT + col] = acc[r];
        }
    }
    __syncthreads();
    const int r = tid >> 2, qd = tid & 3;
    const int m = m_tile0 + r;
    const int bl = (16 * qd) / Tout, t0 = 16 * qd - bl * Tout;
    const int b = b0 + bl;
    const bool valid = m < a.M && b < a.B;
    const int mc = m < a.M ? m : a.M - 1, bc = b < a.B ? b : a.B - 1;   // (clamped: every lane takes part in the shuffles)
    float gv[16];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float4 x = *(const float4*)(tile + r * AVC_IN_LDT + 16 * qd + 4 * k);
        gv[4 * k] = x.x; gv[4 * k + 1] = x.y; gv[4 * k + 2] = x.z; gv[4 * k + 3] = x.w;
    }
    const long rowi = (long)bc * f.C + mc;
    // saved forward row + statistics + AdaIN parameters: requested in front of the residual join
    float yv[16];
    const float* yrow = f.y + rowi * Tout + t0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float4 x = *(const float4*)(yrow + 4 * k);
        yv[4 * k] = x.x; yv[4 * k + 1] = x.y; yv[4 * k + 2] = x.z; yv[4 * k + 3] = x.w;
    }
    const float mean = f.mean[rowi], rstd = f.rstd[rowi];
    float gamma = 1.f, beta = 0.f;
    if (f.cond) {
        const float* cr = f.cond + (long)bc * f.cond_sb + f.cond_off;
        beta = cr[mc];
        gamma = cr[f.C + mc];
    }
    if (a.res_mode != AVC_RES_NONE) {   // residual-gradient join (contiguous rows [B][M][Tres]; res_to_primary)
        const float* rrow = g.res + ((long)bc * a.M + mc) * a.Tres;
        if (a.res_mode == AVC_RES_IDENTITY) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float4 x = *(const float4*)(rrow + t0 + 4 * k);
                gv[4 * k] += x.x; gv[4 * k + 1] += x.y; gv[4 * k + 2] += x.z; gv[4 * k + 3] += x.w;
            }
        } else if (a.res_mode == AVC_RES_POOLT) {   // adjoint of the ceil-mode pool: g[t / 2] / 2 (Tout even: no clipped window)
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const float4 x = *(const float4*)(rrow + (t0 >> 1) + 4 * k);
                gv[8 * k] += x.x * 0.5f; gv[8 * k + 1] += x.x * 0.5f; gv[8 * k + 2] += x.y * 0.5f; gv[8 * k + 3] += x.y * 0.5f;
                gv[8 * k + 4] += x.z * 0.5f; gv[8 * k + 5] += x.z * 0.5f; gv[8 * k + 6] += x.w * 0.5f; gv[8 * k + 7] += x.w * 0.5f;
            }
        } else {   // AVC_RES_UPT: adjoint of nearest x2: g[2t] + g[2t + 1]
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float4 x = *(const float4*)(rrow + 2 * t0 + 4 * k);
                gv[2 * k] += x.x + x.y;
                gv[2 * k + 1] += x.z + x.w;
            }
        }
    }
    if (valid && g.out) {   // the block's residual path reads g too
        float* grow = g.out + ((long)b * a.M + m) * Tout + t0;
#pragma unroll
        for (int k = 0; k < 4; ++k) *(float4*)(grow + 4 * k) = make_float4(gv[4 * k], gv[4 * k + 1], gv[4 * k + 2], gv[4 * k + 3]);
    }
    float xh[16], s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const float hh = in_xhat(yv[i], mean, rstd);
        const float w = in_preact(hh, gamma, beta);
        const float gme = avc_act_grad(gv[i], !f.relu || w > 0.f, a.slope);
        xh[i] = hh;
        gv[i] = gme;
        s1 += gme;
        s2 += gme * hh;
    }
    for (int o = 1; o < lpr; o <<= 1) {
        s1 += __shfl_xor(s1, o);
        s2 += __shfl_xor(s2, o);
    }
    if (!valid) return;
    if (f.dcond && qd % lpr == 0) {
        float* dc = f.dcond + (long)b * f.dcond_sb + f.dcond_off;
        dc[m] = s1;
        dc[f.C + m] = s2;
    }
    const float invT = 1.0f / (float)Tout;
    const float m1 = gamma * s1 * invT, m2 = gamma * s2 * invT;
    float* drow = f.dy + rowi * Tout + t0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float4 o;
        o.x = rstd * (gv[4 * k] * gamma - m1 - xh[4 * k] * m2);
        o.y = rstd * (gv[4 * k + 1] * gamma - m1 - xh[4 * k + 1] * m2);
        o.z = rstd * (gv[4 * k + 2] * gamma - m1 - xh[4 * k + 2] * m2);
        o.w = rstd * (gv[4 * k + 3] * gamma - m1 - xh[4 * k + 3] * m2);
        *(float4*)(drow + 4 * k) = o;
    }
}

// ... on bf16 PAIR tensors (compute_dtype "bf16", bf16_pairs.h; ops == 1): y is rounded to bf16 FIRST -- statistics, x_hat and the activation
// decision are taken from the stored values, exactly what instnorm_fwd_pairs_kernel reads back and what the backward pass recomputes --
// then the lanes of rows (2p, 2p + 1) exchange halves: each packs 8 frames of the pair row (two 16-byte stores per tensor).
typedef unsigned conv_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned conv_u32x2 __attribute__((ext_vector_type(2)));
static __device__ __forceinline__ void conv_in_res_pairs4(const unsigned* rrow, int mode, int t, float (&lo)[4], float (&hi)[4]) {   // = res_pairs4 of rowops_pairs.hip
    if (mode == AVC_RES_IDENTITY) {
        const conv_u32x4 v = *(const conv_u32x4*)(rrow + t);
#pragma unroll
        for (int k = 0; k < 4; ++k) { lo[k] = bh_lo(v[k]); hi[k] = bh_hi(v[k]); }
    } else if (mode == AVC_RES_UP2) {
        const conv_u32x2 v = *(const conv_u32x2*)(rrow + (t >> 1));
        lo[0] = lo[1] = bh_lo(v[0]); hi[0] = hi[1] = bh_hi(v[0]);
        lo[2] = lo[3] = bh_lo(v[1]); hi[2] = hi[3] = bh_hi(v[1]);
    } else if (mode == AVC_RES_AVGPOOL2) {
        const conv_u32x4 p = *(const conv_u32x4*)(rrow + 2 * t), q = *(const conv_u32x4*)(rrow + 2 * t + 4);
        lo[0] = (bh_lo(p[0]) + bh_lo(p[1])) * 0.5f; hi[0] = (bh_hi(p[0]) + bh_hi(p[1])) * 0.5f;
        lo[1] = (bh_lo(p[2]) + bh_lo(p[3])) * 0.5f; hi[1] = (bh_hi(p[2]) + bh_hi(p[3])) * 0.5f;
        lo[2] = (bh_lo(q[0]) + bh_lo(q[1])) * 0.5f; hi[2] = (bh_hi(q[0]) + bh_hi(q[1])) * 0.5f;
        lo[3] = (bh_lo(q[2]) + bh_lo(q[3])) * 0.5f; hi[3] = (bh_hi(q[2]) + bh_hi(q[3])) * 0.5f;
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) lo[k] = hi[k] = 0.f;
    }
}
static __device__ void conv_epilogue_in_pairs(const ConvArgs& a, const ConvGroup& g, const f32x16& acc, float* tile, int tid, int wave_m, int wave_n,
                                              int li, int h, int m_tile0, int b0) {
    const ConvINFuse& f = a.in;
    const int col = wave_n * 32 + li;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = wave_m * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        const int m = m_tile0 + row;
        const float bs = g.bias ? g.bias[m < a.M ? m : a.M - 1] : 0.f;
        tile[row * AVC_IN_LDT + col] = acc[r] + bs;
    }
    __syncthreads();
    const int r = tid >> 2, qd = tid & 3;
    const int m = m_tile0 + r, odd = m & 1;
    const int Tout = a.Tout, lpr = Tout >> 4;
    const int bl = (16 * qd) / Tout, t0 = 16 * qd - bl * Tout;
    const int b = b0 + bl;
    const bool valid = m < a.M && b < a.B;
    float v[16];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float4 x = *(const float4*)(tile + r * AVC_IN_LDT + 16 * qd + 4 * k);
        v[4 * k] = bh_lo(bh_pack(x.x, 0.f)); v[4 * k + 1] = bh_lo(bh_pack(x.y, 0.f));
        v[4 * k + 2] = bh_lo(bh_pack(x.z, 0.f)); v[4 * k + 3] = bh_lo(bh_pack(x.w, 0.f));
    }
    const float invT = 1.0f / (float)Tout;
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) s += (v[4 * k] + v[4 * k + 1]) + (v[4 * k + 2] + v[4 * k + 3]);
    for (int o = 1; o < lpr; o <<= 1) s += __shfl_xor(s, o);
    const float mean = s * invT;
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const float d = v[i] - mean;
        ss += d * d;
    }
    for (int o = 1; o < lpr; o <<= 1) ss += __shfl_xor(ss, o);
    const float rstd = 1.0f / sqrtf(ss * invT + AVC_IN_EPS);
    const int mc = m < a.M ? m : a.M - 1;
    float gamma = 1.f, beta = 0.f;
    if (f.cond) {
        const float* cr = f.cond + (long)(b < a.B ? b : a.B - 1) * f.cond_sb + f.cond_off;
        beta = cr[mc];
        gamma = cr[f.C + mc];
    }
    if (valid && qd % lpr == 0) {
        f.mean[(long)b * f.C + m] = mean;
        f.rstd[(long)b * f.C + m] = rstd;
    }
    // rows (2p, 2p + 1): the even row's lane keeps frames [0, 8) of its segment, the odd row's lane frames [8, 16), each with BOTH channels
    float lo[8], hi[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float rcv = __shfl_xor(odd ? v[i] : v[8 + i], 4);
        lo[i] = odd ? rcv : v[i];
        hi[i] = odd ? v[8 + i] : rcv;
    }
    const float pmean = __shfl_xor(mean, 4), prstd = __shfl_xor(rstd, 4), pgamma = __shfl_xor(gamma, 4), pbeta = __shfl_xor(beta, 4);
    const float mean0 = odd ? pmean : mean, mean1 = odd ? mean : pmean, rstd0 = odd ? prstd : rstd, rstd1 = odd ? rstd : prstd;
    const float g0 = odd ? pgamma : gamma, g1 = odd ? gamma : pgamma, be0 = odd ? pbeta : beta, be1 = odd ? beta : pbeta;
    if (!valid) return;
    const int fb = t0 + 8 * odd;
    const long prow = (long)b * (f.C >> 1) + (m >> 1);
    unsigned* yrow = (unsigned*)g.out + prow * Tout + fb;
    unsigned* orow = (unsigned*)f.out + prow * Tout + fb;
    const unsigned* rrow = f.res ? (const unsigned*)f.res + prow * f.Tres : nullptr;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        float rl[4], rh[4];
        if (rrow) conv_in_res_pairs4(rrow, f.res_mode, fb + 4 * k, rl, rh);
        else { rl[0] = rl[1] = rl[2] = rl[3] = rh[0] = rh[1] = rh[2] = rh[3] = 0.f; }
        conv_u32x4 yv, ov;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float y0 = lo[4 * k + e], y1 = hi[4 * k + e];
            yv[e] = bh_pack(y0, y1);
            const float w0 = in_preact(in_xhat(y0, mean0, rstd0), g0, be0), w1 = in_preact(in_xhat(y1, mean1, rstd1), g1, be1);
            ov[e] = bh_pack((f.relu ? avc_act(w0, a.slope) : w0) + rl[e], (f.relu ? avc_act(w1, a.slope) : w1) + rh[e]);
        }
        *(conv_u32x4*)(yrow + 4 * k) = yv;
        *(conv_u32x4*)(orow + 4 * k) = ov;
    }
}

// ... the backward twin on bf16 PAIR tensors (round 6).  g = conv^T(dy) [+ residual-gradient join] is rounded to bf16 FIRST -- what the
// two-launch path stores and instnorm_bwd_pairs_kernel reads back -- and the InstanceNorm / AdaIN / activation backward is taken from the
// rounded values; y, the join source, g (where kept) and dy are pair rows [B][C/2][T].  A lane owns 16 frames of ONE channel row for the
// arithmetic (conv_epilogue_in_bwd's layout); the lanes of rows (2p, 2p + 1) exchange halves for the stores, each packing 8 frames of both.
static __device__ void conv_epilogue_in_bwd_pairs(const ConvArgs& a, const ConvGroup& g, const f32x16& acc, float* tile, int tid, int wave_m, int h,
                                                  int m_tile0, int b0, int t0_tile, int bl_lane, int t_lane, bool col_valid) {
    const ConvINBwd& f = a.inb;
    const int Tout = a.Tout, lpr = Tout >> 4;
    if (col_valid) {
        const int col = bl_lane * Tout + (t_lane - t0_tile);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = wave_m * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            tile[row * AVC_IN_LDT + col] = acc[r];
        }
    }
    __syncthreads();
    const int r = tid >> 2, qd = tid & 3;
    const int m = m_tile0 + r, odd = m & 1;
    const int bl = (16 * qd) / Tout, t0 = 16 * qd - bl * Tout;
    const int b = b0 + bl;
    const bool valid = m < a.M && b < a.B;
    const int mc = m < a.M ? m : a.M - 1, bc = b < a.B ? b : a.B - 1;   // (clamped: every lane takes part in the exchanges)
    float gv[16];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float4 x = *(const float4*)(tile + r * AVC_IN_LDT + 16 * qd + 4 * k);
        gv[4 * k] = x.x; gv[4 * k + 1] = x.y; gv[4 * k + 2] = x.z; gv[4 * k + 3] = x.w;
    }
    const int C2 = f.C >> 1;
    const long prow = (long)bc * C2 + (mc >> 1);       // the pair row of this lane's channel
    const long rowi = (long)bc * f.C + mc;
    // this lane's channel of the saved forward rows: its half of the 16 dwords (the partner lane reads the same dwords: one cache line)
    float yv[16];
    const unsigned* yrow = (const unsigned*)f.y + prow * Tout + t0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const conv_u32x4 v = *(const conv_u32x4*)(yrow + 4 * k);
#pragma unroll
        for (int e = 0; e < 4; ++e) yv[4 * k + e] = odd ? bh_hi(v[e]) : bh_lo(v[e]);
    }
    const float mean = f.mean[rowi], rstd = f.rstd[rowi];
    float gamma = 1.f, beta = 0.f;
    if (f.cond) {
        const float* cr = f.cond + (long)bc * f.cond_sb + f.cond_off;
        beta = cr[mc];
        gamma = cr[f.C + mc];
    }
    if (a.res_mode != AVC_RES_NONE) {   // residual-gradient join (pair rows [B][M/2][Tres]; res_to_primary)
        const unsigned* rrow = (const unsigned*)g.res + ((long)bc * (a.M >> 1) + (mc >> 1)) * a.Tres;
        if (a.res_mode == AVC_RES_IDENTITY) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const conv_u32x4 v = *(const conv_u32x4*)(rrow + t0 + 4 * k);
#pragma unroll
                for (int e = 0; e < 4; ++e) gv[4 * k + e] += odd ? bh_hi(v[e]) : bh_lo(v[e]);
            }
        } else if (a.res_mode == AVC_RES_POOLT) {   // adjoint of the ceil-mode pool: g[t / 2] / 2 (Tout even: no clipped window)
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const conv_u32x4 v = *(const conv_u32x4*)(rrow + (t0 >> 1) + 4 * k);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float x = (odd ? bh_hi(v[e]) : bh_lo(v[e])) * 0.5f;
                    gv[8 * k + 2 * e] += x;
                    gv[8 * k + 2 * e + 1] += x;
                }
            }
        } else {   // AVC_RES_UPT: adjoint of nearest x2: g[2t] + g[2t + 1]
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const conv_u32x4 v = *(const conv_u32x4*)(rrow + 2 * t0 + 4 * k);
                gv[2 * k] += odd ? bh_hi(v[0]) + bh_hi(v[1]) : bh_lo(v[0]) + bh_lo(v[1]);
                gv[2 * k + 1] += odd ? bh_hi(v[2]) + bh_hi(v[3]) : bh_lo(v[2]) + bh_lo(v[3]);
            }
        }
    }
    // g as the two-launch path stores it: ONE rounding of (accumulator + join)
#pragma unroll
    for (int i = 0; i < 16; ++i) gv[i] = bh_lo(bh_pack(gv[i], 0.f));
    // pack helper: rows (2p, 2p + 1) -- the even row's lane keeps frames [0, 8) of its 16, the odd row's lane frames [8, 16), each with BOTH channels
    auto store_pairs = [&](unsigned* base, const float (&v)[16]) {
        float lo[8], hi[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float rcv = __shfl_xor(odd ? v[i] : v[8 + i], 4);
            lo[i] = odd ? rcv : v[i];
            hi[i] = odd ? v[8 + i] : rcv;
        }
        if (!valid) return;
        unsigned* drow = base + prow * Tout + t0 + 8 * odd;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            conv_u32x4 w;
#pragma unroll
            for (int e = 0; e < 4; ++e) w[e] = bh_pack(lo[4 * k + e], hi[4 * k + e]);
            *(conv_u32x4*)(drow + 4 * k) = w;
        }
    };
    if (g.out) store_pairs((unsigned*)g.out, gv);   // the block's residual path reads g too (launch-uniform)
    float xh[16], s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const float hh = in_xhat(yv[i], mean, rstd);
        const float w = in_preact(hh, gamma, beta);
        const float gme = avc_act_grad(gv[i], !f.relu || w > 0.f, a.slope);
        xh[i] = hh;
        gv[i] = gme;
        s1 += gme;
        s2 += gme * hh;
    }
    for (int o = 1; o < lpr; o <<= 1) {
        s1 += __shfl_xor(s1, o);
        s2 += __shfl_xor(s2, o);
    }
    if (valid && f.dcond && qd % lpr == 0) {
        float* dc = f.dcond + (long)b * f.dcond_sb + f.dcond_off;
        dc[m] = s1;
        dc[f.C + m] = s2;
    }
    const float invT = 1.0f / (float)Tout;
    const float m1 = gamma * s1 * invT, m2 = gamma * s2 * invT;
    float dv[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) dv[i] = rstd * (gv[i] * gamma - m1 - xh[i] * m2);
    store_pairs((unsigned*)f.dy, dv);
}

// ---- the same epilogue on bf16 PAIR tensors (bf16_pairs.h): rows m (even) and m + 1 of the lane's column are one dword.
// Strides are dword strides of the [B][C/2][T] tensors; M is even; time stride 1.
// the (up to two) dwords of a residual pair row that conv_res_value's modes combine, as element offsets inside the row (time stride 1)
static __device__ __forceinline__ void conv_res_index_pair(const ConvEpi& a, int t, int& i0, int& i1) {
    i1 = -1;
    switch (a.res_mode) {
        case AVC_RES_IDENTITY: i0 = t; break;
        case AVC_RES_AVGPOOL2: i0 = 2 * t; i1 = (2 * t + 1 < a.Tres) ? 2 * t + 1 : 2 * t; break;
        case AVC_RES_POOLT: i0 = t >> 1; break;
        case AVC_RES_UPT: i0 = 2 * t; i1 = 2 * t + 1; break;
        default: i0 = 0;
    }
}
static inline __device__ void conv_load_res_pair(const ConvEpi& a, const float* res, int b, int m, int t, float& r0, float& r1) {
    const unsigned* base = (const unsigned*)res + a.rbase + (long)b * a.rb + (long)(m >> 1) * a.rc;
    int i0, i1;
    conv_res_index_pair(a, t, i0, i1);
    const unsigned d0 = base[i0], d1 = i1 >= 0 ? base[i1] : 0u;
    r0 = conv_res_value(a, bh_lo(d0), bh_lo(d1), t);
    r1 = conv_res_value(a, bh_hi(d0), bh_hi(d1), t);
}

// (phases as in conv_store_frag: the loads of four row pairs, one straight-line block, the stores)
static __device__ __forceinline__ void conv_store_frag_pairs(const ConvEpi& a, const ConvGroup& g, const f32x16& acc, int m_base, int h,
                                                             int b, int t) {
    unsigned* out = (unsigned*)g.out;
    unsigned* out2 = (unsigned*)g.out2;
    const unsigned* mask = (const unsigned*)g.mask;
    const bool has_res = a.res_mode != AVC_RES_NONE;
    const bool has_mask = out2 && mask;
    const bool res_two = a.res_mode == AVC_RES_AVGPOOL2 || a.res_mode == AVC_RES_UPT;
    int ri0 = 0, ri1 = -1;
    if (has_res) conv_res_index_pair(a, t, ri0, ri1);
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        float b0[4], b1[4];
        unsigned d0[4], d1[4], md[4];
        long o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int r = 2 * (4 * half + e);                          // accumulator rows r, r + 1 = output rows m, m + 1
            const int m = m_base + (r & 3) + 8 * (r >> 2) + 4 * h;     // even
            const int mc = m < a.M ? m : a.M - 2;
            // (a pixel-shuffling layer, model.py:52-59, stores its conv-output pairs as they are: rows m, m + 1 are frames 2 t, 2 t + 1
            // of channel m / 2, i.e. the natural [B][C][2 T] bf16 layout -- the InstanceNorm that follows reads such "planar" rows)
            o[e] = a.obase + (long)b * a.ob + (long)(mc >> 1) * a.oc + t;
            b0[e] = b1[e] = 0.f;
            d0[e] = d1[e] = md[e] = 0u;
        }
        if (g.bias) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int r = 2 * (4 * half + e);
                const int m = m_base + (r & 3) + 8 * (r >> 2) + 4 * h;
                const int mc = m < a.M ? m : a.M - 2;
                b0[e] = g.bias[mc];
                b1[e] = g.bias[mc + 1];
            }
        }
        if (has_res) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int r = 2 * (4 * half + e);
                const int m = m_base + (r & 3) + 8 * (r >> 2) + 4 * h;
                const int mc = m < a.M ? m : a.M - 2;
                const unsigned* base = (const unsigned*)g.res + a.rbase + (long)b * a.rb + (long)(mc >> 1) * a.rc;
                d0[e] = base[ri0];
                if (res_two) d1[e] = base[ri1];
            }
        }
        if (has_mask) {
#pragma unroll
            for (int e = 0; e < 4; ++e) md[e] = mask[o[e]];
        }
        unsigned w[4], w2[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int r = 2 * (4 * half + e);
            float v0 = acc[r], v1 = acc[r + 1];
            if (g.bias) { v0 += b0[e]; v1 += b1[e]; }
            if (a.act == 1) { v0 = avc_act(v0, a.slope); v1 = avc_act(v1, a.slope); }
            float r0 = 0.f, r1 = 0.f;
            if (has_res) {
                r0 = conv_res_value(a, bh_lo(d0[e]), bh_lo(d1[e]), t);
                r1 = conv_res_value(a, bh_hi(d0[e]), bh_hi(d1[e]), t);
            }
            if (a.res_to_primary) { v0 += r0; v1 += r1; }
            float w0 = a.res_to_primary ? v0 : v0 + r0, w1 = a.res_to_primary ? v1 : v1 + r1;
            if (has_mask) {
                w0 = avc_act_grad(w0, bh_lo(md[e]) > 0.f, a.slope);
                w1 = avc_act_grad(w1, bh_hi(md[e]) > 0.f, a.slope);
            }
            w[e] = bh_pack(v0, v1);
            w2[e] = bh_pack(w0, w1);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int r = 2 * (4 * half + e);
            const int m = m_base + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (m < a.M) {
                if (out) out[o[e]] = w[e];
                if (out2) out2[o[e]] = w2[e];
            }
        }
    }
}
