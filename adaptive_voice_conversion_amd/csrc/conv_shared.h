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

static inline __device__ float conv_load_res(const ConvEpi& a, const float* res, int b, int m, int t) {
    const float* base = res + a.rbase + (long)b * a.rb + (long)m * a.rc;
    switch (a.res_mode) {
        case AVC_RES_IDENTITY:
            return base[(long)t * a.rt];
        case AVC_RES_AVGPOOL2: {
            int i0 = 2 * t, i1 = 2 * t + 1;
            float v0 = base[(long)i0 * a.rt];
            if (i1 < a.Tres) return (v0 + base[(long)i1 * a.rt]) * 0.5f;
            return v0;  // clipped window of ceil_mode: divisor 1
        }
        case AVC_RES_POOLT: {
            float gsrc = base[(long)(t >> 1) * a.rt];
            bool single = (a.Tout & 1) && (t == a.Tout - 1);
            return single ? gsrc : gsrc * 0.5f;
        }
        case AVC_RES_UPT:
            return base[(long)(2 * t) * a.rt] + base[(long)(2 * t + 1) * a.rt];
        default:
            return 0.f;
    }
}


// Epilogue of one 32x32 accumulator fragment (C/D map of the 32x32 MFMAs: column = lane & 31,
// row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)): bias, ReLU, pixel-shuffle store index, residual /
// gradient join, secondary output and ReLU mask of the backward pass.  m_base = first output row of the
// fragment, (b, t) = the lane's column.
static __device__ __forceinline__ void conv_store_frag(const ConvEpi& a, const ConvGroup& g, const f32x16& acc, int m_base, int h,
                                                       int b, int t) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        int m = m_base + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (m >= a.M) continue;
        float v = acc[r];
        if (g.bias) v += g.bias[m];
        if (a.act == 1) v = avc_act(v, a.slope);
        long o;
        if (a.ops == 1)
            o = a.obase + (long)b * a.ob + (long)m * a.oc + (long)t * a.ot;
        else
            o = a.obase + (long)b * a.ob + (long)(m / a.ops) * a.oc + (long)(t * a.ops + (m % a.ops)) * a.ot;
        float rr = 0.f;
        if (a.res_mode != AVC_RES_NONE) rr = conv_load_res(a, g.res, b, m, t);
        if (a.res_to_primary) v += rr;
        if (g.out) g.out[o] = v;
        if (g.out2) {
            float v2 = a.res_to_primary ? v : v + rr;
            if (g.mask) v2 = avc_act_grad(v2, g.mask[o] > 0.f, a.slope);
            g.out2[o] = v2;
        }
    }
}

// ---- the same epilogue on bf16 PAIR tensors (bf16_pairs.h): rows m (even) and m + 1 of the lane's column are one dword.
// Strides are dword strides of the [B][C/2][T] tensors; M is even; time stride 1.
static inline __device__ void conv_load_res_pair(const ConvEpi& a, const float* res, int b, int m, int t, float& r0, float& r1) {
    const unsigned* base = (const unsigned*)res + a.rbase + (long)b * a.rb + (long)(m >> 1) * a.rc;
    switch (a.res_mode) {
        case AVC_RES_IDENTITY: {
            const unsigned d = base[t];
            r0 = bh_lo(d); r1 = bh_hi(d);
            return;
        }
        case AVC_RES_AVGPOOL2: {
            const int i0 = 2 * t, i1 = 2 * t + 1;
            const unsigned d0 = base[i0];
            r0 = bh_lo(d0); r1 = bh_hi(d0);
            if (i1 < a.Tres) {   // (a clipped window of ceil_mode divides by 1)
                const unsigned d1 = base[i1];
                r0 = (r0 + bh_lo(d1)) * 0.5f;
                r1 = (r1 + bh_hi(d1)) * 0.5f;
            }
            return;
        }
        case AVC_RES_POOLT: {
            const unsigned d = base[t >> 1];
            const bool single = (a.Tout & 1) && (t == a.Tout - 1);
            const float k = single ? 1.0f : 0.5f;
            r0 = bh_lo(d) * k; r1 = bh_hi(d) * k;
            return;
        }
        case AVC_RES_UPT: {
            const unsigned d0 = base[2 * t], d1 = base[2 * t + 1];
            r0 = bh_lo(d0) + bh_lo(d1);
            r1 = bh_hi(d0) + bh_hi(d1);
            return;
        }
        default:
            r0 = r1 = 0.f;
    }
}

static __device__ __forceinline__ void conv_store_frag_pairs(const ConvEpi& a, const ConvGroup& g, const f32x16& acc, int m_base, int h,
                                                             int b, int t) {
    unsigned* out = (unsigned*)g.out;
    unsigned* out2 = (unsigned*)g.out2;
    const unsigned* mask = (const unsigned*)g.mask;
#pragma unroll
    for (int rp = 0; rp < 8; ++rp) {
        const int r = 2 * rp;                                     // accumulator rows r, r + 1 = output rows m, m + 1
        const int m = m_base + (r & 3) + 8 * (r >> 2) + 4 * h;    // even
        if (m >= a.M) continue;
        float v0 = acc[r], v1 = acc[r + 1];
        if (g.bias) { v0 += g.bias[m]; v1 += g.bias[m + 1]; }
        if (a.act == 1) { v0 = avc_act(v0, a.slope); v1 = avc_act(v1, a.slope); }
        float r0 = 0.f, r1 = 0.f;
        if (a.res_mode != AVC_RES_NONE) conv_load_res_pair(a, g.res, b, m, t, r0, r1);
        if (a.res_to_primary) { v0 += r0; v1 += r1; }
        // (a pixel-shuffling layer, model.py:52-59, stores its conv-output pairs as they are: rows m, m + 1 are frames 2 t, 2 t + 1
        // of channel m / 2, i.e. the natural [B][C][2 T] bf16 layout -- the InstanceNorm that follows reads such "planar" rows)
        const long o = a.obase + (long)b * a.ob + (long)(m >> 1) * a.oc + t;
        if (out) out[o] = bh_pack(v0, v1);
        if (out2) {
            float w0 = a.res_to_primary ? v0 : v0 + r0, w1 = a.res_to_primary ? v1 : v1 + r1;
            if (mask) {
                const unsigned md = mask[o];
                w0 = avc_act_grad(w0, bh_lo(md) > 0.f, a.slope);
                w1 = avc_act_grad(w1, bh_hi(md) > 0.f, a.slope);
            }
            out2[o] = bh_pack(w0, w1);
        }
    }
}
