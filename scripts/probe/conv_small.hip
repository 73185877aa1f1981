// One-shot Conv1d for SHORT rows (T_l = 16 / 32, k = 5, 128 reduction channels) on the exact-fp32 MFMA.
//
// Same contraction, operands (the packed weight images of conv_gemm.hip) and fused epilogues as conv_gemm.hip
// (reference: model.py:21-32 pad_layer + nn.Conv1d and its input gradient).  At T_l <= 32 a layer of the B = 256 step is
// 4-8k columns: 128-256 tiles of 64x64 whose workgroups walk the K axis in a double-buffered chunk pipeline -- four to
// eight small LDS-DMA rounds, each one exposed global->LDS latency next to 1.2 us of MFMAs, on half-empty CUs
// (measured 23-32 us per launch against a 4-9 us FLOP time).  Here a workgroup owns a 32-row weight slab x 64 / 128
// columns (whole samples) and loads EVERYTHING it needs in one burst: the slab for the whole K axis (80 KiB) and the
// [128][ROW] source tile, all DMAs in flight together, one wait, one barrier.  The K axis is then split over four
// wave groups (each a 32x32 tile per wave, 80 MFMAs), the partial tiles are summed through LDS in a fixed order
// (deterministic) and the first group runs the epilogue.  256 workgroups per launch: one per CU, 8-16 waves each.
#include <hip/hip_runtime.h>

#include "avc_common.h"
#include "avc_internal.h"
#include "conv_shared.h"

// NT: 32-column tiles per workgroup (2 or 4); CK: chunk depth of the packed weight image (8 or 16); KS = 5 taps.
template <int NT, int CK, bool MIRROR>
__global__ void __launch_bounds__(NT * 256) conv_small_kernel(const ConvArgs a) {
    constexpr int KS = 5, BN = 32 * NT, NW = NT * 4, NJ = 3;
    HIP_DYNAMIC_SHARED(float, smem)
    const ConvGroup g = a.g[0];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ct = wave % NT, kq = wave / NT;   // column tile, K quarter
    const int li = lane & 31, h = lane >> 5;
    const int padL = g.padL, padR = g.padR, nchunk = g.nchunk;
    const int Tout = a.Tout, Cred = a.Cred;
    // whole samples per tile.  One sample's segment of an LDS row: forward = its reflect-padded frames; dgrad = KS-1 zeros
    // + its frames, the zeros in front of the NEXT segment (or the null window) are its zero extension behind -- main and
    // mirror windows reach at most 4 positions before and 3 behind the frames
    const int SPT = BN / Tout;
    const int SEG = Tout + KS - 1;
    const int ROWDATA = SPT * SEG, ROW = ROWDATA + KS;   // + the null window (zeros)
    const int b0 = blockIdx.x * SPT, m0 = blockIdx.y * 32;
    const int nrows = nchunk * KS * CK;
    float* As = smem;                  // [nrows][32]: the slab's rows of the packed image
    float* Xs = smem + nrows * 32;     // [Cred][ROW]

    // ---- everything in flight at once: the weight slab (16 bytes per lane) ...
    for (int piece = wave; piece * 8 < nrows; piece += NW) {
        const int row = piece * 8 + (lane >> 3);
        avc_glds16(g.wp + (long)row * a.Mp + m0 + (lane & 7) * 4, As + piece * 256);
    }
    // ... and the source tile: lane l owns positions p = 64 j + l of every row (reflect padding / zero extension
    // resolved here once; structural zeros are written by the lane itself)
    int xoff[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int p = 64 * j + lane;
        int sp = -2;   // -2: past the row, -1: structural zero
        if (p < ROW) {
            sp = -1;
            if (p < ROWDATA) {
                const int seg = p / SEG, qq = p - seg * SEG;
                const int b = b0 + seg;
                if (b < a.B) {
                    if (a.mode == 0) {
                        const int r = avc_reflect(qq - padL, a.Tsrc);
                        if (r >= 0 && r < a.Tsrc) sp = (int)(b * a.x.sb + (long)r * a.x.st);
                    } else {
                        const int v = qq - (KS - 1);
                        if (v >= 0 && v < a.Tsrc) sp = (int)(b * a.x.sb + (long)v * a.x.st);
                    }
                }
            }
        }
        xoff[j] = sp;
    }
    for (int r = wave; r < Cred; r += NW) {
        const float* src = a.x.ptr + (a.x.ps == 1 ? (long)r * a.x.sc : (long)(r / a.x.ps) * a.x.sc + (r % a.x.ps));
        float* Xr = Xs + r * ROW;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            if (xoff[j] >= 0) avc_glds4(src + xoff[j], Xr + 64 * j);
            else if (xoff[j] == -1) Xr[64 * j + lane] = 0.f;
        }
    }

    // ---- this lane's column
    const int n = ct * 32 + li;
    const int bl = n / Tout, t = n - bl * Tout;
    const bool v = (bl < SPT) && (b0 + bl < a.B);
    int cb = ROWDATA, cbm = ROWDATA;   // ROWDATA.. = the null window
    if (v) {
        if (a.mode == 0) {
            cb = bl * SEG + t;
        } else {
            cb = bl * SEG + t + padL;
            if (MIRROR) {   // a column of a sample of >= 10 frames is within pad of at most one edge
                if (t >= 1 && t <= padL) cbm = bl * SEG + (padL - t);
                if (t >= Tout - 1 - padR && t <= Tout - 2) cbm = bl * SEG + (2 * (Tout - 1) - t + padL);
            }
        }
    }

    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();

    // ---- K quarter kq: chunks [kq * cpq, (kq + 1) * cpq)
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int cpq = nchunk >> 2;
    const float* xr = Xs + h * ROW + cb;
    const float* xm = Xs + h * ROW + cbm;
    for (int c = kq * cpq; c < (kq + 1) * cpq; ++c) {
        const float* Ac = As + ((c * KS) * CK + h) * 32 + li;
        const float* Xc = xr + c * CK * ROW;
        const float* Xm = xm + c * CK * ROW;
#pragma unroll
        for (int j = 0; j < KS; ++j) {
            float av[CK / 2], bv[CK / 2];
#pragma unroll
            for (int u = 0; u < CK / 2; ++u) {
                av[u] = Ac[(j * CK + 2 * u) * 32];
                float x = Xc[2 * u * ROW + j];
                if (MIRROR) x += Xm[2 * u * ROW + j];
                bv[u] = x;
            }
#pragma unroll
            for (int u = 0; u < CK / 2; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], bv[u], acc, 0, 0, 0);
        }
    }

    // ---- fixed-order sum of the four partial tiles through the (now free) slab memory
    __syncthreads();
    float* red = smem + ((kq > 0 ? kq - 1 : 0) * NT + ct) * 1024 + lane;
    if (kq > 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) red[r * 64] = acc[r];
    }
    __syncthreads();
    if (kq > 0) return;
#pragma unroll
    for (int k2 = 1; k2 < 4; ++k2) {
        const float* rk = smem + ((k2 - 1) * NT + ct) * 1024 + lane;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] += rk[r * 64];
    }
    if (v) conv_store_frag(a, g, acc, m0, h, b0 + bl, t);
}

// --------------------------------------------------------------------------
// avc_set_tuning("conv_small", v): -1 (default) = launches of <= 64 samples (a step of such a batch is bound by the
// latency of its ~250 dependent launches: 2.31 -> 2.24 ms at B = 4, 3.29 -> 3.24 at B = 64; at B = 256 the kernel is
// faster alone but its 158 KB of LDS keep the concurrent weight-gradient workgroups off the CU: 6.76 vs 6.74 ms);
// 0 = never; bit 0 = forward, bit 1 = dgrad launches
static int g_conv_small = -1;
void avc_set_conv_small(int on) { g_conv_small = on; }   // bit 0: forward launches, bit 1: dgrad launches

static int small_nt(const ConvArgs& a) { return a.Tout == 32 ? 4 : 2; }   // 4 samples of 32 / 16 frames per workgroup
static size_t small_lds(const ConvArgs& a) {
    const int SPT = 32 * small_nt(a) / a.Tout;
    return ((size_t)128 * 5 * 32 + (size_t)128 * (SPT * (a.Tout + 4) + 5)) * 4;   // 158,208 B at T = 32
}

// One workgroup per CU (80 KiB weight slab + the source tile), so only launches that fit the chip in one round:
// measured (profiles/r02_conv_small.log) two rounds lose to the chunk-pipelined kernel, one round wins by 15-40 %.
bool avc_conv_small_eligible(const ConvArgs& a, bool forced) {
    const int bits = g_conv_small < 0 ? (a.B <= 64 ? 3 : 0) : g_conv_small;
    if (!forced && !(bits & (a.mode == 0 ? 1 : 2))) return false;
    if (a.ngroups != 1 || a.in_fuse || a.rs || a.dbg) return false;
    const ConvGroup& g = a.g[0];
    if (g.KS != 5 || a.Cred != 128 || a.stride != 1 || g.padL != 2 || g.padR != 2) return false;
    if ((a.Tout != 16 && a.Tout != 32) || a.Tsrc != a.Tout) return false;
    if ((g.CK != 8 && g.CK != 16) || g.nchunk * g.CK != 128) return false;
    if (a.bf16 != AVC_COMPUTE_F32) return false;
    const int SPT = 32 * small_nt(a) / a.Tout;
    if ((long)avc_cdiv(a.B, SPT) * avc_cdiv(a.M, 32) > 256) return false;
    return small_lds(a) <= 160 * 1024;
}

int avc_launch_conv_small(const ConvArgs& a, hipStream_t stream) {
    const ConvGroup& g = a.g[0];
    const int NT = small_nt(a), BN = 32 * NT, SPT = BN / a.Tout;
    const size_t lds = small_lds(a);
    dim3 grid(avc_cdiv(a.B, SPT), avc_cdiv(a.M, 32)), block(NT * 256);
    const double flops = 2.0 * a.M * a.Cred * g.KS * (double)a.B * a.Tout;
    ProfScope ps(a.mode == 0 ? AVC_K_CONV_FWD : AVC_K_CONV_DGRAD, flops, 0.0, stream);
    const bool mir = a.mode == 1 && a.mirror;
#define AVC_SMALL(NT_, CK_)                                                                                         \
    do {                                                                                                            \
        if (mir) hipLaunchKernelGGL((conv_small_kernel<NT_, CK_, true>), grid, block, lds, stream, a);              \
        else hipLaunchKernelGGL((conv_small_kernel<NT_, CK_, false>), grid, block, lds, stream, a);                 \
    } while (0)
    if (NT == 4 && g.CK == 16) AVC_SMALL(4, 16);
    else if (NT == 4) AVC_SMALL(4, 8);
    else if (g.CK == 16) AVC_SMALL(2, 16);
    else AVC_SMALL(2, 8);
#undef AVC_SMALL
    return (int)hipGetLastError();
}
