// Implicit-GEMM Conv1d for gfx950 on the exact-fp32 MFMA (v_mfma_f32_32x32x2_f32).
//
// One kernel serves both directions of every convolution / Linear of the
// AdaIN-VC autoencoder (reference: model.py:21-32 pad_layer + nn.Conv1d /
// nn.Linear call sites at :223-224,:241-247,:256-259,:268,:276,:304-322,:348-370):
//
//   out[b, m, t] = sum_{c, j} Wp[c, j, m] * src_ext[b, c, t*s + j]
//
//   mode 0 (forward):  src_ext = reflect-padded input, never materialised — the
//                      LDS tile loader mirrors the halo indices.
//   mode 1 (dgrad):    src_ext = zero-extended, zero-upsampled dy; weights are
//                      packed transposed + tap-flipped.  The adjoint of the
//                      reflect padding (the "fold") is applied inside the
//                      B-fragment fetch: a column near an edge adds the LDS
//                      windows of its mirror images before the MFMA, so dx is
//                      produced directly with T columns and no padded buffer.
//
// Tiling: workgroup = 4 waves (2x2), wave tile = (32*WM) x (32*WN), K-chunks of
// CK reduction channels x KS taps staged through LDS (weights arrive pre-packed
// in LDS-image order; the source tile is loaded once per chunk and serves all
// KS taps through shifted windows).  Register-staged double buffering, one
// barrier per chunk.  Epilogue fuses bias, ReLU, pixel-shuffle store, the
// residual join (identity / ceil-mode avg-pool / their adjoints) and the ReLU
// mask of the backward pass.
#include <hip/hip_runtime.h>
#include <stdlib.h>

#include "avc_common.h"
#include "avc_internal.h"

#include "conv_shared.h"
#include "conv_x3_shared.h"

// one K-chunk of MFMAs: A fragments from the packed-weight stage, B fragments as shifted windows
// of the source tile (plus the two mirror windows of the reflect-padding adjoint when MIRROR).
// A "unit" is 4 k-steps (8 reduction channels of one tap).
// MIR: 0 = no mirror window, 1 = ONE mirror window per column (cbl; a column of a sample of >= 6 frames is within pad
// of at most one edge), 2 = both windows (very short samples)
template <int WM, int WN, int MIR>
static __device__ __forceinline__ void conv_load_unit(float (&av)[4][WM], float (&bv)[4][WN], const float* Arow, const float* Xrow,
                                                      int ROW, const int (&cb)[WN], const int (&cbl)[WN], const int (&cbr)[WN]) {
    constexpr int BM = 64 * WM;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
#pragma unroll
        for (int wm = 0; wm < WM; ++wm) av[u][wm] = Arow[(2 * u) * BM + wm * 32];
        const float* xr = Xrow + (2 * u) * ROW;
#pragma unroll
        for (int wn = 0; wn < WN; ++wn) {
            float v = xr[cb[wn]];
            if (MIR == 1) v = v + xr[cbl[wn]];
            if (MIR == 2) v = v + xr[cbl[wn]] + xr[cbr[wn]];
            bv[u][wn] = v;
        }
    }
}
template <int WM, int WN, bool BF>
static __device__ __forceinline__ void conv_mma_unit(f32x16 (&acc)[WM][WN], const float (&av)[4][WM], const float (&bv)[4][WN]) {
    if constexpr (BF) {  // the unit's 8 reduction channels in ONE v_mfma_f32_32x32x8_bf16 (operands rounded here)
        avc_s16x4 ap[WM], bp[WN];
#pragma unroll
        for (int wm = 0; wm < WM; ++wm) ap[wm] = avc_pack_bf16x4(av[0][wm], av[1][wm], av[2][wm], av[3][wm]);
#pragma unroll
        for (int wn = 0; wn < WN; ++wn) bp[wn] = avc_pack_bf16x4(bv[0][wn], bv[1][wn], bv[2][wn], bv[3][wn]);
#pragma unroll
        for (int wm = 0; wm < WM; ++wm)
#pragma unroll
            for (int wn = 0; wn < WN; ++wn) acc[wm][wn] = avc_mfma_bf16(ap[wm], bp[wn], acc[wm][wn]);
        return;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int wm = 0; wm < WM; ++wm)
#pragma unroll
            for (int wn = 0; wn < WN; ++wn)
                acc[wm][wn] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u][wm], bv[u][wn], acc[wm][wn], 0, 0, 0);
}

// KSC > 0: tap count and chunk depth are compile-time (GRC = CK/8), the chunk is one straight-line
// block with the fragments of unit u+1 fetched from LDS before the MFMAs of unit u are issued
// (register double buffering) so that a lone wave per SIMD does not stall on LDS latency.
// TS (straight-line chunks only): 0 = all taps, 1 = taps 0, 2, 4, ..., 2 = taps 1, 3, ... (stride-2 dgrad: the other
// taps of a column meet the zeros of the zero-upsampled dy)
template <int WM, int WN, int MIR, int KSC, int GRC, bool BF, int TS = 0>
static __device__ __forceinline__ void conv_chunk_mma(f32x16 (&acc)[WM][WN], const float* Ab, const float* Xb, int KS, int CK,
                                                      int ROW, int h, int a_lane, const int (&cb)[WN], const int (&cbl)[WN],
                                                      const int (&cbr)[WN]) {
    constexpr int BM = 64 * WM;
    if constexpr (KSC > 0) {
        constexpr int NTAP = TS == 0 ? KSC : (TS == 1 ? (KSC + 1) / 2 : KSC / 2);
        constexpr int U = NTAP * GRC;
        constexpr int CKC = 8 * GRC;
        float av[2][4][WM], bv[2][4][WN];
        auto unit_ptrs = [&](int u, const float*& Arow, const float*& Xrow) {
            const int tap = TS == 0 ? u / GRC : 2 * (u / GRC) + (TS == 2 ? 1 : 0), g4 = u % GRC;
            Arow = Ab + (tap * CKC + 8 * g4 + h) * BM + a_lane;
            Xrow = Xb + (8 * g4 + h) * ROW + tap;
        };
        const float *Ar, *Xr;
        unit_ptrs(0, Ar, Xr);
        conv_load_unit<WM, WN, MIR>(av[0], bv[0], Ar, Xr, ROW, cb, cbl, cbr);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (u + 1 < U) {
                unit_ptrs(u + 1, Ar, Xr);
                conv_load_unit<WM, WN, MIR>(av[(u + 1) & 1], bv[(u + 1) & 1], Ar, Xr, ROW, cb, cbl, cbr);
            }
            // (pinning this order with sched_barrier(0) was measured: no gain with 4 waves/SIMD, r1 log)
            conv_mma_unit<WM, WN, BF>(acc, av[u & 1], bv[u & 1]);
        }
    } else {
        const int groups = CK >> 3;
        for (int tap = 0; tap < KS; ++tap) {
            const float* Arow = Ab + (tap * CK + h) * BM + a_lane;
            const float* Xrow = Xb + h * ROW + tap;
            for (int g4 = 0; g4 < groups; ++g4) {
                float av[4][WM], bv[4][WN];
                conv_load_unit<WM, WN, MIR>(av, bv, Arow, Xrow, ROW, cb, cbl, cbr);
                conv_mma_unit<WM, WN, BF>(acc, av, bv);
                Arow += 8 * BM;
                Xrow += 8 * ROW;
            }
        }
    }
}

// KG > 1: intra-workgroup split-K for layers that cannot fill the chip (T_l <= 32: 128-256 tiles, each
// wave a serial chain of 320 MFMAs).  KG groups of 4 waves work on the SAME output tile; group kg
// runs its own double-buffered pipeline over chunks kg, kg+KG, ... and the groups' accumulators are
// summed through LDS in a fixed order at the end (deterministic).
template <int WM, int WN, bool MIRROR, int KSC, int GRC, int KG, bool BF, bool INF = false, bool PAR = false>
__global__ void __launch_bounds__(AVC_THREADS * KG) conv_gemm_kernel(const ConvArgs a) {
    constexpr int BM = 64 * WM, BN = 64 * WN;
    constexpr int NTHREADS = AVC_THREADS * KG;
    HIP_DYNAMIC_SHARED(float, smem)
    const ConvGroup g = a.g[blockIdx.z];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);  // scalar: LDS-DMA bases must be provably wave-uniform
    const int kg = (KG > 1) ? (wave_all >> 2) : 0;
    const int wave = wave_all & 3;
    const int wave_m = wave >> 1, wave_n = wave & 1;
    const int li = lane & 31, h = lane >> 5;

    const int KS = g.KS, padL = g.padL, padR = g.padR, CK = g.CK, nchunk = g.nchunk;
    const int Tout = a.Tout;
    const ConvGeom q = conv_geom(a.mode, a.stride, Tout, KS, BN, blockIdx.x);
    const int ROW = q.ROW;
    const int m_tile0 = blockIdx.y * BM;

    const int AS = KS * CK * BM;  // floats per A stage
    const int XS = CK * ROW;      // floats per X stage
    float* As = smem + kg * 2 * AS;                 // this group's two A stages
    float* Xs = smem + KG * 2 * AS + kg * 2 * XS;   // ... and X stages

    // ---- per-lane source descriptors of the X tile: lane l owns LDS positions p = 64*j + l of every
    // row (one row = one reduction channel), so the descriptor depends on p only and each chunk's
    // loads are dword LDS-DMAs (no staging registers, no index table): offset of the element inside
    // its channel row, or -1 where the tile holds a structural zero (halo / null window / masked).
    int xoff[AVC_CONV_NJ];
#pragma unroll
    for (int j = 0; j < AVC_CONV_NJ; ++j) {
        const int p = 64 * j + lane;
        int sp = -1;
        if (p < q.ROWDATA) {
            int seg = p / q.SEG, qq = p - seg * q.SEG;
            int b = q.b0 + seg;
            int pp = q.seg_p0 + qq;
            if (b < a.B) {
                if (a.mode == 0) {
                    int v = pp - padL;
                    int r = avc_reflect(v, a.Tsrc);
                    if (r >= 0 && r < a.Tsrc) sp = (int)(b * a.x.sb + (long)r * a.x.st);
                } else {
                    int v = pp - (KS - 1);
                    if (v >= 0) {
                        int vs = v / a.stride;
                        if (vs * a.stride == v && vs < a.Tsrc) sp = (int)(b * a.x.sb + (long)vs * a.x.st);
                    }
                }
            }
        }
        xoff[j] = sp;
    }
    // both X stages start as zeros; structural zeros are never overwritten afterwards
    for (int e = tid; e < KG * 2 * XS; e += NTHREADS) smem[KG * 2 * AS + e] = 0.f;

    // ---- per-lane column bases into an LDS row
    int cb[WN], cbl[WN], cbr[WN], colb[WN], colt[WN];
    bool colv[WN];
#pragma unroll
    for (int wn = 0; wn < WN; ++wn) {
        int n = wave_n * (32 * WN) + wn * 32 + li;
        int bl, t;
        bool v;
        if (PAR) {   // (WN == 1) wave_n = parity of the wave's columns, li = slot inside the parity class
            if (Tout >= BN) {
                bl = 0;
                t = q.t0 + 2 * li + wave_n;
                v = (t < Tout) && (q.b0 < a.B);
            } else {
                const int halfT = Tout >> 1;
                bl = li / halfT;
                t = 2 * (li - bl * halfT) + wave_n;
                v = (bl < q.SPT) && (q.b0 + bl < a.B);
            }
        } else if (q.SPT == 1 && Tout >= BN) {
            bl = 0;
            t = q.t0 + n;
            v = (t < Tout) && (q.b0 < a.B);
        } else {
            bl = n / Tout;
            t = n - bl * Tout;
            v = (bl < q.SPT) && (q.b0 + bl < a.B);
        }
        colb[wn] = q.b0 + bl;
        colt[wn] = t;
        colv[wn] = v;
        int base = q.ROWDATA, bL = q.ROWDATA, bR = q.ROWDATA;
        if (v) {
            if (a.mode == 0) {
                base = bl * q.SEG + (t - q.t0) * a.stride;
            } else {
                base = bl * q.SEG + (t - q.t0) + padL;
                if (MIRROR) {
                    if (t >= 1 && t <= padL) bL = bl * q.SEG + (padL - t - q.seg_p0);
                    if (t >= Tout - 1 - padR && t <= Tout - 2) bR = bl * q.SEG + (2 * (Tout - 1) - t + padL - q.seg_p0);
                }
            }
        }
        cb[wn] = base;
        cbl[wn] = bL;
        cbr[wn] = bR;
    }

    f32x16 acc[WM][WN];
#pragma unroll
    for (int wm = 0; wm < WM; ++wm)
#pragma unroll
        for (int wn = 0; wn < WN; ++wn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[wm][wn][r] = 0.f;

    const int npieces = (KS * CK * BM) >> 8;  // 1 KiB (256 floats) per wave-instruction of the LDS DMA
    const int nj = (ROW + 63) >> 6;

    // reflect-adjoint windows are needed only by waves that own a column within pad of a sample edge
    bool use_mirror = false;
    const bool one_window = Tout >= 2 * (padL + padR) + 2;   // left-edge columns [1, padL] and right-edge columns [Tout-1-padR, Tout-2] are disjoint
    if (MIRROR) {
        bool mine = false;
#pragma unroll
        for (int wn = 0; wn < WN; ++wn) {
            mine |= (cbl[wn] != q.ROWDATA) || (cbr[wn] != q.ROWDATA);
            if (one_window && cbl[wn] == q.ROWDATA) cbl[wn] = cbr[wn];   // the column's only mirror window
        }
        use_mirror = __any(mine);
    }

    __syncthreads();  // zero fill done before the first DMA lands

    // weights: packed image == LDS image -> direct global->LDS DMA, 16 B per lane, no staging registers
    auto load_a = [&](int chunk, int buf) {
        const float* wsrc = g.wp + (long)chunk * KS * CK * a.Mp + m_tile0;
        float* Ad = As + buf * AS;
        for (int piece = wave; piece < npieces; piece += 4) {
            int f = piece * 256 + lane * 4;
            int row = f / BM, col = f - row * BM;
            avc_glds16(wsrc + (long)row * a.Mp + col, Ad + piece * 256);
        }
    };
    // source tile: wave w stages rows w, w+4, ... of the chunk, 64 positions per DMA instruction
    auto load_x = [&](int chunk, int buf) {
        float* Xd = Xs + buf * XS;
        for (int r = wave; r < CK; r += 4) {
            const int c = chunk * CK + r;
            if (c < a.Cred) {
                const long coff = (a.x.ps == 1) ? (long)c * a.x.sc : (long)(c / a.x.ps) * a.x.sc + (c % a.x.ps);
                const float* src = a.x.ptr + coff;
#pragma unroll
                for (int j = 0; j < AVC_CONV_NJ; ++j)
                    if (j < nj && xoff[j] >= 0) avc_glds4(src + xoff[j], Xd + r * ROW + 64 * j);
            } else {  // channel padding of the last chunk
#pragma unroll
                for (int j = 0; j < AVC_CONV_NJ; ++j)
                    if (j < nj && 64 * j + lane < ROW) Xd[r * ROW + 64 * j + lane] = 0.f;
            }
        }
    };

    if (kg < nchunk) {
        load_a(kg, 0);
        load_x(kg, 0);
    }
    __syncthreads();

    const int a_lane = wave_m * (32 * WM) + li;
    const int nit = (nchunk + KG - 1) / KG;
    for (int it = 0; it < nit; ++it) {
        const int chunk = it * KG + kg;
        const bool more = (chunk + KG < nchunk);
        if (more && !((a.dbg & 1) && it >= 1)) {
            load_a(chunk + KG, (it + 1) & 1);  // both land while this chunk is multiplied; drained at the barrier
            load_x(chunk + KG, (it + 1) & 1);
        }
        const float* Ab = As + (it & 1) * AS;
        const float* Xb = Xs + (it & 1) * XS;
        if ((a.dbg & 2) || chunk >= nchunk) {
        } else if constexpr (PAR) {   // even columns: taps 0, 2, 4; odd columns: taps 1, 3 (k = 5, padL = 2)
            if (wave_n == 0) {
                if (MIRROR && use_mirror && one_window) conv_chunk_mma<WM, WN, 1, KSC, GRC, BF, 1>(acc, Ab, Xb, KS, CK, ROW, h, a_lane, cb, cbl, cbr);
                else if (MIRROR && use_mirror) conv_chunk_mma<WM, WN, 2, KSC, GRC, BF, 1>(acc, Ab, Xb, KS, CK, ROW, h, a_lane, cb, cbl, cbr);
                else conv_chunk_mma<WM, WN, 0, KSC, GRC, BF, 1>(acc, Ab, Xb, KS, CK, ROW, h, a_lane, cb, cbl, cbr);
            } else {
                if (MIRROR && use_mirror && one_window) conv_chunk_mma<WM, WN, 1, KSC, GRC, BF, 2>(acc, Ab, Xb, KS, CK, ROW, h, a_lane, cb, cbl, cbr);
                else if (MIRROR && use_mirror) conv_chunk_mma<WM, WN, 2, KSC, GRC, BF, 2>(acc, Ab, Xb, KS, CK, ROW, h, a_lane, cb, cbl, cbr);
                else conv_chunk_mma<WM, WN, 0, KSC, GRC, BF, 2>(acc, Ab, Xb, KS, CK, ROW, h, a_lane, cb, cbl, cbr);
            }
        } else if (MIRROR && use_mirror && one_window)   // wave-uniform: only waves owning a column within pad of a sample edge
            conv_chunk_mma<WM, WN, 1, KSC, GRC, BF>(acc, Ab, Xb, KS, CK, ROW, h, a_lane, cb, cbl, cbr);
        else if (MIRROR && use_mirror)
            conv_chunk_mma<WM, WN, 2, KSC, GRC, BF>(acc, Ab, Xb, KS, CK, ROW, h, a_lane, cb, cbl, cbr);
        else if constexpr (KSC < 0) {
            // grouped launch of layers with different tap counts (the conv bank, k = 1..8): the workgroup's (taps,
            // chunk depth) pair is uniform, so each pair gets its own straight-line chunk
            switch (KS * 8 + (CK >> 3)) {
                case 1 * 8 + 4: conv_chunk_mma<WM, WN, 0, 1, 4, BF>(acc, Ab, Xb, KS, CK, ROW, h, a_lane, cb, cbl, cbr); break;
                case 2 * 8 + 2: conv_chunk_mma<WM, WN, 0, 2, 2, BF>(acc, Ab, Xb, KS, CK, ROW, h, a_lane, cb, cbl, cbr); break;
                case 3 * 8 + 2: conv_chunk_mma<WM, WN, 0, 3, 2, BF>(acc, Ab, Xb, KS, CK, ROW, h, a_lane, cb, cbl, cbr); break;
                case 4 * 8 + 1: conv_chunk_mma<WM, WN, 0, 4, 1, BF>(acc, Ab, Xb, KS, CK, ROW, h, a_lane, cb, cbl, cbr); break;
                case 5 * 8 + 1: conv_chunk_mma<WM, WN, 0, 5, 1, BF>(acc, Ab, Xb, KS, CK, ROW, h, a_lane, cb, cbl, cbr); break;
                case 6 * 8 + 1: conv_chunk_mma<WM, WN, 0, 6, 1, BF>(acc, Ab, Xb, KS, CK, ROW, h, a_lane, cb, cbl, cbr); break;
                case 7 * 8 + 1: conv_chunk_mma<WM, WN, 0, 7, 1, BF>(acc, Ab, Xb, KS, CK, ROW, h, a_lane, cb, cbl, cbr); break;
                case 8 * 8 + 1: conv_chunk_mma<WM, WN, 0, 8, 1, BF>(acc, Ab, Xb, KS, CK, ROW, h, a_lane, cb, cbl, cbr); break;
                default: conv_chunk_mma<WM, WN, 0, 0, 0, BF>(acc, Ab, Xb, KS, CK, ROW, h, a_lane, cb, cbl, cbr);
            }
        } else
            conv_chunk_mma<WM, WN, 0, KSC, GRC, BF>(acc, Ab, Xb, KS, CK, ROW, h, a_lane, cb, cbl, cbr);
        if (!(a.dbg & 4)) __syncthreads();
    }

    if (KG > 1) {  // fixed-order sum of the groups' partial tiles through the (now free) stage memory
        float* red = smem + ((kg > 0 ? kg - 1 : 0) * 4 + wave) * (WM * WN * 16 * 64) + lane;
        if (kg > 0) {
#pragma unroll
            for (int wm = 0; wm < WM; ++wm)
#pragma unroll
                for (int wn = 0; wn < WN; ++wn)
#pragma unroll
                    for (int r = 0; r < 16; ++r) red[((wm * WN + wn) * 16 + r) * 64] = acc[wm][wn][r];
        }
        __syncthreads();
        if (kg > 0) return;
#pragma unroll
        for (int k2 = 1; k2 < KG; ++k2) {
            const float* rk = smem + ((k2 - 1) * 4 + wave) * (WM * WN * 16 * 64) + lane;
#pragma unroll
            for (int wm = 0; wm < WM; ++wm)
#pragma unroll
                for (int wn = 0; wn < WN; ++wn)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[wm][wn][r] += rk[((wm * WN + wn) * 16 + r) * 64];
        }
    }

    // ---- epilogue
    if (a.dbg & 8) return;
    if constexpr (INF && WM == 1 && WN == 1) {  // (its own instantiation: the row statistics cost ~90 registers)
        {
            // Fused InstanceNorm (+ AdaIN affine + ReLU + residual): the tile holds whole (b, m) rows
            // (Tout = 16 / 32: inside one wave's 32 columns; 64: the two wave_n halves, joined through LDS).
            // Two-pass statistics like the row kernel; the normalise step is the shared in_xhat / in_preact.
            const bool v = colv[0];
            const int b = colb[0], t = colt[0];
            const int G = Tout < 32 ? Tout : 32;
            const float invT = 1.0f / (float)Tout;
            float* red = smem + KG * 4 * 1024;  // behind the split-K exchange area: [pass][wave_m][wave_n][32 rows]
            float val[16], mean[16], rstd[16];
            auto row_total = [&](float (&x)[16], int pass) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float s = x[r];
                    for (int o = G >> 1; o >= 1; o >>= 1) s += __shfl_xor(s, o);
                    x[r] = s;
                }
                if (Tout == 64) {
                    float* rp = red + pass * 128 + wave_m * 64;
                    if (li == 0) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) rp[wave_n * 32 + h * 16 + r] = x[r];
                    }
                    __syncthreads();
#pragma unroll
                    for (int r = 0; r < 16; ++r) x[r] = rp[h * 16 + r] + rp[32 + h * 16 + r];
                }
            };
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m_tile0 + wave_m * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                val[r] = acc[0][0][r] + ((g.bias && m < a.M) ? g.bias[m] : 0.f);
                mean[r] = v ? val[r] : 0.f;
            }
            row_total(mean, 0);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                mean[r] *= invT;
                const float d = v ? val[r] - mean[r] : 0.f;
                rstd[r] = d * d;
            }
            row_total(rstd, 1);
#pragma unroll
            for (int r = 0; r < 16; ++r) rstd[r] = 1.0f / sqrtf(rstd[r] * invT + AVC_IN_EPS);  // biased variance
            if (!v) return;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m_tile0 + wave_m * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (m >= a.M) continue;
                const long o = (long)b * a.ob + (long)m * a.oc + (long)t * a.ot;
                g.out[o] = val[r];
                float gamma = 1.f, beta = 0.f;
                if (a.in_cond) {
                    const float* cr = a.in_cond + (long)b * a.in_cond_sb + a.in_cond_off;
                    beta = cr[m];
                    gamma = cr[a.in_C + m];
                }
                float w = fmaxf(in_preact(in_xhat(val[r], mean[r], rstd[r]), gamma, beta), 0.f);
                if (a.res_mode != AVC_RES_NONE) w += conv_load_res(a, g.res, b, m, t);
                a.in_out[o] = w;
                if (t == 0) {
                    a.in_mean[(long)b * a.in_C + m] = mean[r];
                    a.in_rstd[(long)b * a.in_C + m] = rstd[r];
                }
            }
            return;
        }
    }
#pragma unroll
    for (int wn = 0; wn < WN; ++wn) {
        if (!colv[wn]) continue;
#pragma unroll
        for (int wm = 0; wm < WM; ++wm)
            conv_store_frag(a, g, acc[wm][wn], m_tile0 + wave_m * (32 * WM) + wm * 32, h, colb[wn], colt[wn]);
    }
}

// --------------------------------------------------------------------------
// weight packing: W[Cout][Cin][KS] (state_dict layout) -> LDS-image order
//   fwd  : Wp[chunk][j][r][m]  = W[m][chunk*CK + r][j]
//   dgrad: Wp[chunk][j][r][m]  = W[chunk*CK + r][m][KS-1-j]   (transposed, tap-flipped)
// Several source tensors of identical shape can be stacked along the forward
// output-channel axis (the 12 AdaIN affine Linears, the mu/log_sigma heads).
// --------------------------------------------------------------------------

struct PackBatch {
    PackArgs a[AVC_PACK_BATCH];
};

static __device__ __forceinline__ void pack_one(const PackArgs& p, long first, long stride);

// several layers per launch: blockIdx.y selects the layer descriptor
__global__ void __launch_bounds__(AVC_THREADS) pack_weight_batch_kernel(const PackBatch b) {
    pack_one(b.a[blockIdx.y], (long)blockIdx.x * AVC_THREADS + threadIdx.x, (long)gridDim.x * AVC_THREADS);
}

__global__ void __launch_bounds__(AVC_THREADS) pack_weight_kernel(const PackArgs p) {
    pack_one(p, (long)blockIdx.x * AVC_THREADS + threadIdx.x, (long)gridDim.x * AVC_THREADS);
}

static __device__ __forceinline__ void pack_one(const PackArgs& p, long first, long stride) {
    if (p.rs == 2) {
        avc_pack_x3_one(p, first, stride);
        return;
    }
    if (p.rs) {  // register-stationary image (conv_rs.hip): ks = 4q + u = c2 * KS + j, c = 2 * c2 + (lane >> 5), m = 32 * slab + (lane & 31)
        const long total = (long)p.rs_nslab * p.rs_nq * 256;
        const int M = p.dgrad ? p.Cin : p.Cout, Cred = p.dgrad ? p.Cout : p.Cin;
        const int NKS = p.KS * ((Cred + 1) / 2);
        const float* w = p.src[0];
        for (long e = first; e < total + 128; e += stride) {
            float v = 0.f;   // (the last 128 floats: the zero block the DMA reads structural zeros from)
            if (e < total) {
                const int u = (int)(e & 3), lane = (int)((e >> 2) & 63);
                const long rest = e >> 8;
                const int q = (int)(rest % p.rs_nq), slab = (int)(rest / p.rs_nq);
                const int ks = 4 * q + u;
                const int c2 = ks / p.KS, j = ks - c2 * p.KS;
                const int c = 2 * c2 + (lane >> 5), m = 32 * slab + (lane & 31);
                if (ks < NKS && m < M && c < Cred)
                    v = p.dgrad ? w[((long)c * p.Cin + m) * p.KS + (p.KS - 1 - j)] : w[((long)m * p.Cin + c) * p.KS + j];
            }
            p.dst[e] = v;
        }
        return;
    }
    long total = (long)p.nchunk * p.KS * p.CK * p.Mp;
    for (long e = first; e < total; e += stride) {
        int m = (int)(e % p.Mp);
        long rest = e / p.Mp;
        int r = (int)(rest % p.CK);
        rest /= p.CK;
        int j = (int)(rest % p.KS);
        int chunk = (int)(rest / p.KS);
        int red = chunk * p.CK + r;
        float v = 0.f;
        if (!p.dgrad) {
            if (m < p.M && red < p.Cin) {
                int s = m / p.rows_per_src, mm = m - s * p.rows_per_src;
                v = p.src[s][((long)mm * p.Cin + red) * p.KS + j];
            }
        } else {
            if (m < p.M && red < p.Cout) {
                int s = red / p.rows_per_src, rr = red - s * p.rows_per_src;
                v = p.src[s][((long)rr * p.Cin + m) * p.KS + (p.KS - 1 - j)];
            }
        }
        p.dst[e] = v;
    }
}

// --------------------------------------------------------------------------
// host side
// --------------------------------------------------------------------------
static int g_conv_ck5 = 8;  // chunk depth of the k >= 4 layers at the op level (micro-benchmark knob, avc_set_tuning)
void avc_set_conv_ck5(int ck) { g_conv_ck5 = (ck == 8 || ck == 16 || ck == 32) ? ck : 8; }
extern "C" int avc_conv_ck(int KS) { return KS >= 4 ? g_conv_ck5 : (KS >= 2 ? 16 : 32); }

// launch-heuristic thresholds (avc_set_tuning "tile_thr11" / "tile_thr21" / "ck16_wgs" / "ck32_wgs" / "kg_wgs"; measured defaults)
static long g_tile_thr11 = 8192, g_tile_thr21 = 4096, g_ck16_wgs = 256, g_ck32_wgs = 256, g_kg_wgs = 256;   // r2 sweep (profiles/r02_tune_sweeps.log)
void avc_set_conv_heuristic(int which, long v) {
    if (which == 0) g_tile_thr11 = v;
    if (which == 1) g_tile_thr21 = v;
    if (which == 2) g_ck16_wgs = v;
    if (which == 3) g_ck32_wgs = v;
    if (which == 4) g_kg_wgs = v;
}
// tile choice shared by the launcher and the plan (which sizes CK from it)
int avc_conv_pick_tile(int Mp, int B, int Tout, int ngroups, int Kred) {
    // measured on MI355X (profiles/r01_conv_micro_*.log): at equal work the 64x64 tile beats 128x64
    // and 128x128 (more resident waves hide the per-chunk LDS/DMA latency; the fp32 MFMA needs no
    // bigger tile for operand reuse), so take the smallest tile unless the grid gets very large
    auto ntn = [&](int BN) { return Tout >= BN ? (long)B * avc_cdiv(Tout, BN) : (long)avc_cdiv(B, BN / Tout); };
    long t11 = (long)(Mp / 64) * ntn(64) * ngroups;
    long t21 = (long)(Mp / 128) * ntn(64) * ngroups;
    // r2 sweeps (profiles/r02_tune_sweeps.log): launches whose workgroups carry little reduction work -- the grouped bank
    // (k = 1..8 over 80 mel rows) and 1x1 convs over <= 256 channels -- keep gaining from 64x64 tiles up to ~8k
    // workgroups (their cost is the epilogue: more, smaller workgroups overlap it better); the k = 5, 128-channel convs
    // switch to 128x64 beyond ~4k (B = 1024 inference: 7.54 vs 7.66 ms)
    const long thr11 = (ngroups > 1 || (Kred > 0 && Kred <= 256)) ? g_tile_thr11 : (g_tile_thr11 < 4095 ? g_tile_thr11 : 4095);
    if (t11 <= thr11) return 11;
    if (t21 <= g_tile_thr21) return 21;
    return 22;
}
long avc_conv_num_wgs(int tile, int Mp, int B, int Tout, int ngroups) {
    int BM = (tile / 10 == 1) ? 64 : 128, BN = (tile % 10 == 1) ? 64 : 128;
    long ntn = Tout >= BN ? (long)B * avc_cdiv(Tout, BN) : (long)avc_cdiv(B, BN / Tout);
    return (long)(Mp / BM) * ntn * ngroups;
}
// K-chunk depth: layers that cannot put two workgroups on every CU are latency-bound per chunk
// (global -> LDS round trip vs. ~1.3k MFMA cycles), so they take twice the channels per chunk
int avc_conv_ck_for(int KS, long wgs, int mode, int stride, int Tout, int tile) {
    int ck = avc_conv_ck(KS);
    if (KS >= 4 && wgs <= g_ck16_wgs) {  // measured (r1 conv micro): CK=16 wins up to 2 workgroups per CU, loses beyond
        int BN = (tile % 10 == 1) ? 64 : 128;
        ConvGeom q = conv_geom(mode, stride, Tout, KS, BN, 0);
        if (q.ROW <= 64 * AVC_CONV_NJ) ck = 16;
        // at most one workgroup per CU: the forward kernel gains another ~6 % from four chunks of 32
        // (r1 sweep: T_l = 16/32 forward 26.0 -> 24.5 us; the dgrad variant does not move)
        int BM = (tile / 10 == 1) ? 64 : 128;
        if (KS == 5 && mode == 0 && wgs <= g_ck32_wgs && tile != 11 && q.ROW <= 64 * AVC_CONV_NJ && 2 * (size_t)(KS * 32 * BM + 32 * q.ROW) * 4 <= 144 * 1024) ck = 32;
    }
    return ck;
}

static size_t conv_lds_bytes(const ConvArgs& a, int BM, int BN) {
    size_t worst = 0;
    for (int gi = 0; gi < a.ngroups; ++gi) {
        ConvGeom q = conv_geom(a.mode, a.stride, a.Tout, a.g[gi].KS, BN, 0);
        size_t AS = (size_t)a.g[gi].KS * a.g[gi].CK * BM, XS = (size_t)a.g[gi].CK * q.ROW;
        size_t bytes = (2 * AS + 2 * XS) * 4;
        worst = bytes > worst ? bytes : worst;
    }
    return worst + 16;
}

static int conv_ntiles_n(const ConvArgs& a, int BN) {
    if (a.Tout >= BN) return a.B * avc_cdiv(a.Tout, BN);
    int spt = BN / a.Tout;
    return avc_cdiv(a.B, spt);
}

template <int KG, bool BF>
static void conv_launch_infuse(const ConvArgs& a, int fast, dim3 grid, dim3 block, size_t lds, hipStream_t stream) {
    if (fast == 1) hipLaunchKernelGGL((conv_gemm_kernel<1, 1, false, 5, 1, KG, BF, true>), grid, block, lds, stream, a);
    else if (fast == 2) hipLaunchKernelGGL((conv_gemm_kernel<1, 1, false, 5, 2, KG, BF, true>), grid, block, lds, stream, a);
    else hipLaunchKernelGGL((conv_gemm_kernel<1, 1, false, 0, 0, KG, BF, true>), grid, block, lds, stream, a);
}

template <int WM, int WN, int KG, bool BF>
static void conv_launch_variant(const ConvArgs& a, bool mir, int fast, dim3 grid, dim3 block, size_t lds, hipStream_t stream) {
    if (mir && fast == 1) hipLaunchKernelGGL((conv_gemm_kernel<WM, WN, true, 5, 1, KG, BF>), grid, block, lds, stream, a);
    else if (mir && fast == 2) hipLaunchKernelGGL((conv_gemm_kernel<WM, WN, true, 5, 2, KG, BF>), grid, block, lds, stream, a);
    else if (mir) hipLaunchKernelGGL((conv_gemm_kernel<WM, WN, true, 0, 0, KG, BF>), grid, block, lds, stream, a);
    else if (fast == 1) hipLaunchKernelGGL((conv_gemm_kernel<WM, WN, false, 5, 1, KG, BF>), grid, block, lds, stream, a);
    else if (fast == 2) hipLaunchKernelGGL((conv_gemm_kernel<WM, WN, false, 5, 2, KG, BF>), grid, block, lds, stream, a);
    else if (fast == 4 && KG == 1) hipLaunchKernelGGL((conv_gemm_kernel<WM, WN, false, 5, 4, 1, BF>), grid, block, lds, stream, a);
    else if (fast == -1 && KG == 1) hipLaunchKernelGGL((conv_gemm_kernel<WM, WN, false, -1, 0, 1, BF>), grid, block, lds, stream, a);
    else if (fast == 14) hipLaunchKernelGGL((conv_gemm_kernel<WM, WN, false, 1, 4, KG, BF>), grid, block, lds, stream, a);   // 1x1, 32-channel chunks
    else hipLaunchKernelGGL((conv_gemm_kernel<WM, WN, false, 0, 0, KG, BF>), grid, block, lds, stream, a);
}

// stride-2 dgrad with one column parity per wave (half the MFMAs of the zero-upsampled correlation)
template <int KG, bool BF>
static void conv_launch_par(const ConvArgs& a, bool mir, int fast, dim3 grid, dim3 block, size_t lds, hipStream_t stream) {
    if (mir && fast == 1) hipLaunchKernelGGL((conv_gemm_kernel<1, 1, true, 5, 1, KG, BF, false, true>), grid, block, lds, stream, a);
    else if (mir) hipLaunchKernelGGL((conv_gemm_kernel<1, 1, true, 5, 2, KG, BF, false, true>), grid, block, lds, stream, a);
    else if (fast == 1) hipLaunchKernelGGL((conv_gemm_kernel<1, 1, false, 5, 1, KG, BF, false, true>), grid, block, lds, stream, a);
    else hipLaunchKernelGGL((conv_gemm_kernel<1, 1, false, 5, 2, KG, BF, false, true>), grid, block, lds, stream, a);
}
static int g_dgrad_par = 1;   // avc_set_tuning("dgrad_par", 0): stride-2 dgrad multiplies all five taps of the zero-upsampled dy
void avc_set_dgrad_par(int on) { g_dgrad_par = on ? 1 : 0; }

static int g_bank_switch = 1;   // avc_set_tuning("bank_switch", 0): the grouped bank launch on the generic (run-time taps) chunk loop
void avc_set_bank_switch(int on) { g_bank_switch = on ? 1 : 0; }

// ablation bits of scripts/conv_ablate.py (timing experiments; results are wrong by construction when set)
static int g_conv_ablation = 0;
void avc_set_conv_ablation(int bits) { g_conv_ablation = bits; }
int avc_conv_ablation_bits() { return g_conv_ablation; }

// returns 0 on success, negative on unsupported geometry
int avc_launch_conv(const ConvArgs& a_in, hipStream_t stream, int force_tile) {
    if (a_in.rs == 2 || force_tile == 97) return avc_launch_conv_x3(a_in, stream);
    if (a_in.rs || force_tile == 99) return avc_launch_conv_rs(a_in, stream);
    if ((force_tile == 0 || force_tile == 98) && !g_conv_ablation && avc_conv_small_eligible(a_in, force_tile == 98)) return avc_launch_conv_small(a_in, stream);
    if (force_tile == 98) return -8;
    ConvArgs a = a_in;
    a.dbg = g_conv_ablation;
    if (a.ngroups < 1 || a.ngroups > AVC_MAX_GROUPS) return -1;
    if (a.Mp % 128 != 0) return -2;
    for (int gi = 0; gi < a.ngroups; ++gi)
        if (a.mode == 0 && (a.g[gi].padL >= a.Tsrc || a.g[gi].padR >= a.Tsrc)) return -6;  // reference: "Padding size should be less than ..."
    int tile = force_tile == 12 ? 11 : force_tile;  // (the 64x128 tile measured no better than 64x64 and was dropped)
    if (tile == 0) tile = avc_conv_pick_tile(a.Mp, a.B, a.Tout, a.ngroups, a.Cred * a.g[0].KS);
    int BM = (tile / 10 == 1) ? 64 : 128;
    int BN = (tile % 10 == 1) ? 64 : 128;
    for (int gi = 0; gi < a.ngroups; ++gi) {
        ConvGeom q = conv_geom(a.mode, a.stride, a.Tout, a.g[gi].KS, BN, 0);
        if (a.g[gi].CK % 8 != 0) return -2;
        if (q.ROW > 64 * AVC_CONV_NJ) return -3;
    }
    if (a.in_fuse && !(tile == 11 && a.mode == 0 && a.ops == 1 && a.ngroups == 1 && a.ot == 1 && (a.Tout == 16 || a.Tout == 32 || a.Tout == 64) &&
                       a.g[0].out && a.in_out && a.in_mean && a.in_rstd))
        return -7;
    size_t lds = conv_lds_bytes(a, BM, BN);
    dim3 grid(conv_ntiles_n(a, BN), a.Mp / BM, a.ngroups);
    // split-K groups: only where the grid leaves CUs or SIMD slots idle (<= 1 workgroup per CU)
    int kgroups = 1;
    if (tile == 11 && a.ngroups == 1 && (long)grid.x * grid.y <= g_kg_wgs && a.g[0].nchunk >= 4 && 2 * lds <= 160 * 1024 && !a.dbg) kgroups = 2;
    lds *= kgroups;
    if (a.in_fuse) {  // row-statistics exchange area behind the split-K exchange area
        size_t need = ((size_t)kgroups * 4096 + 256) * 4;
        lds = lds > need ? lds : need;
    }
    if (lds > 160 * 1024) return -5;
    dim3 block(AVC_THREADS * kgroups);
    double flops = 0;
    for (int gi = 0; gi < a.ngroups; ++gi)
        flops += 2.0 * a.M * a.Cred * a.g[gi].KS * (double)a.B * (a.mode == 0 ? a.Tout : a.Tsrc);
    ProfScope ps(a.mode == 0 ? AVC_K_CONV_FWD : AVC_K_CONV_DGRAD, flops, 0.0, stream);
    const bool mir = a.mode == 1 && a.mirror;
    // the model's kernel_size (5) with the chunk depths the plan uses gets straight-line chunks
    const int fast = (a.ngroups == 1 && a.g[0].KS == 5) ? (a.g[0].CK == 8 ? 1 : (a.g[0].CK == 16 ? 2 : (a.g[0].CK == 32 ? 4 : 0)))
                     : ((a.ngroups > 1 && a.mode == 0 && g_bank_switch) ? -1
                        : ((a.ngroups == 1 && a.g[0].KS == 1 && a.g[0].CK == 32 && g_bank_switch) ? 14 : 0));
    const bool bf = a.bf16 == AVC_COMPUTE_BF16;
    a.par = g_dgrad_par && a.mode == 1 && a.stride == 2 && tile == 11 && a.ngroups == 1 && (fast == 1 || fast == 2) && !a.in_fuse && !a.dbg &&
            a.g[0].padL == 2 && (a.Tout >= 64 || (a.Tout % 2 == 0 && 64 % a.Tout == 0));
#define AVC_LAUNCH_CONV(WM_, WN_, KG_)                                                                             \
    do {                                                                                                           \
        if (bf) conv_launch_variant<WM_, WN_, KG_, true>(a, mir, fast, grid, block, lds, stream);                  \
        else conv_launch_variant<WM_, WN_, KG_, false>(a, mir, fast, grid, block, lds, stream);                    \
    } while (0)
    if (a.par) {
        if (kgroups == 2) { if (bf) conv_launch_par<2, true>(a, mir, fast, grid, block, lds, stream); else conv_launch_par<2, false>(a, mir, fast, grid, block, lds, stream); }
        else { if (bf) conv_launch_par<1, true>(a, mir, fast, grid, block, lds, stream); else conv_launch_par<1, false>(a, mir, fast, grid, block, lds, stream); }
    } else if (a.in_fuse) {  // (validated above: 64x64 tile, forward)
        const int f = fast == 4 ? 0 : fast;
        if (kgroups == 2) { if (bf) conv_launch_infuse<2, true>(a, f, grid, block, lds, stream); else conv_launch_infuse<2, false>(a, f, grid, block, lds, stream); }
        else { if (bf) conv_launch_infuse<1, true>(a, f, grid, block, lds, stream); else conv_launch_infuse<1, false>(a, f, grid, block, lds, stream); }
    } else if (tile == 22) AVC_LAUNCH_CONV(2, 2, 1);
    else if (tile == 21) AVC_LAUNCH_CONV(2, 1, 1);
    else if (kgroups == 2) AVC_LAUNCH_CONV(1, 1, 2);
    else AVC_LAUNCH_CONV(1, 1, 1);
#undef AVC_LAUNCH_CONV
    return (int)hipGetLastError();
}

long avc_pack_total(const PackArgs& p) {
    if (p.rs == 2) return (long)p.nchunk * x3_arows(p.KS) * p.Mp * 4;
    return p.rs ? (long)p.rs_nslab * p.rs_nq * 256 + 128 : (long)p.nchunk * p.KS * p.CK * p.Mp;
}

int avc_launch_pack_batch(const PackArgs* ps, int n, hipStream_t stream) {
    for (int i = 0; i < n; i += AVC_PACK_BATCH) {
        PackBatch b;
        int m = n - i < AVC_PACK_BATCH ? n - i : AVC_PACK_BATCH;
        long maxtotal = 1, bytes = 0;
        for (int k = 0; k < m; ++k) {
            b.a[k] = ps[i + k];
            long t = avc_pack_total(ps[i + k]);
            maxtotal = t > maxtotal ? t : maxtotal;
            bytes += 8 * t;
        }
        int blocks = (int)((maxtotal + AVC_THREADS * 4 - 1) / (AVC_THREADS * 4));
        if (blocks > 256) blocks = 256;
        ProfScope ps_(AVC_K_PACK, 0.0, (double)bytes, stream);
        hipLaunchKernelGGL(pack_weight_batch_kernel, dim3(blocks, m), dim3(AVC_THREADS), 0, stream, b);
    }
    return (int)hipGetLastError();
}

int avc_launch_pack(const PackArgs& p, hipStream_t stream) {
    long total = avc_pack_total(p);
    int blocks = (int)((total + AVC_THREADS * 4 - 1) / (AVC_THREADS * 4));
    if (blocks < 1) blocks = 1;
    if (blocks > 4096) blocks = 4096;
    ProfScope ps(AVC_K_PACK, 0.0, 8.0 * total, stream);
    hipLaunchKernelGGL(pack_weight_kernel, dim3(blocks), dim3(AVC_THREADS), 0, stream, p);
    return (int)hipGetLastError();
}
