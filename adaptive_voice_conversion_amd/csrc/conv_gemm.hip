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
// Tiling: workgroup = 4 waves (2x2), wave tile = (32*WM) rows x 32 columns, K-chunks of
// CK reduction channels x KS taps staged through LDS by direct global->LDS DMA
// (weights arrive pre-packed in LDS-image order; the source tile is loaded once
// per chunk and serves all KS taps through shifted windows).  Both operands sit in
// LDS with the four k-steps of a lane contiguous: one ds_read_b128 per operand
// per four MFMAs.  Double-buffered stages, one barrier per chunk.  Epilogue fuses bias, ReLU, pixel-shuffle store, the
// residual join (identity / ceil-mode avg-pool / their adjoints) and the ReLU
// mask of the backward pass.
#include <hip/hip_runtime.h>
#include <stdlib.h>

#include "avc_common.h"
#include "avc_internal.h"

#include "conv_shared.h"
#include "conv_x3_shared.h"

// LDS images (floats), per stage:
//   A  [tap][unit][h][BM rows][u]   = the packed weight image (AVC_IMG_K4) of this row tile, copied by 16-byte LDS-DMA
//   X  [unit][h][ROW positions][u]  = the source tile; reduction channel of (unit, h, u) = 8 unit + 2 u + h
// so the four k-steps u = 0..3 of one "unit" (8 reduction channels of one tap: four v_mfma_f32_32x32x2_f32, lane half h
// supplies k = h) are 16 contiguous bytes for every lane of both operands: ONE ds_read_b128 per operand per four MFMAs
// (round 2 issued four ds_read_b32 each; the MFMA issue-rate probe, profiles/r02_mfma_probe.log, says reads set the rate).
// A tap is a 16-byte shift of the X address, so all KS taps are served from the one tile.
// MIR: 0 = no mirror window, 1 = ONE mirror window per column (a column of a sample of >= 2 (padL + padR) + 2 frames is within
// pad of at most one edge), 2 = both windows (very short samples)
// LDS stages per operand: 2 = the next chunk is issued while this one multiplies (the product build); 3 = prefetch distance two with a
// partial s_waitcnt vmcnt(n) + bare s_barrier -- built, correct, and measured SLOWER on MI355X (k = 5, T = 128 forward layer 71.9 vs 67.5 us,
// train step 6.93 vs 6.51 ms, profiles/r03_conv_ablate.log): the cost of the LDS-DMA stream is not exposed latency.
#ifndef AVC_CONV_STAGES
#define AVC_CONV_STAGES 2
#endif
// ... of the bf16 pair-storage instances (BF == 2): their chunks hold 1/8 of the matrix-pipe time of an fp32 chunk, the LDS-DMA round trip
// is what a workgroup waits for
#ifndef AVC_CONV_STAGES_BH
#define AVC_CONV_STAGES_BH 2
#endif
// (round 6: three stages for the 1x1 / 32-dword-channel chunk instance ALONE -- the 1104 -> 128 in_conv: 18 chunks of 24 KB with 0.12 us of
// products each -- measured slower too: 39.7 vs 35.5 us at the op level, 2.46 vs 2.37 ms per step, profiles/r06_conv_1x1_three_stages.log)

// s_waitcnt vmcnt(n): at most n of this wave's vector-memory operations (here: LDS-DMA loads, which complete in issue order)
// still outstanding; lgkmcnt / expcnt untouched.  gfx9 encoding: vmcnt = simm16[15:14] : simm16[3:0].
static __device__ __forceinline__ void conv_wait_dma(int n) {
#define AVC_WAIT_VM(k) __builtin_amdgcn_s_waitcnt(((k) & 15) | (((k) >> 4) << 14) | (7 << 4) | (15 << 8))
    switch (n) {
        case 0: AVC_WAIT_VM(0); break;
        case 1: AVC_WAIT_VM(1); break;
        case 2: AVC_WAIT_VM(2); break;
        case 3: AVC_WAIT_VM(3); break;
        case 4: AVC_WAIT_VM(4); break;
        case 5: AVC_WAIT_VM(5); break;
        case 6: AVC_WAIT_VM(6); break;
        case 7: AVC_WAIT_VM(7); break;
        case 8: AVC_WAIT_VM(8); break;
        case 9: AVC_WAIT_VM(9); break;
        case 10: AVC_WAIT_VM(10); break;
        case 11: AVC_WAIT_VM(11); break;
        case 12: AVC_WAIT_VM(12); break;
        default: AVC_WAIT_VM(12); break;   // (more were issued: waiting for all but 12 is stricter than needed, still correct)
    }
#undef AVC_WAIT_VM
}
// workgroup barrier WITHOUT draining the vector-memory counter: __syncthreads() (and the compiler's own handling of s_barrier on
// gfx9) waits for vmcnt(0), i.e. for the chunk that was issued a moment ago.  LDS is written only by the DMAs (vmcnt) and read
// by ds_read (consumed by the MFMAs before this point), so conv_wait_dma + a bare s_barrier is the complete synchronisation.
static __device__ __forceinline__ void conv_bare_barrier() {
#ifdef AVC_EMU
    emu::block_barrier();
#else
    asm volatile("s_barrier" ::: "memory");
#endif
}

// BF: 0 = exact fp32, 1 = fp32 storage with operands rounded to bf16 here, 2 = bf16 PAIR storage (bf16_pairs.h): the 16 bytes a lane
// reads ARE its 8 k-values of one v_mfma_f32_32x32x16_bf16 (dword u = the bf16 pair of reduction channels 2 dc, 2 dc + 1, dc = 8 unit +
// 2 u + h).  The reflect-adjoint windows of the input gradient cannot be summed as packed pairs, so with BF == 2 each active window is
// one more MFMA against the same weight fragment (NW operand slots per column fragment).
template <int MIR, int BF>
struct ConvNW {
    static constexpr int value = (BF == 2) ? 1 + MIR : 1;
};
template <int WM, int WN, int MIR, int BF, int NB>   // NB = WN * NW (deduced)
static __device__ __forceinline__ void conv_load_unit(f32x4 (&av)[WM], f32x4 (&bv)[NB], const float* Ap, const float* Xp,
                                                      const int (&cb4)[WN], const int (&cbl4)[WN], const int (&cbr4)[WN]) {
    constexpr int NW = ConvNW<MIR, BF>::value;
    static_assert(NB == WN * NW, "operand slots");
#pragma unroll
    for (int wm = 0; wm < WM; ++wm) av[wm] = *(const f32x4*)(Ap + wm * 128);
#pragma unroll
    for (int wn = 0; wn < WN; ++wn) {
        f32x4 v = *(const f32x4*)(Xp + cb4[wn]);
        if constexpr (BF == 2) {
            bv[wn * NW] = v;
            if (MIR >= 1) bv[wn * NW + 1] = *(const f32x4*)(Xp + cbl4[wn]);
            if (MIR == 2) bv[wn * NW + 2] = *(const f32x4*)(Xp + cbr4[wn]);
        } else {
            if (MIR >= 1) v += *(const f32x4*)(Xp + cbl4[wn]);
            if (MIR == 2) v += *(const f32x4*)(Xp + cbr4[wn]);
            bv[wn] = v;
        }
    }
}
template <int WM, int WN, int BF, int NW, int NB>
static __device__ __forceinline__ void conv_mma_unit(f32x16 (&acc)[WM][WN], const f32x4 (&av)[WM], const f32x4 (&bv)[NB]) {
    static_assert(NB == WN * NW, "operand slots");
    if constexpr (BF == 2) {
#pragma unroll
        for (int wm = 0; wm < WM; ++wm) {
            const avc_u32x4 ap = __builtin_bit_cast(avc_u32x4, av[wm]);
#pragma unroll
            for (int wn = 0; wn < WN; ++wn)
#pragma unroll
                for (int w = 0; w < NW; ++w) acc[wm][wn] = avc_mfma_bf16x8(ap, __builtin_bit_cast(avc_u32x4, bv[wn * NW + w]), acc[wm][wn]);
        }
        return;
    }
    if constexpr (BF == 1) {  // the unit's 8 reduction channels in ONE v_mfma_f32_32x32x8_bf16 (operands rounded here)
        avc_s16x4 bp[WN];
#pragma unroll
        for (int wn = 0; wn < WN; ++wn) bp[wn] = avc_pack_bf16x4(bv[wn][0], bv[wn][1], bv[wn][2], bv[wn][3]);
#pragma unroll
        for (int wm = 0; wm < WM; ++wm) {
            const avc_s16x4 ap = avc_pack_bf16x4(av[wm][0], av[wm][1], av[wm][2], av[wm][3]);
#pragma unroll
            for (int wn = 0; wn < WN; ++wn) acc[wm][wn] = avc_mfma_bf16(ap, bp[wn], acc[wm][wn]);
        }
        return;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int wm = 0; wm < WM; ++wm)
#pragma unroll
            for (int wn = 0; wn < WN; ++wn) acc[wm][wn] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[wm][u], bv[wn][u], acc[wm][wn], 0, 0, 0);
}

// KSC > 0: tap count and chunk depth are compile-time (GRC = CK/8), the chunk is one straight-line
// block with the fragments of unit u+1 fetched from LDS before the MFMAs of unit u are issued
// (register double buffering) so that a lone wave per SIMD does not stall on LDS latency.
// TS (straight-line chunks only): 0 = all taps, 1 = taps 0, 2, 4, ..., 2 = taps 1, 3, ... (stride-2 dgrad: the other
// taps of a column meet the zeros of the zero-upsampled dy)
template <int WM, int WN, int MIR, int KSC, int GRC, int BF, int TS = 0>
static __device__ __forceinline__ void conv_chunk_mma(f32x16 (&acc)[WM][WN], const float* Ab, const float* Xb, int KS, int CK, int ROW, int h,
                                                      int a_lane4, const int (&cb4)[WN], const int (&cbl4)[WN], const int (&cbr4)[WN]) {
    constexpr int BM4 = 64 * WM * 4;   // floats of one (tap, unit, h) plane of the A stage
    constexpr int NW = ConvNW<MIR, BF>::value;
    if constexpr (KSC > 0) {
        constexpr int NTAP = TS == 0 ? KSC : (TS == 1 ? (KSC + 1) / 2 : KSC / 2);
        constexpr int U = NTAP * GRC;
        f32x4 av[2][WM], bv[2][WN * NW];
        auto unit_ptrs = [&](int u, const float*& Ap, const float*& Xp) {
            const int tap = TS == 0 ? u / GRC : 2 * (u / GRC) + (TS == 2 ? 1 : 0), g8 = u % GRC;
            Ap = Ab + ((tap * GRC + g8) * 2 + h) * BM4 + a_lane4;
            Xp = Xb + ((g8 * 2 + h) * ROW + tap) * 4;
        };
        const float *Ap, *Xp;
        unit_ptrs(0, Ap, Xp);
        conv_load_unit<WM, WN, MIR, BF>(av[0], bv[0], Ap, Xp, cb4, cbl4, cbr4);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (u + 1 < U) {
                unit_ptrs(u + 1, Ap, Xp);
                conv_load_unit<WM, WN, MIR, BF>(av[(u + 1) & 1], bv[(u + 1) & 1], Ap, Xp, cb4, cbl4, cbr4);
            }
            conv_mma_unit<WM, WN, BF, NW>(acc, av[u & 1], bv[u & 1]);
        }
    } else {
        const int groups = CK >> 3;
        for (int tap = 0; tap < KS; ++tap) {
            const float* Ap = Ab + (tap * groups * 2 + h) * BM4 + a_lane4;
            const float* Xp = Xb + (h * ROW + tap) * 4;
            for (int g8 = 0; g8 < groups; ++g8) {
                f32x4 av[WM], bv[WN * NW];
                conv_load_unit<WM, WN, MIR, BF>(av, bv, Ap, Xp, cb4, cbl4, cbr4);
                conv_mma_unit<WM, WN, BF, NW>(acc, av, bv);
                Ap += 2 * BM4;
                Xp += 2 * ROW * 4;
            }
        }
    }
}

// KG > 1: intra-workgroup split-K for layers that cannot fill the chip (T_l <= 32: 128-256 tiles, each
// wave a serial chain of 320 MFMAs).  KG groups of 4 waves work on the SAME output tile; group kg
// runs its own double-buffered pipeline over chunks kg, kg+KG, ... and the groups' accumulators are
// summed through LDS in a fixed order at the end (deterministic).
// RAG (forward only): ragged launch -- the tile's sample, first column and the sample's own lengths / packed-buffer bases come
// from the ConvRag tables; everything else is the uniform kernel with one sample per tile.
// WN = 2 (64 x 128 and 128 x 128 tiles; uniform stride-1 launches): two 32-column fragments per wave share every weight fragment --
// the weight image is 80 % of a chunk's LDS-DMA bytes and it is re-read by every column tile (section 3.1 of DESIGN.md).
// TILE WALK (round 5; WALK instances: uniform launches with KG == 1): a workgroup is PERSISTENT over `a.walk_n` column tiles of its row slab -- tiles
// blockIdx.x, blockIdx.x + gridDim.x, ... .  The walk keeps the first frame t0 of the tile and steps only the sample (a.walk_db samples per
// step), so everything the prologue derives per lane -- source offsets with their reflect / zero-upsampling index math, structural zeros
// of the X stages, column descriptors, mirror windows -- is computed ONCE; a step is a scalar bump of the source base and of the
// epilogue's sample index.  The chunk pipeline runs THROUGH the tile boundary: the LDS-DMA of the next tile's first chunk is issued
// in front of the last chunk's products of this tile, the wave waits for ITS pieces before the epilogue (they had the whole chunk to
// land), and the barrier in front of the next tile's first products is a bare s_barrier behind the epilogue -- no s_waitcnt vmcnt(0)
// stands between the stores of tile i and the loop of tile i + 1.  Each tile is computed by exactly the instruction sequence of the
// one-tile kernel: results are bit-identical (tests/test_conv_walk.py).
// MEASURED (profiles/r05_conv_walk_ablation.log, one MI355X, same-box A/B against the round-4 library): no gain.  With 1024 tiles on 1024
// resident slots (the model's T = 128 layers at B = 256) there is nothing to pipeline -- two walkers per CU x two tiles lose 6-14 % against
// four one-tile workgroups per CU -- and where a launch holds many tiles per slot (B = 1024: 4096 tiles; the 8192-tile bank) the hardware's
// own dispatch of the next workgroup already overlaps prologue and epilogue with the neighbours' products: walk 201 us vs 202 us, and the
// shared kernel body paid 3-5 % for carrying the walk's state through its epilogue (registers 100 -> 120, ~80 spilled scalars).  Hence a
// compile-time flag: WALK = false IS the round-4 kernel; the WALK instances exist for the tests and `avc_tuning.conv_walk` (default 0).
// INF instances (round 5): the InstanceNorm / AdaIN / activation / residual rows of the conv's output inside the epilogue (ConvINFuse,
// conv_shared.h: conv_epilogue_in) -- forward launches on 64 x 64 tiles whose columns are whole rows.  A compile-time flag like WALK: as a
// run-time branch in the one kernel body the second epilogue cost every instance its register budget (100 -> 280 VGPRs, ~130 spilled
// scalars: the compiler evaluates both epilogues' launch-uniform conditions in front of the chunk loop).
// INB instances: the backward twin -- an input-gradient launch whose output rows are d(loss)/d(out) of an InstanceNorm layer turns them into
// d(loss)/d(y) in its epilogue (ConvINBwd, conv_shared.h: conv_epilogue_in_bwd).
template <int WM, int WN, bool MIRROR, int KSC, int GRC, int KG, int BF, bool PAR = false, bool RAG = false, bool WALK = false, bool INF = false, bool INB = false>
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
    const int GR = CK >> 3;
    int Tout = a.Tout, Tsrc = a.Tsrc, Bv = a.B;
    long xsb = a.x.sb, xsc = a.x.sc;
    const float* xptr = a.x.ptr;
    ConvEpi epi = conv_epi(a);
    ConvGeom q;
    if constexpr (RAG) {
        const int rb_ = a.rag.tile[2 * blockIdx.x], rt0 = a.rag.tile[2 * blockIdx.x + 1];
        Tsrc = a.rag.Tsrc[rb_];
        Tout = a.rag.Tout[rb_];
        Bv = 1;
        xsb = 0;
        xsc = a.x.sc < 0 ? Tsrc : a.x.sc;   // packed activations: channel rows of the sample's own length; (the packed input brings explicit strides)
        xptr += (long)a.rag.cx * a.rag.offsrc[rb_];
        q.b0 = 0; q.t0 = rt0; q.SPT = 1; q.ncols = BN;
        q.SEG = (BN - 1) * a.stride + KS; q.seg_p0 = rt0 * a.stride;
        q.ROWDATA = q.SEG; q.ROW = q.SEG + KS;
        epi.Tout = Tout;
        epi.ob = 0;
        epi.oc = (long)a.ops * Tout;
        epi.obase = (long)a.rag.cout * a.rag.offout[rb_] + (long)g.out_c0 * epi.oc;
        if (a.res_mode != AVC_RES_NONE) {
            epi.Tres = a.rag.Tres[rb_];
            epi.rb = 0;
            epi.rc = epi.Tres;
            epi.rbase = (long)a.rag.cres * a.rag.offres[rb_];
        }
    } else {
        q = conv_geom(a.mode, a.stride, Tout, KS, BN, blockIdx.x);
    }
    const int ROW = q.ROW;
    const int m_tile0 = blockIdx.y * BM;

    const int AS = KS * CK * BM;  // floats per A stage
    const int XS = CK * ROW;      // floats per X stage
    constexpr int NS = (BF == 2) ? AVC_CONV_STAGES_BH : AVC_CONV_STAGES;   // chunk c + NS - 1 is in flight while chunk c multiplies (see the main loop)
    float* As = smem + kg * NS * AS;                 // this group's A stages
    float* Xs = smem + KG * NS * AS + kg * NS * XS;  // ... and X stages

    // ---- per-lane source descriptors of the X tile.  One dword LDS-DMA instruction fills 64 consecutive floats of an
    // (unit, h) plane = 16 positions x 4 k-steps: lane l fetches position 16 j + (l >> 2) of reduction channel
    // 8 unit + 2 (l & 3) + h.  The destination is lane-contiguous, the SOURCE address is the lane's own -- that is what
    // interleaves four channel rows of the [B, C, T] tensor into 16-byte fragments without any staging register.
    // Wave w stages the position groups j = 2 jj + (w >> 1) of the planes pl = (w & 1), (w & 1) + 2, ...: it keeps only ITS
    // offsets (chunk-invariant): xo[jj] = element offset of the position inside a channel row plus the lane's channel
    // offset, or -1 where the tile holds a structural zero (halo / null window / masked / past the row).
    const int ul = lane & 3;
    const long lu = (a.x.ps == 1) ? (long)(2 * ul) * xsc : (long)ul * xsc;   // channel 2u of the unit (ps = 2: pixel-unshuffled view, model.py:52-59)
    const int jpar = wave >> 1;
    int xo[AVC_CONV_NJ4 / 2];
#pragma unroll
    for (int jj = 0; jj < AVC_CONV_NJ4 / 2; ++jj) {
        const int p = 16 * (2 * jj + jpar) + (lane >> 2);
        int sp = -1;
        if (p < q.ROWDATA) {
            int seg = p / q.SEG, qq = p - seg * q.SEG;
            int b = q.b0 + seg;
            int pp = q.seg_p0 + qq;
            const int bs = WALK ? seg : b;   // (a walk steps the base pointer: offsets relative to the tile's first sample)
            if (b < Bv) {
                if (a.mode == 0) {
                    int v = pp - padL;
                    int r = avc_reflect(v, Tsrc);
                    if (r >= 0 && r < Tsrc) sp = (int)(bs * xsb + (long)r * a.x.st + lu);
                } else {
                    int v = pp - (KS - 1);
                    if (v >= 0) {
                        int vs = v / a.stride;
                        if (vs * a.stride == v && vs < Tsrc) sp = (int)(bs * xsb + (long)vs * a.x.st + lu);
                    }
                }
            }
        }
        xo[jj] = sp;
    }
    // the X stages start as zeros; structural zeros are never overwritten afterwards
    for (int e = tid; e < KG * NS * XS; e += NTHREADS) smem[KG * NS * AS + e] = 0.f;

    // ---- this lane's columns (float index of a position inside an X plane = 4 x position)
    int cb4[WN], cbl4[WN], cbr4[WN], colb[WN], colt[WN];
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
                v = (t < Tout) && (q.b0 < Bv);
            } else {
                const int halfT = Tout >> 1;
                bl = li / halfT;
                t = 2 * (li - bl * halfT) + wave_n;
                v = (bl < q.SPT) && (q.b0 + bl < Bv);
            }
        } else if (RAG || (q.SPT == 1 && Tout >= BN)) {
            bl = 0;
            t = q.t0 + n;
            v = (t < Tout) && (q.b0 < Bv);
        } else {
            bl = n / Tout;
            t = n - bl * Tout;
            v = (bl < q.SPT) && (q.b0 + bl < Bv);
        }
        colb[wn] = WALK ? bl : q.b0 + bl;   // (walk: sample inside the tile; the tile's first sample is added in the epilogue)
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
        cb4[wn] = 4 * base;
        cbl4[wn] = 4 * bL;
        cbr4[wn] = 4 * bR;
    }

    const int npieces = (KS * CK * BM) >> 8;  // 1 KiB (256 floats) per wave-instruction of the LDS DMA
    const int nj = (ROW + 15) >> 4;

    // reflect-adjoint windows are needed only by waves that own a column within pad of a sample edge
    bool use_mirror = false;
    const bool one_window = Tout >= 2 * (padL + padR) + 2;   // left-edge columns [1, padL] and right-edge columns [Tout-1-padR, Tout-2] are disjoint
    if (MIRROR) {
        bool mine = false;
#pragma unroll
        for (int wn = 0; wn < WN; ++wn) {
            mine |= (cbl4[wn] != 4 * q.ROWDATA) || (cbr4[wn] != 4 * q.ROWDATA);
            if (one_window && cbl4[wn] == 4 * q.ROWDATA) cbl4[wn] = cbr4[wn];   // the column's only mirror window
        }
        use_mirror = __any(mine);
    }

    __syncthreads();  // zero fill done before the first DMA lands

    // weights: packed image == LDS image -> direct global->LDS DMA, 16 B per lane, no staging registers.
    // Piece = 64 rows x 4 k-steps of one (tap, unit, h) plane.
    auto load_a = [&](int chunk, int buf) {
        const float* wsrc = g.wp + (long)chunk * KS * CK * a.Mp + (long)m_tile0 * 4 + lane * 4;
        float* Ad = As + buf * AS;
        for (int piece = wave; piece < npieces; piece += 4) {
            const int plane = WM == 1 ? piece : piece >> 1, sub = WM == 1 ? 0 : piece & 1;
            avc_glds16(wsrc + (long)plane * a.Mp * 4 + sub * 256, Ad + piece * 256);
        }
    };
    // source tile: this wave's position groups of its planes (see xo above)
    const int njw = (nj - jpar + 1) >> 1;   // position groups j = 2 jj + jpar < nj
    auto load_x = [&](const float* xtile, int chunk, int buf) {
        float* Xd = Xs + buf * XS + 64 * jpar;
        const int c_chunk = chunk * CK;
        for (int pl = wave & 1; pl < 2 * GR; pl += 2) {
            const int c0 = c_chunk + 8 * (pl >> 1), hh = pl & 1;
            if (c0 >= a.Cred) break;   // channel padding of the last chunk: the weight image holds zeros there, the (finite) stale tile is harmless
            const long pbase = (a.x.ps == 1) ? (long)(c0 + hh) * xsc : (long)(c0 >> 1) * xsc + hh;
            long fix = 0;   // a unit that straddles Cred: lanes past the last channel read the last valid one (times zero weights)
            if (c0 + 8 > a.Cred) {
                int c = c0 + 2 * ul + hh;
                c = c < a.Cred ? c : a.Cred - 1;
                fix = ((a.x.ps == 1) ? (long)c * xsc : (long)(c >> 1) * xsc + (c & 1)) - (pbase + lu);
            }
            const float* src = xtile + pbase + fix;
            float* dst = Xd + pl * ROW * 4;
#pragma unroll
            for (int jj = 0; jj < AVC_CONV_NJ4 / 2; ++jj)
                if (jj < njw && xo[jj] >= 0) avc_glds4(src + xo[jj], dst + 128 * jj);
        }
    };

    // LDS-DMA instructions one chunk costs THIS wave (a lower bound is what the partial wait below needs): its weight
    // pieces + its position groups that have at least one active lane, per plane of the chunk that holds real channels
    const int na_w = (npieces - wave + 3) >> 2;
    int nx_w = 0;
#pragma unroll
    for (int jj = 0; jj < AVC_CONV_NJ4 / 2; ++jj) nx_w += (jj < njw && __any(xo[jj] >= 0)) ? 1 : 0;
    auto dma_count = [&](int chunk) {
        int planes = 0;
        for (int pl = wave & 1; pl < 2 * GR; pl += 2) planes += (chunk * CK + 8 * (pl >> 1) < a.Cred) ? 1 : 0;
        return na_w + nx_w * planes;
    };

    constexpr int DIST = NS - 1;   // prefetch distance in chunks
    // ---- tile walk state
    static_assert(!WALK || (KG == 1 && !RAG && NS == 2), "tile walk: uniform launches, one wave group, two stages");
    const int walk_n = WALK ? a.walk_n - ((int)blockIdx.x < a.walk_rem ? 0 : 1) : 1;   // (the first walk_rem walkers take one tile more)
    const long walk_dx = WALK ? (long)a.walk_db * xsb : 0;
    const float* xtile = WALK ? xptr + (long)q.b0 * xsb : xptr;
    int b0_cur = WALK ? q.b0 : 0;
    if (kg < nchunk) {
        load_a(kg, 0);
        load_x(xtile, kg, 0);
    }
    if (DIST == 2 && kg + KG < nchunk && !(a.dbg & 1)) {
        load_a(kg + KG, 1);
        load_x(xtile, kg + KG, 1);
    }
    __syncthreads();

    const int a_lane4 = (wave_m * (32 * WM) + li) * 4;
    const int nit = (nchunk + KG - 1) / KG;
    int st = 0;   // stage of the chunk being multiplied (it % NS)
    for (int wk = 0; wk < walk_n; ++wk) {
    f32x16 acc[WM][WN];
#pragma unroll
    for (int wm = 0; wm < WM; ++wm)
#pragma unroll
        for (int wn = 0; wn < WN; ++wn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[wm][wn][r] = 0.f;
    const bool walk_more = WALK && wk + 1 < walk_n;
    for (int it = 0; it < nit; ++it) {
        const int chunk = it * KG + kg;
        // chunk c + DIST is issued now and lands while chunks c .. c + DIST - 1 are multiplied; chunk c + 1 must have landed by the
        // barrier at the end of this iteration (DIST == 1: that is everything outstanding)
        const bool more = (chunk + DIST * KG < nchunk) && !((a.dbg & 1) && (DIST == 2 || it >= 1));
        const bool last_of_tile = walk_more && it == nit - 1;   // (tile walk: DIST == 1) the next tile's first chunk rides under this tile's last
        int issued = 0;
        if (more) {
            const int stn = (st + DIST) % NS;   // last read during chunk c - 1, free since that chunk's barrier
            load_a(chunk + DIST * KG, stn);
            load_x(xtile, chunk + DIST * KG, stn);
            if (DIST == 2) issued = dma_count(chunk + DIST * KG);
        } else if (last_of_tile && !(a.dbg & 1)) {
            const int stn = (st + 1) % NS;
            load_a(0, stn);
            load_x(xtile + walk_dx, 0, stn);
        }
        const float* Ab = As + st * AS;
        const float* Xb = Xs + st * XS;
        if ((a.dbg & 2) || chunk >= nchunk) {
        } else if constexpr (PAR) {   // even columns: taps 0, 2, 4; odd columns: taps 1, 3 (k = 5, padL = 2)
            if (wave_n == 0) {
                if (MIRROR && use_mirror && one_window) conv_chunk_mma<WM, WN, 1, KSC, GRC, BF, 1>(acc, Ab, Xb, KS, CK, ROW, h, a_lane4, cb4, cbl4, cbr4);
                else if (MIRROR && use_mirror) conv_chunk_mma<WM, WN, 2, KSC, GRC, BF, 1>(acc, Ab, Xb, KS, CK, ROW, h, a_lane4, cb4, cbl4, cbr4);
                else conv_chunk_mma<WM, WN, 0, KSC, GRC, BF, 1>(acc, Ab, Xb, KS, CK, ROW, h, a_lane4, cb4, cbl4, cbr4);
            } else {
                if (MIRROR && use_mirror && one_window) conv_chunk_mma<WM, WN, 1, KSC, GRC, BF, 2>(acc, Ab, Xb, KS, CK, ROW, h, a_lane4, cb4, cbl4, cbr4);
                else if (MIRROR && use_mirror) conv_chunk_mma<WM, WN, 2, KSC, GRC, BF, 2>(acc, Ab, Xb, KS, CK, ROW, h, a_lane4, cb4, cbl4, cbr4);
                else conv_chunk_mma<WM, WN, 0, KSC, GRC, BF, 2>(acc, Ab, Xb, KS, CK, ROW, h, a_lane4, cb4, cbl4, cbr4);
            }
        } else if (MIRROR && use_mirror && one_window)   // wave-uniform: only waves owning a column within pad of a sample edge
            conv_chunk_mma<WM, WN, 1, KSC, GRC, BF>(acc, Ab, Xb, KS, CK, ROW, h, a_lane4, cb4, cbl4, cbr4);
        else if (MIRROR && use_mirror)
            conv_chunk_mma<WM, WN, 2, KSC, GRC, BF>(acc, Ab, Xb, KS, CK, ROW, h, a_lane4, cb4, cbl4, cbr4);
        else if constexpr (KSC < 0) {
            // grouped launch of layers with different tap counts (the conv bank, k = 1..8): the workgroup's (taps,
            // chunk depth) pair is uniform, so each pair gets its own straight-line chunk
            switch (KS * 8 + GR) {
                case 1 * 8 + 4: conv_chunk_mma<WM, WN, 0, 1, 4, BF>(acc, Ab, Xb, KS, CK, ROW, h, a_lane4, cb4, cbl4, cbr4); break;
                case 2 * 8 + 2: conv_chunk_mma<WM, WN, 0, 2, 2, BF>(acc, Ab, Xb, KS, CK, ROW, h, a_lane4, cb4, cbl4, cbr4); break;
                case 3 * 8 + 2: conv_chunk_mma<WM, WN, 0, 3, 2, BF>(acc, Ab, Xb, KS, CK, ROW, h, a_lane4, cb4, cbl4, cbr4); break;
                case 4 * 8 + 1: conv_chunk_mma<WM, WN, 0, 4, 1, BF>(acc, Ab, Xb, KS, CK, ROW, h, a_lane4, cb4, cbl4, cbr4); break;
                case 5 * 8 + 1: conv_chunk_mma<WM, WN, 0, 5, 1, BF>(acc, Ab, Xb, KS, CK, ROW, h, a_lane4, cb4, cbl4, cbr4); break;
                case 6 * 8 + 1: conv_chunk_mma<WM, WN, 0, 6, 1, BF>(acc, Ab, Xb, KS, CK, ROW, h, a_lane4, cb4, cbl4, cbr4); break;
                case 7 * 8 + 1: conv_chunk_mma<WM, WN, 0, 7, 1, BF>(acc, Ab, Xb, KS, CK, ROW, h, a_lane4, cb4, cbl4, cbr4); break;
                case 8 * 8 + 1: conv_chunk_mma<WM, WN, 0, 8, 1, BF>(acc, Ab, Xb, KS, CK, ROW, h, a_lane4, cb4, cbl4, cbr4); break;
                default: conv_chunk_mma<WM, WN, 0, 0, 0, BF>(acc, Ab, Xb, KS, CK, ROW, h, a_lane4, cb4, cbl4, cbr4);
            }
        } else
            conv_chunk_mma<WM, WN, 0, KSC, GRC, BF>(acc, Ab, Xb, KS, CK, ROW, h, a_lane4, cb4, cbl4, cbr4);
        if (last_of_tile) {
            conv_wait_dma(0);   // this wave's pieces of the next tile's first chunk; the workgroup's barrier follows the epilogue
        } else if (!(a.dbg & 4)) {
            conv_wait_dma(issued);   // everything but the DMA instructions issued in THIS iteration has landed
            conv_bare_barrier();
        }
        st = st == NS - 1 ? 0 : st + 1;
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
        // (the waves of groups kg > 0 END here; the remaining four waves still meet at workgroup barriers below -- the fused InstanceNorm
        // epilogues' and the one in front of them.  s_barrier on gfx9 counts the waves of the workgroup that have not terminated, so a barrier
        // behind an exited wave is well defined on this target -- the only one this file is written for; ADVICE r5 notes that the HIP
        // language itself leaves it undefined.)
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
    if constexpr (INF) {   // fused InstanceNorm rows (conv_shared.h): the tile goes through the (now free) stage memory
        static_assert(WM == 1 && WN == 1 && (BF == 0 || BF == 2) && !PAR && !RAG && !WALK && !MIRROR, "fused InstanceNorm epilogue: forward, 64 x 64 tiles, fp32 or bf16 pairs");
        if (KG > 1) __syncthreads();   // (the other waves of this group may still be reading the split-K partials)
        if constexpr (BF == 2) conv_epilogue_in_pairs(a, g, acc[0][0], smem, tid, wave_m, wave_n, li, h, m_tile0, q.b0);
        else conv_epilogue_in(a, g, acc[0][0], smem, tid, wave_m, wave_n, li, h, m_tile0, q.b0);
        return;
    }
    if constexpr (INB) {
        static_assert(WM == 1 && WN == 1 && (BF == 0 || BF == 2) && !RAG && !WALK && !INF, "fused InstanceNorm-backward epilogue: input gradient on 64 x 64 tiles, fp32 or bf16 pairs");
        if (KG > 1) __syncthreads();
        if constexpr (BF == 2) conv_epilogue_in_bwd_pairs(a, g, acc[0][0], smem, tid, wave_m, h, m_tile0, q.b0, q.t0, colb[0] - q.b0, colt[0], colv[0]);
        else conv_epilogue_in_bwd(a, g, acc[0][0], smem, tid, wave_m, h, m_tile0, q.b0, q.t0, colb[0] - q.b0, colt[0], colv[0]);
        return;
    }
    if (!(a.dbg & 8)) {
        if constexpr (WALK) {
            // The epilogue's parameters (strides, bases, pointers: ~40 scalar registers) and its row / column arithmetic are invariant along a
            // walk.  Left visible, the compiler keeps all of it live through the chunk loop -- hoisted address arithmetic (246-398 VGPRs: half
            // the waves per SIMD) and ~100 spilled scalars.  So every tile RE-READS its parameters from the kernel-argument segment through an
            // opaque pointer (scalar loads under the last products) and takes opaque copies of the lane's row / column.
            ConvEpi epi_t;
            ConvGroup g_t;
#ifdef AVC_EMU
            const ConvArgs* ka = &a;
#else
            const __attribute__((address_space(4))) ConvArgs* ka = (const __attribute__((address_space(4))) ConvArgs*)__builtin_amdgcn_kernarg_segment_ptr();
            asm volatile("" : "+s"(ka));
#endif
            epi_t.ob = ka->ob; epi_t.oc = ka->oc; epi_t.rb = ka->rb; epi_t.rc = ka->rc; epi_t.obase = 0; epi_t.rbase = 0;
            epi_t.ot = ka->ot; epi_t.ops = ka->ops; epi_t.rt = ka->rt; epi_t.Tres = ka->Tres; epi_t.Tout = ka->Tout; epi_t.M = ka->M;
            epi_t.act = ka->act; epi_t.res_mode = ka->res_mode; epi_t.res_to_primary = ka->res_to_primary; epi_t.slope = ka->slope;
            epi_t.pairs = ka->pairs;
            const int z = blockIdx.z;
            g_t.wp = nullptr; g_t.bias = ka->g[z].bias; g_t.out = ka->g[z].out; g_t.out2 = ka->g[z].out2; g_t.res = ka->g[z].res;
            g_t.mask = ka->g[z].mask; g_t.KS = 0; g_t.padL = 0; g_t.padR = 0; g_t.nchunk = 0; g_t.CK = 0; g_t.out_c0 = 0;
#pragma unroll
            for (int wn = 0; wn < WN; ++wn) {
                if (!colv[wn]) continue;
#pragma unroll
                for (int wm = 0; wm < WM; ++wm) {
                    int tcol = colt[wn], bcol = b0_cur + colb[wn], mb = m_tile0 + wave_m * (32 * WM) + wm * 32;
#ifndef AVC_EMU
                    asm volatile("" : "+v"(tcol), "+v"(bcol));
                    asm volatile("" : "+s"(mb));
#endif
                    if (BF == 2 && epi_t.pairs) conv_store_frag_pairs(epi_t, g_t, acc[wm][wn], mb, h, bcol, tcol);
                    else conv_store_frag(epi_t, g_t, acc[wm][wn], mb, h, bcol, tcol);
                }
            }
        } else {
#pragma unroll
            for (int wn = 0; wn < WN; ++wn) {
                if (!colv[wn]) continue;
#pragma unroll
                for (int wm = 0; wm < WM; ++wm) {
                    if (BF == 2 && epi.pairs) conv_store_frag_pairs(epi, g, acc[wm][wn], m_tile0 + wave_m * (32 * WM) + wm * 32, h, colb[wn], colt[wn]);
                    else conv_store_frag(epi, g, acc[wm][wn], m_tile0 + wave_m * (32 * WM) + wm * 32, h, colb[wn], colt[wn]);
                }
            }
        }
    }
    if (walk_more) {
        conv_bare_barrier();   // every wave's pieces of the next tile's first chunk have landed (each waited for its own in front of its epilogue)
        xtile += walk_dx;
        b0_cur += a.walk_db;
    }
    }   // tile walk
}

// --------------------------------------------------------------------------
// weight packing: W[Cout][Cin][KS] (state_dict layout) -> LDS-image order.  With red = chunk*CK + r the reduction
// channel, m the output row, j the tap:
//   AVC_IMG_PLAIN  Wp[chunk][j][r][m]
//   AVC_IMG_K4     Wp[chunk][j][unit][h][m][u],  r = 8 unit + 2 u + h      (conv_gemm.hip: 16-byte fragments)
//   value: fwd  W[m][red][j];  dgrad  W[red][m][KS-1-j]   (transposed, tap-flipped)
// Several source tensors of identical shape can be stacked along the forward
// output-channel axis (the 12 AdaIN affine Linears, the mu/log_sigma heads).
// --------------------------------------------------------------------------

struct PackBatch {
    PackArgs a[AVC_PACK_BATCH];
};

static __device__ __forceinline__ void pack_one(const PackArgs& p, long first, long stride, long limit = (1L << 62));

// several layers per launch: blockIdx.y selects the layer descriptor
__global__ void __launch_bounds__(AVC_THREADS) pack_weight_batch_kernel(const PackBatch b) {
    pack_one(b.a[blockIdx.y], (long)blockIdx.x * AVC_THREADS + threadIdx.x, (long)gridDim.x * AVC_THREADS);
}

// ONE launch for every image of a plan: the descriptors live in a device table the plan owns (built once at plan creation); their
// pointers are stored as BYTE OFFSETS from the caller's parameter buffer (sources) and workspace (destinations), rebased here.
// Block b handles piece blk[b].y of image blk[b].x.
//   * staged pieces (PackArgs.mb > 0; round 6): one reduction chunk x `mb` consecutive output rows.  Its source elements are a few
//     CONTIGUOUS runs of the state_dict tensor -- forward: per row m the chunk's channels x taps (CK KS floats); input gradient: per
//     reduction row the block's columns x taps (mb KS floats) -- read coalesced into LDS; the image order is then a gather from LDS and
//     the block's stores are contiguous runs of mb 16-byte groups.  (The round-5 kernel gathered from global memory, one scattered
//     4-byte read per element: every wave instruction touched 64 cache lines, 87 us per step at 0.85 TB/s of useful bytes.)
//   * legacy pieces (mb == 0: the split-bf16 images, and any image whose chunk does not fit the stage): AVC_PACK_PIECE consecutive
//     image elements, gathered from global memory.
static __device__ __forceinline__ void pack_staged(const PackArgs* __restrict__ gp, const float* __restrict__ params, float* __restrict__ ws, int piece, float* stage) {
    // (every field is read once from the table entry -- a block-uniform address -- into a scalar: a by-value PackArgs whose `src` is indexed
    // at run time lives in scratch memory)
    const int tid = threadIdx.x;
    const int MB = gp->mb, Mp = gp->Mp, M = gp->M, KS = gp->KS, CK = gp->CK, Cin = gp->Cin, Cout = gp->Cout, img = gp->img;
    const int nsrc = gp->nsrc, rows_per_src = gp->rows_per_src;
    const bool dgrad = gp->dgrad != 0;
    float* dst = (float*)((char*)ws + (size_t)gp->dst);
    const char* pbase = (const char*)params;
    const int nmb = (Mp + MB - 1) / MB;
    const int chunk = piece / nmb, m0 = (piece - chunk * nmb) * MB;
    const bool pairs = img == AVC_IMG_K4H;   // (CK counts dword channels = bf16 pairs of channels there)
    const int CKr = pairs ? 2 * CK : CK, c0 = chunk * CKr;
    const int seglen = dgrad ? MB * KS : CKr * KS, nseg = dgrad ? CKr : MB;
    const int pitch = seglen | 1;   // (odd: the forward gather below walks the segments lane by lane)
    const int n = nseg * seglen;
    const float inv_seglen = 1.0f / (float)seglen, inv_rows = 1.0f / (float)rows_per_src;
    // segment = forward: output row m0 + seg, elements W[m][c0 ...][*]; input gradient: reduction row c0 + seg (a forward output channel),
    // elements W[c][m0 ...][*].  Either way a run of `seglen` consecutive floats that starts at (row Cin + col0) KS.
    const int row0 = dgrad ? c0 : m0, col0 = dgrad ? m0 : c0;
    const int row_end = dgrad ? Cout : M, off_end = ((dgrad ? M : Cin) - col0) * KS;   // (valid: row < row_end and off < off_end)
    // 16 independent, UNCONDITIONAL reads in flight per thread and round (an out-of-range element reads element 0 of the tensor and is
    // replaced by zero afterwards): as a loop of guarded reads every element was its own round trip -- 55 us per step.
    constexpr int U = 16;
    for (int e0 = tid; e0 < n; e0 += U * AVC_THREADS) {
        float v[U];
        int at_[U];
        bool ok[U];
        long idx[U];
        const float* w[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {   // (index arithmetic + the table reads of the stacked tensors' base pointers: no branch, no wait in between)
            const int e = e0 + u * AVC_THREADS;
            const int seg = avc_fastdiv(e, seglen, inv_seglen), off = e - seg * seglen;
            const int row = row0 + seg;
            ok[u] = e < n && row < row_end && off < off_end;
            at_[u] = e < n ? seg * pitch + off : -1;
            int s_ = avc_fastdiv(row, rows_per_src, inv_rows);   // stacked tensors: which one holds forward output row `row`
            s_ = s_ < nsrc ? s_ : nsrc - 1;
            w[u] = (const float*)(pbase + (size_t)gp->src[s_]);
            idx[u] = ok[u] ? ((long)(row - s_ * rows_per_src) * Cin + col0) * KS + off : 0;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = w[u][idx[u]];
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (at_[u] >= 0) stage[at_[u]] = ok[u] ? v[u] : 0.f;
    }
    __syncthreads();
    const int lg = 31 - __builtin_clz(MB);
    auto at = [&](int mi, int cc, int j) -> float {   // (row m0 + mi, reduction channel c0 + cc, tap j)
        return dgrad ? stage[cc * pitch + mi * KS + (KS - 1 - j)] : stage[mi * pitch + cc * KS + j];
    };
    if (img == AVC_IMG_K4 || img == AVC_IMG_K4H) {
        const int GR = CK >> 3, ngr = KS * GR * 2 * MB;
        const float inv_GR = 1.0f / (float)GR;
        for (int idx = tid; idx < ngr; idx += AVC_THREADS) {
            const int mi = idx & (MB - 1);
            int q = idx >> lg;
            const int h = q & 1;
            q >>= 1;
            const int j = avc_fastdiv(q, GR, inv_GR), unit = q - j * GR;
            if (m0 + mi >= Mp) continue;
            const long g4 = ((((long)chunk * KS + j) * GR + unit) * 2 + h) * Mp + m0 + mi;
            const int r0 = 8 * unit + h;   // reduction channel (K4H: dword channel) of u = 0 inside the chunk; u steps it by 2
            if (pairs) {
                unsigned w[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) w[u] = bh_pack(at(mi, 2 * (r0 + 2 * u), j), at(mi, 2 * (r0 + 2 * u) + 1, j));
                *(avc_u32x4*)((unsigned*)dst + 4 * g4) = avc_u32x4{w[0], w[1], w[2], w[3]};
            } else {
                float v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = at(mi, r0 + 2 * u, j);
                *(f32x4*)(dst + 4 * g4) = f32x4{v[0], v[1], v[2], v[3]};
            }
        }
    } else {   // AVC_IMG_PLAIN: Wp[chunk][j][r][m]
        const int nel = KS * CK * MB;
        const float inv_CK = 1.0f / (float)CK;
        for (int idx = tid; idx < nel; idx += AVC_THREADS) {
            const int mi = idx & (MB - 1), q = idx >> lg;
            const int j = avc_fastdiv(q, CK, inv_CK), r = q - j * CK;
            if (m0 + mi >= Mp) continue;
            dst[(((long)chunk * KS + j) * CK + r) * Mp + m0 + mi] = at(mi, r, j);
        }
    }
}

// rows of a staged piece: the largest power of two <= 128 whose chunk slab fits the stage; 0 = the image takes legacy pieces
int avc_pack_stage_rows(const PackArgs& p) {
    if (p.img != AVC_IMG_K4 && p.img != AVC_IMG_K4H && p.img != AVC_IMG_PLAIN) return 0;
    const long per_row = (long)(p.img == AVC_IMG_K4H ? 2 * p.CK : p.CK) * p.KS;
    if (p.rows_per_src <= 0 || (long)p.Cout * p.Cin * p.KS >= (1L << 22)) return 0;   // (float-reciprocal divisions)
    for (int mb = 128; mb >= 8; mb >>= 1)
        if (mb * per_row <= AVC_PACK_STAGE) return mb;
    return 0;
}
long avc_pack_pieces(const PackArgs& p) {
    if (p.mb > 0) return (long)p.nchunk * ((p.Mp + p.mb - 1) / p.mb);
    return (avc_pack_total(p) + AVC_PACK_PIECE - 1) / AVC_PACK_PIECE;
}

__global__ void __launch_bounds__(AVC_THREADS) pack_weight_table_kernel(const PackArgs* __restrict__ tab, const int2* __restrict__ blk, const float* __restrict__ params, float* __restrict__ ws) {
    __shared__ float stage[AVC_PACK_STAGE + 128];
    const int2 bi = blk[blockIdx.x];
    if (tab[bi.x].mb > 0) {
        pack_staged(tab + bi.x, params, ws, bi.y, stage);
        return;
    }
    PackArgs p = tab[bi.x];
    p.dst = (float*)((char*)ws + (size_t)p.dst);
    for (int k = 0; k < p.nsrc; ++k) p.src[k] = (const float*)((const char*)params + (size_t)p.src[k]);
    const int unit = (p.img == AVC_IMG_K4 || p.img == AVC_IMG_K4H) ? AVC_PACK_PIECE / 4 : AVC_PACK_PIECE;   // (K4 images are packed in 16-byte groups)
    pack_one(p, (long)bi.y * unit + threadIdx.x, AVC_THREADS, (long)(bi.y + 1) * unit);
}

__global__ void __launch_bounds__(AVC_THREADS) pack_weight_kernel(const PackArgs p) {
    pack_one(p, (long)blockIdx.x * AVC_THREADS + threadIdx.x, (long)gridDim.x * AVC_THREADS);
}

static __device__ __forceinline__ void pack_one(const PackArgs& p, long first, long stride, long limit) {
    if (p.img == AVC_IMG_X3) {
        avc_pack_x3_one(p, first, stride, limit);
        return;
    }
    long total = (long)p.nchunk * p.KS * p.CK * p.Mp;
    const int GR = p.CK >> 3;
    // (the gather is one scattered read per element; with run-time integer divisions -- ~25 instructions each, five per element -- it
    // was ALU-bound: exact float-reciprocal divisions instead, valid below 2^22 = every image of the model; larger ones take the slow path)
    const bool fastok = total < (1L << 22);
    const float inv_Mp = 1.0f / (float)p.Mp, inv_KS = 1.0f / (float)p.KS, inv_GR = 1.0f / (float)GR, inv_CK = 1.0f / (float)p.CK;
    auto value = [&](int m, int c, int j) -> float {   // (output row m, reduction channel c, tap j)
        if (!p.dgrad) {
            if (m < p.M && c < p.Cin) {
                int s = p.nsrc == 1 ? 0 : m / p.rows_per_src, mm = m - s * p.rows_per_src;
                return p.src[s][((long)mm * p.Cin + c) * p.KS + j];
            }
        } else {
            if (m < p.M && c < p.Cout) {
                int s = p.nsrc == 1 ? 0 : c / p.rows_per_src, rr = c - s * p.rows_per_src;
                return p.src[s][((long)rr * p.Cin + m) * p.KS + (p.KS - 1 - j)];
            }
        }
        return 0.f;
    };
    if (p.img == AVC_IMG_K4 || p.img == AVC_IMG_K4H) {
        // K4 images, round 5: a thread packs one 16-BYTE GROUP -- the four k-steps u of one (chunk, tap, unit, h, row): four (K4H: eight)
        // independent gathered reads in flight behind ONE index decode, one 16-byte store.  first / stride / limit count GROUPS here.
        // (One element per dependent round trip -- the round-4 kernel -- was a latency chain: 93 us per step at 0.8 TB/s.)
        const long total4 = total >> 2;   // (CK is a multiple of 8)
        const long lim = total4 < limit ? total4 : limit;
        for (long g4 = first; g4 < lim; g4 += stride) {
            int m, h, unit, j, chunk;
            if (fastok) {
                int q = avc_fastdiv((int)g4, p.Mp, inv_Mp);
                m = (int)g4 - q * p.Mp;
                h = q & 1;
                q >>= 1;
                const int q2 = avc_fastdiv(q, GR, inv_GR);
                unit = q - q2 * GR;
                chunk = avc_fastdiv(q2, p.KS, inv_KS);
                j = q2 - chunk * p.KS;
            } else {
                long rest = g4;
                m = (int)(rest % p.Mp);
                rest /= p.Mp;
                h = (int)(rest & 1);
                rest >>= 1;
                unit = (int)(rest % GR);
                rest /= GR;
                j = (int)(rest % p.KS);
                chunk = (int)(rest / p.KS);
            }
            const int red0 = chunk * p.CK + 8 * unit + h;   // reduction channel of u = 0; u steps it by 2
            if (p.img == AVC_IMG_K4H) {   // CK / nchunk count DWORD channels: dword = bf16 pair of channels 2 red, 2 red + 1
                float lo[4], hi[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    lo[u] = value(m, 2 * (red0 + 2 * u), j);
                    hi[u] = value(m, 2 * (red0 + 2 * u) + 1, j);
                }
                avc_u32x4 w = {bh_pack(lo[0], hi[0]), bh_pack(lo[1], hi[1]), bh_pack(lo[2], hi[2]), bh_pack(lo[3], hi[3])};
                *(avc_u32x4*)((unsigned*)p.dst + 4 * g4) = w;
            } else {
                float v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = value(m, red0 + 2 * u, j);
                *(f32x4*)(p.dst + 4 * g4) = f32x4{v[0], v[1], v[2], v[3]};
            }
        }
        return;
    }
    total = total < limit ? total : limit;
    for (long e = first; e < total; e += stride) {
        int m, r, j, chunk;
        if (fastok) {
            int q = avc_fastdiv((int)e, p.Mp, inv_Mp);
            m = (int)e - q * p.Mp;
            int q2 = avc_fastdiv(q, p.CK, inv_CK);
            r = q - q2 * p.CK;
            chunk = avc_fastdiv(q2, p.KS, inv_KS);
            j = q2 - chunk * p.KS;
        } else {
            m = (int)(e % p.Mp);
            long rest = e / p.Mp;
            r = (int)(rest % p.CK);
            rest /= p.CK;
            j = (int)(rest % p.KS);
            chunk = (int)(rest / p.KS);
        }
        p.dst[e] = value(m, chunk * p.CK + r, j);
    }
}

// --------------------------------------------------------------------------
// host side
// --------------------------------------------------------------------------
int avc_conv_ck(const avc_tuning& tun, int KS) {
    const int ck5 = (tun.conv_ck5 == 16 || tun.conv_ck5 == 32) ? tun.conv_ck5 : 8;
    return KS >= 4 ? ck5 : (KS >= 2 ? 16 : 32);
}

// tile choice shared by the launcher and the plan (which sizes CK from it): 11 = 64 rows x 64 columns, 21 = 128 x 64
int avc_conv_pick_tile(const avc_tuning& tun, int Mp, int B, int Tout, int ngroups, int Kred) {
    // measured on MI355X (profiles/r01_conv_micro_*.log): at equal work the 64x64 tile beats 128x64 (more resident waves hide
    // the per-chunk LDS/DMA latency; the fp32 MFMA needs no bigger tile for operand reuse), so take the smallest tile unless
    // the grid gets very large.  (A 128x128 tile existed through round 2; no graded configuration selected it.)
    auto ntn = [&](int BN) { return Tout >= BN ? (long)B * avc_cdiv(Tout, BN) : (long)avc_cdiv(B, BN / Tout); };
    long t11 = (long)(Mp / 64) * ntn(64) * ngroups;
    // r2 sweeps (profiles/r02_tune_sweeps.log): launches whose workgroups carry little reduction work -- the grouped bank
    // (k = 1..8 over 80 mel rows) and 1x1 convs over <= 256 channels -- keep gaining from 64x64 tiles up to ~8k
    // workgroups (their cost is the epilogue: more, smaller workgroups overlap it better); the k = 5, 128-channel convs
    // switch to 128x64 beyond ~4k (B = 1024 inference: 7.54 vs 7.66 ms)
    const long thr11 = (ngroups > 1 || (Kred > 0 && Kred <= 256)) ? tun.tile_thr11 : (tun.tile_thr11 < 4095 ? tun.tile_thr11 : 4095);
    // r3 (profiles/r03_conv_micro.log): a deep 1x1 reduction (the 1104 -> 128 in_conv: 35 chunks of 32 channels) moves 8 KiB of
    // weights per chunk per 64 rows; with 128-row tiles the source tile is fetched once for twice the rows: 111.7 vs 131.5 us
    // forward, 138.5 vs 153.3 us input gradient at B = 256
    if (ngroups == 1 && Kred >= 1024 && Mp >= 128 && t11 >= 512) return 21;
    if (ngroups == 1 && Mp >= 1024 && t11 >= 4096) return 21;   // ... and its input gradient (1024 rows): 138.5 vs 153.3 us
    return t11 <= thr11 ? 11 : 21;
}
long avc_conv_num_wgs(int tile, int Mp, int B, int Tout, int ngroups) {
    int BM = (tile / 10 == 1) ? 64 : 128, BN = 64;
    long ntn = Tout >= BN ? (long)B * avc_cdiv(Tout, BN) : (long)avc_cdiv(B, BN / Tout);
    return (long)(Mp / BM) * ntn * ngroups;
}
// K-chunk depth: layers that cannot put two workgroups on every CU are latency-bound per chunk
// (global -> LDS round trip vs. ~1.3k MFMA cycles), so they take twice the channels per chunk
int avc_conv_ck_for(const avc_tuning& tun, int KS, long wgs, int mode, int stride, int Tout, int tile) {
    int ck = avc_conv_ck(tun, KS);
    if (KS >= 4 && wgs <= tun.ck16_wgs) {  // measured (r1 conv micro): CK=16 wins up to 2 workgroups per CU, loses beyond
        ConvGeom q = conv_geom(mode, stride, Tout, KS, 64, 0);
        if (q.ROW <= 16 * AVC_CONV_NJ4) ck = 16;
        // at most one workgroup per CU: the forward kernel gains another ~6 % from four chunks of 32
        // (r1 sweep: T_l = 16/32 forward 26.0 -> 24.5 us; the dgrad variant does not move)
        int BM = (tile / 10 == 1) ? 64 : 128;
        if (KS == 5 && mode == 0 && wgs <= tun.ck32_wgs && tile != 11 && q.ROW <= 16 * AVC_CONV_NJ4 && AVC_CONV_STAGES * (size_t)(KS * 32 * BM + 32 * q.ROW) * 4 <= 144 * 1024) ck = 32;
    }
    return ck;
}

static size_t conv_lds_bytes(const ConvArgs& a, int BM, int BN) {
    size_t worst = 0;
    for (int gi = 0; gi < a.ngroups; ++gi) {
        ConvGeom q = conv_geom(a.mode, a.stride, a.Tout, a.g[gi].KS, BN, 0);
        size_t AS = (size_t)a.g[gi].KS * a.g[gi].CK * BM, XS = (size_t)a.g[gi].CK * q.ROW;
        size_t bytes = (size_t)(a.bf16 == AVC_COMPUTE_BF16S ? AVC_CONV_STAGES_BH : AVC_CONV_STAGES) * (AS + XS) * 4;
        worst = bytes > worst ? bytes : worst;
    }
    return worst + 16;
}

static int conv_ntiles_n(const ConvArgs& a, int BN) {
    if (a.Tout >= BN) return a.B * avc_cdiv(a.Tout, BN);
    int worst = 1;   // (the groups of a grouped launch may fit different numbers of short samples into a tile)
    for (int gi = 0; gi < a.ngroups; ++gi) {
        const int n = avc_cdiv(a.B, conv_geom(a.mode, a.stride, a.Tout, a.g[gi].KS, BN, 0).SPT);
        worst = n > worst ? n : worst;
    }
    return worst;
}

// ragged forward launches (inference): the straight-line instances the model uses + the generic one
template <int WM, int BF>
static void conv_launch_rag(const ConvArgs& a, int fast, dim3 grid, dim3 block, size_t lds, hipStream_t stream) {
    if (fast == 1) hipLaunchKernelGGL((conv_gemm_kernel<WM, 1, false, 5, 1, 1, BF, false, true>), grid, block, lds, stream, a);
    else if (fast == -1) hipLaunchKernelGGL((conv_gemm_kernel<WM, 1, false, -1, 0, 1, BF, false, true>), grid, block, lds, stream, a);
    else if (fast == 14) hipLaunchKernelGGL((conv_gemm_kernel<WM, 1, false, 1, 4, 1, BF, false, true>), grid, block, lds, stream, a);
    else hipLaunchKernelGGL((conv_gemm_kernel<WM, 1, false, 0, 0, 1, BF, false, true>), grid, block, lds, stream, a);
}

template <int WM, int KG, int BF>
static void conv_launch_variant(const ConvArgs& a, bool mir, int fast, dim3 grid, dim3 block, size_t lds, hipStream_t stream) {
    if (mir && fast == 1) hipLaunchKernelGGL((conv_gemm_kernel<WM, 1, true, 5, 1, KG, BF>), grid, block, lds, stream, a);
    else if (mir && fast == 2) hipLaunchKernelGGL((conv_gemm_kernel<WM, 1, true, 5, 2, KG, BF>), grid, block, lds, stream, a);
    else if (mir) hipLaunchKernelGGL((conv_gemm_kernel<WM, 1, true, 0, 0, KG, BF>), grid, block, lds, stream, a);
    else if (fast == 1) hipLaunchKernelGGL((conv_gemm_kernel<WM, 1, false, 5, 1, KG, BF>), grid, block, lds, stream, a);
    else if (fast == 2) hipLaunchKernelGGL((conv_gemm_kernel<WM, 1, false, 5, 2, KG, BF>), grid, block, lds, stream, a);
    else if (fast == 4 && KG == 1) hipLaunchKernelGGL((conv_gemm_kernel<WM, 1, false, 5, 4, 1, BF>), grid, block, lds, stream, a);
    else if (fast == -1 && KG == 1) hipLaunchKernelGGL((conv_gemm_kernel<WM, 1, false, -1, 0, 1, BF>), grid, block, lds, stream, a);
    else if (fast == 14) hipLaunchKernelGGL((conv_gemm_kernel<WM, 1, false, 1, 4, KG, BF>), grid, block, lds, stream, a);   // 1x1, 32-channel chunks
    else hipLaunchKernelGGL((conv_gemm_kernel<WM, 1, false, 0, 0, KG, BF>), grid, block, lds, stream, a);
}

// tile walk (opt-in, avc_tuning.conv_walk; exact fp32 only): the k = 5 straight-line chunk both ways, the grouped bank launch, the 1x1
// convs on 64- and 128-row tiles.  false: no WALK instance for this launch
static bool conv_launch_walk(const ConvArgs& a, int tile, bool mir, int fast, dim3 grid, dim3 block, size_t lds, hipStream_t stream) {
    if (tile == 11 && fast == 1 && mir) hipLaunchKernelGGL((conv_gemm_kernel<1, 1, true, 5, 1, 1, 0, false, false, true>), grid, block, lds, stream, a);
    else if (tile == 11 && fast == 1) hipLaunchKernelGGL((conv_gemm_kernel<1, 1, false, 5, 1, 1, 0, false, false, true>), grid, block, lds, stream, a);
    else if (tile == 11 && fast == -1 && !mir) hipLaunchKernelGGL((conv_gemm_kernel<1, 1, false, -1, 0, 1, 0, false, false, true>), grid, block, lds, stream, a);
    else if (tile == 11 && fast == 14 && !mir) hipLaunchKernelGGL((conv_gemm_kernel<1, 1, false, 1, 4, 1, 0, false, false, true>), grid, block, lds, stream, a);
    else if (tile == 21 && fast == 14 && !mir) hipLaunchKernelGGL((conv_gemm_kernel<2, 1, false, 1, 4, 1, 0, false, false, true>), grid, block, lds, stream, a);
    else return false;
    return true;
}
// fused InstanceNorm epilogue (ConvINFuse): forward launches on 64 x 64 tiles -- the k = 5 chunk at 8 / 16 channels, the 1x1 chunk, the generic
// chunk loop (other kernel sizes), each with one or two split-K wave groups
template <int KG, int BF>
static void conv_launch_inf(const ConvArgs& a, int fast, dim3 grid, dim3 block, size_t lds, hipStream_t stream) {
    if (fast == 1) hipLaunchKernelGGL((conv_gemm_kernel<1, 1, false, 5, 1, KG, BF, false, false, false, true>), grid, block, lds, stream, a);
    else if (fast == 2) hipLaunchKernelGGL((conv_gemm_kernel<1, 1, false, 5, 2, KG, BF, false, false, false, true>), grid, block, lds, stream, a);
    else if (fast == 14) hipLaunchKernelGGL((conv_gemm_kernel<1, 1, false, 1, 4, KG, BF, false, false, false, true>), grid, block, lds, stream, a);
    else hipLaunchKernelGGL((conv_gemm_kernel<1, 1, false, 0, 0, KG, BF, false, false, false, true>), grid, block, lds, stream, a);
}
// fused InstanceNorm-backward epilogue (ConvINBwd): input-gradient launches on 64 x 64 tiles -- the mirrored k = 5 chunk at 8 / 16 channels with
// and without the stride-2 column-parity split, the 1x1 chunk, the generic chunk loop, each with one or two split-K wave groups
template <int KG, int BF>
static void conv_launch_inb(const ConvArgs& a, bool mir, int fast, dim3 grid, dim3 block, size_t lds, hipStream_t stream) {
    if (a.par && mir && fast == 1) hipLaunchKernelGGL((conv_gemm_kernel<1, 1, true, 5, 1, KG, BF, true, false, false, false, true>), grid, block, lds, stream, a);
    else if (a.par && mir) hipLaunchKernelGGL((conv_gemm_kernel<1, 1, true, 5, 2, KG, BF, true, false, false, false, true>), grid, block, lds, stream, a);
    else if (a.par && fast == 1) hipLaunchKernelGGL((conv_gemm_kernel<1, 1, false, 5, 1, KG, BF, true, false, false, false, true>), grid, block, lds, stream, a);
    else if (a.par) hipLaunchKernelGGL((conv_gemm_kernel<1, 1, false, 5, 2, KG, BF, true, false, false, false, true>), grid, block, lds, stream, a);
    else if (mir && fast == 1) hipLaunchKernelGGL((conv_gemm_kernel<1, 1, true, 5, 1, KG, BF, false, false, false, false, true>), grid, block, lds, stream, a);
    else if (mir && fast == 2) hipLaunchKernelGGL((conv_gemm_kernel<1, 1, true, 5, 2, KG, BF, false, false, false, false, true>), grid, block, lds, stream, a);
    else if (mir) hipLaunchKernelGGL((conv_gemm_kernel<1, 1, true, 0, 0, KG, BF, false, false, false, false, true>), grid, block, lds, stream, a);
    else if (fast == 14) hipLaunchKernelGGL((conv_gemm_kernel<1, 1, false, 1, 4, KG, BF, false, false, false, false, true>), grid, block, lds, stream, a);
    else hipLaunchKernelGGL((conv_gemm_kernel<1, 1, false, 0, 0, KG, BF, false, false, false, false, true>), grid, block, lds, stream, a);
}
static bool conv_walk_instance(int tile, bool mir, int fast) {
    return (tile == 11 && (fast == 1 || ((fast == -1 || fast == 14) && !mir))) || (tile == 21 && fast == 14 && !mir);
}

// 64 x 128 tiles (WN = 2): the k = 5 straight-line chunk, forward and mirrored input gradient
template <int BF>
static void conv_launch_wide(const ConvArgs& a, bool mir, dim3 grid, dim3 block, size_t lds, hipStream_t stream) {
    if (mir) hipLaunchKernelGGL((conv_gemm_kernel<1, 2, true, 5, 1, 1, BF>), grid, block, lds, stream, a);
    else hipLaunchKernelGGL((conv_gemm_kernel<1, 2, false, 5, 1, 1, BF>), grid, block, lds, stream, a);
}

// stride-2 dgrad with one column parity per wave (half the MFMAs of the zero-upsampled correlation)
template <int KG, int BF>
static void conv_launch_par(const ConvArgs& a, bool mir, int fast, dim3 grid, dim3 block, size_t lds, hipStream_t stream) {
    if (mir && fast == 1) hipLaunchKernelGGL((conv_gemm_kernel<1, 1, true, 5, 1, KG, BF, true>), grid, block, lds, stream, a);
    else if (mir) hipLaunchKernelGGL((conv_gemm_kernel<1, 1, true, 5, 2, KG, BF, true>), grid, block, lds, stream, a);
    else if (fast == 1) hipLaunchKernelGGL((conv_gemm_kernel<1, 1, false, 5, 1, KG, BF, true>), grid, block, lds, stream, a);
    else hipLaunchKernelGGL((conv_gemm_kernel<1, 1, false, 5, 2, KG, BF, true>), grid, block, lds, stream, a);
}

// Can the InstanceNorm of this conv's output rows run inside its epilogue (ConvINFuse)?  Exact-fp32 forward launches on 64 x 64 tiles whose
// 64 columns are whole rows of whole samples, contiguous [B][C][T] outputs, nothing else in the epilogue.
bool avc_conv_in_fusable(const ConvArgs& a, const avc_tuning& tun, int res_mode, int Tres) {
    // the residual the fused rows join: whole 16-byte vectors of a row of the matching length (an odd-length source of a ceil-mode pool,
    // model.py:319, keeps the row kernel)
    const int Tn_ = a.Tout * a.ops;
    if (res_mode != AVC_RES_NONE && !((res_mode == AVC_RES_IDENTITY && Tres == Tn_) || (res_mode == AVC_RES_UP2 && 2 * Tres == Tn_) ||
                                      (res_mode == AVC_RES_AVGPOOL2 && Tres == 2 * Tn_))) return false;
    if (!tun.conv_in_fuse || a.mode != 0 || a.ngroups != 1 || a.rag.tile) return false;
    const bool bh = a.bf16 == AVC_COMPUTE_BF16S;   // bf16 pair storage: pair rows in, pair rows out, no pixel shuffle (its rows are "planar")
    if (bh ? (a.img != AVC_IMG_K4H || !a.pairs || a.ops != 1 || (a.M & 1) || a.x.ps != 1 || a.x.st != 1) : (a.bf16 != AVC_COMPUTE_F32 || a.img != AVC_IMG_K4 || a.pairs)) return false;
    if (a.Tout != 16 && a.Tout != 32 && a.Tout != 64) return false;
    if (a.ops != 1 && a.ops != 2) return false;
    if (a.ops == 2 && (a.M & 1)) return false;
    if (a.act || a.res_mode != AVC_RES_NONE || a.g[0].out2 || a.g[0].mask) return false;
    const int C = (bh ? a.M / 2 : a.M) / a.ops, Tn = a.Tout * a.ops;   // rows per sample (pair rows with bh)
    if (a.ot != 1 || a.oc != Tn || a.ob != (long)C * Tn) return false;
    if (conv_geom(0, a.stride, a.Tout, a.g[0].KS, 64, 0).SPT != 64 / a.Tout) return false;
    return avc_conv_pick_tile(tun, a.Mp, a.B, a.Tout, 1, a.Cred * a.g[0].KS * (bh ? 2 : 1)) == 11;
}

// ... and of an input-gradient launch: may the InstanceNorm BACKWARD of its output rows run in its epilogue (ConvINBwd)?  Exact-fp32 dgrad on
// 64 x 64 tiles whose columns are whole rows; contiguous [B][M][T] output and residual-gradient rows; only the "to primary" join.
bool avc_conv_inb_fusable(const ConvArgs& a, const avc_tuning& tun) {
    if (!tun.conv_in_fuse || a.mode != 1 || a.ngroups != 1 || a.rag.tile) return false;
    const bool bh = a.bf16 == AVC_COMPUTE_BF16S;   // bf16 pair storage (round 6): pair rows in, pair rows out; strides in dwords over M / 2 pair rows per sample
    if (bh ? (a.img != AVC_IMG_K4H || !a.pairs || (a.M & 1) || a.x.ps != 1 || a.x.st != 1) : (a.bf16 != AVC_COMPUTE_F32 || a.img != AVC_IMG_K4 || a.pairs)) return false;
    if (a.Tout != 16 && a.Tout != 32 && a.Tout != 64) return false;
    if (a.ops != 1 || a.act || a.g[0].out2 || a.g[0].mask || a.g[0].bias) return false;
    const long Mr = bh ? a.M / 2 : a.M;
    if (a.ot != 1 || a.oc != a.Tout || a.ob != Mr * a.Tout) return false;
    if (a.res_mode != AVC_RES_NONE) {
        if (!a.res_to_primary || a.rt != 1 || a.rc != a.Tres || a.rb != Mr * a.Tres) return false;
        if (!((a.res_mode == AVC_RES_IDENTITY && a.Tres == a.Tout) || (a.res_mode == AVC_RES_POOLT && 2 * a.Tres == a.Tout) ||
              (a.res_mode == AVC_RES_UPT && a.Tres == 2 * a.Tout))) return false;
    }
    if (conv_geom(1, a.stride, a.Tout, a.g[0].KS, 64, 0).SPT != 64 / a.Tout) return false;
    return avc_conv_pick_tile(tun, a.Mp, a.B, a.Tout, 1, a.Cred * a.g[0].KS * (bh ? 2 : 1)) == 11;
}

// returns 0 on success, negative on unsupported geometry
int avc_launch_conv(const ConvArgs& a_in, hipStream_t stream, int force_tile, const avc_tuning& tun) {
    if (a_in.img == AVC_IMG_X3 || force_tile == 97) return avc_launch_conv_x3(a_in, stream, tun);
    ConvArgs a = a_in;
    a.dbg = tun.conv_ablation;
    if (a.ngroups < 1 || a.ngroups > AVC_MAX_GROUPS) return -1;
    if (a.Mp % 128 != 0) return -2;
    if (a.x.ps != 1 && a.x.ps != 2) return -2;   // (pixel-unshuffled dy views of the decoder: upsample factors are 1 or 2)
    for (int gi = 0; gi < a.ngroups; ++gi)
        if (a.mode == 0 && (a.g[gi].padL >= a.Tsrc || a.g[gi].padR >= a.Tsrc)) return -6;  // reference: "Padding size should be less than ..."
    int tile = force_tile == 22 ? 21 : force_tile;  // (the 128x128 tile was measured no better and dropped)
    if (tile == 98) return -8;   // (round 2's one-shot short-row kernel; gone)
    const bool wide_ok = a.ngroups == 1 && a.g[0].KS == 5 && a.g[0].CK == 8 && a.stride == 1 && a.Tout >= 64 && !a.rag.tile && a.x.ps == 1;
    if (tile == 0) {
        tile = avc_conv_pick_tile(tun, a.Mp, a.B, a.Tout, a.ngroups, a.Cred * a.g[0].KS * (a.bf16 == AVC_COMPUTE_BF16S ? 2 : 1));
        // 64 x 128: half the weight-image traffic per column; only while the launch keeps the chip full
        if (tile == 11 && wide_ok && tun.tile12_wgs > 0 && !a.in.out && !a.inb.dy) {   // (the fused InstanceNorm epilogues live on 64 x 64 tiles)
            const long ntn = a.Tout >= 128 ? (long)a.B * avc_cdiv(a.Tout, 128) : (long)avc_cdiv(a.B, 128 / a.Tout);
            if ((long)(a.Mp / 64) * ntn >= tun.tile12_wgs) tile = 12;
        }
    }
    if (a.in.out) {   // fused InstanceNorm epilogue: only what avc_conv_in_fusable admits, on the tile it assumed
        if (!avc_conv_in_fusable(a, tun, a.in.res ? a.in.res_mode : AVC_RES_NONE, a.in.Tres) || (force_tile != 0 && force_tile != 11) || tile != 11) return -2;
        if (!a.in.mean || !a.in.rstd || a.in.C != a.M / a.ops) return -2;
    }
    if (a.inb.dy) {   // fused InstanceNorm-backward epilogue
        if (a.in.out || !avc_conv_inb_fusable(a, tun) || (force_tile != 0 && force_tile != 11) || tile != 11) return -2;
        if (!a.inb.y || !a.inb.mean || !a.inb.rstd || a.inb.C != a.M) return -2;
    }
    if (tile == 12 && !wide_ok) return -2;
    if (tile != 11 && tile != 21 && tile != 12) return -2;
    const int BM = (tile == 21) ? 128 : 64, BN = (tile == 12) ? 128 : 64;
    for (int gi = 0; gi < a.ngroups; ++gi) {
        ConvGeom q = conv_geom(a.mode, a.stride, a.Tout, a.g[gi].KS, BN, 0);
        if (a.g[gi].CK % 8 != 0) return -2;
        if (q.ROW > 16 * AVC_CONV_NJ4) return -3;
    }
    size_t lds = conv_lds_bytes(a, BM, BN);
    const bool rag = a.rag.tile != nullptr;
    if (rag && (a.mode != 0 || a.Tout < BN || a.rag.ntiles < 1)) return -2;   // (ragged launches: forward only; the caller passes Tout >= 64 for the geometry)
    dim3 grid(rag ? a.rag.ntiles : conv_ntiles_n(a, BN), a.Mp / BM, a.ngroups);
    // split-K groups: only where the grid leaves CUs or SIMD slots idle (<= 1 workgroup per CU)
    int kgroups = 1;
    // (bf16 pair storage: the small layers are bound by launch / prologue latency, a second wave group only adds to it -- measured,
    // profiles/r03_bf16s_tune.log)
    if (!rag && a.bf16 != AVC_COMPUTE_BF16S && tile == 11 && a.ngroups == 1 && (long)grid.x * grid.y <= tun.kg_wgs && a.g[0].nchunk >= 4 && 2 * lds <= 160 * 1024 && !a.dbg) kgroups = 2;
    lds *= kgroups;
    if ((a.in.out || a.inb.dy) && lds < AVC_IN_LDS_BYTES) lds = AVC_IN_LDS_BYTES;
    if (lds > 160 * 1024) return -5;
    if (tun.conv_min_lds > 0 && (size_t)tun.conv_min_lds > lds && tun.conv_min_lds <= 160 * 1024) lds = (size_t)tun.conv_min_lds;   // (fewer co-resident workgroups)
    dim3 block(AVC_THREADS * kgroups);
    double flops = 0;
    for (int gi = 0; gi < a.ngroups; ++gi)
        flops += 2.0 * a.M * a.Cred * (a.bf16 == AVC_COMPUTE_BF16S ? 2 : 1) * a.g[gi].KS * (double)a.B * (a.mode == 0 ? a.Tout : a.Tsrc);
    ProfScope ps(a.mode == 0 ? AVC_K_CONV_FWD : AVC_K_CONV_DGRAD, flops, 0.0, stream);
    const bool mir = a.mode == 1 && a.mirror;
    // the model's kernel_size (5) with the chunk depths the plan uses gets straight-line chunks
    const int fast = (a.ngroups == 1 && a.g[0].KS == 5) ? (a.g[0].CK == 8 ? 1 : (a.g[0].CK == 16 ? 2 : (a.g[0].CK == 32 ? 4 : 0)))
                     : ((a.ngroups > 1 && a.mode == 0 && tun.bank_switch) ? -1
                        : ((a.ngroups == 1 && a.g[0].KS == 1 && a.g[0].CK == 32 && tun.bank_switch) ? 14 : 0));
    const int bf = a.bf16 == AVC_COMPUTE_BF16 ? 1 : (a.bf16 == AVC_COMPUTE_BF16S ? 2 : 0);
    if (bf == 2 && (a.img != AVC_IMG_K4H || a.x.ps != 1 || a.x.st != 1)) return -2;   // bf16 pair tensors: time-contiguous dword rows, pair weight image
    if (a.pairs && (bf != 2 || a.ops != 1 || a.ot != 1 || (a.M & 1) || (a.res_mode != AVC_RES_NONE && a.rt != 1))) return -2;
    a.par = tun.dgrad_par && a.mode == 1 && a.stride == 2 && tile == 11 && a.ngroups == 1 && (fast == 1 || fast == 2) && !a.dbg &&
            a.g[0].padL == 2 && (a.Tout >= 64 || (a.Tout % 2 == 0 && 64 % a.Tout == 0));
    // ---- tile walk: persistent workgroups over the column tiles of a row slab (conv_gemm_kernel).  Only where every tile of a walk has
    // the geometry of the first one: rows of >= 64 frames (a walker keeps its first frame), or whole groups of short samples
    a.walk_n = 1;
    a.walk_rem = (int)grid.x;
    a.walk_db = 0;
    bool walk = false;
    if (tun.conv_walk != 0 && !a.in.out && !a.inb.dy && !rag && !a.par && bf == 0 && kgroups == 1 && AVC_CONV_STAGES == 2 && conv_walk_instance(tile, mir, fast)) {
        const int ntn = (int)grid.x;
        const int tps = a.Tout >= BN ? avc_cdiv(a.Tout, BN) : 1;
        const int SPT = a.Tout >= BN ? 1 : conv_geom(a.mode, a.stride, a.Tout, a.g[0].KS, BN, 0).SPT;
        const bool geom_ok = a.Tout >= BN || (a.ngroups == 1 && a.B % SPT == 0);
        long per_cu = (long)(160 * 1024) / (long)lds;                 // resident workgroups per CU: LDS ...
        const long reg_cu = BM == 128 ? 3 : 4;                          // ... and registers (126-129 / 100-115 VGPRs per lane)
        per_cu = per_cu < reg_cu ? per_cu : reg_cu;
        per_cu = per_cu < tun.conv_walk ? per_cu : tun.conv_walk;
        long W = 256 * per_cu / ((long)grid.y * grid.z);
        const bool forced = tun.conv_walk < 0;   // (test aid: exactly -conv_walk walkers per row slab, whatever the chip holds)
        if (forced) W = -tun.conv_walk < tps ? tps : -tun.conv_walk;
        W = W / tps * tps;
        if (geom_ok && W >= tps && W < ntn && (forced || (long)ntn >= (tun.conv_walk_min > 1 ? tun.conv_walk_min : 2) * W)) {
            a.walk_n = avc_cdiv(ntn, (int)W);
            a.walk_rem = ntn - (a.walk_n - 1) * (int)W;
            a.walk_db = a.Tout >= BN ? (int)W / tps : (int)W * SPT;
            grid.x = (unsigned)W;
            walk = true;
        }
    }
#define AVC_BF3(CALL_)                       \
    do {                                     \
        if (bf == 2) { CALL_(2); }           \
        else if (bf == 1) { CALL_(1); }      \
        else { CALL_(0); }                   \
    } while (0)
    if (a.inb.dy) {
        if (bf == 2) conv_launch_inb<1, 2>(a, mir, fast, grid, block, lds, stream);   // (pair storage runs without split-K wave groups)
        else if (kgroups == 2) conv_launch_inb<2, 0>(a, mir, fast, grid, block, lds, stream);
        else conv_launch_inb<1, 0>(a, mir, fast, grid, block, lds, stream);
    } else if (a.in.out) {
        const int f = (fast == 1 || fast == 2 || fast == 14) ? fast : 0;   // (the k = 5 chunk at 32 channels takes the generic loop)
        if (bf == 2) conv_launch_inf<1, 2>(a, f, grid, block, lds, stream);   // (pair storage runs without split-K wave groups)
        else if (kgroups == 2) conv_launch_inf<2, 0>(a, f, grid, block, lds, stream);
        else conv_launch_inf<1, 0>(a, f, grid, block, lds, stream);
    } else if (walk) {
        if (!conv_launch_walk(a, tile, mir, fast, grid, block, lds, stream)) return -2;
    } else if (rag) {
        if (bf == 2) return -2;   // (ragged plans run fp32 storage)
        const int f = (fast == 1 || fast == -1 || fast == 14) ? fast : 0;
#define AVC_C_RAG2(BF_) conv_launch_rag<2, BF_>(a, f, grid, block, lds, stream)
#define AVC_C_RAG1(BF_) conv_launch_rag<1, BF_>(a, f, grid, block, lds, stream)
        if (tile == 21) { if (bf) AVC_C_RAG2(1); else AVC_C_RAG2(0); }
        else { if (bf) AVC_C_RAG1(1); else AVC_C_RAG1(0); }
    } else if (a.par) {
#define AVC_C_PAR2(BF_) conv_launch_par<2, BF_>(a, mir, fast, grid, block, lds, stream)
#define AVC_C_PAR1(BF_) conv_launch_par<1, BF_>(a, mir, fast, grid, block, lds, stream)
        if (kgroups == 2) AVC_BF3(AVC_C_PAR2);
        else AVC_BF3(AVC_C_PAR1);
    } else if (tile == 12) {
#define AVC_C_WIDE(BF_) conv_launch_wide<BF_>(a, mir, grid, block, lds, stream)
        AVC_BF3(AVC_C_WIDE);
    } else if (tile == 21) {
#define AVC_C_V21(BF_) conv_launch_variant<2, 1, BF_>(a, mir, fast, grid, block, lds, stream)
        AVC_BF3(AVC_C_V21);
    } else if (kgroups == 2) {
#define AVC_C_V12(BF_) conv_launch_variant<1, 2, BF_>(a, mir, fast, grid, block, lds, stream)
        AVC_BF3(AVC_C_V12);
    } else {
#define AVC_C_V11(BF_) conv_launch_variant<1, 1, BF_>(a, mir, fast, grid, block, lds, stream)
        AVC_BF3(AVC_C_V11);
    }
#undef AVC_BF3
    return (int)hipGetLastError();
}

long avc_pack_total(const PackArgs& p) {
    if (p.img == AVC_IMG_X3) return (long)p.nchunk * x3_arows(p.KS) * p.Mp * 4;
    return (long)p.nchunk * p.KS * p.CK * p.Mp;
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

int avc_launch_pack_table(const PackArgs* dev_tab, const void* dev_blk, int nblk, double bytes, const float* params, float* ws, hipStream_t stream) {
    ProfScope ps_(AVC_K_PACK, 0.0, bytes, stream);
    hipLaunchKernelGGL(pack_weight_table_kernel, dim3(nblk), dim3(AVC_THREADS), 0, stream, dev_tab, (const int2*)dev_blk, params, ws);
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
