// PROBE (round 3, NOT in the product, NOT measured on the GPU): conv_wgrad.hip with AVC_WGRAD_STAGES LDS stages and the producer waves
// AVC_WGRAD_STAGES - 1 chunks ahead of the consumers (fast path; the generic path keeps distance one).  DESIGN section 7 (1d): the product
// kernel's staging does not overlap its MFMAs (ablation: 19.8 + 16 + 39.6 = 75.3 us) because x / dy come from HBM with a round trip longer
// than one chunk's products.  To try: copy over csrc/conv_wgrad.hip, build, tests/test_ops_conv.py + tests/test_engine.py, bench.
// Verified on the simulator only (tests/emu: LDS-DMA is synchronous there, so stage indexing / reuse order are checked, the partial
// vmcnt waits are correct by construction: every lane of every fast-path DMA piece always loads, so a producer wave issues a launch-constant
// number of DMA instructions per chunk, and LDS-DMA loads complete in issue order).
#ifndef AVC_WGRAD_STAGES
#define AVC_WGRAD_STAGES 3
#endif
// Weight gradient of the reflect-padded Conv1d on the fp32 MFMA
// (reference: autograd of model.py:21-32; dW[co,ci,j] = sum_{b,t} dy[b,co,t] * xpad[b,ci,t*s+j]).
//
// GEMM view: M = co, N = (ci, tap), K = (b, t).  A workgroup owns a 64co x 64ci
// x KS tile (4 waves, each 32co x 32ci x KS accumulators) and a contiguous range
// of 32-column K-chunks; partial tiles go to a slab [split][Cout][Cin][KS] that a
// second kernel sums in a fixed order (deterministic, no atomics).  The bias
// gradient (row sums of dy) is produced by the ci-tile-0 workgroups from the dy
// tile they already hold in LDS.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <string.h>
#include <type_traits>
#include <vector>

#include "avc_common.h"
#include "avc_internal.h"
#include "conv_x3_shared.h"
#include "bf16_pairs.h"

// LDS row strides (floats).  General form: odd (33 / (spc XSEG) | 1): the 32 lanes of a half-wave read one column of 32 different
// rows with ds_read_b32, conflict-free.  LIN instances (whole 32-column chunks of a stride-1 layer): rows are 16-byte aligned with
// (stride / 4) ODD, so that a lane's four k-steps -- columns 8 g + 4 h + u of the chunk in BOTH operands -- are one ds_read_b128
// and the 16 lanes of each ds_read_b128 service group (MI355X_MICROARCH.md, LDS) fall on 16 different 16-byte bank slots.
static constexpr __host__ __device__ int wg_dyrow(bool lin) { return lin ? 36 : 33; }
static constexpr __host__ __device__ int wg_xrow_lin(int KS) {
    int r = (31 + KS + 3) / 4;       // 16-byte units covering XSEG = 31 + KS positions
    return 4 * (r | 1);              // KS = 1: 36, 2..5: 36, 6..8: 44
}
#define WG_THREADS 512   // 4 consumer waves (MFMA) + 4 producer waves (LDS-DMA issue)

static inline __device__ long src_chan_off(const ConvSrc& s, int c) {
    return (s.ps == 1) ? (long)c * s.sc : (long)(c / s.ps) * s.sc + (c % s.ps);
}

// KS taps, NB 32-wide ci blocks per wave, WCO waves along co (4/WCO along ci):
//   workgroup tile = (32*WCO) co  x  (32*NB*(4/WCO)) ci  x  KS taps.
// <5,1,2> is the 64x64 tile of the k=5 layers; WCO=4 (128co x 32ci) suits Cin that is not a multiple
// of 64 (the 80-mel bank convs); <1,4,4> (128co x 128ci) gives the 1x1 convs / Linears four
// accumulators per wave, i.e. the arithmetic intensity per staged element that the taps give k=5.
//
// Warp-specialised: with ~80 accumulator registers per wave the kernel runs one MFMA wave per SIMD,
// and a wave issues in order -- every DMA address computation or exposed LDS round trip inside the
// k-loop is matrix-pipe idle time (measured: 62 % MFMA duty even with DMA and barriers removed).
// So waves 0-3 (consumers) execute nothing but fragment reads and MFMAs, and waves 4-7 (producers)
// issue the next chunk's global->LDS DMAs, drain them and meet the consumers at one barrier per chunk.
//
// X3 (LIN layers -- whole 32-column chunks of a stride-1 conv, k = 1..8; opt-in, avc_set_tuning("wgrad_x3", 1)): the consumers form the products from three bf16 terms per
// operand on v_mfma_f32_32x32x16_bf16 (conv_x3_shared.h: fp32-level accuracy in 2.7x fewer matrix-pipe cycles).  Both operands are
// activations here, so both are split in registers: per 16 columns a lane reads its 8 dy values and the 12 x values that its
// KS shifted windows cover, splits each ONCE, and assembles the KS B fragments by pairing registers (v_perm) -- 20 splits and
// 30 MFMAs per block where the fp32 path issues 40 MFMAs of twice the length.  Producers, tiles, slabs: unchanged.
//
// BF == 2 (bf16 PAIR storage, bf16_pairs.h): x and dy are dword tensors [B][C/2][T].  The producers stage PAIR rows -- the same code
// over half as many rows, every DMA'd dword brings two channels -- and the consumers feed v_mfma_f32_32x32x16_bf16: a lane's
// 8 k-values are 8 consecutive columns of ITS channel, i.e. one half of 8 consecutive dwords of its pair row, gathered with one
// v_perm_b32 per two columns (lanes 2p and 2p + 1 read the same LDS words: a broadcast, no extra bandwidth).
#ifndef AVC_EMU
static __device__ __forceinline__ unsigned bh_sel(unsigned d0, unsigned d1, unsigned sel) { return __builtin_amdgcn_perm(d1, d0, sel); }
#else
static inline unsigned bh_sel(unsigned d0, unsigned d1, unsigned sel) {   // sel = 0x05040100 (low halves) or 0x07060302 (high halves)
    return sel == 0x05040100u ? ((d0 & 0xffffu) | (d1 << 16)) : ((d0 >> 16) | (d1 & 0xffff0000u));
}
#endif
// s_waitcnt vmcnt(n), n = 0..63 (gfx9 encoding: vmcnt = simm16[15:14] : simm16[3:0]; expcnt / lgkmcnt fields at their maxima)
static __device__ __forceinline__ void wg_wait_dma(int n) {
#define WG_W(k) case k: __builtin_amdgcn_s_waitcnt(((k) & 15) | (((k) >> 4) << 14) | (7 << 4) | (15 << 8)); break;
#define WG_W4(k) WG_W(k) WG_W(k + 1) WG_W(k + 2) WG_W(k + 3)
#define WG_W16(k) WG_W4(k) WG_W4(k + 4) WG_W4(k + 8) WG_W4(k + 12)
    switch (n < 0 ? 0 : (n > 63 ? 63 : n)) {   // (more than 63 allowed in flight: waiting for all but 63 is stricter, still correct)
        WG_W16(0) WG_W16(16) WG_W16(32) WG_W16(48)
    }
#undef WG_W16
#undef WG_W4
#undef WG_W
}
// workgroup barrier WITHOUT the compiler's vmcnt(0) drain in front of it (the explicit wait above is the synchronisation of the LDS-DMA)
static __device__ __forceinline__ void wg_bare_barrier() {
#ifdef AVC_EMU
    emu::block_barrier();
#else
    asm volatile("s_barrier" ::: "memory");
#endif
}

template <int KS, int NB, int WCO, bool LIN, int BF, bool X3 = false>
__global__ void __launch_bounds__(WG_THREADS) conv_wgrad_kernel(const WgradBatch bt) {
    // which layer of the batch this workgroup works for (wave-uniform scan of <= 16 entries)
    int layer = 0;
    for (int i = 1; i < bt.nlayers; ++i) layer = ((int)blockIdx.x >= bt.L[i].wg_begin) ? i : layer;
    const WgradArgs& a = bt.L[layer];
    const int dbg = bt.dbg;
    constexpr int WCI = 4 / WCO;            // waves along ci
    constexpr int TCO = 32 * WCO;           // co rows per workgroup
    constexpr int TCI = 32 * NB * WCI;      // ci rows per workgroup
    constexpr int NACC = KS * NB;
    constexpr bool BH = BF == 2;
    constexpr int RCO = BH ? TCO / 2 : TCO, RCI = BH ? TCI / 2 : TCI;   // LDS / source rows of the two operand tiles (pair rows with BH)
    HIP_DYNAMIC_SHARED(float, smem)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);  // scalar: LDS-DMA bases must be provably wave-uniform
    const bool producer = wave8 >= 4;
    const int wave = wave8 & 3;   // consumer: tile position; producer: which pieces of a chunk it stages
    const int ptid = tid & 255;   // thread index inside its role group
    const int wave_m = wave / WCI, wave_n = wave % WCI, li = lane & 31, h = lane >> 5;
    const int ci_tiles = avc_cdiv(a.Cin, TCI);
    const int local = (int)blockIdx.x - a.wg_begin;
    const int z = local / a.tiles, tile = local - z * a.tiles;   // split index, (co, ci) tile
    const int co0 = (tile / ci_tiles) * TCO, ci0 = (tile % ci_tiles) * TCI;
    const int co0r = BH ? co0 >> 1 : co0, ci0r = BH ? ci0 >> 1 : ci0;                   // ... in source rows
    const int CoutR = BH ? a.Cout >> 1 : a.Cout, CinR = BH ? a.Cin >> 1 : a.Cin;
    const float* xptr = a.x.ptr;
    const float* dyptr = a.dy.ptr;
    float* slabp = a.slab;
    float* dbp = a.dbslab;
    const int Tc = a.Tc, spc = a.spc;
    const int lgTc = 31 - __builtin_clz(Tc);
    const int XSEG = (Tc - 1) * a.stride + KS;
    constexpr int WG_DYROW = wg_dyrow(LIN);
    const int XROW = LIN ? wg_xrow_lin(KS) : ((spc * XSEG) | 1);
    const int DYS = RCO * WG_DYROW, XS = RCI * XROW;
    const int DYSP = (DYS + 63) & ~63, XSP = (XS + 63) & ~63;  // stage strides: whole 64-float DMA pieces
    constexpr int NST = AVC_WGRAD_STAGES;
    float* dyT = smem;               // [NST][RCO][WG_DYROW]
    float* xT = smem + NST * DYSP;   // [NST][RCI][XROW]
    const bool do_db = (dbp != nullptr) && (ci0 == 0);
    const float inv_xrow = 1.0f / (float)XROW;

    float dbsum = 0.f, dbsum1 = 0.f;

    // Both operand tiles go global -> LDS by dword DMA (each lane its own source address, so the
    // reflect padding and the padded LDS rows cost no staging registers).  Two stages: chunk c+1
    // lands while chunk c multiplies.  Both stages are zero-filled once.
    for (int e = tid; e < NST * (DYSP + XSP); e += WG_THREADS) smem[e] = 0.f;

    // Fast path (one sample per chunk, whole chunks): every lane of every DMA piece always loads --
    // LDS positions that hold no tile element (row padding, rows past Cout / Cin) get a clamped,
    // valid address instead of an exec-masked branch; they are never read, or feed accumulator rows
    // that are never stored.  The chunk-invariant byte offsets live in producer registers, the
    // chunk origin is a scalar base: one SADDR-form DMA instruction per piece, ~no address VALU.
    constexpr int NPD = (RCO * WG_DYROW + 255) / 256;  // dy pieces per wave
    constexpr int NPX = (RCI * (LIN ? wg_xrow_lin(KS) : (KS == 1 ? 33 : 71)) + 255) / 256;  // x pieces per wave (XROW <= 71, or 33 for stride-1 1x1)
    // ... and the same for chunks that hold spc whole short samples (T_l = 16, 8, ...): there even the
    // reflection is chunk-invariant, so the x offsets are complete and only the base moves.
    const bool fastm = (spc > 1) && (a.Tout == Tc) && (a.B % spc == 0) && (XSP <= NPX * 256);
    const bool fastp = fastm || ((spc == 1) && (a.Tout % 32 == 0) && (XSP <= NPX * 256));
    unsigned dyo[NPD], xo[NPX];
    int xq[NPX];
    if (fastm && producer) {
#pragma unroll
        for (int i = 0; i < NPD; ++i) {
            const int f = (wave + 4 * i) * 64 + lane;
            int row = f / WG_DYROW, qcol = f - row * WG_DYROW;
            row = row < RCO ? row : RCO - 1;
            qcol = qcol < 32 ? qcol : 31;
            int co = co0r + row;
            co = co < CoutR ? co : CoutR - 1;
            const int sl = qcol >> lgTc, tl = qcol & (Tc - 1);
            dyo[i] = 4u * (unsigned)((long)sl * a.dy.sb + src_chan_off(a.dy, co) + (long)tl * a.dy.st);
        }
#pragma unroll
        for (int i = 0; i < NPX; ++i) {
            const int f = (wave + 4 * i) * 64 + lane;
            int row = avc_fastdiv(f, XROW, inv_xrow), pp = f - row * XROW;
            row = row < RCI ? row : RCI - 1;
            int sl = pp / XSEG, p = pp - sl * XSEG;
            if (sl >= spc) { sl = spc - 1; p = XSEG - 1; }  // the odd-stride padding column
            int ci = ci0r + row;
            ci = ci < CinR ? ci : CinR - 1;
            int r = avc_reflect(p - a.padL, a.Tin);
            r = r < 0 ? 0 : (r >= a.Tin ? a.Tin - 1 : r);
            xo[i] = 4u * (unsigned)((long)sl * a.x.sb + src_chan_off(a.x, ci) + (long)r * a.x.st);
            xq[i] = 0;
        }
    } else if (fastp && producer) {
#pragma unroll
        for (int i = 0; i < NPD; ++i) {
            const int f = (wave + 4 * i) * 64 + lane;
            int row = f / WG_DYROW, qcol = f - row * WG_DYROW;
            row = row < RCO ? row : RCO - 1;
            qcol = qcol < 32 ? qcol : 31;
            int co = co0r + row;
            co = co < CoutR ? co : CoutR - 1;
            dyo[i] = 4u * (unsigned)(src_chan_off(a.dy, co) + (long)qcol * a.dy.st);
        }
#pragma unroll
        for (int i = 0; i < NPX; ++i) {
            const int f = (wave + 4 * i) * 64 + lane;
            int row = avc_fastdiv(f, XROW, inv_xrow), p = f - row * XROW;
            row = row < RCI ? row : RCI - 1;
            p = p < XSEG ? p : XSEG - 1;
            int ci = ci0r + row;
            ci = ci < CinR ? ci : CinR - 1;
            xo[i] = 4u * (unsigned)src_chan_off(a.x, ci);
            xq[i] = p;
        }
    }
    auto issue_fast = [&](int chunk, int buf) {
        float* dd = dyT + buf * DYSP;
        float* xd = xT + buf * XSP;
        if (fastm) {
            const float* dyb = dyptr + (long)chunk * spc * a.dy.sb;
            const float* xb = xptr + (long)chunk * spc * a.x.sb;
#pragma unroll
            for (int i = 0; i < NPD; ++i)
                if ((wave + 4 * i) * 64 < DYSP) avc_glds4_s(dyb, dyo[i], dd + (wave + 4 * i) * 64);
#pragma unroll
            for (int i = 0; i < NPX; ++i)
                if ((wave + 4 * i) * 64 < XSP) avc_glds4_s(xb, xo[i], xd + (wave + 4 * i) * 64);
            return;
        }
        const int cb = chunk / a.chunks_per_sample;
        const int t0 = (chunk - cb * a.chunks_per_sample) * 32;
        const float* dyb = dyptr + ((long)cb * a.dy.sb + (long)t0 * a.dy.st);
        const float* xb = xptr + (long)cb * a.x.sb;
        const int v0 = t0 * a.stride - a.padL;
        const unsigned st4 = 4u * (unsigned)a.x.st;
#pragma unroll
        for (int i = 0; i < NPD; ++i)
            if ((wave + 4 * i) * 64 < DYSP) avc_glds4_s(dyb, dyo[i], dd + (wave + 4 * i) * 64);
        if (v0 >= 0 && v0 + XSEG <= a.Tin) {  // interior chunk: no reflection anywhere in the tile
            const float* xbv = xb + (long)v0 * a.x.st;
#pragma unroll
            for (int i = 0; i < NPX; ++i)
                if ((wave + 4 * i) * 64 < XSP) avc_glds4_s(xbv, xo[i] + (unsigned)xq[i] * st4, xd + (wave + 4 * i) * 64);
        } else {
#pragma unroll
            for (int i = 0; i < NPX; ++i)
                if ((wave + 4 * i) * 64 < XSP) {
                    int r = avc_reflect(v0 + xq[i], a.Tin);
                    r = r < 0 ? 0 : (r >= a.Tin ? a.Tin - 1 : r);  // (only positions no valid dy column multiplies)
                    avc_glds4_s(xb, xo[i] + (unsigned)r * st4, xd + (wave + 4 * i) * 64);
                }
        }
    };

    auto issue = [&](int chunk, int buf) {
        if (fastp) {
            issue_fast(chunk, buf);
            return;
        }
        float* dd = dyT + buf * DYSP;
        float* xd = xT + buf * XSP;
        int cb, t0;
        if (spc == 1) {
            cb = chunk / a.chunks_per_sample;
            t0 = (chunk - cb * a.chunks_per_sample) * 32;
        } else {
            cb = chunk * spc;
            t0 = 0;
        }
        for (int piece = wave; piece * 64 < DYS; piece += 4) {
            int f = piece * 64 + lane;
            int row = f / WG_DYROW, qcol = f - row * WG_DYROW;
            if (f < DYS && qcol < 32) {
                int sl = qcol >> lgTc, tl = qcol & (Tc - 1);  // Tc is a power of two
                int b = cb + sl, t = t0 + tl, co = co0r + row;
                if (b < a.B && t < a.Tout && co < CoutR)
                    avc_glds4(dyptr + ((long)b * a.dy.sb + src_chan_off(a.dy, co) + (long)t * a.dy.st), dd + piece * 64);
                else
                    dd[f] = 0.f;
            }
        }
        for (int piece = wave; piece * 64 < XS; piece += 4) {
            int f = piece * 64 + lane;
            if (f < XS) {
                int row = avc_fastdiv(f, XROW, inv_xrow), pp = f - row * XROW;
                int sl = pp / XSEG, p = pp - sl * XSEG;
                int b = cb + sl, ci = ci0r + row;
                int r = avc_reflect(t0 * a.stride + p - a.padL, a.Tin);
                if (sl < spc && b < a.B && ci < CinR && r >= 0 && r < a.Tin)
                    avc_glds4(xptr + ((long)b * a.x.sb + src_chan_off(a.x, ci) + (long)r * a.x.st), xd + piece * 64);
                else
                    xd[f] = 0.f;
            }
        }
    };

    const int c_begin = z * a.chunks_per_wg;
    int c_end = c_begin + a.chunks_per_wg;
    if (c_end > a.total_chunks) c_end = a.total_chunks;

    // producers run `dist` chunks ahead: NST - 1 in the fast path (launch-constant DMA instruction count per chunk and wave -> partial
    // vmcnt waits), one in the generic path (conditional DMAs + LDS zero stores: it drains)
    const int dist = fastp ? NST - 1 : 1;
    int n_w = 0;   // DMA instructions THIS producer wave issues per fast-path chunk
#pragma unroll
    for (int i = 0; i < NPD; ++i) n_w += ((wave + 4 * i) * 64 < DYSP) ? 1 : 0;
#pragma unroll
    for (int i = 0; i < NPX; ++i) n_w += ((wave + 4 * i) * 64 < XSP) ? 1 : 0;
    __syncthreads();  // zero fill complete before the first DMA lands
    if (producer)
        for (int d = 0; d < dist; ++d)
            if (c_begin + d < c_end) issue(c_begin + d, d % NST);
    __syncthreads();  // (drains everything issued so far: the first `dist` chunks have landed)
    constexpr int TPR = 256 / RCO;  // producer threads per dy row in the bias-gradient partial sum
    constexpr int CPT = 32 / TPR;
    // The two roles run separate loops that meet at one s_barrier per chunk (the barrier counts wave
    // arrivals, not call sites).  Producer side of the barrier: the next stage has landed (the DMA is
    // drained by the s_waitcnt in front of it); consumer side: the current stage is free again.
    if (producer) {
        for (int chunk = c_begin; chunk < c_end; ++chunk) {
            const int buf = (chunk - c_begin) % NST;
            // stage of chunk + dist: last read while chunk + dist - NST <= chunk - 1 multiplied, free since the barrier that ended that iteration
            const bool more = (chunk + dist < c_end) && !((dbg & 1) && chunk > c_begin);
            if (more) issue(chunk + dist, (chunk + dist - c_begin) % NST);
            if (do_db) {
                const float* dr = dyT + buf * DYSP + (ptid / TPR) * WG_DYROW + (ptid % TPR) * CPT;
#pragma unroll
                for (int k = 0; k < CPT; ++k) {
                    if constexpr (BH) {
                        const unsigned d = bh_as_u32(dr[k]);
                        dbsum += bh_lo(d);
                        dbsum1 += bh_hi(d);
                    } else {
                        dbsum += dr[k];
                    }
                }
            }
            if (!(dbg & 4)) {
                // chunk + 1 must have landed; the chunks behind it that are already issued may stay in flight
                int last = chunk + dist < c_end ? chunk + dist : c_end - 1;
                if ((dbg & 1) && chunk > c_begin) last = chunk + 1;
                const int ahead = last - (chunk + 1) > 0 ? last - (chunk + 1) : 0;
                wg_wait_dma(fastp ? n_w * ahead : 0);
                wg_bare_barrier();
            }
        }
        if (do_db) {
#pragma unroll
            for (int o = 1; o < TPR; o <<= 1) {
                dbsum += __shfl_xor(dbsum, o);
                if (BH) dbsum1 += __shfl_xor(dbsum1, o);
            }
            if constexpr (BH) {
                const int co = co0 + 2 * (ptid / TPR);
                if ((ptid % TPR) == 0 && co < a.Cout) {
                    dbp[(long)z * a.db_stride + co] = dbsum;
                    dbp[(long)z * a.db_stride + co + 1] = dbsum1;
                }
            } else {
                const int co = co0 + ptid / TPR;
                if ((ptid % TPR) == 0 && co < a.Cout) dbp[(long)z * a.db_stride + co] = dbsum;
            }
        }
        return;
    }

    {
    f32x16 acc[NACC];
#pragma unroll
    for (int j = 0; j < NACC; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    for (int chunk = c_begin; chunk < c_end; ++chunk) {
        const int buf = (chunk - c_begin) % NST;
        if (!(dbg & 2)) {
            const float* arow = dyT + buf * DYSP + (BH ? (wave_m * 32 + li) >> 1 : wave_m * 32 + li) * WG_DYROW;
            const float* brow = xT + buf * XSP + (BH ? (wave_n * NB * 32 + li) >> 1 : wave_n * NB * 32 + li) * XROW;
            constexpr int NBROW = BH ? 16 : 32;   // LDS rows between the ci blocks of a wave
            // fragments of k-step s+1 are requested before the MFMAs of step s are queued.  LIN (template): whole
            // 32-column chunks of a stride-1 layer -- column 2s+h of the chunk is element 2s+h of both
            // LDS rows, so every fragment address is base + immediate (the general form costs ~30
            // hoisted address registers, which decides whether two workgroups fit on a CU).
            {
                const float* arow_h = arow + h;
                const float* brow_h = brow + h;
                auto ldfrag = [&](int s, float& av, float (&bv)[NACC]) {
                    const float* bp;
                    if constexpr (LIN) {
                        av = arow_h[2 * s];
                        bp = brow_h + 2 * s;
                    } else {
                        const int qcol = 2 * s + h;
                        const int sl = qcol >> lgTc, tl = qcol & (Tc - 1);  // Tc is a power of two
                        av = arow[qcol];
                        bp = brow + sl * XSEG + tl * a.stride;
                    }
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                        for (int j = 0; j < KS; ++j) bv[nb * KS + j] = bp[nb * 32 * XROW + j];
                };
                if constexpr (BH) {
                    // two blocks of 16 columns per chunk; lane-half h owns columns 16 kb + 8 h .. + 7 in BOTH operands
                    const unsigned sel = (li & 1) ? 0x07060302u : 0x05040100u;   // this lane's channel = low / high half of its pair row
                    constexpr int NX = 8 + KS - 1;
                    auto fetch = [&](int kb, unsigned (&ad)[8], unsigned (&xd)[NB][LIN ? NX : 8 * KS]) {
                        if constexpr (LIN) {   // 16-byte aligned rows: ds_read_b128
                            const float* ap = arow + 16 * kb + 8 * h;
                            const float* bp = brow + 16 * kb + 8 * h;
#pragma unroll
                            for (int i4 = 0; i4 < 2; ++i4) {
                                const f32x4 v = *(const f32x4*)(ap + 4 * i4);
#pragma unroll
                                for (int i = 0; i < 4; ++i) ad[4 * i4 + i] = bh_as_u32(v[i]);
                            }
#pragma unroll
                            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                                for (int i4 = 0; i4 < (NX + 3) / 4; ++i4) {
                                    const f32x4 v = *(const f32x4*)(bp + nb * NBROW * XROW + 4 * i4);
#pragma unroll
                                    for (int i = 0; i < 4; ++i)
                                        if (4 * i4 + i < NX) xd[nb][4 * i4 + i] = bh_as_u32(v[i]);
                                }
                        } else {               // short samples / strided layers: every column has its own window
#pragma unroll
                            for (int i = 0; i < 8; ++i) {
                                const int qcol = 16 * kb + 8 * h + i;
                                const int sl = qcol >> lgTc, tl = qcol & (Tc - 1);
                                ad[i] = bh_as_u32(arow[qcol]);
                                const float* bp = brow + sl * XSEG + tl * a.stride;
#pragma unroll
                                for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                                    for (int j = 0; j < KS; ++j) xd[nb][i * KS + j] = bh_as_u32(bp[nb * NBROW * XROW + j]);
                            }
                        }
                    };
                    auto block = [&](const unsigned (&ad)[8], const unsigned (&xd)[NB][LIN ? NX : 8 * KS]) {
                        avc_u32x4 at;
#pragma unroll
                        for (int q4 = 0; q4 < 4; ++q4) at[q4] = bh_sel(ad[2 * q4], ad[2 * q4 + 1], sel);
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                            for (int j = 0; j < KS; ++j) {
                                avc_u32x4 b;
#pragma unroll
                                for (int q4 = 0; q4 < 4; ++q4) {
                                    if constexpr (LIN) b[q4] = bh_sel(xd[nb][j + 2 * q4], xd[nb][j + 2 * q4 + 1], sel);
                                    else b[q4] = bh_sel(xd[nb][(2 * q4) * KS + j], xd[nb][(2 * q4 + 1) * KS + j], sel);
                                }
                                acc[nb * KS + j] = avc_mfma_bf16x8(at, b, acc[nb * KS + j]);
                            }
                    };
                    unsigned a0[8], x0[NB][LIN ? NX : 8 * KS], a1[8], x1[NB][LIN ? NX : 8 * KS];
                    fetch(0, a0, x0);
                    fetch(1, a1, x1);   // (requested before the first block's MFMAs are issued)
                    block(a0, x0);
                    block(a1, x1);
                } else if constexpr (X3 && LIN && !BF) {
                    // two blocks of 16 columns: lane-half h owns columns 16 kb + 8 h .. + 7 of the chunk
                    constexpr int NX = 8 + KS - 1;   // x values under the KS shifted windows of 8 columns
                    auto fetch = [&](int kb, float (&av)[8], float (&xv)[NB][NX]) {   // 16-byte aligned rows (LIN): ds_read_b128
                        const float* ap = arow + 16 * kb + 8 * h;
                        const float* bp = brow + 16 * kb + 8 * h;
#pragma unroll
                        for (int i4 = 0; i4 < 2; ++i4) {
                            const f32x4 v = *(const f32x4*)(ap + 4 * i4);
#pragma unroll
                            for (int i = 0; i < 4; ++i) av[4 * i4 + i] = v[i];
                        }
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                            for (int i4 = 0; i4 < (NX + 3) / 4; ++i4) {
                                const f32x4 v = *(const f32x4*)(bp + nb * 32 * XROW + 4 * i4);
#pragma unroll
                                for (int i = 0; i < 4; ++i)
                                    if (4 * i4 + i < NX) xv[nb][4 * i4 + i] = v[i];
                            }
                    };
                    auto block = [&](const float (&av)[8], const float (&xv)[NB][NX]) {
                        unsigned ah[8], am[8], al[8];
#pragma unroll
                        for (int i = 0; i < 8; ++i) x3_split(av[i], ah[i], am[i], al[i]);
                        avc_u32x4 at[3];
#pragma unroll
                        for (int q4 = 0; q4 < 4; ++q4) {
                            at[0][q4] = x3_pair(ah[2 * q4], ah[2 * q4 + 1]);
                            at[1][q4] = x3_pair(am[2 * q4], am[2 * q4 + 1]);
                            at[2][q4] = x3_pair(al[2 * q4], al[2 * q4 + 1]);
                        }
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb) {
                            unsigned xh[NX], xm[NX], xl[NX];
#pragma unroll
                            for (int i = 0; i < NX; ++i) x3_split(xv[nb][i], xh[i], xm[i], xl[i]);
#pragma unroll
                            for (int j = 0; j < KS; ++j) {
                                avc_u32x4 b0, b1, b2;
#pragma unroll
                                for (int q4 = 0; q4 < 4; ++q4) {
                                    b0[q4] = x3_pair(xh[j + 2 * q4], xh[j + 2 * q4 + 1]);
                                    b1[q4] = x3_pair(xm[j + 2 * q4], xm[j + 2 * q4 + 1]);
                                    b2[q4] = x3_pair(xl[j + 2 * q4], xl[j + 2 * q4 + 1]);
                                }
                                f32x16& c = acc[nb * KS + j];
                                // small terms first
                                c = avc_mfma_bf16x8(at[2], b0, c);
                                c = avc_mfma_bf16x8(at[0], b2, c);
                                c = avc_mfma_bf16x8(at[1], b1, c);
                                c = avc_mfma_bf16x8(at[1], b0, c);
                                c = avc_mfma_bf16x8(at[0], b1, c);
                                c = avc_mfma_bf16x8(at[0], b0, c);
                            }
                        }
                    };
                    float a0[8], x0[NB][NX], a1[8], x1[NB][NX];
                    fetch(0, a0, x0);
                    fetch(1, a1, x1);   // (requested before the first block's MFMAs are issued)
                    block(a0, x0);
                    block(a1, x1);
                } else if constexpr (LIN) {
                    // Four groups of 8 columns per chunk; in group g lane-half h owns columns 8 g + 4 h + u, u = k-step 0..3, in BOTH
                    // operands (the sum over columns does not care about the order).  Its four dy values are ONE ds_read_b128, and the
                    // 4 + KS - 1 x values under its KS shifted windows are (KS + 6) / 4 more: 3 reads per 20 MFMAs at k = 5, where
                    // round 2 issued 6 ds_read_b32 per 5 (the consumer waves run alone on their SIMD: every read they issue is
                    // matrix-pipe idle time, profiles/r02_mfma_probe.log).  BF: the same fragments rounded to bf16, one
                    // v_mfma_f32_32x32x8_bf16 per tap and group.
                    constexpr int NX4 = (KS + 6) / 4;          // 16-byte reads covering 4 + KS - 1 values
                    auto ldgrp = [&](int g, f32x4& av, f32x4 (&xv)[NB][NX4]) {
                        const int c = 8 * g + 4 * h;
                        av = *(const f32x4*)(arow + c);
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                            for (int i4 = 0; i4 < NX4; ++i4) xv[nb][i4] = *(const f32x4*)(brow + nb * 32 * XROW + c + 4 * i4);
                    };
                    f32x4 av[2], xv[2][NB][NX4];
                    ldgrp(0, av[0], xv[0]);
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int cur = g & 1;
                        if (g + 1 < 4) ldgrp(g + 1, av[cur ^ 1], xv[cur ^ 1]);
                        __builtin_amdgcn_sched_barrier(0);  // the next group's reads stay in front of the MFMAs they overlap with ...
                        if constexpr (BF) {
                            const avc_s16x4 ap = avc_pack_bf16x4(av[cur][0], av[cur][1], av[cur][2], av[cur][3]);
#pragma unroll
                            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                                for (int j = 0; j < KS; ++j) {
                                    const avc_s16x4 bp = avc_pack_bf16x4(xv[cur][nb][j >> 2][j & 3], xv[cur][nb][(j + 1) >> 2][(j + 1) & 3],
                                                                         xv[cur][nb][(j + 2) >> 2][(j + 2) & 3], xv[cur][nb][(j + 3) >> 2][(j + 3) & 3]);
                                    acc[nb * KS + j] = avc_mfma_bf16(ap, bp, acc[nb * KS + j]);
                                }
                        } else {
#pragma unroll
                            for (int u = 0; u < 4; ++u)
#pragma unroll
                                for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                                    for (int j = 0; j < KS; ++j)
                                        acc[nb * KS + j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cur][u], xv[cur][nb][(u + j) >> 2][(u + j) & 3], acc[nb * KS + j], 0, 0, 0);
                        }
                        __builtin_amdgcn_sched_barrier(0);  // ... and one group ahead only
                    }
                } else if constexpr (BF) {
                    // four k-steps (8 columns) per v_mfma_f32_32x32x8_bf16: slot j of lane-half h carries
                    // column 2(4g + j) + h of the chunk in both operands; operands are rounded to bf16 here
                    float av4[2][4], bv4[2][4][NACC];
#pragma unroll
                    for (int j = 0; j < 4; ++j) ldfrag(j, av4[0][j], bv4[0][j]);
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        const int cur = g4 & 1;
                        if (g4 + 1 < 4) {
#pragma unroll
                            for (int j = 0; j < 4; ++j) ldfrag(4 * (g4 + 1) + j, av4[cur ^ 1][j], bv4[cur ^ 1][j]);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        const avc_s16x4 ap = avc_pack_bf16x4(av4[cur][0], av4[cur][1], av4[cur][2], av4[cur][3]);
#pragma unroll
                        for (int k = 0; k < NACC; ++k)
                            acc[k] = avc_mfma_bf16(ap, avc_pack_bf16x4(bv4[cur][0][k], bv4[cur][1][k], bv4[cur][2][k], bv4[cur][3][k]), acc[k]);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                } else {
                float av[2], bv[2][NACC];
                ldfrag(0, av[0], bv[0]);
#pragma unroll
                for (int s = 0; s < 16; ++s) {
                    const int cur = s & 1;
                    if (s + 1 < 16) ldfrag(s + 1, av[cur ^ 1], bv[cur ^ 1]);
                    __builtin_amdgcn_sched_barrier(0);  // reads stay in front of the MFMAs they overlap with ...
#pragma unroll
                    for (int k = 0; k < NACC; ++k)
                        acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cur], bv[cur][k], acc[k], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);  // ... and one step ahead only (hoisting all 16 steps' reads spills)
                }
                }
            }
        }
        if (!(dbg & 4)) __syncthreads();
    }

    // ---- epilogue: partial tile -> slab[z][tap][co][ci]  (tap-major: the 32 lanes of a half-wave
    // hold 32 consecutive ci of one (tap, co) row -> 128-byte coalesced stores; the reduce kernel
    // restores the [co][ci][tap] parameter layout)
    float* slab = slabp + (long)z * a.slab_stride;
    if (!(dbg & 8))
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int ci = ci0 + (wave_n * NB + nb) * 32 + li;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            int co = co0 + wave_m * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (co < a.Cout && ci < a.Cin) {
#pragma unroll
                for (int j = 0; j < KS; ++j) slab[((long)j * a.Cout + co) * a.Cin + ci] = acc[nb * KS + j][r];
            }
        }
    }
    }
}

// out[e] = sum_z slab[z*stride + e]   (fixed order -> deterministic)
struct ReduceArgs {
    ReduceSeg seg[AVC_REDUCE_MAXSEG];
    int nseg;
};

__global__ void __launch_bounds__(AVC_THREADS) slab_reduce_kernel(const ReduceArgs a) {
    const ReduceSeg s = a.seg[blockIdx.y];
    const int plane = s.n / s.KS;  // slab is [tap][rows*Cin]; dst is [rows*Cin][tap]
    for (int e = blockIdx.x * AVC_THREADS + threadIdx.x; e < s.n; e += gridDim.x * AVC_THREADS) {
        // fixed summation order (deterministic: v + slab 0 + slab 1 + ..., as ever); 16 independent loads in flight per thread -- a
        // thread owns one or two elements, so the kernel's run time is (slabs / loads in flight) dependent HBM round trips: 4 in flight
        // were 6-8 trips for 23-30 slabs, 16 are 2 (slots past the last slab re-read it and are not added)
        float v = 0.f;
        for (int zz = 0; zz < s.nsplit; zz += 16) {
            float q[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int z = zz + k < s.nsplit ? zz + k : s.nsplit - 1;
                q[k] = s.slab[(long)z * s.stride + e];
            }
#pragma unroll
            for (int k = 0; k < 16; ++k)
                if (zz + k < s.nsplit) v += q[k];
        }
        if (s.KS == 1) {
            s.dst[e] = v;
        } else {
            int j = e / plane, rem = e - j * plane;
            s.dst[(long)rem * s.KS + j] = v;
        }
    }
}

// --------------------------------------------------------------------------
// tile shape per layer: returns (NB, WCO); tile = (32*WCO) co x (32*NB*(4/WCO)) ci
static void wgrad_shape(int Cin, int Cout, int KS, int* NB, int* WCO) {
    if (KS == 1 && Cin >= 96) {
        *NB = 4; *WCO = 4;      // 128 x 128
    } else if (Cin % 64 == 0 || KS == 1) {
        *NB = 1; *WCO = 2;      // 64 x 64
    } else {
        *NB = 1; *WCO = 4;      // 128 x 32: Cin = 80 wastes 17 % instead of 38 %
    }
}

static size_t wgrad_lds_bytes(const WgradArgs& a, int NB, int WCO) {
    const int half = a.bf16 == AVC_COMPUTE_BF16S ? 2 : 1;   // pair rows
    const int TCO = 32 * WCO / half, TCI = 32 * NB * (4 / WCO) / half;
    const int XSEG = (a.Tc - 1) * a.stride + a.KS;
    const bool lin = a.Tc == 32 && a.stride == 1;   // (the LIN kernel instances, wgrad_key)
    const int XROW = lin ? wg_xrow_lin(a.KS) : ((a.spc * XSEG) | 1), WG_DYROW = wg_dyrow(lin);
    return (size_t)AVC_WGRAD_STAGES * (((TCO * WG_DYROW + 63) & ~63) + ((TCI * XROW + 63) & ~63)) * 4 + 16;
}

// kernel instance a layer runs on: (KS, NB, WCO, LIN); layers with equal keys can share a launch
struct WgradKey {
    int KS, NB, WCO, lin;
    bool operator==(const WgradKey& o) const { return KS == o.KS && NB == o.NB && WCO == o.WCO && lin == o.lin; }
};
static WgradKey wgrad_key(const WgradArgs& a) {
    WgradKey k;
    k.KS = a.KS;
    wgrad_shape(a.Cin, a.Cout, a.KS, &k.NB, &k.WCO);
    if (k.KS == 1 && k.NB == 4 && wgrad_lds_bytes(a, 4, 4) > 158 * 1024) {  // LDS too small for the wide tile (many short samples)
        k.NB = 1;
        k.WCO = 2;
    }
    k.lin = (a.Tc == 32 && a.stride == 1) ? 1 : 0;
    return k;
}

// K-chunk geometry of one layer (32 columns of the (b, t) axis per chunk; short samples are packed)
void avc_wgrad_geometry(WgradArgs& a) {
    if (a.Tout >= 32) {
        a.Tc = 32;
        a.spc = 1;
        a.chunks_per_sample = avc_cdiv(a.Tout, 32);
        a.total_chunks = a.B * a.chunks_per_sample;
    } else {
        int p = 1;
        while (p < a.Tout) p <<= 1;
        a.Tc = p;
        a.spc = 32 / p;
        a.chunks_per_sample = 1;
        a.total_chunks = avc_cdiv(a.B, a.spc);
    }
    const WgradKey k = wgrad_key(a);
    a.tiles = avc_cdiv(a.Cout, 32 * k.WCO) * avc_cdiv(a.Cin, 32 * k.NB * (4 / k.WCO));
}

// Split-K factors of a batch.  Layers that share a kernel instance share a launch and get the SAME number
// of K-chunks per workgroup (every chunk costs the same there, so the launch is balanced); the count is
// chosen so that the launch has about `target_wgs` workgroups, but at least 4 chunks (128 columns) per
// workgroup so that the slab write + fixed-order reduce stay a small fraction of the work.
void avc_wgrad_plan_batch(WgradArgs* L, int n, int target_wgs) {
    if (target_wgs < 1) target_wgs = 256;
    for (int i = 0; i < n; ++i) avc_wgrad_geometry(L[i]);
    std::vector<char> done((size_t)n, 0);
    for (int i = 0; i < n; ++i) {
        if (done[i]) continue;
        const WgradKey k = wgrad_key(L[i]);
        long units = 0;
        for (int j = i; j < n; ++j)
            if (!done[j] && wgrad_key(L[j]) == k) units += (long)L[j].tiles * L[j].total_chunks;
        int cpw = (int)((units + target_wgs - 1) / target_wgs);
        if (cpw < 4) cpw = 4;
        // the launch must FIT the target (one workgroup per CU is resident: a 257th workgroup is a second round
        // that doubles the launch time): grow the chunk run until the per-layer round-ups fit
        for (;; ++cpw) {
            long wgs = 0;
            for (int j = i; j < n; ++j)
                if (!done[j] && wgrad_key(L[j]) == k) wgs += (long)L[j].tiles * avc_cdiv(L[j].total_chunks, cpw);
            if (wgs <= target_wgs || cpw >= (1 << 20)) break;
        }
        for (int j = i; j < n; ++j)
            if (!done[j] && wgrad_key(L[j]) == k) {
                int c = cpw < L[j].total_chunks ? cpw : L[j].total_chunks;
                L[j].chunks_per_wg = c;
                L[j].nsplit = avc_cdiv(L[j].total_chunks, c);
                done[j] = 1;
            }
    }
}

void avc_wgrad_plan(const avc_tuning& tun, int B, int Cin, int Cout, int Tout, int KS, int* Tc, int* spc, int* chunks_per_sample, int* total_chunks,
                    int* chunks_per_wg, int* nsplit) {
    WgradArgs a;
    memset(&a, 0, sizeof(a));
    a.B = B; a.Cin = Cin; a.Cout = Cout; a.Tout = Tout; a.KS = KS; a.stride = 1;
    avc_wgrad_plan_batch(&a, 1, tun.wgrad_target_wgs);
    *Tc = a.Tc; *spc = a.spc; *chunks_per_sample = a.chunks_per_sample; *total_chunks = a.total_chunks;
    *chunks_per_wg = a.chunks_per_wg; *nsplit = a.nsplit;
}


template <int KS, int NB, int WCO>
static int launch_wgrad_t(const WgradBatch& bt, int total_wgs, bool lin, int bf, bool x3, size_t lds, double flops, hipStream_t stream) {
    if (lds > 158 * 1024) return -3;
    dim3 grid(total_wgs);
    ProfScope ps(AVC_K_CONV_WGRAD, flops, 0.0, stream);
    if (bf == 2) {
        if (lin) hipLaunchKernelGGL((conv_wgrad_kernel<KS, NB, WCO, true, 2>), grid, dim3(WG_THREADS), lds, stream, bt);
        else hipLaunchKernelGGL((conv_wgrad_kernel<KS, NB, WCO, false, 2>), grid, dim3(WG_THREADS), lds, stream, bt);
    } else if (bf) {
        if (lin) hipLaunchKernelGGL((conv_wgrad_kernel<KS, NB, WCO, true, 1>), grid, dim3(WG_THREADS), lds, stream, bt);
        else hipLaunchKernelGGL((conv_wgrad_kernel<KS, NB, WCO, false, 1>), grid, dim3(WG_THREADS), lds, stream, bt);
    } else if (lin) {
        if constexpr (KS * NB <= 8) {
            if (x3) {
                hipLaunchKernelGGL((conv_wgrad_kernel<KS, NB, WCO, true, 0, true>), grid, dim3(WG_THREADS), lds, stream, bt);
                return (int)hipGetLastError();
            }
        }
        hipLaunchKernelGGL((conv_wgrad_kernel<KS, NB, WCO, true, 0>), grid, dim3(WG_THREADS), lds, stream, bt);
    } else hipLaunchKernelGGL((conv_wgrad_kernel<KS, NB, WCO, false, 0>), grid, dim3(WG_THREADS), lds, stream, bt);
    return (int)hipGetLastError();
}

template <int KS>
static int launch_wgrad_ks(const WgradBatch& bt, int total_wgs, const WgradKey& k, int bf, bool x3, size_t lds, double flops, hipStream_t stream) {
    return k.WCO == 4 ? launch_wgrad_t<KS, 1, 4>(bt, total_wgs, k.lin, bf, x3, lds, flops, stream)
                      : launch_wgrad_t<KS, 1, 2>(bt, total_wgs, k.lin, bf, x3, lds, flops, stream);
}

// launches every layer of the batch (planned by avc_wgrad_plan_batch, slabs assigned): one launch per kernel
// instance present, <= AVC_WGRAD_MAXL layers per launch
// ablation: timing-experiment bits of scripts/wgrad_ablate.py (results are wrong by construction when set)
int avc_launch_wgrad_batch(const WgradArgs* L, int n, hipStream_t stream, int ablation) {
    std::vector<char> done((size_t)n, 0);
    for (int i = 0; i < n; ++i) {
        if (done[i]) continue;
        const WgradArgs& a0 = L[i];
        if (a0.KS < 1 || a0.KS > 8) return -1;
        const WgradKey k = wgrad_key(a0);
        WgradBatch bt;
        memset(&bt, 0, sizeof(bt));
        bt.dbg = ablation;
        int wgs = 0;
        size_t lds = 0;
        double flops = 0;
        for (int j = i; j < n && bt.nlayers < AVC_WGRAD_MAXL; ++j) {
            if (done[j] || !(wgrad_key(L[j]) == k) || L[j].bf16 != a0.bf16) continue;
            if (L[j].padL >= L[j].Tin) return -6;
            WgradArgs& d = bt.L[bt.nlayers++];
            d = L[j];
            d.wg_begin = wgs;
            wgs += d.tiles * d.nsplit;
            size_t l = wgrad_lds_bytes(d, k.NB, k.WCO);
            lds = l > lds ? l : lds;
            flops += 2.0 * d.Cout * d.Cin * d.KS * (double)d.B * d.Tout;
            done[j] = 1;
        }
        const int bf = a0.bf16 == AVC_COMPUTE_BF16 ? 1 : (a0.bf16 == AVC_COMPUTE_BF16S ? 2 : 0);
        const bool x3 = a0.bf16 == AVC_COMPUTE_F32X3;
        if (bf == 2)
            for (int j = 0; j < bt.nlayers; ++j)
                if ((bt.L[j].Cin & 1) || (bt.L[j].Cout & 1) || bt.L[j].x.st != 1 || bt.L[j].dy.st != 1 || bt.L[j].x.ps != 1 || bt.L[j].dy.ps != 1) return -2;
        int rc;
        if (k.KS == 1 && k.NB == 4) rc = launch_wgrad_t<1, 4, 4>(bt, wgs, k.lin, bf, x3, lds, flops, stream);
        else switch (k.KS) {
            case 1: rc = launch_wgrad_ks<1>(bt, wgs, k, bf, x3, lds, flops, stream); break;
            case 2: rc = launch_wgrad_ks<2>(bt, wgs, k, bf, x3, lds, flops, stream); break;
            case 3: rc = launch_wgrad_ks<3>(bt, wgs, k, bf, x3, lds, flops, stream); break;
            case 4: rc = launch_wgrad_ks<4>(bt, wgs, k, bf, x3, lds, flops, stream); break;
            case 5: rc = launch_wgrad_ks<5>(bt, wgs, k, bf, x3, lds, flops, stream); break;
            case 6: rc = launch_wgrad_ks<6>(bt, wgs, k, bf, x3, lds, flops, stream); break;
            case 7: rc = launch_wgrad_ks<7>(bt, wgs, k, bf, x3, lds, flops, stream); break;
            default: rc = launch_wgrad_ks<8>(bt, wgs, k, bf, x3, lds, flops, stream); break;
        }
        if (rc) return rc;
        // (layers of this key beyond AVC_WGRAD_MAXL stay !done and open their own launch when the outer loop reaches them)
    }
    return 0;
}

int avc_launch_reduce_segs(const ReduceSeg* segs, int n, hipStream_t stream) {
    if (n < 1 || n > AVC_REDUCE_MAXSEG) return -1;
    ReduceArgs r;
    r.nseg = n;
    int maxn = 0;
    for (int i = 0; i < n; ++i) {
        r.seg[i] = segs[i];
        maxn = segs[i].n > maxn ? segs[i].n : maxn;
    }
    int blocks = avc_cdiv(maxn, AVC_THREADS);
    if (blocks > 256) blocks = 256;
    double rb = 0;
    for (int i = 0; i < n; ++i) rb += 4.0 * segs[i].n * (segs[i].nsplit + 1);
    ProfScope ps(AVC_K_REDUCE, 0.0, rb, stream);
    hipLaunchKernelGGL(slab_reduce_kernel, dim3(blocks, n), dim3(AVC_THREADS), 0, stream, r);
    return (int)hipGetLastError();
}

int avc_launch_reduce(const float* slab, long stride, int nsplit, int n, float* dst, int KS, hipStream_t stream) {
    ReduceArgs r;
    r.nseg = 1;
    r.seg[0].slab = slab;
    r.seg[0].dst = dst;
    r.seg[0].stride = stride;
    r.seg[0].n = n;
    r.seg[0].nsplit = nsplit;
    r.seg[0].KS = KS;
    int blocks = avc_cdiv(n, AVC_THREADS);
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(slab_reduce_kernel, dim3(blocks, 1), dim3(AVC_THREADS), 0, stream, r);
    return (int)hipGetLastError();
}
