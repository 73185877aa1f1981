// Weight gradient of the reflect-padded Conv1d on the fp32 MFMA
// (reference: autograd of model.py:21-32; dW[co,ci,j] = sum_{b,t} dy[b,co,t] * xpad[b,ci,t*s+j]).
//
// GEMM view: M = co, N = (ci, tap), K = (b, t).  A workgroup holds a 64co x 64ci x KS (or 128 x 32, 128 x 128) tile in
// accumulator registers and walks 32-column K-chunks of it.  Since round 4 a launch is a STREAM-K split of ALL its layers
// (avc_common.h, WgradArgs): exactly `grid` persistent workgroups, each owning one contiguous run of the launch's
// (layer, tile, chunk) sequence.  A workgroup that walked a tile's WHOLE K range writes the finished gradient (parameter
// layout) itself; otherwise it stores its partial tile into the tile's slot z, and ONE chip-wide reduce launch behind the
// batch's launches sums every tile's slots in the fixed order z = 0, 1, ... (bit-deterministic) into the flat gradient buffer.
// (An in-kernel reduce by the last workgroup to arrive at a tile was built first and measured SLOWER -- one workgroup reading
// nsplit x 80..131 KB serially, 44 GB/s: profiles/r04_wgrad_inkernel_reduce_ablate.log.)
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
#define WG_THREADS 512   // 4 consumer waves (MFMA) + 4 producer waves (LDS-DMA issue); CW = 8 instances: 8 + 4 waves = 768 threads
static constexpr __host__ __device__ int wg_threads(int CW) { return (CW + 4) * 64; }

// LDS stage slots per operand and chunks per barrier.  fp32 / bf16-operand / fp32x3 products: two slots, one chunk per barrier -- chunk c + 1
// lands while chunk c multiplies (40 us of MFMA per layer hide a round trip per chunk; three / four stages measured neutral there,
// profiles/r03_wgrad_stages_ab.log).  bf16 PAIR storage (BF == 2, round 6): a chunk multiplies for ~0.15 us, so what a 32-column chunk costs is
// everything around the products -- the rendezvous of eight waves (~0.27 us) and the producers' per-chunk round trip
// (profiles/r06_wgrad_steady_*.txt: 0.5 us of "neither" per chunk against 0.15 us of MFMA).  Those instances therefore work in GROUPS of G
// chunks: the producers issue the G chunks of group g + 1 into one half of 2 G slots while the consumers multiply the G chunks of group g out
// of the other half, and the two roles meet ONCE per group (the draining __syncthreads: everything of group g + 1 has landed).  G chunks
// (36 - 48 KB) in flight per workgroup instead of one.  [First built as a 4-slot ring with the producers three chunks ahead, partial
// s_waitcnt vmcnt(n) and bare s_barrier, one barrier per chunk: 2.58 -> 2.54 ms per step -- the in-flight bytes were not the bound, the
// barrier count was.]
static constexpr __host__ __device__ int wg_group(int BF, int NB) { return BF == 2 ? (NB >= 4 ? 2 : 4) : 1; }   // (the 128 x 128 1x1 tile: 20 KB per slot)
static constexpr __host__ __device__ int wg_stages(int BF, int NB) { return 2 * wg_group(BF, NB); }
#ifndef AVC_WGRAD_BHX_WAVES
#define AVC_WGRAD_BHX_WAVES 3   // minimum waves per SIMD of the bf16 k = 5 whole-chunk instance: 3 = 168 registers (it takes 137)
#endif

static inline __device__ long src_chan_off(const ConvSrc& s, int c) {
    return (s.ps == 1) ? (long)c * s.sc : (long)(c / s.ps) * s.sc + (c % s.ps);
}

#ifndef AVC_EMU
static __device__ __forceinline__ unsigned bh_sel(unsigned d0, unsigned d1, unsigned sel) { return __builtin_amdgcn_perm(d1, d0, sel); }
#else
static inline unsigned bh_sel(unsigned d0, unsigned d1, unsigned sel) {   // sel = 0x05040100 (low halves) or 0x07060302 (high halves)
    return sel == 0x05040100u ? ((d0 & 0xffffu) | (d1 << 16)) : ((d0 >> 16) | (d1 & 0xffff0000u));
}
#endif

template <int K>
struct KTag {
    static constexpr int value = K;
};

// ---- one workgroup's walk through the launch's (layer, tile, chunk) sequence.  Pure arithmetic on launch constants and the
// workgroup index: both roles of a workgroup (and every workgroup that shares a tile) derive the same numbers.
struct WgSeg {
    int layer, tile, c_begin, c_end;   // K-chunks [c_begin, c_end) of (co, ci) tile `tile` of layer `layer`
    int nsplit, z;                     // workgroups that share the tile, this one's index among them
};
struct WgSegIter {
    long b0, b1, G, Ctot;
    int w, layer;
    long g, g_end;   // chunk sequence numbers (tile * total_chunks + chunk) of the current layer still to do
    __device__ WgSegIter(const WgradBatch& bt) {
        Ctot = bt.L[0].cost_total;
        G = gridDim.x;
        w = blockIdx.x;
        b0 = ((long)w * Ctot + G - 1) / G;
        b1 = ((long)(w + 1) * Ctot + G - 1) / G;
        layer = -1;
        g = g_end = 0;
    }
    __device__ bool next(const WgradBatch& bt, WgSeg& s) {
        while (g >= g_end) {
            if (++layer >= bt.nlayers) return false;
            const WgradArgs& a = bt.L[layer];
            const int cc = a.chunk_cost;
            const long lay_end = a.cost_begin + (long)a.tiles * a.total_chunks * cc;
            if (b1 <= a.cost_begin || b0 >= lay_end) continue;
            // chunks of this layer whose start cost lies in [b0, b1)
            const long lo = b0 > a.cost_begin ? b0 - a.cost_begin : 0, hi = (b1 < lay_end ? b1 : lay_end) - a.cost_begin;
            g = (lo + cc - 1) / cc;
            g_end = (hi + cc - 1) / cc;
        }
        const WgradArgs& a = bt.L[layer];
        const int cc = a.chunk_cost;
        s.layer = layer;
        s.tile = (int)(g / a.total_chunks);
        s.c_begin = (int)(g - (long)s.tile * a.total_chunks);
        const long tile_last = (long)(s.tile + 1) * a.total_chunks;
        s.c_end = (int)((g_end < tile_last ? g_end : tile_last) - (long)s.tile * a.total_chunks);
        g += s.c_end - s.c_begin;
        // which workgroups share this tile: owner(chunk with start cost c) = floor(c G / C)
        const long cs = a.cost_begin + (long)s.tile * a.total_chunks * cc;
        const int w_first = (int)((cs * G) / Ctot), w_last = (int)(((cs + (long)(a.total_chunks - 1) * cc) * G) / Ctot);
        s.nsplit = w_last - w_first + 1;
        s.z = w - w_first;
        return true;
    }
};

// KS taps (RT: the layer's own tap count 1..KS is a run-time value -- the eight conv-bank members k = 1..8 share ONE launch; the
// chunk body is still straight-line code per tap count, selected by a wave-uniform switch), NB 32-wide ci blocks per wave, WCO
// waves along co (4/WCO along ci):   workgroup tile = (32*WCO) co  x  (32*NB*(4/WCO)) ci  x  taps.
// <5,false,1,2> is the 64x64 tile of the k=5 layers; WCO=4 (128co x 32ci) suits Cin that is not a multiple of 64 (the 80-mel bank
// convs); <1,false,4,4> (128co x 128ci) gives the 1x1 convs / Linears four accumulators per wave.
//
// Warp-specialised: with ~80 accumulator registers per wave the kernel runs one MFMA wave per SIMD, and a wave issues in order --
// every DMA address computation or exposed LDS round trip inside the k-loop is matrix-pipe idle time.  So waves 0-3 (consumers)
// execute nothing but fragment reads and MFMAs, and waves 4-7 (producers) issue the next chunk's global->LDS DMAs, drain them and
// meet the consumers at one barrier per chunk.  The two roles are two separate functions that walk the same segment sequence and
// meet at the same number of barriers (the barrier counts wave arrivals, not call sites): their registers do not add up.
//
// X3 (LIN layers; opt-in, avc_set_tuning("wgrad_x3", 1)): the consumers form the products from three bf16 terms per operand on
// v_mfma_f32_32x32x16_bf16 (conv_x3_shared.h: fp32-level accuracy in 2.7x fewer matrix-pipe cycles).
//
// BF == 2 (bf16 PAIR storage, bf16_pairs.h): x and dy are dword tensors [B][C/2][T].  The producers stage PAIR rows and the consumers
// feed v_mfma_f32_32x32x16_bf16: a lane's 8 k-values are 8 consecutive columns of ITS channel, i.e. one half of 8 consecutive dwords of
// its pair row, gathered with one v_perm_b32 per two columns.
template <int KS, bool RT, int NB, int WCO, bool LIN, int BF, int CW = 4>
struct WgCfg {
    static constexpr int WCI = CW / WCO;           // consumer waves along ci
    static constexpr int TCO = 32 * WCO;           // co rows per workgroup
    static constexpr int TCI = 32 * NB * WCI;      // ci rows per workgroup
    static constexpr int NACC = KS * NB;
    static constexpr bool BH = BF == 2;
    static constexpr int RCO = BH ? TCO / 2 : TCO, RCI = BH ? TCI / 2 : TCI;   // LDS / source rows of the two operand tiles (pair rows with BH)
    static constexpr int WG_DYROW = wg_dyrow(LIN);
    static constexpr int NPD = (RCO * WG_DYROW + 255) / 256;  // dy pieces per producer wave
    static constexpr int NPX = (RCI * (LIN ? wg_xrow_lin(KS) : (KS == 1 ? 33 : 71)) + 255) / 256;  // x pieces per wave (XROW <= 71, or 33 for stride-1 1x1)
    static constexpr int TPR = 256 / RCO;  // producer threads per dy row in the bias-gradient partial sum
    static constexpr int CPT = 32 / TPR;
    static constexpr int NBROW = BH ? 16 : 32;   // LDS rows between the ci blocks of a wave
    static constexpr int GRP = wg_group(BF, NB);        // chunks per barrier (wg_group)
    static constexpr int NSTG = wg_stages(BF, NB);      // LDS stage slots per operand
    // ---- 16-BYTE staging of the bf16 whole-chunk instances (round 6; `wide` layers, wg_wide16 below).  The dword LDS-DMA that builds the
    // padded rows above costs one instruction per 256 bytes: 36 instructions per 9-KB chunk, and their issue / landing cadence -- not
    // latency, not HBM -- was the bf16 weight gradient's bound (0.83 us per chunk with the products ablated, 2.7 TB/s chip-wide:
    // profiles/r06_wgrad_steady_*.txt).  A dword row of a pair tensor is 16-byte aligned in HBM, so both operand tiles are staged as
    // rows of TEN 16-byte pieces = frames t0 - 4 .. t0 + 35 of the chunk (dy: t0 .. t0 + 31 and two pieces nobody reads): 1 KiB per
    // instruction, 10 per chunk.  40-dword rows: the 8 distinct rows a ds_read_b128 service group touches (lanes 2p, 2p + 1 share a pair
    // row) start 8 banks apart.  No reflection in the loader: the row's first / last piece of a sample's first / last chunk would lie
    // outside the row -- those lanes fetch the neighbouring piece instead (valid memory, never used) and the CONSUMER takes the mirrored
    // frames from registers it holds anyway (frame -i is frame i of the same fragment).
    static constexpr bool BHX = BH && LIN && !RT;   // (the run-time-taps instance of the bank was tried on these rows too -- fragments read at the dword-aligned LDS
                                                    // address c0 + 4 - padL, mirrored frames by a switch on the tap count: parity-green and SLOWER, wgrad class 1.06 vs
                                                    // 0.96 ms, step 2.52 vs 2.50 ms, gpurun_out r6n -- and taken out again)
    static constexpr int ROW16 = 40;
    static constexpr int ND16 = (RCO * 10 + 63) / 64, NX16 = (RCI * 10 + 63) / 64;   // DMA instructions per operand tile
    static constexpr int NPD16 = (ND16 + 3) / 4, NPX16 = (NX16 + 3) / 4;               // ... per producer wave
};
// may this layer's tiles be staged by 16-byte pieces?  Whole 32-frame chunks of a stride-1 layer (LIN instances) whose rows are 16-byte
// aligned dword rows.  Both roles of a workgroup evaluate it per segment.
static __device__ __forceinline__ bool wg_wide16(const WgradArgs& a) {
    return (a.Tout & 31) == 0 && a.Tin == a.Tout && a.spc == 1 && a.x.st == 1 && a.dy.st == 1 && a.x.ps == 1 && a.dy.ps == 1 &&
           ((a.x.sc | a.x.sb | a.dy.sc | a.dy.sb) & 3) == 0 && ((((unsigned long)a.x.ptr) | ((unsigned long)a.dy.ptr)) & 15) == 0;
}

// ---------------- producers: waves 4-7.  Both operand tiles go global -> LDS by dword DMA (each lane its own source address, so the
// reflect padding and the padded LDS rows cost no staging registers).  Two stages: chunk c+1 lands while chunk c multiplies.
template <int KS, bool RT, int NB, int WCO, bool LIN, int BF, int CW>
static __device__ __forceinline__ void wg_producer(const WgradBatch& bt, float* smem, int tid, int lane, int wave) {
    using C = WgCfg<KS, RT, NB, WCO, LIN, BF, CW>;
    constexpr int TCO = C::TCO, TCI = C::TCI, RCO = C::RCO, RCI = C::RCI, WG_DYROW = C::WG_DYROW, NPD = C::NPD, NPX = C::NPX, TPR = C::TPR, CPT = C::CPT;
    constexpr int NSTG = C::NSTG, GRP = C::GRP;   // group g + 1 (GRP chunks) is issued while group g multiplies
    constexpr bool BH = C::BH;
    const int dbg = bt.dbg;
    const int ptid = tid & 255;   // thread index inside the producer group (CW * 64 is a multiple of 256; the mask tells the compiler the range)
    unsigned dyo[NPD], xo[NPX];
    int xq[NPX];
    WgSegIter it(bt);
    WgSeg sg;
    while (it.next(bt, sg)) {
        const WgradArgs& a = bt.L[sg.layer];
        const int KSr = RT ? a.KS : KS;
        const int ci_tiles = avc_cdiv(a.Cin, TCI);
        const int CoutR = BH ? a.Cout >> 1 : a.Cout, CinR = BH ? a.Cin >> 1 : a.Cin;
        const float* xptr = a.x.ptr;
        const float* dyptr = a.dy.ptr;
        const int Tc = a.Tc, spc = a.spc;
        const int lgTc = 31 - __builtin_clz(Tc);
        const int XSEG = (Tc - 1) * a.stride + KSr;
        const int XROW = LIN ? wg_xrow_lin(KSr) : ((spc * XSEG) | 1);
        const bool wide = C::BHX && wg_wide16(a);   // 16-byte staging (WgCfg)
        const int DYROWW = wide ? C::ROW16 : WG_DYROW;
        const int DYS = wide ? RCO * C::ROW16 : RCO * WG_DYROW, XS = wide ? RCI * C::ROW16 : RCI * XROW;
        const int DYSP = wide ? (DYS + 255) & ~255 : (DYS + 63) & ~63, XSP = wide ? (XS + 255) & ~255 : (XS + 63) & ~63;  // stage strides: whole DMA pieces
        float* dyT = smem;                // [NSTG][RCO][WG_DYROW]
        float* xT = smem + NSTG * DYSP;   // [NSTG][RCI][XROW]
        const float inv_xrow = 1.0f / (float)XROW;
        const int tile = sg.tile, c_begin = sg.c_begin, c_end = sg.c_end;
        const int co0 = (tile / ci_tiles) * TCO, ci0 = (tile % ci_tiles) * TCI;
        const int co0r = BH ? co0 >> 1 : co0, ci0r = BH ? ci0 >> 1 : ci0;                   // ... in source rows
        const bool do_db = (a.db != nullptr) && (ci0 == 0);
        float dbsum = 0.f, dbsum1 = 0.f;
        // Fast path (one sample per chunk, whole chunks): every lane of every DMA piece always loads -- LDS positions that hold no
        // tile element (row padding, rows past Cout / Cin) get a clamped, valid address instead of an exec-masked branch; they are never
        // read, or feed accumulator rows that are never stored.  The chunk-invariant byte offsets live in producer registers, the chunk
        // origin is a scalar base: one SADDR-form DMA instruction per piece, ~no address VALU.
        // ... and the same for chunks that hold spc whole short samples (T_l = 16, 8, ...): there even the reflection is chunk-invariant,
        // so the x offsets are complete and only the base moves.
        const bool fastm = !wide && (spc > 1) && (a.Tout == Tc) && (a.B % spc == 0) && (XSP <= NPX * 256);
        const bool fastp = wide || fastm || ((spc == 1) && (a.Tout % 32 == 0) && (XSP <= NPX * 256));
        // 16-byte staging: chunk-invariant byte offsets of this wave's pieces; xedge = +1 / -1 for the first / last piece of an x row
        unsigned dyo16[C::BHX ? C::NPD16 : 1], xo16[C::BHX ? C::NPX16 : 1];
        int xedge[C::BHX ? C::NPX16 : 1];
        if constexpr (C::BHX) {
            if (wide) {
#pragma unroll
                for (int i = 0; i < C::NPD16; ++i) {
                    const int f = (wave + 4 * i) * 64 + lane;
                    int row = f / 10;
                    const int pc = f - row * 10;
                    row = row < RCO ? row : RCO - 1;
                    int co = co0r + row;
                    co = co < CoutR ? co : CoutR - 1;
                    dyo16[i] = 4u * (unsigned)((long)co * a.dy.sc + 4 * (pc < 8 ? pc : 0));   // (pieces 8, 9 of a dy row hold nothing: they re-fetch piece 0)
                }
#pragma unroll
                for (int i = 0; i < C::NPX16; ++i) {
                    const int f = (wave + 4 * i) * 64 + lane;
                    int row = f / 10;
                    const int pc = f - row * 10;
                    row = row < RCI ? row : RCI - 1;
                    int ci = ci0r + row;
                    ci = ci < CinR ? ci : CinR - 1;
                    xo16[i] = 4u * (unsigned)((long)ci * a.x.sc + 4 * pc);
                    xedge[i] = pc == 0 ? 1 : (pc == 9 ? -1 : 0);
                }
            }
        }
        auto issue16 = [&](int chunk, int buf) {
            if constexpr (C::BHX) {
                const int cb = chunk / a.chunks_per_sample;
                const int t0 = (chunk - cb * a.chunks_per_sample) * 32;
                const float* dyb = dyptr + ((long)cb * a.dy.sb + t0);
                const float* xb = xptr + ((long)cb * a.x.sb + t0) - 4;   // frame t0 - 4: in front of the row when t0 == 0 -- the lanes of that piece fetch the next one
                const bool first = t0 == 0, last = t0 + 32 == a.Tout;
                float* dd = dyT + buf * DYSP;
                float* xd = xT + buf * XSP;
#pragma unroll
                for (int i = 0; i < C::NPD16; ++i)
                    if (wave + 4 * i < C::ND16) avc_glds16_s(dyb, dyo16[i], dd + (wave + 4 * i) * 256);
#pragma unroll
                for (int i = 0; i < C::NPX16; ++i)
                    if (wave + 4 * i < C::NX16) {
                        unsigned v = xo16[i];
                        if (first && xedge[i] > 0) v += 16u;
                        if (last && xedge[i] < 0) v -= 16u;
                        avc_glds16_s(xb, v, xd + (wave + 4 * i) * 256);
                    }
            }
        };
        if (fastm) {
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
        } else if (fastp && !wide) {
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
            if (wide) {
                issue16(chunk, buf);
                return;
            }
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

        const int nck = c_end - c_begin;
        const int ngr = (nck + GRP - 1) / GRP;
        auto issue_group = [&](int gi) {   // the chunks of group gi into slots (gi & 1) GRP ...
#pragma unroll
            for (int i = 0; i < GRP; ++i) {
                const int chunk = c_begin + gi * GRP + i;
                if (chunk < c_end) issue(chunk, (gi & 1) * GRP + i);
            }
        };
        __syncthreads();  // the zero fill / the previous segment's last reads are complete before the first DMA of this one lands
        issue_group(0);
        __syncthreads();  // (drains the DMA: the compiler's barrier waits for vmcnt(0))
        for (int gi = 0; gi < ngr; ++gi) {
            const bool more = (gi + 1 < ngr) && !((dbg & 1) && gi > 0);
            if (more) issue_group(gi + 1);
            if (do_db) {   // bias gradient = row sums of the dy tiles that are in LDS anyway
#pragma unroll
                for (int i = 0; i < GRP; ++i) {
                    if (gi * GRP + i >= nck) break;
                    const float* dr = dyT + ((gi & 1) * GRP + i) * DYSP + (ptid / TPR) * DYROWW + (ptid % TPR) * CPT;
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
            }
            if (!(dbg & 4)) __syncthreads();
        }
        // ---- segment end
        auto store_db = [&](float v0, float v1) {   // finished bias gradient of this workgroup's co rows
            const int rows = a.rows_per_src;
            if ((ptid % TPR) != 0) return;
            if constexpr (BH) {
                const int co = co0 + 2 * (ptid / TPR);
                if (co < a.Cout) {
                    a.db[(long)(co / rows) * a.db_src_stride + co % rows] = v0;
                    a.db[(long)((co + 1) / rows) * a.db_src_stride + (co + 1) % rows] = v1;
                }
            } else {
                const int co = co0 + ptid / TPR;
                if (co < a.Cout) a.db[(long)(co / rows) * a.db_src_stride + co % rows] = v0;
            }
        };
        if (do_db) {
#pragma unroll
            for (int o = 1; o < TPR; o <<= 1) {
                dbsum += __shfl_xor(dbsum, o);
                if (BH) dbsum1 += __shfl_xor(dbsum1, o);
            }
        }
        if (dbg & 8) continue;   // (ablation: no stores, no reduce)
        if (sg.nsplit == 1) {    // this workgroup walked the tile's whole K range
            if (do_db) store_db(dbsum, dbsum1);
            continue;
        }
        float* dbs_t = do_db ? a.dbslab + ((long)(tile / ci_tiles) * a.slots) * TCO : nullptr;
        if (do_db && (ptid % TPR) == 0) {
            if constexpr (BH) {
                dbs_t[(long)sg.z * TCO + 2 * (ptid / TPR)] = dbsum;
                dbs_t[(long)sg.z * TCO + 2 * (ptid / TPR) + 1] = dbsum1;
            } else {
                dbs_t[(long)sg.z * TCO + ptid / TPR] = dbsum;
            }
        }
    }
}

// ---------------- consumers: waves 0-3 -- fragment reads, MFMAs, and at a segment's end the partial tile / the finished gradient
template <int KS, bool RT, int NB, int WCO, bool LIN, int BF, bool X3, int CW>
static __device__ __forceinline__ void wg_consumer(const WgradBatch& bt, float* smem, int tid, int lane, int wave) {
    using C = WgCfg<KS, RT, NB, WCO, LIN, BF, CW>;
    constexpr int WCI = C::WCI, TCO = C::TCO, TCI = C::TCI, NACC = C::NACC, RCO = C::RCO, RCI = C::RCI, WG_DYROW = C::WG_DYROW, NBROW = C::NBROW;
    constexpr int NSTG = C::NSTG;
    constexpr bool BH = C::BH;
    const int dbg = bt.dbg;
    const int wave_m = wave / WCI, wave_n = wave % WCI, li = lane & 31, h = lane >> 5;
    f32x16 acc[NACC];
    WgSegIter it(bt);
    WgSeg sg;
    while (it.next(bt, sg)) {
        const WgradArgs& a = bt.L[sg.layer];
        const int KSr = RT ? a.KS : KS;
        const int ci_tiles = avc_cdiv(a.Cin, TCI);
        const int Tc = a.Tc, spc = a.spc;
        const int lgTc = 31 - __builtin_clz(Tc);
        const int XSEG = (Tc - 1) * a.stride + KSr;
        const int XROW = LIN ? wg_xrow_lin(KSr) : ((spc * XSEG) | 1);
        const bool wide = C::BHX && wg_wide16(a);   // 16-byte staging: 40-dword rows, frame t0 - 4 + d at dword d (WgCfg)
        const int DYS = wide ? RCO * C::ROW16 : RCO * WG_DYROW, XS = wide ? RCI * C::ROW16 : RCI * XROW;
        const int DYSP = wide ? (DYS + 255) & ~255 : (DYS + 63) & ~63, XSP = wide ? (XS + 255) & ~255 : (XS + 63) & ~63;
        const float* dyT = smem;
        const float* xT = smem + NSTG * DYSP;
        int tcs = wide ? sg.c_begin % a.chunks_per_sample : 0;   // position of the chunk inside its sample (wide: first / last chunk reflect)
        const long tile_floats = (long)CW * KSr * NB * 1024;   // consumer waves x (taps x blocks) accumulators x 16 registers x 64 lanes
        const int tile = sg.tile, c_begin = sg.c_begin, c_end = sg.c_end;
        const int co0 = (tile / ci_tiles) * TCO, ci0 = (tile % ci_tiles) * TCI;
#pragma unroll
        for (int j = 0; j < NACC; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
        __syncthreads();
        __syncthreads();   // the first chunk has landed
        // The chunk loop exists ONCE PER STAGING LAYOUT (compile-time WIDE): with the layout as a run-time branch inside one loop body the two
        // bodies' accumulator chains met in phi nodes and the compiler copied all 80 accumulator registers around every MFMA
        // (v_mfma ... v[98:113], ..., v[66:81]: round 6, the consumers alone 764 instead of 418 ns per chunk).
        auto chunk_loop = [&](auto wtag) {
        constexpr bool WIDE = decltype(wtag)::value;
        constexpr int GRP = C::GRP;
        for (int chunk = c_begin; chunk < c_end; ++chunk) {
            const int rel = chunk - c_begin;
            const int buf = GRP == 1 ? (rel & 1) : ((rel / GRP) & 1) * GRP + rel % GRP;   // slot of the chunk: half (group & 1), place in the group
            const bool first_c = tcs == 0, last_c = tcs == a.chunks_per_sample - 1;
            if (WIDE && ++tcs == a.chunks_per_sample) tcs = 0;
            if (!(dbg & 2)) {
                const float* arow = dyT + buf * DYSP + (BH ? (wave_m * 32 + li) >> 1 : wave_m * 32 + li) * (WIDE ? C::ROW16 : WG_DYROW);
                const float* brow = xT + buf * XSP + (BH ? (wave_n * NB * 32 + li) >> 1 : wave_n * NB * 32 + li) * (WIDE ? C::ROW16 : XROW);
                // one straight-line chunk body.  RT: it is compiled for K = 8 taps and every tap's loads / MFMAs sit behind a WAVE-UNIFORM
                // test of the layer's own tap count (a scalar branch per MFMA group: nothing beside a 64-cycle MFMA).  (A switch over eight
                // per-tap-count bodies made the register allocator keep two copies of accumulators at the merge: spills in the loop.)
                auto body = [&](auto ktag) {
                    constexpr int K = decltype(ktag)::value;
                    auto tap_on = [&](int j) { return !RT || j < KSr; };
                    const float* arow_h = arow + h;
                    const float* brow_h = brow + h;
                    // fragments of k-step s+1 are requested before the MFMAs of step s are queued.  LIN: column 2s+h of the
                    // chunk is element 2s+h of both LDS rows, so every fragment address is base + immediate.
                    auto ldfrag = [&](int s, float& av, float (&bv)[NB * K]) {
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
                            for (int j = 0; j < K; ++j)
                                if (tap_on(j)) bv[nb * K + j] = bp[nb * 32 * XROW + j];
                    };
                    if constexpr (BH) {
                        // two blocks of 16 columns per chunk; lane-half h owns columns 16 kb + 8 h .. + 7 in BOTH operands
                        const unsigned sel = (li & 1) ? 0x07060302u : 0x05040100u;   // this lane's channel = low / high half of its pair row
                        if constexpr (WIDE) {
                            {
                                // 16-byte staged rows: dword d of a row = frame t0 - 4 + d.  The fragment of columns c0 = 16 kb + 8 h .. + 7 is dwords
                                // c0 .. c0 + 15 (frames c0 - 4 .. c0 + 11); tap j of column c0 + i reads frame c0 + i + j - PADL = fragment dword
                                // i + j + 4 - PADL.  At a sample's edges the mirrored frames are other dwords of the SAME fragment.
                                constexpr int PADL = K / 2, PADR = (K & 1) ? K / 2 : K / 2 - 1;
                                constexpr int P0 = (4 - PADL) / 4, P1 = (11 + PADR) / 4;   // 16-byte pieces of the fragment that are read
                                auto fetchw = [&](int kb, unsigned (&ad)[8], unsigned (&xr)[NB][16]) {
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
                                        for (int i4 = P0; i4 <= P1; ++i4) {
                                            const f32x4 v = *(const f32x4*)(bp + nb * NBROW * C::ROW16 + 4 * i4);
#pragma unroll
                                            for (int i = 0; i < 4; ++i) xr[nb][4 * i4 + i] = bh_as_u32(v[i]);
                                        }
                                };
                                auto blockw = [&](int kb, const unsigned (&ad)[8], unsigned (&xr)[NB][16]) {
                                    if (PADL > 0 && first_c && kb == 0) {   // frames -i of the sample: frame i (reflect padding, model.py:28-30)
#pragma unroll
                                        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                                            for (int i = 1; i <= PADL; ++i) xr[nb][4 - i] = h == 0 ? xr[nb][4 + i] : xr[nb][4 - i];
                                    }
                                    if (PADR > 0 && last_c && kb == 1) {    // frames T - 1 + i: frame T - 1 - i
#pragma unroll
                                        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                                            for (int i = 1; i <= PADR; ++i) xr[nb][11 + i] = h == 1 ? xr[nb][11 - i] : xr[nb][11 + i];
                                    }
                                    avc_u32x4 at;
#pragma unroll
                                    for (int q4 = 0; q4 < 4; ++q4) at[q4] = bh_sel(ad[2 * q4], ad[2 * q4 + 1], sel);
#pragma unroll
                                    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                                        for (int j = 0; j < K; ++j) {
                                            avc_u32x4 bq;
#pragma unroll
                                            for (int q4 = 0; q4 < 4; ++q4) bq[q4] = bh_sel(xr[nb][j + 4 - PADL + 2 * q4], xr[nb][j + 4 - PADL + 2 * q4 + 1], sel);
                                            acc[nb * KS + j] = avc_mfma_bf16x8(at, bq, acc[nb * KS + j]);
                                        }
                                };
                                unsigned a0[8], a1[8], x0[NB][16], x1[NB][16];
                                fetchw(0, a0, x0);
                                fetchw(1, a1, x1);   // (requested before the first block's MFMAs are issued)
                                blockw(0, a0, x0);
                                blockw(1, a1, x1);
                                return;
                            }
                        }
                        constexpr int NX = 8 + K - 1;
                        auto fetch = [&](int kb, unsigned (&ad)[8], unsigned (&xd)[NB][LIN ? NX : 8 * K]) {
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
                                            if (4 * i4 + i < NX) xd[nb][4 * i4 + i] = bh_as_u32(v[i]);   // (RT: values past the layer's own 8 + k - 1 feed skipped taps only)
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
                                        for (int j = 0; j < K; ++j)
                                            if (tap_on(j)) xd[nb][i * K + j] = bh_as_u32(bp[nb * NBROW * XROW + j]);
                                }
                            }
                        };
                        auto block = [&](const unsigned (&ad)[8], const unsigned (&xd)[NB][LIN ? NX : 8 * K]) {
                            avc_u32x4 at;
#pragma unroll
                            for (int q4 = 0; q4 < 4; ++q4) at[q4] = bh_sel(ad[2 * q4], ad[2 * q4 + 1], sel);
#pragma unroll
                            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                                for (int j = 0; j < K; ++j) {
                                    if (!tap_on(j)) continue;
                                    avc_u32x4 bq;
#pragma unroll
                                    for (int q4 = 0; q4 < 4; ++q4) {
                                        if constexpr (LIN) bq[q4] = bh_sel(xd[nb][j + 2 * q4], xd[nb][j + 2 * q4 + 1], sel);
                                        else bq[q4] = bh_sel(xd[nb][(2 * q4) * K + j], xd[nb][(2 * q4 + 1) * K + j], sel);
                                    }
                                    acc[nb * KS + j] = avc_mfma_bf16x8(at, bq, acc[nb * KS + j]);
                                }
                        };
                        unsigned a0[8], x0[NB][LIN ? NX : 8 * K], a1[8], x1[NB][LIN ? NX : 8 * K];
                        fetch(0, a0, x0);
                        fetch(1, a1, x1);   // (requested before the first block's MFMAs are issued)
                        block(a0, x0);
                        block(a1, x1);
                    } else if constexpr (X3 && LIN && !BF) {
                        // two blocks of 16 columns: lane-half h owns columns 16 kb + 8 h .. + 7 of the chunk
                        constexpr int NX = 8 + K - 1;   // x values under the K shifted windows of 8 columns
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
                                for (int j = 0; j < K; ++j) {
                                    if (!tap_on(j)) continue;
                                    avc_u32x4 q0, q1, q2;
#pragma unroll
                                    for (int q4 = 0; q4 < 4; ++q4) {
                                        q0[q4] = x3_pair(xh[j + 2 * q4], xh[j + 2 * q4 + 1]);
                                        q1[q4] = x3_pair(xm[j + 2 * q4], xm[j + 2 * q4 + 1]);
                                        q2[q4] = x3_pair(xl[j + 2 * q4], xl[j + 2 * q4 + 1]);
                                    }
                                    f32x16& c = acc[nb * KS + j];
                                    // small terms first
                                    c = avc_mfma_bf16x8(at[2], q0, c);
                                    c = avc_mfma_bf16x8(at[0], q2, c);
                                    c = avc_mfma_bf16x8(at[1], q1, c);
                                    c = avc_mfma_bf16x8(at[1], q0, c);
                                    c = avc_mfma_bf16x8(at[0], q1, c);
                                    c = avc_mfma_bf16x8(at[0], q0, c);
                                }
                            }
                        };
                        float a0[8], x0[NB][NX], a1[8], x1[NB][NX];
                        fetch(0, a0, x0);
                        fetch(1, a1, x1);   // (requested before the first block's MFMAs are issued)
                        block(a0, x0);
                        block(a1, x1);
                    } else if constexpr (LIN) {
                        // Four groups of 8 columns per chunk; in group g lane-half h owns columns 8 g + 4 h + u, u = k-step 0..3, in
                        // BOTH operands (the sum over columns does not care about the order).  Its four dy values are ONE
                        // ds_read_b128, and the 4 + K - 1 x values under its K shifted windows are (K + 6) / 4 more: 3 reads per 20
                        // MFMAs at k = 5.  BF: the same fragments rounded to bf16, one v_mfma_f32_32x32x8_bf16 per tap and group.
                        constexpr int NX4 = (K + 6) / 4;          // 16-byte reads covering 4 + K - 1 values
                        auto ldgrp = [&](int gq, f32x4& av, f32x4 (&xv)[NB][NX4]) {
                            const int c = 8 * gq + 4 * h;
                            av = *(const f32x4*)(arow + c);
#pragma unroll
                            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                                for (int i4 = 0; i4 < NX4; ++i4)
                                    if (!RT || 4 * i4 < 3 + KSr) xv[nb][i4] = *(const f32x4*)(brow + nb * 32 * XROW + c + 4 * i4);
                        };
                        f32x4 av[2], xv[2][NB][NX4];
                        ldgrp(0, av[0], xv[0]);
#pragma unroll
                        for (int gq = 0; gq < 4; ++gq) {
                            const int cur = gq & 1;
                            if (gq + 1 < 4) ldgrp(gq + 1, av[cur ^ 1], xv[cur ^ 1]);
                            __builtin_amdgcn_sched_barrier(0);  // the next group's reads stay in front of the MFMAs they overlap with ...
                            if constexpr (BF != 0) {
                                const avc_s16x4 ap = avc_pack_bf16x4(av[cur][0], av[cur][1], av[cur][2], av[cur][3]);
#pragma unroll
                                for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                                    for (int j = 0; j < K; ++j) {
                                        if (!tap_on(j)) continue;
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
                                        for (int j = 0; j < K; ++j)
                                            if (tap_on(j)) acc[nb * KS + j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cur][u], xv[cur][nb][(u + j) >> 2][(u + j) & 3], acc[nb * KS + j], 0, 0, 0);
                            }
                            __builtin_amdgcn_sched_barrier(0);  // ... and one group ahead only
                        }
                    } else if constexpr (BF != 0) {
                        // four k-steps (8 columns) per v_mfma_f32_32x32x8_bf16: slot j of lane-half h carries
                        // column 2(4g + j) + h of the chunk in both operands; operands are rounded to bf16 here
                        float av4[2][4], bv4[2][4][NB * K];
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
                            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                                for (int j = 0; j < K; ++j) {
                                    if (!tap_on(j)) continue;
                                    const int k = nb * K + j;
                                    acc[nb * KS + j] = avc_mfma_bf16(ap, avc_pack_bf16x4(bv4[cur][0][k], bv4[cur][1][k], bv4[cur][2][k], bv4[cur][3][k]), acc[nb * KS + j]);
                                }
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    } else {
                        float av[2], bv[2][NB * K];
                        ldfrag(0, av[0], bv[0]);
#pragma unroll
                        for (int s = 0; s < 16; ++s) {
                            const int cur = s & 1;
                            if (s + 1 < 16) ldfrag(s + 1, av[cur ^ 1], bv[cur ^ 1]);
                            __builtin_amdgcn_sched_barrier(0);  // reads stay in front of the MFMAs they overlap with ...
#pragma unroll
                            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                                for (int j = 0; j < K; ++j)
                                    if (tap_on(j)) acc[nb * KS + j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cur], bv[cur][nb * K + j], acc[nb * KS + j], 0, 0, 0);
                            __builtin_amdgcn_sched_barrier(0);  // ... and one step ahead only (hoisting all 16 steps' reads spills)
                        }
                    }
                };
                static_assert(!RT || KS == 8, "run-time tap counts are built for up to 8 taps");
                body(KTag<KS>{});
            }
            if (!(dbg & 4) && (GRP == 1 || rel % GRP == GRP - 1 || chunk + 1 == c_end)) __syncthreads();   // once per group
        }
        };
        if constexpr (C::BHX) {
            if (wide) chunk_loop(std::true_type{});
            else chunk_loop(std::false_type{});
        } else {
            chunk_loop(std::false_type{});
        }

        // ---------------- segment end
        // thread (wave, lane) owns, per ci block nb and tap j, the 16 accumulator registers of output rows
        // co = co0 + wave_m 32 + (r & 3) + 8 (r >> 2) + 4 h, column ci = ci0 + (wave_n NB + nb) 32 + li
        auto store_final = [&]() {
            const int rows = a.rows_per_src;
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                const int ci = ci0 + (wave_n * NB + nb) * 32 + li;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int co = co0 + wave_m * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    if (co < a.Cout && ci < a.Cin) {
                        const int src = co / rows, rr = co - src * rows;
                        float* d = a.dw + (long)src * a.dw_src_stride + ((long)rr * a.Cin + ci) * KSr;
#pragma unroll
                        for (int j = 0; j < KS; ++j)
                            if (j < KSr) d[j] = acc[nb * KS + j][r];
                    }
                }
            }
        };
        if (dbg & 8) continue;   // (ablation: no stores, no reduce)
        if (sg.nsplit == 1) {    // this workgroup walked the tile's whole K range: the accumulators ARE the gradient
            store_final();
            continue;
        }
        // partial tile -> slot z of the tile, accumulator layout [consumer wave][accumulator][register quad][lane][4]: every store
        // instruction of a wave is 1 KiB contiguous, and thread (wave, lane) of the reduce launch reads exactly what it will own
        float* slab_t = a.slab + ((long)tile * a.slots) * tile_floats + (long)wave * (KSr * NB * 1024) + lane * 4;
        {
            float* sp = slab_t + (long)sg.z * tile_floats;   // (a running pointer: one address register pair, 1 KiB steps)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int j = 0; j < KS; ++j)
                    if (j < KSr) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            f32x4 v = {acc[nb * KS + j][4 * q], acc[nb * KS + j][4 * q + 1], acc[nb * KS + j][4 * q + 2], acc[nb * KS + j][4 * q + 3]};
                            *(f32x4*)sp = v;
                            sp += 256;
                        }
                    }
        }
    }
}

// Register budget = co-residency: a weight-gradient workgroup sits on its CU for the whole launch (hundreds of microseconds, on a
// low-priority stream) while the latency-critical dgrad / InstanceNorm chain of the same backward pass needs slots on the same CUs.
// At 2 waves per SIMD x ~200 registers (what the compiler takes when nothing stops it) a CU has no room left for a second kernel's
// waves and the chain's small launches queue behind the persistent workgroups (traced in round 4: a 25-us dgrad launch took 440 us).
// The second __launch_bounds__ argument (minimum waves per SIMD) caps the allocation: 128 registers for the k = 5 whole-chunk
// instance (no spill), 168 where that costs no spill in the loop; the bank instance (128 accumulator registers) keeps 256.
template <int KS, bool RT, int NB, int WCO, bool LIN, int BF, bool X3, int CW>
struct WgWaves {
    static constexpr int value = CW == 8 ? 3   // 12 waves = 3 per SIMD: 168 registers (at 128: class 2.31 vs 2.21 ms, spills only in the cold store path)
                                 : (BF == 2 && LIN && !RT && KS == 5) ? AVC_WGRAD_BHX_WAVES   // (137 registers with the chunk loop instantiated per staging layout)
                                 : (KS == 5 && !RT && LIN && !X3 && BF != 2) ? 4
                                 : (((KS == 5 && (LIN || BF == 0)) || (KS == 1 && NB == 1)) ? 3 : 2);   // (run-time-taps bank instance: 256 registers -- at 168 it
                                                                                             // spills in the loop; class and step equal within noise)
};
// CW = 8 (opt-in, avc_tuning.wgrad_cw8): eight consumer waves (two per SIMD) on a 128 co x 64 ci tile + the four producers.  Two MFMA
// waves per SIMD fill each other's gaps (fragment-read latency behind every barrier), and the tile's DMA bytes per MFMA are 3/4 of the
// 64 x 64 tile's: the class drops 2.26 -> 2.21 ms per step, but the STEP rises 5.98 -> 6.03 ms (same box, round 4) -- a 768-thread
// workgroup at 168 registers owns its CU, and the chain's small launches wait for one to leave.  Hence off by default.
// NOTE the role-local indices below are MASKS (tid & 255, wave & 3), not differences: with `tid - CW * 64` the compiler loses the
// value range and every instance's producer address arithmetic got 5-25 % slower (measured per instance, round 4).
template <int KS, bool RT, int NB, int WCO, bool LIN, int BF, bool X3 = false, int CW = 4>
__global__ void __launch_bounds__(wg_threads(CW), (WgWaves<KS, RT, NB, WCO, LIN, BF, X3, CW>::value)) conv_wgrad_kernel(const WgradBatch bt) {
    HIP_DYNAMIC_SHARED(float, smem)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);  // scalar: LDS-DMA bases must be provably wave-uniform
    // both stages start as zeros (the general staging path stores explicit zeros afterwards, the fast paths overwrite every position
    // they read; positions nobody reads may hold an earlier layer's finite data)
    for (int e = tid; e < bt.lds_floats; e += wg_threads(CW)) smem[e] = 0.f;
    if (wave8 >= CW) wg_producer<KS, RT, NB, WCO, LIN, BF, CW>(bt, smem, tid, lane, wave8 & 3);
    else wg_consumer<KS, RT, NB, WCO, LIN, BF, X3, CW>(bt, smem, tid, lane, wave8);
}

// --------------------------------------------------------------------------
// tile shape per layer: returns (NB, WCO); tile = (32*WCO) co x (32*NB*(4/WCO)) ci
static void wgrad_shape(const WgradArgs& a, int* NB, int* WCO, int* CW) {
    *CW = 4;
    if (a.KS == 1 && a.Cin >= 256) {
        *NB = 4; *WCO = 4;      // 128 x 128 (the 1x1 over the 1104-channel concat buffer; a 128-channel 1x1 has too few such tiles: each
                                // would be shared by ~100 workgroups and its reduce would be one long dependent chain)
    } else if (a.KS == 5 && a.Cin % 64 == 0 && a.Cout % 128 == 0 && a.cw8 && a.Tc == 32 && a.stride == 1 && a.bf16 != AVC_COMPUTE_BF16S &&
               a.bf16 != AVC_COMPUTE_F32X3) {
        *NB = 1; *WCO = 4; *CW = 8;   // 128 co x 64 ci, eight consumer waves (the model's k = 5 layers)
    } else if (a.Cin % 64 == 0) {
        *NB = 1; *WCO = 2;      // 64 x 64
    } else {
        *NB = 1; *WCO = 4;      // 128 x 32: Cin = 80 wastes 17 % instead of 38 %
    }
}

// kernel instance a layer runs on; layers with equal keys share a launch.  KST: the tap count the instance is compiled for -- 1 and 5
// (the model's own sizes) have their own instances, every other case runs on the run-time-taps instance (KST = 8)
struct WgradKey {
    int KST, rt, NB, WCO, CW, lin;
    bool operator==(const WgradKey& o) const { return KST == o.KST && rt == o.rt && NB == o.NB && WCO == o.WCO && CW == o.CW && lin == o.lin; }
};
static size_t wgrad_lds_bytes_for(const WgradArgs& a, int NB, int WCO, int CW) {
    const int half = a.bf16 == AVC_COMPUTE_BF16S ? 2 : 1;   // pair rows
    const int TCO = 32 * WCO / half, TCI = 32 * NB * (CW / WCO) / half;
    const int XSEG = (a.Tc - 1) * a.stride + a.KS;
    const bool lin = a.Tc == 32 && a.stride == 1;   // (the LIN kernel instances, wgrad_key)
    const int XROW = lin ? wg_xrow_lin(a.KS) : ((a.spc * XSEG) | 1), WG_DYROW = wg_dyrow(lin);
    size_t stage = (size_t)((TCO * WG_DYROW + 63) & ~63) + ((TCI * XROW + 63) & ~63);
    if (half == 2 && lin) {   // the 16-byte staging of the bf16 whole-chunk instances: 40-dword rows, whole 1-KiB pieces per operand
        const size_t s16 = (size_t)((TCO * 40 + 255) & ~255) + ((TCI * 40 + 255) & ~255);
        stage = s16 > stage ? s16 : stage;
    }
    return (size_t)wg_stages(half == 2 ? 2 : 0, NB) * stage * 4 + 64;
}
static WgradKey wgrad_key(const WgradArgs& a) {
    WgradKey k;
    wgrad_shape(a, &k.NB, &k.WCO, &k.CW);
    if (a.KS == 1 && k.NB == 4 && wgrad_lds_bytes_for(a, 4, 4, 4) > 158 * 1024) {  // LDS too small for the wide tile (many short samples)
        k.NB = 1;
        k.WCO = 2;
    }
    if (k.CW == 8 && wgrad_lds_bytes_for(a, k.NB, k.WCO, 8) > 158 * 1024) {
        k.CW = 4;
        k.WCO = 2;
    }
    if (a.KS == 1) { k.KST = 1; k.rt = 0; }
    else if (a.KS == 5 && (k.WCO == 2 || k.CW == 8)) { k.KST = 5; k.rt = 0; }
    else { k.KST = 8; k.rt = 1; }
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
    a.tiles = avc_cdiv(a.Cout, 32 * k.WCO) * avc_cdiv(a.Cin, 32 * k.NB * (k.CW / k.WCO));
}

// Plans a batch: layers that share a kernel instance (and an operand dtype) form ONE launch (<= AVC_WGRAD_MAXL layers), a stream-K
// split over `target_wgs` workgroups -- fewer when the launch is small: a workgroup walks at least 4 chunks, and never less than one
// chunk of the most expensive layer (so that every workgroup between a tile's first and last owner owns a chunk of it).
// Fills grp / grid / chunk_cost / cost_begin / cost_total / slots / slab_need / dbslab_need / tile shape of every layer.
void avc_wgrad_plan_batch(WgradArgs* L, int n, int target_wgs) {
    if (target_wgs < 1) target_wgs = 256;
    for (int i = 0; i < n; ++i) {
        avc_wgrad_geometry(L[i]);
        L[i].grp = -1;
    }
    int ngrp = 0;
    for (int i = 0; i < n; ++i) {
        if (L[i].grp >= 0) continue;
        const WgradKey k = wgrad_key(L[i]);
        std::vector<int> mem;
        for (int j = i; j < n && (int)mem.size() < AVC_WGRAD_MAXL; ++j)
            if (L[j].grp < 0 && wgrad_key(L[j]) == k && L[j].bf16 == L[i].bf16) mem.push_back(j);
        long C = 0;
        int ccmax = 1;
        for (int j : mem) {
            WgradArgs& a = L[j];
            a.grp = ngrp;
            // cost of one chunk: its MFMAs (taps x ci blocks) + a fixed part (staging, barrier) worth about one tap
            a.chunk_cost = a.KS * k.NB + 1;
            ccmax = a.chunk_cost > ccmax ? a.chunk_cost : ccmax;
            a.cost_begin = C;
            C += (long)a.tiles * a.total_chunks * a.chunk_cost;
        }
        long grid = C / ((long)4 * ccmax);
        if (grid > target_wgs) grid = target_wgs;
        if (grid < 1) grid = 1;
        const int TCOv = 32 * k.WCO;
        // no tile is shared by more than 64 workgroups (its reduce is a dependent chain of slot reads): a launch of very few tiles gets
        // fewer workgroups instead
        // (slots are ALWAYS those of the final grid: the loop ends on a pass that did not shrink it -- a stale count would size the slabs
        // for another split than the one the kernel and the reduce derive from `grid`)
        for (;;) {
            int worst = 1;
            for (int j : mem) {
                WgradArgs& a = L[j];
                int slots = 1;
                for (int t = 0; t < a.tiles; ++t) {
                    const long cs = a.cost_begin + (long)t * a.total_chunks * a.chunk_cost;
                    const long wf = (cs * grid) / C, wl = ((cs + (long)(a.total_chunks - 1) * a.chunk_cost) * grid) / C;
                    slots = (int)(wl - wf + 1) > slots ? (int)(wl - wf + 1) : slots;
                }
                a.slots = slots;
                worst = slots > worst ? slots : worst;
            }
            if (worst <= 64 || grid <= 1) break;
            const long shrunk = grid * 62 / worst;
            grid = shrunk < grid ? (shrunk < 1 ? 1 : shrunk) : grid - 1;   // strictly smaller every pass: terminates
        }
        for (int j : mem) {
            WgradArgs& a = L[j];
            a.grid = (int)grid;
            a.cost_total = C;
            a.tNB = k.NB;
            a.tWCO = k.WCO;
            a.tCW = k.CW;
            const long tile_floats = (long)k.CW * a.KS * k.NB * 1024;
            a.slab_need = a.slots > 1 ? (long)a.tiles * a.slots * tile_floats : 0;
            a.dbslab_need = a.slots > 1 ? (long)avc_cdiv(a.Cout, TCOv) * a.slots * TCOv : 0;
        }
        ++ngrp;
    }
}

template <int KS, bool RT, int NB, int WCO>
static int launch_wgrad_t(const WgradBatch& bt, int grid_wgs, bool lin, int bf, bool x3, size_t lds, double flops, hipStream_t stream);

// the eight-consumer-wave instances (fp32 / bf16-operand products of the k = 5 layers)
static int launch_wgrad_cw8(const WgradBatch& bt, int grid_wgs, bool lin, int bf, size_t lds, double flops, hipStream_t stream) {
    if (lds > 158 * 1024) return -3;
    dim3 grid(grid_wgs), block(wg_threads(8));
    ProfScope ps(AVC_K_CONV_WGRAD, flops, 0.0, stream);
    if (!lin) return -1;   // (wgrad_shape: whole-chunk stride-1 layers only)
    if (bf) hipLaunchKernelGGL((conv_wgrad_kernel<5, false, 1, 4, true, 1, false, 8>), grid, block, lds, stream, bt);
    else hipLaunchKernelGGL((conv_wgrad_kernel<5, false, 1, 4, true, 0, false, 8>), grid, block, lds, stream, bt);
    return (int)hipGetLastError();
}

template <int KS, bool RT, int NB, int WCO>
static int launch_wgrad_t(const WgradBatch& bt, int grid_wgs, bool lin, int bf, bool x3, size_t lds, double flops, hipStream_t stream) {
    if (lds > 158 * 1024) return -3;
    dim3 grid(grid_wgs);
    ProfScope ps(AVC_K_CONV_WGRAD, flops, 0.0, stream);
    if (bf == 2) {
        if (lin) hipLaunchKernelGGL((conv_wgrad_kernel<KS, RT, NB, WCO, true, 2>), grid, dim3(WG_THREADS), lds, stream, bt);
        else hipLaunchKernelGGL((conv_wgrad_kernel<KS, RT, NB, WCO, false, 2>), grid, dim3(WG_THREADS), lds, stream, bt);
    } else if (bf) {
        if (lin) hipLaunchKernelGGL((conv_wgrad_kernel<KS, RT, NB, WCO, true, 1>), grid, dim3(WG_THREADS), lds, stream, bt);
        else hipLaunchKernelGGL((conv_wgrad_kernel<KS, RT, NB, WCO, false, 1>), grid, dim3(WG_THREADS), lds, stream, bt);
    } else if (lin) {
        if constexpr (KS * NB <= 8) {
            if (x3) {
                hipLaunchKernelGGL((conv_wgrad_kernel<KS, RT, NB, WCO, true, 0, true>), grid, dim3(WG_THREADS), lds, stream, bt);
                return (int)hipGetLastError();
            }
        }
        hipLaunchKernelGGL((conv_wgrad_kernel<KS, RT, NB, WCO, true, 0>), grid, dim3(WG_THREADS), lds, stream, bt);
    } else hipLaunchKernelGGL((conv_wgrad_kernel<KS, RT, NB, WCO, false, 0>), grid, dim3(WG_THREADS), lds, stream, bt);
    return (int)hipGetLastError();
}

// ---------------- the reduce launch of a batch: every tile that more than one workgroup worked on.  One 256-thread block per (tile,
// accumulator, consumer wave) sums that wave's registers over the tile's slots in a fixed order and stores the gradient in the
// parameter layout; one more block per co tile does the bias rows.
struct WgReduceItem {
    const float* slab;
    const float* dbslab;
    float* dw;
    float* db;
    long dw_src_stride, db_src_stride, cost_begin, cost_total;
    int tiles, slots, total_chunks, chunk_cost, grid;
    int KS, NB, WCO, CW, Cin, Cout, rows_per_src;
    int blk_begin, nblk_w;   // first block of this layer in the launch, its weight blocks (tiles x taps x ci blocks x 4 consumer waves)
};
struct WgReduceArgs {
    int n, pad_;
    WgReduceItem it[AVC_WGRAD_MAXL];
};
static __device__ __forceinline__ int wg_tile_nsplit(const WgReduceItem& a, int tile) {
    const long cs = a.cost_begin + (long)tile * a.total_chunks * a.chunk_cost;
    const long G = a.grid;
    return (int)(((cs + (long)(a.total_chunks - 1) * a.chunk_cost) * G) / a.cost_total) - (int)((cs * G) / a.cost_total) + 1;
}
__global__ void __launch_bounds__(256) wgrad_reduce_kernel(const WgReduceArgs ra) {
    int li_ = 0;
    for (int i = 1; i < ra.n; ++i) li_ = ((int)blockIdx.x >= ra.it[i].blk_begin) ? i : li_;
    const WgReduceItem& a = ra.it[li_];
    const int blk = (int)blockIdx.x - a.blk_begin;
    const int tid = threadIdx.x;
    const int WCI = a.CW / a.WCO, TCO = 32 * a.WCO, TCI = 32 * a.NB * WCI;
    const int ci_tiles = avc_cdiv(a.Cin, TCI);
    if (blk >= a.nblk_w) {   // bias rows of one co tile
        const int cot = blk - a.nblk_w;
        const int nsplit = wg_tile_nsplit(a, cot * ci_tiles);
        if (nsplit == 1 || tid >= TCO) return;
        const float* p = a.dbslab + ((long)cot * a.slots) * TCO + tid;
        float v = 0.f;
        for (int z = 0; z < nsplit; ++z) v += p[(long)z * TCO];
        const int co = cot * TCO + tid;
        if (co < a.Cout) a.db[(long)(co / a.rows_per_src) * a.db_src_stride + co % a.rows_per_src] = v;
        return;
    }
    // weight block = (tile, accumulator, consumer wave): its 64 lanes' 16 registers each.  The block's four waves each sum a QUARTER of the
    // tile's slots (contiguous z ranges, ascending), the quarters are combined through LDS as ((q0 + q1) + q2) + q3: a fixed order,
    // and a tile that 200 workgroups share (a lone 1x1 layer's only tile) is four times fewer dependent round trips
    const int nacc = a.KS * a.NB;
    const int ta = blk / a.CW, cw = blk - ta * a.CW;           // consumer wave whose registers these are
    const int tile = ta / nacc, ai = ta - tile * nacc;
    const int nsplit = wg_tile_nsplit(a, tile);
    if (nsplit == 1) return;   // the workgroup that walked the whole K range wrote the gradient itself
    const int zq = tid >> 6, lane = tid & 63, li = lane & 31, h = lane >> 5;
    const int wave_m = cw / WCI, wave_n = cw % WCI;
    const int nb = ai / a.KS, j = ai - nb * a.KS;
    const long tile_floats = (long)a.CW * nacc * 1024;
    const float* p = a.slab + ((long)tile * a.slots) * tile_floats + (long)cw * (nacc * 1024) + (long)ai * 1024 + lane * 4;
    const int per = (nsplit + 3) >> 2;
    const int z_lo = zq * per, z_hi = (z_lo + per < nsplit) ? z_lo + per : nsplit;
    f32x4 s[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) s[q] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int z0 = z_lo; z0 < z_hi; z0 += 4) {   // four slots' loads (16 x 16 bytes per thread) in flight at a time
        f32x4 v[4][4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int z = z0 + k < z_hi ? z0 + k : z_hi - 1;
#pragma unroll
            for (int q = 0; q < 4; ++q) v[k][q] = *(const f32x4*)(p + (long)z * tile_floats + q * 256);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (z0 + k < z_hi) {
#pragma unroll
                for (int q = 0; q < 4; ++q) s[q] += v[k][q];
            }
    }
    __shared__ f32x4 part[3][4][64];
    if (zq > 0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) part[zq - 1][q][lane] = s[q];
    }
    __syncthreads();
    if (zq > 0) return;
#pragma unroll
    for (int k = 0; k < 3; ++k)
        if ((k + 1) * per < nsplit) {   // (an empty quarter holds zeros: skipping it or adding it is the same sum)
#pragma unroll
            for (int q = 0; q < 4; ++q) s[q] += part[k][q][lane];
        }
    const int co0 = (tile / ci_tiles) * TCO, ci0 = (tile % ci_tiles) * TCI;
    const int ci = ci0 + (wave_n * a.NB + nb) * 32 + li;
    if (ci >= a.Cin) return;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int co = co0 + wave_m * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (co < a.Cout) {
            const int src = co / a.rows_per_src, rr = co - src * a.rows_per_src;
            a.dw[(long)src * a.dw_src_stride + ((long)rr * a.Cin + ci) * a.KS + j] = s[r >> 2][r & 3];
        }
    }
}

int avc_launch_wgrad_reduce(const WgradArgs* L, int n, hipStream_t stream) {
    for (int i0 = 0; i0 < n;) {
        WgReduceArgs ra;
        memset(&ra, 0, sizeof(ra));
        int blocks = 0;
        double bytes = 0;
        int i = i0;
        for (; i < n && ra.n < AVC_WGRAD_MAXL; ++i) {
            const WgradArgs& a = L[i];
            if (a.slots <= 1) continue;   // every tile of this layer had one owner
            WgReduceItem& t = ra.it[ra.n++];
            t.slab = a.slab; t.dbslab = a.dbslab; t.dw = a.dw; t.db = a.dbslab ? a.db : nullptr;
            t.dw_src_stride = a.dw_src_stride; t.db_src_stride = a.db_src_stride;
            t.cost_begin = a.cost_begin; t.cost_total = a.cost_total;
            t.tiles = a.tiles; t.slots = a.slots; t.total_chunks = a.total_chunks; t.chunk_cost = a.chunk_cost; t.grid = a.grid;
            t.KS = a.KS; t.NB = a.tNB; t.WCO = a.tWCO; t.CW = a.tCW; t.Cin = a.Cin; t.Cout = a.Cout; t.rows_per_src = a.rows_per_src;
            t.blk_begin = blocks;
            t.nblk_w = a.tiles * a.KS * a.tNB * a.tCW;
            blocks += t.nblk_w + (t.db ? avc_cdiv(a.Cout, 32 * a.tWCO) : 0);
            bytes += 4.0 * ((double)a.slab_need + (double)a.Cout * a.Cin * a.KS);
        }
        i0 = i;
        if (ra.n == 0) continue;
        ProfScope ps(AVC_K_REDUCE, 0.0, bytes, stream);
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(blocks), dim3(256), 0, stream, ra);
        int rc = (int)hipGetLastError();
        if (rc) return rc;
    }
    return 0;
}

// launches every layer of the batch (planned by avc_wgrad_plan_batch; slab / dbslab / dw / db assigned by the caller): one launch per
// group, then the batch's reduce launch.
// ablation: timing-experiment bits of scripts/wgrad_ablate.py (results are wrong by construction when set)
int avc_launch_wgrad_batch(const WgradArgs* L, int n, hipStream_t stream, int ablation) {
    int ngrp = 0;
    for (int i = 0; i < n; ++i) ngrp = L[i].grp + 1 > ngrp ? L[i].grp + 1 : ngrp;
    for (int grp = 0; grp < ngrp; ++grp) {
        WgradBatch bt;
        memset(&bt, 0, sizeof(bt));
        bt.dbg = ablation;
        size_t lds = 0;
        double flops = 0;
        WgradKey k;
        memset(&k, 0, sizeof(k));
        for (int j = 0; j < n; ++j) {
            if (L[j].grp != grp) continue;
            const WgradArgs& a0 = L[j];
            if (a0.KS < 1 || a0.KS > 8) return -1;
            if (a0.padL >= a0.Tin) return -6;
            if (bt.nlayers == 0) k = wgrad_key(a0);
            if (bt.nlayers >= AVC_WGRAD_MAXL) return -1;
            if (a0.slots > 1 && (!a0.slab || (a0.db && !a0.dbslab))) return -1;
            bt.L[bt.nlayers++] = a0;
            size_t l = wgrad_lds_bytes_for(a0, k.NB, k.WCO, k.CW);
            lds = l > lds ? l : lds;
            flops += 2.0 * a0.Cout * a0.Cin * a0.KS * (double)a0.B * a0.Tout;
        }
        if (bt.nlayers == 0) continue;
        const WgradArgs& a0 = bt.L[0];
        const int bf = a0.bf16 == AVC_COMPUTE_BF16 ? 1 : (a0.bf16 == AVC_COMPUTE_BF16S ? 2 : 0);
        const bool x3 = a0.bf16 == AVC_COMPUTE_F32X3;
        if (bf == 2)
            for (int j = 0; j < bt.nlayers; ++j)
                if ((bt.L[j].Cin & 1) || (bt.L[j].Cout & 1) || bt.L[j].x.st != 1 || bt.L[j].dy.st != 1 || bt.L[j].x.ps != 1 || bt.L[j].dy.ps != 1) return -2;
        bt.lds_floats = (int)(lds / 4);
        int rc;
        if (k.KST == 1) {
            if (k.NB == 4) rc = launch_wgrad_t<1, false, 4, 4>(bt, a0.grid, k.lin, bf, x3, lds, flops, stream);
            else if (k.WCO == 4) rc = launch_wgrad_t<1, false, 1, 4>(bt, a0.grid, k.lin, bf, x3, lds, flops, stream);
            else rc = launch_wgrad_t<1, false, 1, 2>(bt, a0.grid, k.lin, bf, x3, lds, flops, stream);
        } else if (k.KST == 5 && k.CW == 8) {
            rc = launch_wgrad_cw8(bt, a0.grid, k.lin, bf, lds, flops, stream);
        } else if (k.KST == 5) {
            rc = launch_wgrad_t<5, false, 1, 2>(bt, a0.grid, k.lin, bf, x3, lds, flops, stream);
        } else {
            if (k.WCO == 4) rc = launch_wgrad_t<8, true, 1, 4>(bt, a0.grid, k.lin, bf, x3, lds, flops, stream);
            else rc = launch_wgrad_t<8, true, 1, 2>(bt, a0.grid, k.lin, bf, x3, lds, flops, stream);
        }
        if (rc) return rc;
    }
    if (ablation & 8) return 0;   // (timing experiment: no slab stores -> nothing to reduce)
    return avc_launch_wgrad_reduce(L, n, stream);
}
