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

#include "avc_common.h"
#include "avc_internal.h"

#define WG_DYROW 33

static inline __device__ long src_chan_off(const ConvSrc& s, int c) {
    return (s.ps == 1) ? (long)c * s.sc : (long)(c / s.ps) * s.sc + (c % s.ps);
}

// KS taps, NB 32-wide ci blocks per wave, WCO waves along co (4/WCO along ci):
//   workgroup tile = (32*WCO) co  x  (32*NB*(4/WCO)) ci  x  KS taps.
// <5,1,2> is the 64x64 tile of the k=5 layers; WCO=4 (128co x 32ci) suits Cin that is not a multiple
// of 64 (the 80-mel bank convs); <1,4,4> (128co x 128ci) gives the 1x1 convs / Linears four
// accumulators per wave, i.e. the arithmetic intensity per staged element that the taps give k=5.
template <int KS, int NB, int WCO>
__global__ void __launch_bounds__(AVC_THREADS) conv_wgrad_kernel(const WgradArgs a) {
    constexpr int WCI = 4 / WCO;            // waves along ci
    constexpr int TCO = 32 * WCO;           // co rows per workgroup
    constexpr int TCI = 32 * NB * WCI;      // ci rows per workgroup
    constexpr int NACC = KS * NB;
    HIP_DYNAMIC_SHARED(float, smem)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // scalar: LDS-DMA bases must be provably wave-uniform
    const int wave_m = wave / WCI, wave_n = wave % WCI, li = lane & 31, h = lane >> 5;
    const int ci_tiles = avc_cdiv(a.Cin, TCI);
    const int co0 = (blockIdx.x / ci_tiles) * TCO, ci0 = (blockIdx.x % ci_tiles) * TCI;
    const int z = blockIdx.y;
    const int Tc = a.Tc, spc = a.spc;
    const int lgTc = 31 - __builtin_clz(Tc);
    const int XSEG = (Tc - 1) * a.stride + KS;
    const int XROW = (spc * XSEG) | 1;  // odd row stride: conflict-free column reads
    const int DYS = TCO * WG_DYROW, XS = TCI * XROW;
    float* dyT = smem;            // [2][TCO][WG_DYROW]
    float* xT = smem + 2 * DYS;   // [2][TCI][XROW]
    const bool do_db = (a.dbslab != nullptr) && (ci0 == 0);
    const float inv_xrow = 1.0f / (float)XROW;

    f32x16 acc[NACC];
#pragma unroll
    for (int j = 0; j < NACC; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    float dbsum = 0.f;

    // Both operand tiles go global -> LDS by dword DMA (each lane its own source address, so the
    // reflect padding and the padded LDS rows cost no staging registers).  Two stages: chunk c+1
    // lands while chunk c multiplies.  Both stages are zero-filled once; afterwards only elements
    // that exist are ever written, dy columns past the end of a sample are re-zeroed, and x
    // elements that no valid dy column multiplies may keep stale (finite) data.
    for (int e = tid; e < 2 * (DYS + XS); e += AVC_THREADS) smem[e] = 0.f;

    // chunk-invariant part of every lane's DMA descriptors (fast path: one sample per chunk)
    constexpr int NPD = (TCO * WG_DYROW + 255) / 256;  // dy pieces per wave
    constexpr int NPX = (TCI * (KS == 1 ? 33 : 71) + 255) / 256;  // x pieces per wave (XROW <= 71, or 33 for stride-1 1x1)
    const bool fastp = (spc == 1) && (XS <= NPX * 256);
    int dyo[NPD], xo[NPX], xq[NPX];
    if (fastp) {
#pragma unroll
        for (int i = 0; i < NPD; ++i) {
            int f = (wave + 4 * i) * 64 + lane;
            int row = f / WG_DYROW, qcol = f - row * WG_DYROW;
            bool ok = f < DYS && qcol < 32 && (co0 + row) < a.Cout;
            dyo[i] = ok ? (int)(src_chan_off(a.dy, co0 + row) + (long)qcol * a.dy.st) : -1;
        }
#pragma unroll
        for (int i = 0; i < NPX; ++i) {
            int f = (wave + 4 * i) * 64 + lane;
            int row = avc_fastdiv(f, XROW, inv_xrow), p = f - row * XROW;
            bool ok = f < XS && p < XSEG && (ci0 + row) < a.Cin;
            xo[i] = ok ? (int)src_chan_off(a.x, ci0 + row) : -1;
            xq[i] = p;
        }
    }

    // fast-path pieces (statically indexed descriptors): issued one by one so that the main loop can
    // place each LDS-DMA right behind a group of queued MFMAs
    struct FastOrigin {
        const float* dyb;
        const float* xb;
        int t0, v0;
    };
    auto fast_origin = [&](int chunk) {
        FastOrigin o;
        const int cb = chunk / a.chunks_per_sample;
        o.t0 = (chunk - cb * a.chunks_per_sample) * 32;
        o.dyb = a.dy.ptr + ((long)cb * a.dy.sb + (long)o.t0 * a.dy.st);
        o.xb = a.x.ptr + (long)cb * a.x.sb;
        o.v0 = o.t0 * a.stride - a.padL;
        return o;
    };
    auto fast_dy_piece = [&](const FastOrigin& o, int i, int dyoi, float* dd) {
        const int piece = wave + 4 * i;
        if (piece * 64 < DYS && dyoi >= 0) {
            const int f = piece * 64 + lane;
            const int qcol = f - (f / WG_DYROW) * WG_DYROW;
            if (o.t0 + qcol < a.Tout)
                avc_glds4(o.dyb + dyoi, dd + piece * 64);
            else
                dd[f] = 0.f;
        }
    };
    auto fast_x_piece = [&](const FastOrigin& o, int i, int xoi, int xqi, float* xd) {
        const int piece = wave + 4 * i;
        if (piece * 64 < XS && xoi >= 0) {
            int r = avc_reflect(o.v0 + xqi, a.Tin);
            if (r >= 0 && r < a.Tin) avc_glds4(o.xb + ((long)xoi + (long)r * a.x.st), xd + piece * 64);
        }
    };

    auto issue = [&](int chunk, int buf) {
        float* dd = dyT + buf * DYS;
        float* xd = xT + buf * XS;
        if (fastp) {
            const FastOrigin o = fast_origin(chunk);
#pragma unroll
            for (int i = 0; i < NPD; ++i) fast_dy_piece(o, i, dyo[i], dd);
#pragma unroll
            for (int i = 0; i < NPX; ++i) fast_x_piece(o, i, xo[i], xq[i], xd);
            return;
        }
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
                int b = cb + sl, t = t0 + tl, co = co0 + row;
                if (b < a.B && t < a.Tout && co < a.Cout)
                    avc_glds4(a.dy.ptr + ((long)b * a.dy.sb + src_chan_off(a.dy, co) + (long)t * a.dy.st), dd + piece * 64);
                else
                    dd[f] = 0.f;
            }
        }
        for (int piece = wave; piece * 64 < XS; piece += 4) {
            int f = piece * 64 + lane;
            if (f < XS) {
                int row = avc_fastdiv(f, XROW, inv_xrow), pp = f - row * XROW;
                int sl = pp / XSEG, p = pp - sl * XSEG;
                int b = cb + sl, ci = ci0 + row;
                int r = avc_reflect(t0 * a.stride + p - a.padL, a.Tin);
                if (sl < spc && b < a.B && ci < a.Cin && r >= 0 && r < a.Tin)
                    avc_glds4(a.x.ptr + ((long)b * a.x.sb + src_chan_off(a.x, ci) + (long)r * a.x.st), xd + piece * 64);
                else
                    xd[f] = 0.f;
            }
        }
    };

    const int c_begin = z * a.chunks_per_wg;
    int c_end = c_begin + a.chunks_per_wg;
    if (c_end > a.total_chunks) c_end = a.total_chunks;

    __syncthreads();  // zero fill complete before the first DMA lands
    if (c_begin < c_end) issue(c_begin, 0);
    __syncthreads();
    constexpr int TPR = AVC_THREADS / TCO;  // threads per dy row in the bias-gradient partial sum
    constexpr int CPT = 32 / TPR;
    for (int chunk = c_begin; chunk < c_end; ++chunk) {
        const int buf = (chunk - c_begin) & 1;
        const bool more = chunk + 1 < c_end;
        // generic path: stage the next chunk up front; fast path: one slice of its DMAs behind each
        // k-step's MFMAs (a DMA-only phase would leave the matrix pipe idle: one wave per SIMD here)
        if (more && !fastp) issue(chunk + 1, buf ^ 1);
        FastOrigin fo;
        if (more && fastp) fo = fast_origin(chunk + 1);
        float* ndd = dyT + (buf ^ 1) * DYS;
        float* nxd = xT + (buf ^ 1) * XS;
        const float* arow = dyT + buf * DYS + (wave_m * 32 + li) * WG_DYROW;
        const float* brow = xT + buf * XS + (wave_n * NB * 32 + li) * XROW;
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            int qcol = 2 * s + h;
            int sl = qcol >> lgTc, tl = qcol & (Tc - 1);  // Tc is a power of two
            float av = arow[qcol];
            const float* bp = brow + sl * XSEG + tl * a.stride;
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int j = 0; j < KS; ++j)
                    acc[nb * KS + j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bp[nb * 32 * XROW + j], acc[nb * KS + j], 0, 0, 0);
            if (more && fastp) {
#pragma unroll
                for (int i = s; i < NPD; i += 16) fast_dy_piece(fo, i, dyo[i], ndd);
#pragma unroll
                for (int i = s; i < NPX; i += 16) fast_x_piece(fo, i, xo[i], xq[i], nxd);
            }
        }
        if (do_db) {
            const float* dr = dyT + buf * DYS + (tid / TPR) * WG_DYROW + (tid % TPR) * CPT;
#pragma unroll
            for (int k = 0; k < CPT; ++k) dbsum += dr[k];
        }
        __syncthreads();  // next stage landed (the DMA is drained before the barrier), this one is free
    }

    // ---- epilogue: partial tile -> slab[z][tap][co][ci]  (tap-major: the 32 lanes of a half-wave
    // hold 32 consecutive ci of one (tap, co) row -> 128-byte coalesced stores; the reduce kernel
    // restores the [co][ci][tap] parameter layout)
    float* slab = a.slab + (long)z * a.slab_stride;
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
    if (do_db) {
#pragma unroll
        for (int o = 1; o < TPR; o <<= 1) dbsum += __shfl_xor(dbsum, o);
        int co = co0 + tid / TPR;
        if ((tid % TPR) == 0 && co < a.Cout) a.dbslab[(long)z * a.db_stride + co] = dbsum;
    }
}

// out[e] = sum_z slab[z*stride + e]   (fixed order -> deterministic)
struct ReduceArgs {
    ReduceSeg seg[16];
    int nseg;
};

__global__ void __launch_bounds__(AVC_THREADS) slab_reduce_kernel(const ReduceArgs a) {
    const ReduceSeg s = a.seg[blockIdx.y];
    const int plane = s.n / s.KS;  // slab is [tap][rows*Cin]; dst is [rows*Cin][tap]
    for (int e = blockIdx.x * AVC_THREADS + threadIdx.x; e < s.n; e += gridDim.x * AVC_THREADS) {
        // fixed summation order (deterministic); 4 independent loads in flight per thread
        float v = 0.f;
        int zz = 0;
        for (; zz + 4 <= s.nsplit; zz += 4) {
            const float* q = s.slab + (long)zz * s.stride + e;
            float a0 = q[0], a1 = q[s.stride], a2 = q[2 * s.stride], a3 = q[3 * s.stride];
            v = (((v + a0) + a1) + a2) + a3;
        }
        for (; zz < s.nsplit; ++zz) v += s.slab[(long)zz * s.stride + e];
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

void avc_wgrad_plan(int B, int Cin, int Cout, int Tout, int KS, int* Tc, int* spc, int* chunks_per_sample, int* total_chunks,
                    int* chunks_per_wg, int* nsplit) {
    if (Tout >= 32) {
        *Tc = 32;
        *spc = 1;
        *chunks_per_sample = avc_cdiv(Tout, 32);
        *total_chunks = B * *chunks_per_sample;
    } else {
        int p = 1;
        while (p < Tout) p <<= 1;
        *Tc = p;
        *spc = 32 / p;
        *chunks_per_sample = 1;
        *total_chunks = avc_cdiv(B, *spc);
    }
    // split-K factor: enough workgroups to cover the 256 CUs, but at least 4 chunks (128 columns)
    // per workgroup so that the slab write + fixed-order reduce stay a small fraction of the work
    int NB, WCO;
    wgrad_shape(Cin, Cout, KS, &NB, &WCO);
    int tiles = avc_cdiv(Cout, 32 * WCO) * avc_cdiv(Cin, 32 * NB * (4 / WCO));
    static int target_wgs = 0;
    if (target_wgs == 0) {
        const char* e = getenv("AVC_WGRAD_WGS");  // tuning knob (workgroups per launch the split-K aims for)
        target_wgs = e ? atoi(e) : 256;
        if (target_wgs < 1) target_wgs = 256;
    }
    int want = target_wgs / tiles;
    if (want < 1) want = 1;
    int maxsplit = avc_cdiv(*total_chunks, 4);
    if (want > maxsplit) want = maxsplit;
    *chunks_per_wg = avc_cdiv(*total_chunks, want);
    *nsplit = avc_cdiv(*total_chunks, *chunks_per_wg);
}

template <int KS, int NB, int WCO>
static int launch_wgrad_t(const WgradArgs& a, int nsplit, hipStream_t stream) {
    constexpr int TCO = 32 * WCO, TCI = 32 * NB * (4 / WCO);
    int XSEG = (a.Tc - 1) * a.stride + KS;
    int XROW = (a.spc * XSEG) | 1;
    size_t lds = (size_t)2 * (TCO * WG_DYROW + TCI * XROW) * 4 + 16;
    if (lds > 158 * 1024) return -3;
    dim3 grid(avc_cdiv(a.Cout, TCO) * avc_cdiv(a.Cin, TCI), nsplit);
    ProfScope ps(AVC_K_CONV_WGRAD, 2.0 * a.Cout * a.Cin * a.KS * (double)a.B * a.Tout, 0.0, stream);
    hipLaunchKernelGGL((conv_wgrad_kernel<KS, NB, WCO>), grid, dim3(AVC_THREADS), lds, stream, a);
    return (int)hipGetLastError();
}

template <int KS>
static int launch_wgrad_ks(const WgradArgs& a, int nsplit, int WCO, hipStream_t stream) {
    return WCO == 4 ? launch_wgrad_t<KS, 1, 4>(a, nsplit, stream) : launch_wgrad_t<KS, 1, 2>(a, nsplit, stream);
}

int avc_launch_wgrad(const WgradArgs& a, int nsplit, hipStream_t stream) {
    if (a.KS < 1 || a.KS > 8) return -1;
    if (a.padL >= a.Tin) return -6;
    int NB, WCO;
    wgrad_shape(a.Cin, a.Cout, a.KS, &NB, &WCO);
    if (a.KS == 1 && NB == 4) {
        int rc = launch_wgrad_t<1, 4, 4>(a, nsplit, stream);
        if (rc != -3) return rc;
        return launch_wgrad_t<1, 1, 2>(a, nsplit, stream);  // LDS too small for the wide tile (many short samples)
    }
    switch (a.KS) {
        case 1: return launch_wgrad_ks<1>(a, nsplit, WCO, stream);
        case 2: return launch_wgrad_ks<2>(a, nsplit, WCO, stream);
        case 3: return launch_wgrad_ks<3>(a, nsplit, WCO, stream);
        case 4: return launch_wgrad_ks<4>(a, nsplit, WCO, stream);
        case 5: return launch_wgrad_ks<5>(a, nsplit, WCO, stream);
        case 6: return launch_wgrad_ks<6>(a, nsplit, WCO, stream);
        case 7: return launch_wgrad_ks<7>(a, nsplit, WCO, stream);
        default: return launch_wgrad_ks<8>(a, nsplit, WCO, stream);
    }
}

int avc_launch_reduce_segs(const ReduceSeg* segs, int n, hipStream_t stream) {
    if (n < 1 || n > 16) return -1;
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
