// fp32-accurate Conv1d on the BF16 matrix core: every operand is the sum of THREE bf16 terms
//     x = hi + mid + lo            (8 + 8 + 8 mantissa bits; hi / mid by truncation of the exact remainders)
// and a product keeps the six partial products above 2^-24 of its magnitude,
//     a b ~= hi.hi + hi.mid + mid.hi + hi.lo + lo.hi + mid.mid,
// each a v_mfma_f32_32x32x16_bf16 with fp32 accumulation.  Six of them (6 x 8 passes) cover 16 reduction steps for which
// the exact-fp32 v_mfma_f32_32x32x2_f32 needs 8 x 16 passes: 2.7x fewer matrix-pipe cycles.  Measured on MI355X
// (scripts/probe/bf16x3_probe.hip, profiles/r02_bf16x3_probe.log): max error 1.3e-7 of the result's scale (an fp32 fmaf chain
// of the same products: 1.8e-7) and 185-255 fp32-equivalent TFLOP/s in the inner loop against 100-134 for the fp32 MFMA.
//
// Same contraction, geometry, LDS-DMA staging and fused epilogue as conv_gemm.hip (reference: model.py:21-32 pad_layer +
// nn.Conv1d and its input gradient) for its k = 5 layers (16-channel chunks) and its 1x1 convs (32-channel chunks, the last one
// zero-padded by the image: the 1104 -> 128 in_conv takes 100 instead of 136 us); the workgroup tile is 64 rows x 128 columns with
// the four waves side by side (64 x 32 each), so that one operand split feeds twelve MFMAs.  Differences:
//   * the weights are split ONCE per optimizer step by the pack kernel into a k-contiguous bf16 image
//     [chunk][tap][term][k-half][m][8 bf16]: a lane's A fragment of a term is one 16-byte LDS read;
//   * the source tile stays fp32 [channel][position] (DMA'd as before); a lane reads its 8 channels of a tap, adds the
//     mirror window of the reflect adjoint where needed, and splits the 8 values in registers (~50 VALU per 6 MFMAs).
// STATUS: opt-in.  Op level: tile code 97 of avc_conv1d_fwd / avc_conv1d_dgrad + avc_pack_weight_x3; whole-model plans created
// after avc_set_tuning("conv_x3", 1) run their k = 5 layers that fill the chip on it (the default engine multiplies in exact fp32).
//  Parity is green on hardware (error 0.6-3.6x that of an fp32 convolution against fp64).  Measured against the exact-fp32
// kernel (profiles/r02_conv_micro_x3.log): forward 53.6 vs 68.5 us at B=256, T=128 (100 vs 78 TFLOP/s), 185 vs 228 us at B=1024
// (116 vs 94); mirrored dgrad 63.8 vs 69.1 us; layers with <= 256 workgroups of 64x128 (T_l <= 64 at B=256) are slower.  A
// first version with 32x32 per wave (one split per six MFMAs) was no faster at all; what still separates this one from the probe's
// 240 TFLOP/s is everything around the MFMA loop (30 KB of weight image per chunk through LDS, the barriers, 2 workgroups per CU).
#include <hip/hip_runtime.h>

#include "avc_common.h"
#include "avc_internal.h"
#include "conv_shared.h"

#include "conv_x3_shared.h"

template <bool MIRROR, int KS, int KB>
__global__ void __launch_bounds__(AVC_THREADS) conv_x3_kernel(const ConvArgs a) {
    constexpr int BM = 64, BN = 128, CK = 16 * KB, AROWS = KS * KB * 6;   // 4 waves side by side, each 64 rows x 32 columns: ONE split feeds 12 MFMAs
    // (BM = 128 -- one split per 24 MFMAs, one workgroup per CU -- measured no better: 55.2 vs 53.6 us at T=128)
    constexpr int WM = BM / 32;
    HIP_DYNAMIC_SHARED(float, smem)
    const ConvGroup g = a.g[0];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave_n = wave;
    const int li = lane & 31, h = lane >> 5;
    const int padL = g.padL, padR = g.padR, nchunk = g.nchunk;
    const int Tout = a.Tout;
    const ConvGeom q = conv_geom(a.mode, a.stride, Tout, KS, BN, blockIdx.x);
    const int ROW = q.ROW;
    const int m_tile0 = blockIdx.y * BM;
    constexpr int AS = AROWS * BM * 4;   // floats per A stage (k = 5: 30 KiB, k = 1: 12 KiB)
    const int XS = CK * ROW;
    float* As = smem;
    float* Xs = smem + 2 * AS;

    // per-lane source descriptors of the X tile (as conv_gemm.hip: position p = 64 j + lane of every row)
    int xoff[AVC_CONV_NJ];
#pragma unroll
    for (int j = 0; j < AVC_CONV_NJ; ++j) {
        const int p = 64 * j + lane;
        int sp = -1;
        if (p < q.ROWDATA) {
            const int seg = p / q.SEG, qq = p - seg * q.SEG;
            const int b = q.b0 + seg, pp = q.seg_p0 + qq;
            if (b < a.B) {
                if (a.mode == 0) {
                    const int r = avc_reflect(pp - padL, a.Tsrc);
                    if (r >= 0 && r < a.Tsrc) sp = (int)(b * a.x.sb + (long)r * a.x.st);
                } else {
                    const int v = pp - (KS - 1);
                    if (v >= 0) {
                        const int vs = v / a.stride;
                        if (vs * a.stride == v && vs < a.Tsrc) sp = (int)(b * a.x.sb + (long)vs * a.x.st);
                    }
                }
            }
        }
        xoff[j] = sp;
    }
    for (int e = tid; e < 2 * XS; e += AVC_THREADS) Xs[e] = 0.f;   // structural zeros are never overwritten afterwards

    // the lane's column (a.par: one column parity per wave -- stride-2 dgrad multiplies only the taps of that parity:
    // waves 0, 2 the even columns, waves 1, 3 the odd ones)
    const int n = wave_n * 32 + li;
    const int par = wave_n & 1, pi = (wave_n >> 1) * 32 + li;   // parity class and slot inside it (64 slots per tile)
    int bl, t;
    bool v;
    if (a.par) {
        if (Tout >= BN) { bl = 0; t = q.t0 + 2 * pi + par; v = (t < Tout) && (q.b0 < a.B); }
        else { const int halfT = Tout >> 1; bl = pi / halfT; t = 2 * (pi - bl * halfT) + par; v = (bl < q.SPT) && (q.b0 + bl < a.B); }
    } else if (q.SPT == 1 && Tout >= BN) {
        bl = 0; t = q.t0 + n; v = (t < Tout) && (q.b0 < a.B);
    } else {
        bl = n / Tout; t = n - bl * Tout; v = (bl < q.SPT) && (q.b0 + bl < a.B);
    }
    int cb = q.ROWDATA, cbm = q.ROWDATA;   // ROWDATA.. = the null window
    if (v) {
        if (a.mode == 0) cb = bl * q.SEG + (t - q.t0) * a.stride;
        else {
            cb = bl * q.SEG + (t - q.t0) + padL;
            if (MIRROR) {   // (launcher: Tout >= 2 (padL + padR) + 2 -> at most one mirror window per column)
                if (t >= 1 && t <= padL) cbm = bl * q.SEG + (padL - t - q.seg_p0);
                if (t >= Tout - 1 - padR && t <= Tout - 2) cbm = bl * q.SEG + (2 * (Tout - 1) - t + padL - q.seg_p0);
            }
        }
    }
    const bool use_mirror = MIRROR && __any(cbm != q.ROWDATA);

    f32x16 acc[WM];
#pragma unroll
    for (int wm = 0; wm < WM; ++wm)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[wm][r] = 0.f;
    const int nj = (ROW + 63) >> 6;
    __syncthreads();   // zero fill done before the first DMA lands

    auto load_a = [&](int chunk, int buf) {   // 30 rows of BM m x 16 bytes: the packed image is the LDS image
        const float* wsrc = g.wp + ((long)chunk * AROWS * a.Mp + m_tile0) * 4;
        float* Ad = As + buf * AS;
        for (int piece = wave; piece < AROWS * (BM / 64); piece += 4) {
            const int row = piece / (BM / 64), half = piece % (BM / 64);
            avc_glds16(wsrc + ((long)row * a.Mp + half * 64 + lane) * 4, Ad + (row * BM + half * 64) * 4);
        }
    };
    auto load_x = [&](int chunk, int buf) {
        float* Xd = Xs + buf * XS;
        for (int r = wave; r < CK; r += 4) {
            const int c = chunk * CK + r;
            if (c >= a.Cred) continue;   // channel padding of the last chunk: its weights are zeros, the stage holds finite values
            const long coff = (a.x.ps == 1) ? (long)c * a.x.sc : (long)(c / a.x.ps) * a.x.sc + (c % a.x.ps);
            const float* src = a.x.ptr + coff;
#pragma unroll
            for (int j = 0; j < AVC_CONV_NJ; ++j)
                if (j < nj && xoff[j] >= 0) avc_glds4(src + xoff[j], Xd + r * ROW + 64 * j);
        }
    };
    load_a(0, 0);
    load_x(0, 0);
    __syncthreads();

    for (int chunk = 0; chunk < nchunk; ++chunk) {
        if (chunk + 1 < nchunk && !(a.dbg & 1)) {   // (dbg: ablation switches of scripts/conv_ablate.py, timing only)
            load_a(chunk + 1, (chunk + 1) & 1);
            load_x(chunk + 1, (chunk + 1) & 1);
        }
        const avc_u32x4* Ab = (const avc_u32x4*)(As + (chunk & 1) * AS);
        const float* Xb = Xs + (chunk & 1) * XS + (8 * h) * ROW;   // (+ 16 kb rows for block kb)
        // Straight-line taps, software-pipelined by hand: the eight B values of tap j+1 are requested from LDS before the
        // twelve MFMAs of tap j are issued, so the LDS round trip and the split (~50 VALU) of a tap overlap the matrix
        // pipe working on the previous one (a tap loop with its parity branches compiled to read -> wait -> split -> MFMA).
        auto fetch = [&](int unit, float (&x)[8]) {   // unit = tap * KB + block
            const int tap = unit / KB, kb = unit % KB;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                x[k] = Xb[(16 * kb + k) * ROW + cb + tap];
                if (use_mirror) x[k] += Xb[(16 * kb + k) * ROW + cbm + tap];
            }
        };
        auto mma = [&](int unit, const float (&x)[8]) {
            unsigned hi[8], mid[8], lo[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) x3_split(x[k], hi[k], mid[k], lo[k]);
            avc_u32x4 bt[3];
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
                bt[0][qd] = x3_pair(hi[2 * qd], hi[2 * qd + 1]);
                bt[1][qd] = x3_pair(mid[2 * qd], mid[2 * qd + 1]);
                bt[2][qd] = x3_pair(lo[2 * qd], lo[2 * qd + 1]);
            }
            avc_u32x4 at[WM][3];
#pragma unroll
            for (int wm = 0; wm < WM; ++wm)
#pragma unroll
                for (int term = 0; term < 3; ++term) at[wm][term] = Ab[((unit * 3 + term) * 2 + h) * BM + wm * 32 + li];
            // small terms first; the WM accumulators alternate so that consecutive MFMAs are independent
#pragma unroll
            for (int wm = 0; wm < WM; ++wm) acc[wm] = avc_mfma_bf16x8(at[wm][2], bt[0], acc[wm]);
#pragma unroll
            for (int wm = 0; wm < WM; ++wm) acc[wm] = avc_mfma_bf16x8(at[wm][0], bt[2], acc[wm]);
#pragma unroll
            for (int wm = 0; wm < WM; ++wm) acc[wm] = avc_mfma_bf16x8(at[wm][1], bt[1], acc[wm]);
#pragma unroll
            for (int wm = 0; wm < WM; ++wm) acc[wm] = avc_mfma_bf16x8(at[wm][1], bt[0], acc[wm]);
#pragma unroll
            for (int wm = 0; wm < WM; ++wm) acc[wm] = avc_mfma_bf16x8(at[wm][0], bt[1], acc[wm]);
#pragma unroll
            for (int wm = 0; wm < WM; ++wm) acc[wm] = avc_mfma_bf16x8(at[wm][0], bt[0], acc[wm]);
        };
        float xa[8], xb[8];
        if (a.dbg & 2) {
        } else if constexpr (KS == 1) {   // two blocks of 16 channels
            fetch(0, xa);
            fetch(1, xb); mma(0, xa);
            mma(1, xb);
        } else if (!a.par) {
            fetch(0, xa);
            fetch(1, xb); mma(0, xa);
            fetch(2, xa); mma(1, xb);
            fetch(3, xb); mma(2, xa);
            fetch(4, xa); mma(3, xb);
            mma(4, xa);
        } else if (par == 0) {   // even columns of a stride-2 dgrad: taps 0, 2, 4
            fetch(0, xa);
            fetch(2, xb); mma(0, xa);
            fetch(4, xa); mma(2, xb);
            mma(4, xa);
        } else {                 // odd columns: taps 1, 3
            fetch(1, xa);
            fetch(3, xb); mma(1, xa);
            mma(3, xb);
        }
        __syncthreads();
    }
    if (v) {
#pragma unroll
        for (int wm = 0; wm < WM; ++wm) conv_store_frag(conv_epi(a), g, acc[wm], m_tile0 + 32 * wm, h, q.b0 + bl, t);
    }
}

// --------------------------------------------------------------------------
// avc_tuning.conv_x3: 1 = the big k = 5 layers of a plan run on this kernel, 2 = every layer of an eligible shape, whatever its size (tests)
static bool x3_shape_ok(int mode, int Cred, int KS, int stride, int Tout) {
    if (KS == 1) {   // 1x1 conv / Linear over time: 32-channel chunks, the last one zero-padded by the image
        if (Cred < 32 || stride != 1) return false;
    } else if (KS != 5 || Cred < 16 || Cred % 16 != 0) return false;
    if (stride != 1 && stride != 2) return false;
    if (mode == 1 && Tout < 10) return false;   // one mirror window per column
    const ConvGeom q = conv_geom(mode, stride, Tout, KS, 128, 0);
    return q.ROW <= 64 * AVC_CONV_NJ;
}
// a plan layer takes this kernel when the launch fills the chip with 64 x 128 tiles (measured: 256 workgroups still win,
// 128 lose to the exact-fp32 kernel, profiles/r02_conv_micro_x3.log)
bool avc_conv_x3_eligible(const avc_tuning& tun, int mode, int Cred, int KS, int stride, int Tout, int B, int M) {
    if (!tun.conv_x3 || !x3_shape_ok(mode, Cred, KS, stride, Tout)) return false;
    const long ntn = Tout >= 128 ? (long)B * avc_cdiv(Tout, 128) : avc_cdiv(B, 128 / Tout);
    // (the input-gradient launches carry the mask / residual-join epilogue: in the engine they win only from 512 workgroups on)
    return tun.conv_x3 >= 2 || ntn * (avc_cdiv(M, 128) * 2) >= (mode == 1 ? 512 : 256);
}
long avc_conv_x3_image_floats(int M, int Cred, int KS) { return (long)avc_cdiv(Cred, 16 * x3_kb(KS)) * x3_arows(KS) * (avc_cdiv(M, 128) * 128) * 4; }

void avc_pack_x3_args(PackArgs& p, const float* w, int Cout, int Cin, int KS, int dgrad, float* dst, int M_rows) {
    memset(&p, 0, sizeof(p));
    // M_rows > 0: only the first M_rows output rows of the image (the input gradient of a layer whose leading input channels need one)
    const int M = M_rows > 0 ? M_rows : (dgrad ? Cin : Cout), Cred = dgrad ? Cout : Cin;
    p.src[0] = w;
    p.nsrc = 1; p.rows_per_src = Cout;
    p.Cout = Cout; p.Cin = Cin; p.KS = KS; p.dgrad = dgrad;
    p.CK = 16 * x3_kb(KS); p.nchunk = avc_cdiv(Cred, p.CK);
    p.M = M; p.Mp = avc_cdiv(M, 128) * 128;
    p.dst = dst;
    p.img = AVC_IMG_X3;
}

int avc_launch_conv_x3(const ConvArgs& a_in, hipStream_t stream, const avc_tuning& tun) {
    ConvArgs a = a_in;
    a.dbg = tun.conv_ablation;
    if (a.ngroups != 1) return -1;
    const ConvGroup& g = a.g[0];
    if (!x3_shape_ok(a.mode, a.Cred, g.KS, a.stride, a.Tout) || g.CK != 16 * x3_kb(g.KS) || g.nchunk != avc_cdiv(a.Cred, g.CK) || a.Mp % 128 != 0) return -2;
    if (a.mode == 0 && (g.padL >= a.Tsrc || g.padR >= a.Tsrc)) return -6;
    const ConvGeom q = conv_geom(a.mode, a.stride, a.Tout, g.KS, 128, 0);
    const size_t lds = (size_t)(2 * x3_arows(g.KS) * 256 + 2 * g.CK * q.ROW) * 4 + 16;
    if (lds > 160 * 1024) return -5;
    const int ntn = a.Tout >= 128 ? a.B * avc_cdiv(a.Tout, 128) : avc_cdiv(a.B, 128 / a.Tout);
    a.par = g.KS == 5 && a.mode == 1 && a.stride == 2 && g.padL == 2 && (a.Tout >= 128 || (a.Tout % 2 == 0 && 128 % a.Tout == 0));
    dim3 grid(ntn, a.Mp / 64);
    const double flops = 2.0 * a.M * a.Cred * g.KS * (double)a.B * (a.mode == 0 ? a.Tout : a.Tsrc);
    ProfScope ps(a.mode == 0 ? AVC_K_CONV_FWD : AVC_K_CONV_DGRAD, flops, 0.0, stream);
    if (g.KS == 1) hipLaunchKernelGGL((conv_x3_kernel<false, 1, 2>), grid, dim3(AVC_THREADS), lds, stream, a);
    else if (a.mode == 1 && a.mirror) hipLaunchKernelGGL((conv_x3_kernel<true, 5, 1>), grid, dim3(AVC_THREADS), lds, stream, a);
    else hipLaunchKernelGGL((conv_x3_kernel<false, 5, 1>), grid, dim3(AVC_THREADS), lds, stream, a);
    return (int)hipGetLastError();
}
