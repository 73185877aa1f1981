// Register-stationary Conv1d for gfx950: the WEIGHTS live in the register file, activations stream
// through LDS.
//
// Same contraction and the same fused epilogues as conv_gemm.hip (reference: model.py:21-32 pad_layer +
// nn.Conv1d and its input gradient), restructured around two facts of this chip and this model:
//
//  * every conv of the AdaIN-VC blocks multiplies a SMALL weight (128 x 128 x 5 = 328 KB) with a long
//    activation stream (B x T columns); in the LDS-tiled kernel every workgroup re-stages the whole weight
//    image through LDS (1024 workgroups x 164 KB per launch), ten times the activation traffic;
//  * the exact-fp32 MFMA (v_mfma_f32_32x32x2_f32) takes ONE VGPR per operand and 64 cycles per instruction,
//    and a CDNA4 wave that is alone on its SIMD owns 512 registers.
//
// So a workgroup is 4 waves (one per SIMD); wave w owns output rows [32w, 32w+32) and loads its 32 x K weight
// slab ONCE into registers (K/2 fp32 A-fragments: 320 VGPRs for 128 x 5), then walks 32-column activation
// tiles persistently: per tile the four waves stage ONE [Cred][ROWP] source tile in LDS by dword LDS-DMA
// (reflect padding / zero extension / zero-upsampling resolved in the per-lane source offsets, as in
// conv_gemm.hip) and each MFMA needs a single ds_read_b32 (the B fragment, a shifted window of that tile).
// No weight traffic in the loop, one barrier per tile (320 MFMAs per wave), the next tile's DMA issued in
// the gaps between MFMAs.
//
// dgrad (mode 1) is the same kernel on the transposed / tap-flipped slab; the adjoint of the reflect padding
// is folded into the B fetch exactly as in conv_gemm.hip.
#include <hip/hip_runtime.h>

#include <type_traits>

#include "avc_common.h"
#include "avc_internal.h"
#include "conv_shared.h"

#define RS_BN 32

// A k-step = 2 reduction channels (lane half h) of one tap.
// A image: [slab][q = ks / 4][lane][4]  (ks = c2 * KS + j), 16 bytes per lane per load, 1 KiB per wave-load.
template <int KS, int C2, int ROWP, bool MIRROR>
__global__ void __launch_bounds__(256) conv_rs_kernel(const ConvArgs a) {
    constexpr int NKS = KS * C2;             // k-steps
    constexpr int NQ = (NKS + 3) / 4;        // float4 A registers
    constexpr int CRED = 2 * C2;             // reduction channels staged per tile
    constexpr int XS = CRED * ROWP;          // floats per X stage
    HIP_DYNAMIC_SHARED(float, smem)
    const ConvGroup g = a.g[0];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, h = lane >> 5;
    const int padL = g.padL, padR = g.padR;
    const int Tout = a.Tout;
    const int slab = blockIdx.y * 4 + wave;

    // ---- weights -> registers (once per workgroup)
    f32x4 aw[NQ];
    {
        const f32x4* img = (const f32x4*)g.wp + (long)slab * NQ * 64 + lane;
#pragma unroll
        for (int q = 0; q < NQ; ++q) aw[q] = img[q * 64];
    }

    // this wave's 16 accumulator rows are the same for every tile: bias once
    const int m_base = blockIdx.y * 128 + wave * 32;
    float biasv[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = m_base + (r & 3) + 8 * (r >> 2) + 4 * h;
        biasv[r] = (g.bias && m < a.M) ? g.bias[m] : 0.f;
    }
    const bool has_res = a.res_mode != AVC_RES_NONE, has_mask = g.out2 && g.mask;

    const int ntiles = (Tout >= RS_BN) ? a.B * avc_cdiv(Tout, RS_BN) : avc_cdiv(a.B, RS_BN / Tout);
    const ConvGeom q0 = conv_geom(a.mode, a.stride, Tout, KS, RS_BN, 0);
    const int SEG = q0.SEG, ROWDATA = q0.ROWDATA, SPT = q0.SPT;

    // Staging is branch-free: every lane of every DMA instruction always loads.  Lane l owns LDS positions
    // p = 64 * j + l of every row; its source is a running 64-bit pointer (row 0 of the tile + the position's
    // offset, advanced by 4 channel rows per step) -- or, where the tile holds a structural zero (zero extension
    // / zero-upsampling of dgrad, samples past the batch, the null window), the zero block that closes the
    // weight image, with stride 0.
    const float* zsrc = g.wp + (long)gridDim.y * 4 * NQ * 256 + lane;
    const long rowstep = 4 * a.x.sc;   // this wave stages rows wave, wave + 4, ...
    auto tile_sources = [&](int tile, bool live, const float* (&src)[ROWP / 64], long (&step)[ROWP / 64]) {
        const ConvGeom q = conv_geom(a.mode, a.stride, Tout, KS, RS_BN, tile);
#pragma unroll
        for (int j = 0; j < ROWP / 64; ++j) {
            const int p = 64 * j + lane;
            long sp = -1;
            if (live && p < ROWDATA) {
                int seg = 0, qq = p;
                if (SPT > 1) {
                    seg = p / SEG;
                    qq = p - seg * SEG;
                }
                const int b = q.b0 + seg;
                const int pp = q.seg_p0 + qq;
                if (b < a.B) {
                    if (a.mode == 0) {
                        int r = avc_reflect(pp - padL, a.Tsrc);
                        r = r < 0 ? 0 : (r >= a.Tsrc ? a.Tsrc - 1 : r);   // (only columns whose outputs are discarded)
                        sp = b * a.x.sb + (long)r * a.x.st;
                    } else {
                        const int v = pp - (KS - 1);
                        if (v >= 0) {
                            const int vs = v / a.stride;
                            if (vs * a.stride == v && vs < a.Tsrc) sp = b * a.x.sb + (long)vs * a.x.st;
                        }
                    }
                }
            }
            src[j] = sp >= 0 ? a.x.ptr + (long)wave * a.x.sc + sp : zsrc;
            step[j] = sp >= 0 ? rowstep : 0;
        }
    };
    // stage this wave's next row of a tile (row index advances by 4 per call)
    auto stage_next_row = [&](const float* (&src)[ROWP / 64], const long (&step)[ROWP / 64], float* Xrow) {
#pragma unroll
        for (int j = 0; j < ROWP / 64; ++j) {
            avc_glds4(src[j], Xrow + 64 * j);   // (ROWP = 64 * ceil(ROW / 64): every 64-float piece holds tile positions)
            src[j] += step[j];
        }
    };

    int tile = blockIdx.x;
    if (tile >= ntiles) return;
    const float* src_n[ROWP / 64];
    long step_n[ROWP / 64];
    tile_sources(tile, true, src_n, step_n);
    for (int i = 0; i < CRED / 4; ++i) stage_next_row(src_n, step_n, smem + (wave + 4 * i) * ROWP);
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();

    for (int it = 0; tile < ntiles; tile += gridDim.x, ++it) {
        const float* Xb = smem + (it & 1) * XS;
        float* Xn = smem + ((it + 1) & 1) * XS + wave * ROWP;
        const int next = tile + gridDim.x;
        tile_sources(next < ntiles ? next : tile, next < ntiles, src_n, step_n);   // (past the last tile: zeros into the idle stage)

        // ---- this lane's column of the tile
        const ConvGeom q = conv_geom(a.mode, a.stride, Tout, KS, RS_BN, tile);
        int bl, t;
        bool v;
        if (SPT == 1 && Tout >= RS_BN) {
            bl = 0;
            t = q.t0 + li;
            v = (t < Tout) && (q.b0 < a.B);
        } else {
            bl = li / Tout;
            t = li - bl * Tout;
            v = (bl < SPT) && (q.b0 + bl < a.B);
        }
        int cb = ROWDATA, cbl = ROWDATA, cbr = ROWDATA;   // ROWDATA.. = the null window (zeros)
        if (v) {
            if (a.mode == 0) {
                cb = bl * SEG + (t - q.t0) * a.stride;
            } else {
                cb = bl * SEG + (t - q.t0) + padL;
                if (MIRROR) {
                    if (t >= 1 && t <= padL) cbl = bl * SEG + (padL - t - q.seg_p0);
                    if (t >= Tout - 1 - padR && t <= Tout - 2) cbr = bl * SEG + (2 * (Tout - 1) - t + padL - q.seg_p0);
                }
            }
        }
        bool use_mirror = false;
        if (MIRROR) use_mirror = __any((cbl != ROWDATA) || (cbr != ROWDATA));

        // residual / mask operands of the epilogue are requested now and land under the MFMA loop (a wave
        // alone on its SIMD has nobody to hide an epilogue round trip behind)
        const int bcol = q.b0 + bl;
        float resv[16], maskv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m_base + (r & 3) + 8 * (r >> 2) + 4 * h;
            resv[r] = 0.f;
            maskv[r] = 1.f;
            if (v && m < a.M) {
                if (has_res) resv[r] = conv_load_res(a, g.res, bcol, m, t);
                if (has_mask) maskv[r] = g.mask[(long)bcol * a.ob + (long)m * a.oc + (long)t * a.ot];   // (ops == 1 with a mask)
            }
        }

        // four accumulators in rotation: an MFMA never waits on the one issued just before it
        f32x16 acc[4];
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;

        const float* xr = Xb + h * ROWP + cb;
        // a column is within pad of at most ONE edge of its sample (Tout >= 6, checked by the launcher): one mirror window
        const float* xm = Xb + h * ROWP + (cbl != ROWDATA ? cbl : cbr);
        auto body = [&](auto mir_tag) {
            constexpr bool MIR = decltype(mir_tag)::value;
            // B fragments of channel pair c2 + 1 are requested before the MFMAs of pair c2 are queued
            float bv[2][KS];
            auto ldb = [&](int c2, float (&d)[KS]) {
#pragma unroll
                for (int j = 0; j < KS; ++j) {
                    float x = xr[2 * c2 * ROWP + j];
                    if (MIR) x = x + xm[2 * c2 * ROWP + j];
                    d[j] = x;
                }
            };
            ldb(0, bv[0]);
#pragma unroll
            for (int c2 = 0; c2 < C2; ++c2) {
                if (c2 + 1 < C2) ldb(c2 + 1, bv[(c2 + 1) & 1]);
                // the next tile's rows ride in the gaps between this tile's MFMAs (each wave: CRED / 4 rows)
                if ((c2 & 1) == 0) stage_next_row(src_n, step_n, Xn + 4 * (c2 >> 1) * ROWP);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < KS; ++j) {
                    const int ks = c2 * KS + j;
                    const float av = aw[ks >> 2][ks & 3];
                    acc[ks & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv[c2 & 1][j], acc[ks & 3], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        if (MIRROR && use_mirror) body(std::true_type{});
        else body(std::false_type{});

        if (v) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m_base + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (m >= a.M) continue;
                float val = ((acc[0][r] + acc[1][r]) + (acc[2][r] + acc[3][r])) + biasv[r];
                if (a.act == 1) val = fmaxf(val, 0.f);
                long o;
                if (a.ops == 1) o = (long)bcol * a.ob + (long)m * a.oc + (long)t * a.ot;
                else o = (long)bcol * a.ob + (long)(m / a.ops) * a.oc + (long)(t * a.ops + (m % a.ops)) * a.ot;
                if (a.res_to_primary) val += resv[r];
                if (g.out) g.out[o] = val;
                if (g.out2) {
                    float v2 = a.res_to_primary ? val : val + resv[r];
                    if (has_mask) v2 = (maskv[r] > 0.f) ? v2 : 0.f;
                    g.out2[o] = v2;
                }
            }
        }

        __builtin_amdgcn_s_waitcnt(0);   // this wave's DMAs of the next tile have landed ...
        __syncthreads();                 // ... and everybody is done reading the current stage
    }
}

// weight image (packed by pack_one in conv_gemm.hip, PackArgs.rs): Wrs[slab][q][lane][u], ks = 4q + u = c2 * KS + j,
// c = 2 * c2 + (lane >> 5), m = 32 * slab + (lane & 31), then 128 zeros
//   fwd  : W[m][c][j]            (state_dict layout [Cout][Cin][KS])
//   dgrad: W[c][m][KS - 1 - j]   (reduction over the forward output channels, taps flipped)

// --------------------------------------------------------------------------
// host side
// --------------------------------------------------------------------------
static int g_conv_rs = 0;  // avc_set_tuning("conv_rs", 0|1): plans created while it is 1 use this kernel where eligible.
                           // Default 0: measured SLOWER than the LDS-tiled kernel on MI355X (profiles/r02_mfma_probe.log)
void avc_set_conv_rs(int on) { g_conv_rs = on ? 1 : 0; }

// shapes the register-stationary kernel is instantiated for: the k=5, 128-channel convs of the conv blocks
static bool rs_shape_ok(int mode, int Cred, int KS, int stride, int Tout, int xps);
bool avc_conv_rs_eligible(int mode, int Cred, int KS, int stride, int Tout, int xps) {
    return g_conv_rs && rs_shape_ok(mode, Cred, KS, stride, Tout, xps);
}
static bool rs_shape_ok(int mode, int Cred, int KS, int stride, int Tout, int xps) {
    if (KS != 5 || Cred != 128 || xps != 1) return false;
    if (stride != 1 && stride != 2) return false;
    if (mode == 1 && Tout < 6) return false;   // (a column could be within pad of both edges: the kernel sums one mirror window)
    const ConvGeom q = conv_geom(mode, stride, Tout, KS, RS_BN, 0);
    return q.ROW <= 128;
}

long avc_conv_rs_image_floats(int M, int Cred, int KS) {
    const int C2 = (Cred + 1) / 2, NQ = (KS * C2 + 3) / 4;
    const int nslab = avc_cdiv(M, 128) * 4;
    return (long)nslab * NQ * 256 + 128;   // + a block of zeros: the DMA source of structural zeros
}

void avc_pack_rs_args(PackArgs& p, const float* w, int Cout, int Cin, int KS, int dgrad, float* dst) {
    memset(&p, 0, sizeof(p));
    const int M = dgrad ? Cin : Cout, Cred = dgrad ? Cout : Cin;
    p.src[0] = w;
    p.nsrc = 1; p.rows_per_src = Cout;
    p.Cout = Cout; p.Cin = Cin; p.KS = KS; p.dgrad = dgrad;
    p.dst = dst;
    p.rs = 1;
    p.rs_nq = (KS * ((Cred + 1) / 2) + 3) / 4;
    p.rs_nslab = avc_cdiv(M, 128) * 4;
}

int avc_launch_pack_rs(const float* w, int Cout, int Cin, int KS, int dgrad, float* dst, hipStream_t stream) {
    PackArgs p;
    avc_pack_rs_args(p, w, Cout, Cin, KS, dgrad, dst);
    return avc_launch_pack(p, stream);
}

template <int ROWP>
static void launch_rs_rowp(const ConvArgs& a, bool mir, dim3 grid, size_t lds, hipStream_t stream) {
    if (mir) hipLaunchKernelGGL((conv_rs_kernel<5, 64, ROWP, true>), grid, dim3(256), lds, stream, a);
    else hipLaunchKernelGGL((conv_rs_kernel<5, 64, ROWP, false>), grid, dim3(256), lds, stream, a);
}

int avc_launch_conv_rs(const ConvArgs& a, hipStream_t stream) {
    if (a.ngroups != 1 || a.in_fuse) return -1;
    const ConvGroup& g = a.g[0];
    if (!rs_shape_ok(a.mode, a.Cred, g.KS, a.stride, a.Tout, a.x.ps)) return -2;
    if (a.mode == 0 && (g.padL >= a.Tsrc || g.padR >= a.Tsrc)) return -6;
    const ConvGeom q = conv_geom(a.mode, a.stride, a.Tout, g.KS, RS_BN, 0);
    const int ROWP = q.ROW <= 64 ? 64 : 128;
    const size_t lds = (size_t)2 * 128 * ROWP * 4;
    const int ntiles = (a.Tout >= RS_BN) ? a.B * avc_cdiv(a.Tout, RS_BN) : avc_cdiv(a.B, RS_BN / a.Tout);
    const int ny = a.Mp / 128;
    int nx = ntiles < 256 / ny ? ntiles : 256 / ny;   // one workgroup per CU (4 waves x ~400 registers)
    if (nx < 1) nx = 1;
    // even split of the tiles over the persistent workgroups
    const int per = avc_cdiv(ntiles, nx);
    nx = avc_cdiv(ntiles, per);
    const double flops = 2.0 * a.M * a.Cred * g.KS * (double)a.B * (a.mode == 0 ? a.Tout : a.Tsrc);
    ProfScope ps(a.mode == 0 ? AVC_K_CONV_FWD : AVC_K_CONV_DGRAD, flops, 0.0, stream);
    const bool mir = a.mode == 1 && a.mirror;
    dim3 grid(nx, ny);
    if (ROWP == 64) launch_rs_rowp<64>(a, mir, grid, lds, stream);
    else launch_rs_rowp<128>(a, mir, grid, lds, stream);
    return (int)hipGetLastError();
}
