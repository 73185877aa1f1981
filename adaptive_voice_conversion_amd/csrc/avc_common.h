// Shared device/host definitions for the AdaIN-VC gfx950 kernels.
// Activations are [B, C, T] fp32 with T contiguous (the reference's layout,
// model.py); every kernel takes explicit element strides so that the
// transposed [B,T,M] view handed over by data_utils.py:14-16 needs no copy.
#pragma once
#include <stdint.h>

#define AVC_MAX_GROUPS 8
#define AVC_THREADS 256

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short avc_s16x4 __attribute__((ext_vector_type(4)));

// compute dtype of the conv / Linear matrix products (storage is fp32 either way)
enum {
    AVC_COMPUTE_F32 = 0,   // v_mfma_f32_32x32x2_f32: bit-exact fp32 (reference precision)
    AVC_COMPUTE_BF16 = 1,  // operands rounded to bf16 (RNE) at fragment time, fp32 accumulate: v_mfma_f32_32x32x8_bf16
    AVC_COMPUTE_F32X3 = 2, // (weight-gradient launches) every operand as three bf16 terms, six v_mfma_f32_32x32x16_bf16 per product
                           // block: fp32-level accuracy (conv_x3_shared.h); instances it is not built for run AVC_COMPUTE_F32
    AVC_COMPUTE_BF16S = 3, // bf16 STORAGE (bf16_pairs.h): operand tensors are bf16 channel pairs [B][C/2][T] in HBM and LDS, weight images
                           // bf16 pairs (AVC_IMG_K4H), v_mfma_f32_32x32x16_bf16, fp32 accumulate
};

// ---- residual / gradient-join modes used by conv epilogues and row kernels
enum {
    AVC_RES_NONE = 0,
    AVC_RES_IDENTITY = 1,   // r[t]
    AVC_RES_AVGPOOL2 = 2,   // fwd: F.avg_pool1d(k=2, ceil_mode=True)  (model.py:248,319)
    AVC_RES_POOLT = 3,      // bwd of AVGPOOL2: g[t/2] * (0.5 | 1 for a clipped last window)
    AVC_RES_UPT = 4,        // bwd of nearest x2: g[2t] + g[2t+1]        (model.py:61-63)
    AVC_RES_UP2 = 5,        // fwd nearest x2: r[t/2]
};

// packed weight images (pack_weight_kernel)
enum {
    AVC_IMG_PLAIN = 0,  // [chunk][tap][r][Mp]: one row of Mp output channels per (tap, reduction channel) -- dense.hip, stacked biases, the d_emb GEMM
    AVC_IMG_X3 = 2,     // split-bf16 image of conv_x3.hip
    AVC_IMG_K4 = 3,     // [chunk][tap][unit][h][Mp][u]: reduction channel = chunk CK + 8 unit + 2 u + h; the four k-steps u of a lane are
                        // 16 contiguous bytes (one ds_read_b128 per four MFMAs) -- conv_gemm.hip
    AVC_IMG_K4H = 4,    // the same image over DWORD channels of bf16 pairs: dword (.., m, u) = bf16 weights of reduction channels 2 dc, 2 dc + 1,
                        // dc = chunk CK + 8 unit + 2 u + h: a lane's 16 bytes are its 8 k-values of one v_mfma_f32_32x32x16_bf16
};

struct ConvSrc {
    const float* ptr;
    long sb, sc;  // element strides of batch / channel
    int st;       // element stride of time
    int ps;       // pixel-(un)shuffle factor: channel c lives at (c/ps)*sc + (c%ps)  (model.py:52-59)
};

struct ConvGroup {
    const float* wp;    // packed weights (AVC_IMG_*)
    const float* bias;  // [M] or null
    float* out;         // primary output (may be null)
    float* out2;        // secondary output (may be null)
    const float* res;   // residual / gradient-join source (may be null)
    const float* mask;  // secondary = value * (mask > 0)      (ReLU backward)
    int KS, padL, padR, nchunk, CK;
    int out_c0;         // ragged launches: first channel of this group's rows in the packed output buffer (the conv bank writes a slice of the concat buffer)
    int pad_[2];
};

// Ragged forward launches (AE.inference over utterances of DIFFERENT lengths in one launch set; reference: inference.py:54-70,
// model.py:387-391 -- the reference itself converts one utterance per call).  Every activation tensor is a PACKED buffer:
// sample b owns a contiguous [channels][T_b] block that starts at element channels * off[b], off = prefix sums of the per-sample
// lengths at that level of the network.  One column tile = 64 output frames of ONE sample; tile[2 i], tile[2 i + 1] = its sample
// and first output frame.  All arrays are device pointers into the plan's workspace (uploaded by avc_forward_ragged).
struct ConvRag {
    const int* tile;      // null: uniform lengths (every other field unused)
    const int* Tsrc;      // per-sample length of the source rows
    const int* offsrc;    // ... and first frame of the sample in the packed source buffer
    const int* Tout;      // per-sample conv output length
    const int* offout;    // first frame of the sample in the packed OUTPUT buffer (its rows hold ops * Tout frames)
    const int* Tres;      // residual rows
    const int* offres;
    int cx, cout, cres;   // channels per sample of the packed source / output / residual buffers
    int ntiles;
};

// InstanceNorm1d(affine=False) + AdaIN affine + activation + residual join of the conv's OUTPUT rows, inside the conv epilogue (round 5):
// for output rows of 16 / 32 / 64 frames a 64-column tile holds whole rows of whole samples, so the statistics need no second kernel
// (model.py:296,341 + append_cond :77-83 + the block bodies :309-320 / :353-369).  The conv's own output `g.out` is the pre-norm row y
// (the backward pass reads it), contiguous [B][C][T] with T = Tout x ops; `out`, `res` and the statistics are those of instnorm_fwd_kernel.
struct ConvINFuse {
    float* out;         // act(((y - mean) rstd) gamma + beta) [+ resmap(res)]; null: nothing fused
    float* mean;        // [B][C] (already offset to the launch's first sample)
    float* rstd;
    const float* cond;  // AdaIN affine [B][cond_sb]: beta = cond[off + c], gamma = cond[off + C + c]; null = plain IN
    long cond_sb;
    int cond_off;
    int res_mode, Tres; // residual rows [B][C][Tres] (contiguous), AVC_RES_IDENTITY / AVGPOOL2 / UP2
    const float* res;
    int C;              // channels of the normalised tensor (= M / ops)
    int relu;
};

// ... and its backward twin (round 5): the InstanceNorm / AdaIN / activation BACKWARD of the rows an input-gradient launch produces.  The
// launch's own result g = conv^T(dy) [+ residual-gradient join] is d(loss)/d(out) of an InstanceNorm layer; where its rows are 16 / 32 / 64
// frames long the tile holds them whole and the epilogue turns g into d(loss)/d(y) = `dy` (what that layer's dgrad and wgrad read) and the
// AdaIN gradients -- the formulas of instnorm_bwd_kernel (rowops.hip).  g itself is still written to `g.out` when that is not null (the
// block's residual path reads it too).
struct ConvINBwd {
    float* dy;          // null: nothing fused
    const float* y;     // the normalised conv's saved output rows [B][C][T] (contiguous)
    const float* mean;  // [B][C] (offset to the launch's first sample)
    const float* rstd;
    const float* cond;  // AdaIN affine [B][cond_sb] or null
    long cond_sb;
    float* dcond;       // [B][dcond_sb]: dbeta -> [off + c], dgamma -> [off + C + c]; null for plain IN
    long dcond_sb;
    int cond_off, dcond_off;
    int C, relu;
};

struct ConvArgs {
    ConvSrc x;
    int B, Cred, Tsrc;
    int mode;    // 0 = forward correlation with reflect padding, 1 = dgrad (zero-extended, zero-upsampled dy)
    int stride;  // forward stride / dgrad upsampling factor
    int mirror;  // dgrad only: add the reflect-padding adjoint (fold) inside the B-fragment fetch
    int M, Mp, Tout;
    long ob, oc;
    int ot, ops;  // output strides (+ pixel-shuffle store factor), shared by out/out2/mask
    int act;      // 0 none, 1 relu / leaky relu (slope)
    float slope;  // 0: ReLU; AVC_LRELU_SLOPE: LeakyReLU -- also the slope the `mask` of a dgrad launch applies to masked-out elements
    int res_mode, res_to_primary;
    long rb, rc;
    int rt, Tres;
    int ngroups;
    int dbg;  // ablation switches of the micro-benchmarks (0 in the product path)
    int bf16; // AVC_COMPUTE_*
    ConvRag rag;
    int img;  // weight image g[0].wp points at: AVC_IMG_K4 (conv_gemm.hip) or AVC_IMG_X3 (conv_x3.hip)
    int pairs; // outputs / residual / mask are bf16 pair tensors (bf16_pairs.h; strides in dwords).  0 with AVC_COMPUTE_BF16S: fp32 outputs
               // from bf16 operands (the heads, the decoder's last conv)
    int par;  // stride-2 dgrad: columns of one parity per wave, each wave multiplies only the taps that meet non-zero
              // positions of the zero-upsampled dy (set by the launcher)
    // ---- tile walk (set by the launcher; conv_gemm.hip): workgroup x of gridDim.x walkers computes the column tiles x, x + gridDim.x, ...
    int walk_n;    // tiles of the longest walk (1: one tile per workgroup)
    int walk_rem;  // walkers [0, walk_rem) take walk_n tiles, the others walk_n - 1
    int walk_db;   // samples between two tiles of a walker (the first frame of the tile never changes along a walk)
    ConvINFuse in;  // fused InstanceNorm epilogue (in.out == null: none)
    ConvINBwd inb;  // fused InstanceNorm-backward epilogue of an input-gradient launch (inb.dy == null: none)
    ConvGroup g[AVC_MAX_GROUPS];
};

// One layer of a weight-gradient launch (conv_wgrad.hip).  A launch is a STREAM-K split of all its layers: the K-chunks (32 columns of
// the (b, t) axis) of every (co, ci) tile of every layer form one sequence, weighted by chunk_cost; workgroup w of `grid` owns the chunks
// whose start cost lies in [ceil(w C / grid), ceil((w + 1) C / grid)) -- exactly `grid` workgroups (one per CU), balanced to one chunk,
// whatever the layers' shapes are.  A workgroup that walked a tile's whole K range stores the finished gradient tile; otherwise it
// stores its partial sum (accumulator layout) into that tile's slot z = (its index among the tile's workgroups), and the batch's reduce
// launch sums the slots in the fixed order z = 0, 1, ... (bit-deterministic) into the flat gradient buffer, bias gradients included.
struct WgradArgs {
    ConvSrc x;    // conv input  [B, Cin, Tin]   (reflect padded on the fly)
    ConvSrc dy;   // output grad [B, Cout, Tout]
    int B, Cin, Cout, Tin, Tout;
    int KS, padL, stride;
    int chunks_per_sample, total_chunks, Tc, spc;  // K-chunk geometry (32 columns per chunk)
    int bf16;      // AVC_COMPUTE_*
    int tiles;     // (co, ci) tiles of this layer
    // ---- filled by avc_wgrad_plan_batch
    int grp;          // launch (kernel instance) of the planned batch this layer belongs to
    int grid;         // workgroups of that launch
    int chunk_cost;   // cost units of one K-chunk of this layer (taps + a fixed part)
    int slots;        // slab slots per tile (>= the number of workgroups any tile of this layer is split over)
    int tNB, tWCO, tCW; // tile shape of the layer's kernel instance: tCW consumer waves, (32 tWCO) co x (32 tNB tCW / tWCO) ci
    int cw8;          // (caller) 1: the k = 5 layers may take the eight-consumer-wave 128 x 64 tile (avc_tuning.wgrad_cw8)
    int rows_per_src; // (caller) stacked layers (heads, AdaIN affines): output rows per parameter tensor
    long cost_begin;  // cost units in front of this layer inside its launch
    long cost_total;  // ... of the whole launch
    long slab_need, dbslab_need;   // floats the caller must provide at slab / dbslab
    // ---- filled by the caller
    float* slab;   // [tiles][slots][tile floats] partial tiles, accumulator layout
    float* dbslab; // [co tiles][slots][co rows]  partial bias sums (null: no bias gradient)
    float* dw;     // finished weight gradient [Cout][Cin][KS] (parameter layout); source s of a stacked layer at dw + s * dw_src_stride
    float* db;     // finished bias gradient (may be null)
    long dw_src_stride, db_src_stride;
};
#define AVC_WGRAD_MAXL 16
struct WgradBatch {
    int nlayers;
    int dbg;   // ablation switches of the micro-benchmarks (0 in the product path)
    int lds_floats;  // floats of the dynamic LDS segment (zero-filled once per workgroup)
    int pad_;
    WgradArgs L[AVC_WGRAD_MAXL];
};

#define AVC_DENSE_MAXL 17
struct DenseLayer {
    const float* wp;    // packed weight image [Kp][Mp] (forward image, or the dgrad image for the backward kernel)
    const float* bias;  // forward only
    float* act;         // ReLU output of this layer [C][B] (written forward, read as mask backward)
    float* out2;        // forward, second Linear of a block: h_{l+1} [C][B]
    float* dz;          // backward: gradient wrt this layer's pre-activation [C][B]
    int Cin, Cout, Kp, Mp;
};
struct DenseArgs {
    DenseLayer layer[AVC_DENSE_MAXL];
    int nlayers, B, C, Kmax, Wmax;
    const float* in;    // forward: pooled [C][B]; backward: d(emb) [c_out][B] channel-major
    const float* in2;   // backward: optional upstream d(emb) [B][c_out] row-major, added to `in`
    float* emb;         // forward output [B][c_out]
    float* dpooled;     // backward output [C][B]
    float slope;        // activation slope (0 = ReLU)
};

struct INFwdArgs {
    const float* y;   // conv output rows [R][T]
    float* out;       // relu((y-mean)*rstd*gamma+beta) [+ resmap(res)]
    float* mean;      // [R] saved for backward
    float* rstd;      // [R]
    const float* cond;  // AdaIN affine [B][cond_sb]: beta = cond[off + c], gamma = cond[off + C + c]; null = plain IN
    long cond_sb;
    int cond_off;
    const float* res;   // residual rows [R][Tres] (contiguous) or null
    int res_mode, Tres;
    int R, C, T, relu;
    float slope;
    int planar;         // pair kernels (rowops_pairs.hip): y rows are natural bf16 rows (the output of a pixel-shuffling conv)
    int nv_hint;        // pair kernels: 16-byte vectors per lane (avc_tuning.in_pairs_nv)
};

// ragged InstanceNorm forward (ragged_rows.hip): packed [C][T_b] blocks, see ConvRag
struct RagINArgs {
    const float* y;
    float* out;
    const int* T;       // per-sample row length
    const int* off;     // first frame of the sample in the packed y / out buffers
    const float* cond;  // AdaIN affine [B][cond_sb] or null
    long cond_sb;
    int cond_off;
    const float* res;   // packed residual buffer or null
    const int* Tres;
    const int* offres;
    int res_mode;
    int B, C;
    float slope;
};

struct INBwdArgs {
    const float* g;   // dL/d(out) rows [R][T]
    const float* y;   // conv output (pre-norm), as in forward
    const float* mean;
    const float* rstd;
    const float* cond;
    long cond_sb;
    int cond_off;
    float* dy;        // dL/dy
    float* dcond;     // [B][dcond_sb]: dbeta -> [off + c], dgamma -> [off + C + c]; null for plain IN
    long dcond_sb;
    int dcond_off;
    int R, C, T, relu;
    float slope;
    int planar;       // pair kernels: y and dy rows are natural bf16 rows
    int nv_hint;
};

struct AdamArgs {
    float* p;
    float* g;
    float* m;
    float* v;
    float* vmax;
    long n;
    const float* partial;  // per-block sums of g^2
    int npartial;
    float grad_prescale;   // 1/world_size after a summing all-reduce, else 1
    float max_norm, weight_decay, beta1, beta2, eps, sqrt_bc2, step_size;
    int amsgrad, write_clipped;
    float* gnorm_out;
};

static inline __host__ __device__ int avc_reflect(int v, int T) {
    // F.pad(mode='reflect'): mirror without repeating the edge (model.py:28-30)
    if (v < 0) v = -v;
    if (v >= T) v = 2 * (T - 1) - v;
    return v;
}
// direct global -> LDS DMA (global_load_lds_dwordx4): every lane supplies its own 16-byte source
// address, the destination is the wave-uniform LDS base + lane * 16 (cdna_hip_programming.md §5)
static __device__ __forceinline__ void avc_glds16(const float* gsrc, float* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

static __device__ __forceinline__ void avc_glds4(const float* gsrc, float* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 4, 0, 0);
}

// same, dword, with a wave-uniform base and a per-lane unsigned BYTE offset: selects the SADDR form
// (global_load_lds_dword voff, s[base:base+1]) -- no 64-bit address arithmetic per lane
static __device__ __forceinline__ void avc_glds4_s(const float* sbase, unsigned voff_bytes, float* lds_wave_base) {
    const char* p = (const char*)sbase + voff_bytes;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 4, 0, 0);
}

static __device__ __forceinline__ void avc_glds16_s(const float* sbase, unsigned voff_bytes, float* lds_wave_base) {
    const char* p = (const char*)sbase + voff_bytes;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// nn.ReLU / nn.LeakyReLU(0.01) (model.py:93-99 get_act): slope = 0 -> ReLU (bit-identical to fmaxf), 0.01 -> 'lrelu'.
// The backward masks are (stored activation > 0) either way: a positive slope keeps the sign of the pre-activation.
#define AVC_LRELU_SLOPE 0.01f
static __device__ __forceinline__ float avc_act(float v, float slope) { return slope == 0.f ? fmaxf(v, 0.f) : (v > 0.f ? v : v * slope); }
static __device__ __forceinline__ float avc_act_grad(float g, bool positive, float slope) { return positive ? g : (slope == 0.f ? 0.f : g * slope); }

#define AVC_IN_EPS 1e-5f
// x_hat and the ReLU pre-activation of the InstanceNorm family are computed by the SAME explicitly
// rounded sequence wherever they appear (row kernels forward and backward, fused conv epilogue), so a
// recomputed ReLU mask is bit-identical to the forward decision whatever the compiler contracts.
static __device__ __forceinline__ float in_xhat(float y, float mean, float rstd) { return __fmul_rn(__fsub_rn(y, mean), rstd); }
static __device__ __forceinline__ float in_preact(float xh, float gamma, float beta) { return __fmaf_rn(xh, gamma, beta); }

// four fp32 -> four bf16 (round to nearest even), packed as the A/B operand of v_mfma_f32_32x32x8_bf16:
// slot j of lane-half h carries reduction index k = 2j + h in BOTH operands (any bijection works as
// long as A and B agree, the sum over k does not care about the order)
#ifndef AVC_EMU
static __device__ __forceinline__ avc_s16x4 avc_pack_bf16x4(float a, float b, float c, float d) {
    typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
    bf16x4_t v = {(__bf16)a, (__bf16)b, (__bf16)c, (__bf16)d};  // v_cvt_pk_bf16_f32 x2
    return __builtin_bit_cast(avc_s16x4, v);
}
static __device__ __forceinline__ f32x16 avc_mfma_bf16(avc_s16x4 a, avc_s16x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(a, b, c, 0, 0, 0);
}
#else
static inline short avc_bf16_bits(float f) {
    unsigned u;
    memcpy(&u, &f, 4);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (short)(u >> 16);
}
static inline avc_s16x4 avc_pack_bf16x4(float a, float b, float c, float d) {
    avc_s16x4 v = {avc_bf16_bits(a), avc_bf16_bits(b), avc_bf16_bits(c), avc_bf16_bits(d)};
    return v;
}
static inline f32x16 avc_mfma_bf16(avc_s16x4 a, avc_s16x4 b, f32x16 c) { return emu::mfma_32x32x8_bf16(a, b, c); }
#endif

// ---- lane exchange v[lane ^ O] in the VALU: DPP inside a row of 16 lanes, v_permlane16_swap / v_permlane32_swap (gfx950) across rows.
// __shfl_xor compiles to ds_bpermute_b32 -- an LDS-unit round trip per butterfly step, queued behind whatever the co-resident conv
// workgroups do to the CU's LDS.  Bit-identical to the shuffle (v + partner in either order is the same sum); checked lane by lane on the
// GPU by scripts/probe/xor_lane_test.hip.  (Built while hunting the run-to-run differences of round 6; those turned out to be the packed-fp32
// VALU forms -- csrc/build.sh -- not the shuffles.)
#ifndef AVC_EMU
template <int CTRL, int BANKS>
static __device__ __forceinline__ int avc_dpp_i(int old, int v) { return __builtin_amdgcn_update_dpp(old, v, CTRL, 0xf, BANKS, false); }
template <int O>
static __device__ __forceinline__ float avc_xor_get(float v) {
    static_assert(O == 1 || O == 2 || O == 4 || O == 8 || O == 16 || O == 32, "lane xor by a power of two below 64");
    const int x = __builtin_bit_cast(int, v);
    if constexpr (O == 1) return __builtin_bit_cast(float, avc_dpp_i<0xB1, 0xf>(x, x));          // quad_perm [1,0,3,2]
    else if constexpr (O == 2) return __builtin_bit_cast(float, avc_dpp_i<0x4E, 0xf>(x, x));     // quad_perm [2,3,0,1]
    else if constexpr (O == 4) {   // row_ror:4 (lane i <- lane i - 4 mod 16) for the lanes with bit 2 set, row_ror:12 (lane i <- lane i + 4 mod 16) for the others
        const int t = avc_dpp_i<0x124, 0xA>(x, x);
        return __builtin_bit_cast(float, avc_dpp_i<0x12C, 0x5>(t, x));
    } else if constexpr (O == 8) return __builtin_bit_cast(float, avc_dpp_i<0x128, 0xf>(x, x));  // row_ror:8
    else if constexpr (O == 16) {  // swaps the odd rows of the first operand with the even rows of the second: r[0] = (row0, row0, row2, row2), r[1] = (row1, row1, row3, row3)
        const auto r = __builtin_amdgcn_permlane16_swap((unsigned)x, (unsigned)x, false, false);
        return __builtin_bit_cast(float, (__lane_id() & 16) ? r[0] : r[1]);
    } else {                       // swaps the upper 32 lanes of the first operand with the lower 32 of the second
        const auto r = __builtin_amdgcn_permlane32_swap((unsigned)x, (unsigned)x, false, false);
        return __builtin_bit_cast(float, (__lane_id() & 32) ? r[0] : r[1]);
    }
}
#else
template <int O>
static inline float avc_xor_get(float v) { return __shfl_xor(v, O); }
#endif
// run-time offset (wave-uniform): the offsets the kernels use
static __device__ __forceinline__ float avc_xor_get_rt(float v, int o) {
    switch (o) {
        case 1: return avc_xor_get<1>(v);
        case 2: return avc_xor_get<2>(v);
        case 4: return avc_xor_get<4>(v);
        case 8: return avc_xor_get<8>(v);
        case 16: return avc_xor_get<16>(v);
        default: return avc_xor_get<32>(v);
    }
}
// butterfly sum over groups of LPR lanes (LPR a power of two <= 64), largest offset first -- the order the row kernels always used
template <int LPR>
static __device__ __forceinline__ float avc_group_sum(float v) {
    if constexpr (LPR >= 64) v += avc_xor_get<32>(v);
    if constexpr (LPR >= 32) v += avc_xor_get<16>(v);
    if constexpr (LPR >= 16) v += avc_xor_get<8>(v);
    if constexpr (LPR >= 8) v += avc_xor_get<4>(v);
    if constexpr (LPR >= 4) v += avc_xor_get<2>(v);
    if constexpr (LPR >= 2) v += avc_xor_get<1>(v);
    return v;
}

// f / d for 0 <= f < 2^22 with a precomputed float reciprocal (one fix-up step; integer division
// by a run-time divisor costs ~25 instructions on gfx950, this costs ~6)
static __device__ __forceinline__ int avc_fastdiv(int f, int d, float inv_d) {
    int q = (int)((float)f * inv_d);
    int r = f - q * d;
    if (r < 0) { --q; r += d; }
    if (r >= d) ++q;
    return q;
}

static inline __host__ __device__ int avc_cdiv(int a, int b) { return (a + b - 1) / b; }
