// Op-level C-ABI entry points (declared in include/avc_hip.h): one function per
// kernel family and direction, raw device pointers + explicit sizes + stream.
// Used by the op-level parity tests and by external callers that want single
// kernels; the whole-model path goes through engine.hip.
#include <hip/hip_runtime.h>
#include <math.h>

#include "avc_common.h"
#include "avc_internal.h"

#include <string.h>

// ---- tuning: library defaults + the calling thread's op-level copy (include/avc_hip.h: no process-wide mutable state)
static avc_tuning make_default_tuning() {
    avc_tuning t;
    memset(&t, 0, sizeof(t));
    t.struct_size = (int)sizeof(avc_tuning);
    t.dec_split_min = 128;   // r3 (profiles/r03_tune_sweeps.log): B = 64 is 3.6 % faster unsplit (3.12 vs 3.24 ms), B = 256 0.4 % faster split
    t.dgrad_par = 1;
    t.side_prio = 1;   // round 5 (profiles/r05_tune_sweep.log): 5.89 vs 6.02 ms per step on two boxes, four A/B pairs; neutral in rounds 3-4, before the
                       // stream-K weight gradient and the fused InstanceNorm epilogues changed what shares the chip with the side branch
    t.bank_switch = 1;
    t.conv_ck5 = 8;
    t.wgrad_batch = 12;
    t.wgrad_batch_wgs = 256;
    t.wgrad_target_wgs = 256;
    t.wgrad_batch_units = 1L << 40;
    t.tile_thr11 = 8192;   // r2 sweeps (profiles/r02_tune_sweeps.log)
    t.tile_thr21 = 4096;
    t.ck16_wgs = 256;
    t.ck32_wgs = 256;
    t.kg_wgs = 256;
    t.bh_ck5 = 8;    // r3 (profiles/r03_bf16s_tune.log): 2.75-2.78 ms/step with 8, 2.83 with 16
    t.in_pairs_nv = 1;
    t.wgrad_cw8 = 0;
    t.dec_wgrad_flush = 0;
    t.dec_wgrad_wgs = 128;
    t.conv_walk = 0;
    t.conv_walk_min = 2;
    t.conv_in_fuse = 1;
    return t;
}
const avc_tuning& avc_default_tuning() {
    static const avc_tuning t = make_default_tuning();
    return t;
}
avc_tuning& avc_op_tuning() {
    static thread_local avc_tuning t = make_default_tuning();
    return t;
}

extern "C" {
void avc_tuning_init(avc_tuning* t) {
    if (t) *t = avc_default_tuning();
}
void avc_get_op_tuning(avc_tuning* out) {
    if (out) *out = avc_op_tuning();
}
// one field of the calling thread's op-level tuning, by name
int avc_set_tuning(const char* name, int value) {
    if (!name) return -1;
    avc_tuning& t = avc_op_tuning();
#define AVC_TUNE_FIELD(f) if (!strcmp(name, #f)) { t.f = value; return 0; }
    AVC_TUNE_FIELD(single_stream) AVC_TUNE_FIELD(dec_split_min) AVC_TUNE_FIELD(conv_x3) AVC_TUNE_FIELD(wgrad_x3) AVC_TUNE_FIELD(dgrad_par)
    AVC_TUNE_FIELD(bank_switch) AVC_TUNE_FIELD(conv_ck5) AVC_TUNE_FIELD(wgrad_batch) AVC_TUNE_FIELD(wgrad_batch_wgs) AVC_TUNE_FIELD(wgrad_target_wgs)
    AVC_TUNE_FIELD(conv_ablation) AVC_TUNE_FIELD(wgrad_ablation) AVC_TUNE_FIELD(op_compute_dtype) AVC_TUNE_FIELD(tile12_wgs) AVC_TUNE_FIELD(side_prio) AVC_TUNE_FIELD(wgrad_batch_units)
    AVC_TUNE_FIELD(tile_thr11) AVC_TUNE_FIELD(tile_thr21) AVC_TUNE_FIELD(ck16_wgs) AVC_TUNE_FIELD(ck32_wgs) AVC_TUNE_FIELD(kg_wgs) AVC_TUNE_FIELD(bh_ck5) AVC_TUNE_FIELD(in_pairs_nv) AVC_TUNE_FIELD(conv_min_lds) AVC_TUNE_FIELD(wgrad_cw8) AVC_TUNE_FIELD(dec_wgrad_flush) AVC_TUNE_FIELD(dec_wgrad_wgs)
    AVC_TUNE_FIELD(conv_walk) AVC_TUNE_FIELD(conv_walk_min) AVC_TUNE_FIELD(conv_in_fuse) AVC_TUNE_FIELD(dbg_streams)
#undef AVC_TUNE_FIELD
    if (!strcmp(name, "compute")) { t.op_compute_dtype = (value == AVC_COMPUTE_BF16) ? AVC_COMPUTE_BF16 : AVC_COMPUTE_F32; return 0; }
    return -1;
}

// split-bf16 weight image of conv_x3.hip (tile code 97 of avc_conv1d_fwd / avc_conv1d_dgrad)
long avc_packed_weight_floats_x3(int Cout, int Cin, int KS, int dgrad) {
    const int Cred = dgrad ? Cout : Cin;
    if (!((KS == 5 && Cred % 16 == 0) || (KS == 1 && Cred >= 32))) return -1;
    return avc_conv_x3_image_floats(dgrad ? Cin : Cout, Cred, KS);
}
int avc_pack_weight_x3(const float* w, int Cout, int Cin, int KS, int dgrad, float* dst, void* stream) {
    if (avc_packed_weight_floats_x3(Cout, Cin, KS, dgrad) < 0) return -2;
    PackArgs p;
    avc_pack_x3_args(p, w, Cout, Cin, KS, dgrad, dst);
    return avc_launch_pack(p, (hipStream_t)stream);
}

// device-side segment feed: out[b, m, t] = corpus[starts[b] + t, m]   (data_utils.py:10-22,51-54 on the device)
int avc_gather_segments(const float* corpus, long n_rows, int M, const long* starts, int B, int T, float* out, void* stream) {
    if (!corpus || !starts || !out) return -1;
    return avc_launch_gather_segments(corpus, n_rows, M, starts, B, T, out, (hipStream_t)stream);
}

// op-level compute selector: 3 = bf16 pair storage in and out, 4 = bf16 pair operands with fp32 outputs (the heads / last decoder conv)
static bool op_bh() { const int d = avc_op_tuning().op_compute_dtype; return d == AVC_COMPUTE_BF16S || d == 4; }
static void op_compute(ConvArgs& a) {
    const int d = avc_op_tuning().op_compute_dtype;
    a.bf16 = d == 4 ? AVC_COMPUTE_BF16S : d;
    a.pairs = d == AVC_COMPUTE_BF16S ? 1 : 0;
}

int avc_instnorm_fwd(const float* y, int B, int C, int T, const float* cond, long cond_sb, int cond_off, int relu, const float* res, int res_mode,
                     int Tres, float* out, float* mean, float* rstd, void* stream);
int avc_instnorm_fwd_pairs(const void* y, int B, int C, int T, const float* cond, long cond_sb, int cond_off, int relu, const void* res, int res_mode,
                           int Tres, int planar, void* out, float* mean, float* rstd, void* stream);
int avc_instnorm_bwd(const float* g, const float* y, const float* mean, const float* rstd, int B, int C, int T, const float* cond, long cond_sb,
                     int cond_off, int relu, float* dy, float* dcond, long dcond_sb, int dcond_off, void* stream);

long avc_packed_weight_floats(int Cout, int Cin, int KS, int dgrad) {
    int CK = avc_conv_ck(avc_op_tuning(), KS);
    int red = dgrad ? Cout : Cin, M = dgrad ? Cin : Cout;
    if (op_bh()) red = (red + 1) / 2;   // dword channels of bf16 pairs
    int nchunk = avc_cdiv(red, CK), Mp = avc_cdiv(M, 128) * 128;
    return (long)nchunk * KS * CK * Mp;
}

int avc_pack_weight(const float* const* srcs, int nsrc, int rows_per_src, int Cout, int Cin, int KS, int dgrad,
                    float* dst, void* stream) {
    if (nsrc < 1 || nsrc > 12 || nsrc * rows_per_src != Cout) return -1;
    PackArgs p;
    memset(&p, 0, sizeof(p));
    for (int i = 0; i < nsrc; ++i) p.src[i] = srcs[i];
    p.nsrc = nsrc;
    p.rows_per_src = rows_per_src;
    p.Cout = Cout;
    p.Cin = Cin;
    p.KS = KS;
    p.dgrad = dgrad;
    p.CK = avc_conv_ck(avc_op_tuning(), KS);
    p.img = AVC_IMG_K4;
    int red = dgrad ? Cout : Cin;
    if (op_bh()) { red = (red + 1) / 2; p.img = AVC_IMG_K4H; }
    p.M = dgrad ? Cin : Cout;
    p.nchunk = avc_cdiv(red, p.CK);
    p.Mp = avc_cdiv(p.M, 128) * 128;
    p.dst = dst;
    return avc_launch_pack(p, (hipStream_t)stream);
}

// y = act(conv1d(reflect_pad(x), W) + bias) [; y2 = y + resmap(res)]      (model.py:21-32)
int avc_conv1d_fwd(const float* x, long sxb, long sxc, int sxt, int B, int Cin, int Tin, const float* wp,
                   const float* bias, int Cout, int KS, int stride, int act, float* out, long ob, long oc, int ot,
                   int ops, const float* res, int res_mode, long rb, long rc, int rt, int Tres, float* out2,
                   int tile, void* stream) {
    ConvArgs a;
    memset(&a, 0, sizeof(a));
    op_compute(a);
    const bool bh = op_bh();
    if (bh && (Cin & 1)) return -2;
    a.x.ptr = x; a.x.sb = sxb; a.x.sc = sxc; a.x.st = sxt; a.x.ps = 1;
    a.B = B; a.Cred = bh ? Cin / 2 : Cin; a.Tsrc = Tin;
    a.mode = 0; a.stride = stride; a.mirror = 0;
    int padL = KS / 2, padR = (KS % 2 == 0) ? KS / 2 - 1 : KS / 2;
    a.M = Cout; a.Mp = avc_cdiv(Cout, 128) * 128;
    a.Tout = (Tin + padL + padR - KS) / stride + 1;
    a.ob = ob; a.oc = oc; a.ot = ot; a.ops = ops;
    a.act = act ? 1 : 0;
    a.slope = act == 2 ? AVC_LRELU_SLOPE : 0.f;
    a.res_mode = res_mode; a.res_to_primary = 0;
    a.rb = rb; a.rc = rc; a.rt = rt; a.Tres = Tres;
    a.ngroups = 1;
    a.g[0].CK = avc_conv_ck(avc_op_tuning(), KS);
    a.g[0].wp = wp; a.g[0].bias = bias; a.g[0].out = out; a.g[0].out2 = out2; a.g[0].res = res; a.g[0].mask = nullptr;
    a.g[0].KS = KS; a.g[0].padL = padL; a.g[0].padR = padR; a.g[0].nchunk = avc_cdiv(a.Cred, a.g[0].CK);
    a.img = bh ? AVC_IMG_K4H : AVC_IMG_K4;
    if (tile == 97) { a.img = AVC_IMG_X3; a.g[0].CK = KS == 1 ? 32 : 16; a.g[0].nchunk = avc_cdiv(Cin, a.g[0].CK); }   // wp is a split-bf16 image (avc_pack_weight_x3)
    return avc_launch_conv(a, (hipStream_t)stream, tile, avc_op_tuning());
}

// y = conv1d(reflect_pad(x)) + bias (pixel-shuffled on store when ops == 2), then out = act(InstanceNorm(y) * gamma + beta) [+ resmap(res)] with the
// statistics saved -- ONE launch where the conv's 64-column tile holds whole rows (rows of 16 / 32 / 64 conv frames: the fused epilogue
// of csrc/conv_shared.h), else the conv and the row kernel.  *fused (optional) reports which.  y / out / res are contiguous [B][C][T],
// C = Cout / ops, T = Tout x ops.  Reference: model.py:309-320 / :353-369 (conv -> norm -> [append_cond] -> act [-> + residual]).
int avc_conv1d_in_fwd(const float* x, long sxb, long sxc, int sxt, int B, int Cin, int Tin, const float* wp, const float* bias, int Cout, int KS,
                      int stride, int ops, float* y, const float* cond, long cond_sb, int cond_off, int relu, const float* res, int res_mode,
                      int Tres, float* out, float* mean, float* rstd, int* fused, void* stream) {
    const bool bh = op_bh();   // op_compute_dtype 3: x / y / out / res are bf16 pair tensors (dwords [B][C/2][T]), wp a pair image; no pixel shuffle
    if ((ops != 1 && ops != 2) || Cout % ops) return -2;
    if (bh && (avc_op_tuning().op_compute_dtype != AVC_COMPUTE_BF16S || ops != 1 || (Cin & 1) || (Cout & 1))) return -2;
    ConvArgs a;
    memset(&a, 0, sizeof(a));
    op_compute(a);
    a.x.ptr = x; a.x.sb = sxb; a.x.sc = sxc; a.x.st = sxt; a.x.ps = 1;
    a.B = B; a.Cred = bh ? Cin / 2 : Cin; a.Tsrc = Tin;
    a.mode = 0; a.stride = stride;
    const int padL = KS / 2, padR = (KS % 2 == 0) ? KS / 2 - 1 : KS / 2;
    a.M = Cout; a.Mp = avc_cdiv(Cout, 128) * 128;
    a.Tout = (Tin + padL + padR - KS) / stride + 1;
    const int C = Cout / ops, T = a.Tout * ops;
    a.ob = (long)(bh ? C / 2 : C) * T; a.oc = T; a.ot = 1; a.ops = ops;
    a.slope = relu == 2 ? AVC_LRELU_SLOPE : 0.f;
    a.ngroups = 1;
    a.g[0].CK = avc_conv_ck(avc_op_tuning(), KS);
    a.g[0].wp = wp; a.g[0].bias = bias; a.g[0].out = y;
    a.g[0].KS = KS; a.g[0].padL = padL; a.g[0].padR = padR; a.g[0].nchunk = avc_cdiv(a.Cred, a.g[0].CK);
    a.img = bh ? AVC_IMG_K4H : AVC_IMG_K4;
    const bool fuse = avc_conv_in_fusable(a, avc_op_tuning(), res ? res_mode : AVC_RES_NONE, Tres);
    if (fused) *fused = fuse ? 1 : 0;
    if (fuse) {
        a.in.out = out; a.in.mean = mean; a.in.rstd = rstd;
        a.in.cond = cond; a.in.cond_sb = cond_sb; a.in.cond_off = cond_off;
        a.in.res = res; a.in.res_mode = res ? res_mode : 0; a.in.Tres = Tres;
        a.in.C = C; a.in.relu = relu ? 1 : 0;
        return avc_launch_conv(a, (hipStream_t)stream, 0, avc_op_tuning());
    }
    int rc = avc_launch_conv(a, (hipStream_t)stream, 0, avc_op_tuning());
    if (rc) return rc;
    if (bh) return avc_instnorm_fwd_pairs(y, B, C, T, cond, cond_sb, cond_off, relu, res, res_mode, Tres, 0, out, mean, rstd, stream);
    return avc_instnorm_fwd(y, B, C, T, cond, cond_sb, cond_off, relu, res, res_mode, Tres, out, mean, rstd, stream);
}

// dx = conv1d_input_grad(dy) including the adjoint of the reflect padding
//   [+ resT(res)] ; dx2 = dx * (mask > 0)
int avc_conv1d_dgrad(const float* dy, long syb, long syc, int syt, int yps, int B, int Cout, int Tdy, const float* wpd,
                     int Cin, int KS, int stride, int Tin, float* dx, long ob, long oc, int ot, const float* res,
                     int res_mode, long rb, long rc, int rt, int Tres, float* dx2, const float* mask, int tile,
                     void* stream) {
    ConvArgs a;
    memset(&a, 0, sizeof(a));
    op_compute(a);
    const bool bh = op_bh();
    if (bh && (Cout & 1)) return -2;
    a.x.ptr = dy; a.x.sb = syb; a.x.sc = syc; a.x.st = syt; a.x.ps = yps;
    a.B = B; a.Cred = bh ? Cout / 2 : Cout; a.Tsrc = Tdy;
    a.mode = 1; a.stride = stride;
    int padL = KS / 2, padR = (KS % 2 == 0) ? KS / 2 - 1 : KS / 2;
    a.mirror = (KS > 1) ? 1 : 0;
    a.M = Cin; a.Mp = avc_cdiv(Cin, 128) * 128;
    a.Tout = Tin;
    a.ob = ob; a.oc = oc; a.ot = ot; a.ops = 1;
    a.act = 0;
    a.res_mode = res_mode; a.res_to_primary = 1;
    a.rb = rb; a.rc = rc; a.rt = rt; a.Tres = Tres;
    a.ngroups = 1;
    a.g[0].CK = avc_conv_ck(avc_op_tuning(), KS);
    a.g[0].wp = wpd; a.g[0].bias = nullptr; a.g[0].out = dx; a.g[0].out2 = dx2; a.g[0].res = res; a.g[0].mask = mask;
    a.g[0].KS = KS; a.g[0].padL = padL; a.g[0].padR = padR; a.g[0].nchunk = avc_cdiv(a.Cred, a.g[0].CK);
    a.img = bh ? AVC_IMG_K4H : AVC_IMG_K4;
    if (tile == 97) { a.img = AVC_IMG_X3; a.g[0].CK = KS == 1 ? 32 : 16; a.g[0].nchunk = avc_cdiv(Cout, a.g[0].CK); }
    return avc_launch_conv(a, (hipStream_t)stream, tile, avc_op_tuning());
}

// workspace (floats) needed by avc_conv1d_wgrad: the partial-tile slots of the stream-K launch
static void op_wgrad_args(WgradArgs& a, int B, int Cin, int Cout, int Tin, int Tout, int KS, int stride) {
    memset(&a, 0, sizeof(a));
    const avc_tuning& t = avc_op_tuning();
    a.bf16 = (t.op_compute_dtype == AVC_COMPUTE_F32 && t.wgrad_x3) ? AVC_COMPUTE_F32X3 : (op_bh() ? AVC_COMPUTE_BF16S : t.op_compute_dtype);
    a.B = B; a.Cin = Cin; a.Cout = Cout; a.Tin = Tin; a.Tout = Tout;
    a.KS = KS; a.padL = KS / 2; a.stride = stride;
    a.x.ps = 1; a.dy.ps = 1;
    a.cw8 = t.wgrad_cw8 ? 1 : 0;
}
// The backward twin of avc_conv1d_in_fwd: g = conv1d_input_grad(dy) [+ resT(res)] is d(loss)/d(out) of an InstanceNorm / AdaIN layer whose saved
// forward rows are y / mean / rstd; returns d(loss)/d(y) in dy_out (+ dcond) -- inside the input-gradient launch's epilogue where its tile holds
// whole rows (rows of 16 / 32 / 64 frames, exact fp32; *fused = 1), otherwise as that launch followed by the row kernel.  g itself is written
// to g_out when it is not NULL (a fused launch whose g nobody else reads may pass NULL; the two-launch path needs it as scratch).
int avc_conv1d_dgrad_in_bwd(const float* dy, long syb, long syc, int syt, int yps, int B, int Cout, int Tdy, const float* wpd, int Cin, int KS,
                            int stride, int Tin, float* g_out, const float* res, int res_mode, int Tres, const float* y, const float* mean,
                            const float* rstd, const float* cond, long cond_sb, int cond_off, int relu, float* dy_out, float* dcond,
                            long dcond_sb, int dcond_off, int* fused, void* stream) {
    const bool bh = op_bh();   // op_compute_dtype 3: dy / res / y / g_out / dy_out are bf16 pair tensors (dwords [B][C/2][T]), wpd a pair image
    if (bh && (avc_op_tuning().op_compute_dtype != AVC_COMPUTE_BF16S || (Cin & 1) || (Cout & 1))) return -2;
    ConvArgs a;
    memset(&a, 0, sizeof(a));
    op_compute(a);
    a.x.ptr = dy; a.x.sb = syb; a.x.sc = syc; a.x.st = syt; a.x.ps = yps;
    a.B = B; a.Cred = bh ? Cout / 2 : Cout; a.Tsrc = Tdy;
    a.mode = 1; a.stride = stride;
    const int padL = KS / 2, padR = (KS % 2 == 0) ? KS / 2 - 1 : KS / 2;
    a.mirror = (KS > 1) ? 1 : 0;
    a.M = Cin; a.Mp = avc_cdiv(Cin, 128) * 128;
    a.Tout = Tin;
    const long Cr = bh ? Cin / 2 : Cin;   // rows per sample of the [B, Cin, T] tensors (pair rows with bh)
    a.ob = Cr * Tin; a.oc = Tin; a.ot = 1; a.ops = 1;
    a.slope = relu == 2 ? AVC_LRELU_SLOPE : 0.f;
    a.res_mode = res ? res_mode : AVC_RES_NONE; a.res_to_primary = 1;
    a.rb = Cr * Tres; a.rc = Tres; a.rt = 1; a.Tres = Tres;
    a.ngroups = 1;
    a.g[0].CK = avc_conv_ck(avc_op_tuning(), KS);
    a.g[0].wp = wpd; a.g[0].out = g_out; a.g[0].res = res;
    a.g[0].KS = KS; a.g[0].padL = padL; a.g[0].padR = padR; a.g[0].nchunk = avc_cdiv(a.Cred, a.g[0].CK);
    a.img = bh ? AVC_IMG_K4H : AVC_IMG_K4;
    const bool fuse = avc_conv_inb_fusable(a, avc_op_tuning());
    if (fused) *fused = fuse ? 1 : 0;
    if (fuse) {
        a.inb.dy = dy_out; a.inb.y = y; a.inb.mean = mean; a.inb.rstd = rstd;
        a.inb.cond = cond; a.inb.cond_sb = cond_sb; a.inb.cond_off = cond_off;
        a.inb.dcond = dcond; a.inb.dcond_sb = dcond_sb; a.inb.dcond_off = dcond_off;
        a.inb.C = Cin; a.inb.relu = relu ? 1 : 0;
        return avc_launch_conv(a, (hipStream_t)stream, 0, avc_op_tuning());
    }
    if (!g_out) return -1;
    int rc = avc_launch_conv(a, (hipStream_t)stream, 0, avc_op_tuning());
    if (rc) return rc;
    if (bh) return avc_instnorm_bwd_pairs(g_out, y, mean, rstd, B, Cin, Tin, cond, cond_sb, cond_off, relu, 0, dy_out, dcond, dcond_sb, dcond_off, stream);
    return avc_instnorm_bwd(g_out, y, mean, rstd, B, Cin, Tin, cond, cond_sb, cond_off, relu, dy_out, dcond, dcond_sb, dcond_off, stream);
}

long avc_conv1d_wgrad_ws_floats(int B, int Cin, int Cout, int Tout, int KS) {
    long need = 0;
    for (int stride = 1; stride <= 2; ++stride) {   // (the query does not know the stride: both geometries fit)
        WgradArgs a;
        op_wgrad_args(a, B, Cin, Cout, Tout * stride, Tout, KS, stride);
        avc_wgrad_plan_batch(&a, 1, avc_op_tuning().wgrad_target_wgs);
        const long n = a.slab_need + a.dbslab_need + 64;
        need = n > need ? n : need;
    }
    return need;
}
int avc_conv1d_wgrad(const float* x, long sxb, long sxc, int sxt, const float* dy, long syb, long syc, int syt, int yps,
                     int B, int Cin, int Cout, int Tin, int Tout, int KS, int stride, float* dW, float* db, float* ws,
                     void* stream) {
    WgradArgs a;
    op_wgrad_args(a, B, Cin, Cout, Tin, Tout, KS, stride);
    a.x.ptr = x; a.x.sb = sxb; a.x.sc = sxc; a.x.st = sxt; a.x.ps = 1;
    a.dy.ptr = dy; a.dy.sb = syb; a.dy.sc = syc; a.dy.st = syt; a.dy.ps = yps;
    avc_wgrad_plan_batch(&a, 1, avc_op_tuning().wgrad_target_wgs);
    a.slab = ws;
    a.dbslab = db ? ws + a.slab_need : nullptr;
    a.dw = dW; a.db = db; a.rows_per_src = Cout;
    return avc_launch_wgrad_batch(&a, 1, (hipStream_t)stream, avc_op_tuning().wgrad_ablation);
}

// out = relu((y - mean_T)/sqrt(var_T + 1e-5) * gamma + beta) [+ resmap(res)]; saves mean/rstd
int avc_instnorm_fwd(const float* y, int B, int C, int T, const float* cond, long cond_sb, int cond_off, int relu,
                     const float* res, int res_mode, int Tres, float* out, float* mean, float* rstd, void* stream) {
    INFwdArgs a;
    a.y = y; a.out = out; a.mean = mean; a.rstd = rstd;
    a.cond = cond; a.cond_sb = cond_sb; a.cond_off = cond_off;
    a.res = res; a.res_mode = res ? res_mode : 0; a.Tres = Tres;
    a.R = B * C; a.C = C; a.T = T; a.relu = relu ? 1 : 0;
    a.slope = relu == 2 ? AVC_LRELU_SLOPE : 0.f;
    a.planar = 0; a.nv_hint = 0;
    return avc_launch_in_fwd(a, (hipStream_t)stream);
}

int avc_instnorm_bwd(const float* g, const float* y, const float* mean, const float* rstd, int B, int C, int T,
                     const float* cond, long cond_sb, int cond_off, int relu, float* dy, float* dcond, long dcond_sb,
                     int dcond_off, void* stream) {
    INBwdArgs a;
    a.g = g; a.y = y; a.mean = mean; a.rstd = rstd;
    a.cond = cond; a.cond_sb = cond_sb; a.cond_off = cond_off;
    a.dy = dy; a.dcond = dcond; a.dcond_sb = dcond_sb; a.dcond_off = dcond_off;
    a.R = B * C; a.C = C; a.T = T; a.relu = relu ? 1 : 0;
    a.slope = relu == 2 ? AVC_LRELU_SLOPE : 0.f;
    a.planar = 0; a.nv_hint = 0;
    return avc_launch_in_bwd(a, (hipStream_t)stream);
}

// ---- the same two on bf16 PAIR rows (include/avc_hip.h "bf16 pair storage"): y / out / res / g / dy are dword tensors [B][C/2][T];
// planar != 0: y (and dy) are natural bf16 [B][C][T] rows, the output layout of a pixel-shuffling conv
int avc_instnorm_fwd_pairs(const void* y, int B, int C, int T, const float* cond, long cond_sb, int cond_off, int relu, const void* res, int res_mode,
                           int Tres, int planar, void* out, float* mean, float* rstd, void* stream) {
    INFwdArgs a;
    a.y = (const float*)y; a.out = (float*)out; a.mean = mean; a.rstd = rstd;
    a.cond = cond; a.cond_sb = cond_sb; a.cond_off = cond_off;
    a.res = (const float*)res; a.res_mode = res ? res_mode : 0; a.Tres = Tres;
    a.R = B * (C / 2); a.C = C; a.T = T; a.relu = relu ? 1 : 0;
    a.slope = relu == 2 ? AVC_LRELU_SLOPE : 0.f;
    a.planar = planar ? 1 : 0;
    a.nv_hint = (int)avc_op_tuning().in_pairs_nv;
    return avc_launch_in_fwd_pairs(a, (hipStream_t)stream);
}
int avc_instnorm_bwd_pairs(const void* g, const void* y, const float* mean, const float* rstd, int B, int C, int T, const float* cond, long cond_sb,
                           int cond_off, int relu, int planar, void* dy, float* dcond, long dcond_sb, int dcond_off, void* stream) {
    INBwdArgs a;
    a.g = (const float*)g; a.y = (const float*)y; a.mean = mean; a.rstd = rstd;
    a.cond = cond; a.cond_sb = cond_sb; a.cond_off = cond_off;
    a.dy = (float*)dy; a.dcond = dcond; a.dcond_sb = dcond_sb; a.dcond_off = dcond_off;
    a.R = B * (C / 2); a.C = C; a.T = T; a.relu = relu ? 1 : 0;
    a.slope = relu == 2 ? AVC_LRELU_SLOPE : 0.f;
    a.planar = planar ? 1 : 0;
    a.nv_hint = (int)avc_op_tuning().in_pairs_nv;
    return avc_launch_in_bwd_pairs(a, (hipStream_t)stream);
}
// fp32 [B, C, T] (explicit element strides) -> bf16 pairs [B][C/2][T]
int avc_to_pairs(const float* x, long sxb, long sxc, long sxt, int B, int C, int T, void* dst, void* stream) {
    return avc_launch_to_pairs(x, sxb, sxc, sxt, B, C, T, (float*)dst, (long)(C / 2) * T, T, (hipStream_t)stream);
}

long avc_clip_adam_ws_floats(long n) { return avc_adam_blocks(n); }

// clip_grad_norm_(max_norm) + Adam(amsgrad, coupled weight decay) on flat buffers; step is 1-based
int avc_clip_adam_step(float* p, float* g, float* m, float* v, float* vmax, long n, int step, float lr, float beta1,
                       float beta2, float eps, float weight_decay, int amsgrad, float max_norm, float grad_prescale,
                       int write_clipped, float* ws, float* gnorm_out, void* stream) {
    int rc = avc_launch_sumsq(g, n, ws, (hipStream_t)stream);
    if (rc) return rc;
    AdamArgs a;
    a.p = p; a.g = g; a.m = m; a.v = v; a.vmax = vmax; a.n = n;
    a.partial = ws; a.npartial = avc_adam_blocks(n);
    a.grad_prescale = grad_prescale; a.max_norm = max_norm; a.weight_decay = weight_decay;
    a.beta1 = beta1; a.beta2 = beta2; a.eps = eps;
    double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    a.sqrt_bc2 = (float)sqrt(bc2); a.step_size = (float)((double)lr / bc1);
    a.amsgrad = amsgrad; a.write_clipped = write_clipped; a.gnorm_out = gnorm_out;
    return avc_launch_clip_adam(a, (hipStream_t)stream);
}

}  // extern "C"
