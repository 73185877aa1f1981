// Whole-model orchestration of the AdaIN-VC autoencoder on gfx950: builds the
// launch plan for one (B, T, T_cond) shape and issues the forward / backward
// kernel sequences on the caller's stream.
//
// Reference being replaced (never copied): AE.forward / AE.inference
// (model.py:380-391), SpeakerEncoder.forward (:265-277), ContentEncoder.forward
// (:301-323), Decoder.forward (:347-371) and their autograd; the loss of
// solver.py:84-88.  Parameter order = state_dict registration order.
//
// HBM layout: one caller-owned fp32 workspace.  Packed weights, every saved
// activation ([B,C,T], T contiguous), IN statistics, gradient temporaries and
// the split-K slabs of the weight gradients live at fixed offsets decided at
// plan creation; nothing is allocated, freed or synchronised here.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <map>
#include <string>
#include <vector>

#include "avc_common.h"
#include "avc_hip.h"
#include "avc_internal.h"

static thread_local std::string g_err;
static int fail(int rc, const char* what) {
    char buf[256];
    snprintf(buf, sizeof(buf), "%s (rc=%d)", what, rc);
    g_err = buf;
    return rc;
}
#define RUN(expr)                                  \
    do {                                           \
        int _rc = (expr);                          \
        if (_rc != 0) return fail(_rc, #expr);     \
    } while (0)

extern "C" const char* avc_last_error(void) { return g_err.c_str(); }
extern "C" int avc_version(void) { return 100; }


struct ParamT {
    long off, numel;
    int d[3];
};

struct LayerP {
    int bf16 = 0;  // AVC_COMPUTE_* of this layer's matrix products (avc_plan_set_compute_dtype)
    int Cout = 0, Cin = 0, KS = 1, stride = 1;
    int nsrc = 1, rows = 0;
    int w[12], b[12];
    long wpf = -1, wpd = -1, bpk = -1;
    int CK = 8, CKd = 8, nchunk_f = 0, nchunk_d = 0, Mp_f = 0, Mp_d = 0, dgM = 0;
    bool need_dgrad = true;
    bool x3_f = false, x3_d = false;   // split-bf16 kernel (conv_x3.hip): image at wrs_f / wrs_d, 16-channel chunks
    long wrs_f = -1, wrs_d = -1;
    bool conv_path = true;             // launched through conv_gemm.hip (AVC_IMG_K4 images); false: the dense stack (AVC_IMG_PLAIN)
    long wplain = -1;                  // extra AVC_IMG_PLAIN forward image (the affine layer: its d_emb GEMM reads it as a [Kp][Mp] matrix)
    bool bh = false;                   // bf16 pair operands (AVC_PLAN_BF16S): AVC_IMG_K4H images over Cin / 2 (Cout / 2) dword channels
};

struct EncNet {
    avc_encoder_cfg c;
    float slope = 0.f;   // activation slope of this network (0 = ReLU, AVC_LRELU_SLOPE = 'lrelu')
    int nb = 0, CC = 0, n = 0, nd = 0;
    int T[AVC_MAX_BLOCKS + 1];
    std::vector<int> bank, c1, c2, dn1, dn2;
    int in_conv = -1, outl = -1, heads = -1;
    long cat = -1, dcat = -1, h0 = -1;              // h0: speaker relu(in_conv) / content y0
    long out[AVC_MAX_BLOCKS + 1];                   // block outputs (out[0] = after in_conv stage)
    long a1[AVC_MAX_BLOCKS], a2[AVC_MAX_BLOCKS];    // speaker: relu outputs; content: a1 = relu(IN(y1))
    long y1[AVC_MAX_BLOCKS], y2[AVC_MAX_BLOCKS];    // content conv outputs (pre-norm)
    long st0 = -1, st1[AVC_MAX_BLOCKS], st2[AVC_MAX_BLOCKS];  // IN stats: mean at off, rstd at off + B*C
    long pooled = -1, d1[AVC_MAX_BLOCKS], d2[AVC_MAX_BLOCKS], hd[AVC_MAX_BLOCKS + 1];
};

struct DecNet {
    avc_decoder_cfg c;
    float slope = 0.f;
    int n = 0;
    int T[AVC_MAX_BLOCKS + 1];
    std::vector<int> c1, c2;
    int in_conv = -1, affine = -1, out_conv = -1;
    long z = -1, cond = -1, dcond = -1, y0 = -1;
    long out[AVC_MAX_BLOCKS + 1], y1[AVC_MAX_BLOCKS], a1[AVC_MAX_BLOCKS], y2[AVC_MAX_BLOCKS];
    long st0 = -1, st1[AVC_MAX_BLOCKS], st2[AVC_MAX_BLOCKS];
};

struct avc_plan {
    avc_model_cfg cfg;
    int B, T, Tc, M, Tb, Tout;
    std::vector<ParamT> params;
    long param_floats = 0;
    std::vector<LayerP> layers;
    EncNet spk, enc;
    DecNet dec;
    long ws_top = 0;
    std::map<std::string, long> named;
    // shared gradient temporaries
    long gA = -1, gB = -1, gC = -1;
    long muls = -1, dmuls = -1, emb = -1, demb = -1, decb = -1, ddec = -1, dz = -1;
    long losses = -1, loss_partial = -1;
    long slab = -1, slab_floats = 0;
    long dhA = -1;
    std::vector<avc_relu_site> sites;
    // second gradient-temporary set + side stream: the speaker and content encoders are independent
    // branches (model.py:381-382) and run concurrently so that their small-T layers fill the chip
    long gA2 = -1, gB2 = -1, gC2 = -1;
    mutable hipStream_t side = nullptr;
    mutable hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    int compute = 0;             // AVC_COMPUTE_*
    mutable int side_state = 0;  // 0 = not created, 1 = ready, -1 = disabled
    // weight-gradient kernels depend on nothing downstream: they run on their own (low priority)
    // streams beside the dgrad / InstanceNorm-backward chain of each branch
    mutable hipStream_t wstream[2] = {nullptr, nullptr};
    mutable std::vector<hipEvent_t> wev;   // sized by the dry run of the backward pass (one per ordering edge)
    mutable hipEvent_t wjoin[2] = {nullptr, nullptr};
    mutable hipEvent_t ev_dense = nullptr;       // recorded behind the dense-stack backward kernel (the decoder's weight gradients wait for it)
    mutable hipEvent_t ev_dec_grads = nullptr;   // recorded when the decoder's parameter gradients are final
    mutable hipEvent_t ev_spk_grads = nullptr;   // ... the speaker encoder's
    mutable hipEvent_t ev_all_grads = nullptr;   // recorded at the end of avc_backward
    long dyarena = -1, dyarena_floats = 0;
    int flags = 0;            // AVC_PLAN_*
    // ragged inference plans (avc_plan_create_ragged): per-level length / offset / tile tables
    struct RagLevel {
        std::vector<int> T, off;   // per-sample frames, prefix sums (B + 1 entries)
        int ntiles = 0;
        long dT = -1, doff = -1, dtile = -1;   // workspace offsets (in floats == ints) of the device copies
    };
    std::vector<RagLevel> rl_spk, rl_enc, rl_dec;   // speaker / content encoder levels 0..n, decoder levels 0..n (rl_dec[0] == rl_enc[n])
    std::vector<int> rag_host;                      // host image of all tables (uploaded by avc_forward_ragged)
    long rag_tab = -1;
    avc_tuning tun;           // launch heuristics / diagnostic switches, captured at plan creation (the dry run sizes slabs and events with them)
    int nev_need = 0;         // events one backward pass records on the wgrad streams
    long dec_param_off = 0;   // first float of the decoder's parameters in the flat buffer (they are the tail)
    long enc_param_off = 0;   // first float of the content encoder's (the speaker encoder's are the head)
    // AVC_PLAN_BF16S: every [B, C, T] activation / activation gradient is a bf16 pair tensor (bf16_pairs.h) at the same workspace
    // offsets (half of each allocation is used); muls / dec / emb / cond and their gradients, statistics, slabs stay fp32
    // every weight image of the plan as ONE launch (avc_plan_pack_weights / the head of avc_forward): descriptor table in device memory,
    // pointers stored as byte offsets from the caller's parameter buffer / workspace.  Built at plan creation, owned by the plan.
    mutable PackArgs* pack_tab_dev = nullptr;
    mutable void* pack_blk_dev = nullptr;   // (image, piece) of every block of the launch
    int pack_nblk = 0, pack_nblk_early = 0, pack_early_imgs = 0;   // blocks [0, pack_nblk_early) pack the images the step's first kernels read
    double pack_tab_bytes = 0;
    bool bh = false;
    long ddecp = -1, dmulsp = -1;   // pair copies of d(dec) and d(muls): the conv launches that consume them read pair operands

    long alloc(long n) {
        long o = ws_top;
        ws_top += (n + 63) / 64 * 64;
        return o;
    }
    const float* par(const float* params_, int idx) const { return params_ + params[idx].off; }
};

// --------------------------------------------------------------------------
// plan construction
// --------------------------------------------------------------------------
static int add_param(avc_plan* p, int d0, int d1, int d2) {
    ParamT t;
    t.d[0] = d0;
    t.d[1] = d1;
    t.d[2] = d2;
    t.numel = (long)d0 * (d1 > 0 ? d1 : 1) * (d2 > 0 ? d2 : 1);
    t.off = p->param_floats;
    p->param_floats += (t.numel + 3) / 4 * 4;
    p->params.push_back(t);
    return (int)p->params.size() - 1;
}

static int add_layer(avc_plan* p, int Cout, int Cin, int KS, int stride, bool conv3d) {
    LayerP L;
    L.Cout = Cout;
    L.Cin = Cin;
    L.KS = KS;
    L.stride = stride;
    L.nsrc = 1;
    L.rows = Cout;
    L.w[0] = add_param(p, Cout, Cin, conv3d ? KS : 0);
    L.b[0] = add_param(p, Cout, 0, 0);
    p->layers.push_back(L);
    return (int)p->layers.size() - 1;
}

// Bn/Tf: batch and output length of the forward launch; Td: output length of the dgrad launch
// conv_path: the layer is launched through avc_launch_conv (false: the dense stack, which reads plain fp32 images in its own kernel)
static void finish_layer(avc_plan* p, LayerP& L, bool need_dgrad, int dgM, int Bn, int Tf, int Td, int ngroups = 1, bool conv_path = true,
                         bool pairs_ok = true) {
    const avc_tuning& tun = p->tun;
    L.conv_path = conv_path;
    L.bh = p->bh && conv_path && pairs_ok;
    if (p->bh) L.bf16 = L.bh ? AVC_COMPUTE_BF16S : AVC_COMPUTE_BF16;   // (fp32-stored operands of the affine / dense layers: rounded on the way in)
    L.Mp_f = avc_cdiv(L.Cout, 128) * 128;
    L.need_dgrad = need_dgrad;
    L.dgM = dgM > 0 ? dgM : L.Cin;
    L.Mp_d = avc_cdiv(L.dgM, 128) * 128;
    int tf = avc_conv_pick_tile(tun, L.Mp_f, Bn, Tf, ngroups, L.Cin * L.KS);
    L.CK = avc_conv_ck_for(tun, L.KS, avc_conv_num_wgs(tf, L.Mp_f, Bn, Tf, ngroups), 0, L.stride, Tf, tf);
    int td = avc_conv_pick_tile(tun, L.Mp_d, Bn, Td, 1, L.Cout * L.KS);
    L.CKd = avc_conv_ck_for(tun, L.KS, avc_conv_num_wgs(td, L.Mp_d, Bn, Td, 1), 1, L.stride, Td, td);
    if (L.bh && L.KS >= 4 && ngroups == 1) {   // chunk depth in dword channels (tuning bh_ck5; measured: 8 beats 16 beats 32 on the step)
        const int want = (tun.bh_ck5 == 16 || tun.bh_ck5 == 32) ? (int)tun.bh_ck5 : 8;
        L.CK = want;
        L.CKd = want;
    }
    L.nchunk_f = avc_cdiv(L.bh ? L.Cin / 2 : L.Cin, L.CK);
    L.nchunk_d = avc_cdiv(L.bh ? L.Cout / 2 : L.Cout, L.CKd);
    L.x3_f = !p->bh && conv_path && ngroups == 1 && L.nsrc == 1 && avc_conv_x3_eligible(tun, 0, L.Cin, L.KS, L.stride, Tf, Bn, L.Cout);
    L.x3_d = !p->bh && conv_path && need_dgrad && ngroups == 1 && L.nsrc == 1 && avc_conv_x3_eligible(tun, 1, L.Cout, L.KS, L.stride, Td, Bn, L.dgM);
    if (L.x3_f) { L.CK = L.KS == 1 ? 32 : 16; L.nchunk_f = avc_cdiv(L.Cin, L.CK); }
    if (L.x3_d) { L.CKd = L.KS == 1 ? 32 : 16; L.nchunk_d = avc_cdiv(L.Cout, L.CKd); }
    if (L.x3_f) L.wrs_f = p->alloc(avc_conv_x3_image_floats(L.Cout, L.Cin, L.KS));
    if (L.x3_d) L.wrs_d = p->alloc(avc_conv_x3_image_floats(L.dgM, L.Cout, L.KS));
    L.wpf = p->alloc((long)L.nchunk_f * L.KS * L.CK * L.Mp_f);
    if (need_dgrad) L.wpd = p->alloc((long)L.nchunk_d * L.KS * L.CKd * L.Mp_d);
    if (L.nsrc > 1) L.bpk = p->alloc((long)32 * L.Mp_f);
}

static int validate_enc(const avc_encoder_cfg& c, bool spk) {
    if (c.c_in < 1 || c.c_h < 1 || c.c_out < 1 || c.c_bank < 1) return -1;
    if (c.kernel_size < 1 || c.kernel_size > 8) return -1;
    if (c.act != 0 && c.act != 1) return -1;
    if (c.bank_scale < 1 || c.bank_size < c.bank_scale || c.bank_size > 8) return -1;
    if (c.bank_size / c.bank_scale > AVC_MAX_GROUPS) return -1;
    if (c.n_conv_blocks < 1 || c.n_conv_blocks > AVC_MAX_BLOCKS) return -1;
    if (spk && (c.n_dense_blocks < 0 || c.n_dense_blocks > AVC_MAX_BLOCKS)) return -1;
    for (int l = 0; l < c.n_conv_blocks; ++l)
        if (c.subsample[l] != 1 && c.subsample[l] != 2) return -1;
    return 0;
}

static void build_enc_params(avc_plan* p, EncNet& e, const avc_encoder_cfg& c, bool spk) {
    e.c = c;
    e.slope = c.act == 1 ? AVC_LRELU_SLOPE : 0.f;
    e.n = c.n_conv_blocks;
    e.nd = spk ? c.n_dense_blocks : 0;
    for (int k = c.bank_scale; k <= c.bank_size; k += c.bank_scale) e.bank.push_back(add_layer(p, c.c_bank, c.c_in, k, 1, true));
    e.nb = (int)e.bank.size();
    e.CC = c.c_bank * (c.bank_size / c.bank_scale) + c.c_in;
    e.in_conv = add_layer(p, c.c_h, e.CC, 1, 1, true);
    for (int l = 0; l < e.n; ++l) e.c1.push_back(add_layer(p, c.c_h, c.c_h, c.kernel_size, 1, true));
    for (int l = 0; l < e.n; ++l) e.c2.push_back(add_layer(p, c.c_h, c.c_h, c.kernel_size, c.subsample[l], true));
    if (spk) {
        for (int l = 0; l < e.nd; ++l) e.dn1.push_back(add_layer(p, c.c_h, c.c_h, 1, 1, false));
        for (int l = 0; l < e.nd; ++l) e.dn2.push_back(add_layer(p, c.c_h, c.c_h, 1, 1, false));
        e.outl = add_layer(p, c.c_out, c.c_h, 1, 1, false);
    } else {
        // mean_layer + std_layer stacked into one 2*c_out head (model.py:297-298,321-322)
        int mean = add_layer(p, c.c_out, c.c_h, 1, 1, true);
        LayerP& L = p->layers[mean];
        int wi = add_param(p, c.c_out, c.c_h, 1), bi = add_param(p, c.c_out, 0, 0);
        L.nsrc = 2;
        L.rows = c.c_out;
        L.Cout = 2 * c.c_out;
        L.w[1] = wi;
        L.b[1] = bi;
        e.heads = mean;
    }
}

static void plan_init_streams(avc_plan* p);

extern "C" int avc_plan_create(const avc_model_cfg* cfg, int B, int T, int T_cond, avc_plan** out) {
    return avc_plan_create_ex(cfg, B, T, T_cond, 0, out);
}

extern "C" int avc_plan_create_ex(const avc_model_cfg* cfg, int B, int T, int T_cond, int flags, avc_plan** out) {
    return avc_plan_create_tuned(cfg, B, T, T_cond, flags, nullptr, out);
}

extern "C" int avc_plan_create_tuned(const avc_model_cfg* cfg, int B, int T, int T_cond, int flags, const avc_tuning* tuning, avc_plan** out) {
    if (!cfg || !out || B < 1 || T < 1) return fail(-1, "avc_plan_create: bad arguments");
    if (flags & ~(AVC_PLAN_INFERENCE | AVC_PLAN_SPEAKER_ONLY | AVC_PLAN_X3 | AVC_PLAN_BF16S)) return fail(-1, "avc_plan_create: unknown flag");
    if ((flags & AVC_PLAN_X3) && (flags & AVC_PLAN_BF16S)) return fail(-1, "avc_plan_create: AVC_PLAN_X3 and AVC_PLAN_BF16S exclude each other");
    if (tuning && tuning->struct_size != (int)sizeof(avc_tuning)) return fail(-1, "avc_plan_create_tuned: avc_tuning of another library version (use avc_tuning_init)");
    if (flags & AVC_PLAN_SPEAKER_ONLY) flags |= AVC_PLAN_INFERENCE;
    if (T_cond <= 0) T_cond = T;
    const bool infer = (flags & AVC_PLAN_INFERENCE) != 0, spk_only = (flags & AVC_PLAN_SPEAKER_ONLY) != 0;
    if (validate_enc(cfg->spk, true) || validate_enc(cfg->enc, false)) return fail(-2, "avc_plan_create: unsupported encoder config");
    const avc_decoder_cfg& dc = cfg->dec;
    if (dc.n_conv_blocks < 1 || dc.n_conv_blocks > AVC_MAX_BLOCKS || 2 * dc.n_conv_blocks > 12 || dc.kernel_size < 1 || dc.kernel_size > 8)
        return fail(-2, "avc_plan_create: unsupported decoder config");
    for (int l = 0; l < dc.n_conv_blocks; ++l)
        if (dc.upsample[l] != 1 && dc.upsample[l] != 2) return fail(-2, "avc_plan_create: upsample must be 1 or 2");
    if (cfg->spk.c_in != cfg->enc.c_in || cfg->enc.c_in != dc.c_out || dc.c_in != cfg->enc.c_out || dc.c_cond != cfg->spk.c_out)
        return fail(-2, "avc_plan_create: inconsistent channel sizes between the three networks");

    avc_plan* p = new avc_plan();
    p->cfg = *cfg;
    p->flags = flags;
    p->tun = tuning ? *tuning : avc_default_tuning();
    if (p->tun.wgrad_batch < 1) p->tun.wgrad_batch = 1;
    if (p->tun.wgrad_batch_wgs < 1) p->tun.wgrad_batch_wgs = 256;
    if (p->tun.wgrad_batch_units < 1) p->tun.wgrad_batch_units = 1L << 40;
    if (p->tun.dec_split_min < 2) p->tun.dec_split_min = 2;
    if (flags & AVC_PLAN_X3) {   // compute mode "fp32x3" 
        if (p->tun.conv_x3 < 1) p->tun.conv_x3 = 1;
        p->tun.wgrad_x3 = 1;
    }
    p->B = B;
    p->T = T;
    p->Tc = T_cond;
    p->M = cfg->enc.c_in;
    if (flags & AVC_PLAN_BF16S) {
        p->bh = true;
        p->compute = AVC_COMPUTE_BF16S;
        p->tun.conv_x3 = 0;
        p->tun.wgrad_x3 = 0;
        // schedule defaults of THIS mode (round 6, profiles/r06_bf16_schedule_sweep.log; the fp32 step measured both neutral or worse):
        // its weight gradients cost an eighth of the fp32 ones' matrix time, so (a) the decoder's go out UNDER the decoder's own latency-bound
        // backward chain, six layers at a time on 192 CUs, instead of being held behind the dense-stack kernel (2.44 - 2.46 vs 2.48 - 2.51 ms),
        // and (b) batches of 16 layers per stream-K launch (2.46 - 2.47 ms).  A caller's own values win: -1 = "held", any other batch size.
        if (p->tun.dec_wgrad_flush == 0) { p->tun.dec_wgrad_flush = 6; p->tun.dec_wgrad_wgs = 192; }
        if (p->tun.wgrad_batch == avc_default_tuning().wgrad_batch) p->tun.wgrad_batch = 16;
    }
    if (p->tun.dec_wgrad_flush < 0) p->tun.dec_wgrad_flush = 0;   // (-1: held, explicitly)

    // ---- parameters in reference registration order
    build_enc_params(p, p->spk, cfg->spk, true);
    p->enc_param_off = p->param_floats;
    build_enc_params(p, p->enc, cfg->enc, false);
    DecNet& d = p->dec;
    d.c = dc;
    d.slope = dc.act == 1 ? AVC_LRELU_SLOPE : 0.f;
    d.n = dc.n_conv_blocks;
    p->dec_param_off = p->param_floats;
    d.in_conv = add_layer(p, dc.c_h, dc.c_in, 1, 1, true);
    for (int l = 0; l < d.n; ++l) d.c1.push_back(add_layer(p, dc.c_h, dc.c_h, dc.kernel_size, 1, true));
    for (int l = 0; l < d.n; ++l) d.c2.push_back(add_layer(p, dc.c_h * dc.upsample[l], dc.c_h, dc.kernel_size, 1, true));
    {
        int a0 = add_layer(p, 2 * dc.c_h, dc.c_cond, 1, 1, false);
        LayerP& L = p->layers[a0];
        L.nsrc = 2 * d.n;
        L.rows = 2 * dc.c_h;
        L.Cout = 2 * dc.c_h * L.nsrc;
        for (int i = 1; i < L.nsrc; ++i) {
            L.w[i] = add_param(p, 2 * dc.c_h, dc.c_cond, 0);
            L.b[i] = add_param(p, 2 * dc.c_h, 0, 0);
        }
        d.affine = a0;
    }
    d.out_conv = add_layer(p, dc.c_out, dc.c_h, 1, 1, true);

    // ---- time schedules + reflect-pad validity (reference raises the same way)
    auto sched_enc = [&](EncNet& e, int T0) -> int {
        e.T[0] = T0;
        int maxpad = e.c.bank_size / 2;
        if (maxpad >= T0) return -1;
        for (int l = 0; l < e.n; ++l) {
            if (e.c.kernel_size / 2 >= e.T[l]) return -1;
            e.T[l + 1] = avc_cdiv(e.T[l], e.c.subsample[l]);
        }
        return 0;
    };
    if (sched_enc(p->spk, T_cond) || sched_enc(p->enc, T)) {
        delete p;
        return fail(-6, "Padding size should be less than the corresponding input dimension");
    }
    p->Tb = p->enc.T[p->enc.n];
    d.T[0] = p->Tb;
    for (int l = 0; l < d.n; ++l) {
        if (dc.kernel_size / 2 >= d.T[l]) {
            delete p;
            return fail(-6, "Padding size should be less than the corresponding input dimension");
        }
        d.T[l + 1] = d.T[l] * dc.upsample[l];
    }
    p->Tout = d.T[d.n];
    if (p->bh) {
        // pair rows: two channels per dword, four frames per 16-byte access of the row kernels
        bool ok = !((cfg->enc.c_in | cfg->enc.c_h | cfg->enc.c_bank | cfg->enc.c_out | cfg->spk.c_h | cfg->spk.c_bank | dc.c_h | dc.c_in | dc.c_out) & 1);
        for (int l = 0; l <= p->spk.n; ++l) ok = ok && (p->spk.T[l] % 4 == 0);
        for (int l = 0; l <= p->enc.n; ++l) ok = ok && (p->enc.T[l] % 4 == 0);
        for (int l = 0; l <= d.n; ++l) ok = ok && (d.T[l] % 4 == 0) && (d.T[l] <= 2048);
        for (int l = 0; l < p->enc.n; ++l) ok = ok && (p->enc.T[l] <= 2048) && (p->spk.T[l] <= 2048);
        if (!ok) {
            delete p;
            return fail(AVC_ERR_PAIR_SHAPE, "avc_plan_create: AVC_PLAN_BF16S needs even channel counts and frame counts that are multiples of 4 (<= 2048) at every level");
        }
    }

    // ---- packed weights (inference plans keep no dgrad images; speaker-only plans only the speaker encoder's)
    const bool dg = !infer;
    for (EncNet* e : {&p->spk, &p->enc}) {
        if (spk_only && e == &p->enc) continue;
        for (int id : e->bank) finish_layer(p, p->layers[id], false, 0, B, e->T[0], e->T[0], e->nb);
        finish_layer(p, p->layers[e->in_conv], dg, e->CC - e->c.c_in, B, e->T[0], e->T[0]);
        for (int l = 0; l < e->n; ++l) {
            finish_layer(p, p->layers[e->c1[l]], dg, 0, B, e->T[l], e->T[l]);
            finish_layer(p, p->layers[e->c2[l]], dg, 0, B, e->T[l + 1], e->T[l]);
        }
    }
    for (int l = 0; l < p->spk.nd; ++l) {
        finish_layer(p, p->layers[p->spk.dn1[l]], dg, 0, 1, B, B, 1, false);
        finish_layer(p, p->layers[p->spk.dn2[l]], dg, 0, 1, B, B, 1, false);
    }
    finish_layer(p, p->layers[p->spk.outl], dg, 0, 1, B, B, 1, false);
    if (!spk_only) {
        finish_layer(p, p->layers[p->enc.heads], dg, 0, B, p->Tb, p->Tb);
        finish_layer(p, p->layers[d.in_conv], dg, 0, B, p->Tb, p->Tb);
        for (int l = 0; l < d.n; ++l) {
            finish_layer(p, p->layers[d.c1[l]], dg, 0, B, d.T[l], d.T[l]);
            finish_layer(p, p->layers[d.c2[l]], dg, 0, B, d.T[l], d.T[l]);
        }
        finish_layer(p, p->layers[d.affine], dg, 0, 1, B, B, 1, true, false);
        if (dg) {   // d(emb) = W^T dcond runs through the weight-gradient kernel, which reads W as a plain [Kp][Mp] matrix
            LayerP& La = p->layers[d.affine];
            La.wplain = p->alloc((long)La.nchunk_f * La.CK * La.Mp_f);
        }
        finish_layer(p, p->layers[d.out_conv], dg, 0, B, p->Tout, p->Tout);
    }

    // ---- activations.  Inference plans (AVC_PLAN_INFERENCE) keep only what the forward pass touches;
    // speaker-only plans (AVC_PLAN_SPEAKER_ONLY) only the speaker encoder's buffers.
    const long Bl = B;
    const long H = p->bh ? 2 : 1;   // a stored [B, C, T] activation takes C / H dword rows per sample (bf16 pairs: half the floats)
    auto alloc_enc = [&](EncNet& e, bool spk) {
        const long C = e.c.c_h / H;
        e.cat = p->alloc(Bl * (e.CC / H) * e.T[0]);
        if (!infer) e.dcat = p->alloc(Bl * (e.CC / H) * e.T[0]);
        e.h0 = p->alloc(Bl * C * e.T[0]);
        e.out[0] = spk ? e.h0 : p->alloc(Bl * C * e.T[0]);
        if (!spk) e.st0 = p->alloc(2 * Bl * e.c.c_h);
        for (int l = 0; l < e.n; ++l) {
            e.a1[l] = p->alloc(Bl * C * e.T[l]);
            e.out[l + 1] = p->alloc(Bl * C * e.T[l + 1]);
            if (spk) {
                e.a2[l] = p->alloc(Bl * C * e.T[l + 1]);
                e.y1[l] = e.y2[l] = -1;
            } else {
                e.a2[l] = -1;
                e.y1[l] = p->alloc(Bl * C * e.T[l]);
                e.y2[l] = p->alloc(Bl * C * e.T[l + 1]);
                e.st1[l] = p->alloc(2 * Bl * e.c.c_h);
                e.st2[l] = p->alloc(2 * Bl * e.c.c_h);
            }
        }
        if (spk) {   // (the dense stack is fp32)
            const long Cf = e.c.c_h;
            e.pooled = p->alloc(Cf * Bl);
            e.hd[0] = e.pooled;
            for (int l = 0; l < e.nd; ++l) {
                e.d1[l] = p->alloc(Cf * Bl);
                e.d2[l] = p->alloc(Cf * Bl);
                e.hd[l + 1] = p->alloc(Cf * Bl);
            }
        }
    };
    alloc_enc(p->spk, true);
    const long Cz = dc.c_in, Cd = dc.c_h;
    p->emb = p->alloc(Bl * dc.c_cond);
    if (!spk_only) {
        alloc_enc(p->enc, false);
        p->muls = p->alloc(Bl * 2 * Cz * p->Tb);
        d.z = p->alloc(Bl * (Cz / H) * p->Tb);
        d.cond = p->alloc(Bl * 2 * d.n * 2 * Cd);
        d.y0 = p->alloc(Bl * (Cd / H) * d.T[0]);
        d.out[0] = p->alloc(Bl * (Cd / H) * d.T[0]);
        d.st0 = p->alloc(2 * Bl * Cd);
        for (int l = 0; l < d.n; ++l) {
            d.y1[l] = p->alloc(Bl * (Cd / H) * d.T[l]);
            d.a1[l] = p->alloc(Bl * (Cd / H) * d.T[l]);
            d.y2[l] = p->alloc(Bl * (Cd / H) * d.T[l + 1]);
            d.out[l + 1] = p->alloc(Bl * (Cd / H) * d.T[l + 1]);
            d.st1[l] = p->alloc(2 * Bl * Cd);
            d.st2[l] = p->alloc(2 * Bl * Cd);
        }
        p->decb = p->alloc(Bl * p->M * p->Tout);
    }
    if (!infer) {
        p->dmuls = p->alloc(Bl * 2 * Cz * p->Tb);
        p->demb = p->alloc(Bl * dc.c_cond);
        p->dz = p->alloc(Bl * Cz * p->Tb);
        d.dcond = p->alloc(Bl * 2 * d.n * 2 * Cd);
        p->ddec = p->alloc(Bl * p->M * p->Tout);
        p->losses = p->alloc(64);
        p->loss_partial = p->alloc(2 * 1024);
        // ---- gradient temporaries (sized for the largest [B, C, T] they ever hold)
        long maxCT = 0;
        auto upd = [&](long c, long t) { maxCT = ((c / H) * t > maxCT) ? (c / H) * t : maxCT; };
        for (int l = 0; l <= p->spk.n; ++l) upd(p->spk.c.c_h, p->spk.T[l]);
        for (int l = 0; l <= p->enc.n; ++l) upd(p->enc.c.c_h, p->enc.T[l]);
        for (int l = 0; l <= d.n; ++l) upd(Cd, d.T[l]);
        upd(p->M, p->Tout);
        p->gA = p->alloc(Bl * maxCT);
        p->gB = p->alloc(Bl * maxCT);
        p->gC = p->alloc(Bl * maxCT);
        p->gA2 = p->alloc(Bl * maxCT);
        p->gB2 = p->alloc(Bl * maxCT);
        p->gC2 = p->alloc(Bl * maxCT);
        p->dhA = p->alloc((long)p->spk.c.c_h * Bl);
        if (p->bh) {
            p->ddecp = p->alloc(Bl * p->M * p->Tout / 2);
            p->dmulsp = p->alloc(Bl * Cz * p->Tb);
        }
    }

    p->named["emb"] = p->emb;
    p->named["spk_cat"] = p->spk.cat;
    p->named["spk_out0"] = p->spk.out[0];
    p->named["spk_pooled"] = p->spk.pooled;
    p->named["spk_outN"] = p->spk.out[p->spk.n];
    if (!spk_only) {
        p->named["muls"] = p->muls;
        p->named["dec"] = p->decb;
        p->named["z"] = d.z;
        p->named["cond"] = d.cond;
        p->named["enc_cat"] = p->enc.cat;
        p->named["enc_out0"] = p->enc.out[0];
        p->named["enc_outN"] = p->enc.out[p->enc.n];
        p->named["dec_out0"] = d.out[0];
        p->named["dec_outN"] = d.out[d.n];
    }
    if (!infer) {
        p->named["losses"] = p->losses;
        p->named["d_dec"] = p->ddec;
        p->named["d_z"] = p->dz;
        p->named["d_muls"] = p->dmuls;
        p->named["d_emb"] = p->demb;
        p->named["d_cond"] = d.dcond;
    }

    // ---- ReLU site table in the reference's forward call order (avc_plan_relu_site).  Pair plans: activation / conv-output tensors
    // are bf16 pair tensors (storage 1: strides in dwords) or, after a pixel-shuffling conv, natural bf16 rows (storage 2)
    {
        const long H = p->bh ? 2 : 1;   // channels per row of a stored [B, C, T] tensor
        auto conv_site = [&](long off, int Bn, int C, int T, long sb, long sc, long st, int storage) {
            avc_relu_site r;
            memset(&r, 0, sizeof(r));
            r.kind = 0; r.B = Bn; r.C = C; r.T = T; r.act_off = off; r.sb = sb; r.sc = sc; r.st = st;
            r.y_off = r.stat_off = r.cond_off = -1;
            r.storage = storage;
            p->sites.push_back(r);
        };
        auto in_site = [&](long y, long st, int C, int T, long cond, long csb, int storage) {
            avc_relu_site r;
            memset(&r, 0, sizeof(r));
            r.kind = 1; r.B = B; r.C = C; r.T = T; r.act_off = -1;
            r.y_off = y; r.stat_off = st; r.cond_off = cond; r.cond_sb = csb;
            r.storage = storage;
            p->sites.push_back(r);
        };
        const int PS = p->bh ? 1 : 0;
        const EncNet& sp = p->spk;
        const int Cs_ = sp.c.c_h;
        for (int g = 0; g < sp.nb; ++g) conv_site(sp.cat + (long)g * (sp.c.c_bank / H) * sp.T[0], B, sp.c.c_bank, sp.T[0], (long)(sp.CC / H) * sp.T[0], sp.T[0], 1, PS);
        conv_site(sp.h0, B, Cs_, sp.T[0], (long)(Cs_ / H) * sp.T[0], sp.T[0], 1, PS);
        for (int l = 0; l < sp.n; ++l) {
            conv_site(sp.a1[l], B, Cs_, sp.T[l], (long)(Cs_ / H) * sp.T[l], sp.T[l], 1, PS);
            conv_site(sp.a2[l], B, Cs_, sp.T[l + 1], (long)(Cs_ / H) * sp.T[l + 1], sp.T[l + 1], 1, PS);
        }
        for (int l = 0; l < sp.nd; ++l) {  // dense activations are stored channel-major [C][B] fp32; the reference sees [B, C]
            conv_site(sp.d1[l], B, Cs_, 1, 1, B, 0, 0);
            conv_site(sp.d2[l], B, Cs_, 1, 1, B, 0, 0);
        }
        const EncNet& en = p->enc;
        const int Ce_ = en.c.c_h;
        if (!spk_only) {
        for (int g = 0; g < en.nb; ++g) conv_site(en.cat + (long)g * (en.c.c_bank / H) * en.T[0], B, en.c.c_bank, en.T[0], (long)(en.CC / H) * en.T[0], en.T[0], 1, PS);
        in_site(en.h0, en.st0, Ce_, en.T[0], -1, 0, PS);
        for (int l = 0; l < en.n; ++l) {
            in_site(en.y1[l], en.st1[l], Ce_, en.T[l], -1, 0, PS);
            in_site(en.y2[l], en.st2[l], Ce_, en.T[l + 1], -1, 0, PS);
        }
        const long csb_ = (long)2 * d.n * 2 * Cd;
        in_site(d.y0, d.st0, (int)Cd, d.T[0], -1, 0, PS);
        for (int l = 0; l < d.n; ++l) {
            in_site(d.y1[l], d.st1[l], (int)Cd, d.T[l], d.cond + (long)(2 * l) * 2 * Cd, csb_, PS);
            in_site(d.y2[l], d.st2[l], (int)Cd, d.T[l + 1], d.cond + (long)(2 * l + 1) * 2 * Cd, csb_, (p->bh && dc.upsample[l] > 1) ? 2 : PS);
        }
        }
    }

    // ---- split-K slabs, dy arena and the event pool: sized with a dry run of the backward pass
    if (!infer) {
        p->slab = p->ws_top;
        long need[3] = {0, 0, 0};
        avc_backward_impl(p, nullptr, nullptr, 0, 0, 0, nullptr, 0, 0, 0, nullptr, nullptr, nullptr, nullptr, 0.f, nullptr,
                          nullptr, nullptr, true, need);
        p->slab_floats = need[0];
        p->ws_top += (need[0] + 63) / 64 * 64;
        p->dyarena = p->ws_top;
        p->dyarena_floats = need[1];
        p->ws_top += (need[1] + 63) / 64 * 64;
        p->nev_need = (int)need[2];
        p->named["wgrad_slab"] = p->slab;      // (diagnostic views: scripts/bf16_repro_probe3.py)
        p->named["dy_arena"] = p->dyarena;
        p->named["g_tmp"] = p->gA;
    }
    // helper streams / events belong to the plan from here on (created on the device that is current NOW;
    // a process without a GPU -- host-only plan queries -- simply gets a single-stream plan)
    plan_init_streams(p);
    *out = p;
    return 0;
}

extern "C" void avc_plan_destroy(avc_plan* p) {
    if (!p) return;
    if (p->pack_tab_dev) hipFree(p->pack_tab_dev);
    if (p->pack_blk_dev) hipFree(p->pack_blk_dev);
    // every handle that exists, whatever state plan_init_streams ended in
    // (the helper streams belong to the device's shared set: shared_streams())
    if (p->ev_fork) hipEventDestroy(p->ev_fork);
    if (p->ev_join) hipEventDestroy(p->ev_join);
    for (int i = 0; i < 2; ++i) {
        if (p->wjoin[i]) hipEventDestroy(p->wjoin[i]);
    }
    for (hipEvent_t e : p->wev)
        if (e) hipEventDestroy(e);
    if (p->ev_dense) hipEventDestroy(p->ev_dense);
    if (p->ev_dec_grads) hipEventDestroy(p->ev_dec_grads);
    if (p->ev_spk_grads) hipEventDestroy(p->ev_spk_grads);
    if (p->ev_all_grads) hipEventDestroy(p->ev_all_grads);
    delete p;
}

// fork/join helpers: `side` runs one independent branch while the caller's stream runs the other
extern "C" int avc_plan_set_single_stream(avc_plan* p, int on) {
    if (!p) return fail(-1, "avc_plan_set_single_stream: null plan");
    p->tun.single_stream = on ? 1 : 0;
    return 0;
}

extern "C" int avc_plan_set_compute_dtype(avc_plan* p, int dtype) {
    if (!p || (dtype != AVC_COMPUTE_F32 && dtype != AVC_COMPUTE_BF16)) return fail(-1, "avc_plan_set_compute_dtype: dtype must be 0 (fp32) or 1 (bf16 operands)");
    if (p->bh) return fail(-1, "avc_plan_set_compute_dtype: the plan was created with AVC_PLAN_BF16S (bf16 storage is a property of the workspace layout)");
    p->compute = dtype;
    for (LayerP& L : p->layers) L.bf16 = dtype;
    return 0;
}
extern "C" int avc_plan_compute_dtype(const avc_plan* p) { return p ? p->compute : -1; }

static void pack_layer(const avc_plan* p, const LayerP& L, const float* params, float* ws, std::vector<PackArgs>& out);
static void plan_init_pack_table(avc_plan* p) {
    // descriptors against NULL bases: every pointer field then holds a byte offset
    // The conv banks and in_convs open both encoder branches and run for ~1 ms: their images come first (a short launch on the caller's
    // stream); everything else can be packed on a forward-idle helper stream UNDER those convolutions.
    std::vector<char> early(p->layers.size(), 0);
    if (!(p->flags & AVC_PLAN_RAGGED))
        for (const EncNet* e : {&p->spk, &p->enc}) {
            if (e->in_conv < 0) continue;
            for (int g = 0; g < e->nb; ++g) early[e->bank[g]] = 1;
            early[e->in_conv] = 1;
        }
    std::vector<PackArgs> all;
    for (int pass = 0; pass < 2; ++pass) {
        for (size_t i = 0; i < p->layers.size(); ++i)
            if (p->layers[i].wpf >= 0 && (early[i] != 0) == (pass == 0)) pack_layer(p, p->layers[i], (const float*)nullptr, (float*)nullptr, all);
        if (pass == 0) p->pack_early_imgs = (int)all.size();
    }
    if (all.empty()) return;
    double bytes = 0;
    std::vector<int> blk;   // (image, piece) pairs
    for (size_t i = 0; i < all.size(); ++i) {
        if ((int)i == p->pack_early_imgs) p->pack_nblk_early = (int)(blk.size() / 2);
        all[i].mb = avc_pack_stage_rows(all[i]);
        bytes += 8.0 * avc_pack_total(all[i]);
        const long np = avc_pack_pieces(all[i]);
        for (long q = 0; q < np; ++q) {
            blk.push_back((int)i);
            blk.push_back((int)q);
        }
    }
    if (p->pack_early_imgs >= (int)all.size()) p->pack_nblk_early = (int)(blk.size() / 2);
    PackArgs* dev = nullptr;
    void* dblk = nullptr;
    if (hipMalloc((void**)&dev, all.size() * sizeof(PackArgs)) != hipSuccess || hipMalloc(&dblk, blk.size() * sizeof(int)) != hipSuccess) {
        (void)hipGetLastError();   // no device (host-only plan queries): forward packs with descriptor batches in kernel arguments instead
        if (dev) hipFree(dev);
        return;
    }
    if (hipMemcpy(dev, all.data(), all.size() * sizeof(PackArgs), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(dblk, blk.data(), blk.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) {
        (void)hipGetLastError();
        hipFree(dev);
        hipFree(dblk);
        return;
    }
    p->pack_tab_dev = dev;
    p->pack_blk_dev = dblk;
    p->pack_nblk = (int)(blk.size() / 2);
    p->pack_tab_bytes = bytes;
}

// The three helper streams of a plan -- the side branch's and the two weight-gradient streams -- are ONE set per device, shared by every plan
// of the process and kept for its lifetime (round 5).  The runtime deals streams to a handful of hardware queues in creation order: the
// first plan's three streams and the caller's stream take four of them, a second plan's OWN three land wherever the round-robin has got
// to -- its side stream on the caller's queue, for one: both branches of a pass then run one after the other (bf16 step 4.9 - 6.1 instead of
// 2.59 ms, measured: scripts/two_plans_probe.py).  All three are created together, in a fixed order, by the FIRST plan of the device; the side
// stream's priority is that plan's `side_prio` and stays: a later plan that asks for the other priority gets the existing stream (round 6: a
// fourth helper stream created later put the process back into the queue lottery -- 8.2 instead of 5.9 ms per step,
// profiles/r05_two_plans_probe.log -- so there is none; avc_plan_side_priority reports what a plan really runs with).
// Consequences for callers (include/avc_hip.h says the same): the helper streams are process-lifetime objects; two plans driven from two
// host threads share them, i.e. their side-branch / weight-gradient work is serialised stream by stream (every wait is on an event recorded
// earlier: ordering, never a hazard).  Immutable once created; the lookup is mutex-protected.
#ifndef AVC_EMU
#include <mutex>
#endif
struct DevStreams {
    hipStream_t side = nullptr;
    hipStream_t w[2] = {nullptr, nullptr};
    int side_prio = -1;   // priority class the side stream was created with (0 normal, 1 highest); -1: not created
};
static bool shared_streams(int side_prio, hipStream_t* side, hipStream_t* w0, hipStream_t* w1, int* prio_used) {
#ifdef AVC_EMU
    static DevStreams pool[1];
    const int dev = 0;
#else
    static std::mutex mu;
    static DevStreams pool[64];
    std::lock_guard<std::mutex> lk(mu);
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) {
        (void)hipGetLastError();
        return false;
    }
#endif
    DevStreams& d = pool[dev];
    if (d.side_prio < 0) {
        int lo = 0, hi = 0;
        hipDeviceGetStreamPriorityRange(&lo, &hi);  // lo = least urgent, hi = most urgent
        const int sp = side_prio ? 1 : 0;
        // the side stream carries the speaker-encoder branch: the LONGER pole of both passes (forward: pooling + the latency-bound dense stack
        // after its convs, before the decoder can start; backward: d_emb -> dense stack -> its whole dgrad chain -> the last weight gradients).
        // side_prio = 1 dispatches its workgroups ahead of the content branch's and the weight-gradient streams'.
        hipStream_t s_ = nullptr, w_[2] = {nullptr, nullptr};
        bool ok = (sp ? hipStreamCreateWithPriority(&s_, hipStreamNonBlocking, hi) : hipStreamCreateWithFlags(&s_, hipStreamNonBlocking)) == hipSuccess;
        for (int i = 0; i < 2 && ok; ++i) ok = hipStreamCreateWithPriority(&w_[i], hipStreamNonBlocking, lo) == hipSuccess;
        if (!ok) {
            (void)hipGetLastError();  // no device: single-stream plan
            if (s_) hipStreamDestroy(s_);
            for (int i = 0; i < 2; ++i)
                if (w_[i]) hipStreamDestroy(w_[i]);
            return false;
        }
        d.side = s_; d.w[0] = w_[0]; d.w[1] = w_[1];
        d.side_prio = sp;
    }
    *side = d.side; *w0 = d.w[0]; *w1 = d.w[1];
    *prio_used = d.side_prio;
    return true;
}

static void plan_init_streams(avc_plan* p) {
    plan_init_pack_table(p);
    p->side_state = -1;
    int prio_used = p->tun.side_prio ? 1 : 0;
    if (!shared_streams(p->tun.side_prio, &p->side, &p->wstream[0], &p->wstream[1], &prio_used)) {
        p->side = nullptr;
        p->wstream[0] = p->wstream[1] = nullptr;
        return;
    }
    bool ok = hipEventCreateWithFlags(&p->ev_fork, hipEventDisableTiming) == hipSuccess &&
              hipEventCreateWithFlags(&p->ev_join, hipEventDisableTiming) == hipSuccess &&
              hipEventCreateWithFlags(&p->ev_dense, hipEventDisableTiming) == hipSuccess &&
              hipEventCreateWithFlags(&p->ev_dec_grads, hipEventDisableTiming) == hipSuccess &&
              hipEventCreateWithFlags(&p->ev_spk_grads, hipEventDisableTiming) == hipSuccess &&
              hipEventCreateWithFlags(&p->ev_all_grads, hipEventDisableTiming) == hipSuccess;
    for (int i = 0; i < 2; ++i) {
        ok = ok && hipEventCreateWithFlags(&p->wjoin[i], hipEventDisableTiming) == hipSuccess;
    }
    p->wev.assign((size_t)p->nev_need, nullptr);
    for (hipEvent_t& e : p->wev) ok = ok && hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess;
    p->side_state = ok ? 1 : -1;
    p->tun.side_prio = prio_used;   // (what the device's shared side stream really has)
}
static bool side_ready(const avc_plan* p) { return !p->tun.single_stream && p->side_state == 1; }
static hipStream_t fork_side(const avc_plan* p, hipStream_t mainS) {
    if (!side_ready(p)) return mainS;
    hipEventRecord(p->ev_fork, mainS);
    hipStreamWaitEvent(p->side, p->ev_fork, 0);
    return p->side;
}
static void join_side(const avc_plan* p, hipStream_t mainS, hipStream_t sideS) {
    if (sideS == mainS) return;
    hipEventRecord(p->ev_join, sideS);
    hipStreamWaitEvent(mainS, p->ev_join, 0);
}
extern "C" int avc_plan_side_priority(const avc_plan* p) { return p->side_state == 1 ? p->tun.side_prio : -1; }
extern "C" int avc_plan_num_params(const avc_plan* p) { return (int)p->params.size(); }
extern "C" long avc_plan_param_floats(const avc_plan* p) { return p->param_floats; }
extern "C" int avc_plan_param_info(const avc_plan* p, int i, long* offset, long* numel, int dims[3]) {
    if (i < 0 || i >= (int)p->params.size()) return -1;
    *offset = p->params[i].off;
    *numel = p->params[i].numel;
    for (int k = 0; k < 3; ++k) dims[k] = p->params[i].d[k];
    return 0;
}
extern "C" long avc_plan_workspace_floats(const avc_plan* p) { return p->ws_top; }
extern "C" long avc_plan_buffer(const avc_plan* p, const char* name) {
    auto it = p->named.find(name);
    return it == p->named.end() ? -1 : it->second;
}
extern "C" int avc_plan_num_relu_sites(const avc_plan* p) { return (int)p->sites.size(); }
extern "C" int avc_plan_relu_site(const avc_plan* p, int i, avc_relu_site* out) {
    if (i < 0 || i >= (int)p->sites.size()) return -1;
    *out = p->sites[i];
    return 0;
}
extern "C" int avc_plan_flags(const avc_plan* p) { return p ? p->flags : -1; }
// Data-parallel hook: the decoder's parameters are the tail of the flat buffer and their
// gradients are final well before the encoders' (the backward pass walks decoder -> encoders).
extern "C" int avc_plan_param_range(const avc_plan* p, int part, long* offset, long* numel) {
    if (!p || !offset || !numel) return -1;
    if (part == AVC_GRADS_DECODER) {
        *offset = p->dec_param_off;
        *numel = p->param_floats - p->dec_param_off;
    } else if (part == AVC_GRADS_ENCODERS) {
        *offset = 0;
        *numel = p->dec_param_off;
    } else if (part == AVC_GRADS_SPEAKER) {
        *offset = 0;
        *numel = p->enc_param_off;
    } else if (part == AVC_GRADS_CONTENT) {
        *offset = p->enc_param_off;
        *numel = p->dec_param_off - p->enc_param_off;
    } else if (part == AVC_GRADS_ALL) {
        *offset = 0;
        *numel = p->param_floats;
    } else {
        return -1;
    }
    return 0;
}
// make `stream` wait until the gradients of `part` written by the LAST avc_backward call on this plan are final
// (a no-op before the first call).  -9: the plan could not create helper streams / events -- nothing was ordered, the
// caller must make `stream` wait for the stream avc_backward ran on.
extern "C" int avc_plan_stream_wait_grads(const avc_plan* p, int part, void* stream) {
    if (!p || (p->flags & AVC_PLAN_INFERENCE)) return fail(-1, "avc_plan_stream_wait_grads: not a training plan");
    if (p->side_state != 1) return fail(-9, "avc_plan_stream_wait_grads: the plan has no helper streams / events (order on the stream avc_backward ran on)");
    hipEvent_t e = (part == AVC_GRADS_DECODER) ? p->ev_dec_grads : (part == AVC_GRADS_SPEAKER ? p->ev_spk_grads : p->ev_all_grads);
    return (int)hipStreamWaitEvent((hipStream_t)stream, e, 0);
}
extern "C" int avc_plan_out_len(const avc_plan* p) { return p->Tout; }
extern "C" int avc_plan_latent_len(const avc_plan* p) { return p->Tb; }

// --------------------------------------------------------------------------
// launch helpers
// --------------------------------------------------------------------------
static void set_group(ConvGroup& g, const float* wp, const float* bias, int KS, int CK, int nchunk) {
    memset(&g, 0, sizeof(g));
    g.wp = wp;
    g.bias = bias;
    g.KS = KS;
    g.padL = KS / 2;
    g.padR = (KS % 2 == 0) ? KS / 2 - 1 : KS / 2;
    g.CK = CK;
    g.nchunk = nchunk;
}

static const float* layer_bias(const avc_plan* p, const LayerP& L, const float* params, const float* ws) {
    return L.nsrc == 1 ? p->par(params, L.b[0]) : ws + L.bpk;
}

// forward conv of layer L on a source view; caller fills epilogue extras afterwards
static ConvArgs mk_fwd(const avc_plan* p, float slope, const LayerP& L, const float* params, const float* ws, const float* x, long sb,
                       long sc, int st, int Bn, int Tsrc, float* out, long ob, long oc, int ot, int act) {
    ConvArgs a;
    memset(&a, 0, sizeof(a));
    a.x.ptr = x; a.x.sb = sb; a.x.sc = sc; a.x.st = st; a.x.ps = 1;
    a.B = Bn; a.Cred = L.bh ? L.Cin / 2 : L.Cin; a.Tsrc = Tsrc;
    a.mode = 0; a.stride = L.stride; a.bf16 = L.bf16;
    a.pairs = L.bh ? 1 : 0;   // (callers clear it for the fp32 outputs of the heads and the decoder's last conv)
    a.M = L.Cout; a.Mp = L.Mp_f;
    a.ngroups = 1;
    set_group(a.g[0], ws + (L.x3_f ? L.wrs_f : L.wpf), layer_bias(p, L, params, ws), L.KS, L.CK, L.nchunk_f);
    a.img = L.x3_f ? AVC_IMG_X3 : (L.bh ? AVC_IMG_K4H : AVC_IMG_K4);
    a.Tout = (Tsrc + a.g[0].padL + a.g[0].padR - L.KS) / L.stride + 1;
    a.ob = ob; a.oc = oc; a.ot = ot; a.ops = 1;
    a.act = act;
    a.slope = slope;
    a.g[0].out = out;
    return a;
}

static ConvArgs mk_dgrad(float slope, const LayerP& L, const float* ws, const float* dy, long sb, long sc, int st, int ps, int Bn,
                         int Tdy, int Tin, float* dx, long ob, long oc, int ot) {
    ConvArgs a;
    memset(&a, 0, sizeof(a));
    a.x.ptr = dy; a.x.sb = sb; a.x.sc = sc; a.x.st = st; a.x.ps = ps;
    a.B = Bn; a.Cred = L.bh ? L.Cout / 2 : L.Cout; a.Tsrc = Tdy;
    a.pairs = L.bh ? 1 : 0;
    a.mode = 1; a.stride = L.stride; a.mirror = (L.KS > 1) ? 1 : 0; a.bf16 = L.bf16;
    a.M = L.dgM; a.Mp = L.Mp_d; a.Tout = Tin;
    a.ob = ob; a.oc = oc; a.ot = ot; a.ops = 1;
    a.res_to_primary = 1;
    a.slope = slope;   // (applied where the launch masks by a ReLU output)
    a.ngroups = 1;
    set_group(a.g[0], ws + (L.x3_d ? L.wrs_d : L.wpd), nullptr, L.KS, L.CKd, L.nchunk_d);
    a.img = L.x3_d ? AVC_IMG_X3 : (L.bh ? AVC_IMG_K4H : AVC_IMG_K4);
    a.g[0].out = dx;
    return a;
}

static void set_res(ConvArgs& a, const float* res, int mode, long rb, long rc, int rt, int Tres) {
    a.g[0].res = res;
    a.res_mode = mode;
    a.rb = rb; a.rc = rc; a.rt = rt; a.Tres = Tres;
}

struct BwdCtx {
    const avc_plan* p;
    const float* params;
    float* grads;
    float* ws;
    hipStream_t s;
    bool dry;
    long slab_used;
    hipStream_t wstream;   // stream of the weight-gradient kernels of the current branch (== s when not overlapping)
    int nev;
    long dy_used;
    // weight gradients recorded but not launched yet: they run as batched launches (conv_wgrad.hip) once
    // p->wgrad_batch layers are pending or a branch ends -- few large, balanced launches with short split-K
    // instead of one launch per layer
    struct PendW {
        WgradArgs a;
        const LayerP* L;
    };
    std::vector<PendW> pend;
    bool hold = false;     // keep recording (no automatic flush): the decoder's layers go out together, behind the dense-stack kernel
    hipStream_t s_extra = nullptr;   // a second producer stream the next flush must also be ordered behind (the decoder's other half-batch chain)
    int target_wgs = 0;              // workgroups per launch of the next flush (0 = avc_tuning.wgrad_batch_wgs)
    long pend_units = 0;   // (co, ci) tiles x K-chunks of the pending layers
    // every gradient tensor a (possibly still running) wgrad kernel reads gets its own buffer
    float* fresh(long n) {
        long off = dy_used;
        dy_used += (n + 63) / 64 * 64;
        return dry ? nullptr : ws + p->dyarena + off;
    }
};

// One ordering edge "everything queued on c.s so far -> the branch's wgrad stream"; returns the stream
// to launch on.  Edges are counted in the dry run too: that count sizes the plan's event pool, so the
// index is always valid at run time.
static hipStream_t wgrad_edge(BwdCtx& c, bool two_producers = false) {
    const int i = c.nev++;
    const int i2 = two_producers ? c.nev++ : -1;   // (counted in the dry run whether or not the second stream exists at run time)
    if (c.dry || c.wstream == c.s) return c.s;
    hipEvent_t e = c.p->wev[i];
    hipEventRecord(e, c.s);
    hipStreamWaitEvent(c.wstream, e, 0);
    if (two_producers && c.s_extra && c.s_extra != c.s) {
        hipEvent_t e2 = c.p->wev[i2];
        hipEventRecord(e2, c.s_extra);
        hipStreamWaitEvent(c.wstream, e2, 0);
    }
    return c.wstream;
}

// Launch the pending weight gradients of this branch on the branch's wgrad stream (ordered behind everything queued on c.s so far:
// their dy operands are final): one stream-K launch per kernel instance present (conv_wgrad.hip) + ONE reduce launch that sums the
// partial tiles in a fixed order into the flat gradient buffer (weights and biases).  Runs beside the dgrad / InstanceNorm-backward chain.
static int flush_wgrads(BwdCtx& c, bool two_producers = false) {
    if (c.pend.empty()) return 0;
    hipStream_t ls = wgrad_edge(c, two_producers);
    const int n = (int)c.pend.size();
    std::vector<WgradArgs> L((size_t)n);
    for (int i = 0; i < n; ++i) L[i] = c.pend[i].a;
    avc_wgrad_plan_batch(L.data(), n, c.target_wgs > 0 ? c.target_wgs : c.p->tun.wgrad_batch_wgs);
    for (int i = 0; i < n; ++i) {
        WgradArgs& a = L[i];
        const long off = c.slab_used;
        c.slab_used += (a.slab_need + a.dbslab_need + 63) / 64 * 64;
        if (c.dry) continue;
        a.slab = c.ws + c.p->slab + off;
        a.dbslab = a.slab + a.slab_need;
        const LayerP& Lp = *c.pend[i].L;
        a.rows_per_src = Lp.rows;
        a.dw = c.grads + c.p->params[Lp.w[0]].off;
        a.db = c.grads + c.p->params[Lp.b[0]].off;
        a.dw_src_stride = Lp.nsrc > 1 ? c.p->params[Lp.w[1]].off - c.p->params[Lp.w[0]].off : 0;
        a.db_src_stride = Lp.nsrc > 1 ? c.p->params[Lp.b[1]].off - c.p->params[Lp.b[0]].off : 0;
    }
    c.pend.clear();
    c.pend_units = 0;
    if (c.dry) return 0;
    return avc_launch_wgrad_batch(L.data(), n, ls, c.p->tun.wgrad_ablation);
}

// weight + bias gradient of layer L: x = forward input view, dy = output-gradient view (recorded; see flush_wgrads)
static int wgrad_layer(BwdCtx& c, const LayerP& L, const float* x, long xsb, long xsc, int xst, const float* dy, long ysb,
                       long ysc, int yst, int yps, int Bn, int Tin, int Tout) {
    BwdCtx::PendW pw;
    WgradArgs& a = pw.a;
    memset(&a, 0, sizeof(a));
    a.x.ptr = x; a.x.sb = xsb; a.x.sc = xsc; a.x.st = xst; a.x.ps = 1;
    a.dy.ptr = dy; a.dy.sb = ysb; a.dy.sc = ysc; a.dy.st = yst; a.dy.ps = yps;
    a.B = Bn; a.Cin = L.Cin; a.Cout = L.Cout; a.Tin = Tin; a.Tout = Tout;
    a.KS = L.KS; a.padL = L.KS / 2; a.stride = L.stride;
    a.bf16 = (L.bf16 == AVC_COMPUTE_F32 && c.p->tun.wgrad_x3) ? AVC_COMPUTE_F32X3 : L.bf16;
    a.cw8 = c.p->tun.wgrad_cw8 ? 1 : 0;
    pw.L = &L;
    avc_wgrad_geometry(a);
    c.pend_units += (long)a.tiles * a.total_chunks;
    c.pend.push_back(pw);
    // flush when the batch is worth a launch: enough work to give every CU a long K run, or enough layers
    if (!c.hold && ((int)c.pend.size() >= c.p->tun.wgrad_batch || c.pend_units >= c.p->tun.wgrad_batch_units)) return flush_wgrads(c);
    return 0;
}

static void pack_layer(const avc_plan* p, const LayerP& L, const float* params, float* ws, std::vector<PackArgs>& out) {
    PackArgs a;
    memset(&a, 0, sizeof(a));
    for (int i = 0; i < L.nsrc; ++i) a.src[i] = p->par(params, L.w[i]);
    a.nsrc = L.nsrc; a.rows_per_src = L.rows;
    a.Cout = L.Cout; a.Cin = L.Cin; a.KS = L.KS;
    a.dgrad = 0; a.CK = L.CK; a.nchunk = L.nchunk_f; a.M = L.Cout; a.Mp = L.Mp_f;
    a.dst = ws + L.wpf;
    a.img = L.conv_path ? (L.bh ? AVC_IMG_K4H : AVC_IMG_K4) : AVC_IMG_PLAIN;
    if (L.x3_f) {   // only the image the launch will read
        PackArgs r;
        avc_pack_x3_args(r, p->par(params, L.w[0]), L.Cout, L.Cin, L.KS, 0, ws + L.wrs_f);
        out.push_back(r);
    } else {
        out.push_back(a);
    }
    if (L.wplain >= 0) {
        PackArgs b = a;
        b.img = AVC_IMG_PLAIN;
        b.dst = ws + L.wplain;
        out.push_back(b);
    }
    if (L.need_dgrad && L.x3_d) {
        PackArgs r;
        avc_pack_x3_args(r, p->par(params, L.w[0]), L.Cout, L.Cin, L.KS, 1, ws + L.wrs_d, L.dgM);
        out.push_back(r);
    } else if (L.need_dgrad) {
        a.dgrad = 1; a.CK = L.CKd; a.nchunk = L.nchunk_d; a.M = L.dgM; a.Mp = L.Mp_d;
        a.dst = ws + L.wpd;
        out.push_back(a);
    }
    if (L.nsrc > 1) {  // stacked bias = "weight" with Cin = 1, KS = 1: first Mp floats of the plain image
        PackArgs b;
        memset(&b, 0, sizeof(b));
        for (int i = 0; i < L.nsrc; ++i) b.src[i] = p->par(params, L.b[i]);
        b.nsrc = L.nsrc; b.rows_per_src = L.rows;
        b.Cout = L.Cout; b.Cin = 1; b.KS = 1;
        b.dgrad = 0; b.CK = 32; b.nchunk = 1; b.M = L.Cout; b.Mp = L.Mp_f;
        b.dst = ws + L.bpk;
        b.img = AVC_IMG_PLAIN;
        out.push_back(b);
    }
}

static int in_fwd(float slope, const float* y, int Bn, int C, int T, const float* cond, long cond_sb, int cond_off, const float* res,
                  int res_mode, int Tres, float* out, float* stats, hipStream_t s, int Bfull = 0, int b0 = 0, bool pairs = false, int planar = 0, int nv = 0) {
    // stats = [mean[Bfull*C] | rstd[Bfull*C]]; a sub-batch launch (b0, Bn) of a Bfull-sample tensor passes
    // y/out/res/cond already offset to sample b0
    if (Bfull == 0) Bfull = Bn;
    INFwdArgs a;
    a.y = y; a.out = out; a.mean = stats + (long)b0 * C; a.rstd = stats + (long)Bfull * C + (long)b0 * C;
    a.cond = cond; a.cond_sb = cond_sb; a.cond_off = cond_off;
    a.res = res; a.res_mode = res ? res_mode : 0; a.Tres = Tres;
    a.R = Bn * C; a.C = C; a.T = T; a.relu = 1; a.slope = slope; a.planar = planar; a.nv_hint = nv;
    if (pairs) {
        a.R = Bn * (C / 2);
        return avc_launch_in_fwd_pairs(a, s);
    }
    return avc_launch_in_fwd(a, s);
}

// conv + the InstanceNorm rows of its output: inside the conv's epilogue where a 64-column tile holds whole rows (avc_conv_in_fusable:
// rows of 16 / 32 / 64 frames, exact fp32), else the row kernel behind the conv.  a.g[0].out = y (pre-norm, read by the backward pass).
static int conv_in_fwd(const avc_plan* p, ConvArgs& a, float slope, int Bn, int C, int T, const float* cond, long cond_sb, int cond_off,
                       const float* res, int res_mode, int Tres, float* out, float* stats, hipStream_t s, int Bfull = 0, int b0 = 0,
                       bool pairs = false, int planar = 0, int nv = 0) {
    if (Bfull == 0) Bfull = Bn;
    if (!planar && avc_conv_in_fusable(a, p->tun, res ? res_mode : AVC_RES_NONE, Tres)) {   // (pair tensors too; a pixel-shuffling conv's "planar" rows keep the row kernel)
        a.in.out = out;
        a.in.mean = stats + (long)b0 * C;
        a.in.rstd = stats + (long)Bfull * C + (long)b0 * C;
        a.in.cond = cond; a.in.cond_sb = cond_sb; a.in.cond_off = cond_off;
        a.in.res = res; a.in.res_mode = res ? res_mode : 0; a.in.Tres = Tres;
        a.in.C = C; a.in.relu = 1;
        return avc_launch_conv(a, s, 0, p->tun);
    }
    int rc = avc_launch_conv(a, s, 0, p->tun);
    if (rc) return rc;
    return in_fwd(slope, a.g[0].out, Bn, C, T, cond, cond_sb, cond_off, res, res_mode, Tres, out, stats, s, Bfull, b0, pairs, planar, nv);
}

static int in_bwd(float slope, const float* g, const float* y, const float* stats, int Bn, int C, int T, const float* cond,
                  long cond_sb, int cond_off, float* dy, float* dcond, hipStream_t s, bool pairs = false, int nv = 0) {
    INBwdArgs a;
    a.g = g; a.y = y; a.mean = stats; a.rstd = stats + (long)Bn * C;
    a.cond = cond; a.cond_sb = cond_sb; a.cond_off = cond_off;
    a.dy = dy; a.dcond = dcond; a.dcond_sb = cond_sb; a.dcond_off = cond_off;
    a.R = Bn * C; a.C = C; a.T = T; a.relu = 1; a.slope = slope; a.planar = 0; a.nv_hint = nv;
    if (pairs) {
        a.R = Bn * (C / 2);
        return avc_launch_in_bwd_pairs(a, s);
    }
    return avc_launch_in_bwd(a, s);
}

// --------------------------------------------------------------------------
// forward
// --------------------------------------------------------------------------
static int enc_front(const avc_plan* p, const EncNet& e, const float* params, float* ws, const float* x, long sxb,
                     long sxc, int sxt, hipStream_t s) {
    // conv_bank (model.py:85-91): all bank members in ONE grouped launch writing the concat buffer in place
    const int B = p->B, T0 = e.T[0];
    const float SL = e.slope;
    const bool bh = p->bh;
    const long CCr = bh ? e.CC / 2 : e.CC, Cbr = bh ? e.c.c_bank / 2 : e.c.c_bank;   // rows per sample: pair rows with bh
    float* tail = ws + e.cat + (long)e.nb * Cbr * T0;   // raw input last (model.py:90)
    if (bh) RUN(avc_launch_to_pairs(x, sxb, sxc, sxt, B, e.c.c_in, T0, tail, CCr * T0, T0, s));   // ... as bf16 pairs, and the bank reads THAT copy
    ConvArgs a;
    memset(&a, 0, sizeof(a));
    a.x.ptr = x; a.x.sb = sxb; a.x.sc = sxc; a.x.st = sxt; a.x.ps = 1;
    if (bh) { a.x.ptr = tail; a.x.sb = CCr * T0; a.x.sc = T0; a.x.st = 1; }
    a.B = B; a.Cred = bh ? e.c.c_in / 2 : e.c.c_in; a.Tsrc = T0;
    a.mode = 0; a.stride = 1; a.bf16 = p->compute;
    a.M = e.c.c_bank; a.Mp = avc_cdiv(e.c.c_bank, 128) * 128; a.Tout = T0;
    a.ob = CCr * T0; a.oc = T0; a.ot = 1; a.ops = 1;
    a.act = 1;
    a.slope = SL;
    a.ngroups = e.nb;
    a.img = bh ? AVC_IMG_K4H : AVC_IMG_K4;
    a.pairs = bh ? 1 : 0;
    for (int g = 0; g < e.nb; ++g) {
        const LayerP& L = p->layers[e.bank[g]];
        set_group(a.g[g], ws + L.wpf, p->par(params, L.b[0]), L.KS, L.CK, L.nchunk_f);
        a.g[g].out = ws + e.cat + (long)g * Cbr * T0;
    }
    RUN(avc_launch_conv(a, s, 0, p->tun));
    if (!bh) RUN(avc_launch_copy_rows(x, sxb, sxc, sxt, B, e.c.c_in, T0, tail, (long)e.CC * T0, T0, s));
    return 0;
}

static bool side_ready(const avc_plan* p);
static int pack_all(const avc_plan* p, const float* params, float* ws, hipStream_t s) {
    if (p->pack_tab_dev && !(p->tun.dbg_streams & 16)) {   // (bit 16, diagnostic: the per-image gather kernels the table launch is tested against)
        // ONE launch on the caller's stream.  (Round 4 ran the tail of the table on a helper stream under the next step's bank convs:
        // measured neutral -- 6.000 vs 6.009 ms/step -- and its read of `params` was ordered only against a later forward of the SAME
        // plan, a formal read / write race with an optimizer step or a parameter write that follows on another plan; removed.)
        return avc_launch_pack_table(p->pack_tab_dev, p->pack_blk_dev, p->pack_nblk, p->pack_tab_bytes, params, ws, s);
    }
    std::vector<PackArgs> all;
    for (size_t i = 0; i < p->layers.size(); ++i)
        if (p->layers[i].wpf >= 0) pack_layer(p, p->layers[i], params, ws, all);
    return avc_launch_pack_batch(all.data(), (int)all.size(), s);
}

// Weights -> LDS-image order for every layer of the plan, ONE launch.  A training loop calls this right behind the optimizer step
// (Solver.ae_step) and passes AVC_FWD_WEIGHTS_PACKED to the next avc_forward_ex: the step then opens with its first convolution.
extern "C" int avc_plan_pack_weights(const avc_plan* p, const float* params, float* ws, void* stream) {
    if (!p || !params || !ws) return fail(-1, "avc_plan_pack_weights: null argument");
    RUN(pack_all(p, params, ws, (hipStream_t)stream));
    return 0;
}

static int forward_impl(const avc_plan* p, const float* params, const float* x, long sxb, long sxc, int sxt,
                        const float* xc, long scb, long scc, int sct, const float* eps, float* ws, hipStream_t s, bool packed = false) {
    const int B = p->B;
    const bool bh = p->bh;
    const int NV = (int)p->tun.in_pairs_nv;
    // 0. weights -> LDS-image order (they change every optimizer step): one launch, unless the caller packed them behind its
    // optimizer step already (AVC_FWD_WEIGHTS_PACKED)
    if (!packed) RUN(pack_all(p, params, ws, s));

    const bool spk_only = (p->flags & AVC_PLAN_SPEAKER_ONLY) != 0;  // AE.get_speaker_embeddings (model.py:393-395)
    // first kernels of the content encoder (conv bank, in_conv, InstanceNorm) on the caller's stream
    auto content_front = [&]() -> int {
        if (spk_only) return 0;
        const EncNet& e = p->enc;
        const float SL = e.slope;
        const int Cc = e.c.c_h;
        const long C = bh ? Cc / 2 : Cc, CCr = bh ? e.CC / 2 : e.CC;
        RUN(enc_front(p, e, params, ws, x, sxb, sxc, sxt, s));
        ConvArgs a = mk_fwd(p, SL, p->layers[e.in_conv], params, ws, ws + e.cat, CCr * e.T[0], e.T[0], 1, B, e.T[0], ws + e.h0, (long)C * e.T[0], e.T[0], 1, 0);
        RUN(conv_in_fwd(p, a, SL, B, Cc, e.T[0], nullptr, 0, 0, nullptr, 0, 0, ws + e.out[0], ws + e.st0, s, 0, 0, bh, 0, NV));
        return 0;
    };
    // ---------------- speaker encoder (model.py:265-277), concurrent with the content encoder
    const hipStream_t mainS = s;
    const hipStream_t sideS = fork_side(p, mainS);
    {
        const hipStream_t s = sideS;
        const EncNet& e = p->spk;
        const float SL = e.slope;
        const int Cc = e.c.c_h;                       // channels
        const long C = bh ? Cc / 2 : Cc;              // rows per sample of a [B, c_h, T] tensor (pair rows with bh): every stride below
        const long CCr = bh ? e.CC / 2 : e.CC;
        RUN(enc_front(p, e, params, ws, xc, scb, scc, sct, s));
        RUN(content_front());   // (main stream) issued under this branch's long bank kernel: the content encoder is the longer branch
        {
            const LayerP& L = p->layers[e.in_conv];
            ConvArgs a = mk_fwd(p, SL, L, params, ws, ws + e.cat, CCr * e.T[0], e.T[0], 1, B, e.T[0], ws + e.h0, (long)C * e.T[0], e.T[0], 1, 1);
            RUN(avc_launch_conv(a, s, 0, p->tun));
        }
        for (int l = 0; l < e.n; ++l) {
            const int Ti = e.T[l], To = e.T[l + 1];
            ConvArgs a = mk_fwd(p, SL, p->layers[e.c1[l]], params, ws, ws + e.out[l], (long)C * Ti, Ti, 1, B, Ti, ws + e.a1[l], (long)C * Ti, Ti, 1, 1);
            RUN(avc_launch_conv(a, s, 0, p->tun));
            ConvArgs b = mk_fwd(p, SL, p->layers[e.c2[l]], params, ws, ws + e.a1[l], (long)C * Ti, Ti, 1, B, Ti, ws + e.a2[l], (long)C * To, To, 1, 1);
            b.g[0].out2 = ws + e.out[l + 1];
            set_res(b, ws + e.out[l], e.c.subsample[l] > 1 ? AVC_RES_AVGPOOL2 : AVC_RES_IDENTITY, (long)C * Ti, Ti, 1, Ti);
            RUN(avc_launch_conv(b, s, 0, p->tun));
        }
        const int Tn = e.T[e.n];
        if (bh) RUN(avc_launch_timepool_fwd_pairs(ws + e.out[e.n], B, Cc, Tn, ws + e.pooled, s));
        else RUN(avc_launch_timepool_fwd(ws + e.out[e.n], B, Cc, Tn, ws + e.pooled, s));
        // dense blocks + output layer: ONE fused launch (dense.hip); activations channel-major [C][B]
        {
            DenseArgs da;
            memset(&da, 0, sizeof(da));
            da.nlayers = 2 * e.nd + 1;
            da.B = B; da.C = Cc; da.slope = SL;
            da.in = ws + e.hd[0];
            da.emb = ws + p->emb;
            for (int l = 0; l < da.nlayers; ++l) {
                const bool last = (l == da.nlayers - 1);
                const LayerP& L = p->layers[last ? e.outl : ((l & 1) ? e.dn2[l / 2] : e.dn1[l / 2])];
                DenseLayer& D = da.layer[l];
                D.wp = ws + L.wpf;
                D.bias = p->par(params, L.b[0]);
                D.Cin = L.Cin; D.Cout = L.Cout; D.Kp = L.nchunk_f * L.CK; D.Mp = L.Mp_f;
                if (!last) {
                    D.act = ws + ((l & 1) ? e.d2[l / 2] : e.d1[l / 2]);
                    D.out2 = (l & 1) ? ws + e.hd[l / 2 + 1] : nullptr;
                }
                da.Kmax = D.Kp > da.Kmax ? D.Kp : da.Kmax;
                da.Wmax = D.Kp * D.Mp > da.Wmax ? D.Kp * D.Mp : da.Wmax;
            }
            RUN(avc_launch_dense(da, 0, s));
        }
    }

    // ---------------- content encoder (model.py:301-323)
    if (!spk_only) {
        const EncNet& e = p->enc;
        const float SL = e.slope;
        const int Cc = e.c.c_h;
        const long C = bh ? Cc / 2 : Cc, CCr = bh ? e.CC / 2 : e.CC;
        // (conv bank, in_conv and its InstanceNorm were issued by content_front(), above)
        for (int l = 0; l < e.n; ++l) {
            const int Ti = e.T[l], To = e.T[l + 1];
            ConvArgs a = mk_fwd(p, SL, p->layers[e.c1[l]], params, ws, ws + e.out[l], (long)C * Ti, Ti, 1, B, Ti, ws + e.y1[l], (long)C * Ti, Ti, 1, 0);
            RUN(conv_in_fwd(p, a, SL, B, Cc, Ti, nullptr, 0, 0, nullptr, 0, 0, ws + e.a1[l], ws + e.st1[l], s, 0, 0, bh, 0, NV));
            ConvArgs b = mk_fwd(p, SL, p->layers[e.c2[l]], params, ws, ws + e.a1[l], (long)C * Ti, Ti, 1, B, Ti, ws + e.y2[l], (long)C * To, To, 1, 0);
            const int rmode = e.c.subsample[l] > 1 ? AVC_RES_AVGPOOL2 : AVC_RES_IDENTITY;
            RUN(conv_in_fwd(p, b, SL, B, Cc, To, nullptr, 0, 0, ws + e.out[l], rmode, Ti, ws + e.out[l + 1], ws + e.st2[l], s, 0, 0, bh, 0, NV));
        }
        const int Tb = p->Tb;
        ConvArgs h = mk_fwd(p, SL, p->layers[e.heads], params, ws, ws + e.out[e.n], (long)C * Tb, Tb, 1, B, Tb, ws + p->muls, (long)2 * e.c.c_out * Tb, Tb, 1, 0);
        h.pairs = 0;   // mu | log_sigma stay fp32 (the loss and the reparameterisation read them)
        RUN(avc_launch_conv(h, s, 0, p->tun));
    }

    join_side(p, mainS, sideS);
    // ---------------- reparameterisation (model.py:383-384) + decoder (model.py:347-371)
    if (!spk_only) {
        const DecNet& d = p->dec;
        const float SL = d.slope;
        const int Cc = d.c.c_h, Czc = d.c.c_in, Tb = p->Tb;
        const long C = bh ? Cc / 2 : Cc, Cz = bh ? Czc / 2 : Czc;   // rows per sample (pair rows with bh)
        if (bh) RUN(avc_launch_reparam_fwd_pairs(ws + p->muls, eps, B, Czc, Tb, ws + d.z, s));
        else RUN(avc_launch_reparam_fwd(ws + p->muls, eps, B, Czc, Tb, ws + d.z, s));
        const long csb = (long)2 * d.n * 2 * Cc;
        {   // all 2n AdaIN affine Linears as ONE GEMM on emb (they share their input)
            ConvArgs a = mk_fwd(p, SL, p->layers[d.affine], params, ws, ws + p->emb, 0, 1, d.c.c_cond, 1, B, ws + d.cond, 0, 1, (int)csb, 0);
            RUN(avc_launch_conv(a, s, 0, p->tun));
        }
        // The decoder is one serial chain of ~40 small kernels (T_l = 16..128): alone on the GPU it leaves
        // most CUs waiting on launch / pipeline latency (0.86 ms with one kernel in flight, traced).  Two
        // half-batch chains on two streams interleave their phases; every tensor is [B, ...], so a half is
        // a pointer offset.
        // phase 0: in_conv + IN; phases 1 .. 2n: (first conv + AdaIN) / (second conv + AdaIN + residual) of block (ph - 1) / 2; phase 2n + 1: out_conv
        auto dec_phase = [&](int b0, int Bn, hipStream_t s, int ph) -> int {
            const long oz = (long)b0 * Cz * Tb, ob0 = (long)b0 * C * Tb;
            const float* cond = ws + d.cond + (long)b0 * csb;
            if (ph == 0) {
                ConvArgs a = mk_fwd(p, SL, p->layers[d.in_conv], params, ws, ws + d.z + oz, (long)Cz * Tb, Tb, 1, Bn, Tb, ws + d.y0 + ob0, (long)C * Tb, Tb, 1, 0);
                RUN(conv_in_fwd(p, a, SL, Bn, Cc, Tb, nullptr, 0, 0, nullptr, 0, 0, ws + d.out[0] + ob0, ws + d.st0, s, B, b0, bh, 0, NV));
                return 0;
            }
            if (ph == 2 * d.n + 1) {
                const int To = p->Tout;
                ConvArgs o = mk_fwd(p, SL, p->layers[d.out_conv], params, ws, ws + d.out[d.n] + (long)b0 * C * To, (long)C * To, To, 1, Bn, To,
                                    ws + p->decb + (long)b0 * p->M * To, (long)p->M * To, To, 1, 0);
                o.pairs = 0;   // dec is fp32 (the L1 loss / the caller read it)
                RUN(avc_launch_conv(o, s, 0, p->tun));
                return 0;
            }
            const int l = (ph - 1) / 2;
            const int Ti = d.T[l], To = d.T[l + 1], up = d.c.upsample[l];
            const long oi = (long)b0 * C * Ti, oo = (long)b0 * C * To;
            if ((ph - 1) % 2 == 0) {
                ConvArgs a = mk_fwd(p, SL, p->layers[d.c1[l]], params, ws, ws + d.out[l] + oi, (long)C * Ti, Ti, 1, Bn, Ti, ws + d.y1[l] + oi, (long)C * Ti, Ti, 1, 0);
                RUN(conv_in_fwd(p, a, SL, Bn, Cc, Ti, cond, csb, (2 * l) * 2 * Cc, nullptr, 0, 0, ws + d.a1[l] + oi, ws + d.st1[l], s, B, b0, bh, 0, NV));
                return 0;
            }
            // second conv: C*up channels, pixel-shuffled on store into [B, C, Ti*up]  (model.py:359-361)
            ConvArgs b = mk_fwd(p, SL, p->layers[d.c2[l]], params, ws, ws + d.a1[l] + oi, (long)C * Ti, Ti, 1, Bn, Ti, ws + d.y2[l] + oo, (long)C * To, To, 1, 0);
            b.ops = up;
            if (bh) {   // the conv-output pairs [B][c_h up / 2][Ti] ARE the natural bf16 rows of the shuffled tensor: "planar" y2 (rowops_pairs.hip)
                b.ops = 1;
                b.ob = (long)C * To; b.oc = Ti;
            }
            RUN(conv_in_fwd(p, b, SL, Bn, Cc, To, cond, csb, (2 * l + 1) * 2 * Cc, ws + d.out[l] + oi, up > 1 ? AVC_RES_UP2 : AVC_RES_IDENTITY, Ti,
                            ws + d.out[l + 1] + oo, ws + d.st2[l], s, B, b0, bh, (bh && up > 1) ? 1 : 0, NV));
            return 0;
        };
        const int nph = 2 * d.n + 2;
        if (B >= p->tun.dec_split_min && side_ready(p)) {
            // the host ISSUES the two chains phase by phase in turn: a chain issued whole before the other one starts has run to its end by the
            // time the second chain's first launch reaches the GPU (the kernels are as short as a launch call: traced, profiles/r03_bf16s_timeline.txt)
            const int Bh = B / 2;
            const hipStream_t s2 = fork_side(p, s);
            for (int ph = 0; ph < nph; ++ph) {
                RUN(dec_phase(0, Bh, s, ph));
                RUN(dec_phase(Bh, B - Bh, s2, ph));
            }
            join_side(p, s, s2);
        } else {
            for (int ph = 0; ph < nph; ++ph) RUN(dec_phase(0, B, s, ph));
        }
    }
    return 0;
}

extern "C" int avc_forward(const avc_plan* p, const float* params, const float* x, long sxb, long sxc, int sxt,
                           const float* x_cond, long scb, long scc, int sct, const float* eps, float* ws, void* stream) {
    if (!p || !params || !x || !ws) return fail(-1, "avc_forward: null argument");
    if (p->flags & AVC_PLAN_RAGGED) return fail(-8, "avc_forward: ragged plans run through avc_forward_ragged");
    if (!x_cond) {
        x_cond = x; scb = sxb; scc = sxc; sct = sxt;
    }
    return forward_impl(p, params, x, sxb, sxc, sxt, x_cond, scb, scc, sct, eps, ws, (hipStream_t)stream);
}

extern "C" int avc_forward_ex(const avc_plan* p, const float* params, const float* x, long sxb, long sxc, int sxt,
                              const float* x_cond, long scb, long scc, int sct, const float* eps, float* ws, int flags, void* stream) {
    if (!p || !params || !x || !ws) return fail(-1, "avc_forward_ex: null argument");
    if (p->flags & AVC_PLAN_RAGGED) return fail(-8, "avc_forward_ex: ragged plans run through avc_forward_ragged");
    if (flags & ~AVC_FWD_WEIGHTS_PACKED) return fail(-1, "avc_forward_ex: unknown flag");
    if (!x_cond) {
        x_cond = x; scb = sxb; scc = sxc; sct = sxt;
    }
    return forward_impl(p, params, x, sxb, sxc, sxt, x_cond, scb, scc, sct, eps, ws, (hipStream_t)stream, (flags & AVC_FWD_WEIGHTS_PACKED) != 0);
}

extern "C" int avc_loss(const avc_plan* p, const float* x, long sxb, long sxc, int sxt, float lambda_rec, float* ws,
                        void* stream) {
    if (p->flags & AVC_PLAN_INFERENCE) return fail(-8, "avc_loss: the plan was created with AVC_PLAN_INFERENCE");
    if (p->Tout != p->T) return fail(-7, "avc_loss: L1Loss needs dec and x of equal length (T % 8 == 0 for the stock config)");
    RUN(avc_launch_loss(ws + p->decb, x, sxb, sxc, sxt, p->B, p->M, p->T, ws + p->muls, p->dec.c.c_in, p->Tb, lambda_rec,
                        ws + p->ddec, ws + p->loss_partial, ws + p->losses, (hipStream_t)stream));
    return 0;
}

// --------------------------------------------------------------------------
// backward
// --------------------------------------------------------------------------
static int enc_back_front(BwdCtx& c, const EncNet& e, const float* x, long sxb, long sxc, int sxt, const float* dy_in) {
    // dy_in: gradient wrt the in_conv output [B, C, T0]
    const avc_plan* p = c.p;
    const bool bh = p->bh;
    const int B = p->B, T0 = e.T[0];
    const long C = bh ? e.c.c_h / 2 : e.c.c_h, CCr = bh ? e.CC / 2 : e.CC, Cbr = bh ? e.c.c_bank / 2 : e.c.c_bank;   // rows per sample (pair rows with bh)
    const float SL = e.slope;
    float* ws = c.ws;
    const LayerP& L = p->layers[e.in_conv];
    RUN(wgrad_layer(c, L, ws + e.cat, CCr * T0, T0, 1, dy_in, (long)C * T0, T0, 1, 1, B, T0, T0));
    if (!c.dry) {
        // d(cat) for the bank channels only, masked by the bank ReLU (cat > 0)
        ConvArgs a = mk_dgrad(SL, L, ws, dy_in, (long)C * T0, T0, 1, 1, B, T0, T0, nullptr, CCr * T0, T0, 1);
        a.g[0].out2 = ws + e.dcat;
        a.g[0].mask = ws + e.cat;
        RUN(avc_launch_conv(a, c.s, 0, p->tun));
    }
    for (int g = 0; g < e.nb; ++g) {
        const LayerP& Lb = p->layers[e.bank[g]];
        if (bh)   // the bank read the pair copy of the input at the tail of the concat buffer
            RUN(wgrad_layer(c, Lb, ws + e.cat + (long)e.nb * Cbr * T0, CCr * T0, T0, 1, ws + e.dcat + (long)g * Cbr * T0, CCr * T0, T0, 1, 1, B, T0, T0));
        else
            RUN(wgrad_layer(c, Lb, x, sxb, sxc, sxt, ws + e.dcat + (long)g * e.c.c_bank * T0, (long)e.CC * T0, T0, 1, 1, B, T0, T0));
    }
    return 0;
}

int avc_backward_impl(const avc_plan* p, const float* params, const float* x, long sxb, long sxc, int sxt, const float* xc,
                      long scb, long scc, int sct, const float* eps, const float* d_dec, const float* d_muls_up,
                      const float* d_emb_up, float lambda_kl, float* grads, float* ws, hipStream_t s, bool dry,
                      long* slab_need) {
    BwdCtx c;
    c.p = p; c.params = params; c.grads = grads; c.ws = ws; c.s = s; c.dry = dry; c.slab_used = 0;
    if (!dry) avc_prof_mark(0, s);
    c.nev = 0; c.dy_used = 0; c.pend_units = 0;
    const bool overlap = !dry && side_ready(p) && !(p->tun.dbg_streams & 1);
    const bool use_side = !dry && !(p->tun.dbg_streams & 2);
    c.wstream = overlap ? p->wstream[0] : s;
    const int B = p->B;
    const bool bh = p->bh;
    const int NV = (int)p->tun.in_pairs_nv;
    float* gA = ws + p->gA;
    float* gB = ws + p->gB;
    float* gC = ws + p->gC;
    float* dyA = nullptr;
    float* dyB = nullptr;
    auto rot = [&]() { float* t = gA; gA = gC; gC = t; };

    // ---------------- decoder
    {
        const DecNet& d = p->dec;
        const float SL = d.slope;
        const int Cc = d.c.c_h, Czc = d.c.c_in, Tb = p->Tb, To = p->Tout;
        const long C = bh ? Cc / 2 : Cc, Cz = bh ? Czc / 2 : Czc;   // rows per sample (pair rows with bh): every stride / offset below
        const long Mr = bh ? p->M / 2 : p->M;
        const long csb = (long)2 * d.n * 2 * Cc;
        const float* ddec = d_dec ? d_dec : ws + p->ddec;
        if (bh) {   // the conv launches read pair operands: d(dec) (fp32: written by avc_loss or handed in by the autograd seam) -> pairs
            if (!dry) RUN(avc_launch_to_pairs(ddec, (long)p->M * To, To, 1, B, p->M, To, ws + p->ddecp, Mr * To, To, s));
            ddec = ws + p->ddecp;
        }
        const LayerP& Lo = p->layers[d.out_conv];
        const LayerP& Li = p->layers[d.in_conv];
        // every dy a weight-gradient launch reads gets its own buffer (the launches run later, beside the encoders' backward)
        float* dy2[AVC_MAX_BLOCKS];
        float* dy1[AVC_MAX_BLOCKS];
        for (int l = d.n - 1; l >= 0; --l) {
            dy2[l] = c.fresh((long)B * C * d.T[l + 1]);
            dy1[l] = c.fresh((long)B * C * d.T[l]);
        }
        float* dy0 = c.fresh((long)B * C * Tb);
        // The decoder's backward is one serial chain of ~26 small kernels (T_l = 128 .. 16) that runs ALONE on the GPU -- nothing
        // else is ready before d(emb) and d(muls) exist (traced: 0.7 ms with one kernel in flight).  Like the forward pass it is
        // issued as two half-batch chains on two streams for B >= dec_split_min: every tensor is [B, ...], a half is a pointer
        // offset, and InstanceNorm / AdaIN statistics and d(cond) rows are per sample.
        struct ChainSt {
            float *gA, *gB, *gC;
            bool fused;   // the InstanceNorm backward that opens the NEXT phase already ran inside this phase's input-gradient launch
        };
        // phase 0: out_conv input gradient; phases 1 .. 2n walk the blocks from the last one: (AdaIN backward of the second conv + its input
        // gradient) / (AdaIN backward of the first conv + its input gradient joined with the skip path); phase 2n + 1: IN backward + in_conv
        auto chain_phase = [&](ChainSt& st, int b0, int Bn, hipStream_t s, int ph) -> int {
            auto half_in_bwd = [&](const float* g, long off, long yoff, long stoff, int T, int coff, bool cond, float* dy, int planar = 0) -> int {
                INBwdArgs a;
                a.g = g + off; a.y = ws + yoff + off;
                a.mean = ws + stoff + (long)b0 * Cc; a.rstd = ws + stoff + (long)B * Cc + (long)b0 * Cc;
                a.cond = cond ? ws + d.cond + (long)b0 * csb : nullptr; a.cond_sb = csb; a.cond_off = coff;
                a.dy = dy + off; a.dcond = cond ? ws + d.dcond + (long)b0 * csb : nullptr; a.dcond_sb = csb; a.dcond_off = coff;
                a.R = Bn * Cc; a.C = Cc; a.T = T; a.relu = 1; a.slope = SL; a.planar = planar; a.nv_hint = (int)p->tun.in_pairs_nv;
                if (bh) {
                    a.R = Bn * (Cc / 2);
                    return avc_launch_in_bwd_pairs(a, s);
                }
                return avc_launch_in_bwd(a, s);
            };
            // An input-gradient launch whose output rows are the d(out) of the NEXT phase's InstanceNorm layer runs that layer's backward
            // in its epilogue where the tile holds whole rows (rows of 16 / 32 / 64 frames, fp32: avc_conv_inb_fusable); the next phase
            // then skips its row kernel.  `keep_g`: the launch's own output is read again (the block's skip path).
            auto dgrad_inb = [&](ConvArgs& a, bool keep_g, long off, long yoff, long stoff, int T, int coff, bool cond, float* dy, bool planar = false) -> int {
                st.fused = false;
                if (!planar && a.Tout == T && avc_conv_inb_fusable(a, p->tun)) {   // (pair tensors too since round 6; the "planar" rows behind a pixel shuffle keep the row kernel)
                    a.inb.dy = dy + off; a.inb.y = ws + yoff + off;
                    a.inb.mean = ws + stoff + (long)b0 * Cc; a.inb.rstd = ws + stoff + (long)B * Cc + (long)b0 * Cc;
                    a.inb.cond = cond ? ws + d.cond + (long)b0 * csb : nullptr; a.inb.cond_sb = csb; a.inb.cond_off = coff;
                    a.inb.dcond = cond ? ws + d.dcond + (long)b0 * csb : nullptr; a.inb.dcond_sb = csb; a.inb.dcond_off = coff;
                    a.inb.C = Cc; a.inb.relu = 1;
                    if (!keep_g) a.g[0].out = nullptr;
                    st.fused = true;
                }
                return avc_launch_conv(a, s, 0, p->tun);
            };
            const bool was_fused = st.fused;   // (set by the previous phase of THIS chain)
            if (ph == 0) {
                const long oi = (long)b0 * Mr * To, oo = (long)b0 * C * To;
                ConvArgs a = mk_dgrad(SL, Lo, ws, ddec + oi, Mr * To, To, 1, 1, Bn, To, To, st.gA + oo, (long)C * To, To, 1);
                // next: the AdaIN backward of the last block's second conv (rows of To frames)
                const int ln = d.n - 1;
                RUN(dgrad_inb(a, true, oo, d.y2[ln], d.st2[ln], d.T[ln + 1], (2 * ln + 1) * 2 * Cc, true, dy2[ln], bh && d.c.upsample[ln] > 1));
                return 0;
            }
            if (ph == 2 * d.n + 1) {
                const long ob = (long)b0 * C * Tb;
                if (!was_fused) RUN(half_in_bwd(st.gA, ob, d.y0, d.st0, Tb, 0, false, dy0));
                st.fused = false;
                ConvArgs a = mk_dgrad(SL, Li, ws, dy0 + ob, (long)C * Tb, Tb, 1, 1, Bn, Tb, Tb, ws + p->dz + (long)b0 * Czc * Tb, (long)Czc * Tb, Tb, 1);
                a.pairs = 0;   // d(z) is fp32: the latent backward combines it with the fp32 mu / log_sigma
                RUN(avc_launch_conv(a, s, 0, p->tun));
                return 0;
            }
            const int l = d.n - 1 - (ph - 1) / 2;
            const int Ti = d.T[l], T2 = d.T[l + 1], up = d.c.upsample[l];
            const long o2 = (long)b0 * C * T2, o1 = (long)b0 * C * Ti;
            if ((ph - 1) % 2 == 0) {
                if (!was_fused) RUN(half_in_bwd(st.gA, o2, d.y2[l], d.st2[l], T2, (2 * l + 1) * 2 * Cc, true, dy2[l], (bh && up > 1) ? 1 : 0));
                // dy2 is the pixel-shuffled layout [B, C, Ti*up]; view it as the conv output [B, C*up, Ti]
                // (pair plans: dy2 was written planar = as the conv-output pairs [B][c_h up / 2][Ti], a plain stride-1 source)
                ConvArgs a = bh ? mk_dgrad(SL, p->layers[d.c2[l]], ws, dy2[l] + o2, (long)C * T2, Ti, 1, 1, Bn, Ti, Ti, st.gB + o1, (long)C * Ti, Ti, 1)
                                : mk_dgrad(SL, p->layers[d.c2[l]], ws, dy2[l] + o2, (long)C * T2, T2, up, up, Bn, Ti, Ti, st.gB + o1, (long)C * Ti, Ti, 1);
                // next: the AdaIN backward of this block's first conv (rows of Ti frames); gB has no other reader
                RUN(dgrad_inb(a, false, o1, d.y1[l], d.st1[l], Ti, (2 * l) * 2 * Cc, true, dy1[l]));
                return 0;
            }
            if (!was_fused) RUN(half_in_bwd(st.gB, o1, d.y1[l], d.st1[l], Ti, (2 * l) * 2 * Cc, true, dy1[l]));
            ConvArgs a = mk_dgrad(SL, p->layers[d.c1[l]], ws, dy1[l] + o1, (long)C * Ti, Ti, 1, 1, Bn, Ti, Ti, st.gC + o1, (long)C * Ti, Ti, 1);
            set_res(a, st.gA + o2, up > 1 ? AVC_RES_UPT : AVC_RES_IDENTITY, (long)C * T2, T2, 1, T2);
            // next: the AdaIN backward of the previous block's second conv, or the in_conv's InstanceNorm (rows of Ti frames); gC feeds the skip path too
            if (l > 0) RUN(dgrad_inb(a, true, o1, d.y2[l - 1], d.st2[l - 1], Ti, (2 * (l - 1) + 1) * 2 * Cc, true, dy2[l - 1], bh && d.c.upsample[l - 1] > 1));
            else RUN(dgrad_inb(a, true, o1, d.y0, d.st0, Tb, 0, false, dy0));
            float* t = st.gA; st.gA = st.gC; st.gC = t;
            return 0;
        };
        // The weight gradient of a layer is RECORDED right behind the phase that produces its dy.  By default all of them are held and go
        // out behind the dense-stack backward kernel (below).  avc_tuning.dec_wgrad_flush = N > 0: every N recorded layers go out at once,
        // on the wgrad stream, as launches of only dec_wgrad_wgs persistent workgroups -- the decoder's backward chain is latency-bound
        // and leaves most CUs idle; a launch that occupies only part of the chip is filler for exactly that phase.
        c.hold = true;
        const int early = p->tun.dec_wgrad_flush;
        auto rec_phase = [&](int ph) -> int {   // (the order of the records is the order of round 3: out_conv, blocks n-1 .. 0 (second, first conv), in_conv)
            if (ph == 0) return wgrad_layer(c, Lo, ws + d.out[d.n], (long)C * To, To, 1, ddec, Mr * To, To, 1, 1, B, To, To);
            if (ph == 2 * d.n + 1) return wgrad_layer(c, Li, ws + d.z, (long)Cz * Tb, Tb, 1, dy0, (long)C * Tb, Tb, 1, 1, B, Tb, Tb);
            const int l = d.n - 1 - (ph - 1) / 2;
            const int Ti = d.T[l], T2 = d.T[l + 1], up = d.c.upsample[l];
            if ((ph - 1) % 2 == 0) {
                if (bh) return wgrad_layer(c, p->layers[d.c2[l]], ws + d.a1[l], (long)C * Ti, Ti, 1, dy2[l], (long)C * T2, Ti, 1, 1, B, Ti, Ti);
                return wgrad_layer(c, p->layers[d.c2[l]], ws + d.a1[l], (long)C * Ti, Ti, 1, dy2[l], (long)C * T2, T2, up, up, B, Ti, Ti);
            }
            return wgrad_layer(c, p->layers[d.c1[l]], ws + d.out[l], (long)C * Ti, Ti, 1, dy1[l], (long)C * Ti, Ti, 1, 1, B, Ti, Ti);
        };
        {
            const int nph = 2 * d.n + 2;
            ChainSt st0 = {ws + p->gA, ws + p->gB, ws + p->gC, false}, st1 = st0;
            const bool split = use_side && B >= p->tun.dec_split_min && side_ready(p);
            const int Bh = B / 2;
            const hipStream_t s2 = split ? fork_side(p, s) : s;
            for (int ph = 0; ph < nph; ++ph) {   // the two half-batch chains are issued in turn (see the forward pass)
                if (!dry) {
                    if (split) {
                        RUN(chain_phase(st0, 0, Bh, s, ph));
                        RUN(chain_phase(st1, Bh, B - Bh, s2, ph));
                    } else {
                        RUN(chain_phase(st0, 0, B, s, ph));
                    }
                }
                // phase ph consumed (ph = 0: d(dec)) or produced (its InstanceNorm backward) the dy of this layer -- on BOTH chains' streams
                RUN(rec_phase(ph));
                if (early > 0 && (int)c.pend.size() >= early && ph + 1 < nph) {
                    c.s_extra = s2;
                    c.target_wgs = (int)p->tun.dec_wgrad_wgs;
                    RUN(flush_wgrads(c, true));
                    c.s_extra = nullptr;
                    c.target_wgs = 0;
                }
            }
            if (split) join_side(p, s, s2);
        }
        // affine Linears: dW/db from (emb, dcond), d(emb) = W^T dcond (+ upstream)
        const LayerP& La = p->layers[d.affine];
        RUN(wgrad_layer(c, La, ws + p->emb, 0, 1, d.c.c_cond, ws + d.dcond, 0, 1, (int)csb, 1, 1, B, B));
        {
            // d(emb)[i][b] = sum_o W[o][i] * dcond[b][o]: a 128 x B output with K = 2n*2C = 3072.  As a dgrad
            // launch that is 8 workgroups walking 96 chunks one after the other (206 us on the critical
            // path, measured); as a split-K GEMM through the weight-gradient kernel ("co" = i read from the
            // packed forward image [i][o], "ci" = b read from dcond [b][o], "t" = o) it is ~50 workgroups.
            WgradArgs w;
            memset(&w, 0, sizeof(w));
            w.x.ptr = ws + d.dcond; w.x.sb = 0; w.x.sc = csb; w.x.st = 1; w.x.ps = 1;
            w.dy.ptr = ws + La.wplain; w.dy.sb = 0; w.dy.sc = La.Mp_f; w.dy.st = 1; w.dy.ps = 1;
            w.B = 1; w.Cin = B; w.Cout = d.c.c_cond; w.Tin = La.Cout; w.Tout = La.Cout;
            w.KS = 1; w.padL = 0; w.stride = 1; w.bf16 = bh ? AVC_COMPUTE_BF16 : p->compute;   // (fp32-stored operands either way)
            w.rows_per_src = w.Cout;
            avc_wgrad_plan_batch(&w, 1, 256);
            const long off = c.slab_used;
            c.slab_used += (w.slab_need + 63) / 64 * 64;
            if (!dry) {
                w.slab = ws + p->slab + off;
                w.dbslab = nullptr;
                w.dw = ws + p->demb;   // demb: channel-major [c_cond][B]
                w.db = nullptr;
                RUN(avc_launch_wgrad_batch(&w, 1, s, p->tun.wgrad_ablation));
                if (d_emb_up) RUN(avc_launch_add_transposed(ws + p->demb, d_emb_up, B, d.c.c_cond, s));
            }
        }
        // latent: KL term + reparameterisation (solver.py:86, model.py:384)
        if (!dry) {
            float lk = lambda_kl / (float)((long)B * Czc * Tb);
            RUN(avc_launch_latent_bwd(ws + p->muls, eps, ws + p->dz, d_muls_up, B, Czc, Tb, lk, ws + p->dmuls, s));
            if (bh) RUN(avc_launch_to_pairs(ws + p->dmuls, (long)2 * Czc * Tb, Tb, 1, B, 2 * Czc, Tb, ws + p->dmulsp, (long)Czc * Tb, Tb, s));
        }
        // (the decoder's weight gradients are launched from inside the speaker branch below, BEHIND the dense-stack backward kernel: that
        // kernel opens the speaker branch's critical chain, needs a whole CU's LDS per workgroup and cannot share a CU with a persistent
        // weight-gradient workgroup -- launched after 256 of those it waited for them: 413 instead of 66 us, traced in round 4)
    }

    // ---------------- content encoder (issued from inside the speaker branch below, right after that branch's first kernel: the host
    // issues launches about as fast as these kernels run, so whichever branch is issued second starts late by the other one's issue time)
    auto content_branch = [&]() -> int {
        const EncNet& e = p->enc;
        const float SL = e.slope;
        const int Cc = e.c.c_h, Tb = p->Tb;
        const long C = bh ? Cc / 2 : Cc, Co2 = bh ? e.c.c_out : 2 * e.c.c_out;   // rows per sample (pair rows with bh)
        const float* dmuls = ws + (bh ? p->dmulsp : p->dmuls);
        const LayerP& Lh = p->layers[e.heads];
        RUN(wgrad_layer(c, Lh, ws + e.out[e.n], (long)C * Tb, Tb, 1, dmuls, (long)Co2 * Tb, Tb, 1, 1, B, Tb, Tb));
        // An input-gradient launch whose output rows are the d(out) of the NEXT InstanceNorm layer runs that layer's backward in its
        // epilogue where the tile holds whole rows (avc_conv_inb_fusable); the row kernel then stays out of the chain.
        auto dgrad_inb = [&](ConvArgs& a, bool keep_g, const float* y, const float* st, int T, float* dy) -> bool {
            bool fused = false;
            if (a.Tout == T && avc_conv_inb_fusable(a, p->tun)) {   // (pair tensors too since round 6)
                a.inb.dy = dy; a.inb.y = y; a.inb.mean = st; a.inb.rstd = st + (long)B * Cc;
                a.inb.C = Cc; a.inb.relu = 1;
                if (!keep_g) a.g[0].out = nullptr;
                fused = true;
            }
            return fused;
        };
        bool fused = false;
        dyA = c.fresh((long)B * C * e.T[e.n]);   // d(y2) of the last block
        {
            ConvArgs a = mk_dgrad(SL, Lh, ws, dmuls, (long)Co2 * Tb, Tb, 1, 1, B, Tb, Tb, gA, (long)C * Tb, Tb, 1);
            fused = dgrad_inb(a, true, ws + e.y2[e.n - 1], ws + e.st2[e.n - 1], e.T[e.n], dyA);
            if (!dry) RUN(avc_launch_conv(a, s, 0, p->tun));
        }
        for (int l = e.n - 1; l >= 0; --l) {
            const int Ti = e.T[l], T2 = e.T[l + 1], sub = e.c.subsample[l];
            const LayerP& L1 = p->layers[e.c1[l]];
            const LayerP& L2 = p->layers[e.c2[l]];
            if (!dry && !fused) RUN(in_bwd(SL, gA, ws + e.y2[l], ws + e.st2[l], B, Cc, T2, nullptr, 0, 0, dyA, nullptr, s, bh, NV));
            dyB = c.fresh((long)B * C * Ti);
            {
                ConvArgs a = mk_dgrad(SL, L2, ws, dyA, (long)C * T2, T2, 1, 1, B, T2, Ti, gB, (long)C * Ti, Ti, 1);
                fused = dgrad_inb(a, false, ws + e.y1[l], ws + e.st1[l], Ti, dyB);   // gB has no other reader
                if (!dry) RUN(avc_launch_conv(a, s, 0, p->tun));
            }
            RUN(wgrad_layer(c, L2, ws + e.a1[l], (long)C * Ti, Ti, 1, dyA, (long)C * T2, T2, 1, 1, B, Ti, T2));
            if (!dry && !fused) RUN(in_bwd(SL, gB, ws + e.y1[l], ws + e.st1[l], B, Cc, Ti, nullptr, 0, 0, dyB, nullptr, s, bh, NV));
            dyA = c.fresh((long)B * C * Ti);   // d(y2) of block l - 1, or d(in_conv output) for l == 0: rows of Ti frames either way
            {
                ConvArgs a = mk_dgrad(SL, L1, ws, dyB, (long)C * Ti, Ti, 1, 1, B, Ti, Ti, gC, (long)C * Ti, Ti, 1);
                set_res(a, gA, sub > 1 ? AVC_RES_POOLT : AVC_RES_IDENTITY, (long)C * T2, T2, 1, T2);
                fused = dgrad_inb(a, true, l > 0 ? ws + e.y2[l - 1] : ws + e.h0, l > 0 ? ws + e.st2[l - 1] : ws + e.st0, Ti, dyA);   // gC feeds the skip path too
                if (!dry) RUN(avc_launch_conv(a, s, 0, p->tun));
            }
            RUN(wgrad_layer(c, L1, ws + e.out[l], (long)C * Ti, Ti, 1, dyB, (long)C * Ti, Ti, 1, 1, B, Ti, Ti));
            rot();
        }
        if (!dry && !fused) RUN(in_bwd(SL, gA, ws + e.h0, ws + e.st0, B, Cc, e.T[0], nullptr, 0, 0, dyA, nullptr, s, bh, NV));
        RUN(enc_back_front(c, e, x, sxb, sxc, sxt, dyA));
        if (!dry) avc_prof_mark(3, s);
        RUN(flush_wgrads(c));
        return 0;
    };

    // ---------------- speaker encoder (side stream, own temporaries: concurrent with the content encoder)
    const hipStream_t mainS = s;
    const hipStream_t sideS = use_side ? fork_side(p, mainS) : mainS;
    {
        const hipStream_t s = sideS;
        c.s = sideS;
        float* gA = ws + p->gA2;
        float* gB = ws + p->gB2;
        float* gC = ws + p->gC2;
        float* dyA = nullptr;
        float* dyB = nullptr;
        c.wstream = (overlap && sideS != mainS) ? p->wstream[1] : sideS;
        auto rot = [&]() { float* t = gA; gA = gC; gC = t; };
        const EncNet& e = p->spk;
        const float SL = e.slope;
        const int Cc = e.c.c_h;
        const long C = bh ? Cc / 2 : Cc;   // rows per sample of the [B, c_h, T] tensors (pair rows with bh)
        float* dhA = ws + p->dhA;
        const LayerP& Lo = p->layers[e.outl];
        // output layer + dense blocks: ONE fused dgrad launch produces every dz (and d pooled);
        // the weight gradients are ordinary GEMMs over the batch on the wgrad stream
        {
            DenseArgs da;
            memset(&da, 0, sizeof(da));
            da.nlayers = 2 * e.nd + 1;
            da.B = B; da.C = Cc; da.slope = SL;
            da.in = ws + p->demb;      // [c_out][B] channel-major
            da.in2 = nullptr;           // (upstream d_emb is already folded into demb above)
            da.dpooled = dhA;
            float* dzl[AVC_DENSE_MAXL];
            for (int l = 0; l < da.nlayers; ++l) {
                const bool last = (l == da.nlayers - 1);
                const LayerP& L = p->layers[last ? e.outl : ((l & 1) ? e.dn2[l / 2] : e.dn1[l / 2])];
                DenseLayer& D = da.layer[l];
                D.wp = ws + L.wpd;
                D.Cin = L.Cin; D.Cout = L.Cout; D.Kp = L.nchunk_d * L.CKd; D.Mp = L.Mp_d;
                dzl[l] = last ? nullptr : c.fresh((long)Cc * B);
                if (!last) {
                    D.act = ws + ((l & 1) ? e.d2[l / 2] : e.d1[l / 2]);
                    D.dz = dzl[l];
                }
                da.Kmax = D.Kp > da.Kmax ? D.Kp : da.Kmax;
                da.Wmax = D.Kp * D.Mp > da.Wmax ? D.Kp * D.Mp : da.Wmax;
            }
            if (!dry) RUN(avc_launch_dense(da, 1, s));
            if (!dry) avc_prof_mark(1, s);
            {   // the decoder's pending weight gradients: their dy operands are final on the main stream; they go out now, ordered behind
                // the dense-stack kernel as well
                const hipStream_t ws0 = overlap ? p->wstream[0] : mainS;
                if (overlap && sideS != mainS) {
                    hipEventRecord(p->ev_dense, sideS);
                    hipStreamWaitEvent(ws0, p->ev_dense, 0);
                }
                c.s = mainS;
                c.wstream = ws0;
                c.hold = false;
                RUN(flush_wgrads(c));  // decoder gradients are complete
                // ... which lets a data-parallel caller start their all-reduce under the encoders' backward
                // (avc_plan_stream_wait_grads): the decoder's parameters are the tail of the flat buffer
                if (!dry && p->side_state == 1) hipEventRecord(p->ev_dec_grads, c.wstream);
                if (!dry) avc_prof_mark(5, c.wstream);
                c.s = sideS;
                c.wstream = (overlap && sideS != mainS) ? p->wstream[1] : sideS;
            }
            const LayerP* gl[AVC_DENSE_MAXL];
            const float* gx[AVC_DENSE_MAXL];
            const float* gdy[AVC_DENSE_MAXL];
            int ng = 0;
            gl[ng] = &Lo; gx[ng] = ws + e.hd[e.nd]; gdy[ng] = ws + p->demb; ++ng;
            for (int l = e.nd - 1; l >= 0; --l) {
                gl[ng] = &p->layers[e.dn2[l]]; gx[ng] = ws + e.d1[l]; gdy[ng] = dzl[2 * l + 1]; ++ng;
                gl[ng] = &p->layers[e.dn1[l]]; gx[ng] = ws + e.hd[l]; gdy[ng] = dzl[2 * l]; ++ng;
            }
            // 13 Linear layers on [C][B] channel-major operands: same kernel instance -> ONE batched launch
            for (int i = 0; i < ng; ++i) RUN(wgrad_layer(c, *gl[i], gx[i], 0, B, 1, gdy[i], 0, B, 1, 1, 1, B, B));
        }
        // ... the content encoder's whole branch is issued now (main stream), under the dense-stack kernel that opens this branch
        RUN(flush_wgrads(c));
        c.s = mainS;
        c.wstream = overlap ? p->wstream[0] : mainS;
        if (!dry && (p->tun.dbg_streams & 8) && sideS != mainS) {   // (diagnostic: the content chain starts behind the dense-stack kernel and the speaker's dense weight gradients)
            hipEventRecord(p->ev_dense, sideS);
            hipStreamWaitEvent(mainS, p->ev_dense, 0);
        }
        RUN(content_branch());
        if (!dry && (p->tun.dbg_streams & 4) && sideS != mainS) {   // (diagnostic: the speaker's conv chain starts behind the content chain)
            hipEventRecord(p->ev_join, mainS);
            hipStreamWaitEvent(sideS, p->ev_join, 0);
        }
        c.s = sideS;
        c.wstream = (overlap && sideS != mainS) ? p->wstream[1] : sideS;
        // pooled -> [B,C,Tn] ; dy2 of the last block masked by its ReLU output
        const int Tn = e.T[e.n];
        dyA = c.fresh((long)B * C * Tn);
        if (!dry) {
            if (bh) RUN(avc_launch_timepool_bwd_pairs(dhA, ws + e.a2[e.n - 1], B, Cc, Tn, gA, dyA, SL, s));
            else RUN(avc_launch_timepool_bwd(dhA, ws + e.a2[e.n - 1], B, Cc, Tn, gA, dyA, SL, s));
            avc_prof_mark(2, s);
        }
        for (int l = e.n - 1; l >= 0; --l) {
            const int Ti = e.T[l], T2 = e.T[l + 1], sub = e.c.subsample[l];
            const LayerP& L1 = p->layers[e.c1[l]];
            const LayerP& L2 = p->layers[e.c2[l]];
            // dyA = G_{l+1} * (a2 > 0)
            dyB = c.fresh((long)B * C * Ti);
            if (!dry) {
                ConvArgs a = mk_dgrad(SL, L2, ws, dyA, (long)C * T2, T2, 1, 1, B, T2, Ti, nullptr, (long)C * Ti, Ti, 1);
                a.g[0].out2 = dyB;
                a.g[0].mask = ws + e.a1[l];
                RUN(avc_launch_conv(a, s, 0, p->tun));
            }
            RUN(wgrad_layer(c, L2, ws + e.a1[l], (long)C * Ti, Ti, 1, dyA, (long)C * T2, T2, 1, 1, B, Ti, T2));
            dyA = c.fresh((long)B * C * Ti);  // (the wgrad of conv2 above still reads the previous dyA)
            if (!dry) {
                ConvArgs a = mk_dgrad(SL, L1, ws, dyB, (long)C * Ti, Ti, 1, 1, B, Ti, Ti, gC, (long)C * Ti, Ti, 1);
                set_res(a, gA, sub > 1 ? AVC_RES_POOLT : AVC_RES_IDENTITY, (long)C * T2, T2, 1, T2);
                a.g[0].out2 = dyA;  // next: dy2 of block l-1, or d(in_conv out) for l == 0
                a.g[0].mask = (l > 0) ? ws + e.a2[l - 1] : ws + e.h0;
                RUN(avc_launch_conv(a, s, 0, p->tun));
            }
            RUN(wgrad_layer(c, L1, ws + e.out[l], (long)C * Ti, Ti, 1, dyB, (long)C * Ti, Ti, 1, 1, B, Ti, Ti));
            rot();
        }
        RUN(enc_back_front(c, e, xc, scb, scc, sct, dyA));
        if (!dry) avc_prof_mark(4, s);
        RUN(flush_wgrads(c));
        // the speaker encoder's gradients (head of the flat buffer) are final once its wgrad stream drains: a data-parallel caller
        // reduces them under the content encoder's longer branch (avc_plan_stream_wait_grads(AVC_GRADS_SPEAKER))
        if (!dry && p->side_state == 1) hipEventRecord(p->ev_spk_grads, c.wstream);
        c.s = mainS;
        c.wstream = overlap ? p->wstream[0] : mainS;
    }
    if (!dry) join_side(p, mainS, sideS);
    if (overlap) {
        for (int i = 0; i < 2; ++i) {
            hipEventRecord(p->wjoin[i], p->wstream[i]);
            hipStreamWaitEvent(mainS, p->wjoin[i], 0);
        }
    }
    if (!dry && p->side_state == 1) hipEventRecord(p->ev_all_grads, s);
    if (!dry) avc_prof_mark(6, s);
    if (slab_need) {
        slab_need[0] = c.slab_used;
        slab_need[1] = c.dy_used;
        slab_need[2] = c.nev;
    }
    return 0;
}

extern "C" int avc_backward(const avc_plan* p, const float* params, const float* x, long sxb, long sxc, int sxt,
                            const float* x_cond, long scb, long scc, int sct, const float* eps, const float* d_dec,
                            const float* d_muls_up, const float* d_emb_up, float lambda_kl, float* grads, float* ws,
                            void* stream) {
    if (!p || !params || !x || !ws || !grads) return fail(-1, "avc_backward: null argument");
    if (p->flags & AVC_PLAN_INFERENCE) return fail(-8, "avc_backward: the plan was created with AVC_PLAN_INFERENCE (no gradient buffers)");
    if (!x_cond) {
        x_cond = x; scb = sxb; scc = sxc; sct = sxt;
    }
    return avc_backward_impl(p, params, x, sxb, sxc, sxt, x_cond, scb, scc, sct, eps, d_dec, d_muls_up, d_emb_up, lambda_kl,
                             grads, ws, (hipStream_t)stream, false, nullptr);
}


// --------------------------------------------------------------------------
// ragged inference: utterances of different lengths in ONE launch set
// --------------------------------------------------------------------------
static void rag_level(avc_plan* p, avc_plan::RagLevel& L, const std::vector<int>& T) {
    const int B = (int)T.size();
    L.T = T;
    L.off.assign(B + 1, 0);
    for (int b = 0; b < B; ++b) L.off[b + 1] = L.off[b] + T[b];
    std::vector<int> tiles;
    for (int b = 0; b < B; ++b)
        for (int t0 = 0; t0 < T[b]; t0 += 64) {
            tiles.push_back(b);
            tiles.push_back(t0);
        }
    L.ntiles = (int)tiles.size() / 2;
    auto put = [&](const std::vector<int>& v) {
        long o = (long)p->rag_host.size();
        p->rag_host.insert(p->rag_host.end(), v.begin(), v.end());
        while (p->rag_host.size() % 4) p->rag_host.push_back(0);
        return o;
    };
    L.dT = put(L.T);
    L.doff = put(L.off);
    L.dtile = put(tiles);
}

extern "C" int avc_plan_create_ragged(const avc_model_cfg* cfg, int B, const int* T, const int* T_cond, const avc_tuning* tuning, avc_plan** out) {
    if (!cfg || !out || !T || B < 1) return fail(-1, "avc_plan_create_ragged: bad arguments");
    if (tuning && tuning->struct_size != (int)sizeof(avc_tuning)) return fail(-1, "avc_plan_create_ragged: avc_tuning of another library version (use avc_tuning_init)");
    if (!T_cond) T_cond = T;
    if (validate_enc(cfg->spk, true) || validate_enc(cfg->enc, false)) return fail(-2, "avc_plan_create_ragged: unsupported encoder config");
    const avc_decoder_cfg& dc = cfg->dec;
    if (dc.n_conv_blocks < 1 || dc.n_conv_blocks > AVC_MAX_BLOCKS || 2 * dc.n_conv_blocks > 12 || dc.kernel_size < 1 || dc.kernel_size > 8)
        return fail(-2, "avc_plan_create_ragged: unsupported decoder config");
    for (int l = 0; l < dc.n_conv_blocks; ++l)
        if (dc.upsample[l] != 1 && dc.upsample[l] != 2) return fail(-2, "avc_plan_create_ragged: upsample must be 1 or 2");
    if (cfg->spk.c_in != cfg->enc.c_in || cfg->enc.c_in != dc.c_out || dc.c_in != cfg->enc.c_out || dc.c_cond != cfg->spk.c_out)
        return fail(-2, "avc_plan_create_ragged: inconsistent channel sizes between the three networks");

    avc_plan* p = new avc_plan();
    p->cfg = *cfg;
    p->flags = AVC_PLAN_INFERENCE | AVC_PLAN_RAGGED;
    p->tun = tuning ? *tuning : avc_default_tuning();
    p->tun.ck16_wgs = -1;   // default chunk depths: the ragged launcher has the straight-line instances for those
    p->tun.ck32_wgs = -1;
    p->tun.conv_x3 = 0;
    p->B = B;
    p->M = cfg->enc.c_in;
    p->T = p->Tc = 0;
    for (int b = 0; b < B; ++b) {
        if (T[b] < 1 || T_cond[b] < 1) { delete p; return fail(-1, "avc_plan_create_ragged: lengths must be positive"); }
        p->T = T[b] > p->T ? T[b] : p->T;
        p->Tc = T_cond[b] > p->Tc ? T_cond[b] : p->Tc;
    }
    build_enc_params(p, p->spk, cfg->spk, true);
    p->enc_param_off = p->param_floats;
    build_enc_params(p, p->enc, cfg->enc, false);
    DecNet& d = p->dec;
    d.c = dc;
    d.slope = dc.act == 1 ? AVC_LRELU_SLOPE : 0.f;
    d.n = dc.n_conv_blocks;
    p->dec_param_off = p->param_floats;
    d.in_conv = add_layer(p, dc.c_h, dc.c_in, 1, 1, true);
    for (int l = 0; l < d.n; ++l) d.c1.push_back(add_layer(p, dc.c_h, dc.c_h, dc.kernel_size, 1, true));
    for (int l = 0; l < d.n; ++l) d.c2.push_back(add_layer(p, dc.c_h * dc.upsample[l], dc.c_h, dc.kernel_size, 1, true));
    {
        int a0 = add_layer(p, 2 * dc.c_h, dc.c_cond, 1, 1, false);
        LayerP& L = p->layers[a0];
        L.nsrc = 2 * d.n;
        L.rows = 2 * dc.c_h;
        L.Cout = 2 * dc.c_h * L.nsrc;
        for (int i = 1; i < L.nsrc; ++i) {
            L.w[i] = add_param(p, 2 * dc.c_h, dc.c_cond, 0);
            L.b[i] = add_param(p, 2 * dc.c_h, 0, 0);
        }
        d.affine = a0;
    }
    d.out_conv = add_layer(p, dc.c_out, dc.c_h, 1, 1, true);

    // ---- per-sample time schedules; the reference's reflect-pad rule per utterance
    auto sched = [&](EncNet& e, const int* T0, std::vector<avc_plan::RagLevel>& lv) -> int {
        std::vector<int> cur(T0, T0 + B);
        lv.resize(e.n + 1);
        for (int b = 0; b < B; ++b)
            if (e.c.bank_size / 2 >= cur[b]) return -1;
        for (int l = 0; l <= e.n; ++l) {
            rag_level(p, lv[l], cur);
            if (l == e.n) break;
            for (int b = 0; b < B; ++b) {
                if (e.c.kernel_size / 2 >= cur[b]) return -1;
                cur[b] = avc_cdiv(cur[b], e.c.subsample[l]);
            }
        }
        return 0;
    };
    bool bad = sched(p->spk, T_cond, p->rl_spk) != 0 || sched(p->enc, T, p->rl_enc) != 0;
    if (!bad) {
        std::vector<int> cur = p->rl_enc[p->enc.n].T;
        p->rl_dec.resize(d.n + 1);
        for (int l = 0; l <= d.n && !bad; ++l) {
            rag_level(p, p->rl_dec[l], cur);
            if (l == d.n) break;
            for (int b = 0; b < B; ++b) {
                if (dc.kernel_size / 2 >= cur[b]) bad = true;
                cur[b] *= dc.upsample[l];
            }
        }
    }
    if (bad) {
        delete p;
        return fail(-6, "Padding size should be less than the corresponding input dimension");
    }
    p->Tb = 0;
    p->Tout = 0;

    // ---- packed weights (forward images only); tile heuristics see the launch's column-tile count as its batch
    for (EncNet* e : {&p->spk, &p->enc}) {
        const std::vector<avc_plan::RagLevel>& lv = (e == &p->spk) ? p->rl_spk : p->rl_enc;
        for (int id : e->bank) finish_layer(p, p->layers[id], false, 0, lv[0].ntiles, 64, 64, e->nb);
        finish_layer(p, p->layers[e->in_conv], false, 0, lv[0].ntiles, 64, 64);
        for (int l = 0; l < e->n; ++l) {
            finish_layer(p, p->layers[e->c1[l]], false, 0, lv[l].ntiles, 64, 64);
            finish_layer(p, p->layers[e->c2[l]], false, 0, lv[l + 1].ntiles, 64, 64);
        }
    }
    for (int l = 0; l < p->spk.nd; ++l) {
        finish_layer(p, p->layers[p->spk.dn1[l]], false, 0, 1, B, B, 1, false);
        finish_layer(p, p->layers[p->spk.dn2[l]], false, 0, 1, B, B, 1, false);
    }
    finish_layer(p, p->layers[p->spk.outl], false, 0, 1, B, B, 1, false);
    finish_layer(p, p->layers[p->enc.heads], false, 0, p->rl_dec[0].ntiles, 64, 64);
    finish_layer(p, p->layers[d.in_conv], false, 0, p->rl_dec[0].ntiles, 64, 64);
    for (int l = 0; l < d.n; ++l) {
        finish_layer(p, p->layers[d.c1[l]], false, 0, p->rl_dec[l].ntiles, 64, 64);
        finish_layer(p, p->layers[d.c2[l]], false, 0, p->rl_dec[l].ntiles, 64, 64);
    }
    finish_layer(p, p->layers[d.affine], false, 0, 1, B, B);
    finish_layer(p, p->layers[d.out_conv], false, 0, p->rl_dec[d.n].ntiles, 64, 64);

    // ---- packed activation buffers: [channels][T_b] blocks back to back
    const long Bl = B;
    auto alloc_enc = [&](EncNet& e, const std::vector<avc_plan::RagLevel>& lv, bool spk) {
        const long C = e.c.c_h;
        e.cat = p->alloc((long)e.CC * lv[0].off[B]);
        e.h0 = p->alloc(C * lv[0].off[B]);
        e.out[0] = spk ? e.h0 : p->alloc(C * lv[0].off[B]);
        for (int l = 0; l < e.n; ++l) {
            e.a1[l] = p->alloc(C * lv[l].off[B]);
            e.out[l + 1] = p->alloc(C * lv[l + 1].off[B]);
            if (spk) {
                e.a2[l] = p->alloc(C * lv[l + 1].off[B]);
                e.y1[l] = e.y2[l] = -1;
            } else {
                e.a2[l] = -1;
                e.y1[l] = p->alloc(C * lv[l].off[B]);
                e.y2[l] = p->alloc(C * lv[l + 1].off[B]);
            }
        }
        if (spk) {
            e.pooled = p->alloc(C * Bl);
            e.hd[0] = e.pooled;
            for (int l = 0; l < e.nd; ++l) {
                e.d1[l] = p->alloc(C * Bl);
                e.d2[l] = p->alloc(C * Bl);
                e.hd[l + 1] = p->alloc(C * Bl);
            }
        }
    };
    alloc_enc(p->spk, p->rl_spk, true);
    const long Cz = dc.c_in, Cd = dc.c_h;
    p->emb = p->alloc(Bl * dc.c_cond);
    alloc_enc(p->enc, p->rl_enc, false);
    p->muls = p->alloc(2 * Cz * p->rl_dec[0].off[B]);
    d.cond = p->alloc(Bl * 2 * d.n * 2 * Cd);
    d.y0 = p->alloc(Cd * p->rl_dec[0].off[B]);
    d.out[0] = p->alloc(Cd * p->rl_dec[0].off[B]);
    for (int l = 0; l < d.n; ++l) {
        d.y1[l] = p->alloc(Cd * p->rl_dec[l].off[B]);
        d.a1[l] = p->alloc(Cd * p->rl_dec[l].off[B]);
        d.y2[l] = p->alloc(Cd * p->rl_dec[l + 1].off[B]);
        d.out[l + 1] = p->alloc(Cd * p->rl_dec[l + 1].off[B]);
    }
    p->decb = p->alloc((long)p->M * p->rl_dec[d.n].off[B]);
    p->rag_tab = p->alloc((long)p->rag_host.size());
    p->named["emb"] = p->emb;
    p->named["muls"] = p->muls;
    p->named["dec"] = p->decb;
    p->named["cond"] = d.cond;
    plan_init_streams(p);
    *out = p;
    return 0;
}

extern "C" int avc_plan_ragged_out(const avc_plan* p, int* out_len, long* out_off) {
    if (!p || !(p->flags & AVC_PLAN_RAGGED) || !out_len || !out_off) return fail(-1, "avc_plan_ragged_out: not a ragged plan");
    const avc_plan::RagLevel& L = p->rl_dec[p->dec.n];
    for (int b = 0; b < p->B; ++b) {
        out_len[b] = L.T[b];
        out_off[b] = p->decb + (long)p->M * L.off[b];
    }
    return 0;
}

static void set_rag(ConvArgs& a, const int* tab, const avc_plan::RagLevel& src, int cx, const avc_plan::RagLevel& outl, const avc_plan::RagLevel& convl,
                    int cout) {
    // src: level of the source rows; convl: level whose lengths are the conv's output lengths (tiles run over it); outl: level
    // of the packed OUTPUT buffer (== convl unless the store pixel-shuffles)
    a.rag.tile = tab + convl.dtile;
    a.rag.ntiles = convl.ntiles;
    a.rag.Tsrc = tab + src.dT;
    a.rag.offsrc = tab + src.doff;
    a.rag.Tout = tab + convl.dT;
    a.rag.offout = tab + outl.doff;
    a.rag.cx = cx;
    a.rag.cout = cout;
    a.Tout = 64;    // geometry of a one-sample column tile
    a.Tsrc = 1 << 20;   // (per sample in the tables; the reflect-pad rule was checked per utterance at plan creation)
    a.B = 1;
    a.x.sb = 0;
    a.ob = 0;
}
static void set_rag_res(ConvArgs& a, const int* tab, const float* res, int mode, const avc_plan::RagLevel& rl, int cres) {
    a.g[0].res = res;
    a.res_mode = mode;
    a.rag.Tres = tab + rl.dT;
    a.rag.offres = tab + rl.doff;
    a.rag.cres = cres;
    a.rt = 1;
}

static int rag_in(const avc_plan* p, float slope, const int* tab, const float* y, float* out, const avc_plan::RagLevel& lv, int C, const float* cond, long csb,
                  int coff, const float* res, int res_mode, const avc_plan::RagLevel* rl, hipStream_t s) {
    RagINArgs a;
    memset(&a, 0, sizeof(a));
    a.y = y; a.out = out; a.T = tab + lv.dT; a.off = tab + lv.doff;
    a.cond = cond; a.cond_sb = csb; a.cond_off = coff;
    a.res = res; a.res_mode = res ? res_mode : 0;
    if (res) { a.Tres = tab + rl->dT; a.offres = tab + rl->doff; }
    a.B = p->B; a.C = C; a.slope = slope;
    return avc_launch_rag_in_fwd(a, s);
}

static int rag_enc_front(const avc_plan* p, const EncNet& e, const std::vector<avc_plan::RagLevel>& lv, const int* tab, const float* params, float* ws,
                         const float* x, hipStream_t s) {
    // conv_bank (model.py:85-91) on the packed input [sum T][M] (frames as rows): channel stride 1, frame stride M
    const int M = e.c.c_in;
    const float SL = e.slope;
    ConvArgs a;
    memset(&a, 0, sizeof(a));
    a.x.ptr = x; a.x.sc = 1; a.x.st = M; a.x.ps = 1;
    a.Cred = M; a.mode = 0; a.stride = 1; a.bf16 = p->compute;
    a.M = e.c.c_bank; a.Mp = avc_cdiv(e.c.c_bank, 128) * 128;
    a.ot = 1; a.ops = 1; a.act = 1; a.slope = SL;
    a.ngroups = e.nb;
    a.img = AVC_IMG_K4;
    for (int g = 0; g < e.nb; ++g) {
        const LayerP& L = p->layers[e.bank[g]];
        set_group(a.g[g], ws + L.wpf, p->par(params, L.b[0]), L.KS, L.CK, L.nchunk_f);
        a.g[g].out = ws + e.cat;
        a.g[g].out_c0 = g * e.c.c_bank;
    }
    set_rag(a, tab, lv[0], M, lv[0], lv[0], e.CC);
    RUN(avc_launch_conv(a, s, 0, p->tun));
    RUN(avc_launch_rag_copy_rows(x, 1, M, tab + lv[0].dT, tab + lv[0].doff, p->B, M, lv[0].off[p->B], ws + e.cat, e.CC, e.nb * e.c.c_bank, s));
    return 0;
}

extern "C" int avc_forward_ragged(const avc_plan* p, const float* params, const float* x, const float* x_cond, float* ws, void* stream) {
    if (!p || !params || !x || !ws) return fail(-1, "avc_forward_ragged: null argument");
    if (!(p->flags & AVC_PLAN_RAGGED)) return fail(-8, "avc_forward_ragged: the plan was not created by avc_plan_create_ragged");
    if (!x_cond) x_cond = x;
    hipStream_t s = (hipStream_t)stream;
    const int B = p->B;
    // tables -> workspace (a few KB; stream-ordered in front of everything that reads them)
    RUN((int)hipMemcpyAsync(ws + p->rag_tab, p->rag_host.data(), p->rag_host.size() * sizeof(int), hipMemcpyHostToDevice, s));
    const int* tab = (const int*)(ws + p->rag_tab);
    RUN(pack_all(p, params, ws, s));   // weights -> LDS-image order (one launch)
    // A conv on packed activations: source rows of the sample's own length (x.sc = -1), output block of `cout` channels.
    auto conv = [&](float slope, const LayerP& L, const float* src, const avc_plan::RagLevel& sl, int cx, float* dst, const avc_plan::RagLevel& convl,
                    const avc_plan::RagLevel& outl, int cout, int act, int ops) {
        ConvArgs a;
        memset(&a, 0, sizeof(a));
        a.x.ptr = src; a.x.sc = -1; a.x.st = 1; a.x.ps = 1;
        a.Cred = L.Cin; a.mode = 0; a.stride = L.stride; a.bf16 = L.bf16;
        a.M = L.Cout; a.Mp = L.Mp_f;
        a.ngroups = 1;
        set_group(a.g[0], ws + L.wpf, layer_bias(p, L, params, ws), L.KS, L.CK, L.nchunk_f);
        a.img = AVC_IMG_K4;
        a.ot = 1; a.ops = ops; a.act = act; a.slope = slope;
        a.g[0].out = dst;
        set_rag(a, tab, sl, cx, outl, convl, cout);
        return a;
    };
    const hipStream_t mainS = s;
    const hipStream_t sideS = fork_side(p, mainS);
    {   // ---------------- speaker encoder on the target utterances (model.py:265-277)
        const hipStream_t s = sideS;
        const EncNet& e = p->spk;
        const float SL = e.slope;
        const std::vector<avc_plan::RagLevel>& lv = p->rl_spk;
        const int C = e.c.c_h;
        RUN(rag_enc_front(p, e, lv, tab, params, ws, x_cond, s));
        {
            ConvArgs a = conv(SL, p->layers[e.in_conv], ws + e.cat, lv[0], e.CC, ws + e.h0, lv[0], lv[0], C, 1, 1);
            RUN(avc_launch_conv(a, s, 0, p->tun));
        }
        for (int l = 0; l < e.n; ++l) {
            ConvArgs a = conv(SL, p->layers[e.c1[l]], ws + e.out[l], lv[l], C, ws + e.a1[l], lv[l], lv[l], C, 1, 1);
            RUN(avc_launch_conv(a, s, 0, p->tun));
            ConvArgs b = conv(SL, p->layers[e.c2[l]], ws + e.a1[l], lv[l], C, ws + e.a2[l], lv[l + 1], lv[l + 1], C, 1, 1);
            b.g[0].out2 = ws + e.out[l + 1];
            set_rag_res(b, tab, ws + e.out[l], e.c.subsample[l] > 1 ? AVC_RES_AVGPOOL2 : AVC_RES_IDENTITY, lv[l], C);
            RUN(avc_launch_conv(b, s, 0, p->tun));
        }
        RUN(avc_launch_rag_timepool_fwd(ws + e.out[e.n], tab + lv[e.n].dT, tab + lv[e.n].doff, B, C, ws + e.pooled, s));
        DenseArgs da;
        memset(&da, 0, sizeof(da));
        da.nlayers = 2 * e.nd + 1;
        da.B = B; da.C = C; da.slope = SL;
        da.in = ws + e.hd[0];
        da.emb = ws + p->emb;
        for (int l = 0; l < da.nlayers; ++l) {
            const bool last = (l == da.nlayers - 1);
            const LayerP& L = p->layers[last ? e.outl : ((l & 1) ? e.dn2[l / 2] : e.dn1[l / 2])];
            DenseLayer& D = da.layer[l];
            D.wp = ws + L.wpf;
            D.bias = p->par(params, L.b[0]);
            D.Cin = L.Cin; D.Cout = L.Cout; D.Kp = L.nchunk_f * L.CK; D.Mp = L.Mp_f;
            if (!last) {
                D.act = ws + ((l & 1) ? e.d2[l / 2] : e.d1[l / 2]);
                D.out2 = (l & 1) ? ws + e.hd[l / 2 + 1] : nullptr;
            }
            da.Kmax = D.Kp > da.Kmax ? D.Kp : da.Kmax;
            da.Wmax = D.Kp * D.Mp > da.Wmax ? D.Kp * D.Mp : da.Wmax;
        }
        RUN(avc_launch_dense(da, 0, s));
    }
    {   // ---------------- content encoder on the source utterances (model.py:301-323)
        const EncNet& e = p->enc;
        const float SL = e.slope;
        const std::vector<avc_plan::RagLevel>& lv = p->rl_enc;
        const int C = e.c.c_h;
        RUN(rag_enc_front(p, e, lv, tab, params, ws, x, s));
        {
            ConvArgs a = conv(SL, p->layers[e.in_conv], ws + e.cat, lv[0], e.CC, ws + e.h0, lv[0], lv[0], C, 0, 1);
            RUN(avc_launch_conv(a, s, 0, p->tun));
            RUN(rag_in(p, SL, tab, ws + e.h0, ws + e.out[0], lv[0], C, nullptr, 0, 0, nullptr, 0, nullptr, s));
        }
        for (int l = 0; l < e.n; ++l) {
            ConvArgs a = conv(SL, p->layers[e.c1[l]], ws + e.out[l], lv[l], C, ws + e.y1[l], lv[l], lv[l], C, 0, 1);
            RUN(avc_launch_conv(a, s, 0, p->tun));
            RUN(rag_in(p, SL, tab, ws + e.y1[l], ws + e.a1[l], lv[l], C, nullptr, 0, 0, nullptr, 0, nullptr, s));
            ConvArgs b = conv(SL, p->layers[e.c2[l]], ws + e.a1[l], lv[l], C, ws + e.y2[l], lv[l + 1], lv[l + 1], C, 0, 1);
            RUN(avc_launch_conv(b, s, 0, p->tun));
            RUN(rag_in(p, SL, tab, ws + e.y2[l], ws + e.out[l + 1], lv[l + 1], C, nullptr, 0, 0, ws + e.out[l],
                       e.c.subsample[l] > 1 ? AVC_RES_AVGPOOL2 : AVC_RES_IDENTITY, &lv[l], s));
        }
        ConvArgs h = conv(SL, p->layers[e.heads], ws + e.out[e.n], lv[e.n], C, ws + p->muls, lv[e.n], lv[e.n], 2 * e.c.c_out, 0, 1);
        RUN(avc_launch_conv(h, s, 0, p->tun));
    }
    join_side(p, mainS, sideS);
    {   // ---------------- decoder(mu, emb) (model.py:347-371, :387-391: no noise)
        const DecNet& d = p->dec;
        const float SL = d.slope;
        const std::vector<avc_plan::RagLevel>& lv = p->rl_dec;
        const int C = d.c.c_h, Cz = d.c.c_in;
        const long csb = (long)2 * d.n * 2 * C;
        {   // all 2n AdaIN affine Linears as ONE GEMM on emb (uniform: one row per utterance)
            ConvArgs a = mk_fwd(p, SL, p->layers[d.affine], params, ws, ws + p->emb, 0, 1, d.c.c_cond, 1, B, ws + d.cond, 0, 1, (int)csb, 0);
            RUN(avc_launch_conv(a, s, 0, p->tun));
        }
        {   // z = mu: the first Cz channels of every sample's (mu | log_sigma) block
            ConvArgs a = conv(SL, p->layers[d.in_conv], ws + p->muls, lv[0], 2 * Cz, ws + d.y0, lv[0], lv[0], C, 0, 1);
            RUN(avc_launch_conv(a, s, 0, p->tun));
            RUN(rag_in(p, SL, tab, ws + d.y0, ws + d.out[0], lv[0], C, nullptr, 0, 0, nullptr, 0, nullptr, s));
        }
        for (int l = 0; l < d.n; ++l) {
            const int up = d.c.upsample[l];
            ConvArgs a = conv(SL, p->layers[d.c1[l]], ws + d.out[l], lv[l], C, ws + d.y1[l], lv[l], lv[l], C, 0, 1);
            RUN(avc_launch_conv(a, s, 0, p->tun));
            RUN(rag_in(p, SL, tab, ws + d.y1[l], ws + d.a1[l], lv[l], C, ws + d.cond, csb, (2 * l) * 2 * C, nullptr, 0, nullptr, s));
            // second conv: C * up channels, pixel-shuffled on store into the level-(l+1) buffer (model.py:359-361)
            ConvArgs b = conv(SL, p->layers[d.c2[l]], ws + d.a1[l], lv[l], C, ws + d.y2[l], lv[l], lv[l + 1], C, 0, up);
            RUN(avc_launch_conv(b, s, 0, p->tun));
            RUN(rag_in(p, SL, tab, ws + d.y2[l], ws + d.out[l + 1], lv[l + 1], C, ws + d.cond, csb, (2 * l + 1) * 2 * C, ws + d.out[l],
                       up > 1 ? AVC_RES_UP2 : AVC_RES_IDENTITY, &lv[l], s));
        }
        ConvArgs o = conv(SL, p->layers[d.out_conv], ws + d.out[d.n], lv[d.n], C, ws + p->decb, lv[d.n], lv[d.n], p->M, 0, 1);
        RUN(avc_launch_conv(o, s, 0, p->tun));
    }
    return 0;
}
