/* libavc_hip.so — C ABI of the MI355X-native AdaIN-VC forward/backward engine.
 *
 * The reference (jjery2243542/adaptive_voice_conversion) has no FFI of its own:
 * its hot path sits behind the torch.nn.Module `AE` (model.py:373-395) and
 * `Solver.ae_step` (solver.py:81-97).  This header is the boundary a binding
 * for that path attaches to: plain pointers, sizes and a hipStream_t; no torch
 * types.  The Python mirror of `AE`/`Solver` in adaptive_voice_conversion_amd/
 * binds it with ctypes (see INTEGRATION.md for the stub a reference maintainer
 * would add).
 *
 * Conventions
 *  - every pointer is a DEVICE pointer unless stated; tensors are fp32.
 *  - activations are [B, C, T] with explicit element strides where a view may
 *    be handed in (the collate view of data_utils.py:14-16 has strides (T*M, 1, M)).
 *  - scratch memory is a caller-owned workspace whose size the plan reports; the data-path entry points (avc_forward* / avc_loss /
 *    avc_backward / avc_clip_adam_step / avc_plan_pack_weights and every op-level call) never allocate, free or synchronise -- they only
 *    enqueue work (graph-capture safe).  Three exceptions, all outside the per-step path, stated here because earlier revisions of this
 *    header claimed there were none:
 *      (1) avc_plan_create* allocates a small DEVICE table (hipMalloc: the weight-image descriptors of avc_plan_pack_weights, a few KB)
 *          and fills it with a blocking hipMemcpy; avc_plan_destroy frees it.  Create plans outside captured / latency-critical regions.
 *      (2) the FIRST plan created on a device creates that device's three helper HIP streams (below); they live until the process ends.
 *      (3) avc_forward_ragged uploads its per-utterance length / offset / tile tables (a few KB) with hipMemcpyAsync from pageable host
 *          memory: the call may block the host until the copy is staged and is NOT graph-capture safe (the uniform entry points are).
 *  - return value: 0 = ok, < 0 = bad argument / unsupported shape,
 *    > 0 = hipError_t.  avc_last_error() describes the last failure.
 *  - the stream is always an argument (backward runs on PyTorch's autograd
 *    thread).  A plan overlaps independent branches on three helper HIP streams -- one "side" stream (the
 *    speaker-encoder branch, the decoder's second half-batch chain) and two low-priority weight-gradient streams -- and on events of
 *    its own (created by avc_plan_create*, destroyed by avc_plan_destroy); every entry point joins them back into the caller's stream
 *    before it returns.  The three helper STREAMS are ONE set per device, created by the first plan on that device, shared by every
 *    plan of the process and never destroyed (a second set lands on whatever hardware queues the runtime's round-robin has reached
 *    and was measured 1.8 - 2.4x slower: scripts/two_plans_probe.py).  The side stream's priority is the FIRST plan's
 *    avc_tuning.side_prio; later plans get that stream whatever they ask for (avc_plan_side_priority reports it).  ONE call per plan
 *    may be in flight on the host at a time (two host threads need two plans); two plans driven from two threads are functionally
 *    independent but share the helper streams, i.e. their side-branch / weight-gradient work is serialised stream by stream
 *    (tests/test_engine.py::test_two_plans_from_two_threads: results bit-equal to the serial runs).
 *  - the library has no process-wide mutable state that changes RESULTS or heuristics: the only process-lifetime objects are the helper
 *    streams above.  Launch heuristics and
 *    diagnostic switches are an `avc_tuning` value that a plan captures at
 *    creation (avc_plan_create_tuned); the op-level entry points at the end of
 *    this header read a THREAD-LOCAL avc_tuning that avc_set_tuning edits for
 *    the calling thread only (micro-benchmarks and kernel tests).  No
 *    environment variable is read by the LIBRARY.  (The Python loader, _lib.py, honours
 *    AVC_HIP_LIB = path of an alternative build of this library: which .so is loaded for
 *    same-box A/B measurements, never how it behaves.)
 *  - the gradient all-reduce of data-parallel training deliberately lives in
 *    the host framework (torch.distributed "nccl" = RCCL over xGMI):
 *    this library exposes where to cut (avc_plan_param_range) and when each
 *    part of the flat gradient buffer is final (avc_plan_stream_wait_grads).
 */
#ifndef AVC_HIP_H
#define AVC_HIP_H
#ifdef __cplusplus
extern "C" {
#endif

#define AVC_MAX_BLOCKS 8

/* config.yaml:1-24 (SpeakerEncoder / ContentEncoder blocks; n_dense_blocks = 0 for the content encoder) */
typedef struct avc_encoder_cfg {
    int c_in, c_h, c_out, kernel_size, bank_size, bank_scale, c_bank, n_conv_blocks, n_dense_blocks;
    int subsample[AVC_MAX_BLOCKS];
    int act; /* config.yaml `act` (model.py:93-99 get_act): 0 = relu, 1 = lrelu (nn.LeakyReLU, slope 0.01) */
} avc_encoder_cfg;

/* config.yaml:25-36 (Decoder) */
typedef struct avc_decoder_cfg {
    int c_in, c_cond, c_h, c_out, kernel_size, n_conv_blocks;
    int upsample[AVC_MAX_BLOCKS];
    int act; /* 0 = relu, 1 = lrelu */
} avc_decoder_cfg;

typedef struct avc_model_cfg {
    avc_encoder_cfg spk; /* model.py:209-277 */
    avc_encoder_cfg enc; /* model.py:279-323 */
    avc_decoder_cfg dec; /* model.py:325-371 */
} avc_model_cfg;

typedef struct avc_plan avc_plan; /* host-side launch plan for one (B, T, T_cond) shape */

int avc_version(void);
const char* avc_last_error(void);

/* ---- plan (host only) -------------------------------------------------- */
/* T = frames of the content/source segment, T_cond = frames of the speaker
 * (target) segment; training uses T_cond == T (model.py:380-385), inference may
 * differ (model.py:387-391).  Fails (like the reference, a1 in SURVEY §8a) when a
 * reflect pad is not smaller than the length it pads. */
int avc_plan_create(const avc_model_cfg* cfg, int B, int T, int T_cond, avc_plan** out);
/* flags: AVC_PLAN_INFERENCE = forward only (AE.inference, model.py:387-391): no gradient / slab / dy buffers in
 * the workspace, avc_loss / avc_backward are refused; AVC_PLAN_SPEAKER_ONLY (implies INFERENCE) = only the
 * speaker encoder runs (AE.get_speaker_embeddings, model.py:393-395): T is ignored, the result is ws["emb"]. */
#define AVC_PLAN_INFERENCE 1
#define AVC_PLAN_SPEAKER_ONLY 2
int avc_plan_create_ex(const avc_model_cfg* cfg, int B, int T, int T_cond, int flags, avc_plan** out);
int avc_plan_flags(const avc_plan* p);
/* priority class of the device's shared side stream this plan runs its side branch on: 1 = highest, 0 = normal, -1 = the plan has no
 * helper streams (no device at creation: every kernel goes to the caller's stream) */
int avc_plan_side_priority(const avc_plan* p);
void avc_plan_destroy(avc_plan* p);
/* flat parameter buffer: the 166 state_dict tensors in reference
 * registration order, each at a 16-byte aligned offset */
int avc_plan_num_params(const avc_plan* p);
long avc_plan_param_floats(const avc_plan* p);                                   /* total floats incl. padding */
int avc_plan_param_info(const avc_plan* p, int i, long* offset, long* numel, int dims[3]);
long avc_plan_workspace_floats(const avc_plan* p);
/* named workspace regions (float offsets): "muls" [B,2*c_out,Tb] (mu | log_sigma),
 * "emb" [B,c_cond], "dec" [B,M,T'], "z", "losses" [2], "grad_norm" [1], "d_dec", ... ; -1 if unknown */
long avc_plan_buffer(const avc_plan* p, const char* name);
int avc_plan_out_len(const avc_plan* p);    /* T' of dec: 8*ceil(T/8) for the stock config */
int avc_plan_latent_len(const avc_plan* p); /* Tb */

/* ReLU sites of the forward pass in the reference's call order (test/diagnostic aid: lets a checker
 * evaluate its own gradient on the SAME piecewise-linear branch the engine took).  kind 0: the
 * mask is (stored activation > 0) at ws[act_off] (strides sb, sc, st over [B, C, T]); kind 1
 * (InstanceNorm sites): the pre-activation is ((y - mean) * rstd) * gamma + beta with y at
 * ws[y_off] ([B,C,T] contiguous), mean/rstd at ws[stat_off] / ws[stat_off + B*C], and, when
 * cond_off >= 0, beta = ws[cond_off + b*cond_sb + c], gamma = ws[cond_off + b*cond_sb + C + c]. */
typedef struct avc_relu_site {
    int kind, B, C, T;
    long act_off, sb, sc, st;
    long y_off, stat_off, cond_off, cond_sb;
    long storage;   /* 0: fp32 tensors; AVC_PLAN_BF16S plans: 1 = bf16 pair tensor at act_off / y_off (dwords [B][C/2][T], sb / sc in dwords),
                     * 2 = natural bf16 rows [B][C][T] at y_off (the output of a pixel-shuffling conv) */
} avc_relu_site;
int avc_plan_num_relu_sites(const avc_plan* p);
int avc_plan_relu_site(const avc_plan* p, int i, avc_relu_site* out);

/* ---- data-parallel hooks: the flat gradient buffer is all-reduced by the caller (RCCL).
 * The backward pass finishes the decoder's parameter gradients first; they are the TAIL of the flat buffer.
 * avc_plan_param_range gives the float range of a part; avc_plan_stream_wait_grads makes `stream` wait until
 * that part, as written by the last avc_backward call on this plan, is final -- so the reduce of the decoder
 * range can run on a communication stream under the encoders' backward. */
#define AVC_GRADS_ALL 0
#define AVC_GRADS_DECODER 1
#define AVC_GRADS_ENCODERS 2
#define AVC_GRADS_SPEAKER 3   /* the speaker encoder's parameters (head of the flat buffer): final when its branch of the backward ends, */
#define AVC_GRADS_CONTENT 4   /* ... well before the content encoder's (the longer branch, InstanceNorm kernels in its chain) */
int avc_plan_param_range(const avc_plan* p, int part, long* offset, long* numel);
int avc_plan_stream_wait_grads(const avc_plan* p, int part, void* stream);

/* ---- launch heuristics and diagnostic switches (plan-scoped) ---------------------------------------------
 * avc_tuning_init fills the library defaults; a plan copies the struct at creation and never looks at the
 * caller's copy again.  Every field is an A/B-measurement or test aid: production code passes NULL. */
typedef struct avc_tuning {
    int struct_size;        /* sizeof(avc_tuning) of the caller (set by avc_tuning_init; a mismatch is refused) */
    int single_stream;      /* 1: every kernel on the caller's stream (profiling / per-class event brackets) */
    int dec_split_min;      /* smallest batch whose decoder forward AND backward run as two half-batch chains on two streams (128) */
    int conv_x3;            /* 1: split-bf16 conv kernel (csrc/conv_x3.hip) for the k = 5 layers that fill the chip; 2: every eligible layer */
    int wgrad_x3;           /* 1: split-bf16 products in the whole-chunk weight-gradient launches */
    int dgrad_par;          /* 0: stride-2 dgrad multiplies all taps of the zero-upsampled dy (default 1: one column parity per wave) */
    int bank_switch;        /* 0: generic run-time-taps chunk loop for the grouped bank launch / 1x1 convs (default 1) */
    int conv_ck5;           /* chunk depth of the k >= 4 convs at the op level: 8 (default) | 16 | 32 */
    int wgrad_batch;        /* weight gradients per batched launch (12) */
    int wgrad_batch_wgs;    /* workgroups a batched weight-gradient launch aims for (256) */
    int wgrad_target_wgs;   /* op level: split-K workgroups per weight-gradient launch (256) */
    int conv_ablation;      /* timing experiments only, WRONG results when set: bit0 no DMA, bit1 no MFMA, bit2 no barrier, bit3 no store */
    int wgrad_ablation;
    int op_compute_dtype;   /* op-level conv entry points: 0 fp32, 1 bf16 operands (plans: avc_plan_set_compute_dtype) */
    int side_prio;          /* 1 (default since round 5): the device's side stream (speaker-encoder branch, the longer pole of forward and backward) is
                             * created with the highest stream priority -- its kernels are dispatched ahead of the weight-gradient launches and the
                             * other branch's when CU slots free up (-2.2 % on the B = 256 step); 0: normal priority.  Honoured by the FIRST plan
                             * created on a device only: the helper streams are one set per device and process (see "Conventions") */
    int tile12_wgs;         /* 64 x 128 column tiles (two fragments per wave share each weight fragment) for stride-1 k = 5 layers whose launch still has
                             * at least this many workgroups; 0 = never */
    long wgrad_batch_units; /* pending (tile x K-chunk) units that trigger a batched launch early (1 << 40 = never) */
    long tile_thr11, tile_thr21, ck16_wgs, ck32_wgs, kg_wgs;   /* conv tile / chunk-depth / split-K-group thresholds in workgroups */
    long conv_min_lds;      /* occupancy experiment: every conv_gemm launch asks for at least this many BYTES of LDS (0 = what the tiles need) */
    long in_pairs_nv;       /* bf16 pair InstanceNorm rows: 16-byte vectors per lane, 1 (default: one lane group spans the row) | 2 | 4 */
    long bh_ck5;            /* AVC_PLAN_BF16S plans: chunk depth (DWORD channels = bf16 channel pairs) of the k >= 4 convs: 8 (default) | 16 | 32 */
    long wgrad_cw8;         /* 1: the weight gradients of the k = 5 layers (Cin % 64 == 0, Cout % 128 == 0) run on 128 x 64 tiles with EIGHT consumer
                             * waves (two MFMA waves per SIMD) + four producers.  0 (default): 64 x 64 tiles, four consumers.  Measured on MI355X,
                             * same box: the wgrad class is 2.21 vs 2.26 ms per step with it, the STEP 6.03 vs 5.98 ms (768 threads x 168 registers
                             * leave the chain's kernels no room on a CU the persistent workgroup sits on) */
    long dec_wgrad_flush;   /* decoder backward: every N recorded weight gradients go out at once, under the decoder's own (latency-bound) backward chain,
                             * as launches of only dec_wgrad_wgs persistent workgroups; 0 = all of them are held until the dense-stack backward
                             * kernel of the speaker branch has been launched (round 4's first schedule) */
    long dec_wgrad_wgs;     /* workgroups (= CUs occupied) of those early launches */
    long conv_walk;         /* conv_gemm tile walk (round 5, opt-in; exact-fp32 k = 5 / bank / 1x1 launches): a launch with at least conv_walk_min
                             * tiles per resident workgroup runs as PERSISTENT workgroups -- at most this many per CU (further bounded by LDS and
                             * registers) -- that each walk a list of column tiles, the next tile's first chunk in flight under the epilogue of
                             * this one.  Bit-identical results.  0 (default) = one tile per workgroup: measured no slower anywhere
                             * (profiles/r05_conv_walk_ablation.log).  < 0 (tests): exactly -conv_walk walkers per row slab */
    long conv_walk_min;     /* smallest number of tiles per walker worth a walk (2) */
    long conv_in_fuse;      /* 1: InstanceNorm / AdaIN / activation / residual of rows of 16 / 32 / 64 frames run inside the producing conv's epilogue
                             * (the 64-column tile holds whole rows; csrc/conv_shared.h: conv_epilogue_in) instead of a row kernel of their own */
    long dbg_streams;       /* diagnostic (scripts/bf16_repro_probe2.py): bit 0 = the backward pass launches its weight gradients on the stream that
                             * produced their operands (no weight-gradient streams); bit 1 = the backward pass runs the speaker branch and the
                             * decoder's second half-batch chain on the caller's stream (no side stream); bit 4 (16) = avc_plan_pack_weights packs image by image with
                             * the op-level gather kernel instead of the plan's one-launch table (tests/test_engine.py compares the two).  0 = the product schedule */
} avc_tuning;
void avc_tuning_init(avc_tuning* t);
/* avc_plan_create_ex with explicit tuning (NULL = defaults).  Additional flags: AVC_PLAN_X3 = compute mode "fp32x3"
 * (tuning->conv_x3 = 1, wgrad_x3 = 1 unless the caller's tuning already asks for more). */
#define AVC_PLAN_X3 4
#define AVC_PLAN_RAGGED 8   /* set by avc_plan_create_ragged (reported by avc_plan_flags) */
/* AVC_PLAN_BF16S = compute mode "bf16" with bf16 STORAGE (BASELINE configs[2]: bf16 compute, fp32 master weights and optimizer state):
 * every [B, C, T] activation and activation gradient in the workspace is a bf16 channel-pair tensor (dwords [B][C/2][T], see "bf16
 * PAIR storage" below), conv / Linear products run on v_mfma_f32_32x32x16_bf16 from bf16 pair weight images, accumulation, InstanceNorm
 * statistics, mu / log_sigma, dec, emb, the AdaIN affine parameters, all parameter gradients and the optimizer stay fp32.  Inputs and
 * outputs of the entry points keep their fp32 types.  Needs even channel counts and frame counts that are multiples of 4 at every level
 * (AVC_ERR_PAIR_SHAPE otherwise: a return code of its own, so that a caller can fall back to operand rounding -- avc_plan_set_compute_dtype(1) --
 * for THAT reason only); avc_plan_compute_dtype reports 3; avc_plan_buffer offsets of activation tensors then address pair tensors. */
#define AVC_PLAN_BF16S 16
#define AVC_ERR_PAIR_SHAPE (-12)
int avc_plan_create_tuned(const avc_model_cfg* cfg, int B, int T, int T_cond, int flags, const avc_tuning* tuning, avc_plan** out);
/* profiling aid on an existing plan: 1 = every kernel of this plan on the caller's stream */
int avc_plan_set_single_stream(avc_plan* p, int on);

/* Compute dtype of the Conv1d / Linear matrix products (BASELINE config 3: "bf16 compute, fp32 master
 * and optimizer state").  0 = fp32 MFMA, bit-exact fp32 arithmetic (default, the reference's precision);
 * 1 = operands rounded to bf16 (round-to-nearest-even) when they enter the matrix core, fp32 accumulate.
 * Parameters, activations, statistics, gradients and the optimizer stay fp32 in memory either way. */
int avc_plan_set_compute_dtype(avc_plan* p, int dtype);
int avc_plan_compute_dtype(const avc_plan* p);

/* ---- whole-model entry points (replace AE.forward / AE.inference, model.py:380-391) */
/* x: source mel [B,M,T]; x_cond: speaker mel [B,M,T_cond] (may alias x); eps: [B,c_out,Tb]
 * reparameterisation noise (model.py:383) or NULL for z = mu (AE.inference).
 * Results are left in the workspace regions "muls", "emb", "dec". */
int avc_forward(const avc_plan* p, const float* params, const float* x, long sxb, long sxc, int sxt,
                const float* x_cond, long scb, long scc, int sct, const float* eps, float* ws, void* stream);
/* The forward pass opens by re-packing every weight tensor into the plan's LDS-image order (ONE launch; the parameters change with
 * every optimizer step, solver.py:93).  A training loop moves that launch behind its optimizer step instead:
 *     avc_clip_adam_step(...); avc_plan_pack_weights(plan, params, ws, stream);          // end of step n
 *     avc_forward_ex(plan, params, ..., ws, AVC_FWD_WEIGHTS_PACKED, stream);             // step n + 1 opens with its first conv
 * AVC_FWD_WEIGHTS_PACKED is a promise that `params` has not changed since the last avc_plan_pack_weights into THIS workspace. */
#define AVC_FWD_WEIGHTS_PACKED 1
int avc_plan_pack_weights(const avc_plan* p, const float* params, float* ws, void* stream);
int avc_forward_ex(const avc_plan* p, const float* params, const float* x, long sxb, long sxc, int sxt,
                   const float* x_cond, long scb, long scc, int sct, const float* eps, float* ws, int flags, void* stream);

/* ---- ragged inference: B (source, target) pairs of DIFFERENT lengths in ONE launch set (the batched generalisation of
 * Inferencer.inference_one_utterance, inference.py:54-70; the reference converts one utterance per call).  Nothing is padded --
 * reflect padding, ceil-mode pooling and the InstanceNorm statistics all see each utterance's true length -- so result b equals
 * AE.inference(x_b, x_cond_b) (model.py:387-391).  T[b] / T_cond[b]: frames of source / target utterance b (host arrays).
 * x, x_cond: the utterances back to back as rows of frames, [sum T][M] fp32 (mel bins contiguous: the [T, M] arrays the
 * reference's utt_make_frames views).  Result: ws["dec"] holds the converted utterances back to back, utterance b as a
 * [M][out_len[b]] block (frames contiguous) at float offset out_off[b]; out_len[b] = 8 ceil(T[b] / 8) for the stock config. */
int avc_plan_create_ragged(const avc_model_cfg* cfg, int B, const int* T, const int* T_cond, const avc_tuning* tuning, avc_plan** out);
int avc_plan_ragged_out(const avc_plan* p, int* out_len, long* out_off);
int avc_forward_ragged(const avc_plan* p, const float* params, const float* x, const float* x_cond, float* ws, void* stream);

/* L1 + KL losses of solver.py:84-86 -> ws["losses"] = {loss_rec, loss_kl}; writes
 * d(lambda_rec*loss_rec)/d(dec) into ws["d_dec"] for avc_backward. */
int avc_loss(const avc_plan* p, const float* x, long sxb, long sxc, int sxt, float lambda_rec, float* ws, void* stream);

/* backward of avc_forward (autograd of model.py:380-385).  d_dec [B,M,T'] (NULL = use
 * ws["d_dec"] from avc_loss), d_muls_up [B,2*c_out,Tb] and d_emb_up [B,c_cond] are
 * upstream gradients (may be NULL); lambda_kl adds the KL term of solver.py:86-88
 * analytically (0 = none).  Writes every parameter gradient into the flat `grads`
 * buffer (same layout as params). */
int avc_backward(const avc_plan* p, const float* params, const float* x, long sxb, long sxc, int sxt,
                 const float* x_cond, long scb, long scc, int sct, const float* eps, const float* d_dec,
                 const float* d_muls_up, const float* d_emb_up, float lambda_kl, float* grads, float* ws, void* stream);

/* ---- bf16 PAIR storage (compute_dtype "bf16", BASELINE configs[2]): an activation tensor [B, C, T] is a DWORD tensor [B][C/2][T],
 * dword (b, p, t) = bf16(channel 2p) in the low half, bf16(channel 2p + 1) in the high half; statistics and accumulation stay fp32.
 * Op-level conv launches select it with avc_set_tuning("op_compute_dtype", 3) (pair tensors in and out; strides in dwords; 4 = pair
 * operands with fp32 outputs, as the heads and the decoder's last conv run); avc_pack_weight then emits bf16 pair weight images.
 * InstanceNorm / AdaIN rows (reference: model.py:296,341,77-83 and autograd): planar != 0 reads y (writes dy) as natural bf16
 * [B][C][T] rows -- the layout a pixel-shuffling conv (model.py:52-59) leaves its output in. */
int avc_instnorm_fwd_pairs(const void* y, int B, int C, int T, const float* cond, long cond_sb, int cond_off, int relu, const void* res, int res_mode,
                           int Tres, int planar, void* out, float* mean, float* rstd, void* stream);
int avc_instnorm_bwd_pairs(const void* g, const void* y, const float* mean, const float* rstd, int B, int C, int T, const float* cond, long cond_sb,
                           int cond_off, int relu, int planar, void* dy, float* dcond, long dcond_sb, int dcond_off, void* stream);
int avc_to_pairs(const float* x, long sxb, long sxc, long sxt, int B, int C, int T, void* dst, void* stream);

/* clip_grad_norm_(max_norm) + torch.optim.Adam(amsgrad, coupled L2) of solver.py:75-77,
 * :91-93 on flat buffers.  step is 1-based.  grad_prescale = 1/world_size when g holds an
 * all-reduced SUM.  ws: avc_clip_adam_ws_floats(n) floats.  gnorm_out: device float or NULL. */
long avc_clip_adam_ws_floats(long n);
int avc_clip_adam_step(float* p, float* g, float* m, float* v, float* vmax, long n, int step, float lr, float beta1,
                       float beta2, float eps, float weight_decay, int amsgrad, float max_norm, float grad_prescale,
                       int write_clipped, float* ws, float* gnorm_out, void* stream);

/* Edits ONE field (by name, as in the struct above; also "compute" = op_compute_dtype) of the CALLING THREAD's op-level tuning:
 * it affects only the op-level entry points at the end of this header when called from the same thread (micro-benchmark
 * scripts, kernel tests), never a plan.  Returns -1 for an unknown name.  avc_get_op_tuning copies the current value. */
int avc_set_tuning(const char* name, int value);
void avc_get_op_tuning(avc_tuning* out);

/* ---- device-side segment feed (replaces PickleDataset.__getitem__ + CollateFn, data_utils.py:10-22,51-54,
 * for an HBM-resident corpus): out[b, m, t] = corpus[starts[b] + t, m], corpus = [n_rows, M] fp32 (mel bins
 * contiguous), starts = B device int64 row indices, out = [B, M, T] contiguous. */
int avc_gather_segments(const float* corpus, long n_rows, int M, const long* starts, int B, int T, float* out,
                        void* stream);

/* ---- mel <-> waveform DSP (replaces preprocess/tacotron/utils.py:27-155, there librosa on the CPU)
 * A complex spectrogram is [2F][T] fp32 (row 2f = Re, 2f+1 = Im of bin f, F = n_fft/2 + 1, T contiguous); magnitudes /
 * mels are [C][T]; the normalised features the pickles and the model use are [T][C] (utils.py:84-85).
 * The transforms are GEMMs against DFT bases that carry the centre-padded periodic Hann window (built once per
 * (n_fft, win_length) by avc_dsp_make_basis into caller memory of avc_dsp_basis_floats floats; dense_scratch:
 * avc_dsp_basis_scratch_floats floats, free afterwards).  Nothing here allocates or synchronises. */
int avc_dsp_num_frames(long L, int hop_length);                      /* librosa.stft(center=True): 1 + L / hop */
long avc_dsp_basis_floats(int n_fft, int win_length, int inverse);
long avc_dsp_basis_scratch_floats(int n_fft, int win_length);
int avc_dsp_make_basis(int n_fft, int hop_length, int win_length, int inverse, float* dense_scratch, float* packed, void* stream);
/* librosa.stft(y, n_fft, hop_length, win_length) (utils.py:63-66,141): reflect padding of n_fft/2, T = 1 + L/hop frames.
 * frames_ws: win_length * T floats.  -6 when L <= n_fft/2 (the reference's call raises there). */
int avc_dsp_stft(const float* y, long L, int n_fft, int hop_length, int win_length, const float* basis_fwd, float* frames_ws,
                 float* spec, void* stream);
/* librosa.istft(spec, hop_length, win_length, window="hann") (utils.py:150-154): y has hop * (T - 1) samples. */
int avc_dsp_istft(const float* spec, int T, int n_fft, int hop_length, int win_length, const float* basis_inv, float* tf_ws, float* y,
                  void* stream);
/* griffin_lim (utils.py:136-147) on magnitudes S [F][T]: n_iter x (istft, stft, phase projection) + the final istft. */
long avc_dsp_griffin_lim_ws_floats(int T, int n_fft, int hop_length, int win_length);
int avc_dsp_griffin_lim(const float* S, int T, int n_fft, int hop_length, int win_length, int n_iter, const float* basis_fwd,
                        const float* basis_inv, float* ws, float* y, void* stream);
/* B equally long utterances at once: their frames are the columns of ONE GEMM per transform (a lone 400-frame utterance is
 * 133-238 workgroups).  y: [B][L] signals -> spec [2F][B T] with utterance b in columns b T .. b T + T - 1; S likewise [F][B T];
 * waveforms [B][hop (T - 1)].  Workspaces: the single-utterance sizes with T replaced by B T. */
int avc_dsp_stft_batch(const float* y, long L, int B, int n_fft, int hop_length, int win_length, const float* basis_fwd, float* frames_ws,
                       float* spec, void* stream);
int avc_dsp_istft_batch(const float* spec, int B, int T, int n_fft, int hop_length, int win_length, const float* basis_inv, float* tf_ws,
                        float* y, void* stream);
int avc_dsp_griffin_lim_batch(const float* S, int B, int T, int n_fft, int hop_length, int win_length, int n_iter, const float* basis_fwd,
                              const float* basis_inv, float* ws, float* y, void* stream);
/* ... and of DIFFERENT lengths: toff = B + 1 device ints (first frame of every utterance, toff[B] = Ttot), toff_host = the same
 * offsets in host memory (validated here: -1 unless monotone from 0 to Ttot, -6 unless every utterance has hop (T_b - 1) > n_fft / 2,
 * the reflect padding of its STFT; the kernels index through the device copy unchecked); S is [F][Ttot]; the waveforms come back to
 * back, utterance b (hop (T_b - 1) samples) at sample hop (toff[b] - b).  ws: avc_dsp_griffin_lim_ws_floats(Ttot, ...). */
int avc_dsp_griffin_lim_ragged(const float* S, const int* toff, const int* toff_host, int B, int Ttot, int n_fft, int hop_length, int win_length,
                               int n_iter, const float* basis_fwd, const float* basis_inv, float* ws, float* y, void* stream);
int avc_dsp_magnitude(const float* spec, int n_fft, int T, float* mag, void* stream);                                   /* utils.py:69 */
/* out[t][c] = clip((20 log10(max(1e-5, in[c][t])) - ref_db + max_db) / max_db, 1e-8, 1)   (utils.py:76-85) */
int avc_dsp_db_normalize(const float* in, int C, int T, float ref_db, float max_db, float* out, void* stream);
/* out[c][t] = 10 ^ (0.05 (clip(in[t][c], 0, 1) max_db - max_db + ref_db))                 (utils.py:92-98) */
int avc_dsp_denormalize_amp(const float* in, int C, int T, float ref_db, float max_db, float* out, void* stream);
int avc_dsp_preemphasis(const float* y, long L, float a, float* out, void* stream);        /* utils.py:60 (out != y) */
int avc_dsp_deemphasis(const float* x, long L, float a, float* out, void* stream);         /* utils.py:104 lfilter([1], [1, -a]) */
/* mean square of the frames of librosa.effects.trim (utils.py:57,107): out[1 + L/hop] */
int avc_dsp_frame_power(const float* y, long L, int frame_length, int hop_length, float* out, void* stream);

/* ---- op-level entry points (one per kernel family and direction) -------- */
long avc_packed_weight_floats(int Cout, int Cin, int KS, int dgrad);
/* W[Cout][Cin][KS] (nn.Conv1d / nn.Linear state_dict layout; nsrc tensors stacked on Cout) -> LDS-image order of
 * csrc/conv_gemm.hip: [chunk][tap][8-channel unit][lane half h][row m][k-step u], channel = 8 unit + 2 u + h (16-byte fragments) */
int avc_pack_weight(const float* const* srcs, int nsrc, int rows_per_src, int Cout, int Cin, int KS, int dgrad,
                    float* dst, void* stream);
/* weight image of the split-bf16 conv kernel (csrc/conv_x3.hip; k = 5, reduction channels a multiple of 16: every operand as
 * three bf16 terms, six bf16 MFMAs per product block, fp32-level accuracy): pass it as `wp` / `wpd` together with tile = 97.
 * Whole-model plans created with AVC_PLAN_X3 (or tuning.conv_x3) use it for their k = 5 layers that fill the chip (opt-in; the
 * default engine multiplies in exact fp32). */
long avc_packed_weight_floats_x3(int Cout, int Cin, int KS, int dgrad);
int avc_pack_weight_x3(const float* w, int Cout, int Cin, int KS, int dgrad, float* dst, void* stream);
/* pad_layer (model.py:21-32): y = act(conv1d(reflect_pad(x), W) + b) (act: 0 none, 1 ReLU, 2 LeakyReLU(0.01)); ops = pixel_shuffle_1d factor of the store
 * (model.py:52-59); res/res_mode: y2 = y + resmap(res) (1 identity, 2 avg_pool1d(2, ceil_mode) model.py:248) */
int avc_conv1d_fwd(const float* x, long sxb, long sxc, int sxt, int B, int Cin, int Tin, const float* wp,
                   const float* bias, int Cout, int KS, int stride, int act, float* out, long ob, long oc, int ot,
                   int ops, const float* res, int res_mode, long rb, long rc, int rt, int Tres, float* out2, int tile,
                   void* stream);
/* input gradient incl. the adjoint of the reflect padding; res_mode 1 identity, 3 adjoint of avg-pool,
 * 4 adjoint of nearest-x2 upsample; dx2 = dx * (mask > 0) */
int avc_conv1d_dgrad(const float* dy, long syb, long syc, int syt, int yps, int B, int Cout, int Tdy, const float* wpd,
                     int Cin, int KS, int stride, int Tin, float* dx, long ob, long oc, int ot, const float* res,
                     int res_mode, long rb, long rc, int rt, int Tres, float* dx2, const float* mask, int tile,
                     void* stream);
long avc_conv1d_wgrad_ws_floats(int B, int Cin, int Cout, int Tout, int KS);
int avc_conv1d_wgrad(const float* x, long sxb, long sxc, int sxt, const float* dy, long syb, long syc, int syt, int yps,
                     int B, int Cin, int Cout, int Tin, int Tout, int KS, int stride, float* dW, float* db, float* ws,
                     void* stream);
/* nn.InstanceNorm1d(affine=False) (model.py:296,341) [+ append_cond model.py:77-83] [+ ReLU] [+ residual:
 * 1 identity, 2 avg-pool(ceil), 5 nearest x2]; cond row b: beta = cond[b*cond_sb + cond_off + c], gamma = [.. + C + c] */
/* relu: 0 = no activation, 1 = ReLU, 2 = LeakyReLU(0.01) */
int avc_instnorm_fwd(const float* y, int B, int C, int T, const float* cond, long cond_sb, int cond_off, int relu,
                     const float* res, int res_mode, int Tres, float* out, float* mean, float* rstd, void* stream);
int avc_instnorm_bwd(const float* g, const float* y, const float* mean, const float* rstd, int B, int C, int T,
                     const float* cond, long cond_sb, int cond_off, int relu, float* dy, float* dcond, long dcond_sb,
                     int dcond_off, void* stream);
/* conv -> InstanceNorm -> [append_cond] -> act [-> + residual] of one block half (model.py:309-320 / :353-369), exact fp32:
 * y = conv1d(reflect_pad(x)) + bias, pixel-shuffled on store when ops == 2 (model.py:52-59); out / mean / rstd as avc_instnorm_fwd over the
 * rows of y.  y, out, res are contiguous [B][C][T] with C = Cout / ops, T = Tout * ops.  Where the conv's output rows are 16 / 32 / 64
 * frames long, a 64-column tile of the conv holds whole rows and the normalisation runs INSIDE the conv's epilogue (one launch;
 * avc_tuning.conv_in_fuse); otherwise the conv launch is followed by the row kernel.  *fused (may be NULL) reports which happened. */
int avc_conv1d_in_fwd(const float* x, long sxb, long sxc, int sxt, int B, int Cin, int Tin, const float* wp, const float* bias, int Cout, int KS,
                      int stride, int ops, float* y, const float* cond, long cond_sb, int cond_off, int relu, const float* res, int res_mode,
                      int Tres, float* out, float* mean, float* rstd, int* fused, void* stream);
/* ... and its autograd counterpart one layer up: g = conv1d_input_grad(dy) [+ resT(res): res_mode 1 identity, 3 pool^T, 4 up^T] is the gradient
 * wrt the OUTPUT of an InstanceNorm / AdaIN / activation layer whose saved forward rows are y / mean / rstd ([B][Cin][Tin] contiguous); the call
 * returns the gradient wrt that layer's INPUT rows in dy_out, and dbeta / dgamma in dcond (avc_instnorm_bwd's layout).  One launch where the
 * input-gradient tile holds whole rows (Tin = 16 / 32 / 64: the normalisation's backward runs in the epilogue), two otherwise.  g_out (may be
 * NULL when fused): g itself, for callers that read it again (the skip path of a block). */
int avc_conv1d_dgrad_in_bwd(const float* dy, long syb, long syc, int syt, int yps, int B, int Cout, int Tdy, const float* wpd, int Cin, int KS,
                            int stride, int Tin, float* g_out, const float* res, int res_mode, int Tres, const float* y, const float* mean,
                            const float* rstd, const float* cond, long cond_sb, int cond_off, int relu, float* dy_out, float* dcond,
                            long dcond_sb, int dcond_off, int* fused, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* AVC_HIP_H */
