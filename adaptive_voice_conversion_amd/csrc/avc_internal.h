// Internal (non-ABI) launcher prototypes shared by the .hip translation units.
#pragma once
#include <string.h>

#include "avc_common.h"
#include "avc_hip.h"

struct PackArgs {
    const float* src[12];
    int nsrc, rows_per_src;  // forward output channels per source tensor
    int Cout, Cin, KS;       // stacked forward shape
    int dgrad, CK, nchunk, M, Mp;
    float* dst;
    int img;      // AVC_IMG_*
    int mb;       // one-launch table only: rows per staged piece (avc_pack_stage_rows), 0 = legacy pieces
};

// library defaults / the calling thread's op-level tuning (ops_api.hip)
const avc_tuning& avc_default_tuning();
avc_tuning& avc_op_tuning();
long avc_pack_total(const PackArgs& p);

int avc_conv_ck(const avc_tuning& tun, int KS);
int avc_conv_pick_tile(const avc_tuning& tun, int Mp, int B, int Tout, int ngroups, int Kred);   // Kred = reduction channels x taps
long avc_conv_num_wgs(int tile, int Mp, int B, int Tout, int ngroups);
int avc_conv_ck_for(const avc_tuning& tun, int KS, long wgs, int mode, int stride, int Tout, int tile);
int avc_launch_conv(const ConvArgs& a, hipStream_t stream, int force_tile, const avc_tuning& tun);
bool avc_conv_in_fusable(const ConvArgs& a, const avc_tuning& tun, int res_mode, int Tres);
bool avc_conv_inb_fusable(const ConvArgs& a, const avc_tuning& tun);   // ConvINBwd: may the InstanceNorm backward of this dgrad launch's output rows run in its epilogue?   // ConvINFuse: may the InstanceNorm of this launch's output rows run in its epilogue?
int avc_launch_pack(const PackArgs& p, hipStream_t stream);
// split-bf16 conv (conv_x3.hip): ConvArgs.img == AVC_IMG_X3
bool avc_conv_x3_eligible(const avc_tuning& tun, int mode, int Cred, int KS, int stride, int Tout, int B, int M);
long avc_conv_x3_image_floats(int M, int Cred, int KS);
void avc_pack_x3_args(PackArgs& p, const float* w, int Cout, int Cin, int KS, int dgrad, float* dst, int M_rows = 0);
int avc_launch_conv_x3(const ConvArgs& a, hipStream_t stream, const avc_tuning& tun);
// mel <-> waveform DSP (dsp.hip)
int avc_launch_dsp_basis(int which, int n_fft, int win, float* W, hipStream_t s);
int avc_launch_dsp_frames(const float* y, long L, int B, int T, int hop, int n_fft, int win, float* frames, hipStream_t s);
int avc_launch_dsp_ola(const float* tf, int B, int T, int hop, int n_fft, int win, float* y, hipStream_t s);
int avc_launch_dsp_frames_ragged(const float* y, const int* toff, int B, int Ttot, int hop, int n_fft, int win, float* frames, hipStream_t s);
int avc_launch_dsp_ola_ragged(const float* tf, const int* toff, int B, int Ttot, int hop, int n_fft, int win, float* y, hipStream_t s);
int avc_launch_dsp_phase(const float* est, const float* S, int F, int T, float* out, hipStream_t s);
int avc_launch_dsp_mag(const float* spec, int F, int T, float* mag, hipStream_t s);
int avc_launch_dsp_db_norm(const float* in, int C, int T, float ref_db, float max_db, float* out, hipStream_t s);
int avc_launch_dsp_denorm_amp(const float* in, int C, int T, float ref_db, float max_db, float* out, hipStream_t s);
int avc_launch_dsp_preemph(const float* y, long L, float a, float* out, hipStream_t s);
int avc_launch_dsp_deemph(const float* x, long L, float a, float* out, hipStream_t s);
int avc_launch_dsp_frame_power(const float* y, long L, int frame_length, int hop, int n_frames, float* out, hipStream_t s);
#define AVC_PACK_BATCH 16
int avc_launch_pack_batch(const PackArgs* ps, int n, hipStream_t stream);
#define AVC_PACK_PIECE 2048   // image elements per block of the one-launch pack (8 per thread): legacy pieces
#define AVC_PACK_STAGE 8192   // floats of LDS stage per block: staged pieces
int avc_pack_stage_rows(const PackArgs& p);
long avc_pack_pieces(const PackArgs& p);
int avc_launch_pack_table(const PackArgs* dev_tab, const void* dev_blk, int nblk, double bytes, const float* params, float* ws, hipStream_t stream);

void avc_wgrad_geometry(WgradArgs& a);
void avc_wgrad_plan_batch(WgradArgs* layers, int n, int target_wgs);
int avc_launch_wgrad_batch(const WgradArgs* layers, int n, hipStream_t stream, int ablation = 0);   // the stream-K launches + the batch's reduce launch

int avc_launch_in_fwd(const INFwdArgs& a, hipStream_t s);
int avc_launch_in_bwd(const INBwdArgs& a, hipStream_t s);
// bf16 pair rows (rowops_pairs.hip): a.R = B * C / 2 dword rows
int avc_launch_in_fwd_pairs(const INFwdArgs& a, hipStream_t s);
int avc_launch_in_bwd_pairs(const INBwdArgs& a, hipStream_t s);
int avc_launch_to_pairs(const float* x, long sxb, long sxc, long sxt, int B, int C, int T, float* dst, long db, long dc, hipStream_t s);
int avc_launch_reparam_fwd_pairs(const float* muls, const float* eps, int B, int C, int Tb, float* z, hipStream_t s);
int avc_launch_timepool_fwd_pairs(const float* in, int B, int C, int T, float* out, hipStream_t s);
int avc_launch_timepool_bwd_pairs(const float* dP, const float* amask, int B, int C, int T, float* G, float* dy, float slope, hipStream_t s);
int avc_launch_rag_in_fwd(const RagINArgs& a, hipStream_t s);
int avc_launch_rag_copy_rows(const float* x, long xsc, long xst, const int* T, const int* off, int B, int M, int sumT, float* dst, int CC, int c0,
                             hipStream_t s);
int avc_launch_rag_timepool_fwd(const float* in, const int* T, const int* off, int B, int C, float* out, hipStream_t s);
int avc_launch_copy_rows(const float* x, long sxb, long sxc, int sxt, int B, int M, int T, float* dst, long db, long dc,
                         hipStream_t s);
int avc_launch_timepool_fwd(const float* in, int B, int C, int T, float* out, hipStream_t s);
int avc_launch_timepool_bwd(const float* dP, const float* amask, int B, int C, int T, float* G, float* dy, float slope, hipStream_t s);
int avc_launch_reparam_fwd(const float* muls, const float* eps, int B, int C, int Tb, float* z, hipStream_t s);
int avc_launch_latent_bwd(const float* muls, const float* eps, const float* dz, const float* dmuls_up, int B, int C,
                          int Tb, float lambda_kl_over_n, float* dmuls, hipStream_t s);
int avc_loss_blocks(long n);
int avc_launch_loss(const float* dec, const float* x, long sxb, long sxc, int sxt, int B, int M, int T, const float* muls,
                    int C, int Tb, float lambda_rec, float* ddec, float* partial, float* losses, hipStream_t s);
int avc_adam_blocks(long n);
int avc_launch_sumsq(const float* g, long n, float* partial, hipStream_t s);
int avc_launch_clip_adam(const AdamArgs& a, hipStream_t s);
struct avc_plan;
int avc_backward_impl(const avc_plan*, const float*, const float*, long, long, int, const float*, long, long, int,
                      const float*, const float*, const float*, const float*, float, float*, float*, hipStream_t, bool,
                      long*);

// ---- optional per-class event timing (prof.hip)
enum { AVC_K_CONV_FWD = 0, AVC_K_CONV_DGRAD, AVC_K_CONV_WGRAD, AVC_K_REDUCE, AVC_K_IN_FWD, AVC_K_IN_BWD, AVC_K_PACK,
       AVC_K_ADAM, AVC_K_MISC, AVC_K_NCLASS };
bool avc_prof_on();
// named points of a backward pass (prof.hip: avc_prof_marks_begin / _end): 0 backward starts, 1 dense-stack backward issued (side stream),
// 2 the speaker branch's first chain kernel issued (side), 3 content branch chain done (main), 4 speaker branch chain done (side),
// 5 decoder weight gradients done (wgrad stream 0), 6 everything joined (main)
#define AVC_PROF_NMARK 8
void avc_prof_mark(int id, hipStream_t s);
struct ProfScope {
    ProfScope(int cls, double flops, double bytes, hipStream_t s);
    ~ProfScope();
    bool active_;
    hipStream_t s_;
};
int avc_launch_dense(const DenseArgs& a, int backward, hipStream_t s);
int avc_launch_gather_segments(const float* corpus, long n_rows, int M, const long* starts, int B, int T, float* out,
                               hipStream_t s);
int avc_launch_add_transposed(float* dst, const float* src, int B, int C, hipStream_t s);

