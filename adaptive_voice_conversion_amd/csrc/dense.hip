// Fused dense stack of the speaker encoder (reference: SpeakerEncoder.dense_blocks +
// output_layer, model.py:252-263,274-276, and their autograd).
//
// The stack is 2*n_dense_blocks + 1 Linear layers on a [B, c_h] matrix: ~0.03 MFLOP per sample
// and layer, i.e. pure launch/latency cost when issued as 13 (+27 backward) separate GEMM
// launches.  Here ONE workgroup carries a tile of 32 samples through every layer: activations stay
// in LDS ([channel][sample], so the MFMA B fragment is a conflict-free row read), each layer's
// packed weight image ([k][m], the same LDS-image order the conv kernels use) is pulled in by
// 16-byte LDS-DMA, and only the tensors the backward pass needs are written to HBM.
//   forward : h <- relu(W2 relu(W1 h + b1) + b2) + h   (x n_dense), emb = Wo h + bo
//   backward: dz2 = dH * [d2 > 0]; dz1 = (W2^T dz2) * [d1 > 0]; dH <- W1^T dz1 + dH   (reversed)
// Weight gradients stay separate (they are plain GEMMs over the batch and run on the wgrad stream).
#include <hip/hip_runtime.h>

#include "avc_common.h"
#include "avc_internal.h"

#define DS_NS 32  // samples per workgroup (one 32-wide MFMA column block)

// out[m][n] (+)= sum_k Wl[k][m] * Hin[k][n] for the m-blocks of this wave; result left in acc[]
template <int MAXMB>
static __device__ __forceinline__ void dense_gemm(f32x16 (&acc)[MAXMB], const float* Wl, int Mp, const float* Hin, int Kp,
                                                  int nmb, int wave, int li, int h) {
#pragma unroll
    for (int i = 0; i < MAXMB; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    // Kp is a multiple of 32: 8 k-steps per unrolled group so that the 16 LDS fragment reads of a
    // group are in flight together instead of one exposed LDS latency per MFMA
    for (int s0 = 0; s0 < Kp / 2; s0 += 8) {
        float bv[8], av[8][MAXMB];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int k = 2 * (s0 + u) + h;
            bv[u] = Hin[k * DS_NS + li];
#pragma unroll
            for (int i = 0; i < MAXMB; ++i) {
                const int mb = wave + 4 * i;
                av[u][i] = (mb < nmb) ? Wl[k * Mp + mb * 32 + li] : 0.f;
            }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < MAXMB; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u][i], bv[u], acc[i], 0, 0, 0);
    }
}

static __device__ __forceinline__ void dense_load_w(const float* wp, float* Wl, int floats, int wave, int lane) {
    for (int piece = wave; piece * 256 < floats; piece += 4) avc_glds16(wp + piece * 256 + lane * 4, Wl + piece * 256);
}

template <int MAXMB>
__global__ void __launch_bounds__(AVC_THREADS) dense_stack_fwd_kernel(const DenseArgs a) {
    HIP_DYNAMIC_SHARED(float, smem)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, h = lane >> 5;
    const int b0 = blockIdx.x * DS_NS;
    const int HB = a.Kmax * DS_NS;  // floats per activation buffer
    float* Wl = smem;
    float* Hb = smem + a.Wmax;      // three rotating activation buffers
    for (int e = tid; e < 3 * HB; e += AVC_THREADS) Hb[e] = 0.f;
    __syncthreads();
    // input: pooled [C][B] channel-major
    for (int e = tid; e < a.C * DS_NS; e += AVC_THREADS) {
        int c = e / DS_NS, n = e - c * DS_NS;
        if (b0 + n < a.B) Hb[c * DS_NS + n] = a.in[(long)c * a.B + b0 + n];
    }
    int cur = 0;  // buffer holding h_l
    f32x16 acc[MAXMB];
    for (int l = 0; l < a.nlayers; ++l) {
        const DenseLayer L = a.layer[l];
        __syncthreads();  // previous layer's readers of Wl / writers of H are done
        dense_load_w(L.wp, Wl, L.Kp * L.Mp, wave, lane);
        __syncthreads();  // (the DMA is drained before the barrier)
        const bool last = (l == a.nlayers - 1);
        const bool second = !last && (l & 1);
        // block layer 1: in = cur, out = cur+1 ; layer 2: in = cur+1, out = cur+2 (+ residual cur)
        const float* Hin = Hb + ((second ? cur + 1 : cur) % 3) * HB;
        float* Hout = Hb + ((second ? cur + 2 : cur + 1) % 3) * HB;
        const float* Hres = Hb + (cur % 3) * HB;
        const int nmb = avc_cdiv(L.Cout, 32);
        dense_gemm<MAXMB>(acc, Wl, L.Mp, Hin, L.Kp, nmb, wave, li, h);
        const int n = li, b = b0 + n;
#pragma unroll
        for (int i = 0; i < MAXMB; ++i) {
            const int mb = wave + 4 * i;
            if (mb >= nmb) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (m >= L.Cout) continue;
                float v = acc[i][r] + L.bias[m];
                if (last) {
                    if (b < a.B) a.emb[(long)b * L.Cout + m] = v;  // emb [B][c_out] row-major
                } else {
                    v = avc_act(v, a.slope);
                    if (b < a.B) L.act[(long)m * a.B + b] = v;      // d1 / d2 (ReLU outputs, masks of the backward)
                    if (second) {
                        v += Hres[m * DS_NS + n];
                        if (b < a.B) L.out2[(long)m * a.B + b] = v;  // h_{l+1}
                    }
                    Hout[m * DS_NS + n] = v;
                }
            }
        }
        if (second) cur = (cur + 2) % 3;
    }
}

template <int MAXMB>
__global__ void __launch_bounds__(AVC_THREADS) dense_stack_bwd_kernel(const DenseArgs a) {
    HIP_DYNAMIC_SHARED(float, smem)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, h = lane >> 5;
    const int b0 = blockIdx.x * DS_NS;
    const int HB = a.Kmax * DS_NS;
    float* Wl = smem;
    float* Hb = smem + a.Wmax;
    for (int e = tid; e < 3 * HB; e += AVC_THREADS) Hb[e] = 0.f;
    __syncthreads();
    // input: d(emb) [c_out][B] channel-major (+ optional upstream [B][c_out]) -> [c][n]
    const int Co = a.layer[a.nlayers - 1].Cout;
    for (int e = tid; e < Co * DS_NS; e += AVC_THREADS) {
        int c = e / DS_NS, n = e - c * DS_NS;
        if (b0 + n < a.B) {
            float v = a.in[(long)c * a.B + b0 + n];
            if (a.in2) v += a.in2[(long)(b0 + n) * Co + c];
            Hb[c * DS_NS + n] = v;
        }
    }
    // buffer roles: X = gradient fed to the current layer (dz), G = dH carried along the residual path
    int xi = 0, gi = 1;
    f32x16 acc[MAXMB];
    for (int l = a.nlayers - 1; l >= 0; --l) {
        const DenseLayer L = a.layer[l];   // L.wp = dgrad image [Kp = Cout][Mp >= Cin]
        __syncthreads();
        dense_load_w(L.wp, Wl, L.Kp * L.Mp, wave, lane);
        __syncthreads();
        const bool last = (l == a.nlayers - 1);   // output layer
        const bool second = !last && (l & 1);      // second Linear of a block (applied after the first in forward)
        const float* Xin = Hb + xi * HB;
        float* G = Hb + gi * HB;
        const int free_i = 3 - xi - gi;
        float* Xout = Hb + free_i * HB;
        const int nmb = avc_cdiv(L.Cin, 32);
        dense_gemm<MAXMB>(acc, Wl, L.Mp, Xin, L.Kp, nmb, wave, li, h);
        const int n = li, b = b0 + n;
#pragma unroll
        for (int i = 0; i < MAXMB; ++i) {
            const int mb = wave + 4 * i;
            if (mb >= nmb) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (m >= L.Cin) continue;
                float v = acc[i][r];
                const bool ok = b < a.B;
                if (last || !second) {
                    // output layer or FIRST Linear of a block: v (+ residual dH) is the new dH
                    if (!last) v += G[m * DS_NS + n];
                    G[m * DS_NS + n] = v;   // each element is read and written by the same lane only
                    if (l == 0) {
                        if (ok) a.dpooled[(long)m * a.B + b] = v;
                    } else {
                        // gradient entering the second Linear of the previous block: dH * [d2 > 0]
                        const DenseLayer Lp = a.layer[l - 1];
                        float mk = ok ? Lp.act[(long)m * a.B + b] : 0.f;
                        float dz = ok ? avc_act_grad(v, mk > 0.f, a.slope) : 0.f;
                        if (ok) Lp.dz[(long)m * a.B + b] = dz;
                        Xout[m * DS_NS + n] = dz;
                    }
                } else {
                    // SECOND Linear: v = W2^T dz2 -> dz1 = v * [d1 > 0]
                    const DenseLayer Lp = a.layer[l - 1];
                    float mk = ok ? Lp.act[(long)m * a.B + b] : 0.f;
                    float dz = ok ? avc_act_grad(v, mk > 0.f, a.slope) : 0.f;
                    if (ok) Lp.dz[(long)m * a.B + b] = dz;
                    Xout[m * DS_NS + n] = dz;
                }
            }
        }
        if (last) {
            // dH now lives in buffer gi; X for the next layer in free_i
            xi = free_i;
        } else if (!second) {
            xi = free_i;   // dz2 of the previous block
        } else {
            xi = free_i;   // dz1 of this block; G untouched
        }
    }
}

// --------------------------------------------------------------------------
int avc_launch_dense(const DenseArgs& a, int backward, hipStream_t s) {
    if (a.nlayers < 1 || a.nlayers > AVC_DENSE_MAXL) return -1;
    int maxc = a.C;
    for (int l = 0; l < a.nlayers; ++l) {
        maxc = a.layer[l].Cin > maxc ? a.layer[l].Cin : maxc;
        maxc = a.layer[l].Cout > maxc ? a.layer[l].Cout : maxc;
    }
    if (maxc > 512) return -2;
    size_t lds = (size_t)(a.Wmax + 3 * a.Kmax * DS_NS) * 4 + 16;
    if (lds > 158 * 1024) return -3;
    dim3 grid(avc_cdiv(a.B, DS_NS)), block(AVC_THREADS);
    ProfScope ps(backward ? AVC_K_CONV_DGRAD : AVC_K_CONV_FWD, 0.0, 0.0, s);
    const int nmb = avc_cdiv(maxc, 32);
    if (!backward) {
        if (nmb <= 4) hipLaunchKernelGGL((dense_stack_fwd_kernel<1>), grid, block, lds, s, a);
        else hipLaunchKernelGGL((dense_stack_fwd_kernel<4>), grid, block, lds, s, a);
    } else {
        if (nmb <= 4) hipLaunchKernelGGL((dense_stack_bwd_kernel<1>), grid, block, lds, s, a);
        else hipLaunchKernelGGL((dense_stack_bwd_kernel<4>), grid, block, lds, s, a);
    }
    return (int)hipGetLastError();
}
