// Fused dense stack of the speaker encoder (reference: SpeakerEncoder.dense_blocks +
// output_layer, model.py:252-263,274-276, and their autograd).
//
// The stack is 2*n_dense_blocks + 1 Linear layers on a [B, c_h] matrix: ~0.03 MFLOP per sample
// and layer, i.e. pure launch/latency cost when issued as 13 (+27 backward) separate GEMM
// launches -- and it sits on the step's critical chain in BOTH passes (forward: the decoder waits
// for emb; backward: it opens the speaker branch, the longer pole).  ONE workgroup carries a tile of
// 16 samples through every layer, and everything in it is arranged for latency, not throughput:
//   * activations stay in LDS ([channel][sample] with the channels of every group of 8 permuted so that
//     the four k-rows one v_mfma_f32_16x16x4_f32 consumes are 64 consecutive floats: conflict-free);
//   * each layer's packed weight image ([k][m], the plain LDS-image order) is pulled in by 16-byte
//     LDS-DMA into one of TWO buffers: layer l+1's image is in flight under layer l's products.  Every
//     1-KiB DMA piece (two k-rows at Mp = 128) is padded by 64 bytes, so the four k-rows of a product
//     (k = 8t + 2q + parity for lane quarter q: four consecutive pieces) hit disjoint banks;
//   * 16x16x4 tiles: two independent accumulators per wave at c_h = 128 (32-cycle products, no
//     dependent-chain stalls), a quarter of the 32x32x2 formulation's matrix-pipe time per layer;
//   * every global LOAD of a layer (bias / ReLU masks) is issued in front of the products and every
//     global STORE behind them: loads interleaved with stores made the compiler drain the vector-memory
//     counter once per accumulator register (16 exposed round trips per layer in round 2's kernel).
//   forward : h <- relu(W2 relu(W1 h + b1) + b2) + h   (x n_dense), emb = Wo h + bo
//   backward: dz2 = dH * [d2 > 0]; dz1 = (W2^T dz2) * [d1 > 0]; dH <- W1^T dz1 + dH   (reversed)
// Weight gradients stay separate (they are plain GEMMs over the batch and run on the wgrad stream).
#include <hip/hip_runtime.h>

#include "avc_common.h"
#include "avc_internal.h"

#define DS_NS 16        // samples per workgroup (one 16-wide MFMA column block)
#define DS_PIECE 256    // floats per LDS-DMA piece (64 lanes x 16 bytes)
#define DS_PSTRIDE 272  // LDS floats per piece: 64 bytes of padding behind every piece
// s_waitcnt vmcnt(0) only (expcnt / lgkmcnt fields at their maxima = no wait)
#define DS_WAIT_VM0() __builtin_amdgcn_s_waitcnt(0 | (7 << 4) | (15 << 8))

// LDS float offset of weight-image element (k, m)
static __device__ __forceinline__ int ds_waddr(int k, int m, int Mp) {
    const int e = k * Mp + m;
    return e + ((e >> 8) << 4);
}
// LDS row of activation channel c: within every aligned group of 8, channel 2q + par sits in row 4*par + q
static __device__ __forceinline__ int ds_hpos(int c) { return (c & ~7) | ((c & 1) << 2) | ((c >> 1) & 3); }

// acc[i][r] = sum_k Wl[k][m] * Hin[k][n] with m = (wave + 4i)*16 + 4*(lane>>4) + r, n = lane & 15
template <int MAXMB>
static __device__ __forceinline__ void dense_gemm(f32x4 (&acc)[MAXMB], const float* Wl, int Mp, const float* Hin, int Kp,
                                                  int nmb, int wave, int lane) {
    const int j = lane & 15, kq = lane >> 4;
#pragma unroll
    for (int i = 0; i < MAXMB; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
    int mrow[MAXMB];
    bool mok[MAXMB];
#pragma unroll
    for (int i = 0; i < MAXMB; ++i) {
        mok[i] = (wave + 4 * i) < nmb;
        mrow[i] = mok[i] ? (wave + 4 * i) * 16 + j : 0;
    }
    const int nt = Kp >> 3;  // groups of 8 k-rows = 2 products each
    int t0 = 0;
    // 8 products per unrolled group so that its 8 + 8*MAXMB LDS fragment reads are in flight together
    for (; t0 + 4 <= nt; t0 += 4) {
        float bv[8], av[8][MAXMB];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int t = t0 + (u >> 1), par = u & 1;
            bv[u] = Hin[(8 * t + 4 * par) * DS_NS + lane];
            const int k = 8 * t + 2 * kq + par;
#pragma unroll
            for (int i = 0; i < MAXMB; ++i) {
                const float w = Wl[ds_waddr(k, mrow[i], Mp)];
                av[u][i] = mok[i] ? w : 0.f;
            }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < MAXMB; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][i], bv[u], acc[i], 0, 0, 0);
    }
    for (; t0 < nt; ++t0) {
#pragma unroll
        for (int par = 0; par < 2; ++par) {
            const float bvv = Hin[(8 * t0 + 4 * par) * DS_NS + lane];
            const int k = 8 * t0 + 2 * kq + par;
#pragma unroll
            for (int i = 0; i < MAXMB; ++i) {
                const float w = Wl[ds_waddr(k, mrow[i], Mp)];
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(mok[i] ? w : 0.f, bvv, acc[i], 0, 0, 0);
            }
        }
    }
}

static __device__ __forceinline__ void dense_load_w(const float* wp, float* Wl, int floats, int wave, int lane) {
    for (int piece = wave; piece * DS_PIECE < floats; piece += 4) avc_glds16(wp + piece * DS_PIECE + lane * 4, Wl + piece * DS_PSTRIDE);
}

template <int MAXMB>
__global__ void __launch_bounds__(AVC_THREADS) dense_stack_fwd_kernel(const DenseArgs a, const int nbuf) {
    HIP_DYNAMIC_SHARED(float, smem)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 15, kq = lane >> 4;
    const int b0 = blockIdx.x * DS_NS;
    const int b = b0 + j;
    const bool ok = b < a.B;
    const int WB = (a.Wmax / DS_PIECE) * DS_PSTRIDE;  // LDS floats per weight buffer
    const int HB = a.Kmax * DS_NS;                    // floats per activation buffer
    float* Hb = smem + nbuf * WB;                     // two activation buffers: h (block input / output), relu1 output
    for (int e = tid; e < 2 * HB; e += AVC_THREADS) Hb[e] = 0.f;
    dense_load_w(a.layer[0].wp, smem, a.layer[0].Kp * a.layer[0].Mp, wave, lane);
    __syncthreads();
    // input: pooled [C][B] channel-major
    for (int e = tid; e < a.C * DS_NS; e += AVC_THREADS) {
        const int c = e / DS_NS, n = e - c * DS_NS;
        if (b0 + n < a.B) Hb[ds_hpos(c) * DS_NS + n] = a.in[(long)c * a.B + b0 + n];
    }
    f32x4 acc[MAXMB];
    float hres[MAXMB][4];  // h_l of the block being computed, in the accumulator layout (the residual)
#pragma unroll
    for (int i = 0; i < MAXMB; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) hres[i][r] = 0.f;
    for (int l = 0; l < a.nlayers; ++l) {
        const DenseLayer& L = a.layer[l];
        float* Wl = smem + ((nbuf == 2) ? (l & 1) * WB : 0);
        if (nbuf == 1 && l > 0) {
            __syncthreads();  // every wave is done with the previous layer's image
            dense_load_w(L.wp, Wl, L.Kp * L.Mp, wave, lane);
        }
        DS_WAIT_VM0();    // this wave's share of the layer's image has landed (and its earlier stores are out)
        __syncthreads();  // ... everybody's; the previous layer's activations are complete
        const bool last = (l == a.nlayers - 1);
        const bool second = !last && (l & 1);
        const int nmb = avc_cdiv(L.Cout, 16);
        // (clamped index, no branch: a load inside a conditional block makes the compiler drain the counter again at every later join)
        float bias[MAXMB][4];
#pragma unroll
        for (int i = 0; i < MAXMB; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = (wave + 4 * i) * 16 + 4 * kq + r;
                bias[i][r] = L.bias[m < L.Cout ? m : L.Cout - 1];
            }
        // next layer's image into the other buffer (its last readers passed the barrier above)
        if (nbuf == 2 && l + 1 < a.nlayers)
            dense_load_w(a.layer[l + 1].wp, smem + ((l + 1) & 1) * WB, a.layer[l + 1].Kp * a.layer[l + 1].Mp, wave, lane);
        // block layer 1: h -> r ; layer 2: r -> h (+ residual h, held in registers) ; output layer: h -> emb
        const float* Hin = second ? Hb + HB : Hb;
        float* Hout = second ? Hb : Hb + HB;
        if (!last && !second) {
#pragma unroll
            for (int i = 0; i < MAXMB; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = (wave + 4 * i) * 16 + 4 * kq + r;
                    hres[i][r] = ((wave + 4 * i) < nmb && m < L.Cout) ? Hb[ds_hpos(m) * DS_NS + j] : 0.f;
                }
        }
        dense_gemm<MAXMB>(acc, Wl, L.Mp, Hin, L.Kp, nmb, wave, lane);
        // straight-line first: everything that waits for a load (ONE vector-memory wait per layer) ...
        float v1[MAXMB][4], v2[MAXMB][4];
#pragma unroll
        for (int i = 0; i < MAXMB; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float v = acc[i][r] + bias[i][r];
                v1[i][r] = last ? v : avc_act(v, a.slope);
                v2[i][r] = v1[i][r] + hres[i][r];
            }
        // ... then the stores
#pragma unroll
        for (int i = 0; i < MAXMB; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = (wave + 4 * i) * 16 + 4 * kq + r;
                const bool valid = (wave + 4 * i) < nmb && m < L.Cout;
                if (last) {
                    if (valid && ok) a.emb[(long)b * L.Cout + m] = v1[i][r];  // emb [B][c_out] row-major
                } else {
                    if (valid && ok) L.act[(long)m * a.B + b] = v1[i][r];                 // d1 / d2 (ReLU outputs, masks of the backward)
                    if (valid && ok && second) L.out2[(long)m * a.B + b] = v2[i][r];      // h_{l+1}
                    if (valid) Hout[ds_hpos(m) * DS_NS + j] = second ? v2[i][r] : v1[i][r];
                }
            }
    }
}

template <int MAXMB>
__global__ void __launch_bounds__(AVC_THREADS) dense_stack_bwd_kernel(const DenseArgs a, const int nbuf) {
    HIP_DYNAMIC_SHARED(float, smem)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 15, kq = lane >> 4;
    const int b0 = blockIdx.x * DS_NS;
    const int b = b0 + j;
    const bool ok = b < a.B;
    const int WB = (a.Wmax / DS_PIECE) * DS_PSTRIDE;
    const int HB = a.Kmax * DS_NS;
    float* Hb = smem + nbuf * WB;  // two buffers: the gradient fed to the current layer (dz) ping-pongs between them
    for (int e = tid; e < 2 * HB; e += AVC_THREADS) Hb[e] = 0.f;
    {
        const DenseLayer& L0 = a.layer[a.nlayers - 1];   // L.wp = dgrad image [Kp = Cout][Mp >= Cin]
        dense_load_w(L0.wp, smem, L0.Kp * L0.Mp, wave, lane);
    }
    __syncthreads();
    // input: d(emb) [c_out][B] channel-major (+ optional upstream [B][c_out]) -> [c][n]
    const int Co = a.layer[a.nlayers - 1].Cout;
    for (int e = tid; e < Co * DS_NS; e += AVC_THREADS) {
        const int c = e / DS_NS, n = e - c * DS_NS;
        if (b0 + n < a.B) {
            float v = a.in[(long)c * a.B + b0 + n];
            if (a.in2) v += a.in2[(long)(b0 + n) * Co + c];
            Hb[ds_hpos(c) * DS_NS + n] = v;
        }
    }
    int xi = 0;
    f32x4 acc[MAXMB];
    float G[MAXMB][4];  // dH carried along the residual path: every element belongs to one lane, it never leaves the registers
#pragma unroll
    for (int i = 0; i < MAXMB; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) G[i][r] = 0.f;
    for (int l = a.nlayers - 1; l >= 0; --l) {
        const DenseLayer& L = a.layer[l];
        const int idx = a.nlayers - 1 - l;
        float* Wl = smem + ((nbuf == 2) ? (idx & 1) * WB : 0);
        if (nbuf == 1 && idx > 0) {
            __syncthreads();
            dense_load_w(L.wp, Wl, L.Kp * L.Mp, wave, lane);
        }
        DS_WAIT_VM0();
        __syncthreads();
        const bool last = (l == a.nlayers - 1);   // output layer
        const bool second = !last && (l & 1);      // second Linear of a block (applied after the first in forward)
        const int nmb = avc_cdiv(L.Cin, 16);
        // ReLU outputs of the layer below (the masks of the gradient this layer hands down)
        const DenseLayer& below = a.layer[l > 0 ? l - 1 : 0];
        float mk[MAXMB][4];
#pragma unroll
        for (int i = 0; i < MAXMB; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = (wave + 4 * i) * 16 + 4 * kq + r;
                // (clamped, no lane-dependent branch.  l == 0 hands down d(pooled) unmasked -- and with n_dense_blocks = 0 the only layer is the
                // output layer, which HAS no ReLU output to read: the test on l is wave-uniform)
                mk[i][r] = (l > 0) ? below.act[(long)(m < L.Cin ? m : L.Cin - 1) * a.B + (ok ? b : a.B - 1)] : 0.f;
            }
        if (nbuf == 2 && l > 0)
            dense_load_w(a.layer[l - 1].wp, smem + ((idx + 1) & 1) * WB, a.layer[l - 1].Kp * a.layer[l - 1].Mp, wave, lane);
        const float* Xin = Hb + xi * HB;
        float* Xout = Hb + (xi ^ 1) * HB;
        dense_gemm<MAXMB>(acc, Wl, L.Mp, Xin, L.Kp, nmb, wave, lane);
        // straight-line first (the ONE vector-memory wait of the layer: the masks), then the stores
        float dzv[MAXMB][4];
#pragma unroll
        for (int i = 0; i < MAXMB; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = acc[i][r];
                if (last || !second) {
                    // output layer or FIRST Linear of a block: v (+ residual dH) is the new dH
                    if (!last) v += G[i][r];
                    G[i][r] = v;
                }
                // gradient entering the layer below: dH * [d2 > 0] (below = second Linear of the previous block),
                // or (W2^T dz2) * [d1 > 0] (below = first Linear of this block); at l == 0: d(pooled) = dH
                dzv[i][r] = (l == 0) ? v : (ok ? avc_act_grad(v, mk[i][r] > 0.f, a.slope) : 0.f);
            }
#pragma unroll
        for (int i = 0; i < MAXMB; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = (wave + 4 * i) * 16 + 4 * kq + r;
                const bool valid = (wave + 4 * i) < nmb && m < L.Cin;
                if (l == 0) {
                    if (valid && ok) a.dpooled[(long)m * a.B + b] = dzv[i][r];
                } else {
                    if (valid && ok) below.dz[(long)m * a.B + b] = dzv[i][r];
                    if (valid) Xout[ds_hpos(m) * DS_NS + j] = dzv[i][r];
                }
            }
        xi ^= 1;
    }
}

// --------------------------------------------------------------------------
int avc_launch_dense(const DenseArgs& a, int backward, hipStream_t s) {
    if (a.nlayers < 1 || a.nlayers > AVC_DENSE_MAXL) return -1;
    int maxc = a.C;
    for (int l = 0; l < a.nlayers; ++l) {
        maxc = a.layer[l].Cin > maxc ? a.layer[l].Cin : maxc;
        maxc = a.layer[l].Cout > maxc ? a.layer[l].Cout : maxc;
        if (a.layer[l].Kp % 8 || (a.layer[l].Kp * a.layer[l].Mp) % DS_PIECE) return -2;
    }
    if (a.Kmax % 8 || a.Wmax % DS_PIECE) return -2;
    if (maxc > 128) return -3;  // (two 16-row blocks per wave; a wider image would not fit the LDS either)
    const size_t WB = (size_t)(a.Wmax / DS_PIECE) * DS_PSTRIDE, HB = (size_t)a.Kmax * DS_NS;
    const size_t limit = 158 * 1024;
    int nbuf = 2;  // next layer's weight image in flight under this layer's products, when two images fit
    size_t lds = (2 * WB + 2 * HB) * 4 + 16;
    if (lds > limit) {
        nbuf = 1;
        lds = (WB + 2 * HB) * 4 + 16;
    }
    if (lds > limit) return -3;
    dim3 grid(avc_cdiv(a.B, DS_NS)), block(AVC_THREADS);
    ProfScope ps(backward ? AVC_K_CONV_DGRAD : AVC_K_CONV_FWD, 0.0, 0.0, s);
    if (!backward) hipLaunchKernelGGL((dense_stack_fwd_kernel<2>), grid, block, lds, s, a, nbuf);
    else hipLaunchKernelGGL((dense_stack_bwd_kernel<2>), grid, block, lds, s, a, nbuf);
    return (int)hipGetLastError();
}
