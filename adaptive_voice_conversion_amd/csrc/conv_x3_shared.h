// Split-bf16 operand helpers and the weight-image packer of conv_x3.hip (the pack kernels live in conv_gemm.hip).
#pragma once
#include "avc_common.h"
#include "avc_internal.h"

typedef unsigned avc_u32x4 __attribute__((ext_vector_type(4)));
#ifndef AVC_EMU
typedef __bf16 avc_bf16x8 __attribute__((ext_vector_type(8)));
static __device__ __forceinline__ f32x16 avc_mfma_bf16x8(avc_u32x4 a, avc_u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(avc_bf16x8, a), __builtin_bit_cast(avc_bf16x8, b), c, 0, 0, 0);
}
static __device__ __forceinline__ unsigned x3_bits(float x) { return __float_as_uint(x); }
static __device__ __forceinline__ float x3_float(unsigned u) { return __uint_as_float(u); }
#else
static inline f32x16 avc_mfma_bf16x8(avc_u32x4 a, avc_u32x4 b, f32x16 c) { return emu::mfma_32x32x16_bf16(a, b, c); }
static inline unsigned x3_bits(float x) { unsigned u; memcpy(&u, &x, 4); return u; }
static inline float x3_float(unsigned u) { float x; memcpy(&x, &u, 4); return x; }
#endif

// hi = the top 16 bits of x, mid = the top 16 bits of the exact remainder, lo = what is left rounded to 16 bits
static __device__ __forceinline__ void x3_split(float x, unsigned& hi, unsigned& mid, unsigned& lo) {
    hi = x3_bits(x) & 0xffff0000u;
    const float r = x - x3_float(hi);
    mid = x3_bits(r) & 0xffff0000u;
    lo = (x3_bits(r - x3_float(mid)) + 0x8000u) & 0xffff0000u;
}
#ifndef AVC_EMU
static __device__ __forceinline__ unsigned x3_pair(unsigned a, unsigned b) { return __builtin_amdgcn_perm(b, a, 0x07060302u); }   // (bf16 a, bf16 b): one v_perm_b32
#else
static inline unsigned x3_pair(unsigned a, unsigned b) { return (a >> 16) | (b & 0xffff0000u); }
#endif

// chunk geometry: KB blocks of 16 reduction channels per chunk (k = 5: one, k = 1: two) x KS taps; a chunk of the weight
// image is KS * KB * 6 rows (tap, block, term, k-half) of Mp x 8 bf16
static inline __host__ __device__ int x3_kb(int KS) { return KS == 1 ? 2 : 1; }
static inline __host__ __device__ int x3_arows(int KS) { return KS * x3_kb(KS) * 6; }

// weight image: W[Cout][Cin][KS] -> [chunk][tap][block][term][k-half][Mp][4 dwords], dword q = channels chunk*CK + 16 block + 8 h + 2q, +1
//   fwd  : value(m, c, j) = W[m][c][j]          dgrad: value(m, c, j) = W[c][m][KS-1-j]   (transposed, tap-flipped)
static __device__ __forceinline__ void avc_pack_x3_one(const PackArgs& p, long first, long stride, long limit = (1L << 62)) {
    const int KB = x3_kb(p.KS);
    long total = (long)p.nchunk * x3_arows(p.KS) * p.Mp * 4;
    total = total < limit ? total : limit;
    const float* w = p.src[0];
    const int Cred = p.dgrad ? p.Cout : p.Cin;
    unsigned* dst = (unsigned*)p.dst;
    for (long e = first; e < total; e += stride) {
        const int q = (int)(e & 3);
        long rest = e >> 2;
        const int m = (int)(rest % p.Mp);
        rest /= p.Mp;
        const int h = (int)(rest & 1);
        rest >>= 1;
        const int term = (int)(rest % 3);
        rest /= 3;
        const int kb = (int)(rest % KB);
        rest /= KB;
        const int j = (int)(rest % p.KS), chunk = (int)(rest / p.KS);
        unsigned t[2] = {0u, 0u};
        for (int u = 0; u < 2; ++u) {
            const int c = chunk * 16 * KB + 16 * kb + 8 * h + 2 * q + u;
            float v = 0.f;
            if (m < p.M && c < Cred) v = p.dgrad ? w[((long)c * p.Cin + m) * p.KS + (p.KS - 1 - j)] : w[((long)m * p.Cin + c) * p.KS + j];
            unsigned hi, mid, lo;
            x3_split(v, hi, mid, lo);
            t[u] = term == 0 ? hi : (term == 1 ? mid : lo);
        }
        dst[e] = x3_pair(t[0], t[1]);
    }
}

