// bf16 PAIR storage (compute_dtype "bf16": BASELINE configs[2]'s data path).
//
// An activation tensor [B, C, T] is stored as DWORDS [B][C/2][T]: dword (b, p, t) = (bf16 of channel 2p at frame t) in its low
// half, (bf16 of channel 2p+1) in its high half.  Byte for byte this has the structure of an fp32 [B][C/2][T] tensor, so every
// piece of the fp32 staging machinery applies unchanged to "dword channels": the per-lane-source dword LDS-DMA of conv_gemm.hip
// brings TWO reduction channels per lane, a 16-byte LDS fragment holds the 8 k-values one lane feeds to
// v_mfma_f32_32x32x16_bf16, an InstanceNorm row of T dwords is two channels' rows interleaved frame by frame.
#pragma once
#include "avc_common.h"

#ifndef AVC_EMU
static __device__ __forceinline__ unsigned bh_pack(float lo, float hi) {   // round to nearest even (v_cvt_pk_bf16_f32)
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    bf16x2_t v = {(__bf16)lo, (__bf16)hi};
    return __builtin_bit_cast(unsigned, v);
}
static __device__ __forceinline__ float bh_lo(unsigned d) { return __uint_as_float(d << 16); }
static __device__ __forceinline__ float bh_hi(unsigned d) { return __uint_as_float(d & 0xffff0000u); }
static __device__ __forceinline__ unsigned short bh_bits(float f) {
    __bf16 b = (__bf16)f;
    return __builtin_bit_cast(unsigned short, b);
}
#else
static inline unsigned bh_pack(float lo, float hi) {
    return ((unsigned)(unsigned short)avc_bf16_bits(lo)) | (((unsigned)(unsigned short)avc_bf16_bits(hi)) << 16);
}
static inline float bh_lo(unsigned d) { unsigned u = d << 16; float f; memcpy(&f, &u, 4); return f; }
static inline float bh_hi(unsigned d) { unsigned u = d & 0xffff0000u; float f; memcpy(&f, &u, 4); return f; }
static inline unsigned short bh_bits(float f) { return (unsigned short)avc_bf16_bits(f); }
#endif
static __device__ __forceinline__ unsigned bh_as_u32(float f) {
#ifndef AVC_EMU
    return __float_as_uint(f);
#else
    unsigned u; memcpy(&u, &f, 4); return u;
#endif
}
static __device__ __forceinline__ float bh_as_f32(unsigned u) {
#ifndef AVC_EMU
    return __uint_as_float(u);
#else
    float f; memcpy(&f, &u, 4); return f;
#endif
}
