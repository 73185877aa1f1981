// Lane-level CPU simulator of the subset of HIP used by csrc/*.hip.
//
// TEST INFRASTRUCTURE ONLY.  There is no GPU in the build container, so the
// kernel *logic* (index maps, LDS layouts, MFMA fragment maps, barriers) is
// debugged by compiling the very same .hip sources for the host with
//     clang++ -x c++ -include tests/emu/hip_emu.h ...
// Every HIP thread becomes a fiber; __syncthreads / wave shuffles / MFMA are
// rendezvous points.  The resulting libavc_emu.so is loaded only by tests/ —
// the product loader (adaptive_voice_conversion_amd/_lib.py) never looks at it.
#pragma once
#define AVC_EMU 1
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>

struct dim3 {
    unsigned x, y, z;
    constexpr dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
extern dim3 threadIdx, blockIdx, blockDim, gridDim;

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __shared__ static
#define __launch_bounds__(...)

typedef int hipError_t;
typedef void* hipStream_t;
#define hipSuccess 0
static inline hipError_t hipGetLastError() { return 0; }
static inline hipError_t hipPeekAtLastError() { return 0; }
static inline const char* hipGetErrorString(hipError_t) { return "emu"; }
static inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return 0; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, int, hipStream_t) { memcpy(d, s, n); return 0; }
static inline hipError_t hipMalloc(void** p, size_t n) { *p = malloc(n); return *p ? 0 : 2; }
static inline hipError_t hipFree(void* p) { free(p); return 0; }
static inline hipError_t hipMemcpy(void* d, const void* s, size_t n, int) { memcpy(d, s, n); return 0; }
#define hipMemcpyDeviceToDevice 3
#define hipMemcpyHostToDevice 1

struct float2 { float x, y; };
struct float4 { float x, y, z, w; };
struct int2 { int x, y; };
struct int4 { int x, y, z, w; };
static inline float2 make_float2(float x, float y) { return {x, y}; }
static inline float4 make_float4(float x, float y, float z, float w) { return {x, y, z, w}; }
static inline int4 make_int4(int x, int y, int z, int w) { return {x, y, z, w}; }

namespace emu {
void launch(dim3 grid, dim3 block, size_t smem, const std::function<void()>& body);
void block_barrier();
void wave_sync();
void* dyn_smem();
void* wave_buf();  // 64 x 16-byte slots shared by the current wave
int lane();
int wave_lanes();

template <class T>
static inline T exchange(T v, int src_lane) {
    static_assert(sizeof(T) <= 16, "slot");
    char* buf = (char*)wave_buf();
    memcpy(buf + 16 * lane(), &v, sizeof(T));
    wave_sync();
    T r;
    int n = wave_lanes();
    if (src_lane < 0 || src_lane >= n) src_lane = lane();
    memcpy(&r, buf + 16 * src_lane, sizeof(T));
    wave_sync();
    return r;
}

typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));

// v_mfma_f32_32x32x2_f32: A[i=l&31][k=l>>5], B[k=l>>5][j=l&31],
// C/D: col = l&31, row = (r&3) + 8*(r>>2) + 4*(l>>5)   (cdna_hip_programming.md §3)
static inline f32x16_t mfma_32x32x2(float a, float b, f32x16_t c) {
    float2* buf = (float2*)wave_buf();  // 16-byte slots -> use as float2 at stride 2
    int l = lane();
    buf[2 * l] = make_float2(a, b);
    wave_sync();
    int col = l & 31, hi = l >> 5;
    for (int r = 0; r < 16; ++r) {
        int row = (r & 3) + 8 * (r >> 2) + 4 * hi;
        float acc = c[r];
        for (int k = 0; k < 2; ++k) {
            float av = buf[2 * (row + 32 * k)].x;
            float bv = buf[2 * (col + 32 * k)].y;
            acc = fmaf(av, bv, acc);
        }
        c[r] = acc;
    }
    wave_sync();
    return c;
}
// v_mfma_f32_32x32x8_bf16 (_1k): A[i=l&31][k=4*(l>>5)+j], B[k=4*(l>>5)+j][col=l&31], j = 0..3 in the
// lane's four bf16 slots; C/D as the fp32 32x32 shape; products exact in fp32, fp32 accumulate
typedef short s16x4_t __attribute__((ext_vector_type(4)));
static inline float bf16_to_f32(short s) {
    unsigned u = ((unsigned)(unsigned short)s) << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}
static inline f32x16_t mfma_32x32x8_bf16(s16x4_t a, s16x4_t b, f32x16_t c) {
    struct Slot { s16x4_t a, b; };
    Slot* buf = (Slot*)wave_buf();  // 16-byte slots
    int l = lane();
    buf[l].a = a;
    buf[l].b = b;
    wave_sync();
    int col = l & 31, hi = l >> 5;
    for (int r = 0; r < 16; ++r) {
        int row = (r & 3) + 8 * (r >> 2) + 4 * hi;
        float acc = c[r];
        for (int k = 0; k < 8; ++k) {
            float av = bf16_to_f32(buf[row + 32 * (k >> 2)].a[k & 3]);
            float bv = bf16_to_f32(buf[col + 32 * (k >> 2)].b[k & 3]);
            acc = fmaf(av, bv, acc);
        }
        c[r] = acc;
    }
    wave_sync();
    return c;
}
// v_mfma_f32_32x32x16_bf16: A[i=l&31][k=8*(l>>5)+j], B[k=8*(l>>5)+j][col=l&31], j = 0..7 in the lane's eight bf16 slots
// (slot j = half j&1 of dword j>>1, low half first); C/D as the fp32 32x32 shape (checked on hardware by
// scripts/probe/bf16x3_probe.hip's accuracy test)
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
static inline f32x16_t mfma_32x32x16_bf16(u32x4_t a, u32x4_t b, f32x16_t c) {
    // two rounds through the 16-byte wave slots: A fragments, then B fragments
    u32x4_t* buf = (u32x4_t*)wave_buf();
    int l = lane();
    u32x4_t afr[64], bfr[64];
    buf[l] = a;
    wave_sync();
    for (int i = 0; i < 64; ++i) afr[i] = buf[i];
    wave_sync();
    buf[l] = b;
    wave_sync();
    for (int i = 0; i < 64; ++i) bfr[i] = buf[i];
    wave_sync();
    auto el = [](const u32x4_t& v, int j) { unsigned d = v[j >> 1]; return bf16_to_f32((short)((j & 1) ? (d >> 16) : (d & 0xffffu))); };
    int col = l & 31, hi = l >> 5;
    for (int r = 0; r < 16; ++r) {
        int row = (r & 3) + 8 * (r >> 2) + 4 * hi;
        float acc = c[r];
        for (int k = 0; k < 16; ++k) acc = fmaf(el(afr[row + 32 * (k >> 3)], k & 7), el(bfr[col + 32 * (k >> 3)], k & 7), acc);
        c[r] = acc;
    }
    return c;
}
// v_mfma_f32_16x16x4_f32: A[i=l&15][k=l>>4], B[k=l>>4][j=l&15], C: col=l&15,row=4*(l>>4)+r
static inline f32x4_t mfma_16x16x4(float a, float b, f32x4_t c) {
    float2* buf = (float2*)wave_buf();
    int l = lane();
    buf[2 * l] = make_float2(a, b);
    wave_sync();
    int col = l & 15, q = l >> 4;
    for (int r = 0; r < 4; ++r) {
        int row = 4 * q + r;
        float acc = c[r];
        for (int k = 0; k < 4; ++k) acc = fmaf(buf[2 * (row + 16 * k)].x, buf[2 * (col + 16 * k)].y, acc);
        c[r] = acc;
    }
    wave_sync();
    return c;
}
}  // namespace emu

// device math that libm lacks (dsp.hip)
static inline double cospi(double x) { return cos(M_PI * x); }
static inline float cospif(float x) { return (float)cos(M_PI * (double)x); }
static inline void sincospi(double x, double* s, double* c) { *s = sin(M_PI * x); *c = cos(M_PI * x); }

#define __syncthreads() emu::block_barrier()
#define __builtin_amdgcn_s_barrier() emu::block_barrier()
#define __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, x, y, z) emu::mfma_32x32x2((a), (b), (c))
#define __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, x, y, z) emu::mfma_16x16x4((a), (b), (c))

template <class T> static inline T __shfl_xor(T v, int mask, int width = 64) {
    int l = emu::lane();
    int src = l ^ mask;
    if ((src / width) != (l / width)) src = l;
    return emu::exchange(v, src);
}
template <class T> static inline T __shfl_down(T v, unsigned d, int width = 64) {
    int l = emu::lane();
    int src = l + (int)d;
    if ((src / width) != (l / width)) src = l;
    return emu::exchange(v, src);
}
template <class T> static inline T __shfl(T v, int src, int width = 64) {
    int l = emu::lane();
    return emu::exchange(v, (l / width) * width + (src % width));
}
static inline int __builtin_amdgcn_readfirstlane(int v) { return emu::exchange(v, 0); }
static inline float atomicAdd(float* p, float v) { float o = *p; *p = o + v; return o; }
static inline int atomicAdd(int* p, int v) { int o = *p; *p = o + v; return o; }
static inline unsigned atomicAdd(unsigned* p, unsigned v) { unsigned o = *p; *p = o + v; return o; }
static inline float __fmaf_rn(float a, float b, float c) { return fmaf(a, b, c); }
static inline float __fmul_rn(float a, float b) { volatile float r = a * b; return r; }
static inline float __fsub_rn(float a, float b) { volatile float r = a - b; return r; }
static inline float __fadd_rn(float a, float b) { volatile float r = a + b; return r; }
static inline float __frcp_rn(float a) { return 1.0f / a; }
static inline float __fsqrt_rn(float a) { return sqrtf(a); }
static inline float __fdividef(float a, float b) { return a / b; }
static inline void __threadfence() {}

// LDS DMA: lane l copies `size` bytes from its own global address to lds_base + l*size
#define __builtin_amdgcn_global_load_lds(g, l, size, off, aux) \
    memcpy((char*)(uintptr_t)(l) + emu::lane() * (size), (const void*)(uintptr_t)(g), (size))
template <class T> static inline bool __any(T pred) {
    int v = pred ? 1 : 0;
    for (int o = 1; o < 64; o <<= 1) v |= __shfl_xor(v, o);
    return v != 0;
}

#define HIP_DYNAMIC_SHARED(type, var) type* var = (type*)emu::dyn_smem();
#define hipLaunchKernelGGL(kernel, grid, block, smem, stream, ...) \
    emu::launch((grid), (block), (smem), [&]() { kernel(__VA_ARGS__); })

// events (used only by the optional profiler): no-ops in the simulator
typedef void* hipEvent_t;
static inline hipError_t hipEventCreate(hipEvent_t* e) { *e = nullptr; return 0; }
static inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return 0; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return 0; }
static inline hipError_t hipEventElapsedTime(float* t, hipEvent_t, hipEvent_t) { *t = 0.f; return 0; }
static inline hipError_t hipEventDestroy(hipEvent_t) { return 0; }

// streams: the simulator executes every launch synchronously, so a "side stream" is just a tag
#define hipStreamNonBlocking 1
#define hipEventDisableTiming 2
static inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = (hipStream_t)0x1; return 0; }
static inline hipError_t hipStreamDestroy(hipStream_t) { return 0; }
static inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = nullptr; return 0; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return 0; }
static inline hipError_t hipDeviceGetStreamPriorityRange(int* lo, int* hi) { *lo = 0; *hi = 0; return 0; }
static inline hipError_t hipStreamCreateWithPriority(hipStream_t* s, unsigned, int) { *s = (hipStream_t)0x2; return 0; }
#define __builtin_amdgcn_sched_barrier(x) ((void)0)
#define __builtin_amdgcn_s_waitcnt(x) ((void)0)
