// Does ds_read_b128 (gfx950) return the right 16 bytes from an LDS address that is only DWORD aligned, and at what cost?
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(float* out, long long* cyc, int shift) {
    __shared__ __attribute__((aligned(16))) float lds[64 * 44 + 64];
    for (int i = threadIdx.x; i < 64 * 44 + 64; i += 64) lds[i] = (float)i;
    __syncthreads();
    const unsigned addr = (unsigned)(size_t)(lds) + 4u * (threadIdx.x * 44 + shift);   // rows of 44 dwords, like the staged tiles
    f32x4 v, acc = {0.f, 0.f, 0.f, 0.f};
    long long t0 = clock64();
#pragma unroll 1
    for (int r = 0; r < 256; ++r) {
        asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
        acc += v;
    }
    long long t1 = clock64();
    out[4 * threadIdx.x + 0] = acc[0] / 256.f; out[4 * threadIdx.x + 1] = acc[1] / 256.f;
    out[4 * threadIdx.x + 2] = acc[2] / 256.f; out[4 * threadIdx.x + 3] = acc[3] / 256.f;
    if (threadIdx.x == 0) *cyc = (t1 - t0) / 256;
}
int main() {
    float *o, r[256]; long long *c, hc;
    hipMalloc(&o, sizeof(r)); hipMalloc(&c, 8);
    int bad_total = 0;
    for (int shift = 0; shift < 8; ++shift) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, o, c, shift);
        hipError_t e = hipDeviceSynchronize();
        hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost); hipMemcpy(&hc, c, 8, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int l = 0; l < 64; ++l) for (int i = 0; i < 4; ++i) bad += r[4 * l + i] != (float)(l * 44 + shift + i);
        printf("LDS address shifted by %d dwords: %s (%d wrong; lane 0 got %g %g %g %g), %lld cycles per dependent read %s\n", shift, bad ? "WRONG" : "ok", bad, r[0], r[1], r[2], r[3], hc, e == hipSuccess ? "" : hipGetErrorString(e));
        bad_total += bad;
    }
    return bad_total ? 1 : 0;
}
