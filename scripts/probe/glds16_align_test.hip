// Does global_load_lds_dwordx4 (gfx950) accept a global source address that is only DWORD aligned?  (The bf16 weight gradient's run-time-taps
// instance wants to stage x rows starting at frame t0 - padL, i.e. 16-byte pieces whose source is shifted by padL dwords.)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include "avc_common.h"
__global__ void k(const float* src, float* out, int shift) {
    __shared__ __attribute__((aligned(16))) float lds[256];
    const int lane = threadIdx.x;
    avc_glds16(src + shift + 4 * lane, lds);   // lane l fetches dwords shift + 4 l .. + 3 into LDS dwords 4 l .. 4 l + 3
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    for (int i = lane; i < 256; i += 64) out[i] = lds[i];
}
int main() {
    float h[512], *d, *o, r[256];
    for (int i = 0; i < 512; ++i) h[i] = (float)i;
    hipMalloc(&d, sizeof(h)); hipMalloc(&o, sizeof(r));
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    int bad_total = 0;
    for (int shift = 0; shift < 8; ++shift) {
        hipMemset(o, 0, sizeof(r));
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, shift);
        hipError_t e = hipDeviceSynchronize();
        hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
        int bad = 0;
        for (int i = 0; i < 256; ++i) bad += r[i] != (float)(i + shift);
        printf("source shifted by %d dwords (%2d bytes): %s (%d of 256 dwords wrong, first %g %g %g %g) %s\n", shift, 4 * shift, bad ? "WRONG" : "ok", bad, r[0], r[1], r[2], r[3], e == hipSuccess ? "" : hipGetErrorString(e));
        bad_total += bad;
    }
    return bad_total ? 1 : 0;
}
