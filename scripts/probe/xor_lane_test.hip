// standalone check of avc_xor_get<O> on the GPU: every lane must receive lane ^ O
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include "avc_common.h"
__global__ void k(float* out) {
    const float v = (float)threadIdx.x;
    out[0 * 64 + threadIdx.x] = avc_xor_get<1>(v);
    out[1 * 64 + threadIdx.x] = avc_xor_get<2>(v);
    out[2 * 64 + threadIdx.x] = avc_xor_get<4>(v);
    out[3 * 64 + threadIdx.x] = avc_xor_get<8>(v);
    out[4 * 64 + threadIdx.x] = avc_xor_get<16>(v);
    out[5 * 64 + threadIdx.x] = avc_xor_get<32>(v);
    out[6 * 64 + threadIdx.x] = avc_group_sum<32>(v);
    out[7 * 64 + threadIdx.x] = avc_group_sum<64>(v);
}
int main() {
    float* d; hipMalloc(&d, 8 * 64 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    float h[8 * 64]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 6; ++i) for (int l = 0; l < 64; ++l) if ((int)h[i * 64 + l] != (l ^ (1 << i))) { ++bad; if (bad < 10) printf("xor %d lane %d: got %d\n", 1 << i, l, (int)h[i * 64 + l]); }
    for (int l = 0; l < 64; ++l) { int g = l / 32; float e = 0; for (int k2 = 0; k2 < 32; ++k2) e += g * 32 + k2; if (h[6 * 64 + l] != e) ++bad; }
    for (int l = 0; l < 64; ++l) if (h[7 * 64 + l] != 2016.f) ++bad;
    printf("avc_xor_get / avc_group_sum: %s (%d mismatches)\n", bad ? "FAIL" : "ok", bad);
    return bad ? 1 : 0;
}
