// Micro-probe: what fraction of the fp32-MFMA rate (v_mfma_f32_32x32x2_f32) does an in-order wave sustain with
// N accumulators, LDS operand reads and filler VALU between the MFMAs?  (design input for conv_rs.hip / conv_wgrad.hip)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC, int READS, int VALU, int RND>
__global__ void __launch_bounds__(256) probe(const float* src, float* out, int iters) {
    __shared__ float lds[8192];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int e = tid; e < 8192; e += 256) lds[e] = src[e];
    __syncthreads();
    f32x16 acc[NACC];
#pragma unroll
    for (int k = 0; k < NACC; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
    float a = src[lane], b = src[64 + lane], f = 1.0f;
    const float* p = lds + lane;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 20; ++u) {
            float bb = b;
            if (READS) {
#pragma unroll
                for (int r = 0; r < READS; ++r) bb += p[((it + u) & 15) * 64 + r * 1024];
            }
#pragma unroll
            for (int v = 0; v < VALU; ++v) f = f * 1.0001f + 0.5f;
            acc[u % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, READS ? bb : b, acc[u % NACC], 0, 0, 0);
        }
        if (VALU) a += f * 1e-30f;
    }
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < NACC; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[k][r];
    out[blockIdx.x * 256 + tid] = s + f;
}

template <int NACC, int READS, int VALU, int RND>
static void run(const char* name, const float* src, float* out, int wgs) {
    const int iters = 400;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((probe<NACC, READS, VALU, RND>), dim3(wgs), dim3(256), 0, 0, src, out, iters);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
    }
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)wgs * 4 * iters * 20 * 2.0 * 32 * 32 * 2;
    printf("%-34s wgs=%5d  %8.3f ms  %7.1f TF\n", name, wgs, ms, flops / (ms * 1e-3) / 1e12);
    fflush(stdout);
}

int main(int argc, char** argv) {
    float *src, *out;
    hipMalloc(&src, 8192 * 4);
    hipMalloc(&out, 4096 * 256 * 4);
    std::vector<float> h(8192);
    for (int zero = 0; zero < 2; ++zero) {
        for (int i = 0; i < 8192; ++i) h[i] = zero ? 0.f : (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.f - 0.5f;
        hipMemcpy(src, h.data(), 8192 * 4, hipMemcpyHostToDevice);
        printf("---- operands: %s\n", zero ? "zeros" : "random");
        for (int wgs : {256, 512, 1024}) {
            run<1, 0, 0, 0>("1 acc, pure", src, out, wgs);
            run<2, 0, 0, 0>("2 acc, pure", src, out, wgs);
            run<4, 0, 0, 0>("4 acc, pure", src, out, wgs);
            run<1, 1, 0, 0>("1 acc + 1 ds_read", src, out, wgs);
            run<4, 1, 0, 0>("4 acc + 1 ds_read", src, out, wgs);
            run<1, 1, 2, 0>("1 acc + 1 ds_read + 2 valu", src, out, wgs);
            run<2, 1, 2, 0>("2 acc + 1 ds_read + 2 valu", src, out, wgs);
            run<4, 1, 2, 0>("4 acc + 1 ds_read + 2 valu", src, out, wgs);
            run<5, 1, 2, 0>("5 acc + 1 ds_read + 2 valu", src, out, wgs);
            run<4, 3, 4, 0>("4 acc + 3 ds_read + 4 valu", src, out, wgs);
            run<1, 3, 4, 0>("1 acc + 3 ds_read + 4 valu", src, out, wgs);
        }
    }
    return 0;
}
