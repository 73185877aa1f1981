// Micro-probe for the next round: fp32-accurate products from THREE bf16 terms per operand on the bf16 matrix core.
//   x = hi + mid + lo (each a bf16, 24 bits of mantissa together);  a*b ~= hi.hi + hi.mid + mid.hi + hi.lo + lo.hi + mid.mid
// (the dropped terms are below 2^-24 of the product), accumulated in fp32 by v_mfma_f32_32x32x16_bf16.  Six bf16 MFMAs
// cover 16 reduction steps in 6 x 8 passes; the exact-fp32 MFMA needs 8 x 16 passes for the same 16 steps -- 2.7x fewer
// matrix-pipe cycles IF the operand split (VALU) and the LDS reads keep up.  This probe measures exactly that: a wave
// reads WN x 8 fp32 B values from LDS, splits them in registers, reads WM x 3 pre-split A fragments (16 bytes each) and
// issues WM x WN x 6 MFMAs per step.  It reports fp32-EQUIVALENT TFLOP/s (2 * M * N * K per step), to be compared with
// the 157.3 TFLOP/s peak / the ~84 TFLOP/s the exact-fp32 conv kernel reaches.  Also checks the accuracy of one product.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// truncation split: hi = top 16 bits, mid = top 16 bits of the (exact) remainder, lo = the rest rounded
static __device__ __forceinline__ void split3(float x, unsigned& hi, unsigned& mid, unsigned& lo) {
    const unsigned xb = __float_as_uint(x);
    hi = xb & 0xffff0000u;
    const float r = x - __uint_as_float(hi);
    const unsigned rb = __float_as_uint(r);
    mid = rb & 0xffff0000u;
    const float r2 = r - __uint_as_float(mid);
    lo = __float_as_uint(r2) & 0xffff0000u;
}
static __device__ __forceinline__ unsigned pack_hi16(unsigned a, unsigned b) { return (a >> 16) | (b & 0xffff0000u); }   // (bf16(a), bf16(b))

template <int WM, int WN, int TERMS>
__global__ void __launch_bounds__(256) probe(const float* src, float* out, int iters) {
    __shared__ float ldsB[64 * 64];        // fp32 activations [row][64]
    __shared__ u32x4 ldsA[3 * 4 * 64];     // pre-split weights: [term][wm][lane] 16-byte fragments
    const int tid = threadIdx.x, lane = tid & 63;
    for (int e = tid; e < 64 * 64; e += 256) ldsB[e] = src[e];
    for (int e = tid; e < 3 * 4 * 64; e += 256) {
        u32x4 v = {__float_as_uint(src[e]) , __float_as_uint(src[e + 1]), __float_as_uint(src[e + 2]), __float_as_uint(src[e + 3])};
        ldsA[e] = v;
    }
    __syncthreads();
    f32x16 acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            u32x4 bt[WN][3];
#pragma unroll
            for (int j = 0; j < WN; ++j) {
                unsigned h[8], m[8], l[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) split3(ldsB[((it + u + k) & 63) * 64 + ((lane + 32 * j) & 63)], h[k], m[k], l[k]);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    bt[j][0][q] = pack_hi16(h[2 * q], h[2 * q + 1]);
                    bt[j][1][q] = pack_hi16(m[2 * q], m[2 * q + 1]);
                    bt[j][2][q] = pack_hi16(l[2 * q], l[2 * q + 1]);
                }
            }
#pragma unroll
            for (int i = 0; i < WM; ++i) {
                u32x4 at[3];
#pragma unroll
                for (int t = 0; t < 3; ++t) at[t] = ldsA[(t * 4 + ((i + u) & 3)) * 64 + lane];
#pragma unroll
                for (int j = 0; j < WN; ++j) {
                    auto mm = [&](int ta, int tb) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, at[ta]), __builtin_bit_cast(bf16x8, bt[j][tb]),
                                                                            acc[i][j], 0, 0, 0);
                    };
                    mm(0, 0);
                    if (TERMS >= 3) { mm(0, 1); mm(1, 0); }
                    if (TERMS >= 6) { mm(0, 2); mm(2, 0); mm(1, 1); }
                }
            }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    out[blockIdx.x * 256 + tid] = s;
}

// accuracy: C = A x B (32 x 32 x 16) with the 6-term split against a double-precision product
__global__ void accuracy(const float* A, const float* B, float* C, int terms) {
    const int lane = threadIdx.x;
    unsigned ah[8], am[8], al[8], bh[8], bm[8], bl[8];
    for (int k = 0; k < 8; ++k) {   // A[i = lane & 31][k = 8 (lane >> 5) + k'],  B[k][j = lane & 31]
        split3(A[(lane & 31) * 16 + 8 * (lane >> 5) + k], ah[k], am[k], al[k]);
        split3(B[(8 * (lane >> 5) + k) * 32 + (lane & 31)], bh[k], bm[k], bl[k]);
    }
    u32x4 a[3], b[3];
    for (int q = 0; q < 4; ++q) {
        a[0][q] = pack_hi16(ah[2 * q], ah[2 * q + 1]); a[1][q] = pack_hi16(am[2 * q], am[2 * q + 1]); a[2][q] = pack_hi16(al[2 * q], al[2 * q + 1]);
        b[0][q] = pack_hi16(bh[2 * q], bh[2 * q + 1]); b[1][q] = pack_hi16(bm[2 * q], bm[2 * q + 1]); b[2][q] = pack_hi16(bl[2 * q], bl[2 * q + 1]);
    }
    f32x16 c;
    for (int r = 0; r < 16; ++r) c[r] = 0.f;
    auto mm = [&](int ta, int tb) { c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[ta]), __builtin_bit_cast(bf16x8, b[tb]), c, 0, 0, 0); };
    // small terms first
    if (terms >= 6) { mm(2, 0); mm(0, 2); mm(1, 1); }
    if (terms >= 3) { mm(1, 0); mm(0, 1); }
    mm(0, 0);
    for (int r = 0; r < 16; ++r) C[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * 32 + (lane & 31)] = c[r];
}

template <int WM, int WN, int TERMS>
static void run(const char* name, const float* src, float* out, int wgs) {
    const int iters = 200;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((probe<WM, WN, TERMS>), dim3(wgs), dim3(256), 0, 0, src, out, iters);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
    }
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)wgs * 4 * iters * 4 * 2.0 * (32 * WM) * (32 * WN) * 16;
    printf("%-44s wgs=%5d  %8.3f ms  %7.1f fp32-equivalent TF\n", name, wgs, ms, flops / (ms * 1e-3) / 1e12);
    fflush(stdout);
}

int main() {
    float *src, *out, *A, *B, *C;
    hipMalloc(&src, 8192 * 4);
    hipMalloc(&out, 4096 * 256 * 4);
    std::vector<float> h(8192);
    for (int i = 0; i < 8192; ++i) h[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.f - 0.5f;
    hipMemcpy(src, h.data(), 8192 * 4, hipMemcpyHostToDevice);
    // accuracy of the split product
    hipMalloc(&A, 32 * 16 * 4); hipMalloc(&B, 16 * 32 * 4); hipMalloc(&C, 32 * 32 * 4);
    std::vector<float> ha(512), hb(512), hc(1024);
    for (int i = 0; i < 512; ++i) { ha[i] = sinf(0.37f * i) * (1 + (i % 7)); hb[i] = cosf(0.11f * i) / (1 + (i % 5)); }
    hipMemcpy(A, ha.data(), 2048, hipMemcpyHostToDevice);
    hipMemcpy(B, hb.data(), 2048, hipMemcpyHostToDevice);
    for (int terms : {1, 3, 6}) {
        hipLaunchKernelGGL(accuracy, dim3(1), dim3(64), 0, 0, A, B, C, terms);
        hipMemcpy(hc.data(), C, 4096, hipMemcpyDeviceToHost);
        double worst = 0, scale = 0, worst32 = 0;
        for (int i = 0; i < 32; ++i)
            for (int j = 0; j < 32; ++j) {
                double ref = 0;
                float f32 = 0.f;
                for (int k = 0; k < 16; ++k) { ref += (double)ha[i * 16 + k] * hb[k * 32 + j]; f32 = fmaf(ha[i * 16 + k], hb[k * 32 + j], f32); }
                worst = fmax(worst, fabs(hc[i * 32 + j] - ref));
                worst32 = fmax(worst32, fabs((double)f32 - ref));
                scale = fmax(scale, fabs(ref));
            }
        printf("accuracy, %d term(s): max |err| / max |C| = %.3e   (an fp32 fmaf chain: %.3e)\n", terms, worst / scale, worst32 / scale);
    }
    for (int wgs : {256, 512, 1024}) {
        run<1, 1, 6>("32x32 per wave, 6 terms", src, out, wgs);
        run<2, 1, 6>("64x32 per wave, 6 terms", src, out, wgs);
        run<2, 2, 6>("64x64 per wave, 6 terms", src, out, wgs);
        run<2, 2, 3>("64x64 per wave, 3 terms (16-bit mantissa)", src, out, wgs);
        run<2, 2, 1>("64x64 per wave, 1 term (plain bf16)", src, out, wgs);
    }
    return 0;
}
