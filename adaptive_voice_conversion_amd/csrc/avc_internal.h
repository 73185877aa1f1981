// Internal (non-ABI) launcher prototypes shared by the .hip translation units.
#pragma once
#include <string.h>

#include "avc_common.h"

struct PackArgs {
    const float* src[12];
    int nsrc, rows_per_src;  // forward output channels per source tensor
    int Cout, Cin, KS;       // stacked forward shape
    int dgrad, CK, nchunk, M, Mp;
    float* dst;
};

extern "C" int avc_conv_ck(int KS);
int avc_launch_conv(const ConvArgs& a, hipStream_t stream, int force_tile);
int avc_launch_pack(const PackArgs& p, hipStream_t stream);

void avc_wgrad_plan(int B, int Cin, int Cout, int Tout, int* Tc, int* spc, int* chunks_per_sample, int* total_chunks,
                    int* chunks_per_wg, int* nsplit);
int avc_launch_wgrad(const WgradArgs& a, int nsplit, hipStream_t stream);
int avc_launch_reduce(const float* slab, long stride, int nsplit, int n, float* dst, hipStream_t stream);
