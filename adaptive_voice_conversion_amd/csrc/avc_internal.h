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
