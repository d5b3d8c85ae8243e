// Parameters of the attention kernels (attention.hip, attention_x6.hip): strides in elements.
#pragma once
#include "aed_common.h"

struct AttnParams {
    const float* q; const float* k; const float* v; const float* bias; float* o;
    int Nq, Nk, H;
    int ldq, ldk, ldv, ldo, ld_bias;
    long bsq, bsk, bsv, bso;
    float scale;
};

// attention_x6.hip: the transposed-score kernel on split-bf16 MFMAs.  Returns -1 when it does not take the head dim.
int launch_attention_x6(const AttnParams& p, int B, int D, hipStream_t s);
