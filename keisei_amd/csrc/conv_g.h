// Interface of the GEMM-class 3x3 convolution (conv_g.hip) towards the dispatcher in conv3x3.hip.
#pragma once
#include "common.h"

struct ConvGArgs {
    const void* in;           // (B, 81, 256) bf16
    const void* wpack;        // fragment-ordered weights (ka_pack_conv3x3, Nout = Kin = 256)
    void* out;                // (B, 81, 256) bf16
    // fused input transform (one of): x' = relu?(x*scale + shift) + bias[b]      (forward conv2)
    //                                 x' = x*scale + shift + in2*k3, written to in_out when non-NULL (data-gradient convs)
    const float* in_scale; const float* in_shift; const float* in_bias; int relu;
    const void* in2; const float* in_k3; void* in_out;
    // statistics of the output tile, per board and channel: bsum = sum, sqpart = sum of squares
    float* bsum; float* sqpart;
    // masked epilogue (conv2's data gradient): out = acc * [ep_scale*ep_y + ep_shift > 0]; ep_s1 = sum out,
    // ep_s2 = sum out * (ep_y - ep_mean) * ep_invstd
    const void* ep_y; const float* ep_scale; const float* ep_shift; const float* ep_mean; const float* ep_invstd;
    float* ep_s1; float* ep_s2;
    int B;
};

// true when conv_g covers this launch (bf16, Cin = Cout = 256, a batch that fills the chip, a launch kind it was built for)
bool conv_g_applies(int B, int Cin, int Cout, int dtype, bool two_tensor_input);
int conv_g_run(const ConvGArgs& a, hipStream_t st);
