// Split-bf16 3x3 convolution family (conv_split.hip), reached through drba_conv3x3 with cfg ids that follow the fp32
// table of conv.hip.
#pragma once
#include <stddef.h>

namespace drba {

int conv_split_num_cfgs();
bool conv_split_supports(int Cin, int Cout, int id);  // stride 1, Cin a multiple of 32
size_t conv_split_packed_floats(int Cin, int Cout, int id);
int conv_split_pack(const float *w, float *packed, int Cin, int Cout, int id);
int conv_split_launch(int id, const float *in, const float *packed_w, const float *bias, const float *beta,
                      const float *residual, const float *residual2, float *out, int N, int Cin, int H, int W, int Cout,
                      int act, float post_slope, int pre_act, float pre_slope, void *stream);


// the same arithmetic with every operand streamed by LDS-DMA and the activations split on the way into the MFMAs
// (conv_dma.hip); cfg ids follow the conv_split family
int conv_dma_num_cfgs();
bool conv_dma_supports(int Cin, int Cout, int id);
size_t conv_dma_packed_floats(int Cin, int Cout, int id);
int conv_dma_pack(const float *w, float *packed, int Cin, int Cout, int id);
int conv_dma_launch(int id, const float *in, const float *packed_w, const float *bias, const float *beta,
                    const float *residual, const float *residual2, float *out, int N, int Cin, int H, int W, int Cout,
                    int act, float post_slope, int pre_act, float pre_slope, void *stream);

// the same per-wave program with the K dimension split across the waves of a workgroup, for multi-chunk layers on small
// maps (conv_ks.hip); cfg ids follow the conv_dma family
int conv_ks_num_cfgs();
bool conv_ks_supports(int Cin, int Cout, int id);
size_t conv_ks_packed_floats(int Cin, int Cout, int id);
int conv_ks_pack(const float *w, float *packed, int Cin, int Cout, int id);
int conv_ks_launch(int id, const float *in, const float *packed_w, const float *bias, const float *beta, const float *residual,
                   const float *residual2, float *out, int N, int Cin, int H, int W, int Cout, int act, float post_slope,
                   int pre_act, float pre_slope, void *stream);

// transposed convolution 4x4 s2 p1 (cfg ids after conv.hip's fp32 deconv table)
int deconv_split_num_cfgs();
bool deconv_split_supports(int Cin, int Cout, int id);
size_t deconv_split_packed_floats(int Cin, int Cout, int id);
int deconv_split_pack(const float *w, float *packed, int Cin, int Cout, int id);
int deconv_split_launch(int id, const float *in, const float *packed_w, const float *bias, float *out, int N, int Cin, int H,
                        int W, int Cout, int pixel_shuffle, int pre_act, float pre_slope, void *stream);

}  // namespace drba
