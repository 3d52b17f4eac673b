// Split-bf16 3x3 convolution family (conv_split.hip), reached through drba_conv3x3 with cfg ids that follow the fp32
// table of conv.hip.
#pragma once
#include <stddef.h>
#include <string.h>

namespace drba {

// Two-term fp16 form of the split (conv_split.hip "Two-term form"): activations are pre-scaled by 2^-kSplitActShift
// before the split (undone exactly in the epilogue), so that they stay finite up to 65504 * 2^kSplitActShift.
constexpr int kSplitActShift = 4;

// the PL 16-bit terms of a weight as the kernels expect them: bf16 h, m, l with x = h + m + l (PL = 3), or fp16 h and
// (x - h) * 2^11 (PL = 2); round-to-nearest-even at every step (host side of every split family's pack function)
static inline void split_weight_terms(float x, int PL, unsigned short *bits) {
  if (PL == 3) {
    float r = x;
    for (int t = 0; t < 3; ++t) {
      unsigned u;
      memcpy(&u, &r, 4);
      u += 0x7fffu + ((u >> 16) & 1u);
      u &= 0xffff0000u;
      float h;
      memcpy(&h, &u, 4);
      bits[t] = (unsigned short)(u >> 16);
      r -= h;
    }
  } else {
    const _Float16 h = (_Float16)x;
    const _Float16 l = (_Float16)((x - (float)h) * 2048.f);
    memcpy(&bits[0], &h, 2);
    memcpy(&bits[1], &l, 2);
  }
}

int conv_split_num_cfgs();   // ids 0 .. n-1: three-term bf16 split; ids n .. 2n-1: the same tiles, two-term fp16 split
int conv_split_f16_first();  // = n
int conv_split_s2_first();   // = 2n: then conv_split_s2_num_cfgs() stride-2 tiles in the two-term form (any Cin; H, W = the INPUT map)
int conv_split_s2_num_cfgs();
int conv_split_cs_first();     // behind the stride-2 tiles: stride-1 two-term tiles whose waves split rows and couts (round 6)
int conv_split_cs_num_cfgs();
int conv_split_cfg_stride(int id);  // 1 or 2, for any id of this file's table
bool conv_split_supports(int Cin, int Cout, int id);  // stride 1, Cin a multiple of 32
size_t conv_split_packed_floats(int Cin, int Cout, int id);
int conv_split_pack(const float *w, float *packed, int Cin, int Cout, int id);
int conv_split_launch(int id, const float *in, const float *packed_w, const float *bias, const float *beta,
                      const float *residual, const float *residual2, float *out, int N, int Cin, int H, int W, int Cout,
                      int act, float post_slope, int pre_act, float pre_slope, void *stream, int pixel_shuffle = 0);
// (pixel_shuffle: MODE 0 tiles that carry the PixelShuffle(2) store form -- the two-term 4 x 32 x 64 ones -- else DRBA_EUNSUPPORTED)


// the same arithmetic with every operand streamed by LDS-DMA and the activations split on the way into the MFMAs
// (conv_dma.hip); cfg ids follow the conv_split family
int conv_dma_num_cfgs();   // n three-term ids, then n two-term ones (conv_dma_f16_first)
int conv_dma_f16_first();
bool conv_dma_supports(int Cin, int Cout, int id);
size_t conv_dma_packed_floats(int Cin, int Cout, int id);
int conv_dma_pack(const float *w, float *packed, int Cin, int Cout, int id);
int conv_dma_launch(int id, const float *in, const float *packed_w, const float *bias, const float *beta,
                    const float *residual, const float *residual2, float *out, int N, int Cin, int H, int W, int Cout,
                    int act, float post_slope, int pre_act, float pre_slope, void *stream);

// the same per-wave program with the K dimension split across the waves of a workgroup, for multi-chunk layers on small
// maps (conv_ks.hip); cfg ids follow the conv_dma family
int conv_ks_num_cfgs();   // n three-term ids, then n two-term ones (conv_ks_f16_first)
int conv_ks_f16_first();
bool conv_ks_supports(int Cin, int Cout, int id);
size_t conv_ks_packed_floats(int Cin, int Cout, int id);
int conv_ks_pack(const float *w, float *packed, int Cin, int Cout, int id);
int conv_ks_launch(int id, const float *in, const float *packed_w, const float *bias, const float *beta, const float *residual,
                   const float *residual2, float *out, int N, int Cin, int H, int W, int Cout, int act, float post_slope,
                   int pre_act, float pre_slope, void *stream);

// transposed convolution 4x4 s2 p1 (cfg ids after conv.hip's fp32 deconv table)
int deconv_split_num_cfgs();   // as above: n three-term ids, then n two-term ones
int deconv_split_f16_first();
int deconv_split_total_cfgs();  // 2 n + the two-term tiles whose waves split rows and couts (appended, round 6)
bool deconv_split_supports(int Cin, int Cout, int id);
size_t deconv_split_packed_floats(int Cin, int Cout, int id);
int deconv_split_pack(const float *w, float *packed, int Cin, int Cout, int id);
int deconv_split_launch(int id, const float *in, const float *packed_w, const float *bias, float *out, int N, int Cin, int H,
                        int W, int Cout, int pixel_shuffle, int pre_act, float pre_slope, void *stream);

}  // namespace drba
