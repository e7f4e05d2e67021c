// winograd.h -- Winograd F(4x4, 3x3) path of the folded 5x5 upsampling convolutions
// (reference models/dcgan.py:33-46: resize_nearest_neighbor + 5x5 conv, three times) and of the 5x5 stride-2
// convolutions of the critic (models/dcgan.py:12-14).
// Internal interface between conv.hip (dispatch, fold/unfold) and winograd.hip (kernels).
//
// After upsample folding each of the four output-parity classes is an ordinary 3x3 'SAME'
// convolution on the SMALL image, so the minimal-filtering form applies: per 4x4 output tile
// 36 multiplies instead of 144 (4x fewer MFMA FLOP than the folded direct form; F(2x2,3x3), the
// round-1 choice, needs 64).  Interpolation points {0, 1, -1, 1/2, -2, inf}: dyadic data / output
// transforms; measured 6.8e-7 relative L2 against fp64 on the split-precision GEMM -- between
// F(2x2,3x3) (2.2e-7) and a plain fp32 MFMA chain of the direct convolution (1.3e-6).  (Round 1
// rejected F(4x4,3x3) at 3.4e-6: textbook points {0,+-1,+-2}, fp32 filter transform, fp32 MFMA.)
#pragma once
#include "common.h"

struct WinoGeo {
  int N, H, W;      // small (stored) image
  int Cin, Cout;    // real channels (pre-activation NONE only)
  int ldx;          // channel stride of x
  int ldy, y_coff;  // output buffer [N, 2H, 2W, ldy], channel offset
  const float* x_amax = nullptr;    // amax records of x / dy when the caller has them (otgan_layers.h), else null
  const float* dy_amax = nullptr;
  // otgan_conv_desc::x_operand: the forward pass leaves its transformed input here (instead of in the workspace) and
  // the weight gradient of the same x reads it instead of transforming x again; null = each pass transforms
  float* x_op = nullptr;
  const float* w_amax = nullptr;   // otgan_conv_desc::w_amax (filters made from the un-folded weights)
  // forward only, otgan_conv_desc::glu_out / glu_amax_out: [N, 2H, 2W, Cout/2] gated output written beside y (Cout % 8 == 0)
  float* glu_out = nullptr;
  float* glu_amax = nullptr;
};

// amax record (otgan_layers.h) of x[rows][C], row stride ld
constexpr int kWinoM = 4;          // output tile edge of F(4x4, 3x3)
constexpr int kWinoFreq = 36;      // (kWinoM + 2)^2 batched GEMMs
constexpr int kWinoS2Blocks = 121; // non-zero (class, frequency) blocks of a strided layer, of 4 * 36
// tiles = N * (H/4) * (W/4)
inline long wino_tiles(const WinoGeo& g) { return (long)g.N * (g.H / kWinoM) * (g.W / kWinoM); }
// scratch floats of each pass

// y = folded conv of x with the class weights given as weffT[cls][Cout][9*Cin] (class stride cls_stride)
// `prep` (optional): the Winograd-domain filters of this pass, made once by wino_prepare_filters() and reused while
// the weights do not change (wino_filter_floats() floats); null: derived from the weights into the workspace
// dx[N,H,W,lddx] (+)= gradient w.r.t. the small input; weff[cls][9][Cin][Cout]
// dweff[cls][9][Cin][Cout] (class stride cls_stride) = folded weight gradient

// ---- 5x5 stride-2 layers (DCGAN critic, models/dcgan.py:12-14) --------------------------------
struct WinoS2Geo {
  int N, H, W;             // input image (H, W multiples of 8); output is H/2 x W/2
  int C, Ceff;             // real / effective input channels
  int doubled, act;        // CReLU/CELU doubling; 0 none, 1 relu-type, 2 elu-type
  int ldx;
  int Cout, ldy, y_coff;
  const float* x_amax = nullptr;
  const float* dy_amax = nullptr;
  int x_amax_count = 1, dy_amax_count = 1;   // consecutive records behind x_amax / dy_amax (their maximum counts)
  // 1: a 3x3 stride-1 layer instead (the block-input convolution of a DenseNet block, ops.py DenseBlockFunction):
  // ONE class on the full H x W grid (multiples of 4), taps taken as they are, nothing structurally zero;
  // wT: [Cout][9*Ceff], w: [9][Ceff][Cout].  Same kernels, same three passes.
  int plain = 0;
  // plain only, 1: x is stored at half resolution ([N, H/2, W/2, ldx]) and read through a 2x nearest-neighbour
  // upsample (the DenseNet generator's transition layers); H, W stay the grid the convolution runs on.  Weight
  // gradient (the input transform reads through the upsample) and input gradient (the output transform sums the
  // 2x2 groups of its 4x4 tile onto the stored pixel).
  int up = 0;
  int y_accumulate = 0;   // forward: y += result + bias (otgan_conv_desc::y_accumulate)
  float* x_op = nullptr;  // as in WinoGeo
  // otgan_conv_desc::y_amax_out / dx_amax_out: the output transform of the forward pass / the input gradient also
  // max-accumulates the magnitudes it writes into record[0]
  float* y_amax_out = nullptr;
  float* dx_amax_out = nullptr;
  const float* w_amax = nullptr;   // otgan_conv_desc::w_amax
};
inline int wino_s2_classes(const WinoS2Geo& g) { return g.plain ? 1 : 4; }
inline int wino_s2_out_h(const WinoS2Geo& g) { return g.plain ? g.H : g.H / 2; }
inline int wino_s2_out_w(const WinoS2Geo& g) { return g.plain ? g.W : g.W / 2; }
inline long wino_s2_tiles(const WinoS2Geo& g) { return (long)g.N * (wino_s2_out_h(g) / kWinoM) * (wino_s2_out_w(g) / kWinoM); }
// wT: [Cout][25*Ceff];  w: HWIO [25][Ceff][Cout];  single-tensor inputs only (default channel map: [act(x), act(-x)])

// ---- 3x3 convolution of a 2x nearest-neighbour upsampled image with a doubled ReLU pre-activation ([relu(x),
// relu(-x)], the reference concatenates a list input BEFORE the activation here): the DenseNet generator's
// transition layers.  Forward only: F(4x4,3x3) on the UPSAMPLED grid (2.25 products per output; the folded
// implicit GEMM: 4) with the un-folded filters, on the split-precision engine.  dgrad / wgrad keep the folded path.
struct WinoUp3Geo {
  int N, H, W;        // small (stored) image; output is 2H x 2W
  int C, Ceff;        // real / effective (2 C) input channels
  int ldx;
  int Cout, ldy, y_coff;
  const float* x_amax = nullptr;
  float* x_op = nullptr;  // as in WinoGeo (read back by the one-class weight gradient, WinoS2Geo plain + up)
  float* y_amax_out = nullptr;   // otgan_conv_desc::y_amax_out: the output transform leaves the record of y (round 4)
  const float* w_amax = nullptr; // otgan_conv_desc::w_amax (round 6: the filter transform reduced the weights itself until then)
};
inline long wino_up3_tiles(const WinoUp3Geo& g) { return (long)g.N * (2 * g.H / kWinoM) * (2 * g.W / kWinoM); }
// wT: un-folded [Cout][9 * Ceff]

// the functions of winograd.hip: one namespace per piece count (winograd_api.inc)
#ifdef WINO_NS
#include "winograd_api.inc"
#else
#define WINO_NS wino_p2
#include "winograd_api.inc"
#undef WINO_NS
#define WINO_NS wino_p3
#include "winograd_api.inc"
#undef WINO_NS
#endif
