// winograd.h -- Winograd F(2x2, 3x3) path of the folded 5x5 upsampling convolutions
// (reference models/dcgan.py:33-46: resize_nearest_neighbor + 5x5 conv, three times).
// Internal interface between conv.hip (dispatch, fold/unfold) and winograd.hip (kernels).
//
// After upsample folding each of the four output-parity classes is an ordinary 3x3 'SAME'
// convolution on the SMALL image, so the minimal-filtering form applies: per 2x2 output tile
// 16 multiplies instead of 36 (2.25x fewer MFMA FLOP), exact up to fp32 rounding of the
// +-1, +-1/2 transform coefficients (measured 5e-7 relative L2 against fp64, the same class
// as the direct fp32 chain; the F(4x4,3x3) variant measured 3.4e-6 and was rejected).
#pragma once
#include "common.h"

struct WinoGeo {
  int N, H, W;      // small (stored) image
  int Cin, Cout;    // real channels (pre-activation NONE only)
  int ldx;          // channel stride of x
  int ldy, y_coff;  // output buffer [N, 2H, 2W, ldy], channel offset
};

bool winograd_enabled();
// tiles = N * (H/2) * (W/2)
inline long wino_tiles(const WinoGeo& g) { return (long)g.N * (g.H / 2) * (g.W / 2); }
// scratch floats of each pass
size_t wino_fwd_ws_floats(const WinoGeo& g);
size_t wino_dgrad_ws_floats(const WinoGeo& g);
size_t wino_wgrad_ws_floats(const WinoGeo& g);

// y = folded conv of x with the class weights given as weffT[cls][Cout][9*Cin] (class stride cls_stride)
// `prep` (optional): the Winograd-domain filters of this pass, made once by wino_prepare_filters() and reused while
// the weights do not change (wino_filter_floats() floats); null: derived from the weights into the workspace
size_t wino_filter_floats(const WinoGeo& g, int which);   // which: 0 forward (from weffT), 1 dgrad (from weff)
int wino_prepare_filters(const WinoGeo& g, int which, const float* w, long cls_stride, float* out, hipStream_t s);
int wino_fwd(const WinoGeo& g, const float* x, const float* weffT, long cls_stride, const float* bias, float* y,
             float* ws, hipStream_t s, const float* prep = nullptr);
// dx[N,H,W,lddx] (+)= gradient w.r.t. the small input; weff[cls][9][Cin][Cout]
int wino_dgrad(const WinoGeo& g, const float* dy, const float* weff, long cls_stride, float* dx, int lddx,
               int accumulate, float* ws, hipStream_t s, const float* prep = nullptr);
// dweff[cls][9][Cin][Cout] (class stride cls_stride) = folded weight gradient
int wino_wgrad(const WinoGeo& g, const float* x, const float* dy, float* dweff, long cls_stride, float* ws,
               hipStream_t s);

// ---- 5x5 stride-2 layers (DCGAN critic, models/dcgan.py:12-14) --------------------------------
struct WinoS2Geo {
  int N, H, W;             // input image (H, W multiples of 4); output is H/2 x W/2
  int C, Ceff;             // real / effective input channels
  int doubled, act;        // CReLU/CELU doubling; 0 none, 1 relu-type, 2 elu-type
  int ldx;
  int Cout, ldy, y_coff;
};
inline long wino_s2_tiles(const WinoS2Geo& g) { return (long)g.N * (g.H / 4) * (g.W / 4); }
size_t wino_s2_fwd_ws_floats(const WinoS2Geo& g);
size_t wino_s2_dgrad_ws_floats(const WinoS2Geo& g);
size_t wino_s2_wgrad_ws_floats(const WinoS2Geo& g);
// wT: [Cout][25*Ceff];  w: HWIO [25][Ceff][Cout];  single-tensor inputs only (default channel map)
size_t wino_s2_filter_floats(const WinoS2Geo& g, int which);   // which: 0 forward (from wT), 1 dgrad (from w)
int wino_s2_prepare_filters(const WinoS2Geo& g, int which, const float* w, float* out, hipStream_t s);
int wino_s2_fwd(const WinoS2Geo& g, const float* x, const float* wT, const float* bias, float* y, float* ws,
                hipStream_t s, const float* prep = nullptr);
int wino_s2_dgrad(const WinoS2Geo& g, const float* dy, const float* w, const float* x, float* dx, int lddx,
                  int accumulate, float* ws, hipStream_t s, const float* prep = nullptr);
int wino_s2_wgrad(const WinoS2Geo& g, const float* x, const float* dy, float* dw, float* ws, hipStream_t s);
