// dense16.h -- LDS-free MFMA kernels for the DenseNet growth layers (3x3, stride 1, 16 output
// channels; reference utils/nn.py:243-262 `dense_block`, models/densenet.py:11-16,59-73).
// Internal interface between conv.hip (dispatch) and dense16.hip (kernels).
#pragma once
#include "common.h"

struct Dense16Geo {
  int N, H, W, logH, logW;  // stored spatial size (powers of two)
  int C;                    // real input channels (multiple of 8)
  int Ceff;                 // effective channels (2C for CReLU/CELU)
  int doubled;              // 1: [x, -x] channel doubling
  int act;                  // 0 none, 1 relu-type, 2 elu-type
  int ldx;                  // channel stride of the input buffer
  const int32_t* cmap;      // effective channel -> source channel | sign<<31 (nullable)
};

bool dense16_enabled();
// y[pix, coff + n] = bias[n] + sum_{tap, e} act(+-x[pix + tap, c(e)]) * wT[n][tap*Ceff + e]
int dense16_fwd(const Dense16Geo& g, const float* x, const float* wT, const float* bias, float* y,
                int ldy, int coff, hipStream_t s);
