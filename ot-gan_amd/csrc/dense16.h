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
// y[pix, coff + n] (+)= bias[n] + sum_{tap, e} act(+-x[pix + tap, c(e)]) * wT[n][tap*Ceff + e]
// amax_out (nullable): amax record (common.h) that max-accumulates the magnitudes of the values written
int dense16_fwd(const Dense16Geo& g, const float* x, const float* wT, const float* bias, float* y,
                int ldy, int coff, int accumulate, hipStream_t s, float* amax_out = nullptr);

// ---- the chains of a split dense block on two scaled fp16 pieces (round 4) --------------------
// CReLU chains over `nsl` list elements of exactly 16 channels: x = first channel of the first slice, wq = the
// layer's weights prepared by dense16_h2_prepare (dense16_h2_filter_bytes(nsl) bytes, 16-byte aligned), rec = `nrec`
// consecutive amax records whose maximum bounds |x|.  y[pix, coff + n] += sum (always accumulating, no bias).
size_t dense16_h2_filter_bytes(int nsl);
bool dense16_h2_shape_ok(int N, int H, int W);
int dense16_h2_prepare(const float* const* wT, const int* nsl, void* const* out, int count, hipStream_t s);
// the chain of a group (layers 1 .. nslices - 1 over the slices in front of each) in ONE launch where a workgroup covers an
// image (8 x 8, 16 x 16); false: shape not taken, the caller launches layer by layer
bool dense16_chain_fwd_h2(int N, int H, int W, int nslices, float* buf, int ld, const void* const* filters, float* records,
                          hipStream_t s);
bool dense16_chain_bwd_h2(int N, int H, int W, int nslices, float* g, int ldg, const float* x, int ldx,
                          const void* const* filters, const float* rec0, float* slice_records, hipStream_t s);
int dense16_fwd_h2(int N, int H, int W, int nsl, const float* x, int ldx, const void* wq, const float* rec, int nrec,
                   float* y, int ldy, int coff, hipStream_t s, float* amax_out);

// input gradient of the chains by slice: dG_c += [x_c > 0] G+ - [x_c < 0] G-, gathered from the nsl later layers of the
// group (g = gradient buffer at slice c + 1, dx = at slice c, x = forward buffer at slice c; wq = weights of the pairs
// (c, c + 1 ...) prepared by dense16_h2_bwd_prepare; rec0 / rec1: two ranges of amax records bounding the sources)
struct Dense16BwdPair {
  const float* w;        // HWIO chain weights of the source layer [9][32 * nch][16]
  const void* fwd;       // the source layer's forward prepared buffer
  void* out_base;        // prepared buffer of the output slice (dense16_h2_bwd_filter_bytes(pairs of that slice))
  int nch, cidx, pair_index;
};
size_t dense16_h2_bwd_filter_bytes(int nsl);
int dense16_h2_bwd_prepare(const Dense16BwdPair* pairs, int npairs, const void* const* allfwd, int nall, hipStream_t s);
int dense16_bwd_h2(int N, int H, int W, int nsl, const float* g, int ldg, const void* wq, const float* x, int ldx, float* dx,
                   const float* rec0, int nrec0, const float* rec1, int nrec1, hipStream_t s, float* amax_out);

// ---- weight gradient ----------------------------------------------------------------------
// Tiling shared by the LDS kernels: a block tile is TR full rows (64*PT pixels) of one image.
struct Dense16Tiling {
  int ok;        // 0: geometry not supported by the LDS kernels
  int PT, TR, RS;
  int tiles;     // N * H / TR
  int nchunk;    // ceil(Ceff / 32)
  int nsplit;    // pixel splits of the wgrad grid (slabs)
  int tiles_per_split;
};
// pure function of the geometry (used for the workspace query as well)
Dense16Tiling dense16_tiling(int N, int H, int W, int Ceff);
// slab[split][tap][e][n] = sum over the split's pixels of act(+-x[pix + tap, c(e)]) * dy[pix, n]
// (nsplit slabs of 9*Ceff*16 floats; the caller reduces them)
// x_rec / dy_rec (round 4; both or neither): amax records bounding the x slices read and dy -- the kernel then runs on the
// fp16 matrix pipe with two scaled fp16 pieces per operand (OTGAN_DENSE16_WGRAD_H2=0: always the fp32 pipe)
int dense16_wgrad(const Dense16Geo& g, const float* x, const float* dy, int ldy, int coff, float* slabs,
                  hipStream_t s, const float* x_rec = nullptr, int x_nrec = 0, const float* dy_rec = nullptr, int dy_nrec = 0);

// ---- input gradient -----------------------------------------------------------------------
// dx[q, c] (+)= act'(x[q,c]) * G+[q,c] - act'(-x[q,c]) * G-[q,c],
// G+-[q, c] = sum_{tap, n} dy[q - tap, n] * w[tap][e+-(c)][n]   (w: HWIO, e+- from `inv` or c, C + c)
int dense16_dgrad(const Dense16Geo& g, const float* dy, int ldy, int coff, const float* w, const float* x,
                  const int32_t* inv, float* dx, int lddx, int accumulate, hipStream_t s);
