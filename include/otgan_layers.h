/*
 * otgan_layers.h -- layer kernels of the OT-GAN generator / critic (included from otgan.h).
 *
 * Activations are NHWC fp32, weights HWIO fp32 ([KH][KW][Cin_eff][Cout]), exactly the
 * layouts of the reference's TensorFlow graph, so parameter tensors are interchangeable.
 * Spatial sizes must be powers of two (32x32 / 64x64 images and their pyramids).
 */
#ifndef OTGAN_LAYERS_H
#define OTGAN_LAYERS_H

/* pre-activations of reference utils/nn.py:190-206 */
#define OTGAN_ACT_NONE 0
#define OTGAN_ACT_CRELU 1 /* relu(concat([x0,-x0,x1,-x1,...]))  -- doubles the channels */
#define OTGAN_ACT_CELU 2  /* elu (concat([x0,-x0,x1,-x1,...]))  -- doubles the channels */
#define OTGAN_ACT_ELU 3
#define OTGAN_ACT_RELU 4

/*
 * Geometry of one conv2d layer (reference utils/nn.py:327-338 + :234-241).
 * The (list of) input tensor(s) is ONE NHWC buffer with channel stride ldx whose first C
 * channels are the concatenated list elements; the output is written at channel offset
 * y_coff of a buffer with channel stride ldy (DenseNet blocks grow in place instead of
 * re-concatenating, reference models/densenet.py:11-16).
 * Derived: Cin_eff = C * (2 for CRELU/CELU, else 1); Hin = H << upsample;
 * OH = ceil(Hin / stride); TF 'SAME': pad_before = max((OH-1)*stride + KH - Hin, 0) / 2.
 */
typedef struct otgan_conv_desc {
  int N;        /* batch */
  int H, W;     /* stored spatial size of the input buffer */
  int C;        /* real input channels */
  int ldx;      /* channel stride of the input buffer (>= C) */
  int upsample; /* 1: 2x nearest-neighbour resize before the conv (nn.py:235-236) */
  int KH, KW;
  int stride;   /* 1 or 2 */
  int Cout;
  int ldy;      /* channel stride of the output buffer */
  int y_coff;   /* channel offset of this layer's output inside the output buffer */
  int preact;   /* OTGAN_ACT_* */
  int list_quads; /* CRELU/CELU only: 1 = every list element (or C itself for a single tensor)
                     is a multiple of 4 channels wide, i.e. aligned groups of 4 effective
                     channels map to 4 consecutive source channels with one sign -> 16-byte
                     gathers; 0 = arbitrary widths -> per-channel gathers */
  /* Optional, Winograd passes only (otgan_conv2d_filter_bytes > 0): device pointers to "amax records" of the
   * tensors the pass reads -- otgan_absmax_f32 of x (forward, wgrad) and of dy (dgrad, wgrad).  The scaled
   * two-piece operands of those passes need the largest magnitude of their source tensor; NULL = the library
   * reduces the tensor itself inside the call (one more read of it).  A caller that runs forward and wgrad on the
   * same x, or dgrad and wgrad on the same dy, computes each record once. */
  const float* x_amax;
  const float* dy_amax;
  /* forward only: 1 = y += conv(x) + bias instead of y = ...  Implemented for the 3x3 / stride-1 / 16-output growth
   * layers (dense16 kernels) and for the wide 3x3 / stride-1 layers that take the Winograd path, which is what a dense
   * block split into "wide convolutions of finished channel groups + short growth chains" needs (ops.py
   * DenseBlockFunction); any other layer with this flag set is rejected with OTGAN_ERR_ARG. */
  int y_accumulate;
  /* Optional: device buffer of otgan_conv2d_operand_bytes(d) bytes (16-byte aligned) shared by the forward call and the
   * weight-gradient call on the SAME x.  Both passes of a Winograd layer start with the same transform of x into the
   * GEMM's operand format; with this buffer the forward call leaves that operand here instead of in its workspace
   * and the weight-gradient call reads it instead of transforming x again (the buffer must stay untouched in
   * between; x_amax is then not needed by the weight gradient).  Ignored when otgan_conv2d_operand_bytes(d) == 0;
   * when it is > 0 and a call cannot take the Winograd path (list input, misaligned operands, short workspace) the
   * call fails with OTGAN_ERR_ARG instead of silently writing / reading nothing.  NULL = each pass transforms x. */
  void* x_operand;
  /* Optional OUTPUT amax records (round 3): the kernel that WRITES y (forward) / dx (input gradient) also leaves the
   * largest |value| it wrote in the record -- atomically max-accumulated as the float's bit pattern, so the caller must
   * have zeroed the record (all OTGAN_AMAX_RECORD_FLOATS floats) before the call; NaN / infinity propagate as in otgan_absmax_f32.  The next Winograd layer
   * takes the record as its x_amax / dy_amax and the separate reduction pass over the tensor (otgan_absmax_f32: one
   * more read of it) disappears.  otgan_conv2d_amax_fused(d, which) tells whether the pass does this inside its own
   * output kernel; otherwise a given record is filled by a separate reduction launch (same result).  Requires
   * Cout % 4 == 0 (forward) / C % 4 == 0 (input gradient), 4-aligned channel strides.  y_amax_out with y_accumulate:
   * the record receives the SUMS the call leaves in memory, so that several calls that finish different parts of a
   * buffer can share one record (the growth layers and wide convolutions of a dense block); dx_amax_out: not with
   * `accumulate` -- round 4: allowed with `accumulate` too, the record then max-accumulates the SUMS left in dx. */
  float* y_amax_out;
  float* dx_amax_out;
  /* Optional, otgan_conv2d_prepare_filters_f32 only: amax record of the NORMALISED weights the filters are made from
   * (un-folded w / wT: which = 2, 3 of the folded layers, which = 0, 1 of the strided and wide 3x3 layers), e.g. from
   * otgan_weightnorm_fwd_amax_f32.  NULL = the call reduces the weights itself (one more read of them).  Ignored for
   * filters made from FOLDED weights (a different tensor). */
  const float* w_amax;
  /* (round 4) x_amax may point to x_amax_count consecutive records (OTGAN_AMAX_RECORD_FLOATS floats apart); the bound used
   * is their maximum.  0 or 1 = one record.  Read by the growth-layer kernels (Cout = 16) and by the Winograd passes of the
   * strided 5x5 and the wide 3x3 stride-1 layers (as is dy_amax_count below); every other pass reads the first record only. */
  int x_amax_count;
  /* (round 4) CRELU / CELU list inputs: nonzero = every list element is exactly this many channels wide and lies at channel
   * i * list_width of x, i.e. the channel map is the per-element interleave [x_0, -x_0, x_1, -x_1, ...] of equal slices.
   * With list_width = 16, CRELU, y_accumulate, no bias, an x_amax record and `filters` prepared by
   * otgan_dense16_prepare_filters_f32, a growth layer (3x3, stride 1, Cout = 16) runs on the two-scaled-fp16-piece kernel
   * (otgan_dense16_h2_ok tells).  0 = unknown (the channel map decides). */
  int list_width;
  /* (round 4) forward only: the layer is followed by a gated linear unit over the channel halves (reference
   * models/dcgan.py:35-36: a, b = split(y, 2, axis = 3); a * sigmoid(b)).  Non-NULL glu_out: the kernel that writes y
   * also writes the gated product to glu_out[N, OH, OW, Cout / 2] (contiguous, 16-byte aligned) -- y itself is still
   * written, the backward pass of the unit needs it -- and, if glu_amax_out is non-NULL, max-accumulates the largest
   * |gated value| into that (zeroed) amax record.  Only where otgan_conv2d_glu_fused(d) returns 1 (the Winograd path of
   * the 5x5 upsampling layers, Cout % 8 == 0, y_coff == 0, ldy == Cout); any other layer with glu_out set is rejected
   * with OTGAN_ERR_ARG.  Same arithmetic as otgan_glu_fwd_amax_f32 on y: bit-identical. */
  float* glu_out;
  float* glu_amax_out;
  /* (round 4) dy_amax points to this many consecutive records (see x_amax_count); 0 or 1 = one record. */
  int dy_amax_count;
} otgan_conv_desc;

/*
 * Channel maps (device int32 arrays, may be NULL for a single-tensor input):
 *   cmap[d], d in [0, Cin_eff): source channel (low 31 bits) and sign (bit 31 set = negate)
 *            of effective channel d -- encodes the reference's per-list-element interleave
 *            [x0,-x0,x1,-x1,...] (nn.py:198,200).  NULL = [x, -x] / identity.
 *   inv[c], inv[C + c]: effective index of (+x_c) and (-x_c); used by dgrad.  NULL = c, C+c.
 */

/* which: 0 fwd, 1 dgrad, 2 wgrad */
size_t otgan_conv2d_workspace_bytes(const otgan_conv_desc* d, int which);
/* 1: the forward pass of this layer can write the gated product of otgan_conv_desc::glu_out itself */
int otgan_conv2d_glu_fused(const otgan_conv_desc* d);
/* size of otgan_conv_desc::x_operand for this layer; 0 = forward and weight gradient do not share an operand */
size_t otgan_conv2d_operand_bytes(const otgan_conv_desc* d);

/*
 * amax record of x[rows][C] (row stride ld floats, C and ld multiples of 4, 16-byte aligned): the largest |x| (NaN if
 * any element is NaN).  record: OTGAN_AMAX_RECORD_FLOATS floats of device memory, written asynchronously on `stream`;
 * pass it as otgan_conv_desc::x_amax / dy_amax.  Deterministic.  Format: 16 sub-slots at a stride of 32 floats, each
 * the bit pattern of a non-negative float; the record's value is their maximum (this call writes it to sub-slot 0 and
 * zeroes the others; the kernels behind y_amax_out / dx_amax_out max-accumulate into all 16, one cache line each, so
 * that their atomics do not queue up in one L2 channel).
 */
#define OTGAN_AMAX_RECORD_FLOATS 512
int otgan_absmax_f32(const float* x, long rows, int C, long ld, float* record, void* stream);
/* which: 0 forward (y_amax_out), 1 input gradient (dx_amax_out): 1 = the pass fills the record in its own output
 * kernel (no extra launch, no extra read), 0 = it would take a separate reduction. */
int otgan_conv2d_amax_fused(const otgan_conv_desc* d, int which);

/*
 * Winograd F(4x4,3x3) paths (fwd, dgrad and wgrad; scratch for the transformed operands is
 * reported by otgan_conv2d_workspace_bytes; nothing else changes for the caller):
 *   - folded 5x5 upsampling layers without pre-activation (the DCGAN generator,
 *     models/dcgan.py:33-46): each output-parity class is a 3x3 convolution on the small image;
 *   - 5x5 stride-2 layers with a single-tensor input (the DCGAN critic, models/dcgan.py:12-14):
 *     four 3x3 sub-convolutions of the input-parity sub-images, classes folded into the
 *     contraction index, structurally-zero blocks skipped (49 instead of 100 products per tile).
 *   - 3x3 stride-1 layers with >= 128 outputs and a single-tensor input (the wide convolutions a dense block is cut
 *     into, ops.py DenseBlockFunction) and 3x3 layers on a 2x upsampled input with CReLU (the DenseNet generator's
 *     transitions, models/densenet.py:67-73): the same passes with one class on the (upsampled) grid.
 * The batched GEMMs of these paths run on the fp16 matrix pipe with operands stored as two fp16 pieces of the
 * power-of-two-scaled value (hi + lo = 22 significand bits, one scale per Winograd frequency derived from the
 * largest magnitude of the tensor the operand is a transform of): three MFMAs per product, fp32 accumulation,
 * exact rescale on the way out -- measured errors at or below those of the fp32 MFMA chain of the direct path.
 * OTGAN_WINO_PIECES=3 selects the other build of the same source: three bf16 pieces (24 significand bits), six
 * MFMAs per product, no scales (read per call; a prepared filter buffer belongs to the mode it was made in).
 * Environment switches (debugging / A-B measurements): OTGAN_DISABLE_WINOGRAD=1,
 * OTGAN_WINO_FP32=1 (Winograd GEMMs on the fp32 engine), OTGAN_WINO_WGRAD_X3=0 (weight-gradient GEMMs on
 * the fp32 engine), OTGAN_WINO_WGRAD_TL=0 (weight-gradient operands from the transposing producers instead
 * of the forward-layout ones), OTGAN_DISABLE_DENSE16=1, OTGAN_DENSE16_V1=1; tuning knobs kept for measurements:
 * OTGAN_X3_SPLIT_TARGET (workgroups a K-split weight-gradient GEMM aims for, default 256), OTGAN_OUTER_CHUNKS
 * (pixel chunks of the few-channel weight gradient, default 512), OTGAN_X3_STREAM (0 / 1 / 2: the persistent
 * stream-K GEMM never / where it pays / always; read per launch), OTGAN_X3_STREAM_HALF_ROUNDS (its selection
 * threshold in half tiles per compute unit, default 3), OTGAN_X3_FMAP=0 (tile-residue instead of frequency-major
 * GEMM grid), OTGAN_AMAX_BLOCKS (workgroups of the largest-magnitude reduction, default 256);
 * OTGAN_DISABLE_WINO_PLAIN3=1 (wide 3x3 stride-1 layers back on the implicit GEMM), OTGAN_PLAIN3_MIN_CEFF / _MIN_COUT
 * (their eligibility thresholds, 64 / 128), OTGAN_DISABLE_WINO_UP3=1, OTGAN_DISABLE_WINO_UP3_WGRAD=1,
 * OTGAN_DISABLE_WINO_UP3_DGRAD=1 (3x3 upsampling layers: all passes / weight gradient / input gradient back on the
 * folded implicit GEMM), OTGAN_DISABLE_X_OPERAND=1 (otgan_conv2d_operand_bytes reports 0: no operand sharing),
 * OTGAN_IGEMM_X3=0 (implicit-GEMM forward / input gradient back on the fp32 MFMA loop instead of three bf16 pieces).
 */

/*
 * Upsample folding.  conv(k x k) applied to a 2x nearest-neighbour upsampled image equals four
 * output-parity-class convolutions on the SMALL image whose taps are sums of the original
 * ones (k = 5: 3x3 per class instead of 5x5; k = 3: 2x2 instead of 3x3) -- identical
 * mathematics with 25/9 (9/4) fewer multiply-adds in fwd, dgrad and wgrad.
 * A layer is folded iff otgan_conv2d_folded_weight_elems(d) > 0 (a pure function of the
 * descriptor: upsample == 1, stride == 1, Cin_eff % 16 == 0, 4-aligned channel strides).
 * For a folded layer the caller runs otgan_conv2d_fold_weights_f32 once per weight update and
 * passes `weffT` as the `wT` argument of otgan_conv2d_fwd_f32 and `weff` as the `w` argument
 * of otgan_conv2d_dgrad_f32; otgan_conv2d_wgrad_f32 still returns the un-folded HWIO gradient.
 */
size_t otgan_conv2d_folded_weight_elems(const otgan_conv_desc* d);
int otgan_conv2d_fold_weights_f32(const otgan_conv_desc* d, const float* w, float* weff,
                                  float* weffT, void* stream);

/* y = conv2d(preact(upsample(x)), W) + bias.   wT: [Cout][KH*KW*Cin_eff] (transposed copy
 * of the HWIO weight, produced by otgan_weightnorm_fwd_f32).  nn.py:241,337. */
int otgan_conv2d_fwd_f32(const otgan_conv_desc* d, const float* x, const int32_t* cmap,
                         const float* wT, const float* bias, float* y, void* workspace,
                         size_t workspace_bytes, void* stream);

/* dx (+)= d(loss)/d(x) given dy; w is the HWIO weight.  x (the layer input) is needed for
 * the activation derivative.  dx: [N,H,W,lddx] (first C channels written).
 * accumulate != 0 adds into dx (DenseNet gradient buffers). */
int otgan_conv2d_dgrad_f32(const otgan_conv_desc* d, const float* dy, const float* w,
                           const float* x, const int32_t* inv, float* dx, int lddx,
                           int accumulate, void* workspace, size_t workspace_bytes, void* stream);

/*
 * Winograd-domain filters made once per weight update.  The layers that run as Winograd GEMMs derive
 * their filter operand from the weights inside every forward / dgrad call; a caller that knows the
 * weights are unchanged (the critic during the generator steps: five steps out of six) can derive it
 * once into its own buffer and pass it to the _pf variants.
 *   otgan_conv2d_filter_bytes(d, which)   which = 0 forward, 1 dgrad; 0: the layer / pass has none
 *   otgan_conv2d_prepare_filters_f32      `w` = what the pass itself takes (wT for forward, w for dgrad;
 *                                         folded layers: weffT / weff).  Folded layers also accept
 *                                         which = 2 / 3: forward / dgrad filters from the UN-folded wT / w
 *                                         (bit-identical; the fold is then not needed by these two passes:
 *                                         with `filters` given they do not read their weight argument).
 *                                         3x3 layers on an upsampled CReLU input have ONLY these two (their
 *                                         forward / dgrad run as Winograd GEMMs exactly when `filters` is given;
 *                                         a dgrad call with `filters` must then satisfy that path's alignment /
 *                                         workspace requirements -- it is rejected otherwise, never rerouted)
 *   otgan_conv2d_fwd_pf_f32 / otgan_conv2d_dgrad_pf_f32: as the plain calls; `filters` may be NULL
 *                                         (then identical to them) and is ignored by the non-Winograd paths.
 * The buffer's content is tied to the descriptor and to the OTGAN_WINO_* switches of the process.
 */
size_t otgan_conv2d_filter_bytes(const otgan_conv_desc* d, int which);
/* Growth layers of a split dense block (ops.py DenseBlockFunction): the chain weights wT [16][9 * 32 * nslices] (CReLU over
 * `nslices` list elements of 16 channels) pre-split into two scaled fp16 pieces in MFMA fragment order, up to
 * OTGAN_DENSE16_MAX_BATCH layers per launch; pass the buffer as `filters` of otgan_conv2d_fwd_pf_f32.
 * otgan_dense16_h2_ok(d): 1 if a forward call with this descriptor (plus filters and x_amax) takes that kernel. */
#define OTGAN_DENSE16_MAX_BATCH 16
size_t otgan_dense16_filter_bytes(int nslices);
int otgan_dense16_prepare_filters_f32(const float* const* wT, const int* nslices, void* const* filters, int count,
                                      void* stream);
int otgan_dense16_h2_ok(const otgan_conv_desc* d);
/*
 * Input gradient of those chains BY SLICE (round 4): the gradient of slice c of a group gathers from the `npairs` later
 * layers of the group in one launch,  dG_c += [x_c > 0] G+ - [x_c < 0] G-  (CReLU; reference utils/nn.py:198-200 backward),
 * instead of every layer adding into every earlier slice.  g / dx: the gradient buffer [N, H, W, ldg] at the first channel of
 * slice c + 1 / of slice c; x: the forward buffer at slice c (row stride ldx).  Slices must be processed last to first
 * (a source slice must be final).  filters: otgan_dense16_bwd_filter_bytes(npairs) bytes prepared by
 * otgan_dense16_prepare_bwd_filters_f32 -- one otgan_dense16_bwd_pair per (output slice, source layer): `w` the source
 * layer's chain weights HWIO [9][32 * nslices_src][16], `fwd_filters` that layer's forward buffer (scale exponent),
 * `slice_index` the position of slice c in that chain, `pair_index` the pair's position in `filters`;
 * all_fwd_filters: the forward buffers of every chain layer of the block (one scale for the block).
 * rec0 / rec1: two ranges of consecutive amax records whose maximum bounds the source slices (rec1 nullable);
 * amax_out (nullable, zeroed or shared): max-accumulates the sums written.  Geometry as otgan_dense16_h2_ok.
 */
/*
 * Whole chains in one call (the launches of a group's layers issued back to back from C: at 8 x 8 a chain kernel takes 9 us,
 * less than one Python-level call).  A group = `nslices` consecutive 16-channel slices of the block buffer starting at
 * `buf_group` (row stride ld floats); records = (nslices + 1) consecutive amax records [W, c_0, c_1, ...] as laid out by
 * ops.py DenseBlockFunction (W: the wide convolutions' sums; c_j: left by the kernel that finishes slice j).
 *   otgan_dense16_chain_fwd_f32: layers j = 1 .. nslices - 1 in order, layer j adds its chain over slices [0, j) onto slice
 *     j (filters[j - 1]: its prepared weights), reads records [0, 1 + j), writes record 1 + j.
 *   otgan_dense16_chain_bwd_f32: slices c = nslices - 2 .. 0, otgan_dense16_bwd_slice_f32 each (filters[c]; sources bounded
 *     by rec0 (one record) and slice_records[c + 1 ..]; slice_records[c] receives the sums left).
 */
int otgan_dense16_chain_fwd_f32(int N, int H, int W, int nslices, float* buf_group, int ld, const void* const* filters,
                                float* records, void* stream);
int otgan_dense16_chain_bwd_f32(int N, int H, int W, int nslices, float* g_group, int ldg, const float* x_group, int ldx,
                                const void* const* filters, const float* rec0, float* slice_records, void* stream);
typedef struct otgan_dense16_bwd_pair {
  const float* w;
  const void* fwd_filters;
  void* filters;
  int nslices_src, slice_index, pair_index;
} otgan_dense16_bwd_pair;
size_t otgan_dense16_bwd_filter_bytes(int npairs);
int otgan_dense16_prepare_bwd_filters_f32(const otgan_dense16_bwd_pair* pairs, int npairs, const void* const* all_fwd_filters,
                                          int nall, void* stream);
int otgan_dense16_bwd_slice_f32(int N, int H, int W, int npairs, const float* g, int ldg, const void* filters, const float* x,
                                int ldx, float* dx, const float* rec0, int nrec0, const float* rec1, int nrec1,
                                float* amax_out, void* stream);
int otgan_conv2d_prepare_filters_f32(const otgan_conv_desc* d, int which, const float* w, void* filters,
                                     size_t filter_bytes, void* stream);
int otgan_conv2d_fwd_pf_f32(const otgan_conv_desc* d, const float* x, const int32_t* cmap,
                            const float* wT, const void* filters, const float* bias, float* y,
                            void* workspace, size_t workspace_bytes, void* stream);
int otgan_conv2d_dgrad_pf_f32(const otgan_conv_desc* d, const float* dy, const float* w, const void* filters,
                              const float* x, const int32_t* inv, float* dx, int lddx,
                              int accumulate, void* workspace, size_t workspace_bytes, void* stream);

/* dw[KH][KW][Cin_eff][Cout] = sum over pixels of preact(x)^T . dy  (overwrites dw). */
int otgan_conv2d_wgrad_f32(const otgan_conv_desc* d, const float* x, const int32_t* cmap,
                           const float* dy, float* dw, void* workspace, size_t workspace_bytes,
                           void* stream);

/* ---- weight normalisation (nn.py:176-181): w = g * V * rsqrt(max(sum_col V^2, 1e-12)) ------------
 * V, w: [K][Cout] (K = KH*KW*Cin_eff), wT: [Cout][K] (nullable), inv_norm: [Cout].        */
int otgan_weightnorm_fwd_f32(const float* V, const float* g, int K, int Cout, float* w,
                             float* wT, float* inv_norm, void* stream);
/* the same, also max-accumulating |w| into an amax record (zeroed by the caller; NULL = none): pass it as
 * otgan_conv_desc::w_amax to otgan_conv2d_prepare_filters_f32 */
int otgan_weightnorm_fwd_amax_f32(const float* V, const float* g, int K, int Cout, float* w, float* wT,
                                  float* inv_norm, float* w_amax, void* stream);
/* dV, dg from dw (scratch: Cout floats). */
int otgan_weightnorm_bwd_f32(const float* V, const float* g, const float* inv_norm,
                             const float* dw, int K, int Cout, float* dV, float* dg,
                             float* scratch, void* stream);

/*
 * The same two passes for MANY layers with 16 outputs each (the growth layers of a dense block,
 * models/densenet.py:11-16) in one launch per OTGAN_WN_MAX_LAYERS layers: `layers` is a HOST array, the pointers in it
 * are device pointers (16-byte aligned).  Deterministic.
 *   forward: w[K][16], wT[16][K] (nullable), inv[16] from V[K][16], g[16].
 *   backward: dV[taps*Ceff][16], dg[16] from the weight gradient given as up to three PARTS per layer, each holding
 *     nrows consecutive effective-channel rows of every tap: row (tap, e) of part i is the 16 floats at
 *     p + ((tap * nrows + perm[e]) * rstride)   (perm: device int32 array or NULL = identity; unused parts: nrows 0)
 *     -- the layout in which a dense block computed as wide convolutions + growth chains (ops.py DenseBlockFunction)
 *     leaves its weight gradients: a column slice of each wide convolution's dw, then the layer's own chain.
 */
#define OTGAN_WN_MAX_LAYERS 16
typedef struct otgan_wn_fwd_layer {
  const float* V;
  const float* g;
  float* w;
  float* wT;
  float* inv;
  int K;
} otgan_wn_fwd_layer;
typedef struct otgan_wn_part {
  const float* p;
  const int32_t* perm;
  int nrows;
  int rstride;
} otgan_wn_part;
typedef struct otgan_wn_bwd_layer {
  const float* V;
  const float* g;
  const float* inv;
  float* dV;
  float* dg;
  int Ceff, taps;
  otgan_wn_part part[3];
} otgan_wn_bwd_layer;
int otgan_weightnorm_fwd_batched16_f32(const otgan_wn_fwd_layer* layers, int n_layers, void* stream);
int otgan_weightnorm_bwd_batched16_f32(const otgan_wn_bwd_layer* layers, int n_layers, void* stream);

/* out[c] = sum_r a[r*lda + c]   (bias gradients; rows = pixels).  scratch: 256*cols floats */
int otgan_colsum_f32(const float* a, long rows, int cols, long lda, float* out, float* scratch,
                     void* stream);

/* ---- pointwise blocks ------------------------------------------------------------------ */
/* GLU (models/dcgan.py:35-36): y[r, c] = x[r, c] * sigmoid(x[r, C + c]), x: [rows, 2C].    */
int otgan_glu_fwd_f32(const float* x, long rows, int C, float* y, void* stream);
int otgan_glu_bwd_f32(const float* x, const float* dy, long rows, int C, float* dx, void* stream);
/* the same, also max-accumulating |y| / |dx| into an amax record (otgan_conv_desc::y_amax_out; NULL = none;
 * needs C % 4 == 0 and 16-byte aligned tensors) */
int otgan_glu_fwd_amax_f32(const float* x, long rows, int C, float* y, float* y_amax, void* stream);
int otgan_glu_bwd_amax_f32(const float* x, const float* dy, long rows, int C, float* dx, float* dx_amax, void* stream);
/* The same, also writing colsum[2 C] = the column sums of dx (the bias gradient of the convolution in front of the GLU,
 * models/dcgan.py:39-47: conv -> GLU) without another pass over dx.  scratch: 256 * 2 C floats.  C % 4 == 0, 16-byte
 * aligned tensors; dx_amax nullable. */
int otgan_glu_bwd_colsum_f32(const float* x, const float* dy, long rows, int C, float* dx, float* dx_amax, float* colsum,
                             float* scratch, void* stream);
/* tanh output (models/dcgan.py:50) */
int otgan_tanh_fwd_f32(const float* x, long n, float* y, void* stream);
int otgan_tanh_bwd_f32(const float* y, const float* dy, long n, float* dx, void* stream);
/* Feature head (models/dcgan.py:16-19): f = concat([relu(x), relu(-x)], channel), flatten
 * (h, w, c), L2-normalise rows (no epsilon).  x: [N, HW, C] -> f: [N, HW*2C].
 * norm: [N] saved row norms for the backward pass. */
int otgan_feature_head_fwd_f32(const float* x, int N, int HW, int C, float* f, float* norm,
                               void* stream);
int otgan_feature_head_bwd_f32(const float* x, const float* f, const float* norm,
                               const float* df, int N, int HW, int C, float* dx, void* stream);
/* the same, also max-accumulating |dx| into an amax record (NULL = none) */
int otgan_feature_head_bwd_amax_f32(const float* x, const float* f, const float* norm, const float* df, int N, int HW,
                                    int C, float* dx, float* dx_amax, void* stream);

/* ---- optimiser / EMA (utils/nn.py:50-73, train.py:63-64) --------------------------------
 * Adam with the reference's epsilon placement:  p -= lr * vhat / sqrt(mghat + 1e-8),
 * vhat = v/(1-mom1^t), mghat = mg/(1-mom2^t); mom1 == 0 skips the first moment.          */
/* Hyper-parameters travel as doubles: the reference forms (1 - mom) and (1 - mom^t) in
 * Python doubles before they become fp32 graph constants.                                  */
int otgan_adam_step_f32(float* p, const float* grad, float* v, float* mg, long n, double lr,
                        double mom1, double mom2, double t, void* stream);
/* The same step over a FLAT parameter buffer p whose gradient comes as one device tensor per variable (what the
 * backward pass returns): variable i owns p[offsets[i] .. offsets[i+1]) and reads grads[i][0 ..]; `grads` and `offsets`
 * (nseg + 1 entries, ascending) are HOST arrays, nseg <= OTGAN_ADAM_MAX_SEGMENTS.  No concatenation of the gradients
 * (151 MB per DCGAN generator step).  ema_shadow (nullable, same layout as p): shadow = ema_decay * shadow +
 * (1 - ema_decay) * p_new in the same pass (train.py:63-64,223). */
#define OTGAN_ADAM_MAX_SEGMENTS 32
int otgan_adam_step_gather_f32(float* p, const float* const* grads, const long* offsets, int nseg, float* v, float* mg,
                               double lr, double mom1, double mom2, double t, float* ema_shadow, double ema_decay,
                               void* stream);
/* The bias corrections {1 - mom1^t, 1 - mom2^t} of that step as the two entries above evaluate them (fp32, nn.py:62,67), on
 * the HOST (no launch), and the gathered step reading them from DEVICE memory (`coef_dev` [2]; null = from `t` as above):
 * a step captured in a hipGraph (trainer.py: one graph per step kind, replayed) keeps its launch arguments, so what changes
 * from step to step must come from memory -- the trainer writes the next step's pair into `coef_dev` before each replay and
 * the replayed step is bit-identical to the eager one (reference train.py:142-143: one `t` per optimiser). */
void otgan_adam_coefficients(double mom1, double mom2, double t, float* out2);
int otgan_adam_step_coef_f32(float* p, const float* grad, float* v, float* mg, long n, double lr,
                             double mom1, double mom2, double t, const float* coef_dev, void* stream);
int otgan_adam_step_gather_coef_f32(float* p, const float* const* grads, const long* offsets, int nseg, float* v, float* mg,
                                    double lr, double mom1, double mom2, double t, const float* coef_dev, float* ema_shadow,
                                    double ema_decay, void* stream);
/* Up to OTGAN_COPY2D_MAX_SEGMENTS strided 2-D copies in ONE launch: dst[s][r * dst_ld[s] + c] = src[s][r * src_ld[s] + c]
 * for r < rows[s], c < cols[s].  All arrays are HOST arrays of nseg entries.  (The growth layers of a DenseNet block read
 * sub-blocks of their normalised weights -- rows of every filter tap, models/densenet.py:11-16 through
 * ops.py DenseBlockFunction -- as contiguous tensors: 28 slices per block and weight version, one launch instead of
 * 28 framework copies.) */
#define OTGAN_COPY2D_MAX_SEGMENTS 64
int otgan_copy2d_batched_f32(const float* const* src, float* const* dst, const int* rows, const int* cols,
                             const long* src_ld, const long* dst_ld, int nseg, void* stream);
/* Up to OTGAN_GATHER3D_MAX_SEGMENTS strided 3-D copies with a gathered middle index in ONE launch:
 *   dst[s][i0 * dst_s0[s] + i1 * dst_s1[s] + i2] = src[s][i0 * src_s0[s] + (base1 + map1[i1]) * src_s1[s] + i2]
 * for i0 < n0[s], i1 < n1, i2 < n2 (map1: n1 device ints, or NULL = identity); the per-segment arrays are HOST arrays of nseg
 * entries, strides in elements.  (The wide convolutions of a split DenseNet block -- reference models/densenet.py:11-16 through
 * ops.py _split_block_plan -- read the rows of one channel group out of the weights of all later growth layers, side by side
 * and re-ordered from the reference's per-list-element CReLU order [x0, -x0, x1, -x1, ...] (utils/nn.py:198-200) to the
 * single-tensor order: one launch per operand instead of a cat, an index_select and their copies per layer.)  amax_out
 * (nullable): a zeroed amax record (OTGAN_AMAX_RECORD_FLOATS floats) that receives the largest magnitude written -- the
 * record `otgan_conv_desc::w_amax` wants for the gathered weights, without a reduction pass over them. */
#define OTGAN_GATHER3D_MAX_SEGMENTS 32
int otgan_gather3d_batched_f32(const float* const* src, float* const* dst, const int* n0, int n1, int n2,
                               const long* src_s0, const long* src_s1, const long* dst_s0, const long* dst_s1,
                               int base1, const int* map1_dev, float* amax_out, int nseg, void* stream);
int otgan_adamax_step_f32(float* p, const float* grad, float* v, float* mg, long n, double lr,
                          double mom1, double mom2, void* stream);
int otgan_nesterov_step_f32(float* p, const float* grad, float* v, long n, double lr, double mom1,
                            void* stream);
/* shadow = decay*shadow + (1-decay)*p */
int otgan_ema_update_f32(float* shadow, const float* p, long n, double decay, void* stream);

#endif /* OTGAN_LAYERS_H */
