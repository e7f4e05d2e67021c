/*
 * otgan.h -- C ABI of the MI355X-native OT-GAN hot path (libotgan_hip.so).
 *
 * The reference (openai/ot-gan) has no FFI: its "operator API" is Python call sites on stock
 * TensorFlow ops.  Each entry point below replaces one group of those call sites; the
 * reference file:line it stands in for is cited per function.  The Python host layer
 * (ot-gan_amd/) mirrors the reference's own names (utils/matching.py, utils/nn.py,
 * models/, train.py) on top of this ABI.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no C++/torch types.
 *   - every function returns 0 on success or a negative OTGAN_ERR_* code; the text of the
 *     last error on the calling thread is available from otgan_last_error().
 *   - all buffers are caller-owned DEVICE pointers (fp32 unless stated), row-major,
 *     with explicit leading dimensions in elements.
 *   - `stream` is a hipStream_t passed as void*; all work is asynchronous on it.
 *   - no internal allocation: scratch comes from a caller-provided workspace sized by the
 *     matching *_workspace_bytes() query.  Distinct workspaces/streams may be used from
 *     distinct threads concurrently.
 *   - scalar results (entropy, distance) are written to DEVICE memory so that the caller
 *     decides when to synchronise.
 */
#ifndef OTGAN_H
#define OTGAN_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OTGAN_ABI_VERSION 1

/* error codes */
#define OTGAN_OK 0
#define OTGAN_ERR_INVALID (-1)
#define OTGAN_ERR_WORKSPACE (-2)
#define OTGAN_ERR_LAUNCH (-3)
#define OTGAN_ERR_UNSUPPORTED (-4)

int otgan_version(void);
const char* otgan_last_error(void);

/* ---------------------------------------------------------------------------------------
 * Per-kernel-class HIP-event timing (measurement support for bench.py's roofline leg).
 * Classes: 0 conv_fwd, 1 conv_dgrad, 2 conv_wgrad, 3 cost_gemm, 4 sinkhorn, 5 plan_apply,
 * 6 pointwise, 7 wino_gemm (the batched Winograd-domain GEMM kernel alone, fp32 MFMA; its launches
 * are also part of the conv class they belong to), 8 wino_gemm_bf16x3 (the same GEMM on the bf16
 * pipe with split-precision operands; FLOP counted as executed bf16 FLOP = 6 per fp32 product).
 * While enabled every launch of a class is bracketed by hipEvents on its
 * launch stream; otgan_prof_collect() synchronises and returns the totals since the last
 * reset: out[0]=launches, out[1]=sum of milliseconds, out[2]=sum of algorithmic FLOP,
 * out[3]=sum of algorithmic bytes.
 * ------------------------------------------------------------------------------------- */
int otgan_prof_enable(int on);
int otgan_prof_reset(void);
int otgan_prof_collect(int cls, double* out4);

/* Sweep statistics of the Sinkhorn kernels (reference utils/matching.py:50-57: exactly L sweeps; here every sweep runs in
 * the log-domain form or, once the potentials have settled, in the linear form -- DESIGN.md section 3).  on = 1 allocates
 * (once) and zeroes eight device counters that every problem solved from then on adds to: [0] problems, [1] log-domain
 * sweeps, [2] linear sweeps, [3] entries into the linear form, [4] fold-backs to the log-domain form, [5] sum over problems
 * of the sweep at which the linear form was first entered, [6] problems that never entered it, [7] unused; on = 0
 * detaches them.  otgan_sinkhorn_counters_read copies them to the host (synchronises the device; reset != 0 zeroes them). */
int otgan_sinkhorn_counters(int on);
int otgan_sinkhorn_counters_read(long long* out8, int reset);

/* ---------------------------------------------------------------------------------------
 * Matching operator (mini-batch Sinkhorn energy distance).
 * Replaces reference utils/matching.py:11-85 (two-batch), :88-136 (single-batch) and the
 * toy variant toy_example/matching_cpu.py:4-95.
 * ------------------------------------------------------------------------------------- */
#define OTGAN_COST_COSINE 0        /* C = 1 - x.y            (utils/matching.py:31)              */
#define OTGAN_COST_SQEUCLID_MEAN 1 /* C = |x-y|^2 / (2 D)    (toy_example/matching_cpu.py:17-21) */

#define OTGAN_MATCH_TWO_BATCH 0
#define OTGAN_MATCH_SINGLE_BATCH 1

/* Scratch bytes needed by otgan_matching_*_f32 for a problem of `rows` rows per side
 * (two-batch: rows = N = half the samples of each kind; single-batch: rows = all samples). */
size_t otgan_matching_workspace_bytes(int mode, int rows, int D);

/*
 * Two-batch matching.  fa, fb: [2N, D] (rows [0,N) = mini-batch 1, [N,2N) = mini-batch 2;
 * `a` = generated, `b` = data), leading dimension ldf.  Outputs f_aa, f_bb, f_ab, f_ba:
 * [2N, D] with leading dimension ldo.  entropy: 1 float (mean of the six mean row
 * entropies, matching.py:57,61).  dist: 1 double = calc_distance (matching.py:139-153; for
 * OTGAN_COST_SQEUCLID_MEAN the toy normalisation matching_cpu.py:155-164), evaluated with
 * fp64 accumulation.  stats (nullable): [6][4] doubles per problem in the reference order
 * (a1a2, b2b1, a1b1, a1b2, a2b1, a2b2): {sum of row entropies, <M,C>, sum(M), 0}.
 * Exactly `iters` row->column sweeps are run, then a row softmax (no early exit).
 * Cosine cost: the features are what the reference's critics return (models/dcgan.py:16-19, models/densenet.py:40-42):
 * rows of unit length.  The cost and plan-application GEMMs multiply operands split into two scaled fp16 pieces with an
 * a-priori scale for such rows (the one-tile kernels of N <= 128; rows need not be of unit length, only |x| < 8); an element of
 * magnitude >= 8 overflows a piece there and the outputs come out NaN (loud, never silently wrong).  The pre-split engine of
 * N >= 256 measures its operands and re-splits when the largest magnitude leaves the expected band: it has no such limit
 * (tests/test_matching_h2_gpu.py::test_features_outside_the_cosine_contract_are_loud).  OTGAN_MATCH_FP32=1 (environment, read once) keeps these GEMMs on the exact-fp32 MFMA engine, which
 * has no such limit; the toy cost always runs there.
 */
int otgan_matching_two_batch_f32(const float* fa, const float* fb, int N, int D, long ldf,
                                 float sinkhorn_lambda, int iters, int cost_kind,
                                 float* f_aa, float* f_bb, float* f_ab, float* f_ba, long ldo,
                                 float* entropy, double* dist, double* stats,
                                 void* workspace, size_t workspace_bytes, void* stream);

/*
 * Row-range variant for data-parallel training: the six problems are solved in full, but only
 * rows [row_begin, row_begin + row_count) of the [2N, D] outputs are produced (into
 * [row_count, D] buffers) -- the rows of the samples a rank owns; the range must lie inside one
 * mini-batch.  dist is the cancellation-free closed form
 * [sum_ab (sum(M)-<M,C>) - ...] / (4N) of the same quantity (cosine cost only).
 * K_pre (nullable): the six log-kernels [6][N][N] = -lambda*cost in the reference's problem
 * order, when the caller already has them (ranks compute their own row slices with
 * otgan_cost_matrix_f32 and all-gather them, as the reference shards matching.py:29-39).
 */
int otgan_matching_two_batch_rows_f32(const float* fa, const float* fb, int N, int D, long ldf,
                                      float sinkhorn_lambda, int iters, int row_begin,
                                      int row_count, const float* K_pre, float* f_aa, float* f_bb,
                                      float* f_ab, float* f_ba, long ldo, float* entropy,
                                      double* dist, double* stats, void* workspace,
                                      size_t workspace_bytes, void* stream);

/*
 * Training-mode two-batch matching: the injected gradients of the reference's step directly --
 *   grad_a = f_aa - f_ab   (train.py:111, generated shards)      grad_b = f_bb - f_ba   (train.py:125-126, data shards)
 * -- as [2N, D] arrays (leading dimension ldo), grad_b nullable (generator steps: five out of six, train.py:214).  The
 * four matched arrays are never formed: each half of a difference is ONE plan application with three terms; dist is
 * the closed form [2 T(a1a2) + 2 T(b2b1) - T(a1b1) - T(a1b2) - T(a2b1) - T(a2b2)] / (4N), T = sum(M) - <M,C>, from
 * the Sinkhorn kernel's statistics (no pass over the features).  Cosine cost.  Workspace:
 * otgan_matching_grad_workspace_bytes.  The _rows_ variant produces rows [row_begin, +row_count) (inside one
 * mini-batch) into [row_count, D] buffers, K_pre as in otgan_matching_two_batch_rows_f32.
 */
size_t otgan_matching_grad_workspace_bytes(int N, int D);
int otgan_matching_two_batch_grad_f32(const float* fa, const float* fb, int N, int D, long ldf,
                                      float sinkhorn_lambda, int iters, float* grad_a, float* grad_b, long ldo,
                                      float* entropy, double* dist, double* stats, void* workspace,
                                      size_t workspace_bytes, void* stream);
int otgan_matching_two_batch_rows_grad_f32(const float* fa, const float* fb, int N, int D, long ldf,
                                           float sinkhorn_lambda, int iters, int row_begin, int row_count,
                                           const float* K_pre, float* grad_a, float* grad_b, long ldo, float* entropy,
                                           double* dist, double* stats, void* workspace, size_t workspace_bytes,
                                           void* stream);

/*
 * One split of the features per step for a data-parallel rank (round 5).  A rank of the global matching scope multiplies the
 * gathered features in two library calls per step -- its three cost row slices (matching.py:29-39) and the plans applied to
 * its own rows (:64-83) -- and each call used to split the fp32 blocks it read into the GEMM engine's operand (two scaled
 * fp16 planes) itself.  The STACK is that operand as an object of the caller: six N-row blocks [a1 b1 b2 a2 a1 b1] (the layout
 * inside otgan_matching_two_batch_grad_f32: every injected difference contracts over three adjacent blocks).
 *   otgan_matching_stack_bytes      size of the buffer; 0 when the engine does not take the shape (N < 256, ...): use the
 *                                   entries above.
 *   otgan_matching_stack_split_f32  fills the given stack-row ranges (multiples of 32 rows) from fa / fb [2N, D]; a
 *                                   first-half rank in a generator step needs rows [N, 4N) (the Y blocks of its cost slices
 *                                   and the contraction blocks of g(a1)) plus its own rows of a1; all ranges of a step in ONE
 *                                   call (they share the operand's scale).
 *   otgan_cost_slices_stack_f32     K[p] = -lambda (1 - x.y) for P <= 6 problems: X = stack rows [xrow[p], +nrows),
 *                                   Y = the N-row block at stack row yrow[p] -> K [P][nrows][N].
 *   otgan_matching_two_batch_rows_grad_stack_f32   otgan_matching_two_batch_rows_grad_f32 reading the stack (K_pre required).
 */
size_t otgan_matching_stack_bytes(int N, int D);
int otgan_matching_stack_split_f32(const float* fa, const float* fb, int N, int D, long ldf, int nranges,
                                   const int* range_begin, const int* range_rows, void* stack, void* stream);
size_t otgan_cost_slices_stack_workspace_bytes(int P, int nrows, int N, int D);
int otgan_cost_slices_stack_f32(const void* stack, int N, int D, int P, const long* xrow, const long* yrow, int nrows,
                                float sinkhorn_lambda, float* K, void* workspace, size_t workspace_bytes, void* stream);
int otgan_matching_two_batch_rows_grad_stack_f32(const void* stack, int N, int D, float sinkhorn_lambda, int iters,
                                                 int row_begin, int row_count, const float* K_pre, float* grad_a,
                                                 float* grad_b, long ldo, float* entropy, double* dist, double* stats,
                                                 void* workspace, size_t workspace_bytes, void* stream);

/*
 * Single-batch matching (matching.py:88-136): fa, fb [n, D]; 999 is added to the a-a and
 * b-b cost diagonals.  stats: [3][4] doubles (aa, bb, ab).
 */
int otgan_matching_single_batch_f32(const float* fa, const float* fb, int n, int D, long ldf,
                                    float sinkhorn_lambda, int iters,
                                    float* f_aa, float* f_bb, float* f_ab, float* f_ba, long ldo,
                                    float* entropy, double* dist, double* stats,
                                    void* workspace, size_t workspace_bytes, void* stream);

/*
 * Training-mode single-batch matching: what the --single_batch step consumes (train.py:111,125-126 on the outputs of
 * matching.py:131-134) -- grad_a = f_aa - f_ab = M_aa a - M_ab b and grad_b = f_bb - f_ba = M_bb b - M_ab^T a (grad_b
 * nullable: generator steps) as two-term plan applications, dist = (T_bb + T_aa - 2 T_ab) / (2n), T = sum(M) - <M,C>, from the
 * Sinkhorn statistics.  The _rows_ variant produces rows [row_begin, +row_count) of [0, n) -- the samples of one
 * data-parallel rank -- into [row_count, D] buffers; K_pre (nullable) = the three [n, n] log-kernels a-a, b-b (both with
 * -lambda*999 on the diagonal), a-b, e.g. assembled from the ranks' row slices (the reference shards exactly these
 * GEMMs over its towers, matching.py:99-104).  Workspace: otgan_matching_single_batch_grad_workspace_bytes.
 */
size_t otgan_matching_single_batch_grad_workspace_bytes(int n, int D);
int otgan_matching_single_batch_grad_f32(const float* fa, const float* fb, int n, int D, long ldf, float sinkhorn_lambda,
                                         int iters, float* grad_a, float* grad_b, long ldo, float* entropy, double* dist,
                                         double* stats, void* workspace, size_t workspace_bytes, void* stream);
int otgan_matching_single_batch_rows_grad_f32(const float* fa, const float* fb, int n, int D, long ldf,
                                              float sinkhorn_lambda, int iters, int row_begin, int row_count,
                                              const float* K_pre, float* grad_a, float* grad_b, long ldo, float* entropy,
                                              double* dist, double* stats, void* workspace, size_t workspace_bytes,
                                              void* stream);

/* Staged entry points (same kernels, exposed for parity tests and custom pipelines). */

/* K[n,m] = -lambda * (cost(X[n,D], Y[m,D]) + diag_add * I)     (matching.py:31,50,109) */
size_t otgan_cost_matrix_workspace_bytes(int n, int m, int D);
int otgan_cost_matrix_f32(const float* X, const float* Y, int n, int m, int D, long ldf,
                          float sinkhorn_lambda, int cost_kind, float diag_add, float* K,
                          void* workspace, size_t workspace_bytes, void* stream);

/* The same for P <= 6 blocks of one shape in ONE launch: K[P][n][m], K[p] = -lambda * (cost(X[p], Y[p]) + diag_add[p] * I)
 * (diag_add nullable).  X / Y are HOST arrays of P device pointers; blocks named more than once are staged once.
 * This is what a data-parallel rank calls for its three row slices of the cost matrices (the reference shards
 * them over the towers the same way, matching.py:29-39). */
size_t otgan_cost_matrix_batched_workspace_bytes(int P, int n, int m, int D);
int otgan_cost_matrix_batched_f32(const float* const* X, const float* const* Y, int P, int n, int m, int D,
                                  long ldf, float sinkhorn_lambda, int cost_kind, const float* diag_add,
                                  float* K, void* workspace, size_t workspace_bytes, void* stream);

/* P log-kernels K[P][n][m] -> plans M[P][n][m], transposed plans MT[P][m][n] and stats[P][4]
 * (matching.py:52-57).  lambda is only used to report <M,C> with C = -K/lambda. */
size_t otgan_sinkhorn_workspace_bytes(int P, int n, int m);
int otgan_sinkhorn_plan_f32(const float* K, int P, int n, int m, int iters,
                            float sinkhorn_lambda, float* plan, float* planT, double* stats,
                            void* workspace, size_t workspace_bytes, void* stream);

/* out[rows, D] = alpha * plan[rows, kdim] . feat[kdim, D]       (matching.py:64-75) */
int otgan_plan_apply_f32(const float* plan, long ldp, int rows, int kdim, const float* feat,
                         long ldf, int D, float alpha, float* out, long ldo, void* stream);

/* dist = (sum(b*bb) + sum(a*aa) - 2 sum(a*ab)) / denom over contiguous [rows, D] arrays,
 * fp64 accumulation (matching.py:147-152).  scratch3: 3 doubles of device scratch. */
int otgan_calc_distance_f32(const float* a, const float* b, const float* aa, const float* bb,
                            const float* ab, long rows, int D, double denom, double* dist,
                            double* scratch3, void* stream);

/* ---------------------------------------------------------------------------------------
 * Layer kernels (generator / critic blocks).  Activations NHWC, weights HWIO.
 * Replace reference utils/nn.py:103-183 (weight norm), :190-206 (pre-activation),
 * :234-241 (conv), :208-209 (dense), models/dcgan.py:16-19,35-36 (feature head, GLU).
 * Declared in otgan_layers.h (included below) to keep this header readable.
 * ------------------------------------------------------------------------------------- */
#include "otgan_layers.h"

#ifdef __cplusplus
}
#endif
#endif /* OTGAN_H */
