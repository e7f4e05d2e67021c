/*
 * oracle/sinkhorn_c.c -- plain-C CPU restatement of the OT-GAN matching block.
 *
 * TEST INFRASTRUCTURE ONLY.  Linked/loaded only by tests/, __graft_entry__.smoke() and
 * the cpu_baseline leg of bench.py -- never by the product path (ot-gan_amd/).
 *
 * Restates (does not copy) the algorithm of the reference:
 *   cost + Sinkhorn + matched products   /root/reference/utils/matching.py:29-83
 *   single-batch variant                 /root/reference/utils/matching.py:95-134
 *   distance                             /root/reference/utils/matching.py:139-153
 *   toy sq-Euclid cost                   /root/reference/toy_example/matching_cpu.py:17-45
 *
 * Arithmetic: fp32 storage and products (as TensorFlow's default), max-shifted
 * log-sum-exp (tf.reduce_logsumexp), in-place row/column sweeps in exactly the
 * reference order (rows, then columns, L times; then a row softmax).  Scalar
 * reductions (entropy, distance) accumulate in double.
 *
 * Pinned by tests/test_oracle_c.py against tests/golden/ (vectors produced by the
 * reference's own code, see oracle/make_golden.py).
 *
 * Build: make -C oracle      (gcc -O3 -fopenmp -shared -fPIC)
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define CLONES __attribute__((target_clones("avx2,fma", "default")))

int otgan_oracle_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* C[n,m] = X[n,d] . Y[m,d]^T */
CLONES static void gemm_nt(const float *X, const float *Y, float *C, int n, int m, int d) {
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; ++i) {
        const float *x = X + (size_t)i * d;
        for (int j = 0; j < m; ++j) {
            const float *y = Y + (size_t)j * d;
            float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            int k = 0;
            for (; k + 8 <= d; k += 8)
                for (int u = 0; u < 8; ++u) acc[u] += x[k + u] * y[k + u];
            float s = 0.f;
            for (; k < d; ++k) s += x[k] * y[k];
            for (int u = 0; u < 8; ++u) s += acc[u];
            C[(size_t)i * m + j] = s;
        }
    }
}

/* O[n,d] (+)= alpha * M[n,m] . F[m,d]   (accumulate != 0 adds into O) */
CLONES static void gemm_nn(const float *M, const float *F, float *O, int n, int m, int d,
                           float alpha, int accumulate) {
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; ++i) {
        float *o = O + (size_t)i * d;
        if (!accumulate) memset(o, 0, sizeof(float) * d);
        for (int j = 0; j < m; ++j) {
            const float w = alpha * M[(size_t)i * m + j];
            const float *f = F + (size_t)j * d;
            for (int k = 0; k < d; ++k) o[k] += w * f[k];
        }
    }
}

/* O[m,d] (+)= alpha * M[n,m]^T . F[n,d] */
CLONES static void gemm_tn(const float *M, const float *F, float *O, int n, int m, int d,
                           float alpha, int accumulate) {
#pragma omp parallel for schedule(static)
    for (int j = 0; j < m; ++j) {
        float *o = O + (size_t)j * d;
        if (!accumulate) memset(o, 0, sizeof(float) * d);
        for (int i = 0; i < n; ++i) {
            const float w = alpha * M[(size_t)i * m + j];
            const float *f = F + (size_t)i * d;
            for (int k = 0; k < d; ++k) o[k] += w * f[k];
        }
    }
}

/* cost kinds */
enum { COST_COSINE = 0, COST_SQEUCLID_MEAN = 1 };

static void make_cost(const float *X, const float *Y, float *C, int n, int m, int d, int kind,
                      float diag_add) {
    gemm_nt(X, Y, C, n, m, d);
    if (kind == COST_COSINE) {
        for (size_t t = 0; t < (size_t)n * m; ++t) C[t] = 1.f - C[t];
    } else {
        float *xs = (float *)malloc(sizeof(float) * n), *ys = (float *)malloc(sizeof(float) * m);
        for (int i = 0; i < n; ++i) {
            double s = 0;
            for (int k = 0; k < d; ++k) s += (double)X[(size_t)i * d + k] * X[(size_t)i * d + k];
            xs[i] = (float)(0.5 * s / d);
        }
        for (int j = 0; j < m; ++j) {
            double s = 0;
            for (int k = 0; k < d; ++k) s += (double)Y[(size_t)j * d + k] * Y[(size_t)j * d + k];
            ys[j] = (float)(0.5 * s / d);
        }
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < m; ++j)
                C[(size_t)i * m + j] = xs[i] + ys[j] - C[(size_t)i * m + j] / (float)d;
        free(xs);
        free(ys);
    }
    if (diag_add != 0.f) {
        int q = n < m ? n : m;
        for (int i = 0; i < q; ++i) C[(size_t)i * m + i] += diag_add;
    }
}

/* In-place log-domain Sinkhorn (matching.py:50-57).  On return A holds the plan M
 * (row softmax of the final log_a); returns the mean row entropy. */
static double sinkhorn_inplace(float *A, int n, int m, float lam, int iters) {
    for (size_t t = 0; t < (size_t)n * m; ++t) A[t] = -lam * A[t];
    float *col = (float *)malloc(sizeof(float) * m);
    for (int it = 0; it < iters; ++it) {
#pragma omp parallel for schedule(static)
        for (int i = 0; i < n; ++i) {
            float *a = A + (size_t)i * m;
            float mx = a[0];
            for (int j = 1; j < m; ++j) mx = a[j] > mx ? a[j] : mx;
            float s = 0.f;
            for (int j = 0; j < m; ++j) s += expf(a[j] - mx);
            const float l = logf(s) + mx;
            for (int j = 0; j < m; ++j) a[j] -= l;
        }
#pragma omp parallel for schedule(static)
        for (int j = 0; j < m; ++j) {
            float mx = A[j];
            for (int i = 1; i < n; ++i) { float v = A[(size_t)i * m + j]; mx = v > mx ? v : mx; }
            float s = 0.f;
            for (int i = 0; i < n; ++i) s += expf(A[(size_t)i * m + j] - mx);
            col[j] = logf(s) + mx;
        }
#pragma omp parallel for schedule(static)
        for (int i = 0; i < n; ++i) {
            float *a = A + (size_t)i * m;
            for (int j = 0; j < m; ++j) a[j] -= col[j];
        }
    }
    free(col);
    double ent = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : ent)
    for (int i = 0; i < n; ++i) {
        float *a = A + (size_t)i * m;
        float mx = a[0];
        for (int j = 1; j < m; ++j) mx = a[j] > mx ? a[j] : mx;
        float s = 0.f;
        for (int j = 0; j < m; ++j) s += expf(a[j] - mx);
        const float l = logf(s) + mx;
        double h = 0.0;
        for (int j = 0; j < m; ++j) {
            const float lm = a[j] - l;
            const float p = expf(lm);
            h -= (double)p * (double)lm;
            a[j] = p;
        }
        ent += h;
    }
    return ent / n;
}

static double dot_d(const float *a, const float *b, size_t n) {
    double s = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : s)
    for (long t = 0; t < (long)n; ++t) s += (double)a[t] * (double)b[t];
    return s;
}

/*
 * Two-batch matching.  fa, fb: [2N, D] row-major (rows [0,N) = mini-batch 1, [N,2N) = 2).
 * Outputs f_aa, f_bb, f_ab, f_ba: [2N, D].  entropy = mean over the six problems.
 * dist: cosine -> reference calc_distance (sum / (2*2N)); toy -> mean-normalised / 2.
 */
int otgan_oracle_two_batch_f32(const float *fa, const float *fb, int N, int D, float lam,
                               int iters, int cost_kind, float *f_aa, float *f_bb,
                               float *f_ab, float *f_ba, double *entropy, double *dist) {
    const size_t ND = (size_t)N * D;
    const float *fa1 = fa, *fa2 = fa + ND, *fb1 = fb, *fb2 = fb + ND;
    float *P = (float *)malloc(sizeof(float) * (size_t)N * N);
    if (!P) return -1;
    double ent = 0.0;
    /* a1-a2 */
    make_cost(fa1, fa2, P, N, N, D, cost_kind, 0.f);
    ent += sinkhorn_inplace(P, N, N, lam, iters);
    gemm_nn(P, fa2, f_aa, N, N, D, 1.f, 0);
    gemm_tn(P, fa1, f_aa + ND, N, N, D, 1.f, 0);
    /* b2-b1 (rows b2, cols b1) */
    make_cost(fb2, fb1, P, N, N, D, cost_kind, 0.f);
    ent += sinkhorn_inplace(P, N, N, lam, iters);
    gemm_tn(P, fb2, f_bb, N, N, D, 1.f, 0);
    gemm_nn(P, fb1, f_bb + ND, N, N, D, 1.f, 0);
    /* a1-b1 */
    make_cost(fa1, fb1, P, N, N, D, cost_kind, 0.f);
    ent += sinkhorn_inplace(P, N, N, lam, iters);
    gemm_nn(P, fb1, f_ab, N, N, D, 0.5f, 0);
    gemm_tn(P, fa1, f_ba, N, N, D, 0.5f, 0);
    /* a1-b2 */
    make_cost(fa1, fb2, P, N, N, D, cost_kind, 0.f);
    ent += sinkhorn_inplace(P, N, N, lam, iters);
    gemm_nn(P, fb2, f_ab, N, N, D, 0.5f, 1);
    gemm_tn(P, fa1, f_ba + ND, N, N, D, 0.5f, 0);
    /* a2-b1 */
    make_cost(fa2, fb1, P, N, N, D, cost_kind, 0.f);
    ent += sinkhorn_inplace(P, N, N, lam, iters);
    gemm_nn(P, fb1, f_ab + ND, N, N, D, 0.5f, 0);
    gemm_tn(P, fa2, f_ba, N, N, D, 0.5f, 1);
    /* a2-b2 */
    make_cost(fa2, fb2, P, N, N, D, cost_kind, 0.f);
    ent += sinkhorn_inplace(P, N, N, lam, iters);
    gemm_nn(P, fb2, f_ab + ND, N, N, D, 0.5f, 1);
    gemm_tn(P, fa2, f_ba + ND, N, N, D, 0.5f, 1);
    free(P);
    *entropy = ent / 6.0;
    const double nd_aa = dot_d(fa, f_aa, 2 * ND), nd_bb = dot_d(fb, f_bb, 2 * ND),
                 nd_ab = dot_d(fa, f_ab, 2 * ND);
    if (cost_kind == COST_COSINE)
        *dist = (nd_bb + nd_aa - 2.0 * nd_ab) / (2.0 * (2.0 * N));
    else
        *dist = (nd_bb + nd_aa - 2.0 * nd_ab) / (2.0 * ND) / 2.0;
    return 0;
}

/* Single-batch matching (matching.py:88-136): fa, fb [n, D]; 999 added on the a-a and
 * b-b diagonals. */
int otgan_oracle_single_batch_f32(const float *fa, const float *fb, int n, int D, float lam,
                                  int iters, float *f_aa, float *f_bb, float *f_ab,
                                  float *f_ba, double *entropy, double *dist) {
    float *P = (float *)malloc(sizeof(float) * (size_t)n * n);
    if (!P) return -1;
    double ent = 0.0;
    make_cost(fa, fa, P, n, n, D, COST_COSINE, 999.f);
    ent += sinkhorn_inplace(P, n, n, lam, iters);
    gemm_nn(P, fa, f_aa, n, n, D, 1.f, 0);
    make_cost(fb, fb, P, n, n, D, COST_COSINE, 999.f);
    ent += sinkhorn_inplace(P, n, n, lam, iters);
    gemm_nn(P, fb, f_bb, n, n, D, 1.f, 0);
    make_cost(fa, fb, P, n, n, D, COST_COSINE, 0.f);
    ent += sinkhorn_inplace(P, n, n, lam, iters);
    gemm_nn(P, fb, f_ab, n, n, D, 1.f, 0);
    gemm_tn(P, fa, f_ba, n, n, D, 1.f, 0);
    free(P);
    *entropy = ent / 3.0;
    const size_t nD = (size_t)n * D;
    *dist = (dot_d(fb, f_bb, nD) + dot_d(fa, f_aa, nD) - 2.0 * dot_d(fa, f_ab, nD)) / (2.0 * n);
    return 0;
}
