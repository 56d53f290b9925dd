/*
 * doa_oracle.c — plain-C CPU restatement of the gr-doa hot path.
 *
 * TEST / BASELINE INFRASTRUCTURE ONLY.  Nothing in the product (gr-doa_amd/, include/) links,
 * loads or calls this file; it is used by tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg.  PARITY STATUS: parity unpinned against the reference's own outputs (the
 * reference stores no golden vectors and cannot be built here: Armadillo, GNU Radio and Octave
 * are absent) — see the header of oracle/doa_oracle.py for what pins the oracle instead; this C
 * file is itself checked against that numpy/LAPACK oracle in tests/test_cpu_oracle_c.py.
 *
 * Each routine follows the per-item algorithm of the reference block it names (paths relative
 * to the reference tree):
 *   oracle_autocorrelate   lib/autocorrelate_impl.cc:83-118
 *   oracle_music_*         lib/MUSIC_lin_array_impl.cc:47-87,98-104,108-150
 *   oracle_find_local_max  lib/find_local_max_impl.cc:47-71,80-165,167-194 (+ _impl.h:53-58)
 * The Hermitian eigendecomposition the reference obtains from LAPACK cheevd (via arma::eig_sym,
 * lib/MUSIC_lin_array_impl.cc:128) is, by default, a cyclic complex Jacobi written here (no
 * LAPACK headers exist in the image); oracle_use_lapack() can bind the very routines Armadillo
 * forwards to (cheevd_, cgemm_) from a BLAS/LAPACK shared object at run time.
 */
#include <complex.h>
#include <dlfcn.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef float complex cf32;
#define ORACLE_MAX_N 64

/* ------------------------------------------------------------------------------------------ */
/* optional LAPACK/BLAS binding (Fortran ABI)                                                   */
/* ------------------------------------------------------------------------------------------ */
typedef void (*cheevd_fn)(const char *jobz, const char *uplo, const int *n, cf32 *a, const int *lda,
                          float *w, cf32 *work, const int *lwork, float *rwork, const int *lrwork,
                          int *iwork, const int *liwork, int *info, size_t, size_t);
typedef void (*cgemm_fn)(const char *ta, const char *tb, const int *m, const int *n, const int *k,
                         const cf32 *alpha, const cf32 *a, const int *lda, const cf32 *b,
                         const int *ldb, const cf32 *beta, cf32 *c, const int *ldc, size_t, size_t);
static cheevd_fn g_cheevd = NULL;
static cgemm_fn g_cgemm = NULL;

/* Bind cheevd/cgemm from `path` (e.g. scipy's bundled OpenBLAS, symbol prefix "scipy_").
 * Returns 0 on success.  Passing NULL unbinds (back to the built-in Jacobi / loops). */
int oracle_use_lapack(const char *path, const char *prefix)
{
    g_cheevd = NULL;
    g_cgemm = NULL;
    if (!path) return 0;
    void *h = dlopen(path, RTLD_NOW | RTLD_LOCAL);
    if (!h) return -1;
    char name[128];
    snprintf(name, sizeof name, "%scheevd_", prefix ? prefix : "");
    cheevd_fn f1 = (cheevd_fn)dlsym(h, name);
    snprintf(name, sizeof name, "%scgemm_", prefix ? prefix : "");
    cgemm_fn f2 = (cgemm_fn)dlsym(h, name);
    if (!f1 || !f2) return -2;
    /* keep OpenBLAS single-threaded inside our own OpenMP loops */
    snprintf(name, sizeof name, "%sopenblas_set_num_threads", prefix ? prefix : "");
    void (*setn)(int) = (void (*)(int))dlsym(h, name);
    if (!setn) setn = (void (*)(int))dlsym(h, "openblas_set_num_threads");
    if (setn) setn(1);
    g_cheevd = f1;
    g_cgemm = f2;
    return 0;
}
int oracle_lapack_bound(void) { return g_cheevd != NULL; }

/* ------------------------------------------------------------------------------------------ */
/* autocorrelate                                                                                */
/* ------------------------------------------------------------------------------------------ */
/* in[k] points at the first sample of window 0 of stream k (history included), exactly like
 * input_items[k] in general_work; window i starts at in[k] + i*(K-ovl) (:98).  out receives
 * n_out column-major N x N matrices (:103). */
int oracle_autocorrelate(const cf32 *const *in, int N, int K, int ovl, int avg_method, int n_out,
                         cf32 *out)
{
    if (N <= 0 || N > ORACLE_MAX_N || K <= 0 || ovl >= K) return -1;
    const int S = K - ovl;
    const float invK = (float)(1.0 / K);
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n_out; i++) {
        cf32 *R = out + (size_t)i * N * N;
        if (g_cgemm) {
            /* the reference's path: memcpy into K x N (:95-100), conj(X) temporary, cgemm('T','N') */
            cf32 *X = (cf32 *)malloc(sizeof(cf32) * 2 * (size_t)K * N);
            cf32 *Xc = X + (size_t)K * N;
            for (int k = 0; k < N; k++) memcpy(X + (size_t)k * K, in[k] + (size_t)i * S, sizeof(cf32) * K);
            for (size_t t = 0; t < (size_t)K * N; t++) Xc[t] = conjf(X[t]);
            cf32 alpha = invK, beta = 0.0f;
            g_cgemm("T", "N", &N, &N, &K, &alpha, X, &K, Xc, &K, &beta, R, &N, 1, 1);
            free(X);
        } else {
            for (int b = 0; b < N; b++) {
                const cf32 *xb = in[b] + (size_t)i * S;
                for (int a = 0; a < N; a++) {
                    const cf32 *xa = in[a] + (size_t)i * S;
                    float re = 0.0f, im = 0.0f;
                    for (int t = 0; t < K; t++) {          /* x_a[t] * conj(x_b[t])  (:106) */
                        float ar = crealf(xa[t]), ai = cimagf(xa[t]);
                        float br = crealf(xb[t]), bi = cimagf(xb[t]);
                        re += ar * br + ai * bi;
                        im += ai * br - ar * bi;
                    }
                    R[a + (size_t)b * N] = (re * invK) + (im * invK) * I;
                }
            }
        }
        if (avg_method == 1) {                              /* :107-108 (second term /K again) */
            cf32 T[ORACLE_MAX_N * ORACLE_MAX_N];
            const float h = 0.5f, hk = (float)(0.5 / K);
            for (int b = 0; b < N; b++)
                for (int a = 0; a < N; a++)
                    T[a + b * N] = h * R[a + (size_t)b * N] + hk * conjf(R[(N - 1 - a) + (size_t)(N - 1 - b) * N]);
            memcpy(R, T, sizeof(cf32) * N * N);
        }
    }
    return n_out;
}

/* ------------------------------------------------------------------------------------------ */
/* MUSIC tables (constructor)                                                                    */
/* ------------------------------------------------------------------------------------------ */
/* array_loc (:57-61), theta grid (:64-72), steering table A[n + N*i] = a_i[n] (:75-86,98-104) */
void oracle_music_tables(float norm_spacing, int N, int P, float *array_loc, float *theta, cf32 *A)
{
    for (int nn = 0; nn < N; nn++) array_loc[nn] = (float)(norm_spacing * 0.5 * (N - 1 - 2 * nn));
    theta[0] = 0.0f;
    float theta_prev = 0.0f, th;
    for (int ii = 1; ii < P; ii++) {
        th = (float)(theta_prev + 180.0 / P);
        theta_prev = th;
        theta[ii] = (float)(M_PI * th / 180.0);
    }
    for (int ii = 0; ii < P; ii++) {
        float k = (float)(-1.0 * 2 * M_PI * cos((double)theta[ii]));
        for (int nn = 0; nn < N; nn++) {
            float ph = k * array_loc[nn];
            A[nn + (size_t)N * ii] = cosf(ph) + sinf(ph) * I;
        }
    }
}

/* ------------------------------------------------------------------------------------------ */
/* Hermitian EVD: cyclic complex Jacobi on the upper triangle, ascending eigenvalues           */
/* ------------------------------------------------------------------------------------------ */
static void herm_evd_jacobi(const cf32 *Rin, int N, float *w, cf32 *V)
{
    cf32 A[ORACLE_MAX_N * ORACLE_MAX_N];
    /* only the upper triangle of the input is significant (cheevd uplo='U') */
    for (int c = 0; c < N; c++)
        for (int r = 0; r < N; r++)
            A[r + c * N] = (r < c) ? Rin[r + c * N] : (r == c ? crealf(Rin[r + c * N]) : conjf(Rin[c + r * N]));
    for (int c = 0; c < N; c++)
        for (int r = 0; r < N; r++) V[r + c * N] = (r == c) ? 1.0f : 0.0f;
    for (int sweep = 0; sweep < 30; sweep++) {
        float off = 0.0f, diag = 0.0f;
        for (int c = 0; c < N; c++)
            for (int r = 0; r < N; r++) {
                float m = crealf(A[r + c * N]) * crealf(A[r + c * N]) + cimagf(A[r + c * N]) * cimagf(A[r + c * N]);
                if (r == c) diag += m; else off += m;
            }
        if (off <= 1e-30f || off <= 1e-15f * diag) break;
        for (int p = 0; p < N - 1; p++)
            for (int q = p + 1; q < N; q++) {
                cf32 apq = A[p + q * N];
                float g = cabsf(apq);
                if (g == 0.0f) continue;
                float app = crealf(A[p + p * N]), aqq = crealf(A[q + q * N]);
                float tau = (aqq - app) / (2.0f * g);
                float t = (tau >= 0.0f ? 1.0f : -1.0f) / (fabsf(tau) + sqrtf(1.0f + tau * tau));
                float c = 1.0f / sqrtf(1.0f + t * t), s = t * c;
                cf32 ph = apq / g;                      /* e^{i phi} */
                /* A <- J^H A J,  J[p][p]=c, J[p][q]=s*ph, J[q][p]=-s*conj(ph), J[q][q]=c */
                for (int k = 0; k < N; k++) {           /* columns p,q:  A J */
                    cf32 akp = A[k + p * N], akq = A[k + q * N];
                    A[k + p * N] = c * akp - s * conjf(ph) * akq;
                    A[k + q * N] = s * ph * akp + c * akq;
                }
                for (int k = 0; k < N; k++) {           /* rows p,q:  J^H A */
                    cf32 apk = A[p + k * N], aqk = A[q + k * N];
                    A[p + k * N] = c * apk - s * ph * aqk;
                    A[q + k * N] = s * conjf(ph) * apk + c * aqk;
                }
                A[p + q * N] = 0.0f;
                A[q + p * N] = 0.0f;
                A[p + p * N] = crealf(A[p + p * N]);
                A[q + q * N] = crealf(A[q + q * N]);
                for (int k = 0; k < N; k++) {           /* V <- V J */
                    cf32 vkp = V[k + p * N], vkq = V[k + q * N];
                    V[k + p * N] = c * vkp - s * conjf(ph) * vkq;
                    V[k + q * N] = s * ph * vkp + c * vkq;
                }
            }
    }
    /* ascending order (eig_sym contract) */
    int idx[ORACLE_MAX_N];
    for (int i = 0; i < N; i++) { idx[i] = i; w[i] = crealf(A[i + i * N]); }
    for (int i = 1; i < N; i++) {
        int k = idx[i]; float wk = w[k]; int j = i - 1;
        while (j >= 0 && w[idx[j]] > wk) { idx[j + 1] = idx[j]; j--; }
        idx[j + 1] = k;
    }
    cf32 Vs[ORACLE_MAX_N * ORACLE_MAX_N]; float ws[ORACLE_MAX_N];
    for (int c = 0; c < N; c++) { ws[c] = w[idx[c]]; memcpy(Vs + c * N, V + idx[c] * N, sizeof(cf32) * N); }
    memcpy(V, Vs, sizeof(cf32) * N * N);
    memcpy(w, ws, sizeof(float) * N);
}

static int herm_evd(const cf32 *Rin, int N, float *w, cf32 *V)
{
    if (g_cheevd) {
        int lwork = 2 * N + N * N, lrwork = 1 + 5 * N + 2 * N * N, liwork = 3 + 5 * N, info = 0;
        cf32 work[2 * ORACLE_MAX_N + ORACLE_MAX_N * ORACLE_MAX_N];
        float rwork[1 + 5 * ORACLE_MAX_N + 2 * ORACLE_MAX_N * ORACLE_MAX_N];
        int iwork[3 + 5 * ORACLE_MAX_N];
        memcpy(V, Rin, sizeof(cf32) * N * N);           /* in_matrix copy, :124 */
        g_cheevd("V", "U", &N, V, &N, w, work, &lwork, rwork, &lrwork, iwork, &liwork, &info, 1, 1);
        return info;
    }
    herm_evd_jacobi(Rin, N, w, V);
    return 0;
}

/* P_N = U_N U_N^H, U_N = first N-M eigenvectors (:131-133).  Exposed for tests. */
int oracle_noise_projector(const cf32 *R, int N, int M, cf32 *PN)
{
    if (N <= 0 || N > ORACLE_MAX_N || M < 0 || M >= N) return -1;
    float w[ORACLE_MAX_N];
    cf32 V[ORACLE_MAX_N * ORACLE_MAX_N];
    int info = herm_evd(R, N, w, V);
    if (info) return -2;
    const int nn = N - M;
    for (int b = 0; b < N; b++)
        for (int a = 0; a < N; a++) {
            cf32 acc = 0.0f;
            for (int k = 0; k < nn; k++) acc += V[a + k * N] * conjf(V[b + k * N]);
            PN[a + b * N] = acc;
        }
    return 0;
}

/* one item of MUSIC_lin_array::work (:121-144): spectrum from one covariance matrix */
static int music_item(const cf32 *R, int N, int M, int P, const cf32 *A, float *out)
{
    cf32 PN[ORACLE_MAX_N * ORACLE_MAX_N];
    int rc = oracle_noise_projector(R, N, M, PN);
    if (rc) return rc;
    float mx = -INFINITY;
    for (int ii = 0; ii < P; ii++) {                       /* :137-141 */
        const cf32 *a = A + (size_t)N * ii;
        cf32 t[ORACLE_MAX_N];
        for (int n = 0; n < N; n++) {                       /* t = a^H P_N */
            cf32 acc = 0.0f;
            for (int m = 0; m < N; m++) acc += conjf(a[m]) * PN[m + n * N];
            t[n] = acc;
        }
        cf32 q = 0.0f;
        for (int n = 0; n < N; n++) q += t[n] * a[n];       /* (a^H P_N) a */
        float o = (float)(1.0 / (double)crealf(q));
        out[ii] = o;
        if (o > mx) mx = o;
    }
    for (int ii = 0; ii < P; ii++) out[ii] = 10.0f * log10f(out[ii] / mx);   /* :142 */
    return 0;
}

int oracle_music_lin_array(const cf32 *R_items, int n_items, float norm_spacing, int M, int N, int P,
                           float *spec)
{
    if (N <= 0 || N > ORACLE_MAX_N || M <= 0 || M >= N || P <= 0) return -1;
    float *loc = (float *)malloc(sizeof(float) * N), *theta = (float *)malloc(sizeof(float) * P);
    cf32 *A = (cf32 *)malloc(sizeof(cf32) * (size_t)N * P);
    oracle_music_tables(norm_spacing, N, P, loc, theta, A);
    int bad = 0;
#pragma omp parallel for schedule(static) reduction(| : bad)
    for (int i = 0; i < n_items; i++)
        bad |= music_item(R_items + (size_t)i * N * N, N, M, P, A, spec + (size_t)i * P) != 0;
    free(loc); free(theta); free(A);
    return bad ? -2 : n_items;
}

/* ------------------------------------------------------------------------------------------ */
/* find_local_max                                                                               */
/* ------------------------------------------------------------------------------------------ */
void oracle_find_local_max_x_axis(int L, float x_min, float x_max, float *x)
{
    x[0] = x_min;
    float x_prev = x_min, xr = x_max - x_min;
    for (int ii = 1; ii < L; ii++) { float v = x_prev + xr / L; x_prev = v; x[ii] = v; }   /* :60-69 */
}

/* arma index_max (op_max::direct_max): best starts at -inf and is replaced only by a strictly
 * greater element -> first occurrence of the maximum; NaNs never win; nothing > -inf -> 0 */
static int argmax_first(const float *v, int L)
{
    int k = 0;
    float best = -INFINITY;
    for (int i = 0; i < L; i++) if (v[i] > best) { best = v[i]; k = i; }
    return k;
}

/* pk[M] <- peak indices of one vector (:80-165, or index_max for M==1: _impl.h:53-56).
 * s: scratch of L floats, cand: scratch of L ints. */
static void peak_indices(const float *v, int L, int M, int *pk, float *s, int *cand)
{
    if (M == 1) { pk[0] = argmax_first(v, L); return; }
    const int Ls = L - 1;
    for (int i = 0; i < Ls; i++) { float d = v[i + 1] - v[i]; s[i] = (d > 0) ? 1.0f : ((d < 0) ? -1.0f : 0.0f); }
    for (int i = Ls - 1; i >= 0; i--)                        /* flats, right to left (:94-107) */
        if (s[i] == 0.0f) { int nx = (i + 1 < Ls - 1) ? i + 1 : Ls - 1; s[i] = (s[nx] >= 0) ? 1.0f : -1.0f; }
    int nc = 0;
    for (int i = 0; i + 1 < Ls; i++) if (s[i + 1] - s[i] == -2.0f) cand[nc++] = i + 1;   /* :114 */
    /* top-M of the candidates by value, descending; ties -> lowest index first */
    int order[16];
    int nsel = nc < M ? nc : M;
    for (int j = 0; j < nsel; j++) {
        int best = -1;
        for (int c = 0; c < nc; c++) {
            int used = 0;
            for (int u = 0; u < j; u++) used |= (order[u] == c);
            if (used) continue;
            if (best < 0 || v[cand[c]] > v[cand[best]]) best = c;
        }
        order[j] = best;
    }
    if (nc >= M) { for (int j = 0; j < M; j++) pk[j] = cand[order[j]]; return; }   /* :141-144 */
    int fill = (nc == 0) ? argmax_first(v, L) : order[0];   /* :150-153 — list position, as written */
    for (int j = 0; j < M; j++) pk[j] = (j < nc) ? cand[order[j]] : fill;
}

int oracle_find_local_max(const float *in, int n_items, int M, int L, float x_min, float x_max,
                          float *max_vals, float *arg_max)
{
    if (M <= 0 || M > 16 || L < 3) return -1;
    float *x = (float *)malloc(sizeof(float) * L);
    oracle_find_local_max_x_axis(L, x_min, x_max, x);
#pragma omp parallel
    {
        float *s = (float *)malloc(sizeof(float) * L);
        int *cand = (int *)malloc(sizeof(int) * L);
        int pk[16];
#pragma omp for schedule(static)
        for (int i = 0; i < n_items; i++) {
            const float *v = in + (size_t)i * L;
            peak_indices(v, L, M, pk, s, cand);
            float loc[16];
            for (int j = 0; j < M; j++) { max_vals[(size_t)i * M + j] = v[pk[j]]; loc[j] = x[pk[j]]; }
            for (int a = 1; a < M; a++) {                    /* sort(..., "descend") (:188) */
                float t = loc[a]; int b = a - 1;
                while (b >= 0 && loc[b] < t) { loc[b + 1] = loc[b]; b--; }
                loc[b + 1] = t;
            }
            for (int j = 0; j < M; j++) arg_max[(size_t)i * M + j] = loc[j];
        }
        free(s); free(cand);
    }
    free(x);
    return n_items;
}

/* ------------------------------------------------------------------------------------------ */
/* whole path: autocorrelate -> MUSIC -> find_local_max(M, P, 0, 180)                           */
/* ------------------------------------------------------------------------------------------ */
int oracle_music_pipeline(const cf32 *const *in, int N, int K, int ovl, int avg_method, float norm_spacing,
                          int M, int P, int n_out, cf32 *R, float *spec, float *max_vals, float *arg_max)
{
    int rc = oracle_autocorrelate(in, N, K, ovl, avg_method, n_out, R);
    if (rc < 0) return rc;
    rc = oracle_music_lin_array(R, n_out, norm_spacing, M, N, P, spec);
    if (rc < 0) return rc;
    return oracle_find_local_max(spec, n_out, M, P, 0.0f, 180.0f, max_vals, arg_max);
}

int oracle_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
void oracle_set_num_threads(int n)
{
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}
