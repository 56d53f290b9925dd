// music.hip — K2+K3 (batched Hermitian EVD -> noise projector -> diagonal sums) and K4 (pseudo-
// spectrum scan) of MUSIC_lin_array on gfx950.
//
// Replaces gr::doa::MUSIC_lin_array (reference lib/MUSIC_lin_array_impl.cc):
//   ctor  :47-87,98-104  element positions, float-accumulated theta grid, steering vectors
//   work  :121-144       eig_sym (LAPACK cheevd 'V','U', ascending) -> U_N = first N-M vectors
//                        -> P_N = U_N U_N^H -> out[i] = 1/Re(a_i^H P_N a_i) -> 10 log10(out/max)
//
// How it is laid out here:
//   * EVD (music_evd_kernel): one lane per covariance matrix.  Cyclic complex Jacobi on the full
//     Hermitian matrix held in registers (N <= 4, fully unrolled) or per-lane scratch (N <= 16),
//     in double by default (float selectable: doa_set_evd_precision).  Rotations are built from
//     rsqrt only, so J is unitary to working precision.  Epilogue: rank eigenvalues ascending,
//     P_N = sum over the N-M smallest of v v^H, and the 2N-1 diagonal sums
//         u_l = sum_r P_N[r+l, r]   (u_0 real, u_l complex)
//     written as one 8N-byte record per item (and P_N itself when asked for).  These u_l are exactly
//     the Root-MUSIC polynomial coefficients (lib/rootMUSIC_linear_array_impl.cc:74-79).
//   * Scan (music_scan_kernel): for a ULA a_i^H P_N a_i is Hermitian-Toeplitz in the element index,
//         Q(psi_i) = u_0 + 2 Re sum_{l=1}^{N-1} u_l z_i^l,   z_i = exp(j psi_i), psi_i = k_i d,
//     so each angle costs N-1 complex Horner steps instead of N^2+N complex MACs; that drops the
//     kernel below the fp32 ridge (HBM-bound: 8N B in, 4P B out per item).  One wave owns one item:
//     each lane evaluates 4 consecutive angles per 256-angle chunk (z_i stays in registers across
//     items), the item maximum is a wave all-reduce, and each store instruction writes 1 KiB
//     contiguous.  Values match the reference's a^H P a up to fp32 rounding (~1e-7 relative).
#include "kernels.hpp"
#include "peak_device.hpp"

#include <cmath>
#include <cstdlib>
#include <vector>

namespace doa {

// ---------------------------------------------------------------------------------------------
// host-side tables (constructor work of the reference block)
// ---------------------------------------------------------------------------------------------
int MusicTables::build(float norm_spacing_, int num_targets, int num_ant_ele, int pspectrum_len)
{
    N = num_ant_ele; M = num_targets; P = pspectrum_len; norm_spacing = norm_spacing_;
    // theta grid: float accumulator, sum formed in double (lib/MUSIC_lin_array_impl.cc:64-72)
    std::vector<float> theta(P);
    theta[0] = 0.0f;
    float theta_prev = 0.0f;
    for (int ii = 1; ii < P; ii++) {
        float th = (float)(theta_prev + 180.0 / P);
        theta_prev = th;
        theta[ii] = (float)(M_PI * th / 180.0);
    }
    // amv (:98-104): a_i[n] = exp(j * (-2 pi cos(theta_i)) * loc_n), loc_n = d*0.5*(N-1-2n) (:57-61), so
    // the phase step between neighbouring elements is psi_i = -2 pi cos(theta_i) * d.  Evaluated in
    // double from the float theta grid; the reference's float roundings of the scalar, of loc_n and of
    // each phase (~2e-7 rad) are not replayed (they are part of its own fp32 error budget).
    std::vector<float2> z(P);
    std::vector<double2> zd(P);
    const double d = (double)norm_spacing;
    for (int ii = 0; ii < P; ii++) {
        const double psi = -1.0 * 2 * M_PI * std::cos((double)theta[ii]) * d;
        z[ii] = make_float2((float)std::cos(psi), (float)std::sin(psi));
        zd[ii] = make_double2(std::cos(psi), std::sin(psi));
    }
    int rc = d_z.reserve(sizeof(float2) * (size_t)P);
    if (rc == DOA_OK) rc = d_zd.reserve(sizeof(double2) * (size_t)P);
    if (rc != DOA_OK) return rc;
    DOA_HIP_TRY(hipMemcpy(d_z.p, z.data(), sizeof(float2) * (size_t)P, hipMemcpyHostToDevice));
    DOA_HIP_TRY(hipMemcpy(d_zd.p, zd.data(), sizeof(double2) * (size_t)P, hipMemcpyHostToDevice));
    return DOA_OK;
}

// ---------------------------------------------------------------------------------------------
// K2+K3: batched Hermitian Jacobi EVD, projector, diagonal sums
// ---------------------------------------------------------------------------------------------
template <typename T> struct Real;
template <> struct Real<float> {
    static __device__ __forceinline__ float rsqrt(float x)
    {
        float y = __builtin_amdgcn_rsqf(x);
        float e = fmaf(-x * y, y, 1.0f);
        return fmaf(0.5f * y, e, y);
    }
    static constexpr float tol = 6e-14f;     // (off-norm / diag-norm)^2 at convergence, ~4 eps^2
    static constexpr float tiny = 1e-36f;
    static constexpr float tau_max = 1e15f;
    static constexpr int max_sweeps = 10;
};
template <> struct Real<double> {
    static __device__ __forceinline__ double rsqrt(double x)
    {
        double y = __builtin_amdgcn_rsq(x);
        double e = fma(-x * y, y, 1.0);
        y = fma(0.5 * y, e, y);
        e = fma(-x * y, y, 1.0);
        return fma(0.5 * y, e, y);
    }
    static constexpr double tol = 1e-27;     // off-norm/diag-norm <= 3e-14: eigenvectors good to ~1e-13
    static constexpr double tiny = 1e-290;
    static constexpr double tau_max = 1e140;
    static constexpr int max_sweeps = 14;
};

// Cyclic complex Jacobi on a Hermitian matrix kept as its real diagonal dg[] and strict upper
// triangle (ur, ui)[r][c], r < c (the lower triangle is never formed: A[c][r] = conj(A[r][c])).
// On return dg holds the eigenvalues and the columns of V = (vr, vi) the eigenvectors.
// One rotation (p,q): J[p][p] = J[q][q] = c, J[p][q] = sigma = s e^{j phi}, J[q][p] = -conj(sigma)
// with phi = arg A[p][q] and t = s/c the smaller root of t^2 + 2 tau t - 1 = 0,
// tau = (A[q][q]-A[p][p]) / (2|A[p][q]|).  c, s and e^{j phi} are built from rsqrt only, so J is
// unitary to working precision and no division appears.  A <- J^H A J touches, for every k not in
// {p,q}, the pair (A[k][p], A[k][q]) and the two diagonal entries; V <- V J touches columns p, q.
template <int N, typename T, bool UNROLL>
__device__ __forceinline__ void herm_jacobi(T (&dg)[N], T (&ur)[N][N], T (&ui)[N][N], T (&vr)[N][N], T (&vi)[N][N])
{
    constexpr int U = UNROLL ? N : 1;
    const int max_sweeps = (N <= 4) ? Real<T>::max_sweeps : Real<T>::max_sweeps + 2 * N;
    for (int sweep = 0; sweep < max_sweeps; sweep++) {
        T off = 0, dn = 0;
#pragma unroll U
        for (int p = 0; p < N; p++) {
            dn = fma(dg[p], dg[p], dn);
#pragma unroll U
            for (int q = 0; q < N; q++)
                if (q > p) off += ur[p][q] * ur[p][q] + ui[p][q] * ui[p][q];
        }
        if (!(off > Real<T>::tol * dn) || !(off > Real<T>::tiny)) break;
#pragma unroll U
        for (int p = 0; p < N - 1; p++) {
#pragma unroll U
            for (int q = 1; q < N; q++) {
                if (q <= p) continue;
                const T apr = ur[p][q], api = ui[p][q];
                const T g2 = apr * apr + api * api;
                // branch-free: a pivot that is already (numerically) zero gets the identity rotation
                const bool live = g2 > Real<T>::tiny;
                const T inv_g = Real<T>::rsqrt(live ? g2 : (T)1);
                const T g = live ? g2 * inv_g : (T)0;
                const T phr = apr * inv_g, phi = api * inv_g;          // e^{j phi}
                T tau = (dg[q] - dg[p]) * (T)0.5 * inv_g;
                tau = fmin(fmax(tau, -Real<T>::tau_max), Real<T>::tau_max);
                const T x1 = fma(tau, tau, (T)1);
                const T r = x1 * Real<T>::rsqrt(x1);                    // sqrt(1+tau^2)
                const T h = fabs(tau) + r;                              // 1/|t|
                const T w = Real<T>::rsqrt(fma(h, h, (T)1));
                const T c = live ? h * w : (T)1;
                const T s = live ? copysign(w, tau) : (T)0;
                const T spr = s * phr, spi = s * phi;                   // sigma
                // 2x2 block: a_pp' = c^2 a_pp - 2cs g + s^2 a_qq,  a_qq' = s^2 a_pp + 2cs g + c^2 a_qq,  a_pq' = 0
                {
                    const T cc = c * c, ss = s * s, csg = (T)2 * c * s * g;
                    const T app = dg[p], aqq = dg[q];
                    dg[p] = fma(cc, app, fma(ss, aqq, -csg));
                    dg[q] = fma(ss, app, fma(cc, aqq, csg));
                    ur[p][q] = 0; ui[p][q] = 0;
                }
                // off-diagonal pairs (A[k][p], A[k][q]), k not in {p,q}:
                //   x' = c x - conj(sigma) y,   y' = sigma x + c y
#pragma unroll U
                for (int k = 0; k < N; k++) {
                    if (k == p || k == q) continue;
                    // A[k][p] lives at (k,p) if k < p, else as the conjugate of (p,k); same for q
                    T xr, xi, yr, yi;
                    if (k < p) { xr = ur[k][p]; xi = ui[k][p]; } else { xr = ur[p][k]; xi = -ui[p][k]; }
                    if (k < q) { yr = ur[k][q]; yi = ui[k][q]; } else { yr = ur[q][k]; yi = -ui[q][k]; }
                    const T nxr = c * xr - (spr * yr + spi * yi);
                    const T nxi = c * xi - (spr * yi - spi * yr);
                    const T nyr = c * yr + (spr * xr - spi * xi);
                    const T nyi = c * yi + (spr * xi + spi * xr);
                    if (k < p) { ur[k][p] = nxr; ui[k][p] = nxi; } else { ur[p][k] = nxr; ui[p][k] = -nxi; }
                    if (k < q) { ur[k][q] = nyr; ui[k][q] = nyi; } else { ur[q][k] = nyr; ui[q][k] = -nyi; }
                }
                // V <- V J
#pragma unroll U
                for (int k = 0; k < N; k++) {
                    const T kpr = vr[k][p], kpi = vi[k][p], kqr = vr[k][q], kqi = vi[k][q];
                    vr[k][p] = c * kpr - (spr * kqr + spi * kqi);
                    vi[k][p] = c * kpi - (spr * kqi - spi * kqr);
                    vr[k][q] = c * kqr + (spr * kpr - spi * kpi);
                    vi[k][q] = c * kqi + (spr * kpi + spi * kpr);
                }
            }
        }
    }
}

template <int N, typename T>
__global__ __launch_bounds__(64) void music_evd_kernel(const float2 *__restrict__ R, float *__restrict__ coef,
                                                       double *__restrict__ coef_d, float2 *__restrict__ pn_out,
                                                       int n_items, int M)
{
    constexpr bool UNROLL = (N <= 4);
    constexpr int U = UNROLL ? N : 1;
    const int item = blockIdx.x * blockDim.x + threadIdx.x;
    if (item >= n_items) return;
    const float2 *Ri = R + (size_t)item * (N * N);

    T dg[N], ar[N][N], ai[N][N], vr[N][N], vi[N][N];
    // only the upper triangle of the item is significant (cheevd uplo='U'); element (r,c) at r + c*N
#pragma unroll U
    for (int c = 0; c < N; c++) {
#pragma unroll U
        for (int r = 0; r < N; r++) {
            if (r > c) continue;
            const float2 x = Ri[r + c * N];
            if (r == c) dg[r] = (T)x.x;
            else { ar[r][c] = (T)x.x; ai[r][c] = (T)x.y; }
        }
    }
#pragma unroll U
    for (int r = 0; r < N; r++)
#pragma unroll U
        for (int c = 0; c < N; c++) { vr[r][c] = (r == c) ? (T)1 : (T)0; vi[r][c] = 0; }

    herm_jacobi<N, T, UNROLL>(dg, ar, ai, vr, vi);

    // ascending rank of each eigenvalue (eig_sym contract); noise set = ranks < N-M
    T sel[N];
#pragma unroll U
    for (int i = 0; i < N; i++) {
        int rank = 0;
#pragma unroll U
        for (int j = 0; j < N; j++) {
            const bool before = (dg[j] < dg[i]) || (dg[j] == dg[i] && j < i);
            rank += before ? 1 : 0;
        }
        sel[i] = (rank < N - M) ? (T)1 : (T)0;
    }
    // P_N[a][b] = sum_i sel_i v[a][i] conj(v[b][i]), Hermitian: upper triangle + diagonal only
    // (reusing dg / ar / ai for P_N)
    T svr[N][N], svi[N][N];
#pragma unroll U
    for (int a = 0; a < N; a++)
#pragma unroll U
        for (int i = 0; i < N; i++) { svr[a][i] = sel[i] * vr[a][i]; svi[a][i] = sel[i] * vi[a][i]; }
#pragma unroll U
    for (int a = 0; a < N; a++) {
#pragma unroll U
        for (int b = 0; b < N; b++) {
            if (b < a) continue;
            T pr = 0, pi = 0;
#pragma unroll U
            for (int i = 0; i < N; i++) {
                pr = fma(svr[a][i], vr[b][i], fma(svi[a][i], vi[b][i], pr));
                pi = fma(svi[a][i], vr[b][i], fma(-svr[a][i], vi[b][i], pi));
            }
            if (a == b) dg[a] = pr;
            else { ar[a][b] = pr; ai[a][b] = pi; }
        }
    }
    if (pn_out) {
        float2 *po = pn_out + (size_t)item * (N * N);
#pragma unroll U
        for (int c = 0; c < N; c++)
#pragma unroll U
            for (int r = 0; r < N; r++) {
                float2 e;
                if (r == c) e = make_float2((float)dg[r], 0.f);
                else if (r < c) e = make_float2((float)ar[r][c], (float)ai[r][c]);
                else e = make_float2((float)ar[c][r], -(float)ai[c][r]);
                po[r + c * N] = e;
            }
    }
    // diagonal sums u_l = sum_r P_N[r+l][r] = conj(sum_r P_N[r][r+l]); float record for the scan, double
    // record for the root finder (Root-MUSIC's near-double roots amplify a float rounding of u_l by
    // ~1e3-1e4) and for the double scan
    float *co = coef ? coef + (size_t)item * (2 * N) : nullptr;
    double *cd = coef_d ? coef_d + (size_t)item * (2 * N) : nullptr;
#pragma unroll U
    for (int l = 0; l < N; l++) {
        T ur_ = 0, ui_ = 0;
#pragma unroll U
        for (int r = 0; r < N; r++)
            if (r + l < N) {
                if (l == 0) ur_ += dg[r];
                else { ur_ += ar[r][r + l]; ui_ -= ai[r][r + l]; }
            }
        if (l == 0) {
            if (co) co[0] = (float)ur_;
            if (cd) cd[0] = (double)ur_;
        } else {
            if (co) { co[2 * l - 1] = (float)ur_; co[2 * l] = (float)ui_; }
            if (cd) { cd[2 * l - 1] = (double)ur_; cd[2 * l] = (double)ui_; }
        }
    }
    if (co) co[2 * N - 1] = 0.f;
    if (cd) cd[2 * N - 1] = 0.0;
}

// ---------------------------------------------------------------------------------------------
// N > 4: the same Jacobi with the matrices in LDS instead of registers.  One lane per item still
// (no barriers: a lane only ever touches its own column of the [element][lane] LDS image, which is
// also bank-conflict free), but A's packed upper triangle, V and the result scratch live in LDS
// where run-time row/column indices cost nothing; 3N^2+3N values per item bound the lanes per
// block (double: 64 lanes at N = 8, 25 at N = 16, one block per CU).
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(64) void music_evd_lds_kernel(const float2 *__restrict__ R, float *__restrict__ coef,
                                                           double *__restrict__ coef_d, float2 *__restrict__ pn_out,
                                                           int n_items, int N, int M, int LB)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T *sm = reinterpret_cast<T *>(smem_raw);
    const int lane = threadIdx.x;
    const int item = blockIdx.x * LB + lane;
    if (lane >= LB || item >= n_items) return;
    const int NT = N * (N - 1) / 2;
    // LDS image, element-major: dg[N] | ur[NT] | ui[NT] | vr[N*N] | vi[N*N] | cu_r[N] | cu_i[N]
    auto at = [&](int e) -> T & { return sm[(size_t)e * LB + lane]; };
    const int oDG = 0, oUR = N, oUI = N + NT, oVR = N + 2 * NT, oVI = oVR + N * N, oCR = oVI + N * N, oCI = oCR + N;
    auto tri = [&](int r, int c) { return r * N - r * (r + 1) / 2 + (c - r - 1); };   // r < c

    const float2 *Ri = R + (size_t)item * (N * N);
    for (int c = 0; c < N; c++)
        for (int r = 0; r <= c; r++) {
            const float2 x = Ri[r + c * N];
            if (r == c) at(oDG + r) = (T)x.x;
            else { at(oUR + tri(r, c)) = (T)x.x; at(oUI + tri(r, c)) = (T)x.y; }
        }
    for (int r = 0; r < N; r++)
        for (int c = 0; c < N; c++) { at(oVR + r * N + c) = (r == c) ? (T)1 : (T)0; at(oVI + r * N + c) = 0; }

    const int max_sweeps = Real<T>::max_sweeps + 2 * N;
    for (int sweep = 0; sweep < max_sweeps; sweep++) {
        T off = 0, dn = 0;
        for (int p = 0; p < N; p++) dn = fma(at(oDG + p), at(oDG + p), dn);
        for (int e = 0; e < NT; e++) off += at(oUR + e) * at(oUR + e) + at(oUI + e) * at(oUI + e);
        if (!(off > Real<T>::tol * dn) || !(off > Real<T>::tiny)) break;
        for (int p = 0; p < N - 1; p++) {
            for (int q = p + 1; q < N; q++) {
                const int epq = tri(p, q);
                const T apr = at(oUR + epq), api = at(oUI + epq);
                const T g2 = apr * apr + api * api;
                if (!(g2 > Real<T>::tiny)) continue;
                const T inv_g = Real<T>::rsqrt(g2);
                const T g = g2 * inv_g;
                const T phr = apr * inv_g, phi = api * inv_g;
                const T app = at(oDG + p), aqq = at(oDG + q);
                T tau = (aqq - app) * (T)0.5 * inv_g;
                tau = fmin(fmax(tau, -Real<T>::tau_max), Real<T>::tau_max);
                const T x1 = fma(tau, tau, (T)1);
                const T r = x1 * Real<T>::rsqrt(x1);
                const T h = fabs(tau) + r;
                const T w = Real<T>::rsqrt(fma(h, h, (T)1));
                const T c = h * w;
                const T s = copysign(w, tau);
                const T spr = s * phr, spi = s * phi;
                const T cc = c * c, ss = s * s, csg = (T)2 * c * s * g;
                at(oDG + p) = fma(cc, app, fma(ss, aqq, -csg));
                at(oDG + q) = fma(ss, app, fma(cc, aqq, csg));
                at(oUR + epq) = 0; at(oUI + epq) = 0;
                for (int k = 0; k < N; k++) {
                    if (k == p || k == q) continue;
                    const int ekp = (k < p) ? tri(k, p) : tri(p, k);
                    const int ekq = (k < q) ? tri(k, q) : tri(q, k);
                    const T sgp = (k < p) ? (T)1 : (T)-1, sgq = (k < q) ? (T)1 : (T)-1;   // conj when stored transposed
                    const T xr = at(oUR + ekp), xi = sgp * at(oUI + ekp);
                    const T yr = at(oUR + ekq), yi = sgq * at(oUI + ekq);
                    const T nxr = c * xr - (spr * yr + spi * yi);
                    const T nxi = c * xi - (spr * yi - spi * yr);
                    const T nyr = c * yr + (spr * xr - spi * xi);
                    const T nyi = c * yi + (spr * xi + spi * xr);
                    at(oUR + ekp) = nxr; at(oUI + ekp) = sgp * nxi;
                    at(oUR + ekq) = nyr; at(oUI + ekq) = sgq * nyi;
                }
                for (int k = 0; k < N; k++) {
                    const T kpr = at(oVR + k * N + p), kpi = at(oVI + k * N + p);
                    const T kqr = at(oVR + k * N + q), kqi = at(oVI + k * N + q);
                    at(oVR + k * N + p) = c * kpr - (spr * kqr + spi * kqi);
                    at(oVI + k * N + p) = c * kpi - (spr * kqi - spi * kqr);
                    at(oVR + k * N + q) = c * kqr + (spr * kpr - spi * kpi);
                    at(oVI + k * N + q) = c * kqi + (spr * kpi + spi * kpr);
                }
            }
        }
    }
    // noise set: ascending rank < N-M (bit i of sel)
    unsigned sel = 0;
    for (int i = 0; i < N; i++) {
        int rank = 0;
        const T wi = at(oDG + i);
        for (int j = 0; j < N; j++) {
            const T wj = at(oDG + j);
            rank += ((wj < wi) || (wj == wi && j < i)) ? 1 : 0;
        }
        if (rank < N - M) sel |= 1u << i;
    }
    for (int l = 0; l < N; l++) { at(oCR + l) = 0; at(oCI + l) = 0; }
    float2 *po = pn_out ? pn_out + (size_t)item * (N * N) : nullptr;
    for (int a = 0; a < N; a++)
        for (int b = a; b < N; b++) {
            T pr = 0, pi = 0;
            for (int i = 0; i < N; i++)
                if ((sel >> i) & 1u) {
                    const T ar_ = at(oVR + a * N + i), ai_ = at(oVI + a * N + i);
                    const T br_ = at(oVR + b * N + i), bi_ = at(oVI + b * N + i);
                    pr = fma(ar_, br_, fma(ai_, bi_, pr));
                    pi = fma(ai_, br_, fma(-ar_, bi_, pi));
                }
            // u_l = sum_r P[r+l][r] = conj(sum_r P[r][r+l])
            at(oCR + (b - a)) += pr;
            at(oCI + (b - a)) -= pi;
            if (po) {
                po[a + b * N] = make_float2((float)pr, (float)pi);
                if (a != b) po[b + a * N] = make_float2((float)pr, -(float)pi);
            }
        }
    float *co = coef ? coef + (size_t)item * (2 * N) : nullptr;
    double *cd = coef_d ? coef_d + (size_t)item * (2 * N) : nullptr;
    for (int l = 0; l < N; l++) {
        const T ur_ = at(oCR + l), ui_ = at(oCI + l);
        if (l == 0) {
            if (co) co[0] = (float)ur_;
            if (cd) cd[0] = (double)ur_;
        } else {
            if (co) { co[2 * l - 1] = (float)ur_; co[2 * l] = (float)ui_; }
            if (cd) { cd[2 * l - 1] = (double)ur_; cd[2 * l] = (double)ui_; }
        }
    }
    if (co) co[2 * N - 1] = 0.f;
    if (cd) cd[2 * N - 1] = 0.0;
}

template <typename T>
static int launch_evd_lds(int N, int M, int n_items, const void *d_R, void *d_coef, void *d_coef_d, void *d_pn,
                          hipStream_t st)
{
    const size_t per_item = (size_t)(3 * N * N + 3 * N) * sizeof(T);   // generous bound on the LDS image
    const size_t budget = 156 * 1024;
    int LB = (int)(budget / per_item);
    if (LB > 64) LB = 64;
    if (LB < 1) { set_error("MUSIC: LDS image of one %dx%d item does not fit", N, N); return DOA_ERR_UNSUPPORTED; }
    const size_t bytes = per_item * LB;
    static size_t configured_f = 0, configured_d = 0;
    size_t &configured = (sizeof(T) == 4) ? configured_f : configured_d;
    if (bytes > configured) {
        DOA_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&music_evd_lds_kernel<T>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024)));
        configured = 160 * 1024;
    }
    dim3 block(64), grid((n_items + LB - 1) / LB);
    hipLaunchKernelGGL((music_evd_lds_kernel<T>), grid, block, bytes, st, (const float2 *)d_R, (float *)d_coef,
                       (double *)d_coef_d, (float2 *)d_pn, n_items, N, M, LB);
    return DOA_OK;
}

template <int N> static void launch_evd_n(int M, int n_items, const void *d_R, void *d_coef, void *d_coef_d, void *d_pn,
                                          int bits, hipStream_t st)
{
    dim3 block(64), grid((n_items + 63) / 64);
    if (bits == 32)
        hipLaunchKernelGGL((music_evd_kernel<N, float>), grid, block, 0, st, (const float2 *)d_R, (float *)d_coef,
                           (double *)d_coef_d, (float2 *)d_pn, n_items, M);
    else
        hipLaunchKernelGGL((music_evd_kernel<N, double>), grid, block, 0, st, (const float2 *)d_R, (float *)d_coef,
                           (double *)d_coef_d, (float2 *)d_pn, n_items, M);
}

int launch_music_evd(int N, int M, int n_items, const void *d_R, void *d_coef, void *d_coef_d, void *d_pn,
                     int evd_bits, hipStream_t st)
{
    if (n_items <= 0) return DOA_OK;
    if (N < 2 || N > DOA_MAX_ANT_ELE) {
        set_error("MUSIC: num_ant_ele=%d outside the built range 2..%d", N, DOA_MAX_ANT_ELE);
        return DOA_ERR_UNSUPPORTED;
    }
    switch (N) {
    case 2: launch_evd_n<2>(M, n_items, d_R, d_coef, d_coef_d, d_pn, evd_bits, st); break;
    case 3: launch_evd_n<3>(M, n_items, d_R, d_coef, d_coef_d, d_pn, evd_bits, st); break;
    case 4: launch_evd_n<4>(M, n_items, d_R, d_coef, d_coef_d, d_pn, evd_bits, st); break;
    default: {
        const int rc = (evd_bits == 32) ? launch_evd_lds<float>(N, M, n_items, d_R, d_coef, d_coef_d, d_pn, st)
                                        : launch_evd_lds<double>(N, M, n_items, d_R, d_coef, d_coef_d, d_pn, st);
        if (rc != DOA_OK) return rc;
    }
    }
    DOA_HIP_TRY(hipGetLastError());
    return DOA_OK;
}

// ---------------------------------------------------------------------------------------------
// K4: spectrum scan
// ---------------------------------------------------------------------------------------------
// Q = u0 + 2 Re( sum_{l=1}^{N-1} u_l z^l ) by Horner, in T (float or double).  c = the item's
// coefficient record [u0, Re u1, Im u1, ...].
template <int N, typename T> __device__ __forceinline__ T null_spectrum(const T (&c)[2 * N], T zr, T zi)
{
    if constexpr (N == 1) return c[0];
    T hr = c[2 * (N - 1) - 1], hi = c[2 * (N - 1)];
#pragma unroll
    for (int l = N - 2; l >= 1; l--) {
        const T tr = fma(hr, zr, fma(-hi, zi, c[2 * l - 1]));
        const T ti = fma(hr, zi, fma(hi, zr, c[2 * l]));
        hr = tr; hi = ti;
    }
    const T re = fma(hr, zr, -hi * zi);
    return fma((T)2, re, c[0]);
}

__device__ __forceinline__ float db_from_ratio(float out, float mx, float inv_mx)
{
    // 10*log10(out/max).  The maximum itself must come out as exactly 0 dB (x/x == 1 in the
    // reference; inf/inf stays NaN as there), everything else is out*(1/max) through the hardware
    // log2: 10*log10(r) = (10*log10(2)) * log2(r).
    // out <= max, so out/max <= 1 in the reference; the reciprocal multiply is clamped to keep that
    float ratio = (out == mx && mx != INFINITY) ? 1.0f : fminf(out * inv_mx, 1.0f);
    return 3.0102999566398120f * __log2f(ratio);
}

// Fast path: P % 4 == 0 and P <= 256*CH.  One wave per item, grid-stride over items so that the
// z table (4*CH angles per lane) is loaded once per wave and stays in registers.  With PEAK the
// find_local_max step (K5) runs on the dB values while they are still in registers, so the spectrum
// is written once and never read back.
template <int N, int CH, typename T, bool HAS_Q, bool PEAK>
__global__ __launch_bounds__(256) void music_scan_kernel(const T *__restrict__ coef, const T *__restrict__ ztab,
                                                         float *__restrict__ spec, float *__restrict__ qout, int P,
                                                         int n_items, const float *__restrict__ xaxis,
                                                         float *__restrict__ pk_val, float *__restrict__ pk_loc, int M)
{
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(blockIdx.x * (blockDim.x / kWave) + (threadIdx.x / kWave));
    const int n_waves = gridDim.x * (blockDim.x / kWave);

    T zr[CH][4], zi[CH][4];
#pragma unroll
    for (int j = 0; j < CH; j++) {
        const int i0 = 4 * lane + 256 * j;
#pragma unroll
        for (int e = 0; e < 4; e++) {
            if (i0 < P) { zr[j][e] = ztab[2 * (i0 + e)]; zi[j][e] = ztab[2 * (i0 + e) + 1]; }
            else { zr[j][e] = 1; zi[j][e] = 0; }
        }
    }
    // coefficient records arrive through scalar loads (wave-uniform address); the next item's record
    // is requested before this item's arithmetic so its latency hides behind it
    T c[2 * N], c_next[2 * N];
    if (wave < n_items) {
#pragma unroll
        for (int k = 0; k < 2 * N; k++) c_next[k] = coef[(size_t)wave * (2 * N) + k];
    }
    for (int item = wave; item < n_items; item += n_waves) {
#pragma unroll
        for (int k = 0; k < 2 * N; k++) c[k] = c_next[k];
        const int nxt = item + n_waves;
        if (nxt < n_items) {
#pragma unroll
            for (int k = 0; k < 2 * N; k++) c_next[k] = coef[(size_t)nxt * (2 * N) + k];
        }
        float out[CH][4];
        float mx = -INFINITY;
#pragma unroll
        for (int j = 0; j < CH; j++) {
            const bool live = (4 * lane + 256 * j) < P;
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const float q = (float)null_spectrum<N, T>(c, zr[j][e], zi[j][e]);
                if constexpr (HAS_Q) {
                    if (live) qout[(size_t)item * P + 4 * lane + 256 * j + e] = q;
                }
                out[j][e] = __builtin_amdgcn_rcpf(q);           // 1.0/Q  (:140)
                mx = live ? fmaxf(mx, out[j][e]) : mx;
            }
        }
        mx = wave_allreduce_max(mx);
        const float inv_mx = __builtin_amdgcn_rcpf(mx);
        float *row = spec + (size_t)item * P;
#pragma unroll
        for (int j = 0; j < CH; j++) {
            const int i0 = 4 * lane + 256 * j;
#pragma unroll
            for (int e = 0; e < 4; e++) out[j][e] = db_from_ratio(out[j][e], mx, inv_mx);   // now dB
            if (i0 < P) *reinterpret_cast<float4 *>(row + i0) = make_float4(out[j][0], out[j][1], out[j][2], out[j][3]);
        }
        if constexpr (PEAK) peak_pick<CH>(out, lane, P, M, xaxis, pk_val + (size_t)item * M, pk_loc + (size_t)item * M);
    }
}

// Any P: one wave per item, one angle per lane per step, two passes (max, then write).
template <int N, typename T>
__global__ __launch_bounds__(256) void music_scan_generic_kernel(const T *__restrict__ coef, const T *__restrict__ ztab,
                                                                 float *__restrict__ spec, float *__restrict__ qout, int P,
                                                                 int n_items)
{
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(blockIdx.x * (blockDim.x / kWave) + (threadIdx.x / kWave));
    const int n_waves = gridDim.x * (blockDim.x / kWave);
    for (int item = wave; item < n_items; item += n_waves) {
        const T *co = coef + (size_t)item * (2 * N);
        T c[2 * N];
#pragma unroll
        for (int k = 0; k < 2 * N; k++) c[k] = co[k];
        float mx = -INFINITY;
        for (int i = lane; i < P; i += kWave) {
            const float q = (float)null_spectrum<N, T>(c, ztab[2 * i], ztab[2 * i + 1]);
            if (qout) qout[(size_t)item * P + i] = q;
            mx = fmaxf(mx, __builtin_amdgcn_rcpf(q));
        }
        mx = wave_allreduce_max(mx);
        const float inv_mx = __builtin_amdgcn_rcpf(mx);
        for (int i = lane; i < P; i += kWave) {
            const float o = __builtin_amdgcn_rcpf((float)null_spectrum<N, T>(c, ztab[2 * i], ztab[2 * i + 1]));
            spec[(size_t)item * P + i] = db_from_ratio(o, mx, inv_mx);
        }
    }
}

struct ScanPeakArgs {           // optional fused K5
    const float *xaxis = nullptr;
    float *val = nullptr, *loc = nullptr;
    int M = 0;
};

template <int N, int CH, typename T>
static void launch_scan_fast(dim3 grid, dim3 block, hipStream_t st, const T *co, const T *z, float *sp, float *q, int P,
                             int n_items, const ScanPeakArgs &pk)
{
    if (q)
        hipLaunchKernelGGL((music_scan_kernel<N, CH, T, true, false>), grid, block, 0, st, co, z, sp, q, P, n_items,
                           nullptr, nullptr, nullptr, 0);
    else if (pk.val)
        hipLaunchKernelGGL((music_scan_kernel<N, CH, T, false, true>), grid, block, 0, st, co, z, sp, q, P, n_items,
                           pk.xaxis, pk.val, pk.loc, pk.M);
    else
        hipLaunchKernelGGL((music_scan_kernel<N, CH, T, false, false>), grid, block, 0, st, co, z, sp, q, P, n_items,
                           nullptr, nullptr, nullptr, 0);
}

// returns true when the fused peak pick ran (fast path only)
template <int N, typename T>
static bool launch_scan_nt(const T *co, const T *z, int P, int n_items, void *d_spec, void *d_q, const ScanPeakArgs &pk,
                           hipStream_t st)
{
    float *sp = (float *)d_spec, *q = (float *)d_q;
    const int waves_per_block = 4;
    const bool aligned = (P % 4 == 0) && (reinterpret_cast<uintptr_t>(d_spec) % 16 == 0);
    // enough waves to fill the chip several times over, few enough that each wave amortises its
    // z-table load over several items
    int blocks = (n_items + waves_per_block - 1) / waves_per_block;
    static const int wpc = [] { const char *e = getenv("DOA_SCAN_WAVES_PER_CU"); return e ? atoi(e) : 8; }();
    const int max_blocks = 256 * wpc / waves_per_block;   // default 8 waves per CU: 2 items per wave at batch 4096
    if (blocks > max_blocks) blocks = max_blocks;
    dim3 grid(blocks), block(waves_per_block * kWave);
    const int max_ch = (sizeof(T) == 4) ? 16 : 4;        // double z table: 4 chunks fit the register file
    if (aligned && P <= 256 * max_ch) {
        if (P <= 256) launch_scan_fast<N, 1, T>(grid, block, st, co, z, sp, q, P, n_items, pk);
        else if (P <= 512) launch_scan_fast<N, 2, T>(grid, block, st, co, z, sp, q, P, n_items, pk);
        else if (P <= 1024) launch_scan_fast<N, 4, T>(grid, block, st, co, z, sp, q, P, n_items, pk);
        else if constexpr (sizeof(T) == 4) {
            if (P <= 2048) launch_scan_fast<N, 8, T>(grid, block, st, co, z, sp, q, P, n_items, pk);
            else launch_scan_fast<N, 16, T>(grid, block, st, co, z, sp, q, P, n_items, pk);
        }
        return pk.val != nullptr && q == nullptr;
    }
    hipLaunchKernelGGL((music_scan_generic_kernel<N, T>), grid, block, 0, st, co, z, sp, q, P, n_items);
    return false;
}

template <int N> static bool launch_scan_n(const MusicTables &t, int bits, int n_items, const void *d_coef, void *d_spec,
                                           void *d_q, const ScanPeakArgs &pk, hipStream_t st)
{
    if (bits == 32)
        return launch_scan_nt<N, float>((const float *)d_coef, t.d_z.as<float>(), t.P, n_items, d_spec, d_q, pk, st);
    return launch_scan_nt<N, double>((const double *)d_coef, t.d_zd.as<double>(), t.P, n_items, d_spec, d_q, pk, st);
}

int launch_music_scan(const MusicTables &t, int bits, int n_items, const void *d_coef, void *d_spec, void *d_q,
                      hipStream_t st, const PeakTables *peaks, void *d_max, void *d_argmax, bool *peaks_done)
{
    if (peaks_done) *peaks_done = false;
    if (n_items <= 0) return DOA_OK;
    ScanPeakArgs pk;
    if (peaks && d_max && d_argmax && peaks->L == t.P) {
        pk.xaxis = peaks->d_x.as<float>(); pk.val = (float *)d_max; pk.loc = (float *)d_argmax; pk.M = peaks->M;
    }
    bool done = false;
    switch (t.N) {
#define DOA_SCAN_CASE(n) case n: done = launch_scan_n<n>(t, bits, n_items, d_coef, d_spec, d_q, pk, st); break;
        DOA_SCAN_CASE(2) DOA_SCAN_CASE(3) DOA_SCAN_CASE(4) DOA_SCAN_CASE(5) DOA_SCAN_CASE(6) DOA_SCAN_CASE(7)
        DOA_SCAN_CASE(8) DOA_SCAN_CASE(9) DOA_SCAN_CASE(10) DOA_SCAN_CASE(11) DOA_SCAN_CASE(12) DOA_SCAN_CASE(13)
        DOA_SCAN_CASE(14) DOA_SCAN_CASE(15) DOA_SCAN_CASE(16)
#undef DOA_SCAN_CASE
    default:
        set_error("MUSIC: num_ant_ele=%d outside the built range 2..%d", t.N, DOA_MAX_ANT_ELE);
        return DOA_ERR_UNSUPPORTED;
    }
    if (peaks_done) *peaks_done = done;
    DOA_HIP_TRY(hipGetLastError());
    return DOA_OK;
}

}  // namespace doa

// ---------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------
struct doa_MUSIC_lin_array {
    doa::MusicTables tab;
    int bits = 64;   // internal precision of EVD + scan (doa_set_internal_precision)
    int device = 0;
    long long items_total = 0;
    hipStream_t stream = nullptr;
    doa::DevBuf d_in, d_out, d_coef, d_pn, d_q;
};

static int music_validate(const char *who, float norm_spacing, int num_targets, int num_ant_ele)
{
    // grc/doa_MUSIC_lin_array.xml:33-35 (inputs > 0, inputs > num_targets, norm_spacing <= 0.5); the
    // reference ctor does not validate (num_targets >= N would index cols(0,-1)), so create fails here.
    if (num_ant_ele <= 0 || num_targets <= 0 || num_targets >= num_ant_ele) {
        doa::set_error("%s: need 0 < num_targets < num_ant_ele (got %d, %d)", who, num_targets, num_ant_ele);
        return DOA_ERR_INVALID_ARG;
    }
    if (!(norm_spacing > 0.0f) || norm_spacing > 0.5f) {
        doa::set_error("%s: need 0 < norm_spacing <= 0.5 (got %g)", who, (double)norm_spacing);
        return DOA_ERR_INVALID_ARG;
    }
    if (num_ant_ele > DOA_MAX_ANT_ELE) {
        doa::set_error("%s: num_ant_ele=%d exceeds DOA_MAX_ANT_ELE=%d", who, num_ant_ele, DOA_MAX_ANT_ELE);
        return DOA_ERR_UNSUPPORTED;
    }
    return DOA_OK;
}

extern "C" {

doa_MUSIC_lin_array_t *doa_MUSIC_lin_array_create(float norm_spacing, int num_targets, int num_ant_ele,
                                                  int pspectrum_len)
{
    doa::clear_error();
    if (music_validate("MUSIC_lin_array", norm_spacing, num_targets, num_ant_ele) != DOA_OK) return nullptr;
    if (pspectrum_len <= 0) {
        doa::set_error("MUSIC_lin_array: pspectrum_len must be > 0 (got %d)", pspectrum_len);
        return nullptr;
    }
    int dev = 0;
    if (doa::ensure_device(&dev) != DOA_OK) return nullptr;
    auto *h = new (std::nothrow) doa_MUSIC_lin_array();
    if (!h) { doa::set_error("out of memory"); return nullptr; }
    h->device = dev;
    h->bits = doa::internal_precision_bits();
    if (h->tab.build(norm_spacing, num_targets, num_ant_ele, pspectrum_len) != DOA_OK ||
        hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) {
        if (!*doa_last_error()) doa::set_error("MUSIC_lin_array: device setup failed");
        doa_MUSIC_lin_array_destroy(h);
        return nullptr;
    }
    return h;
}

void doa_MUSIC_lin_array_destroy(doa_MUSIC_lin_array_t *h)
{
    if (!h) return;
    h->tab.release();
    h->d_in.release(); h->d_out.release(); h->d_coef.release(); h->d_pn.release(); h->d_q.release();
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

long long doa_MUSIC_lin_array_items_total(const doa_MUSIC_lin_array_t *h) { return h ? h->items_total : 0; }

int doa_MUSIC_lin_array_work_dev(doa_MUSIC_lin_array_t *h, int noutput_items, const void *d_input_items0,
                                 void *d_output_items0, void *hip_stream)
{
    doa::clear_error();
    if (!h || noutput_items < 0 || (noutput_items > 0 && (!d_input_items0 || !d_output_items0))) {
        doa::set_error("MUSIC_lin_array_work_dev: bad arguments");
        return DOA_ERR_INVALID_ARG;
    }
    if (noutput_items == 0) return 0;
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    const int N = h->tab.N;
    int rc = h->d_coef.reserve((size_t)noutput_items * doa::coef_stride(N) * sizeof(double));
    if (rc != DOA_OK) return rc;
    const bool dbl = (h->bits == 64);
    rc = doa::launch_music_evd(N, h->tab.M, noutput_items, d_input_items0, dbl ? nullptr : h->d_coef.p,
                               dbl ? h->d_coef.p : nullptr, nullptr, h->bits, st);
    if (rc != DOA_OK) return rc;
    rc = doa::launch_music_scan(h->tab, h->bits, noutput_items, h->d_coef.p, d_output_items0, nullptr, st);
    if (rc != DOA_OK) return rc;
    h->items_total += noutput_items;
    return noutput_items;
}

int doa_MUSIC_lin_array_work(doa_MUSIC_lin_array_t *h, int noutput_items, const void *input_items0,
                             void *output_items0)
{
    doa::clear_error();
    if (!h || noutput_items < 0 || (noutput_items > 0 && (!input_items0 || !output_items0))) {
        doa::set_error("MUSIC_lin_array_work: bad arguments");
        return DOA_ERR_INVALID_ARG;
    }
    if (noutput_items == 0) return 0;
    const int N = h->tab.N, P = h->tab.P;
    const size_t in_bytes = (size_t)noutput_items * N * N * sizeof(float2);
    const size_t out_bytes = (size_t)noutput_items * P * sizeof(float);
    int rc = h->d_in.reserve(in_bytes);
    if (rc == DOA_OK) rc = h->d_out.reserve(out_bytes);
    if (rc != DOA_OK) return rc;
    DOA_HIP_TRY(hipMemcpyAsync(h->d_in.p, input_items0, in_bytes, hipMemcpyHostToDevice, h->stream));
    rc = doa_MUSIC_lin_array_work_dev(h, noutput_items, h->d_in.p, h->d_out.p, h->stream);
    if (rc < 0) return rc;
    DOA_HIP_TRY(hipMemcpyAsync(output_items0, h->d_out.p, out_bytes, hipMemcpyDeviceToHost, h->stream));
    DOA_HIP_TRY(hipStreamSynchronize(h->stream));
    return noutput_items;
}

int doa_MUSIC_lin_array_debug(doa_MUSIC_lin_array_t *h, int noutput_items, const void *input_items0,
                              void *projector_out, void *null_spectrum_out)
{
    doa::clear_error();
    if (!h || noutput_items <= 0 || !input_items0) {
        doa::set_error("MUSIC_lin_array_debug: bad arguments");
        return DOA_ERR_INVALID_ARG;
    }
    const int N = h->tab.N, P = h->tab.P;
    const size_t in_bytes = (size_t)noutput_items * N * N * sizeof(float2);
    const size_t sp_bytes = (size_t)noutput_items * P * sizeof(float);
    int rc = h->d_in.reserve(in_bytes);
    if (rc == DOA_OK) rc = h->d_out.reserve(sp_bytes);
    if (rc == DOA_OK) rc = h->d_q.reserve(sp_bytes);
    if (rc == DOA_OK) rc = h->d_pn.reserve(in_bytes);
    if (rc == DOA_OK) rc = h->d_coef.reserve((size_t)noutput_items * doa::coef_stride(N) * sizeof(double));
    if (rc != DOA_OK) return rc;
    DOA_HIP_TRY(hipMemcpyAsync(h->d_in.p, input_items0, in_bytes, hipMemcpyHostToDevice, h->stream));
    const bool dbl = (h->bits == 64);
    rc = doa::launch_music_evd(N, h->tab.M, noutput_items, h->d_in.p, dbl ? nullptr : h->d_coef.p,
                               dbl ? h->d_coef.p : nullptr, h->d_pn.p, h->bits, h->stream);
    if (rc != DOA_OK) return rc;
    rc = doa::launch_music_scan(h->tab, h->bits, noutput_items, h->d_coef.p, h->d_out.p, h->d_q.p, h->stream);
    if (rc != DOA_OK) return rc;
    if (projector_out)
        DOA_HIP_TRY(hipMemcpyAsync(projector_out, h->d_pn.p, in_bytes, hipMemcpyDeviceToHost, h->stream));
    if (null_spectrum_out)
        DOA_HIP_TRY(hipMemcpyAsync(null_spectrum_out, h->d_q.p, sp_bytes, hipMemcpyDeviceToHost, h->stream));
    DOA_HIP_TRY(hipStreamSynchronize(h->stream));
    return noutput_items;
}

}  // extern "C"
