// music.hip — K2+K3 (batched Hermitian EVD -> noise projector -> diagonal sums) and K4 (pseudo-
// spectrum scan) of MUSIC_lin_array on gfx950.
//
// Replaces gr::doa::MUSIC_lin_array (reference lib/MUSIC_lin_array_impl.cc):
//   ctor  :47-87,98-104  element positions, float-accumulated theta grid, steering vectors
//   work  :121-144       eig_sym (LAPACK cheevd 'V','U', ascending) -> U_N = first N-M vectors
//                        -> P_N = U_N U_N^H -> out[i] = 1/Re(a_i^H P_N a_i) -> 10 log10(out/max)
//
// How it is laid out here:
//   * EVD: cyclic complex Jacobi, in double by default (float selectable:
//     doa_set_internal_precision).  N <= 4 (music_evd_kernel): one lane per covariance matrix, diagonal +
//     strict upper triangle and V in registers, fully unrolled.  N > 4 (music_evd_group_kernel): 8 or 16
//     lanes per matrix, one row of A and V per lane, round-robin parallel ordering.  Rotations are built from
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
#include "music_scan.hpp"
#include "peak_device.hpp"

#include <climits>
#include <cmath>
#include <cstdlib>
#include <type_traits>
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
// Group-parallel Jacobi: G lanes (G = 4, 8 or 16 >= N) share one item.  Lane r of the group holds
// row r of A and row r of V in registers; a sweep is the G-1 rounds of a round-robin tournament,
// each round rotating G/2 disjoint pivot pairs at once:
//   * every lane builds the rotation of its own pair from (A[r][r], A[q][q], A[r][q]), q = its
//     partner this round (partner diagonal: one cross-lane fetch);
//   * A <- A J and V <- V J are column operations, i.e. lane-local, with compile-time column pairs
//     (rounds are unrolled) and the pair parameters broadcast from the owning lane;
//   * A <- J^H A mixes row r with the partner's row: one cross-lane fetch of that row.
// 4 (G = 16) to 16 (G = 4) items per wave instead of 64, but the dependent instruction chain per item
// shrinks by ~G/2 and there is no run-time register indexing, LDS image or scratch.
// Indices >= N are padding: zero off-diagonals and a huge diagonal, so they never rotate and rank
// last.
// ---------------------------------------------------------------------------------------------
template <int G> struct Tournament {
    // pair j of round t of the circle method on G players: (G-1, t) and ((t+k) % (G-1), (t-k) % (G-1))
    static constexpr int a(int t, int j) { return j == 0 ? G - 1 : (t + j) % (G - 1); }
    static constexpr int b(int t, int j) { return j == 0 ? t : (t - j + (G - 1)) % (G - 1); }
    static constexpr int p(int t, int j) { return a(t, j) < b(t, j) ? a(t, j) : b(t, j); }
    static constexpr int q(int t, int j) { return a(t, j) < b(t, j) ? b(t, j) : a(t, j); }
};

template <typename T> __device__ __forceinline__ T lane_fetch(T v, int src_lane);
template <> __device__ __forceinline__ float lane_fetch<float>(float v, int src_lane) { return __shfl(v, src_lane, kWave); }
template <> __device__ __forceinline__ double lane_fetch<double>(double v, int src_lane) { return __shfl(v, src_lane, kWave); }

// sum over the G lanes of a group (G = 4, 8, 16; groups are aligned inside a DPP row of 16)
template <int G, typename T> __device__ __forceinline__ T group_sum(T v, int lane)
{
#pragma unroll
    for (int m = 1; m < G; m <<= 1) v += lane_fetch<T>(v, lane ^ m);
    return v;
}

// G == 4: a group is a DPP quad, so every cross-lane fetch of the round is a quad_perm with a compile-time pattern
// (a few cycles on the vector pipe instead of a ~100-cycle ds_bpermute round trip -- this kernel is a pure latency chain)
template <int PATTERN> __device__ __forceinline__ float quad_fetch(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), PATTERN, 0xF, 0xF, false));
}
template <int PATTERN> __device__ __forceinline__ double quad_fetch(double v)
{
    const long long x = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp((int)(x & 0xFFFFFFFFll), (int)(x & 0xFFFFFFFFll), PATTERN, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp((int)(x >> 32), (int)(x >> 32), PATTERN, 0xF, 0xF, false);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
constexpr int quad_pattern(int s0, int s1, int s2, int s3) { return s0 | (s1 << 2) | (s2 << 4) | (s3 << 6); }
// partner of quad lane r in round t of the 4-player tournament: t=0: 0<->3, 1<->2; t=1: 0<->2, 1<->3; t=2: 0<->1, 2<->3
constexpr int quad_partner(int t, int r) { return (r == 3) ? t : ((r == t) ? 3 : (2 * t - r + 6) % 3); }
constexpr int quad_lower(int t, int r) { return r < quad_partner(t, r) ? r : quad_partner(t, r); }

template <int G, typename T, int ROUND>
__device__ __forceinline__ void jacobi_round(T (&ar)[G], T (&ai)[G], T (&vr)[G], T (&vi)[G], int r, int base, bool active)
{
    using TT = Tournament<G>;
    // partner of lane r in this round
    const int partner = (r == G - 1) ? ROUND : ((r == ROUND) ? G - 1 : (2 * ROUND - r + 2 * (G - 1)) % (G - 1));
    constexpr int RT = (G == 4) ? ROUND : 0;
    constexpr int PAT_PARTNER = quad_pattern(quad_partner(RT, 0), quad_partner(RT, 1), quad_partner(RT, 2), quad_partner(RT, 3));
    constexpr int PAT_LOWER = quad_pattern(quad_lower(RT, 0), quad_lower(RT, 1), quad_lower(RT, 2), quad_lower(RT, 3));
    auto from_partner = [&](T v) { if constexpr (G == 4) return quad_fetch<PAT_PARTNER>(v); else return lane_fetch<T>(v, base + partner); };
    auto from_lower = [&](T v, int lo_lane) { if constexpr (G == 4) return quad_fetch<PAT_LOWER>(v); else return lane_fetch<T>(v, base + lo_lane); };
    // own diagonal, partner diagonal, pivot element A[r][partner]
    T d_own = 0, xr = 0, xi = 0;
#pragma unroll
    for (int k = 0; k < G; k++) {
        d_own = (k == r) ? ar[k] : d_own;
        xr = (k == partner) ? ar[k] : xr;
        xi = (k == partner) ? ai[k] : xi;
    }
    const T d_oth = from_partner(d_own);
    const T g2 = xr * xr + xi * xi;
    const bool live = active && (g2 > Real<T>::tiny);
    const T inv_g = Real<T>::rsqrt(live ? g2 : (T)1);
    const T phr = xr * inv_g, phi = xi * inv_g;
    T tau = (d_oth - d_own) * (T)0.5 * inv_g;
    tau = fmin(fmax(tau, -Real<T>::tau_max), Real<T>::tau_max);
    const T x1 = fma(tau, tau, (T)1);
    const T rt = x1 * Real<T>::rsqrt(x1);
    const T h = fabs(tau) + rt;
    const T w = Real<T>::rsqrt(fma(h, h, (T)1));
    // Both lanes of a pair evaluate this, but only the lower lane's result is used (below): its
    // partner sees A[q][p], which equals conj(A[p][q]) only to rounding, and two almost-equal
    // rotations are not one unitary rotation once the pivot has shrunk to that level.
    const T c_mine = live ? h * w : (T)1;
    const T s_mine = live ? copysign(w, tau) : (T)0;
    const int lo = (r < partner) ? r : partner;
    const T c = from_lower(c_mine, lo);
    const T slr = from_lower(s_mine * phr, lo), sli = from_lower(s_mine * phi, lo);   // sigma = J[lo][hi]
    const T sgr = (r == lo) ? slr : -slr, sgi = (r == lo) ? sli : sli;    // J[r][partner]: sigma, or -conj(sigma)
    // column operations A <- A J, V <- V J: canonical (c, sigma) of each pair come from its lower lane
#pragma unroll
    for (int j = 0; j < G / 2; j++) {
        constexpr int dummy = 0; (void)dummy;
        const int P = TT::p(ROUND, j), Q = TT::q(ROUND, j);
        constexpr int P0 = TT::p(RT, 0) & 3, P1 = TT::p(RT, 1) & 3;       // G == 4: the two pairs' lower lanes
        constexpr int PB0 = quad_pattern(P0, P0, P0, P0), PB1 = quad_pattern(P1, P1, P1, P1);
        auto bcast = [&](T v) {
            if constexpr (G == 4) return (j == 0) ? quad_fetch<PB0>(v) : quad_fetch<PB1>(v);
            else return lane_fetch<T>(v, base + P);
        };
        const T cj = bcast(c);
        const T sr = bcast(sgr), si = bcast(sgi);
        {
            const T pr = ar[P], pi = ai[P], qr = ar[Q], qi = ai[Q];
            ar[P] = cj * pr - (sr * qr + si * qi);
            ai[P] = cj * pi - (sr * qi - si * qr);
            ar[Q] = cj * qr + (sr * pr - si * pi);
            ai[Q] = cj * qi + (sr * pi + si * pr);
        }
        {
            const T pr = vr[P], pi = vi[P], qr = vr[Q], qi = vi[Q];
            vr[P] = cj * pr - (sr * qr + si * qi);
            vi[P] = cj * pi - (sr * qi - si * qr);
            vr[Q] = cj * qr + (sr * pr - si * pi);
            vi[Q] = cj * qi + (sr * pi + si * pr);
        }
    }
    // row operation A <- J^H A: row_r' = c row_r - sigma_r row_partner (same form on both lanes of a pair)
#pragma unroll
    for (int k = 0; k < G; k++) {
        const T orr = from_partner(ar[k]), oi = from_partner(ai[k]);
        const T nr = c * ar[k] - (sgr * orr - sgi * oi);
        const T ni = c * ai[k] - (sgr * oi + sgi * orr);
        ar[k] = nr; ai[k] = ni;
    }
}

template <int G, typename T, int ROUND> struct RoundLoop {
    static __device__ __forceinline__ void run(T (&ar)[G], T (&ai)[G], T (&vr)[G], T (&vi)[G], int r, int base, bool active)
    {
        jacobi_round<G, T, ROUND>(ar, ai, vr, vi, r, base, active);
        if constexpr (ROUND + 1 < G - 1) RoundLoop<G, T, ROUND + 1>::run(ar, ai, vr, vi, r, base, active);
    }
};

// Shared epilogue of the group kernels: lane r of a G-lane group (lanes base .. base+G-1) holds row r
// of V (columns = eigenvector slots) and the eigenvalue lam of slot r; real_col = slot r belongs to
// the N x N problem (not padding).  Ranks the eigenvalues ascending (the eig_sym contract), takes the
// N-M smallest as the noise set and emits the diagonal sums u_l of P_N (or, in calibrate mode, the
// de-rotated top eigenvector).
template <int G, typename T>
__device__ __forceinline__ void evd_group_epilogue(T (&vr)[G], T (&vi)[G], T lam, bool real_col, int r, int base, int lane,
                                                   int item, bool real_item, int N, int M, float *__restrict__ coef,
                                                   double *__restrict__ coef_d, float2 *__restrict__ pn_out,
                                                   const float2 *__restrict__ pilot, float2 *__restrict__ cal_out)
{
    // eigenvalue of lane r = A[r][r]; ascending rank inside the group; noise set = ranks < N-M
    int rank = 0;
#pragma unroll
    for (int j = 0; j < G; j++) {
        const T lj = lane_fetch<T>(lam, base + j);
        rank += ((lj < lam) || (lj == lam && j < r)) ? 1 : 0;
    }
    if (cal_out) {
        // calibrate_lin_array (reference lib/calibrate_lin_array_impl.cc:98-134): U_S = eigenvector of the
        // largest eigenvalue; W = diag(conj v) U_S U_S^H diag(v) is rank one, so its unit-eigenvalue
        // eigenvector is conj(v) .* U_S (normalised).  Phase convention: element 0 real, non-negative.
        const unsigned long long top_mask = __ballot(real_col && (rank == N - 1));
        const int imax = __builtin_ctzll(((top_mask >> base) & ((1ull << G) - 1ull)) | (1ull << G));   // column of the top eigenvector
        T er = 0, ei = 0;
#pragma unroll
        for (int k = 0; k < G; k++) { er = (k == imax) ? vr[k] : er; ei = (k == imax) ? vi[k] : ei; }
        const float2 pv = (r < N) ? pilot[r] : make_float2(1.f, 0.f);
        // conj(v_r) * u_r
        T wr = (T)pv.x * er + (T)pv.y * ei, wi = (T)pv.x * ei - (T)pv.y * er;
        if (!(r < N)) { wr = 0; wi = 0; }
        const T nrm2 = group_sum<G, T>(wr * wr + wi * wi, lane);
        const T inv = Real<T>::rsqrt(nrm2 > (T)0 ? nrm2 : (T)1);
        wr *= inv; wi *= inv;
        const T w0r = lane_fetch<T>(wr, base), w0i = lane_fetch<T>(wi, base);
        const T m0 = w0r * w0r + w0i * w0i;
        if (m0 > (T)0) {                                   // rotate so that element 0 is real positive
            const T im0 = Real<T>::rsqrt(m0);
            const T cr = w0r * im0, ci = -w0i * im0;       // conj(w0)/|w0|
            const T tr = wr * cr - wi * ci, ti = wr * ci + wi * cr;
            wr = tr; wi = ti;
        }
        if (real_item && r < N) cal_out[(size_t)item * N + r] = make_float2((float)wr, (float)((r == 0) ? (T)0 : wi));
        return;
    }
    const bool is_noise = real_col && (rank < N - M);
    const unsigned long long noise_mask = __ballot(is_noise);
    const unsigned sel = (unsigned)((noise_mask >> base) & ((1ull << G) - 1ull));     // bit i: column i is a noise vector
    // masked columns: Y = V S
    T yr[G], yi[G];
#pragma unroll
    for (int i = 0; i < G; i++) {
        const bool on = (sel >> i) & 1u;
        yr[i] = on ? vr[i] : (T)0; yi[i] = on ? vi[i] : (T)0;
    }
    if (pn_out) {                                         // diagnostics: P_N[r][b] = sum_i Y[r][i] conj(V[b][i])
        float2 *po = pn_out + (size_t)item * (N * N);
        for (int b = 0; b < N; b++) {
            T pr = 0, pi = 0;
#pragma unroll
            for (int i = 0; i < G; i++) {
                const T br = lane_fetch<T>(vr[i], base + b), bi = lane_fetch<T>(vi[i], base + b);
                pr = fma(yr[i], br, fma(yi[i], bi, pr));
                pi = fma(yi[i], br, fma(-yr[i], bi, pi));
            }
            if (real_item && r < N) po[r + b * N] = make_float2((float)pr, (float)pi);
        }
    }
    // u_l = sum_r P_N[r+l][r] = sum_r sum_i Y[r+l][i] conj(V[r][i]): fetch row r+l, dot with own row, reduce
    float *co = (coef && real_item) ? coef + (size_t)item * (2 * N) : nullptr;
    double *cd = (coef_d && real_item) ? coef_d + (size_t)item * (2 * N) : nullptr;
    for (int l = 0; l < N; l++) {
        T tr = 0, ti = 0;
        const int src = (r + l < G) ? base + r + l : lane;
#pragma unroll
        for (int i = 0; i < G; i++) {
            const T ur = lane_fetch<T>(yr[i], src), ui = lane_fetch<T>(yi[i], src);
            tr = fma(ur, vr[i], fma(ui, vi[i], tr));
            ti = fma(ui, vr[i], fma(-ur, vi[i], ti));
        }
        if (!(r + l < N)) { tr = 0; ti = 0; }
        tr = group_sum<G, T>(tr, lane);
        ti = group_sum<G, T>(ti, lane);
        if (r == 0) {
            if (l == 0) {
                if (co) co[0] = (float)tr;
                if (cd) cd[0] = (double)tr;
            } else {
                if (co) { co[2 * l - 1] = (float)tr; co[2 * l] = (float)ti; }
                if (cd) { cd[2 * l - 1] = (double)tr; cd[2 * l] = (double)ti; }
            }
        }
    }
    if (r == 0) {
        if (co) co[2 * N - 1] = 0.f;
        if (cd) cd[2 * N - 1] = 0.0;
    }
}

template <int G, typename T>
__global__ __launch_bounds__(64) void music_evd_group_kernel(const float2 *__restrict__ R, float *__restrict__ coef,
                                                             double *__restrict__ coef_d, float2 *__restrict__ pn_out,
                                                             int n_items, int N, int M,
                                                             const float2 *__restrict__ pilot, float2 *__restrict__ cal_out)
{
    constexpr int IPW = kWave / G;                       // items per wave
    const int lane = threadIdx.x & (kWave - 1);
    const int r = lane % G, base = lane - r;
    int item = blockIdx.x * IPW + lane / G;
    const bool real_item = item < n_items;
    if (!real_item) item = n_items - 1;                  // idle groups shadow the last item (no stores)
    const float2 *Ri = R + (size_t)item * (N * N);

    T ar[G], ai[G], vr[G], vi[G];
#pragma unroll
    for (int c = 0; c < G; c++) {
        T xr = 0, xi = 0;
        if (r < N && c < N) {
            // upper triangle only (cheevd uplo='U'): A[r][c] = R[r + c N] for r <= c, else conj(R[c + r N])
            const float2 x = (r <= c) ? Ri[r + c * N] : Ri[c + r * N];
            xr = (T)x.x;
            xi = (r == c) ? (T)0 : ((r < c) ? (T)x.y : -(T)x.y);
        } else if (r == c) {
            xr = (T)1e30;                                // padding: isolated, ranks after every real eigenvalue
        }
        ar[c] = xr; ai[c] = xi;
        vr[c] = (r == c) ? (T)1 : (T)0; vi[c] = 0;
    }
    const int max_sweeps = Real<T>::max_sweeps + G;
    bool active = true;
    for (int sweep = 0; sweep < max_sweeps; sweep++) {
        T off = 0, dn = 0;
#pragma unroll
        for (int k = 0; k < G; k++) {
            const T m = ar[k] * ar[k] + ai[k] * ai[k];
            if (r < N && k < N) { if (k == r) dn += m; else off += m; }
        }
        off = group_sum<G, T>(off, lane);
        dn = group_sum<G, T>(dn, lane);
        active = active && (off > Real<T>::tol * dn) && (off > Real<T>::tiny);
        if (!__any(active)) break;
        RoundLoop<G, T, 0>::run(ar, ai, vr, vi, r, base, active);
    }
    // eigenvalue of lane r = A[r][r]
    T lam = 0;
#pragma unroll
    for (int k = 0; k < G; k++) lam = (k == r) ? ar[k] : lam;
    evd_group_epilogue<G, T>(vr, vi, lam, r < N, r, base, lane, item, real_item, N, M, coef, coef_d, pn_out, pilot, cal_out);
}

// ---------------------------------------------------------------------------------------------
// 8 < N <= 16: one WAVE per item, lane (a, b) = (lane >> 3, lane & 7) holds the 2 x 2 block
// A[2a..2a+1][2b..2b+1] and the same block of V.  Pivot pairs sit at fixed physical positions
// (2k, 2k+1) (Brent-Luk): every round the diagonal lane (k, k) builds the rotation of its own block,
// lane (a, b) applies J_a^H from the left and J_b from the right (V: J_b only), and then rows and
// columns move one step along the round-robin "caterpillar" (position 0 fixed), so that after 15
// rounds every index pair has met once.  Rounds are a rolled loop of identical code; the state is 16
// registers per matrix instead of the 64 the row-per-lane layout needs at G = 16 (which compiles to
// 512 VGPRs + scratch, i.e. one wave per SIMD whatever the batch), so 4096 items run as 4+ waves per
// SIMD and the cross-lane latency of one item hides behind the arithmetic of the others.
// Padding (N < 16): zero rows/columns -- never rotated (pivot 0 -> identity), they only travel; a
// 16-bit mask that takes the same permutation tells the epilogue which slots are padding.
// ---------------------------------------------------------------------------------------------
// DPP move of a float or double: lanes whose source lies outside their 16-lane row, or whose bank is masked
// off, receive `old`
template <int CTRL, int BANK_MASK> __device__ __forceinline__ float dpp_row_shift(float old, float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), CTRL, 0xF, BANK_MASK, false));
}
template <int CTRL, int BANK_MASK> __device__ __forceinline__ double dpp_row_shift(double old, double v)
{
    const long long o = __double_as_longlong(old), x = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp((int)(o & 0xFFFFFFFFll), (int)(x & 0xFFFFFFFFll), CTRL, 0xF, BANK_MASK, false);
    const int hi = __builtin_amdgcn_update_dpp((int)(o >> 32), (int)(x >> 32), CTRL, 0xF, BANK_MASK, false);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

template <typename T> __device__ __forceinline__ T wave_sum_t(T v)
{
#pragma unroll
    for (int m = 1; m < kWave; m <<= 1) v += lane_fetch<T>(v, (int)((threadIdx.x & (kWave - 1)) ^ m));
    return v;
}

// one caterpillar step: new[0]=old[0], new[2]=old[1], new[2m]=old[2m-2] (m>=2), new[2m-1]=old[2m+1] (m<=7), new[15]=old[14]
__device__ __forceinline__ unsigned caterpillar_mask(unsigned m)
{
    unsigned n = (m & 1u) | (((m >> 1) & 1u) << 2) | (((m >> 14) & 1u) << 15);
#pragma unroll
    for (int k = 2; k <= 7; k++) n |= ((m >> (2 * k - 2)) & 1u) << (2 * k);
#pragma unroll
    for (int k = 1; k <= 7; k++) n |= ((m >> (2 * k + 1)) & 1u) << (2 * k - 1);
    return n;
}

// LEAN: production outputs only (the coefficient records); its epilogue works out of LDS on all 64
// lanes, so the kernel's register allocation is that of the sweeps (the row-per-lane epilogue alone
// needs ~200 VGPRs in double, which would halve the resident waves).  !LEAN: diagnostics (P_N) and
// calibrate mode through the shared row-per-lane epilogue.
template <typename T, bool LEAN>
__global__ __launch_bounds__(64) void music_evd_block16_kernel(const float2 *__restrict__ R, float *__restrict__ coef,
                                                               double *__restrict__ coef_d, float2 *__restrict__ pn_out,
                                                               int n_items, int N, int M,
                                                               const float2 *__restrict__ pilot, float2 *__restrict__ cal_out)
{
    constexpr int G = 16;
    __shared__ T sVr[G * G], sVi[G * G], sLam[G];
    const int lane = threadIdx.x & (kWave - 1);
    // lane = 16 (a >> 1) + 2 b + (a & 1): the two block rows that share a 16-lane DPP row are interleaved, so that
    // "column block b -> b +- 1" is a DPP row shift by 2 lanes whose out-of-row lanes are exactly b = 0 / b = 7
    const int a = ((lane >> 4) << 1) | (lane & 1), b = (lane & 15) >> 1;
    const int item = blockIdx.x;                         // grid = n_items
    const float2 *Ri = R + (size_t)item * (N * N);
    T xr[2][2], xi[2][2], vr[2][2], vi[2][2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int row = 2 * a + i, col = 2 * b + j;
            T re = 0, im = 0;
            if (row < N && col < N) {
                // upper triangle only (cheevd uplo='U'): A[r][c] = R[r + c N] for r <= c, else conj(R[c + r N])
                const float2 x = (row <= col) ? Ri[row + col * N] : Ri[col + row * N];
                re = (T)x.x;
                im = (row == col) ? (T)0 : ((row < col) ? (T)x.y : -(T)x.y);
            }
            xr[i][j] = re; xi[i][j] = im;
            vr[i][j] = (row == col) ? (T)1 : (T)0; vi[i][j] = 0;
        }
    unsigned pad = (N >= G) ? 0u : (((1u << G) - 1u) & ~((1u << N) - 1u));      // bit k: physical slot k is padding
    auto lane_of = [](int aa, int bb) { return 16 * (aa >> 1) + 2 * bb + (aa & 1); };
    const int src_a = lane_of(a, a), src_b = lane_of(b, b);                 // diagonal lanes (a,a) and (b,b)
    const int ln_u = lane_of((a + 7) & 7, b), ln_d = lane_of((a + 1) & 7, b); // block rows a-1 / a+1 (ends unused)
    const int max_sweeps = Real<T>::max_sweeps + G;
    for (int sweep = 0; sweep < max_sweeps; sweep++) {
        T off = 0, dn = 0;
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const T m = xr[i][j] * xr[i][j] + xi[i][j] * xi[i][j];
                if (a == b && i == j) dn += m; else off += m;
            }
        off = wave_sum_t<T>(off);
        dn = wave_sum_t<T>(dn);
        if (!((off > Real<T>::tol * dn) && (off > Real<T>::tiny))) break;       // wave-uniform: one item per wave
#pragma unroll 1
        for (int round = 0; round < G - 1; round++) {
            // rotation of the own 2 x 2 block (meaningful on the diagonal lanes, harmless elsewhere)
            const T d_p = xr[0][0], d_q = xr[1][1], pr = xr[0][1], pi = xi[0][1];
            const T g2 = pr * pr + pi * pi;
            const bool live = g2 > Real<T>::tiny;
            const T inv_g = Real<T>::rsqrt(live ? g2 : (T)1);
            const T phr = pr * inv_g, phi = pi * inv_g;
            T tau = (d_q - d_p) * (T)0.5 * inv_g;
            tau = fmin(fmax(tau, -Real<T>::tau_max), Real<T>::tau_max);
            const T x1 = fma(tau, tau, (T)1);
            const T rt = x1 * Real<T>::rsqrt(x1);
            const T h = fabs(tau) + rt;
            const T w = Real<T>::rsqrt(fma(h, h, (T)1));
            const T c_mine = live ? h * w : (T)1;
            const T s_mine = live ? copysign(w, tau) : (T)0;
            const T sr_mine = s_mine * phr, si_mine = s_mine * phi;              // sigma = J[p][q]
            const T ca = lane_fetch<T>(c_mine, src_a), sar = lane_fetch<T>(sr_mine, src_a), sai = lane_fetch<T>(si_mine, src_a);
            const T cb = lane_fetch<T>(c_mine, src_b), sbr = lane_fetch<T>(sr_mine, src_b), sbi = lane_fetch<T>(si_mine, src_b);
            // columns: (P, Q) <- (c P - conj(sigma) Q, c Q + sigma P) with the pair of column block b
#pragma unroll
            for (int i = 0; i < 2; i++) {
                {
                    const T p_r = xr[i][0], p_i = xi[i][0], q_r = xr[i][1], q_i = xi[i][1];
                    xr[i][0] = cb * p_r - (sbr * q_r + sbi * q_i);
                    xi[i][0] = cb * p_i - (sbr * q_i - sbi * q_r);
                    xr[i][1] = cb * q_r + (sbr * p_r - sbi * p_i);
                    xi[i][1] = cb * q_i + (sbr * p_i + sbi * p_r);
                }
                {
                    const T p_r = vr[i][0], p_i = vi[i][0], q_r = vr[i][1], q_i = vi[i][1];
                    vr[i][0] = cb * p_r - (sbr * q_r + sbi * q_i);
                    vi[i][0] = cb * p_i - (sbr * q_i - sbi * q_r);
                    vr[i][1] = cb * q_r + (sbr * p_r - sbi * p_i);
                    vi[i][1] = cb * q_i + (sbr * p_i + sbi * p_r);
                }
            }
            // rows of A: (p, q) <- (c p - sigma q, c q + conj(sigma) p) with the pair of row block a
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const T p_r = xr[0][j], p_i = xi[0][j], q_r = xr[1][j], q_i = xi[1][j];
                xr[0][j] = ca * p_r - (sar * q_r - sai * q_i);
                xi[0][j] = ca * p_i - (sar * q_i + sai * q_r);
                xr[1][j] = ca * q_r + (sar * p_r + sai * p_i);
                xi[1][j] = ca * q_i + (sar * p_i - sai * p_r);
            }
            // caterpillar step, columns (A and V): slot 0 <- left neighbour, slot 1 <- right neighbour
            // DPP row shifts by 2 lanes (one block column): lanes without a source (b = 0 for the shift right,
            // b = 7 for the shift left) keep `old`, which is exactly the boundary rule -- no selects.  b = 1 takes
            // slot 1 of b = 0 instead of slot 0: a second shift restricted to DPP bank 0 (lanes 0-3 of the row).
            auto move_cols = [&](T (&m)[2][2]) {
#pragma unroll
                for (int i = 0; i < 2; i++) {
                    const T s0 = m[i][0], s1 = m[i][1];
                    T n0 = dpp_row_shift<0x112, 0xF>(s0, s0);              // row_shr:2, old = own slot 0 (b = 0 keeps it)
                    n0 = dpp_row_shift<0x112, 0x1>(n0, s1);                // bank 0 only: b = 1 <- slot 1 of b = 0
                    const T n1 = dpp_row_shift<0x102, 0xF>(s0, s1);        // row_shl:2, old = own slot 0 (b = 7 takes it)
                    m[i][0] = n0;
                    m[i][1] = n1;
                }
            };
            move_cols(xr); move_cols(xi); move_cols(vr); move_cols(vi);
            // rows (A only): row 0 <- block row above, row 1 <- block row below
            auto move_rows = [&](T (&m)[2][2]) {
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    const T to_down = (a == 0) ? m[1][j] : m[0][j];
                    const T from_up = lane_fetch<T>(to_down, ln_u);
                    const T from_down = lane_fetch<T>(m[1][j], ln_d);
                    const T keep = m[0][j];
                    m[0][j] = (a == 0) ? keep : from_up;
                    m[1][j] = (a == 7) ? keep : from_down;
                }
            };
            move_rows(xr); move_rows(xi);
            pad = caterpillar_mask(pad);
        }
    }
    // hand the result to the row-per-lane epilogue through LDS: V[row][slot], lam[slot]
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) {
            sVr[(2 * a + i) * G + 2 * b + j] = vr[i][j];
            sVi[(2 * a + i) * G + 2 * b + j] = vi[i][j];
        }
    if (a == b) {
        sLam[2 * a] = ((pad >> (2 * a)) & 1u) ? (T)1e30 : xr[0][0];             // padding ranks after every real eigenvalue
        sLam[2 * a + 1] = ((pad >> (2 * a + 1)) & 1u) ? (T)1e30 : xr[1][1];
    }
    __syncthreads();
    if constexpr (LEAN) {
        // ranks (ascending, ties -> lower slot) on lanes 0..15; noise set = the N-M smallest real slots
        bool is_noise = false;
        if (lane < G) {
            const T lam = sLam[lane];
            int rank = 0;
#pragma unroll
            for (int j = 0; j < G; j++) {
                const T lj = sLam[j];
                rank += ((lj < lam) || (lj == lam && j < lane)) ? 1 : 0;
            }
            is_noise = !((pad >> lane) & 1u) && (rank < N - M);
        }
        const unsigned sel = (unsigned)(__ballot(is_noise) & 0xFFFFull);
        // u_l = sum_r sum_{i in noise} V[r+l][i] conj(V[r][i]):  lane = 4 l + c takes the rows r = c (mod 4)
        const int l = lane >> 2, c4 = lane & 3;
        T tr = 0, ti = 0;
        if (l < N) {
            for (int r = c4; r + l < N; r += 4) {
                const T *ur = sVr + (r + l) * G, *ui = sVi + (r + l) * G, *wr = sVr + r * G, *wi = sVi + r * G;
#pragma unroll
                for (int i = 0; i < G; i++)
                    if ((sel >> i) & 1u) {
                        tr = fma(ur[i], wr[i], fma(ui[i], wi[i], tr));
                        ti = fma(ui[i], wr[i], fma(-ur[i], wi[i], ti));
                    }
            }
        }
        tr += lane_fetch<T>(tr, lane ^ 1); ti += lane_fetch<T>(ti, lane ^ 1);
        tr += lane_fetch<T>(tr, lane ^ 2); ti += lane_fetch<T>(ti, lane ^ 2);
        if (c4 == 0 && l < N) {
            float *co = coef ? coef + (size_t)item * (2 * N) : nullptr;
            double *cd = coef_d ? coef_d + (size_t)item * (2 * N) : nullptr;
            if (l == 0) {
                if (co) { co[0] = (float)tr; co[2 * N - 1] = 0.f; }
                if (cd) { cd[0] = (double)tr; cd[2 * N - 1] = 0.0; }
            } else {
                if (co) { co[2 * l - 1] = (float)tr; co[2 * l] = (float)ti; }
                if (cd) { cd[2 * l - 1] = (double)tr; cd[2 * l] = (double)ti; }
            }
        }
    } else if (lane < G) {
        T er[G], ei[G];
#pragma unroll
        for (int k = 0; k < G; k++) { er[k] = sVr[lane * G + k]; ei[k] = sVi[lane * G + k]; }
        evd_group_epilogue<G, T>(er, ei, sLam[lane], !((pad >> lane) & 1u), lane, 0, lane, item, true, N, M, coef, coef_d,
                                 pn_out, pilot, cal_out);
    }
}

// ---------------------------------------------------------------------------------------------
// 4 < N <= 8: the same block scheme on an 8 x 8 problem -- 16 lanes per item (lane 4 b + a of its 16-lane DPP
// row holds the 2 x 2 blocks A[2a..][2b..], V[..]), one item per DPP row, four items per wave.  Column block
// moves are DPP row shifts by 4 lanes (out-of-row lanes = the end blocks, which keep `old`), row block moves
// stay inside a quad (quad_perm); only the rotation parameters travel through ds_bpermute.  7 rounds per
// sweep.  ~75 VGPRs instead of the 280 of the 8-lanes-per-item row layout, and twice as many waves for the
// same batch.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned caterpillar_mask8(unsigned m)
{
    // new[0]=old[0], new[2]=old[1], new[4]=old[2], new[6]=old[4], new[1]=old[3], new[3]=old[5], new[5]=old[7], new[7]=old[6]
    return (m & 1u) | (((m >> 1) & 1u) << 2) | (((m >> 2) & 1u) << 4) | (((m >> 4) & 1u) << 6) | (((m >> 3) & 1u) << 1) |
           (((m >> 5) & 1u) << 3) | (((m >> 7) & 1u) << 5) | (((m >> 6) & 1u) << 7);
}

template <int CTRL> __device__ __forceinline__ float dpp_quad(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), CTRL, 0xF, 0xF, false));
}
template <int CTRL> __device__ __forceinline__ double dpp_quad(double v)
{
    const long long x = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp((int)(x & 0xFFFFFFFFll), (int)(x & 0xFFFFFFFFll), CTRL, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp((int)(x >> 32), (int)(x >> 32), CTRL, 0xF, 0xF, false);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

template <typename T, bool LEAN>
__global__ __launch_bounds__(64) void music_evd_block8_kernel(const float2 *__restrict__ R, float *__restrict__ coef,
                                                              double *__restrict__ coef_d, float2 *__restrict__ pn_out,
                                                              int n_items, int N, int M,
                                                              const float2 *__restrict__ pilot, float2 *__restrict__ cal_out)
{
    constexpr int G = 8;
    __shared__ T sVr[4][G * G], sVi[4][G * G], sLam[4][G];
    const int lane = threadIdx.x & (kWave - 1);
    const int grp = lane >> 4, q = lane & 15, base = lane - q;
    const int a = q & 3, b = q >> 2;
    int item = blockIdx.x * 4 + grp;
    const bool real_item = item < n_items;
    if (!real_item) item = n_items - 1;                  // idle groups shadow the last item (no stores)
    const float2 *Ri = R + (size_t)item * (N * N);
    T xr[2][2], xi[2][2], vr[2][2], vi[2][2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int row = 2 * a + i, col = 2 * b + j;
            T re = 0, im = 0;
            if (row < N && col < N) {
                const float2 x = (row <= col) ? Ri[row + col * N] : Ri[col + row * N];      // upper triangle only
                re = (T)x.x;
                im = (row == col) ? (T)0 : ((row < col) ? (T)x.y : -(T)x.y);
            }
            xr[i][j] = re; xi[i][j] = im;
            vr[i][j] = (row == col) ? (T)1 : (T)0; vi[i][j] = 0;
        }
    unsigned pad = (N >= G) ? 0u : (((1u << G) - 1u) & ~((1u << N) - 1u));
    const int src_a = base + 5 * a, src_b = base + 5 * b;            // diagonal lanes (a,a) and (b,b): q = 4 k + k
    const int max_sweeps = Real<T>::max_sweeps + G;
    bool active = true;
    for (int sweep = 0; sweep < max_sweeps; sweep++) {
        T off = 0, dn = 0;
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const T m = xr[i][j] * xr[i][j] + xi[i][j] * xi[i][j];
                if (a == b && i == j) dn += m; else off += m;
            }
        off = group_sum<16, T>(off, lane);
        dn = group_sum<16, T>(dn, lane);
        active = active && (off > Real<T>::tol * dn) && (off > Real<T>::tiny);
        if (!__any(active)) break;
#pragma unroll 1
        for (int round = 0; round < G - 1; round++) {
            const T d_p = xr[0][0], d_q = xr[1][1], pr = xr[0][1], pi = xi[0][1];
            const T g2 = pr * pr + pi * pi;
            const bool live = active && (g2 > Real<T>::tiny);
            const T inv_g = Real<T>::rsqrt(live ? g2 : (T)1);
            const T phr = pr * inv_g, phi = pi * inv_g;
            T tau = (d_q - d_p) * (T)0.5 * inv_g;
            tau = fmin(fmax(tau, -Real<T>::tau_max), Real<T>::tau_max);
            const T x1 = fma(tau, tau, (T)1);
            const T rt = x1 * Real<T>::rsqrt(x1);
            const T h = fabs(tau) + rt;
            const T w = Real<T>::rsqrt(fma(h, h, (T)1));
            const T c_mine = live ? h * w : (T)1;
            const T s_mine = live ? copysign(w, tau) : (T)0;
            const T sr_mine = s_mine * phr, si_mine = s_mine * phi;              // sigma = J[p][q]
            const T ca = lane_fetch<T>(c_mine, src_a), sar = lane_fetch<T>(sr_mine, src_a), sai = lane_fetch<T>(si_mine, src_a);
            const T cb = lane_fetch<T>(c_mine, src_b), sbr = lane_fetch<T>(sr_mine, src_b), sbi = lane_fetch<T>(si_mine, src_b);
#pragma unroll
            for (int i = 0; i < 2; i++) {
                {
                    const T p_r = xr[i][0], p_i = xi[i][0], q_r = xr[i][1], q_i = xi[i][1];
                    xr[i][0] = cb * p_r - (sbr * q_r + sbi * q_i);
                    xi[i][0] = cb * p_i - (sbr * q_i - sbi * q_r);
                    xr[i][1] = cb * q_r + (sbr * p_r - sbi * p_i);
                    xi[i][1] = cb * q_i + (sbr * p_i + sbi * p_r);
                }
                {
                    const T p_r = vr[i][0], p_i = vi[i][0], q_r = vr[i][1], q_i = vi[i][1];
                    vr[i][0] = cb * p_r - (sbr * q_r + sbi * q_i);
                    vi[i][0] = cb * p_i - (sbr * q_i - sbi * q_r);
                    vr[i][1] = cb * q_r + (sbr * p_r - sbi * p_i);
                    vi[i][1] = cb * q_i + (sbr * p_i + sbi * p_r);
                }
            }
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const T p_r = xr[0][j], p_i = xi[0][j], q_r = xr[1][j], q_i = xi[1][j];
                xr[0][j] = ca * p_r - (sar * q_r - sai * q_i);
                xi[0][j] = ca * p_i - (sar * q_i + sai * q_r);
                xr[1][j] = ca * q_r + (sar * p_r + sai * p_i);
                xi[1][j] = ca * q_i + (sar * p_i - sai * p_r);
            }
            // caterpillar step, columns: DPP row shifts by 4 lanes (one block column); b = 0 / b = 3 keep `old`
            auto move_cols = [&](T (&m)[2][2]) {
#pragma unroll
                for (int i = 0; i < 2; i++) {
                    const T s0 = m[i][0], s1 = m[i][1];
                    T n0 = dpp_row_shift<0x114, 0xF>(s0, s0);              // row_shr:4
                    n0 = dpp_row_shift<0x114, 0x2>(n0, s1);                // bank 1 (b = 1) <- slot 1 of b = 0
                    const T n1 = dpp_row_shift<0x104, 0xF>(s0, s1);        // row_shl:4, b = 3 takes its own slot 0
                    m[i][0] = n0;
                    m[i][1] = n1;
                }
            };
            move_cols(xr); move_cols(xi); move_cols(vr); move_cols(vi);
            // rows (A only): inside the quad a = 0..3
            auto move_rows = [&](T (&m)[2][2]) {
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    const T to_down = (a == 0) ? m[1][j] : m[0][j];
                    const T from_up = dpp_quad<0x90>(to_down);             // quad_perm [0,0,1,2]: lane a <- a-1
                    const T from_down = dpp_quad<0xF9>(m[1][j]);           // quad_perm [1,2,3,3]: lane a <- a+1
                    const T keep = m[0][j];
                    m[0][j] = (a == 0) ? keep : from_up;
                    m[1][j] = (a == 3) ? keep : from_down;
                }
            };
            move_rows(xr); move_rows(xi);
            pad = caterpillar_mask8(pad);
        }
    }
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) {
            sVr[grp][(2 * a + i) * G + 2 * b + j] = vr[i][j];
            sVi[grp][(2 * a + i) * G + 2 * b + j] = vi[i][j];
        }
    if (a == b) {
        sLam[grp][2 * a] = ((pad >> (2 * a)) & 1u) ? (T)1e30 : xr[0][0];
        sLam[grp][2 * a + 1] = ((pad >> (2 * a + 1)) & 1u) ? (T)1e30 : xr[1][1];
    }
    __syncthreads();
    if constexpr (LEAN) {
        bool is_noise = false;
        if (q < G) {
            const T lam = sLam[grp][q];
            int rank = 0;
#pragma unroll
            for (int j = 0; j < G; j++) {
                const T lj = sLam[grp][j];
                rank += ((lj < lam) || (lj == lam && j < q)) ? 1 : 0;
            }
            is_noise = !((pad >> q) & 1u) && (rank < N - M);
        }
        const unsigned sel = (unsigned)((__ballot(is_noise) >> base) & 0xFFull);
        // u_l = sum_r sum_{i in noise} V[r+l][i] conj(V[r][i]):  lane q = 2 l + c takes the rows r = c (mod 2)
        const int l = q >> 1, c2 = q & 1;
        T tr = 0, ti = 0;
        if (l < N) {
            for (int r = c2; r + l < N; r += 2) {
                const T *ur = sVr[grp] + (r + l) * G, *ui = sVi[grp] + (r + l) * G, *wr = sVr[grp] + r * G, *wi = sVi[grp] + r * G;
#pragma unroll
                for (int i = 0; i < G; i++)
                    if ((sel >> i) & 1u) {
                        tr = fma(ur[i], wr[i], fma(ui[i], wi[i], tr));
                        ti = fma(ui[i], wr[i], fma(-ur[i], wi[i], ti));
                    }
            }
        }
        tr += lane_fetch<T>(tr, lane ^ 1); ti += lane_fetch<T>(ti, lane ^ 1);
        if (c2 == 0 && l < N && real_item) {
            float *co = coef ? coef + (size_t)item * (2 * N) : nullptr;
            double *cd = coef_d ? coef_d + (size_t)item * (2 * N) : nullptr;
            if (l == 0) {
                if (co) { co[0] = (float)tr; co[2 * N - 1] = 0.f; }
                if (cd) { cd[0] = (double)tr; cd[2 * N - 1] = 0.0; }
            } else {
                if (co) { co[2 * l - 1] = (float)tr; co[2 * l] = (float)ti; }
                if (cd) { cd[2 * l - 1] = (double)tr; cd[2 * l] = (double)ti; }
            }
        }
    } else if (q < G) {
        T er[G], ei[G];
#pragma unroll
        for (int k = 0; k < G; k++) { er[k] = sVr[grp][q * G + k]; ei[k] = sVi[grp][q * G + k]; }
        evd_group_epilogue<G, T>(er, ei, sLam[grp][q], !((pad >> q) & 1u), q, base, lane, item, real_item, N, M, coef, coef_d,
                                 pn_out, pilot, cal_out);
    }
}

template <typename T>
static void launch_evd_block8(int N, int M, int n_items, const void *d_R, void *d_coef, void *d_coef_d, void *d_pn,
                              hipStream_t st, const void *d_pilot = nullptr, void *d_cal = nullptr)
{
    const dim3 grid((n_items + 3) / 4), block(64);
    if (!d_pn && !d_cal)
        hipLaunchKernelGGL((music_evd_block8_kernel<T, true>), grid, block, 0, st, (const float2 *)d_R, (float *)d_coef,
                           (double *)d_coef_d, nullptr, n_items, N, M, nullptr, nullptr);
    else
        hipLaunchKernelGGL((music_evd_block8_kernel<T, false>), grid, block, 0, st, (const float2 *)d_R, (float *)d_coef,
                           (double *)d_coef_d, (float2 *)d_pn, n_items, N, M, (const float2 *)d_pilot, (float2 *)d_cal);
}

template <typename T>
static void launch_evd_block16(int N, int M, int n_items, const void *d_R, void *d_coef, void *d_coef_d, void *d_pn,
                               hipStream_t st, const void *d_pilot = nullptr, void *d_cal = nullptr)
{
    if (!d_pn && !d_cal)
        hipLaunchKernelGGL((music_evd_block16_kernel<T, true>), dim3(n_items), dim3(64), 0, st, (const float2 *)d_R,
                           (float *)d_coef, (double *)d_coef_d, nullptr, n_items, N, M, nullptr, nullptr);
    else
        hipLaunchKernelGGL((music_evd_block16_kernel<T, false>), dim3(n_items), dim3(64), 0, st, (const float2 *)d_R,
                           (float *)d_coef, (double *)d_coef_d, (float2 *)d_pn, n_items, N, M, (const float2 *)d_pilot,
                           (float2 *)d_cal);
}

template <int G, typename T>
static void launch_evd_group(int N, int M, int n_items, const void *d_R, void *d_coef, void *d_coef_d, void *d_pn,
                             hipStream_t st, const void *d_pilot = nullptr, void *d_cal = nullptr)
{
    constexpr int IPW = kWave / G;
    dim3 block(64), grid((n_items + IPW - 1) / IPW);
    hipLaunchKernelGGL((music_evd_group_kernel<G, T>), grid, block, 0, st, (const float2 *)d_R, (float *)d_coef,
                       (double *)d_coef_d, (float2 *)d_pn, n_items, N, M, (const float2 *)d_pilot, (float2 *)d_cal);
}

// calibrate_lin_array: top eigenvector of each covariance item, de-rotated by the pilot steering vector
int launch_calibrate(int N, int n_items, const void *d_R, const void *d_pilot, void *d_out, int bits, hipStream_t st)
{
    if (n_items <= 0) return DOA_OK;
    const bool f32 = (bits == 32);
    if (N > 8) { if (f32) launch_evd_block16<float>(N, 1, n_items, d_R, nullptr, nullptr, nullptr, st, d_pilot, d_out); else launch_evd_block16<double>(N, 1, n_items, d_R, nullptr, nullptr, nullptr, st, d_pilot, d_out); }
    else if (N > 4) { if (f32) launch_evd_group<8, float>(N, 1, n_items, d_R, nullptr, nullptr, nullptr, st, d_pilot, d_out); else launch_evd_group<8, double>(N, 1, n_items, d_R, nullptr, nullptr, nullptr, st, d_pilot, d_out); }
    else { if (f32) launch_evd_group<4, float>(N, 1, n_items, d_R, nullptr, nullptr, nullptr, st, d_pilot, d_out); else launch_evd_group<4, double>(N, 1, n_items, d_R, nullptr, nullptr, nullptr, st, d_pilot, d_out); }
    DOA_HIP_TRY(hipGetLastError());
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
    // N <= 4: one lane per item, everything in registers (measured 10.5 us vs 11.0 us for the
    // 4-lane group kernel at batch 4096: at this size the cross-lane traffic eats the shorter
    // dependency chain).  N > 4: 8 or 16 lanes per item (15x / 17x faster than one lane per item with
    // the matrices in scratch).  DOA_EVD_KERNEL=1 forces the group kernel for N <= 4 (A/B runs).
    static const int force_group = [] { const char *e = getenv("DOA_EVD_KERNEL"); return e ? atoi(e) : 0; }();
    const bool f32 = (evd_bits == 32);
    static const int block16 = [] { const char *e = getenv("DOA_EVD16_BLOCK"); return e ? atoi(e) : 1; }();
    // 4 < N <= 8: the 16-lanes-per-item block kernel is the faster one on its own (46 vs 63 us per 4096 items at N = 8)
    // but costs 10 % of the 4-stream pipeline throughput (80 vs 72 us per step): its data movement runs on the vector
    // pipe (DPP), which the N = 8 covariance kernel also needs, where the row-per-lane kernel uses the LDS pipe and only
    // half of the SIMDs.  Throughput wins the default; DOA_EVD8_BLOCK=1 selects the low-latency kernel.
    static const int block8 = [] { const char *e = getenv("DOA_EVD8_BLOCK"); return e ? atoi(e) : 0; }();
    if (N > 8 && block16) {
        if (f32) launch_evd_block16<float>(N, M, n_items, d_R, d_coef, d_coef_d, d_pn, st);
        else launch_evd_block16<double>(N, M, n_items, d_R, d_coef, d_coef_d, d_pn, st);
    } else if (N > 8) {
        if (f32) launch_evd_group<16, float>(N, M, n_items, d_R, d_coef, d_coef_d, d_pn, st);
        else launch_evd_group<16, double>(N, M, n_items, d_R, d_coef, d_coef_d, d_pn, st);
    } else if (N > 4 && block8) {
        if (f32) launch_evd_block8<float>(N, M, n_items, d_R, d_coef, d_coef_d, d_pn, st);
        else launch_evd_block8<double>(N, M, n_items, d_R, d_coef, d_coef_d, d_pn, st);
    } else if (N > 4) {
        if (f32) launch_evd_group<8, float>(N, M, n_items, d_R, d_coef, d_coef_d, d_pn, st);
        else launch_evd_group<8, double>(N, M, n_items, d_R, d_coef, d_coef_d, d_pn, st);
    } else if (force_group == 1) {
        if (f32) launch_evd_group<4, float>(N, M, n_items, d_R, d_coef, d_coef_d, d_pn, st);
        else launch_evd_group<4, double>(N, M, n_items, d_R, d_coef, d_coef_d, d_pn, st);
    } else {
        switch (N) {
        case 2: launch_evd_n<2>(M, n_items, d_R, d_coef, d_coef_d, d_pn, evd_bits, st); break;
        case 3: launch_evd_n<3>(M, n_items, d_R, d_coef, d_coef_d, d_pn, evd_bits, st); break;
        default: launch_evd_n<4>(M, n_items, d_R, d_coef, d_coef_d, d_pn, evd_bits, st); break;
        }
    }
    DOA_HIP_TRY(hipGetLastError());
    return DOA_OK;
}

// ---------------------------------------------------------------------------------------------
// K4: spectrum scan — kernels in music_scan_impl.hpp, one translation unit per polynomial size
// ---------------------------------------------------------------------------------------------
DOA_SCAN_SIZES(DOA_SCAN_EXTERN)

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
    // compiled polynomial sizes: 2, 3, 4, 6, 8, 12, 16 (an array of n elements uses the next size up
    // with zero high-order coefficients)
    const int n = t.N;
    if (n < 2 || n > DOA_MAX_ANT_ELE) {
        set_error("MUSIC: num_ant_ele=%d outside the built range 2..%d", n, DOA_MAX_ANT_ELE);
        return DOA_ERR_UNSUPPORTED;
    }
    if (n == 2) done = launch_scan_n<2>(t, bits, n_items, d_coef, d_spec, d_q, pk, st);
    else if (n == 3) done = launch_scan_n<3>(t, bits, n_items, d_coef, d_spec, d_q, pk, st);
    else if (n == 4) done = launch_scan_n<4>(t, bits, n_items, d_coef, d_spec, d_q, pk, st);
    else if (n <= 6) done = launch_scan_n<6>(t, bits, n_items, d_coef, d_spec, d_q, pk, st);
    else if (n <= 8) done = launch_scan_n<8>(t, bits, n_items, d_coef, d_spec, d_q, pk, st);
    else if (n <= 12) done = launch_scan_n<12>(t, bits, n_items, d_coef, d_spec, d_q, pk, st);
    else done = launch_scan_n<16>(t, bits, n_items, d_coef, d_spec, d_q, pk, st);
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
    if (int brc = doa::bind_device(h->device); brc != DOA_OK) return brc;
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
    if (int brc = doa::bind_device(h->device); brc != DOA_OK) return brc;
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
    if (int brc = doa::bind_device(h->device); brc != DOA_OK) return brc;
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
