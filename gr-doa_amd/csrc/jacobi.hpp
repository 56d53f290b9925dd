// jacobi.hpp — device code of the batched Hermitian Jacobi EVD (K2+K3): working-precision traits, the rotation parameters, the pre-scaling, the cyclic sweep
// on one matrix per lane, and the per-item "covariance -> coefficient record" routine.
//
// Replaces eig_sym (LAPACK cheevd 'V','U') + U_N U_N^H of the reference (lib/MUSIC_lin_array_impl.cc:128-133,
// lib/rootMUSIC_linear_array_impl.cc:112-116) and the diagonal sums of lib/rootMUSIC_linear_array_impl.cc:74-79.
#pragma once
#include "common.hpp"

namespace doa {

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

// Rotation parameters of one Jacobi pivot (a_pp, a_qq real; a_pq = xr + j xi):
//     J[p][p] = J[q][q] = c,   J[p][q] = sigma,   J[q][p] = -conj(sigma),   sigma = c kappa a_pq,
//     kappa = sgn(d) / (|d| + sqrt(d^2 + |a_pq|^2)),  d = (a_qq - a_pp)/2      (t = kappa |a_pq| is the smaller root
// of t^2 + 2 tau t - 1 = 0, tau = d/|a_pq|: the classical choice, |t| <= 1).
// Only the ANGLE decides how much of the pivot a rotation removes; what the accuracy of the eigenvectors
// needs is that J be unitary to working precision, i.e. c^2 (1 + kappa^2 |a_pq|^2) = 1.  So kappa is formed in
// float with the hardware v_sqrt_f32 / v_rcp_f32 (it leaves a pivot at ~1e-7 of its size instead of 1e-16; the
// next sweep takes it the rest of the way, exactly as it does for the pivots the later rotations of the same sweep
// refill) and only c = rsqrt(1 + kappa^2 |a_pq|^2) is formed in T: ONE rsqrt on [1, 2] seeded by v_rsq_f32 and
// polished by one third-order step (full T precision).  The pivot's exact residue rho a_pq is kept in A.  This
// replaces three dependent T-precision rsqrt (each a 2^-24 seed plus two Newton steps in double) per rotation:
// the parameter chain is what a batched small-matrix Jacobi waits on.
// Matrices are pre-scaled to max |entry| in [1, 2) (jacobi_prescale), so the float intermediates cannot
// overflow and a pivot with |a_pq|^2 < 1e-30 is zero at any working precision.
// rho: (J^H A J)[p][q] = rho a_pq, the pivot's residue; ss = |sigma|^2; csg = 2 c Re(conj(sigma) a_pq)
template <typename T> struct JacobiRot { T c, sr, si, rho, ss, csg; };
template <typename T>
__device__ __forceinline__ JacobiRot<T> jacobi_rotation(T app, T aqq, T xr, T xi, bool enable = true)
{
    const T d = (T)0.5 * (aqq - app);
    const T g2 = fma(xi, xi, xr * xr);
    const float df = (float)d, g2f = (float)g2;
    const float rad = __builtin_amdgcn_sqrtf(fmaf(df, df, g2f));
    const float kf = copysignf(__builtin_amdgcn_rcpf(fabsf(df) + rad), df);
    const bool live = enable && (g2f > 1e-30f) && (fabsf(kf) < 3e38f);       // (NaN compares false)
    const T kappa = live ? (T)kf : (T)0;
    const T x = fma(kappa * g2, kappa, (T)1);               // 1 + t^2, in [1, 2]
    T y = (T)__builtin_amdgcn_rsqf((float)x);
    const T e = fma(-x * y, y, (T)1);
    y = fma(y * e, fma((T)0.375, e, (T)0.5), y);            // third-order step: 2^-23 seed -> full T precision
    const T ck = y * kappa;
    // (J^H A J)[p][q] = c^2 a_pq (1 - 2 kappa d - kappa^2 |a_pq|^2): zero for the exact kappa, ~1e-7 for the float one
    const T rho = (y * y) * fma((T)-2 * kappa, d, (T)2 - x);
    const T sr = ck * xr, si = ck * xi;
    return {y, sr, si, rho, fma(sr, sr, si * si), (T)2 * y * fma(sr, xr, si * xi)};
}

// The classical parameters, everything in T: c, s and e^{j phi} from rsqrt only (three dependent T-precision rsqrt),
// the pivot annihilated to working precision (rho = 0).  Slower on its own than jacobi_rotation (10.4 against 9.6 us
// per 4096 4 x 4 items, one lane per item), but the one-lane-per-item kernel built on it costs the 4-stream pipeline
// 1-2 us LESS per step (same-box A/B, three repetitions: 26.1-26.3 against 26.9-28.1 us): that kernel shares its SIMDs
// with the covariance kernel of the next batch, and what counts there is how its instruction stream interleaves, not
// its length.  So the lane-per-item kernel uses this one; the lanes-per-item kernels (N > 4) use jacobi_rotation.
template <typename T>
__device__ __forceinline__ JacobiRot<T> jacobi_rotation_classic(T app, T aqq, T xr, T xi)
{
    const T g2 = xr * xr + xi * xi;
    const bool live = g2 > Real<T>::tiny;                        // a pivot that is already zero gets the identity
    const T inv_g = Real<T>::rsqrt(live ? g2 : (T)1);
    const T phr = xr * inv_g, phi = xi * inv_g;                   // e^{j phi}
    T tau = (aqq - app) * (T)0.5 * inv_g;
    tau = fmin(fmax(tau, -Real<T>::tau_max), Real<T>::tau_max);
    const T x1 = fma(tau, tau, (T)1);
    const T r = x1 * Real<T>::rsqrt(x1);                          // sqrt(1 + tau^2)
    const T h = fabs(tau) + r;                                    // 1/|t|
    const T w = Real<T>::rsqrt(fma(h, h, (T)1));
    const T g = live ? g2 * inv_g : (T)0;
    const T c = live ? h * w : (T)1;
    const T sn = live ? copysign(w, tau) : (T)0;
    return {c, sn * phr, sn * phi, (T)0, sn * sn, (T)2 * c * sn * g};
}

// Exact power-of-two scale that brings max |entry| into [1, 2) (eigenvectors, ranks and therefore P_N do not
// depend on the scale; eigenvalues themselves are never output).  m = max |entry| as float; 0, inf, NaN -> 1.
template <typename T> __device__ __forceinline__ T jacobi_prescale(float m)
{
    const int e = (__float_as_int(m) >> 23) & 0xff;
    if (e == 0 || e == 255) return (T)1;
    return (T)__int_as_float((254 - e) << 23);              // 2^-(e-127)
}

// Cyclic complex Jacobi on a Hermitian matrix kept as its real diagonal dg[] and strict upper
// triangle (ur, ui)[r][c], r < c (the lower triangle is never formed: A[c][r] = conj(A[r][c])).
// On return dg holds the eigenvalues and the columns of V = (vr, vi) the eigenvectors.
// One rotation (p,q): J[p][p] = J[q][q] = c, J[p][q] = sigma = s e^{j phi}, J[q][p] = -conj(sigma)
// with phi = arg A[p][q] and t = s/c the smaller root of t^2 + 2 tau t - 1 = 0,
// tau = (A[q][q]-A[p][p]) / (2|A[p][q]|).  c, s and e^{j phi} are built from rsqrt only, so J is
// unitary to working precision and no division appears.  A <- J^H A J touches, for every k not in
// {p,q}, the pair (A[k][p], A[k][q]) and the two diagonal entries; V <- V J touches columns p, q.
template <int N, typename T, bool UNROLL, bool CLASSIC = false>
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
                const JacobiRot<T> rot = CLASSIC ? jacobi_rotation_classic<T>(dg[p], dg[q], apr, api)
                                                 : jacobi_rotation<T>(dg[p], dg[q], apr, api);
                const T c = rot.c, spr = rot.sr, spi = rot.si;         // sigma
                // 2x2 block: a_pp' = c^2 a_pp - 2c Re(conj(sigma) a_pq) + |sigma|^2 a_qq,  a_qq' = |sigma|^2 a_pp + 2c Re(..) + c^2 a_qq,
                // a_pq' = rho a_pq: what J^H A J really leaves there (the float angle does not annihilate it exactly)
                {
                    const T cc = c * c, ss = rot.ss, csg = rot.csg;
                    const T app = dg[p], aqq = dg[q];
                    dg[p] = fma(cc, app, fma(ss, aqq, -csg));
                    dg[q] = fma(ss, app, fma(cc, aqq, csg));
                    if constexpr (CLASSIC) { ur[p][q] = 0; ui[p][q] = 0; }
                    else { ur[p][q] = rot.rho * apr; ui[p][q] = rot.rho * api; }
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

// One covariance item -> its coefficient record, everything in this lane's registers (N <= 4 unrolled):
// upper triangle of R (cheevd uplo='U'; element (r,c) at r + c*N) -> prescale -> cyclic Jacobi ->
// ascending ranks (the eig_sym contract; ties -> lower index) -> P_N = sum over the N-M smallest of v v^H
// -> u_l = sum_r P_N[r+l][r].  u[0] = u_0, u[2l-1] + j u[2l] = u_l, u[2N-1] = 0.  pn_out (optional):
// column-major P_N as float2.
template <int N, typename T>
__device__ __forceinline__ void evd_item_coefficients(const float2 *__restrict__ Ri, int M, T (&u)[2 * N],
                                                      float2 *__restrict__ pn_out = nullptr)
{
    constexpr bool UNROLL = (N <= 4);
    constexpr int U = UNROLL ? N : 1;
    T dg[N], ar[N][N], ai[N][N], vr[N][N], vi[N][N];
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
    // 0, or NaN when the item holds a non-finite entry (0 * inf = 0 * NaN = NaN).  (No pre-scaling here: the classical
    // rotation works in T throughout, and float-origin data cannot leave double's range.)
    T poison = 0;
#pragma unroll U
    for (int c = 0; c < N; c++) {
        poison = fma(dg[c], (T)0, poison);
#pragma unroll U
        for (int r = 0; r < N; r++)
            if (r < c) poison = fma(ar[r][c], (T)0, fma(ai[r][c], (T)0, poison));
    }

    herm_jacobi<N, T, UNROLL, true>(dg, ar, ai, vr, vi);

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
    // a non-finite item yields a non-finite projector (the reference: eig_sym fails and the block throws), never
    // a plausible-looking one built from an identity V
#pragma unroll U
    for (int a = 0; a < N; a++) {
        dg[a] += poison;
#pragma unroll U
        for (int b = 0; b < N; b++)
            if (b > a) { ar[a][b] += poison; ai[a][b] += poison; }
    }
    if (pn_out) {
        float2 *po = pn_out;
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
    // diagonal sums u_l = sum_r P_N[r+l][r] = conj(sum_r P_N[r][r+l])
#pragma unroll U
    for (int l = 0; l < N; l++) {
        T ur_ = 0, ui_ = 0;
#pragma unroll U
        for (int r = 0; r < N; r++)
            if (r + l < N) {
                if (l == 0) ur_ += dg[r];
                else { ur_ += ar[r][r + l]; ui_ -= ai[r][r + l]; }
            }
        if (l == 0) u[0] = ur_;
        else { u[2 * l - 1] = ur_; u[2 * l] = ui_; }
    }
    u[2 * N - 1] = 0;
}

}  // namespace doa
