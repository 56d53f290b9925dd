// evd_subspace.hpp — K2+K3 for wide arrays (4 < N <= 16) with few sources (M <= 4): the noise projector from the SIGNAL
// subspace, found by shifted orthogonal (subspace) iteration, one wave per covariance item.
//
// What the reference computes (lib/MUSIC_lin_array_impl.cc:128-133, lib/rootMUSIC_linear_array_impl.cc:112-116): eig_sym,
// U_N = the eigenvectors of the N-M smallest eigenvalues, P_N = U_N U_N^H.  Whenever eigenvalue M and M+1 (descending) are
// distinct, P_N = I - U_S U_S^H with U_S any orthonormal basis of the invariant subspace of the M LARGEST eigenvalues --
// and that subspace is what orthogonal iteration X <- orth((A - mu I) X) converges to, at the rate
// max_noise |lambda_j - mu| / (lambda_M - mu) per step.  With mu = the mean of the N-M remaining eigenvalues (from the
// traces: mu = (tr A - tr X^H A X) / (N-M)) that ratio is the noise eigenvalues' SPREAD over the signal-to-noise gap:
// 1e-2..1e-3 on array data, i.e. 4-6 steps to a residual of 3e-14 ||A||, where the cyclic Jacobi needs ~11 sweeps of 15
// rounds (41 k wave instructions per 16 x 16 item against ~3 k here).
//
// Every exit of the fast path is CHECKED, and anything unusual falls back to the full Jacobi EVD of this file's includer
// (evd_block16_item, the kernel this replaces as the default), so results never depend on the iteration having worked:
//   * convergence: ||A X - X T||_F <= 3e-14 ||A||_F with T = X^H A X formed explicitly (checked after 4, 6, 8, 11, 15, 20 steps);
//   * certificate that X spans the TOP-M eigenspace and not some other invariant subspace: with E = ||A - mu I||_F^2 -
//     ||T - mu I||_F^2 (= the sum of (lambda_j - mu)^2 over the N-M eigenvalues outside span X) every outside eigenvalue
//     satisfies |lambda_j - mu| <= sqrt(E); the smallest Ritz value satisfies theta_min - mu >= det(T') ((M-1)/tr T')^(M-1),
//     T' = T - mu I (AM-GM on the other M-1 eigenvalues); if that bound exceeds sqrt(E) every Ritz value lies above every
//     outside eigenvalue.  Equal eigenvalues across the signal/noise boundary (R = c I, ...) can never pass, so the
//     reference's tie behaviour stays the Jacobi kernel's (ranks by index);
//   * Cholesky breakdown, non-finite input, no convergence in 20 steps: fall back.
//
// Layout: lane = 16 g + r'.  DPP row g (16 lanes) owns column g of X (g < MC = M, other rows idle in the product); lane r'
// holds row r = r' mod G of that column, G = 8 (N <= 8: both halves of the DPP row hold the same data, so a 16-lane
// rotation is an 8-row rotation) or 16.  A is kept SKEWED in registers, As[s] = A[r][(r+s) mod G], so the product
// y = A x needs x rotated by s lanes (DPP row_ror) and a register with a compile-time index: G x (4 DPP moves + 4 FMA64).
// Orthonormalisation is Cholesky-QR (Gram matrix by DPP row sums, Cholesky and the triangular solve redundantly in every
// lane); the other columns of a row come through ds_bpermute.  Epilogue: X in LDS, u_l = N delta_l0 - sum_r sum_c
// X[r+l][c] conj(X[r][c]) (the Root-MUSIC / scan coefficient records), optionally P_N itself.
#pragma once

namespace doa {

template <int CTRL> __device__ __forceinline__ double dpp_mov_d(double v)
{
    const long long x = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(x & 0xFFFFFFFFll), CTRL, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(x >> 32), CTRL, 0xF, 0xF, false);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
// sum over the 16 lanes of a DPP row, result in every lane of the row
__device__ __forceinline__ double row_sum16(double v)
{
    v += dpp_mov_d<0xB1>(v);        // quad_perm [1,0,3,2]
    v += dpp_mov_d<0x4E>(v);        // quad_perm [2,3,0,1]
    v += dpp_mov_d<0x124>(v);       // row_ror:4
    v += dpp_mov_d<0x128>(v);       // row_ror:8
    return v;
}
__device__ __forceinline__ double shfl_d(double v, int src_lane) { return __shfl(v, src_lane, kWave); }

// y += As[S] * rot_S(x), S = 1..G-1 (rot_S: lane r' takes the value of lane (r' + S) mod 16 = row_ror:(16 - S))
template <int G, int S> struct SkewMac {
    static __device__ __forceinline__ void run(const double (&ar)[G], const double (&ai)[G], double xr, double xi, double &yr, double &yi)
    {
        const double pr = dpp_mov_d<0x120 + (16 - S)>(xr), pi = dpp_mov_d<0x120 + (16 - S)>(xi);
        yr = fma(ar[S], pr, fma(-ai[S], pi, yr));
        yi = fma(ar[S], pi, fma(ai[S], pr, yi));
        if constexpr (S + 1 < G) SkewMac<G, S + 1>::run(ar, ai, xr, xi, yr, yi);
    }
};

// returns true when the item's records were written by the fast path; false = the caller must run the Jacobi kernel body
template <int G, int MC, bool PN>
__device__ __forceinline__ bool evd_subspace_item(const float2 *__restrict__ Ri, int item, float *__restrict__ coef,
                                                  double *__restrict__ coef_d, float2 *__restrict__ pn_out, int N,
                                                  double *__restrict__ sXr, double *__restrict__ sXi)
{
    static_assert(G == 8 || G == 16, "rows per column");
    static_assert(MC >= 1 && MC <= 4, "columns");
    const int lane = threadIdx.x & (kWave - 1);
    const int g = lane >> 4, rp = lane & 15, r = rp & (G - 1);
    constexpr double kDup = (G == 8) ? 0.5 : 1.0;              // a 16-lane row sum counts every row twice when G = 8 (exact)
    auto elem = [&](int row, int col) -> float2 {              // A[row][col] from the upper triangle (cheevd uplo = 'U')
        if (row >= N || col >= N) return make_float2(0.f, 0.f);
        if (row == col) return make_float2(Ri[row + col * N].x, 0.f);
        const float2 x = (row < col) ? Ri[row + col * N] : Ri[col + row * N];
        return make_float2(x.x, (row < col) ? x.y : -x.y);
    };
    double ar[G], ai[G];
    float m = 0.f;
#pragma unroll
    for (int s = 0; s < G; s++) {
        const float2 e = elem(r, (r + s) & (G - 1));
        ar[s] = (double)e.x; ai[s] = (double)e.y;
        m = fmaxf(m, fmaxf(fabsf(e.x), fabsf(e.y)));
    }
    m = wave_allreduce_max(m);
    if (!(m > 0.f) || !(m < INFINITY)) return false;           // zero, NaN (fmaxf drops it: checked again below) or inf
    const double sc = jacobi_prescale<double>(m);
    double nrm2 = 0.0, poison = 0.0;
#pragma unroll
    for (int s = 0; s < G; s++) {
        poison = fma(ar[s], 0.0, fma(ai[s], 0.0, poison));     // NaN if any entry is non-finite
        ar[s] *= sc; ai[s] *= sc;
        nrm2 = fma(ar[s], ar[s], fma(ai[s], ai[s], nrm2));
    }
    nrm2 = row_sum16(nrm2) * kDup;                             // ||A||_F^2 (every DPP row holds all of A)
    poison = row_sum16(poison);
    if (__builtin_amdgcn_ballot_w64(poison != 0.0) != 0ull) return false;
    const double trA = row_sum16(ar[0]) * kDup;

    // xc[c] / yc[c]: columns 0..MC-1 of X / Y at this lane's row (every DPP row keeps all of them); own column = g
    double xcr[MC], xci[MC], ycr[MC], yci[MC];
    // Cholesky-QR of the columns in (ycr, yci) -> (xcr, xci); false on breakdown
    auto cholqr = [&]() -> bool {
        double gr[MC][MC], gi[MC][MC];                         // Gram matrix, upper triangle; then L (lower) in place
#pragma unroll
        for (int i = 0; i < MC; i++)
#pragma unroll
            for (int j = i; j < MC; j++) {
                double pr = fma(ycr[i], ycr[j], yci[i] * yci[j]);               // conj(y_i) y_j
                pr = row_sum16(pr) * kDup;
                double pi = 0.0;
                if (j > i) { pi = fma(ycr[i], yci[j], -yci[i] * ycr[j]); pi = row_sum16(pi) * kDup; }
                gr[i][j] = pr; gi[i][j] = pi;
            }
        // L[j][k], k <= j, with L L^H = Gram: stored as lr/li[j][k]; inv[j] = 1 / L[j][j]
        double lr[MC][MC], li[MC][MC], inv[MC];
        bool ok = true;
#pragma unroll
        for (int j = 0; j < MC; j++) {
            double d = gr[j][j];
#pragma unroll
            for (int k = 0; k < j; k++) d -= fma(lr[j][k], lr[j][k], li[j][k] * li[j][k]);
            ok = ok && (d > 1e-280) && (d < 1e280);
            const double dd = ok ? d : 1.0;
            inv[j] = Real<double>::rsqrt(dd);
            lr[j][j] = dd * inv[j]; li[j][j] = 0.0;
#pragma unroll
            for (int i = j + 1; i < MC; i++) {
                // L[i][j] = (G[i][j] - sum_k L[i][k] conj(L[j][k])) / L[j][j],  G[i][j] = conj(G[j][i])
                double tr = gr[j][i], ti = -gi[j][i];
#pragma unroll
                for (int k = 0; k < j; k++) {
                    tr -= fma(lr[i][k], lr[j][k], li[i][k] * li[j][k]);
                    ti -= fma(li[i][k], lr[j][k], -lr[i][k] * li[j][k]);
                }
                lr[i][j] = tr * inv[j]; li[i][j] = ti * inv[j];
            }
        }
        if (__builtin_amdgcn_ballot_w64(!ok) != 0ull) return false;
        // Y = Q R, R = L^H: q_j = (y_j - sum_{i<j} q_i conj(L[j][i])) / L[j][j]
#pragma unroll
        for (int j = 0; j < MC; j++) {
            double tr = ycr[j], ti = yci[j];
#pragma unroll
            for (int i = 0; i < j; i++) {
                // q_i * conj(L[j][i])
                tr -= fma(xcr[i], lr[j][i], xci[i] * li[j][i]);
                ti -= fma(xci[i], lr[j][i], -xcr[i] * li[j][i]);
            }
            xcr[j] = tr * inv[j]; xci[j] = ti * inv[j];
        }
        return true;
    };
    // start: the first MC columns of A (A e_c = sum lambda_i v_i conj(v_i[c]): dominated by the large eigenvalues)
#pragma unroll
    for (int c = 0; c < MC; c++) {
        const float2 e = elem(r, c);
        ycr[c] = (double)e.x * sc; yci[c] = (double)e.y * sc;
    }
    if (!cholqr()) return false;
    double mu = 0.0;
    const double inv_nm = 1.0 / (double)(N - MC);
    bool done = false;
    int next_check = 4, gap = 2;
    for (int it = 1; it <= 20; it++) {
        // own column of X, then y = A x - mu x
        double xr = xcr[0], xi = xci[0];
#pragma unroll
        for (int c = 1; c < MC; c++) { xr = (g == c) ? xcr[c] : xr; xi = (g == c) ? xci[c] : xi; }
        double yr = fma(ar[0], xr, fma(-ai[0], xi, -mu * xr));
        double yi = fma(ar[0], xi, fma(ai[0], xr, -mu * xi));
        SkewMac<G, 1>::run(ar, ai, xr, xi, yr, yi);
        // all columns of Y at this row
#pragma unroll
        for (int c = 0; c < MC; c++) { ycr[c] = shfl_d(yr, 16 * c + rp); yci[c] = shfl_d(yi, 16 * c + rp); }
        if (it == next_check) {
            // T' = X^H Y = X^H A X - mu I (upper triangle, mirrored), residual ||Y - X T'||_F
            double tr_[MC][MC], ti_[MC][MC];
#pragma unroll
            for (int i = 0; i < MC; i++)
#pragma unroll
                for (int j = i; j < MC; j++) {
                    double pr = row_sum16(fma(xcr[i], ycr[j], xci[i] * yci[j])) * kDup;
                    double pi = 0.0;
                    if (j > i) pi = row_sum16(fma(xcr[i], yci[j], -xci[i] * ycr[j])) * kDup;
                    tr_[i][j] = pr; ti_[i][j] = pi;
                    tr_[j][i] = pr; ti_[j][i] = -pi;
                }
            double res2 = 0.0, t2 = 0.0, trT = 0.0;
#pragma unroll
            for (int j = 0; j < MC; j++) {
                double rr = ycr[j], ri = yci[j];
#pragma unroll
                for (int i = 0; i < MC; i++) {
                    rr -= fma(xcr[i], tr_[i][j], -xci[i] * ti_[i][j]);
                    ri -= fma(xcr[i], ti_[i][j], xci[i] * tr_[i][j]);
                    t2 = fma(tr_[i][j], tr_[i][j], fma(ti_[i][j], ti_[i][j], t2));
                }
                res2 = fma(rr, rr, fma(ri, ri, res2));
                trT += tr_[j][j];
            }
            res2 = row_sum16(res2) * kDup;
            // wave-uniform by construction (every DPP row holds the same numbers); the ballot makes it uniform by force
            const bool conv = res2 <= (3e-14 * 3e-14) * nrm2;
            if (__builtin_amdgcn_ballot_w64(!conv) == 0ull) {
                // certificate (header comment): Cholesky of T' for its determinant, AM-GM bound on theta_min - mu
                double lr[MC][MC], li[MC][MC], det = 1.0;
                bool pd = true;
#pragma unroll
                for (int j = 0; j < MC; j++) {
                    double d = tr_[j][j];
#pragma unroll
                    for (int k = 0; k < j; k++) d -= fma(lr[j][k], lr[j][k], li[j][k] * li[j][k]);
                    pd = pd && (d > 0.0);
                    const double dd = pd ? d : 1.0;
                    det *= dd;
                    const double iv = Real<double>::rsqrt(dd);
                    lr[j][j] = dd * iv; li[j][j] = 0.0;
#pragma unroll
                    for (int i = j + 1; i < MC; i++) {
                        double a_ = tr_[i][j], b_ = ti_[i][j];
#pragma unroll
                        for (int k = 0; k < j; k++) {
                            a_ -= fma(lr[i][k], lr[j][k], li[i][k] * li[j][k]);
                            b_ -= fma(li[i][k], lr[j][k], -lr[i][k] * li[j][k]);
                        }
                        lr[i][j] = a_ * iv; li[i][j] = b_ * iv;
                    }
                }
                double bound = det;
                if constexpr (MC > 1) {
                    const double f = (double)(MC - 1) / trT;
#pragma unroll
                    for (int k = 0; k < MC - 1; k++) bound *= f;
                }
                const double En = fma((double)N * mu, mu, fma(-2.0 * mu, trA, nrm2)) - t2;
                const bool cert = pd && (trT > 0.0) && (bound > 0.0) && (bound * bound > 1.02 * fmax(En, 0.0) + 1e-12 * nrm2);
                if (__builtin_amdgcn_ballot_w64(!cert) != 0ull) return false;
                done = true;
                break;
            }
            next_check += gap;
            gap = (gap < 5) ? gap + (it >= 8 ? 1 : 0) : gap;      // checks after 4, 6, 8, 11, 15, 20 steps
            if (it == 8) gap = 3;
            if (it == 11) gap = 4;
            if (it == 15) gap = 5;
        }
        // shift = mean of the eigenvalues outside span X, from the traces
        double tsum = 0.0;
#pragma unroll
        for (int c = 0; c < MC; c++) tsum = fma(xcr[c], ycr[c], fma(xci[c], yci[c], tsum));
        tsum = row_sum16(tsum) * kDup;                         // tr(X^H A X) - MC mu
        mu = (trA - tsum - (double)MC * mu) * inv_nm;
        if (!cholqr()) return false;
    }
    if (!done) return false;

    // ---- epilogue: X (orthonormal, MC columns) -> records --------------------------------------------------------
    if (g == 0 && rp < G) {
#pragma unroll
        for (int c = 0; c < MC; c++) { sXr[r * 4 + c] = xcr[c]; sXi[r * 4 + c] = xci[c]; }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_barrier();                              // one wave per workgroup: orders the LDS writes before the reads
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    {
        // u_l = N delta_l0 - sum_r sum_c X[r+l][c] conj(X[r][c]); lane = 4 l + c4 takes the rows r = c4 (mod 4)
        const int l = lane >> 2, c4 = lane & 3;
        double tr = 0.0, ti = 0.0;
        if (l < N) {
            for (int rr = c4; rr + l < N; rr += 4) {
#pragma unroll
                for (int c = 0; c < MC; c++) {
                    const double ur = sXr[(rr + l) * 4 + c], ui = sXi[(rr + l) * 4 + c], wr = sXr[rr * 4 + c], wi = sXi[rr * 4 + c];
                    tr = fma(ur, wr, fma(ui, wi, tr));
                    ti = fma(ui, wr, fma(-ur, wi, ti));
                }
            }
        }
        tr += shfl_d(tr, lane ^ 1); ti += shfl_d(ti, lane ^ 1);
        tr += shfl_d(tr, lane ^ 2); ti += shfl_d(ti, lane ^ 2);
        if (c4 == 0 && l < N) {
            float *co = coef ? coef + (size_t)item * (2 * N) : nullptr;
            double *cd = coef_d ? coef_d + (size_t)item * (2 * N) : nullptr;
            if (l == 0) {
                const double u0 = (double)N - tr;
                if (co) { co[0] = (float)u0; co[2 * N - 1] = 0.f; }
                if (cd) { cd[0] = u0; cd[2 * N - 1] = 0.0; }
            } else {
                if (co) { co[2 * l - 1] = (float)-tr; co[2 * l] = (float)-ti; }
                if (cd) { cd[2 * l - 1] = -tr; cd[2 * l] = -ti; }
            }
        }
    }
    if constexpr (PN) {
        if (pn_out) {
            // P_N[a][b] = delta_ab - sum_c X[a][c] conj(X[b][c]), column-major N x N
            float2 *po = pn_out + (size_t)item * (N * N);
            for (int e = lane; e < N * N; e += kWave) {
                const int a = e % N, b = e / N;
                double pr = (a == b) ? 1.0 : 0.0, pi = 0.0;
#pragma unroll
                for (int c = 0; c < MC; c++) {
                    const double ur = sXr[a * 4 + c], ui = sXi[a * 4 + c], wr = sXr[b * 4 + c], wi = sXi[b * 4 + c];
                    pr -= fma(ur, wr, ui * wi);
                    pi -= fma(ui, wr, -ur * wi);
                }
                po[e] = make_float2((float)pr, (float)pi);
            }
        }
    }
    return true;
}

}  // namespace doa

namespace doa {

// ---------------------------------------------------------------------------------------------------------------------
// N <= 4: the same signal-subspace iteration with ONE LANE per covariance item, everything in that lane's registers and no
// cross-lane traffic (the Gram matrices are MC x MC sums over N <= 4 rows).  Same checks as above (residual 3e-14 ||A||,
// top-M certificate, Cholesky breakdown, non-finite input); returns false for an item that must take the Jacobi routine
// (evd_item_coefficients), which the kernel then runs for the lanes that need it.  ~100 FP64 instructions per step at
// N = 4, M = 1 and 4-6 steps on array data, against ~4000 for the cyclic Jacobi.
// u: [u0, Re u1, Im u1, ...] (2N values, the last one 0), u_l = sum_r P_N[r+l][r], P_N = I - X X^H.
// ---------------------------------------------------------------------------------------------------------------------
template <int N, int MC>
__device__ __forceinline__ bool evd_small_subspace(const float2 *__restrict__ Ri, double (&u)[2 * N], float2 *__restrict__ pn_out)
{
    static_assert(N >= 2 && N <= 4 && MC >= 1 && MC < N, "sizes");
    // A (Hermitian, from the upper triangle: cheevd uplo = 'U'), full storage
    double ar[N][N], ai[N][N];
    double nrm2 = 0.0, trA = 0.0, poison = 0.0;
    float m = 0.f;
#pragma unroll
    for (int c = 0; c < N; c++)
#pragma unroll
        for (int r = 0; r <= c; r++) {
            const float2 e = Ri[r + c * N];
            const double re = (double)e.x, im = (r == c) ? 0.0 : (double)e.y;
            ar[r][c] = re; ai[r][c] = im; ar[c][r] = re; ai[c][r] = -im;
            m = fmaxf(m, fmaxf(fabsf(e.x), fabsf((r == c) ? 0.f : e.y)));
            poison = fma(re, 0.0, fma(im, 0.0, poison));
        }
    if (!(m > 0.f) || !(m < INFINITY) || poison != 0.0) return false;
    const double sc = jacobi_prescale<double>(m);
#pragma unroll
    for (int r = 0; r < N; r++)
#pragma unroll
        for (int c = 0; c < N; c++) {
            ar[r][c] *= sc; ai[r][c] *= sc;
            nrm2 = fma(ar[r][c], ar[r][c], fma(ai[r][c], ai[r][c], nrm2));
            if (r == c) trA += ar[r][c];
        }
    double xr[N][MC], xi[N][MC], yr[N][MC], yi[N][MC];
    // Cholesky-QR of Y -> X; false on breakdown
    auto cholqr = [&]() -> bool {
        double lr[MC][MC], li[MC][MC], inv[MC];
        bool ok = true;
#pragma unroll
        for (int j = 0; j < MC; j++) {
            // column j of the Gram matrix below the diagonal: G[i][j] = sum_r conj(y[r][i]) y[r][j], i >= j
            double gr[MC], gi[MC];
#pragma unroll
            for (int i = j; i < MC; i++) {
                double pr = 0.0, pi = 0.0;
#pragma unroll
                for (int r = 0; r < N; r++) {
                    pr = fma(yr[r][i], yr[r][j], fma(yi[r][i], yi[r][j], pr));
                    pi = fma(yr[r][i], yi[r][j], fma(-yi[r][i], yr[r][j], pi));
                }
                gr[i] = pr; gi[i] = pi;
            }
            double d = gr[j];
#pragma unroll
            for (int k = 0; k < j; k++) d -= fma(lr[j][k], lr[j][k], li[j][k] * li[j][k]);
            ok = ok && (d > 1e-280) && (d < 1e280);
            const double dd = ok ? d : 1.0;
            inv[j] = Real<double>::rsqrt(dd);
            lr[j][j] = dd * inv[j]; li[j][j] = 0.0;
#pragma unroll
            for (int i = j + 1; i < MC; i++) {
                // L[i][j] = (G[i][j] - sum_k L[i][k] conj(L[j][k])) / L[j][j];  G[i][j] = sum conj(y_i) y_j
                double tr = gr[i], ti = gi[i];
#pragma unroll
                for (int k = 0; k < j; k++) {
                    tr -= fma(lr[i][k], lr[j][k], li[i][k] * li[j][k]);
                    ti -= fma(li[i][k], lr[j][k], -lr[i][k] * li[j][k]);
                }
                lr[i][j] = tr * inv[j]; li[i][j] = ti * inv[j];
            }
        }
        if (!ok) return false;
        // note the index order above: G[i][j] as computed is conj(y_i)^T y_j, i.e. the (i, j) entry of Y^H Y; L L^H = Y^H Y
        // with L lower triangular, and Q = Y L^-H: q_j = (y_j - sum_{i<j} q_i conj(L[j][i])) / L[j][j]
#pragma unroll
        for (int j = 0; j < MC; j++)
#pragma unroll
            for (int r = 0; r < N; r++) {
                double tr = yr[r][j], ti = yi[r][j];
#pragma unroll
                for (int i = 0; i < j; i++) {
                    tr -= fma(xr[r][i], lr[j][i], xi[r][i] * li[j][i]);
                    ti -= fma(xi[r][i], lr[j][i], -xr[r][i] * li[j][i]);
                }
                xr[r][j] = tr * inv[j]; xi[r][j] = ti * inv[j];
            }
        return true;
    };
#pragma unroll
    for (int r = 0; r < N; r++)
#pragma unroll
        for (int c = 0; c < MC; c++) { yr[r][c] = ar[r][c]; yi[r][c] = ai[r][c]; }
    if (!cholqr()) return false;
    double mu = 0.0;
    const double inv_nm = 1.0 / (double)(N - MC);
    bool done = false;
    int next_check = 4;
    for (int it = 1; it <= 20; it++) {
        double tsum = 0.0;
#pragma unroll
        for (int c = 0; c < MC; c++)
#pragma unroll
            for (int r = 0; r < N; r++) {
                double sr = -mu * xr[r][c], si = -mu * xi[r][c];
#pragma unroll
                for (int k = 0; k < N; k++) {
                    sr = fma(ar[r][k], xr[k][c], fma(-ai[r][k], xi[k][c], sr));
                    si = fma(ar[r][k], xi[k][c], fma(ai[r][k], xr[k][c], si));
                }
                yr[r][c] = sr; yi[r][c] = si;
                tsum = fma(xr[r][c], sr, fma(xi[r][c], si, tsum));
            }
        if (it == next_check) {
            double tr_[MC][MC], ti_[MC][MC];
#pragma unroll
            for (int i = 0; i < MC; i++)
#pragma unroll
                for (int j = i; j < MC; j++) {
                    double pr = 0.0, pi = 0.0;
#pragma unroll
                    for (int r = 0; r < N; r++) {
                        pr = fma(xr[r][i], yr[r][j], fma(xi[r][i], yi[r][j], pr));
                        pi = fma(xr[r][i], yi[r][j], fma(-xi[r][i], yr[r][j], pi));
                    }
                    if (i == j) pi = 0.0;
                    tr_[i][j] = pr; ti_[i][j] = pi; tr_[j][i] = pr; ti_[j][i] = -pi;
                }
            double res2 = 0.0, t2 = 0.0, trT = 0.0;
#pragma unroll
            for (int j = 0; j < MC; j++) {
#pragma unroll
                for (int r = 0; r < N; r++) {
                    double rr = yr[r][j], ri = yi[r][j];
#pragma unroll
                    for (int i = 0; i < MC; i++) {
                        rr -= fma(xr[r][i], tr_[i][j], -xi[r][i] * ti_[i][j]);
                        ri -= fma(xr[r][i], ti_[i][j], xi[r][i] * tr_[i][j]);
                    }
                    res2 = fma(rr, rr, fma(ri, ri, res2));
                }
#pragma unroll
                for (int i = 0; i < MC; i++) t2 = fma(tr_[i][j], tr_[i][j], fma(ti_[i][j], ti_[i][j], t2));
                trT += tr_[j][j];
            }
            if (res2 <= (3e-14 * 3e-14) * nrm2) {
                // certificate: Cholesky of T' for its determinant, AM-GM bound on theta_min - mu against sqrt(E)
                double lr[MC][MC], li[MC][MC], det = 1.0;
                bool pd = true;
#pragma unroll
                for (int j = 0; j < MC; j++) {
                    double d = tr_[j][j];
#pragma unroll
                    for (int k = 0; k < j; k++) d -= fma(lr[j][k], lr[j][k], li[j][k] * li[j][k]);
                    pd = pd && (d > 0.0);
                    const double dd = pd ? d : 1.0;
                    det *= dd;
                    const double iv = Real<double>::rsqrt(dd);
                    lr[j][j] = dd * iv; li[j][j] = 0.0;
#pragma unroll
                    for (int i = j + 1; i < MC; i++) {
                        double a_ = tr_[i][j], b_ = ti_[i][j];
#pragma unroll
                        for (int k = 0; k < j; k++) {
                            a_ -= fma(lr[i][k], lr[j][k], li[i][k] * li[j][k]);
                            b_ -= fma(li[i][k], lr[j][k], -lr[i][k] * li[j][k]);
                        }
                        lr[i][j] = a_ * iv; li[i][j] = b_ * iv;
                    }
                }
                double bound = det;
                if constexpr (MC > 1) {
                    const double f = (double)(MC - 1) / trT;
#pragma unroll
                    for (int k = 0; k < MC - 1; k++) bound *= f;
                }
                const double En = fma((double)N * mu, mu, fma(-2.0 * mu, trA, nrm2)) - t2;
                if (!(pd && (trT > 0.0) && (bound > 0.0) && (bound * bound > 1.02 * fmax(En, 0.0) + 1e-12 * nrm2))) return false;
                done = true;
                break;
            }
            next_check += (it < 8) ? 2 : (it == 8 ? 3 : (it == 11 ? 4 : 5));      // checks after 4, 6, 8, 11, 15, 20 steps
        }
        mu = (trA - tsum - (double)MC * mu) * inv_nm;
        if (!cholqr()) return false;
    }
    if (!done) return false;
    // P_N = I - X X^H and its diagonal sums
#pragma unroll
    for (int l = 0; l < N; l++) {
        double sr = 0.0, si = 0.0;
#pragma unroll
        for (int r = 0; r + l < N; r++)
#pragma unroll
            for (int c = 0; c < MC; c++) {
                sr = fma(xr[r + l][c], xr[r][c], fma(xi[r + l][c], xi[r][c], sr));
                si = fma(xi[r + l][c], xr[r][c], fma(-xr[r + l][c], xi[r][c], si));
            }
        if (l == 0) u[0] = (double)N - sr;
        else { u[2 * l - 1] = -sr; u[2 * l] = -si; }
    }
    u[2 * N - 1] = 0.0;
    if (pn_out) {
#pragma unroll
        for (int b = 0; b < N; b++)
#pragma unroll
            for (int a = 0; a < N; a++) {
                double pr = (a == b) ? 1.0 : 0.0, pi = 0.0;
#pragma unroll
                for (int c = 0; c < MC; c++) {
                    pr -= fma(xr[a][c], xr[b][c], xi[a][c] * xi[b][c]);
                    pi -= fma(xi[a][c], xr[b][c], -xr[a][c] * xi[b][c]);
                }
                pn_out[a + b * N] = make_float2((float)pr, (float)pi);
            }
    }
    return true;
}

}  // namespace doa
