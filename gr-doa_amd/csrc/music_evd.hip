// music_evd.hip — K2+K3 of MUSIC_lin_array / rootMUSIC_linear_array / calibrate_lin_array on gfx950: batched
// Hermitian EVD (cyclic complex Jacobi, double by default) -> noise projector -> diagonal sums u_l.
//
// Replaces (reference): lib/MUSIC_lin_array_impl.cc:121-133 (eig_sym -> LAPACK cheevd 'V','U', ascending;
// U_N = first N-M vectors; P_N = U_N U_N^H), lib/rootMUSIC_linear_array_impl.cc:74-79,108-116.
//   N <= 4   music_evd_kernel          one lane per covariance matrix, everything in registers (jacobi.hpp)
//   N <= 8   music_evd_group_kernel    8 lanes per matrix, one row of A and V per lane, round-robin ordering
//            (also instantiated for 4 and 16 lanes: calibrate mode, A/B runs)
//   N <= 16  music_evd_block16_kernel  one wave per matrix, 2 x 2 blocks per lane (Brent-Luk)
// Rotations: jacobi_rotation (float angle, T-precision unitarity); matrices pre-scaled to max |entry| in [1, 2).
// Output: one 2N-value coefficient record per item ([u0, Re u1, Im u1, ...]); P_N itself on request.
#include "jacobi.hpp"
#include "kernels.hpp"
#include <mutex>
#include <type_traits>

#include <cstdlib>

namespace doa {

}  // namespace doa
#include "evd_subspace.hpp"
namespace doa {

// Q(psi) = u0 + 2 sum_l (x_l cos(l psi) - y_l sin(l psi)), u_l = x_l + j y_l, in powers of c = cos psi and s = sin psi
// (cos 2x = 2c^2 - 1, cos 3x = 4c^3 - 3c, sin 2x = 2sc, sin 3x = s(4c^2 - 1)):  Q = A(c) + s B(c),
// A = (u0 - 2x2) + (2x1 - 6x3) c + 4x2 c^2 + 8x3 c^3,  B = (2y3 - 2y1) - 4y2 c - 8y3 c^2   (music_scan_impl.hpp: ChebQ)
// ux[l], uy[l]: u_l for l = 0..3 (zero beyond the array size)
__device__ __forceinline__ void write_cheb_record(double *__restrict__ o, const double (&ux)[4], const double (&uy)[4])
{
    o[0] = ux[0] - 2 * ux[2]; o[1] = 2 * ux[1] - 6 * ux[3]; o[2] = 4 * ux[2]; o[3] = 8 * ux[3];
    o[4] = 2 * uy[3] - 2 * uy[1]; o[5] = -4 * uy[2]; o[6] = -8 * uy[3]; o[7] = 0.0;
}

template <int N, typename T>
__global__ __launch_bounds__(64) void music_evd_kernel(const float2 *__restrict__ R, float *__restrict__ coef,
                                                       double *__restrict__ coef_d, float2 *__restrict__ pn_out,
                                                       int n_items, int M, double *__restrict__ cheb_d,
                                                       unsigned long long *__restrict__ fallback_count)
{
    const int item = blockIdx.x * blockDim.x + threadIdx.x;
    if (item >= n_items) return;
    T u[2 * N];
    const float2 *Ri = R + (size_t)item * (N * N);
    float2 *pn_i = pn_out ? pn_out + (size_t)item * (N * N) : nullptr;
    // double: the signal-subspace iteration first (evd_subspace.hpp, one lane per item); the lanes it does not certify run the
    // cyclic Jacobi below (the whole wave skips it when every lane is done)
    // -- for ONE source only: with one lane per item a wave runs as long as its slowest lane, and with two or three sources
    // the spread of step counts over 64 items (median 4-6, some 11-15, a few fall-backs) costs more than it saves: measured
    // 11.9 against 10.8 us per 4096 items at M = 2 (configs[2]) and 30 against 10.4 us on the simulation flowgraph's shape
    bool done = false;
    if constexpr (sizeof(T) == 8) {
        if (M == 1) {
            done = evd_small_subspace<N, 1>(Ri, u, pn_i);
            if (!done && fallback_count) atomicAdd(fallback_count, 1ull);
        }
    }
    if (!done) evd_item_coefficients<N, T>(Ri, M, u, pn_i);
    // float record for the float scan; double record for the root finder (Root-MUSIC's near-double roots amplify
    // a float rounding of u_l by ~1e3-1e4) and for the double scan
    if (coef) {
#pragma unroll
        for (int k = 0; k < 2 * N; k++) coef[(size_t)item * (2 * N) + k] = (float)u[k];
    }
    if (coef_d) {
#pragma unroll
        for (int k = 0; k < 2 * N; k++) coef_d[(size_t)item * (2 * N) + k] = (double)u[k];
    }
    if (cheb_d) {
        const double ux[4] = {(double)u[0], (N > 1) ? (double)u[1] : 0.0, (N > 2) ? (double)u[3] : 0.0, (N > 3) ? (double)u[5] : 0.0};
        const double uy[4] = {0.0, (N > 1) ? (double)u[2] : 0.0, (N > 2) ? (double)u[4] : 0.0, (N > 3) ? (double)u[6] : 0.0};
        write_cheb_record(cheb_d + (size_t)item * kChebRecord, ux, uy);
    }
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
    // Both lanes of a pair evaluate this, but only the lower lane's result is used (below): its
    // partner sees A[q][p], which equals conj(A[p][q]) only to rounding, and two almost-equal
    // rotations are not one unitary rotation once the pivot has shrunk to that level.
    const JacobiRot<T> rot = jacobi_rotation<T>(d_own, d_oth, xr, xi, active);
    const int lo = (r < partner) ? r : partner;
    const T c = from_lower(rot.c, lo);
    const T slr = from_lower(rot.sr, lo), sli = from_lower(rot.si, lo);               // sigma = J[lo][hi]
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
                                                   const float2 *__restrict__ pilot, float2 *__restrict__ cal_out,
                                                   double *__restrict__ cheb_d = nullptr)
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
    double ux[4] = {0, 0, 0, 0}, uy[4] = {0, 0, 0, 0};       // u_l = ux + j uy for the lean scan's record (G == 4 only)
    auto diagonal_sum = [&](int l, T &tr, T &ti) {
        tr = 0; ti = 0;
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
    };
    if constexpr (G == 4) {                                  // unrolled: the record copy keeps static indices
#pragma unroll
        for (int l = 0; l < 4; l++) {
            if (l < N) {
                T tr, ti;
                diagonal_sum(l, tr, ti);
                ux[l] = (double)tr; uy[l] = (l == 0) ? 0.0 : (double)ti;
            }
        }
    } else {
        for (int l = 0; l < N; l++) {
            T tr, ti;
            diagonal_sum(l, tr, ti);
        }
    }
    if (r == 0) {
        if (co) co[2 * N - 1] = 0.f;
        if (cd) cd[2 * N - 1] = 0.0;
        if constexpr (G == 4) {
            if (cheb_d && real_item) write_cheb_record(cheb_d + (size_t)item * kChebRecord, ux, uy);
        }
    }
}

// the whole group Jacobi of one wave: G lanes per item, `real_item` = this lane's group stores its results
template <int G, typename T>
__device__ __forceinline__ void evd_group_wave(const float2 *__restrict__ Ri, int item, bool real_item, int N, int M,
                                               float *__restrict__ coef, double *__restrict__ coef_d, float2 *__restrict__ pn_out,
                                               const float2 *__restrict__ pilot, float2 *__restrict__ cal_out,
                                               double *__restrict__ cheb_d = nullptr);

template <int G, typename T>
__global__ __launch_bounds__(64) void music_evd_group_kernel(const float2 *__restrict__ R, float *__restrict__ coef,
                                                             double *__restrict__ coef_d, float2 *__restrict__ pn_out,
                                                             int n_items, int N, int M,
                                                             const float2 *__restrict__ pilot, float2 *__restrict__ cal_out)
{
    constexpr int IPW = kWave / G;                       // items per wave
    const int lane = threadIdx.x & (kWave - 1);
    int item = blockIdx.x * IPW + lane / G;
    const bool real_item = item < n_items;
    if (!real_item) item = n_items - 1;                  // idle groups shadow the last item (no stores)
    evd_group_wave<G, T>(R + (size_t)item * (N * N), item, real_item, N, M, coef, coef_d, pn_out, pilot, cal_out);
}

template <int G, typename T>
__device__ __forceinline__ void evd_group_wave(const float2 *__restrict__ Ri, int item, bool real_item, int N, int M,
                                               float *__restrict__ coef, double *__restrict__ coef_d, float2 *__restrict__ pn_out,
                                               const float2 *__restrict__ pilot, float2 *__restrict__ cal_out,
                                               double *__restrict__ cheb_d)
{
    const int lane = threadIdx.x & (kWave - 1);
    const int r = lane % G, base = lane - r;

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
    T poison = 0;            // 0, or NaN when the item holds a non-finite entry
    {
        float m = 0.f;
#pragma unroll
        for (int c = 0; c < G; c++)
            if (r < N && c < N) {
                m = fmaxf(m, fmaxf(fabsf((float)ar[c]), fabsf((float)ai[c])));
                poison = fma(ar[c], (T)0, fma(ai[c], (T)0, poison));
            }
        poison = group_sum<G, T>(poison, lane);
#pragma unroll
        for (int k = 1; k < G; k <<= 1) m = fmaxf(m, __shfl(m, lane ^ k, kWave));
        const T sc = jacobi_prescale<T>(m);
#pragma unroll
        for (int c = 0; c < G; c++)
            if (r < N && c < N) { ar[c] *= sc; ai[c] *= sc; }
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
    // a non-finite item yields non-finite outputs (the reference: eig_sym fails and the block throws), never a
    // plausible-looking record built from an identity V
#pragma unroll
    for (int k = 0; k < G; k++) vr[k] += poison;
    evd_group_epilogue<G, T>(vr, vi, lam, r < N, r, base, lane, item, real_item, N, M, coef, coef_d, pn_out, pilot, cal_out, cheb_d);
}

// ---------------------------------------------------------------------------------------------
// N <= 4 with two or three sources (round 4): the signal-subspace iteration of evd_subspace.hpp with FOUR LANES PER ITEM --
// a DPP quad, 16 items per wave.  The one-lane form (evd_small_subspace) loses at M >= 2: a wave runs as long as the slowest of
// its 64 items and every lane carries the whole 4 x 4 . 4 x M product as one dependent chain (11.9 against the Jacobi's 10.8 us
// per 4096 items at M = 2, 30 against 10.4 on forward-backward data).  Here lane r of a quad holds ROW r: of A, skewed
// (As[s] = A[r][(r + s) mod 4], so that y = A x is three quad rotations of x against registers with compile-time indices), and
// of every column of X and Y; the Gram / Ritz sums are two-step quad reductions, the M x M Cholesky and the triangular
// solve run redundantly in the four lanes, nothing touches LDS.  Same checks, same constants as the other two forms
// (residual 3e-14 ||A||, top-M certificate, Cholesky breakdown, non-finite or zero input, 20 steps) and the same
// consequence: a quad that fails any of them takes the cyclic Jacobi -- evd_group_wave<4> in the same wave, which keeps
// the reference's behaviour for ties (ranks by index).  Rows r >= N are zero padding (they stay zero through the iteration).
// ---------------------------------------------------------------------------------------------
template <int S> __device__ __forceinline__ double quad_rot(double v)      // lane r takes the value of quad lane (r + S) mod 4
{
    return quad_fetch<quad_pattern(S & 3, (1 + S) & 3, (2 + S) & 3, (3 + S) & 3)>(v);
}
__device__ __forceinline__ double quad_sum(double v)
{
    v += quad_fetch<quad_pattern(1, 0, 3, 2)>(v);
    v += quad_fetch<quad_pattern(2, 3, 0, 1)>(v);
    return v;                                                  // the same bits in the four lanes (a + b == b + a)
}

// ok (quad-uniform): X holds an orthonormal basis of the top-MC eigenspace, certified
template <int MC>
__device__ __forceinline__ bool evd_quad_subspace(const float2 *__restrict__ Ri, int N, int r, double (&xcr)[MC], double (&xci)[MC])
{
    static_assert(MC >= 1 && MC <= 3, "columns");
    auto elem = [&](int row, int col) -> float2 {              // A[row][col] from the upper triangle (cheevd uplo = 'U')
        if (row >= N || col >= N) return make_float2(0.f, 0.f);
        if (row == col) return make_float2(Ri[row + col * N].x, 0.f);
        const float2 x = (row < col) ? Ri[row + col * N] : Ri[col + row * N];
        return make_float2(x.x, (row < col) ? x.y : -x.y);
    };
    double ar[4], ai[4];
    float m = 0.f;
#pragma unroll
    for (int s = 0; s < 4; s++) {
        const float2 e = elem(r, (r + s) & 3);
        ar[s] = (double)e.x; ai[s] = (double)e.y;
        m = fmaxf(m, fmaxf(fabsf(e.x), fabsf(e.y)));
    }
    m = fmaxf(m, quad_fetch<quad_pattern(1, 0, 3, 2)>(m));
    m = fmaxf(m, quad_fetch<quad_pattern(2, 3, 0, 1)>(m));
    bool ok = (m > 0.f) && (m < INFINITY);                     // zero, NaN (fmaxf drops it: the poison below) or inf -> Jacobi
    const double sc = jacobi_prescale<double>(ok ? m : 1.f);
    double nrm2 = 0.0, poison = 0.0;
#pragma unroll
    for (int s = 0; s < 4; s++) {
        poison = fma(ar[s], 0.0, fma(ai[s], 0.0, poison));     // NaN if any entry is non-finite
        ar[s] *= sc; ai[s] *= sc;
        nrm2 = fma(ar[s], ar[s], fma(ai[s], ai[s], nrm2));
    }
    nrm2 = quad_sum(nrm2);                                     // ||A||_F^2
    poison = quad_sum(poison);
    ok = ok && (poison == 0.0);
    const double trA = quad_sum(ar[0]);

    double ycr[MC], yci[MC];
    // Cholesky-QR of the columns in (ycr, yci) -> (xcr, xci); false on breakdown (quad-uniform)
    auto cholqr = [&]() -> bool {
        double gr[MC][MC], gi[MC][MC];                         // Gram matrix, upper triangle
#pragma unroll
        for (int i = 0; i < MC; i++)
#pragma unroll
            for (int j = i; j < MC; j++) {
                gr[i][j] = quad_sum(fma(ycr[i], ycr[j], yci[i] * yci[j]));              // conj(y_i) y_j
                gi[i][j] = (j > i) ? quad_sum(fma(ycr[i], yci[j], -yci[i] * ycr[j])) : 0.0;
            }
        double lr[MC][MC], li[MC][MC], inv[MC];
        bool good = true;
#pragma unroll
        for (int j = 0; j < MC; j++) {
            double d = gr[j][j];
#pragma unroll
            for (int k = 0; k < j; k++) d -= fma(lr[j][k], lr[j][k], li[j][k] * li[j][k]);
            good = good && (d > 1e-280) && (d < 1e280);
            const double dd = good ? d : 1.0;
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
        // Y = Q R, R = L^H: q_j = (y_j - sum_{i<j} q_i conj(L[j][i])) / L[j][j]   (harmless numbers when !good)
#pragma unroll
        for (int j = 0; j < MC; j++) {
            double tr = ycr[j], ti = yci[j];
#pragma unroll
            for (int i = 0; i < j; i++) {
                tr -= fma(xcr[i], lr[j][i], xci[i] * li[j][i]);
                ti -= fma(xci[i], lr[j][i], -xcr[i] * li[j][i]);
            }
            xcr[j] = tr * inv[j]; xci[j] = ti * inv[j];
        }
        return good;
    };
    // start: the first MC columns of A
#pragma unroll
    for (int c = 0; c < MC; c++) {
        const float2 e = elem(r, c);
        ycr[c] = (double)e.x * sc; yci[c] = (double)e.y * sc;
    }
    ok = cholqr() && ok;
    double mu = 0.0;
    const double inv_nm = 1.0 / (double)(N - MC);
    bool done = false;                                         // done / ok are quad-uniform; the wave leaves the loop on a ballot
    int next_check = 4;
    for (int it = 1; it <= 20; it++) {
        // y_c = A x_c - mu x_c for every column, at this lane's row
        double tsum = 0.0;
#pragma unroll
        for (int c = 0; c < MC; c++) {
            double yr = fma(ar[0], xcr[c], fma(-ai[0], xci[c], -mu * xcr[c]));
            double yi = fma(ar[0], xci[c], fma(ai[0], xcr[c], -mu * xci[c]));
            {
                const double pr = quad_rot<1>(xcr[c]), pi = quad_rot<1>(xci[c]);
                yr = fma(ar[1], pr, fma(-ai[1], pi, yr)); yi = fma(ar[1], pi, fma(ai[1], pr, yi));
            }
            {
                const double pr = quad_rot<2>(xcr[c]), pi = quad_rot<2>(xci[c]);
                yr = fma(ar[2], pr, fma(-ai[2], pi, yr)); yi = fma(ar[2], pi, fma(ai[2], pr, yi));
            }
            {
                const double pr = quad_rot<3>(xcr[c]), pi = quad_rot<3>(xci[c]);
                yr = fma(ar[3], pr, fma(-ai[3], pi, yr)); yi = fma(ar[3], pi, fma(ai[3], pr, yi));
            }
            // a finished (or failed) quad keeps its X: its Y is not used again
            ycr[c] = yr; yci[c] = yi;
            tsum = fma(xcr[c], yr, fma(xci[c], yi, tsum));
        }
        if (it == next_check) {
            // T' = X^H Y = X^H A X - mu I (upper triangle, mirrored), residual ||Y - X T'||_F
            double tr_[MC][MC], ti_[MC][MC];
#pragma unroll
            for (int i = 0; i < MC; i++)
#pragma unroll
                for (int j = i; j < MC; j++) {
                    const double pr = quad_sum(fma(xcr[i], ycr[j], xci[i] * yci[j]));
                    const double pi = (j > i) ? quad_sum(fma(xcr[i], yci[j], -xci[i] * ycr[j])) : 0.0;
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
            res2 = quad_sum(res2);
            const bool conv = res2 <= (3e-14 * 3e-14) * nrm2;
            {
                // certificate (evd_subspace.hpp, header): Cholesky of T' for its determinant, AM-GM bound on theta_min - mu.
                // Evaluated at every check, converged or not: the same two numbers -- sqrt(E) >= every outside |lambda_j - mu|,
                // `bound` <= theta_min - mu -- also bound the iteration's RATE from above, and an item whose rate says "ten more
                // steps" is handed to the Jacobi now instead of after 20 steps (a wave pays for its slowest quad)
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
                const bool sane = pd && (trT > 0.0) && (bound > 0.0);
                const bool cert = sane && (bound * bound > 1.02 * fmax(En, 0.0) + 1e-12 * nrm2);
                if (ok && !done) {
                    if (conv) {
                        if (cert) done = true;                  // X stays as it is from here on
                        else ok = false;                        // converged to SOME invariant subspace: the Jacobi decides
                    } else {
                        // steps still to go, from above: the squared residual falls by at least bound^2 / En per step.  A step
                        // costs this wave 0.8 us, the Jacobi it would run instead 9: more than kMoreSteps to go is not worth
                        // iterating (the same comparison at every check; the old rule -- rate above 0.2 -- let quads run to the
                        // 20-step limit AND take the Jacobi: 21.6 us per 4096 items at 5 dB SNR against the one-lane Jacobi's
                        // 11.2, profiles/r04_lab_evd_quad_low_snr.txt)
                        constexpr float kMoreSteps = 11.f;
                        bool slow = !sane;
                        if (sane) {
                            const float need = __log2f((float)(res2 / ((3e-14 * 3e-14) * nrm2)));
                            const float gain = __log2f((float)((bound * bound) / fmax(En, 1e-300)));
                            slow = !(gain * kMoreSteps >= need);                 // (NaN: slow)
                        }
                        if (slow) ok = false;
                    }
                }
                // the Jacobi runs for the whole wave or not at all: once one quad needs it, iterating on in the others only adds
                // to the wave's time
                if (__builtin_amdgcn_ballot_w64(!ok) != 0ull && !done) ok = false;
            }
            next_check += (it < 8) ? 2 : (it == 8 ? 3 : (it == 11 ? 4 : 5));      // checks after 4, 6, 8, 11, 15, 20 steps
        }
        if (__builtin_amdgcn_ballot_w64(ok && !done) == 0ull) break;
        // shift = mean of the eigenvalues outside span X, from the traces; next X (only for quads still iterating)
        tsum = quad_sum(tsum);                                 // tr(X^H A X) - MC mu
        const double mu_next = (trA - tsum - (double)MC * mu) * inv_nm;
        double kr[MC], ki[MC];
#pragma unroll
        for (int c = 0; c < MC; c++) { kr[c] = xcr[c]; ki[c] = xci[c]; }
        const bool good = cholqr();
        const bool live = ok && !done;
        if (live) { mu = mu_next; ok = good; }
        else {
#pragma unroll
            for (int c = 0; c < MC; c++) { xcr[c] = kr[c]; xci[c] = ki[c]; }
        }
    }
    return ok && done;
}

// one DPP quad per item, 16 items per wave; MC = num_targets (2 or 3; N = MC + 1 .. 4)
template <int MC, bool PN>
__global__ __launch_bounds__(64) void music_evd_quad_kernel(const float2 *__restrict__ R, float *__restrict__ coef,
                                                            double *__restrict__ coef_d, float2 *__restrict__ pn_out, int n_items,
                                                            int N, double *__restrict__ cheb_d,
                                                            unsigned long long *__restrict__ fallback_count)
{
    const int lane = threadIdx.x & (kWave - 1);
    const int r = lane & 3;
    int item = blockIdx.x * 16 + (lane >> 2);
    const bool real_item = item < n_items;
    if (!real_item) item = n_items - 1;                        // idle quads shadow the last item (no stores)
    const float2 *Ri = R + (size_t)item * (N * N);
    double xcr[MC], xci[MC];
    const bool fast = evd_quad_subspace<MC>(Ri, N, r, xcr, xci);
    if (fast && real_item) {
        // u_l = N delta_l0 - sum_r sum_c X[r+l][c] conj(X[r][c]), r + l < N: row r + l is l lanes further on in the quad
        double ux[4] = {0, 0, 0, 0}, uy[4] = {0, 0, 0, 0};
        auto diag = [&](auto l_tag) {
            constexpr int l = decltype(l_tag)::value;
            double tr = 0.0, ti = 0.0;
#pragma unroll
            for (int c = 0; c < MC; c++) {
                const double ur = quad_rot<l>(xcr[c]), ui = quad_rot<l>(xci[c]);
                tr = fma(ur, xcr[c], fma(ui, xci[c], tr));
                ti = fma(ui, xcr[c], fma(-ur, xci[c], ti));
            }
            if (!(r + l < N)) { tr = 0.0; ti = 0.0; }
            ux[l] = ((l == 0) ? (double)N : 0.0) - quad_sum(tr);
            uy[l] = (l == 0) ? 0.0 : -quad_sum(ti);
        };
        diag(std::integral_constant<int, 0>{});
        diag(std::integral_constant<int, 1>{});
        diag(std::integral_constant<int, 2>{});
        diag(std::integral_constant<int, 3>{});
        if (r == 0) {
            float *co = coef ? coef + (size_t)item * (2 * N) : nullptr;
            double *cd = coef_d ? coef_d + (size_t)item * (2 * N) : nullptr;
#pragma unroll
            for (int l = 0; l < 4; l++) {
                if (l < N) {
                    if (l == 0) {
                        if (co) co[0] = (float)ux[0];
                        if (cd) cd[0] = ux[0];
                    } else {
                        if (co) { co[2 * l - 1] = (float)ux[l]; co[2 * l] = (float)uy[l]; }
                        if (cd) { cd[2 * l - 1] = ux[l]; cd[2 * l] = uy[l]; }
                    }
                }
            }
            if (co) co[2 * N - 1] = 0.f;
            if (cd) cd[2 * N - 1] = 0.0;
            if (cheb_d) write_cheb_record(cheb_d + (size_t)item * kChebRecord, ux, uy);
        }
        if constexpr (PN) {
            if (pn_out) {
                // P_N[a][b] = delta_ab - sum_c X[a][c] conj(X[b][c]); this lane: a = r, b = (r + s) mod 4
                float2 *po = pn_out + (size_t)item * (N * N);
                auto col = [&](auto s_tag) {
                    constexpr int s = decltype(s_tag)::value;
                    const int b = (r + s) & 3;
                    double pr = (s == 0) ? 1.0 : 0.0, pi = 0.0;
#pragma unroll
                    for (int c = 0; c < MC; c++) {
                        const double wr = quad_rot<s>(xcr[c]), wi = quad_rot<s>(xci[c]);      // X[b][c]
                        pr -= fma(xcr[c], wr, xci[c] * wi);
                        pi -= fma(xci[c], wr, -xcr[c] * wi);
                    }
                    if (r < N && b < N) po[r + b * N] = make_float2((float)pr, (float)pi);
                };
                col(std::integral_constant<int, 0>{});
                col(std::integral_constant<int, 1>{});
                col(std::integral_constant<int, 2>{});
                col(std::integral_constant<int, 3>{});
            }
        }
    }
    // quads the iteration did not certify: the cyclic Jacobi, in this wave (skipped when there is none)
    const bool need = real_item && !fast;
    if (__builtin_amdgcn_ballot_w64(need) != 0ull) {
        if (fallback_count && need && r == 0) atomicAdd(fallback_count, 1ull);
        evd_group_wave<4, double>(Ri, item, need, N, MC, coef, coef_d, PN ? pn_out : nullptr, nullptr, nullptr, cheb_d);
    }
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
// One item on one wave (the body of music_evd_block16_kernel; also the fall-back of the subspace-iteration kernel below).
// sVr / sVi / sLam: 16 x 16 + 16 values of LDS owned by this wave.
template <typename T, bool LEAN>
__device__ __forceinline__ void evd_block16_item(const float2 *__restrict__ Ri, int item, float *__restrict__ coef,
                                                 double *__restrict__ coef_d, float2 *__restrict__ pn_out, int N, int M,
                                                 const float2 *__restrict__ pilot, float2 *__restrict__ cal_out,
                                                 T *__restrict__ sVr, T *__restrict__ sVi, T *__restrict__ sLam)
{
    constexpr int G = 16;
    const int lane = threadIdx.x & (kWave - 1);
    // lane = 16 (a >> 1) + 2 b + (a & 1): the two block rows that share a 16-lane DPP row are interleaved, so that
    // "column block b -> b +- 1" is a DPP row shift by 2 lanes whose out-of-row lanes are exactly b = 0 / b = 7
    const int a = ((lane >> 4) << 1) | (lane & 1), b = (lane & 15) >> 1;
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
    T poison = 0;            // 0, or NaN when the item holds a non-finite entry
    {
        float m = 0.f;
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int j = 0; j < 2; j++) m = fmaxf(m, fmaxf(fabsf((float)xr[i][j]), fabsf((float)xi[i][j])));
        const T sc = jacobi_prescale<T>(wave_allreduce_max(m));
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int j = 0; j < 2; j++) {
                poison = fma(xr[i][j], (T)0, fma(xi[i][j], (T)0, poison));
                xr[i][j] *= sc; xi[i][j] *= sc;
            }
        poison = wave_sum_t<T>(poison);
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
            const JacobiRot<T> rot = jacobi_rotation<T>(xr[0][0], xr[1][1], xr[0][1], xi[0][1]);
            const T c_mine = rot.c, sr_mine = rot.sr, si_mine = rot.si;          // sigma = J[p][q]
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
            sVr[(2 * a + i) * G + 2 * b + j] = vr[i][j] + poison;       // non-finite item -> non-finite outputs
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

template <typename T, bool LEAN>
__global__ __launch_bounds__(64) void music_evd_block16_kernel(const float2 *__restrict__ R, float *__restrict__ coef,
                                                               double *__restrict__ coef_d, float2 *__restrict__ pn_out,
                                                               int n_items, int N, int M,
                                                               const float2 *__restrict__ pilot, float2 *__restrict__ cal_out)
{
    __shared__ T sVr[16 * 16], sVi[16 * 16], sLam[16];
    const int item = blockIdx.x;                         // grid = n_items
    evd_block16_item<T, LEAN>(R + (size_t)item * (N * N), item, coef, coef_d, pn_out, N, M, pilot, cal_out, sVr, sVi, sLam);
}

// One wave per item: the signal-subspace iteration (evd_subspace.hpp) first; whatever it does not certify takes the
// block Jacobi above on the same wave.  G = 8 (N <= 8) / 16, MC = num_targets (1..4); double only.
template <int G, int MC, bool PN>
__global__ __launch_bounds__(64) void music_evd_subspace_kernel(const float2 *__restrict__ R, float *__restrict__ coef,
                                                                double *__restrict__ coef_d, float2 *__restrict__ pn_out,
                                                                int n_items, int N, unsigned long long *__restrict__ fallback_count)
{
    __shared__ double sVr[16 * 16], sVi[16 * 16], sLam[16];
    const int item = blockIdx.x;                         // grid = n_items
    const float2 *Ri = R + (size_t)item * (N * N);
    if (evd_subspace_item<G, MC, PN>(Ri, item, coef, coef_d, pn_out, N, sVr, sVi)) return;
    if (fallback_count && (threadIdx.x & (kWave - 1)) == 0) atomicAdd(fallback_count, 1ull);
    if constexpr (PN) evd_block16_item<double, false>(Ri, item, coef, coef_d, pn_out, N, MC, nullptr, nullptr, sVr, sVi, sLam);
    else evd_block16_item<double, true>(Ri, item, coef, coef_d, nullptr, N, MC, nullptr, nullptr, sVr, sVi, sLam);
}

template <int G, int MC>
static void launch_evd_subspace_gm(int N, int n_items, const void *d_R, void *d_coef, void *d_coef_d, void *d_pn, hipStream_t st,
                                   unsigned long long *d_fallback_count)
{
    if (d_pn)
        hipLaunchKernelGGL((music_evd_subspace_kernel<G, MC, true>), dim3(n_items), dim3(64), 0, st, (const float2 *)d_R,
                           (float *)d_coef, (double *)d_coef_d, (float2 *)d_pn, n_items, N, d_fallback_count);
    else
        hipLaunchKernelGGL((music_evd_subspace_kernel<G, MC, false>), dim3(n_items), dim3(64), 0, st, (const float2 *)d_R,
                           (float *)d_coef, (double *)d_coef_d, nullptr, n_items, N, d_fallback_count);
}
// Diagnostics: items of subspace-kernel launches that took the Jacobi fall-back.  ONE COUNTER PER DEVICE, allocated on first use
// with that device current (the launchers run with the handle's device bound: bind_device): a kernel only ever adds into
// memory of the device it runs on.  (Until round 3 there was one counter per process on whichever device came first; a handle
// on another device would have had its fall-back items -- and only those -- atomicAdd into a foreign allocation: VERDICT r3
// #12.)  64-bit: a long low-SNR run does not wrap.  nullptr (no usable device, allocation failed): the kernels skip the count.
namespace {
constexpr int kMaxCounterDevices = 64;
std::mutex g_fb_mutex;
unsigned long long *g_fb_counter[kMaxCounterDevices] = {};
bool g_fb_failed[kMaxCounterDevices] = {};
}
unsigned long long *evd_fallback_counter()
{
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxCounterDevices) { (void)hipGetLastError(); return nullptr; }
    std::lock_guard<std::mutex> lock(g_fb_mutex);
    if (!g_fb_counter[dev] && !g_fb_failed[dev]) {
        unsigned long long *q = nullptr;
        if (hipMalloc(&q, sizeof(unsigned long long)) != hipSuccess) { (void)hipGetLastError(); g_fb_failed[dev] = true; return nullptr; }
        if (hipMemset(q, 0, sizeof(unsigned long long)) != hipSuccess) { (void)hipGetLastError(); (void)hipFree(q); g_fb_failed[dev] = true; return nullptr; }
        g_fb_counter[dev] = q;
    }
    return g_fb_counter[dev];
}
// the device a counter pointer lives on, or -1 (tests: the launch on a handle's device gets that device's counter)
int evd_fallback_counter_device(const void *p)
{
    if (!p) return -1;
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, p) != hipSuccess) { (void)hipGetLastError(); return -1; }
    return at.device;
}
// sum over the devices that have a counter; every counter is read (and cleared) with its own device bound
long long evd_fallback_count(bool reset)
{
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess) { (void)hipGetLastError(); return -1; }
    unsigned long long *ptrs[kMaxCounterDevices];
    {
        std::lock_guard<std::mutex> lock(g_fb_mutex);
        for (int d = 0; d < kMaxCounterDevices; d++) ptrs[d] = g_fb_counter[d];
    }
    long long total = 0;
    bool moved = false, failed = false;
    for (int d = 0; d < kMaxCounterDevices; d++) {
        if (!ptrs[d]) continue;
        if (d != cur || moved) { if (hipSetDevice(d) != hipSuccess) { failed = true; continue; } moved = true; }
        unsigned long long v = 0;
        if (hipMemcpy(&v, ptrs[d], sizeof v, hipMemcpyDeviceToHost) != hipSuccess) { failed = true; continue; }   // (synchronises with the device)
        if (reset) (void)hipMemset(ptrs[d], 0, sizeof v);
        total += (long long)v;
    }
    if (moved) (void)hipSetDevice(cur);
    if (failed) { (void)hipGetLastError(); return -1; }
    return total;
}

// 4 < N <= 16, 1 <= M <= 4, 2 M <= N (the iteration pays when the signal subspace is the small one)
static bool launch_evd_subspace(int N, int M, int n_items, const void *d_R, void *d_coef, void *d_coef_d, void *d_pn,
                                hipStream_t st, unsigned long long *d_fallback_count = nullptr)
{
    if (N <= 4 || N > 16 || M < 1 || M > 4 || 2 * M > N) return false;
#define DOA_SUB(G_, M_) launch_evd_subspace_gm<G_, M_>(N, n_items, d_R, d_coef, d_coef_d, d_pn, st, d_fallback_count)
    if (N <= 8) { if (M == 1) DOA_SUB(8, 1); else if (M == 2) DOA_SUB(8, 2); else if (M == 3) DOA_SUB(8, 3); else DOA_SUB(8, 4); }
    else { if (M == 1) DOA_SUB(16, 1); else if (M == 2) DOA_SUB(16, 2); else if (M == 3) DOA_SUB(16, 3); else DOA_SUB(16, 4); }
#undef DOA_SUB
    return true;
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

// N <= 4, two or three sources, double: four lanes per item (music_evd_quad_kernel)
static bool launch_evd_quad(int N, int M, int n_items, const void *d_R, void *d_coef, void *d_coef_d, void *d_pn, hipStream_t st,
                            void *d_cheb)
{
    // two sources only: with three (N = 4: ONE noise eigenvalue) 1.6 % of random-direction items fall back and take their
    // waves with them -- 26 against the Jacobi's 11 us per 4096 items (profiles/r04_lab_evd_quad.txt)
    // and four antennas only: at N = 3 the one-lane Jacobi is the faster one by far (5.4 against 14.7 us per 4096 random-direction
    // items: a 3 x 3 sweep is three rotations, and the quad spends a lane on padding; profiles/r04_lab_evd_noise_vector.txt)
    if (N != 4 || M != 2) return false;
    const dim3 grid((n_items + 15) / 16), block(64);
#define DOA_QUAD(M_, PN_)                                                                                                    \
    hipLaunchKernelGGL((music_evd_quad_kernel<M_, PN_>), grid, block, 0, st, (const float2 *)d_R, (float *)d_coef,              \
                       (double *)d_coef_d, (float2 *)d_pn, n_items, N, (double *)d_cheb, evd_fallback_counter())
    if (d_pn) DOA_QUAD(2, true); else DOA_QUAD(2, false);
#undef DOA_QUAD
    return true;
}

template <int N> static void launch_evd_n(int M, int n_items, const void *d_R, void *d_coef, void *d_coef_d, void *d_pn,
                                          int bits, hipStream_t st, void *d_cheb)
{
    dim3 block(64), grid((n_items + 63) / 64);
    if (bits == 32)
        hipLaunchKernelGGL((music_evd_kernel<N, float>), grid, block, 0, st, (const float2 *)d_R, (float *)d_coef,
                           (double *)d_coef_d, (float2 *)d_pn, n_items, M, (double *)nullptr, (unsigned long long *)nullptr);
    else
        hipLaunchKernelGGL((music_evd_kernel<N, double>), grid, block, 0, st, (const float2 *)d_R, (float *)d_coef,
                           (double *)d_coef_d, (float2 *)d_pn, n_items, M, (double *)d_cheb, evd_fallback_counter());
}

int launch_music_evd(int N, int M, int n_items, const void *d_R, void *d_coef, void *d_coef_d, void *d_pn,
                     int evd_bits, hipStream_t st, void *d_cheb)
{
    if (n_items <= 0) return DOA_OK;
    if (N < 2 || N > DOA_MAX_ANT_ELE) {
        set_error("MUSIC: num_ant_ele=%d outside the built range 2..%d", N, DOA_MAX_ANT_ELE);
        return DOA_ERR_UNSUPPORTED;
    }
    // N <= 4: one lane per item, everything in registers (measured 10.5 us vs 11.0 us for the
    // 4-lane group kernel at batch 4096: at this size the cross-lane traffic eats the shorter
    // dependency chain).  N > 4: 8 lanes per item (N <= 8) or one wave per item (block Jacobi), 15x / 17x faster than
    // one lane per item with the matrices in scratch.
    const bool f32 = (evd_bits == 32);
    // wide arrays, few sources, double: signal-subspace iteration with the block Jacobi as its per-item fall-back
    if (!f32 && launch_evd_subspace(N, M, n_items, d_R, d_coef, d_coef_d, d_pn, st, evd_fallback_counter())) {
        DOA_HIP_TRY(hipGetLastError());
        return DOA_OK;
    }
    if (N > 8) {
        if (f32) launch_evd_block16<float>(N, M, n_items, d_R, d_coef, d_coef_d, d_pn, st);
        else launch_evd_block16<double>(N, M, n_items, d_R, d_coef, d_coef_d, d_pn, st);
    } else if (N > 4) {
        if (f32) launch_evd_group<8, float>(N, M, n_items, d_R, d_coef, d_coef_d, d_pn, st);
        else launch_evd_group<8, double>(N, M, n_items, d_R, d_coef, d_coef_d, d_pn, st);
    } else if (!f32 && DOA_LAB_ENV_INT("DOA_EVD_QUAD", 1) && launch_evd_quad(N, M, n_items, d_R, d_coef, d_coef_d, d_pn, st, d_cheb)) {
        // (N <= 4 with M >= 2 in double; M = 1 keeps the one-lane iteration below)
    } else {
        switch (N) {
        case 2: launch_evd_n<2>(M, n_items, d_R, d_coef, d_coef_d, d_pn, evd_bits, st, d_cheb); break;
        case 3: launch_evd_n<3>(M, n_items, d_R, d_coef, d_coef_d, d_pn, evd_bits, st, d_cheb); break;
        default: launch_evd_n<4>(M, n_items, d_R, d_coef, d_coef_d, d_pn, evd_bits, st, d_cheb); break;
        }
    }
    DOA_HIP_TRY(hipGetLastError());
    return DOA_OK;
}

}  // namespace doa
