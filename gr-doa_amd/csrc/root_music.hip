// root_music.hip — K6: Root-MUSIC for a uniform linear array on gfx950.
//
// Replaces gr::doa::rootMUSIC_linear_array (reference lib/rootMUSIC_linear_array_impl.cc:46-152):
//   EVD + noise projector as MUSIC (:108-116; here K2+K3 of music.hip), then
//   u[k], k = 0..2N-2: sums of the diagonals of P_N (:68-79), roots of sum_k u[k] z^k — the
//   reference gets them as eigenvalues of the Frobenius companion matrix through LAPACK cgeev
//   (:80-86) — keep the roots strictly inside the unit circle (:122-127), take the num_targets
//   closest to it (:131-141), angle = acos(arg(z) / (2 pi d)) in degrees, sorted ascending (:144).
//
// Here the roots come from a batched Aberth-Ehrlich iteration in double, one root per lane (the
// polynomial has degree 2N-2 <= 30; root_music_group_kernel).  Root-MUSIC's root pairs
// (z, 1/conj(z)) sit within ~1e-4 of each other at the unit circle, which is why a float solver
// (the reference's cgeev) is itself only ~0.01 degree accurate; see DESIGN.md "parity".
// Edge cases follow the reference: with fewer than num_targets interior roots the missing slots
// come out as 90 degrees (arg(inf+0i) = 0, :135-141); with none, the reference raises inside
// Armadillo (index_min on an empty vector) — here the item's status word is set and its angles
// are NaN, and the host entry point returns DOA_ERR_NUMERIC.
#include "kernels.hpp"

#include <cstdlib>

namespace doa {

// |p(z)| <= kResidualFloor * (degree+1) * sum |c_m||z|^m counts as "zero to working precision" (16 eps)
constexpr double kResidualFloor = 16.0 * 2.220446049250313e-16;

// ---------------------------------------------------------------------------------------------
// Group-parallel Aberth-Ehrlich: GR lanes (the power of two >= 2N-2) share one item, one root per
// lane.  Every lane evaluates p and p' at its own root by Horner (the 2N-1 coefficients are held
// redundantly in registers, padded with zeros up to GR so the loop bounds are compile-time), reads
// the other roots with cross-lane fetches for the Aberth correction, and all roots move
// simultaneously.  The dependent chain per item is ~1/(2N-2) of the one-lane-per-item kernel's.
// ---------------------------------------------------------------------------------------------
template <int GR> __device__ __forceinline__ double group_max_d(double v, int lane)
{
#pragma unroll
    for (int m = 1; m < GR; m <<= 1) v = fmax(v, __shfl(v, lane ^ m, kWave));
    return v;
}

// 1/x for a finite positive double well inside the exponent range: hardware seed + two Newton steps
// (what the compiler's IEEE division expands to, minus its scaling / fix-up instructions: ~6 instead
// of ~18 instructions, and one reciprocal serves both parts of a complex quotient)
__device__ __forceinline__ double fast_rcp(double x)
{
    double y = __builtin_amdgcn_rcp(x);
    double e = fma(-x, y, 1.0);
    y = fma(y, e, y);
    e = fma(-x, y, 1.0);
    return fma(y, e, y);
}

// Selection stage (reference lib/rootMUSIC_linear_array_impl.cc:122-145) on the GR lanes of one group, lane k holding
// root k (is_root: k < 2N-2): dist = 1 - |z| (:122), keep dist > 0 -- strictly inside (:125-127) --, num_targets times the
// interior root closest to the circle (index_min: ties -> the first, i.e. the lowest root index), angle =
// 180 acos(arg z / (2 pi d)) / pi (:135), a used root becomes inf + 0i with dist inf (:138-139) -- so once the interior
// roots are exhausted index_min lands on an inf entry, arg(inf + 0i) = 0 and the slot reads 90 degrees --, ascending
// sort (:144; NaN last).  No interior root at all: the reference raises inside arma::index_min; here NaN angles and
// status 1.  Shared by the solver kernel and by root_select_kernel (the selection on caller-supplied roots).
template <int GR>
__device__ __forceinline__ void root_select(double zr, double zi, bool is_root, int k, int base, int lane, int item,
                                            bool real_item, int M, double two_pi_d, float *__restrict__ out,
                                            int *__restrict__ status)
{
    // dist = 1 - |z|; keep dist > 0; the M smallest, one at a time (:122-141)
    double dist = is_root ? 1.0 - sqrt(zr * zr + zi * zi) : -1.0;
    if (!(dist > 0.0)) dist = -1.0;
    const unsigned long long inside_mask = __ballot(dist > 0.0);
    const int n_inside = __popcll((inside_mask >> base) & ((GR == 64) ? ~0ull : ((1ull << GR) - 1ull)));
    float my_aoa = 0.f;
    for (int j = 0; j < M; j++) {
        // group arg-min of dist over the remaining interior roots (ties -> lowest lane)
        double bd = (dist > 0.0) ? dist : 1e300;
        int bk = (dist > 0.0) ? k : GR;
#pragma unroll
        for (int m = 1; m < GR; m <<= 1) {
            const double od = __shfl(bd, lane ^ m, kWave);
            const int ok = __shfl(bk, lane ^ m, kWave);
            if (od < bd || (od == bd && ok < bk)) { bd = od; bk = ok; }
        }
        double ang = 0.0;                                   // exhausted: arg(inf + 0i) = 0 -> 90 degrees
        if (bk < GR) {
            const double br = __shfl(zr, base + bk, kWave), bi = __shfl(zi, base + bk, kWave);
            ang = atan2(bi, br);
            if (k == bk) dist = -1.0;
        }
        const float a = (float)(180.0 * acos(ang / two_pi_d) / M_PI);
        if (k == j) my_aoa = a;
    }
    // ascending sort of the M picks held by lanes 0..M-1 (NaN sorts last)
    const float key = (my_aoa != my_aoa) ? INFINITY : my_aoa;
    int rank = 0;
    for (int j = 0; j < M; j++) {
        const float kj = __shfl(key, base + j, kWave);
        rank += (kj < key || (kj == key && j < k)) ? 1 : 0;
    }
    if (real_item) {
        float *o = out + (size_t)item * M;
        if (n_inside == 0) {
            if (k < M) o[k] = __builtin_nanf("");
            if (k == 0 && status) status[item] = 1;
        } else {
            if (k < M) o[rank] = my_aoa;
            if (k == 0 && status) status[item] = 0;
        }
    }
}

// DEG = compile-time polynomial degree when it is known to be below GR (N = 4: degree 6 on 8 lanes),
// so that neither the Horner recurrence nor the root-pair loop spends steps on padding.
template <int GR, int DEG = GR, bool FLOAT_PHASE = true>
__global__ __launch_bounds__(64) void root_music_group_kernel(const double *__restrict__ coef, float *__restrict__ out,
                                                              int *__restrict__ status, int n_items, int N, int M,
                                                              double two_pi_d, double2 *__restrict__ roots_out,
                                                              int float_iters, float float_tol2)
{
    constexpr int IPW = kWave / GR;
    const int lane = threadIdx.x & (kWave - 1);
    const int k = lane % GR, base = lane - k;
    int item = blockIdx.x * IPW + lane / GR;
    const bool real_item = item < n_items;
    if (!real_item) item = n_items - 1;
    const int D = 2 * N - 2;
    const bool is_root = k < D;
    const double *co = coef + (size_t)item * (2 * N);
    // polynomial c[m], m = 0..D: c[N-1-l] = u_l, c[N-1+l] = conj(u_l); zero above D
    double cr[GR + 1], ci[GR + 1], ca[GR + 1];
#pragma unroll
    for (int m = 0; m <= GR; m++) {
        double vr = 0.0, vi = 0.0;
        if (m <= D) {
            const int l = (m <= N - 1) ? (N - 1 - m) : (m - (N - 1));
            if (l == 0) vr = co[0];
            else { vr = co[2 * l - 1]; vi = (m < N - 1) ? co[2 * l] : -co[2 * l]; }
        }
        cr[m] = vr; ci[m] = vi; ca[m] = fabs(vr) + fabs(vi);
    }
    double zr, zi;
    {
        const double ang = 2.0 * M_PI * (k + 0.37) / D;
        const double rad = 0.75 + 0.5 * ((k * 0.6180339887498949) - floor(k * 0.6180339887498949));
        double sn, cs;
        sincos(ang, &sn, &cs);
        zr = is_root ? rad * cs : 1e6 * (k + 1);          // idle lanes park far away and never move
        zi = is_root ? rad * sn : 0.0;
    }
    if constexpr (FLOAT_PHASE) {
        // Round 4: the first iterations in FLOAT.  This kernel is one dependent chain per wave (512 waves on 1024 SIMDs), so
        // what an iteration costs is the latency of its instructions, and most of the 10-13 iterations only carry the roots from
        // the generic starting points into their basins: the same Aberth step on float copies of the coefficients (one
        // v_rcp_f32 per quotient instead of a seed and two Newton steps in double) until the steps are at float's own floor --
        // relative 1e-5, or 12 iterations --, then the double iteration below with its residual-based stopping test, unchanged,
        // which needs two or three more steps from there (cubic convergence).  What comes out has passed exactly the test it
        // passed before; the roots of a pair z, 1/conj(z), 1e-3..1e-5 apart, are told apart by float well enough to start from.
        float fzr = (float)zr, fzi = (float)zi;
        float fcr[DEG + 1], fci[DEG + 1];
        {
            // scale by the largest coefficient: float has the range, this keeps p and p' O(1)
            double mx = 0.0;
#pragma unroll
            for (int m = 0; m <= DEG; m++) mx = fmax(mx, ca[m]);
            const double inv = (mx > 0.0 && mx < 1e300) ? 1.0 / mx : 1.0;
#pragma unroll
            for (int m = 0; m <= DEG; m++) { fcr[m] = (float)(cr[m] * inv); fci[m] = (float)(ci[m] * inv); }
        }
        for (int it = 0; it < float_iters; it++) {
            float pr = 0.f, pi = 0.f, dr = 0.f, di = 0.f;
#pragma unroll
            for (int m = DEG; m >= 0; m--) {
                const float ndr = fmaf(dr, fzr, fmaf(-di, fzi, pr));
                const float ndi = fmaf(dr, fzi, fmaf(di, fzr, pi));
                dr = ndr; di = ndi;
                const float npr = fmaf(pr, fzr, fmaf(-pi, fzi, fcr[m]));
                const float npi = fmaf(pr, fzi, fmaf(pi, fzr, fci[m]));
                pr = npr; pi = npi;
            }
            const float dn = fmaf(dr, dr, di * di);
            float wr = 1e-3f, wi = 1e-3f;
            if (dn > 0.f) { const float inv = __builtin_amdgcn_rcpf(dn); wr = (pr * dr + pi * di) * inv; wi = (pi * dr - pr * di) * inv; }
            float sr = 0.f, si = 0.f;
#pragma unroll
            for (int j = 0; j < DEG; j++) {
                const float ojr = __shfl(fzr, base + j, kWave), oji = __shfl(fzi, base + j, kWave);
                const float er = fzr - ojr, ei = fzi - oji;
                const float en = fmaf(er, er, ei * ei);
                if (j != k && j < D && en > 0.f) { const float inv = __builtin_amdgcn_rcpf(en); sr = fmaf(er, inv, sr); si = fmaf(-ei, inv, si); }
            }
            const float qr = 1.f - (wr * sr - wi * si), qi = -(wr * si + wi * sr);
            const float qn = fmaf(qr, qr, qi * qi);
            float er = wr, ei = wi;
            if (qn > 0.f) { const float inv = __builtin_amdgcn_rcpf(qn); er = (wr * qr + wi * qi) * inv; ei = (wi * qr - wr * qi) * inv; }
            float rel = 0.f;
            const bool finite_step = (er == er) && (ei == ei) && (fabsf(er) < 1e30f) && (fabsf(ei) < 1e30f);
            if (is_root && finite_step) {
                fzr -= er; fzi -= ei;
                rel = (er * er + ei * ei) / (1.f + fzr * fzr + fzi * fzi);
            }
            if (__ballot(rel >= float_tol2) == 0ull) break;  // (squared relative step)
        }
        // a float phase that went astray (overflow, NaN) hands back the generic starting point
        const bool sane = (fzr == fzr) && (fzi == fzi) && (fabsf(fzr) < 1e6f) && (fabsf(fzi) < 1e6f);
        if (is_root && sane) { zr = (double)fzr; zi = (double)fzi; }
        // two lanes on the SAME float (the two roots of a pair, unresolved): the repulsion term between them would be
        // skipped (distance 0) and they would travel together for good; the later one steps aside by 1e-4
        bool twin = false;
#pragma unroll
        for (int j = 0; j < DEG; j++) {
            const double ojr = __shfl(zr, base + j, kWave), oji = __shfl(zi, base + j, kWave);
            twin = twin || (j < k && j < D && ojr == zr && oji == zi);
        }
        if (is_root && twin) { zr += 1e-4 * (k + 1); zi -= 0.7e-4 * (k + 1); }
    }
    for (int it = 0; it < 80; it++) {
        double pr = 0.0, pi = 0.0, dr = 0.0, di = 0.0, eb = 0.0;
        const double rk = (double)sqrtf((float)(zr * zr + zi * zi));
#pragma unroll
        for (int m = DEG; m >= 0; m--) {
            const double ndr = dr * zr - di * zi + pr;
            const double ndi = dr * zi + di * zr + pi;
            dr = ndr; di = ndi;
            const double npr = pr * zr - pi * zi + cr[m];
            const double npi = pr * zi + pi * zr + ci[m];
            pr = npr; pi = npi;
            eb = fma(eb, rk, ca[m]);             // rounding-error bound of p(z): sum |c_m| |z|^m
        }
        const double floor_k = kResidualFloor * (D + 1) * eb;
        const bool at_floor = (pr * pr + pi * pi <= floor_k * floor_k);
        // (coefficients above degree D are zero, so starting the recurrence at DEG >= D changes nothing)
        const double dn = dr * dr + di * di;
        double wr = 1e-3, wi = 1e-3;
        if (dn > 0.0) { const double inv = fast_rcp(dn); wr = (pr * dr + pi * di) * inv; wi = (pi * dr - pr * di) * inv; }
        double sr = 0.0, si = 0.0;
#pragma unroll
        for (int j = 0; j < DEG; j++) {
            const double ojr = __shfl(zr, base + j, kWave), oji = __shfl(zi, base + j, kWave);
            const double er = zr - ojr, ei = zi - oji;
            const double en = er * er + ei * ei;
            if (j != k && j < D && en > 0.0) { const double inv = fast_rcp(en); sr = fma(er, inv, sr); si = fma(-ei, inv, si); }
        }
        const double qr = 1.0 - (wr * sr - wi * si), qi = -(wr * si + wi * sr);
        const double qn = qr * qr + qi * qi;
        double er = wr, ei = wi;
        if (qn > 0.0) { const double inv = fast_rcp(qn); er = (wr * qr + wi * qi) * inv; ei = (wi * qr - wr * qi) * inv; }
        double rel = 0.0;
        if (is_root && !at_floor) {
            zr -= er; zi -= ei;
            rel = (er * er + ei * ei) / (1.0 + zr * zr + zi * zi);
        }
        if (__ballot(rel >= 1e-29) == 0ull) break;   // every root of every group of the wave has converged
    }
    {
        // a non-finite polynomial (non-finite covariance item) has no roots: without this the iteration above stops at
        // once -- every comparison with NaN is false -- and the starting points would pass for roots
        double csum = 0.0;
#pragma unroll
        for (int m = 0; m <= GR; m++) csum += ca[m];
        if (!(csum < INFINITY)) { zr = __builtin_nan(""); zi = __builtin_nan(""); }
    }
    if (roots_out && real_item && is_root) roots_out[(size_t)item * D + k] = make_double2(zr, zi);    // diagnostics
    root_select<GR>(zr, zi, is_root, k, base, lane, item, real_item, M, two_pi_d, out, status);
}

template <int GR, int DEG = GR>
static void launch_root_group(int N, int M, int n_items, const void *d_coef, void *d_out, void *d_status, double two_pi_d,
                              hipStream_t st, void *d_roots)
{
    constexpr int IPW = kWave / GR;
    dim3 block(64), grid((n_items + IPW - 1) / IPW);
    // float phase: at most kFloatIters iterations, left as soon as every root of the wave moves by less than 1e-3 relative
    // (profiles/r04_lab_root_float_phase.txt has the sweep behind the two numbers)
    const int float_iters = DOA_LAB_ENV_INT("DOA_ROOT_FLOAT_ITERS", 12);
    const float float_tol2 = exp2f(-(float)DOA_LAB_ENV_INT("DOA_ROOT_FLOAT_TOL_LOG2", 33));      // 2^-33 = (1.1e-5)^2
    if (DOA_LAB_ENV_INT("DOA_ROOT_FLOAT_PHASE", 1))
        hipLaunchKernelGGL((root_music_group_kernel<GR, DEG, true>), grid, block, 0, st, (const double *)d_coef, (float *)d_out,
                           (int *)d_status, n_items, N, M, two_pi_d, (double2 *)d_roots, float_iters, float_tol2);
#ifdef DOA_LAB
    else
        hipLaunchKernelGGL((root_music_group_kernel<GR, DEG, false>), grid, block, 0, st, (const double *)d_coef, (float *)d_out,
                           (int *)d_status, n_items, N, M, two_pi_d, (double2 *)d_roots, 0, 0.f);
#endif
}

int launch_root_music(int N, int M, float norm_spacing, int n_items, const void *d_coef, void *d_out, void *d_status,
                      hipStream_t st, void *d_roots)
{
    if (n_items <= 0) return DOA_OK;
    const double two_pi_d = 2 * M_PI * (double)norm_spacing;   // 2*datum::pi*d_norm_spacing, float promoted (:135)
    if (N < 2 || N > DOA_MAX_ANT_ELE) {
        set_error("rootMUSIC: num_ant_ele=%d outside the built range 2..%d", N, DOA_MAX_ANT_ELE);
        return DOA_ERR_UNSUPPORTED;
    }
    const int D = 2 * N - 2;
    if (D <= 2) launch_root_group<2>(N, M, n_items, d_coef, d_out, d_status, two_pi_d, st, d_roots);
    else if (D <= 4) launch_root_group<4>(N, M, n_items, d_coef, d_out, d_status, two_pi_d, st, d_roots);
    else if (D == 6) launch_root_group<8, 6>(N, M, n_items, d_coef, d_out, d_status, two_pi_d, st, d_roots);
    else if (D <= 8) launch_root_group<8>(N, M, n_items, d_coef, d_out, d_status, two_pi_d, st, d_roots);
    else if (D <= 16) launch_root_group<16>(N, M, n_items, d_coef, d_out, d_status, two_pi_d, st, d_roots);
    else launch_root_group<32>(N, M, n_items, d_coef, d_out, d_status, two_pi_d, st, d_roots);
    DOA_HIP_TRY(hipGetLastError());
    return DOA_OK;
}

// The selection stage alone, on caller-supplied roots (diagnostics: doa_rootMUSIC_linear_array_select_debug): the same
// root_select<GR> instantiation the solver kernel of this array size ends in, fed from memory instead of from the
// Aberth iteration, so every branch of the rule can be driven with hand-made root lists.
template <int GR>
__global__ __launch_bounds__(64) void root_select_kernel(const double2 *__restrict__ roots, float *__restrict__ out,
                                                         int *__restrict__ status, int n_items, int N, int M, double two_pi_d)
{
    constexpr int IPW = kWave / GR;
    const int lane = threadIdx.x & (kWave - 1);
    const int k = lane % GR, base = lane - k;
    int item = blockIdx.x * IPW + lane / GR;
    const bool real_item = item < n_items;
    if (!real_item) item = n_items - 1;
    const int D = 2 * N - 2;
    const bool is_root = k < D;
    double zr = 0.0, zi = 0.0;
    if (is_root) { const double2 z = roots[(size_t)item * D + k]; zr = z.x; zi = z.y; }
    root_select<GR>(zr, zi, is_root, k, base, lane, item, real_item, M, two_pi_d, out, status);
}

int launch_root_select(int N, int M, float norm_spacing, int n_items, const void *d_roots, void *d_out, void *d_status,
                       hipStream_t st)
{
    if (n_items <= 0) return DOA_OK;
    const double two_pi_d = 2 * M_PI * (double)norm_spacing;
    const int D = 2 * N - 2;
#define DOA_SELECT(GR_)                                                                                             \
    hipLaunchKernelGGL((root_select_kernel<GR_>), dim3((n_items + kWave / GR_ - 1) / (kWave / GR_)), dim3(64), 0, st, \
                       (const double2 *)d_roots, (float *)d_out, (int *)d_status, n_items, N, M, two_pi_d)
    if (D <= 2) DOA_SELECT(2);
    else if (D <= 4) DOA_SELECT(4);
    else if (D <= 8) DOA_SELECT(8);
    else if (D <= 16) DOA_SELECT(16);
    else DOA_SELECT(32);
#undef DOA_SELECT
    DOA_HIP_TRY(hipGetLastError());
    return DOA_OK;
}

}  // namespace doa

// ---------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------
struct doa_rootMUSIC_linear_array {
    float norm_spacing = 0.f;
    int M = 0, N = 0;
    int bits = 64;
    int device = 0;
    hipStream_t stream = nullptr;
    doa::DevBuf d_in, d_out, d_coef, d_status, d_roots;
    doa::PinnedBuf h_status;
};

extern "C" {

doa_rootMUSIC_linear_array_t *doa_rootMUSIC_linear_array_create(float norm_spacing, int num_targets, int num_ant_ele)
{
    doa::clear_error();
    // grc/doa_rootMUSIC_linear_array.xml:28-30
    if (num_ant_ele <= 0 || num_targets <= 0 || num_targets >= num_ant_ele) {
        doa::set_error("rootMUSIC_linear_array: need 0 < num_targets < num_ant_ele (got %d, %d)", num_targets, num_ant_ele);
        return nullptr;
    }
    if (!(norm_spacing > 0.0f) || norm_spacing > 0.5f) {
        doa::set_error("rootMUSIC_linear_array: need 0 < norm_spacing <= 0.5 (got %g)", (double)norm_spacing);
        return nullptr;
    }
    if (num_ant_ele > DOA_MAX_ANT_ELE || num_targets > DOA_MAX_PEAKS) {
        doa::set_error("rootMUSIC_linear_array: num_ant_ele=%d / num_targets=%d exceed the built caps (%d / %d)",
                       num_ant_ele, num_targets, DOA_MAX_ANT_ELE, DOA_MAX_PEAKS);
        return nullptr;
    }
    int dev = 0;
    if (doa::ensure_device(&dev) != DOA_OK) return nullptr;
    auto *h = new (std::nothrow) doa_rootMUSIC_linear_array();
    if (!h) { doa::set_error("out of memory"); return nullptr; }
    h->norm_spacing = norm_spacing; h->M = num_targets; h->N = num_ant_ele; h->device = dev;
    h->bits = doa::internal_precision_bits();
    if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) {
        doa::set_error("rootMUSIC_linear_array: hipStreamCreate failed");
        delete h;
        return nullptr;
    }
    return h;
}

void doa_rootMUSIC_linear_array_destroy(doa_rootMUSIC_linear_array_t *h)
{
    if (!h) return;
    h->d_in.release(); h->d_out.release(); h->d_coef.release(); h->d_status.release(); h->d_roots.release();
    h->h_status.release();
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

int doa_rootMUSIC_linear_array_work_dev(doa_rootMUSIC_linear_array_t *h, int noutput_items, const void *d_input_items0,
                                        void *d_output_items0, void *hip_stream)
{
    doa::clear_error();
    if (!h || noutput_items < 0 || (noutput_items > 0 && (!d_input_items0 || !d_output_items0))) {
        doa::set_error("rootMUSIC_linear_array_work_dev: bad arguments");
        return DOA_ERR_INVALID_ARG;
    }
    if (noutput_items == 0) return 0;
    if (int brc = doa::bind_device(h->device); brc != DOA_OK) return brc;
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    int rc = h->d_coef.reserve((size_t)noutput_items * doa::coef_stride(h->N) * sizeof(double));
    if (rc == DOA_OK) rc = h->d_status.reserve((size_t)noutput_items * sizeof(int));
    if (rc != DOA_OK) return rc;
    rc = doa::launch_music_evd(h->N, h->M, noutput_items, d_input_items0, nullptr, h->d_coef.p, nullptr, h->bits, st);
    if (rc != DOA_OK) return rc;
    rc = doa::launch_root_music(h->N, h->M, h->norm_spacing, noutput_items, h->d_coef.p, d_output_items0,
                                h->d_status.p, st);
    return rc == DOA_OK ? noutput_items : rc;
}

int doa_rootMUSIC_linear_array_work(doa_rootMUSIC_linear_array_t *h, int noutput_items, const void *input_items0,
                                    void *output_items0)
{
    doa::clear_error();
    if (!h || noutput_items < 0 || (noutput_items > 0 && (!input_items0 || !output_items0))) {
        doa::set_error("rootMUSIC_linear_array_work: bad arguments");
        return DOA_ERR_INVALID_ARG;
    }
    if (noutput_items == 0) return 0;
    if (int brc = doa::bind_device(h->device); brc != DOA_OK) return brc;
    const size_t in_bytes = (size_t)noutput_items * h->N * h->N * sizeof(float2);
    const size_t out_bytes = (size_t)noutput_items * h->M * sizeof(float);
    int rc = h->d_in.reserve(in_bytes);
    if (rc == DOA_OK) rc = h->d_out.reserve(out_bytes);
    if (rc == DOA_OK) rc = h->h_status.reserve((size_t)noutput_items * sizeof(int));
    if (rc != DOA_OK) return rc;
    DOA_HIP_TRY(hipMemcpyAsync(h->d_in.p, input_items0, in_bytes, hipMemcpyHostToDevice, h->stream));
    rc = doa_rootMUSIC_linear_array_work_dev(h, noutput_items, h->d_in.p, h->d_out.p, h->stream);
    if (rc < 0) return rc;
    DOA_HIP_TRY(hipMemcpyAsync(output_items0, h->d_out.p, out_bytes, hipMemcpyDeviceToHost, h->stream));
    DOA_HIP_TRY(hipMemcpyAsync(h->h_status.p, h->d_status.p, (size_t)noutput_items * sizeof(int),
                               hipMemcpyDeviceToHost, h->stream));
    DOA_HIP_TRY(hipStreamSynchronize(h->stream));
    const int *stt = h->h_status.as<int>();
    for (int i = 0; i < noutput_items; i++)
        if (stt[i] != 0) {
            doa::set_error("rootMUSIC_linear_array: item %d has no root strictly inside the unit circle "
                           "(the reference raises in arma::index_min here)", i);
            return DOA_ERR_NUMERIC;
        }
    return noutput_items;
}

int doa_rootMUSIC_linear_array_debug(doa_rootMUSIC_linear_array_t *h, int noutput_items, const void *input_items0,
                                     void *output_items0, void *roots_out, int *status_out)
{
    doa::clear_error();
    if (!h || noutput_items <= 0 || !input_items0 || !output_items0) {
        doa::set_error("rootMUSIC_linear_array_debug: bad arguments");
        return DOA_ERR_INVALID_ARG;
    }
    if (int brc = doa::bind_device(h->device); brc != DOA_OK) return brc;
    const int D = 2 * h->N - 2;
    const size_t in_bytes = (size_t)noutput_items * h->N * h->N * sizeof(float2);
    const size_t out_bytes = (size_t)noutput_items * h->M * sizeof(float);
    const size_t root_bytes = (size_t)noutput_items * D * sizeof(double2);
    int rc = h->d_in.reserve(in_bytes);
    if (rc == DOA_OK) rc = h->d_out.reserve(out_bytes);
    if (rc == DOA_OK) rc = h->d_roots.reserve(root_bytes);
    if (rc == DOA_OK) rc = h->d_coef.reserve((size_t)noutput_items * doa::coef_stride(h->N) * sizeof(double));
    if (rc == DOA_OK) rc = h->d_status.reserve((size_t)noutput_items * sizeof(int));
    if (rc != DOA_OK) return rc;
    DOA_HIP_TRY(hipMemcpyAsync(h->d_in.p, input_items0, in_bytes, hipMemcpyHostToDevice, h->stream));
    rc = doa::launch_music_evd(h->N, h->M, noutput_items, h->d_in.p, nullptr, h->d_coef.p, nullptr, h->bits, h->stream);
    if (rc == DOA_OK)
        rc = doa::launch_root_music(h->N, h->M, h->norm_spacing, noutput_items, h->d_coef.p, h->d_out.p, h->d_status.p,
                                    h->stream, h->d_roots.p);
    if (rc != DOA_OK) return rc;
    DOA_HIP_TRY(hipMemcpyAsync(output_items0, h->d_out.p, out_bytes, hipMemcpyDeviceToHost, h->stream));
    if (roots_out) DOA_HIP_TRY(hipMemcpyAsync(roots_out, h->d_roots.p, root_bytes, hipMemcpyDeviceToHost, h->stream));
    if (status_out)
        DOA_HIP_TRY(hipMemcpyAsync(status_out, h->d_status.p, (size_t)noutput_items * sizeof(int), hipMemcpyDeviceToHost,
                                   h->stream));
    DOA_HIP_TRY(hipStreamSynchronize(h->stream));
    return noutput_items;
}

int doa_rootMUSIC_linear_array_select_debug(doa_rootMUSIC_linear_array_t *h, int noutput_items, const void *roots_in,
                                            void *output_items0, int *status_out)
{
    doa::clear_error();
    if (!h || noutput_items <= 0 || !roots_in || !output_items0) {
        doa::set_error("rootMUSIC_linear_array_select_debug: bad arguments");
        return DOA_ERR_INVALID_ARG;
    }
    if (int brc = doa::bind_device(h->device); brc != DOA_OK) return brc;
    const int D = 2 * h->N - 2;
    const size_t out_bytes = (size_t)noutput_items * h->M * sizeof(float);
    const size_t root_bytes = (size_t)noutput_items * D * sizeof(double2);
    int rc = h->d_out.reserve(out_bytes);
    if (rc == DOA_OK) rc = h->d_roots.reserve(root_bytes);
    if (rc == DOA_OK) rc = h->d_status.reserve((size_t)noutput_items * sizeof(int));
    if (rc != DOA_OK) return rc;
    DOA_HIP_TRY(hipMemcpyAsync(h->d_roots.p, roots_in, root_bytes, hipMemcpyHostToDevice, h->stream));
    rc = doa::launch_root_select(h->N, h->M, h->norm_spacing, noutput_items, h->d_roots.p, h->d_out.p, h->d_status.p, h->stream);
    if (rc != DOA_OK) { (void)hipStreamSynchronize(h->stream); return rc; }
    DOA_HIP_TRY(hipMemcpyAsync(output_items0, h->d_out.p, out_bytes, hipMemcpyDeviceToHost, h->stream));
    if (status_out)
        DOA_HIP_TRY(hipMemcpyAsync(status_out, h->d_status.p, (size_t)noutput_items * sizeof(int), hipMemcpyDeviceToHost,
                                   h->stream));
    DOA_HIP_TRY(hipStreamSynchronize(h->stream));
    return noutput_items;
}

int doa_rootMUSIC_linear_array_set_internal_precision(doa_rootMUSIC_linear_array_t *h, int bits)
{
    doa::clear_error();
    if (!h || (bits != 32 && bits != 64)) { doa::set_error("rootMUSIC_linear_array_set_internal_precision: need a handle and bits = 32 or 64"); return DOA_ERR_INVALID_ARG; }
    h->bits = bits;
    return DOA_OK;
}

}  // extern "C"
