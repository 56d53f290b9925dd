// autocorrelate.hip — K1: sliding-window sample covariance of N complex streams on gfx950.
//
// Replaces gr::doa::autocorrelate (reference lib/autocorrelate_impl.cc:47-118):
//   R_i[a,b] = (1/K) * sum_{t<K} x_a[i*S + t] * conj(x_b[i*S + t]),   S = K - overlap,
//   optional forward-backward step R <- 0.5 R + (0.5/K) J conj(R) J   (:107-108, the second
//   term is divided by K a second time in the reference; reproduced).
// Output items are column-major N x N complex64 (:103).
//
// Kernel shape (HBM-bound, ~1 flop/B): one 64-lane wave owns one snapshot.  Each lane streams
// 16-byte (two-sample) loads from every channel — a wave instruction covers 1 KiB contiguous per
// channel — and keeps the Hermitian upper triangle (N real + N(N-1)/2 complex partial sums) in
// registers; partials meet in one DPP wave all-reduce and lanes 0..N^2-1 write the 8N^2-byte item
// as one contiguous segment.  No LDS, no atomics.  Overlapping windows take a read-once two-kernel
// path (cov_piece_kernel + cov_combine_kernel); 8 < N <= 16 runs on the matrix cores (cov_mfma_kernel).
#include "common.hpp"

#include <cstdlib>

namespace doa {

struct CovArgs {
    const float2 *in[DOA_MAX_ANT_ELE];
    float2 *out;
    int n_ch;      // N
    int K;         // snapshot_size
    int S;         // snapshot_size - overlap_size
    int n_out;     // windows to produce
    int avg;       // 1 = forward-backward
    float inv_k;   // (float)(1.0/K)
    float fb_hk;   // (float)(0.5/K)
    const float2 *gain;   // optional [N*N] g_a conj(g_b) (fused antenna_correction), or nullptr
    // read-once path for overlapping windows (cov_piece_kernel + cov_combine_kernel)
    float2 *pieces;       // [n_steps][2][N*N] raw sums of the pieces A_j, B_j
    int q, r;             // K = q*S + r
    int n_steps;          // n_out + q
};

template <int TN> struct TriAcc {
    float d[TN];
    float re[TN * (TN - 1) / 2 > 0 ? TN * (TN - 1) / 2 : 1];
    float im[TN * (TN - 1) / 2 > 0 ? TN * (TN - 1) / 2 : 1];
};

template <int TN> __device__ __forceinline__ void tri_accumulate(TriAcc<TN> &acc, const float2 (&x)[TN])
{
    int idx = 0;
#pragma unroll
    for (int a = 0; a < TN; a++) {
        acc.d[a] = fmaf(x[a].x, x[a].x, fmaf(x[a].y, x[a].y, acc.d[a]));
#pragma unroll
        for (int b = a + 1; b < TN; b++) {
            // x_a * conj(x_b)
            acc.re[idx] = fmaf(x[a].x, x[b].x, fmaf(x[a].y, x[b].y, acc.re[idx]));
            acc.im[idx] = fmaf(x[a].y, x[b].x, fmaf(-x[a].x, x[b].y, acc.im[idx]));
            idx++;
        }
    }
}

// One wave per snapshot, N = TN <= 8, Hermitian symmetry exploited.
// VEC2: all streams 16-B aligned at every window start (base % 16 == 0, S even) -> float4 loads.
template <int TN, bool VEC2, int UN, bool NT = false>
__global__ __launch_bounds__(256) void cov_wave_kernel(CovArgs g)
{
    const int lane = threadIdx.x & (kWave - 1);
    const int wave0 = blockIdx.x * (blockDim.x / kWave) + (threadIdx.x / kWave);
    const int n_waves = gridDim.x * (blockDim.x / kWave);
    // grid-stride over snapshots: the launch may use fewer waves than snapshots so that other kernels
    // (the EVD / scan of the previous batch on another stream) find free wave slots on every CU
    for (int snap = wave0; snap < g.n_out; snap += n_waves) {
    const size_t base = (size_t)snap * (size_t)g.S;

    TriAcc<TN> acc;
#pragma unroll
    for (int a = 0; a < TN; a++) acc.d[a] = 0.f;
#pragma unroll
    for (int i = 0; i < TN * (TN - 1) / 2; i++) { acc.re[i] = 0.f; acc.im[i] = 0.f; }

    if constexpr (VEC2) {
        const int npair = g.K >> 1;
        // UN wave-iterations (UN*128 samples) per trip: all UN*TN 16-byte loads are issued before the
        // first one is consumed, so each wave keeps UN*TN KiB in flight (HBM latency x bandwidth needs
        // ~50 KiB per CU; 16 waves x 16 KiB gives 5x that while staying under 96 VGPRs).
        int p = lane;
        for (; p + (UN - 1) * kWave < npair; p += UN * kWave) {
            float4 v[UN][TN];
#pragma unroll
            for (int u = 0; u < UN; u++)
#pragma unroll
                for (int a = 0; a < TN; a++)
                    v[u][a] = load_f4<NT>(reinterpret_cast<const float4 *>(g.in[a] + base + 2 * (size_t)(p + u * kWave)));
#pragma unroll
            for (int u = 0; u < UN; u++) {
                float2 x0[TN], x1[TN];
#pragma unroll
                for (int a = 0; a < TN; a++) { x0[a] = make_float2(v[u][a].x, v[u][a].y); x1[a] = make_float2(v[u][a].z, v[u][a].w); }
                tri_accumulate<TN>(acc, x0);
                tri_accumulate<TN>(acc, x1);
            }
        }
        for (; p < npair; p += kWave) {
            float4 v[TN];
#pragma unroll
            for (int a = 0; a < TN; a++) v[a] = load_f4<NT>(reinterpret_cast<const float4 *>(g.in[a] + base + 2 * (size_t)p));
            float2 x0[TN], x1[TN];
#pragma unroll
            for (int a = 0; a < TN; a++) { x0[a] = make_float2(v[a].x, v[a].y); x1[a] = make_float2(v[a].z, v[a].w); }
            tri_accumulate<TN>(acc, x0);
            tri_accumulate<TN>(acc, x1);
        }
        if ((g.K & 1) && lane == 0) {
            float2 x[TN];
#pragma unroll
            for (int a = 0; a < TN; a++) x[a] = g.in[a][base + (size_t)(g.K - 1)];
            tri_accumulate<TN>(acc, x);
        }
    } else {
#pragma unroll 4
        for (int t = lane; t < g.K; t += kWave) {
            float2 x[TN];
#pragma unroll
            for (int a = 0; a < TN; a++) x[a] = g.in[a][base + (size_t)t];
            tri_accumulate<TN>(acc, x);
        }
    }

    // wave all-reduce of the N^2 real partial sums
#pragma unroll
    for (int a = 0; a < TN; a++) acc.d[a] = wave_allreduce_sum(acc.d[a]);
#pragma unroll
    for (int i = 0; i < TN * (TN - 1) / 2; i++) {
        acc.re[i] = wave_allreduce_sum(acc.re[i]);
        acc.im[i] = wave_allreduce_sum(acc.im[i]);
    }

    // lane e = a + b*N picks element R[a,b]
    float2 r = make_float2(0.f, 0.f);
    {
        int idx = 0;
#pragma unroll
        for (int a = 0; a < TN; a++) {
            if (lane == a + a * TN) r = make_float2(acc.d[a], 0.f);
#pragma unroll
            for (int b = a + 1; b < TN; b++) {
                if (lane == a + b * TN) r = make_float2(acc.re[idx], acc.im[idx]);
                if (lane == b + a * TN) r = make_float2(acc.re[idx], -acc.im[idx]);
                idx++;
            }
        }
    }
    r.x = __fmul_rn(r.x, g.inv_k);
    r.y = __fmul_rn(r.y, g.inv_k);
    if (g.gain && lane < TN * TN) {            // fused antenna correction: R[a,b] *= g_a conj(g_b)
        const float2 w = g.gain[lane];
        r = make_float2(fmaf(w.x, r.x, -w.y * r.y), fmaf(w.x, r.y, w.y * r.x));
    }
    if (g.avg == 1) {
        // (J conj(R) J)[a,b] = conj(R[N-1-a, N-1-b])  ->  element index N^2-1-e
        const int src = (lane < TN * TN) ? (TN * TN - 1 - lane) : lane;
        const float px = __shfl(r.x, src, kWave);
        const float py = __shfl(r.y, src, kWave);
        r.x = __fadd_rn(__fmul_rn(0.5f, r.x), __fmul_rn(g.fb_hk, px));
        r.y = __fadd_rn(__fmul_rn(0.5f, r.y), __fmul_rn(g.fb_hk, -py));
    }
    if (lane < TN * TN) g.out[(size_t)snap * (TN * TN) + lane] = r;
    }  // snapshot loop
}

// ---------------------------------------------------------------------------------------------
// Overlapping windows, read-once.  With S = K - overlap and K = q S + r, cut every stream at j S and
// j S + r: step j owns piece A_j = [j S, j S + r) and piece B_j = [j S + r, (j+1) S), and
//     window i  =  sum_{j=i}^{i+q-1} (A_j + B_j)  +  A_{i+q}.
// cov_piece_kernel: one wave per step streams its S new samples ONCE (same loads, accumulators and DPP
// reduction as cov_wave_kernel) and writes the two raw N x N piece sums (2 x 8N^2 B per step);
// cov_combine_kernel adds the q+... pieces of each window and applies 1/K, the fused antenna
// correction and the forward-backward step.  HBM traffic = the new samples only (the single-kernel
// path re-fetches the overlap of almost every window: measured 253 MB against 201 MB algorithmic at
// K = 2048, overlap = 512), and the overlap's multiply-adds are not repeated either.  The price is
// one more (tiny) launch, so windows without overlap stay on cov_wave_kernel.
// ---------------------------------------------------------------------------------------------
template <int TN, int UN, bool NT>
__device__ __forceinline__ void accumulate_pairs(TriAcc<TN> &acc, const CovArgs &g, size_t first, int npair, int lane)
{
    const float2 *const *in = g.in;
    int p = lane;
    for (; p + (UN - 1) * kWave < npair; p += UN * kWave) {
        float4 v[UN][TN];
#pragma unroll
        for (int u = 0; u < UN; u++)
#pragma unroll
            for (int a = 0; a < TN; a++)
                v[u][a] = load_f4<NT>(reinterpret_cast<const float4 *>(in[a] + first + 2 * (size_t)(p + u * kWave)));
#pragma unroll
        for (int u = 0; u < UN; u++) {
            float2 x0[TN], x1[TN];
#pragma unroll
            for (int a = 0; a < TN; a++) { x0[a] = make_float2(v[u][a].x, v[u][a].y); x1[a] = make_float2(v[u][a].z, v[u][a].w); }
            tri_accumulate<TN>(acc, x0);
            tri_accumulate<TN>(acc, x1);
        }
    }
    for (; p < npair; p += kWave) {
        float4 v[TN];
#pragma unroll
        for (int a = 0; a < TN; a++) v[a] = load_f4<NT>(reinterpret_cast<const float4 *>(in[a] + first + 2 * (size_t)p));
        float2 x0[TN], x1[TN];
#pragma unroll
        for (int a = 0; a < TN; a++) { x0[a] = make_float2(v[a].x, v[a].y); x1[a] = make_float2(v[a].z, v[a].w); }
        tri_accumulate<TN>(acc, x0);
        tri_accumulate<TN>(acc, x1);
    }
}

template <int TN> __device__ __forceinline__ void tri_clear(TriAcc<TN> &acc)
{
#pragma unroll
    for (int a = 0; a < TN; a++) acc.d[a] = 0.f;
#pragma unroll
    for (int i = 0; i < TN * (TN - 1) / 2; i++) { acc.re[i] = 0.f; acc.im[i] = 0.f; }
}

// wave all-reduce, then lane e = a + b*N holds the raw sum of x_a conj(x_b)
template <int TN> __device__ __forceinline__ float2 tri_reduce_to_lane(TriAcc<TN> &acc, int lane)
{
#pragma unroll
    for (int a = 0; a < TN; a++) acc.d[a] = wave_allreduce_sum(acc.d[a]);
#pragma unroll
    for (int i = 0; i < TN * (TN - 1) / 2; i++) {
        acc.re[i] = wave_allreduce_sum(acc.re[i]);
        acc.im[i] = wave_allreduce_sum(acc.im[i]);
    }
    float2 r = make_float2(0.f, 0.f);
    int idx = 0;
#pragma unroll
    for (int a = 0; a < TN; a++) {
        if (lane == a + a * TN) r = make_float2(acc.d[a], 0.f);
#pragma unroll
        for (int b = a + 1; b < TN; b++) {
            if (lane == a + b * TN) r = make_float2(acc.re[idx], acc.im[idx]);
            if (lane == b + a * TN) r = make_float2(acc.re[idx], -acc.im[idx]);
            idx++;
        }
    }
    return r;
}

// requires: S and r even, every stream base 16-B aligned (float4 loads at every piece start)
template <int TN, int UN, bool NT>
__global__ __launch_bounds__(256) void cov_piece_kernel(CovArgs g)
{
    const int lane = threadIdx.x & (kWave - 1);
    const int wave0 = blockIdx.x * (blockDim.x / kWave) + (threadIdx.x / kWave);
    const int n_waves = gridDim.x * (blockDim.x / kWave);
    for (int step = wave0; step < g.n_steps; step += n_waves) {
        const size_t base = (size_t)step * (size_t)g.S;
        float2 *po = g.pieces + (size_t)step * (2 * TN * TN);
        TriAcc<TN> acc;
        if (g.r > 0) {                                     // piece A_j
            tri_clear<TN>(acc);
            accumulate_pairs<TN, UN, NT>(acc, g, base, g.r >> 1, lane);
            const float2 v = tri_reduce_to_lane<TN>(acc, lane);
            if (lane < TN * TN) po[lane] = v;
        }
        if (step + 1 < g.n_steps) {                        // piece B_j (the last step only contributes its A)
            tri_clear<TN>(acc);
            accumulate_pairs<TN, UN, NT>(acc, g, base + (size_t)g.r, (g.S - g.r) >> 1, lane);
            const float2 v = tri_reduce_to_lane<TN>(acc, lane);
            if (lane < TN * TN) po[TN * TN + lane] = v;
        }
    }
}

__global__ __launch_bounds__(256) void cov_combine_kernel(CovArgs g)
{
    const int nn = g.n_ch * g.n_ch;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)g.n_out * nn) return;
    const int i = (int)(idx / nn), e = (int)(idx % nn);
    auto window_sum = [&](int el) {
        float sx = 0.f, sy = 0.f;
        for (int j = i; j < i + g.q; j++) {
            const float2 *pj = g.pieces + (size_t)j * (2 * nn);
            if (g.r > 0) { const float2 a = pj[el]; sx = __fadd_rn(sx, a.x); sy = __fadd_rn(sy, a.y); }
            const float2 b = pj[nn + el];
            sx = __fadd_rn(sx, b.x); sy = __fadd_rn(sy, b.y);
        }
        if (g.r > 0) { const float2 a = g.pieces[(size_t)(i + g.q) * (2 * nn) + el]; sx = __fadd_rn(sx, a.x); sy = __fadd_rn(sy, a.y); }
        float2 r = make_float2(__fmul_rn(sx, g.inv_k), __fmul_rn(sy, g.inv_k));
        if (g.gain) {                                      // fused antenna correction: R[a,b] *= g_a conj(g_b)
            const float2 w = g.gain[el];
            r = make_float2(fmaf(w.x, r.x, -w.y * r.y), fmaf(w.x, r.y, w.y * r.x));
        }
        return r;
    };
    float2 r = window_sum(e);
    if (g.avg == 1) {                                      // (J conj(R) J)[a,b] = conj(R[N-1-a, N-1-b]) = element N^2-1-e
        const float2 m = window_sum(nn - 1 - e);
        r.x = __fadd_rn(__fmul_rn(0.5f, r.x), __fmul_rn(g.fb_hk, m.x));
        r.y = __fadd_rn(__fmul_rn(0.5f, r.y), __fmul_rn(g.fb_hk, -m.y));
    }
    g.out[idx] = r;
}

// Wide arrays (8 < N <= 16): the per-snapshot outer-product sum is a 16 x K by K x 16 complex GEMM,
// run on the matrix cores as four real v_mfma_f32_16x16x4_f32 per 4-sample step (exact fp32, same
// rate as the vector FMA pipe but with the whole 16 x 16 accumulator held in 8 registers):
//     Re R += Xr Xr^T + Xi Xi^T,      Im R += Xi Xr^T - Xr Xi^T.
// One wave owns one snapshot.  Lane l = (channel l&15, sample group l>>4) loads 4 consecutive samples
// (two 16-byte loads; 128 B contiguous per channel per wave-iteration) and feeds the SAME register as
// the A and the B operand (A[a][k] = x_a[t_k], B[k][b] = x_b[t_k]); channels >= N contribute zeros.
// No cross-lane reduction is needed: the K dimension is summed inside the MFMA accumulators, whose
// C/D layout (row = 4*(l>>4)+reg, col = l&15) is written straight to the column-major item.
typedef float f32x4_t __attribute__((ext_vector_type(4)));

template <bool VEC2> __global__ __launch_bounds__(256) void cov_mfma_kernel(CovArgs g)
{
    const int lane = threadIdx.x & (kWave - 1);
    const int wave0 = blockIdx.x * (blockDim.x / kWave) + (threadIdx.x / kWave);
    const int n_waves = gridDim.x * (blockDim.x / kWave);
    const int ch = lane & 15, grp = lane >> 4;
    const bool live = ch < g.n_ch;
    const float2 *src = g.in[live ? ch : 0];
    for (int snap = wave0; snap < g.n_out; snap += n_waves) {
        // VEC2: per 16-sample block this lane takes samples {2 grp, 2 grp + 1, 8 + 2 grp, 9 + 2 grp}: each 16-byte
        // load instruction then covers 64 contiguous bytes per channel (two full 32-byte sectors) instead of four
        // half-used ones; which four samples meet in one MFMA step does not matter for the sum
        const float2 *p = src + (size_t)snap * (size_t)g.S + (VEC2 ? 2 : 4) * grp;
        f32x4_t acc_re = {0.f, 0.f, 0.f, 0.f}, acc_im = {0.f, 0.f, 0.f, 0.f};
        auto step = [&](float xr, float xi) {
            acc_re = __builtin_amdgcn_mfma_f32_16x16x4f32(xr, xr, acc_re, 0, 0, 0);
            acc_im = __builtin_amdgcn_mfma_f32_16x16x4f32(xi, xr, acc_im, 0, 0, 0);
            acc_re = __builtin_amdgcn_mfma_f32_16x16x4f32(xi, xi, acc_re, 0, 0, 0);
            acc_im = __builtin_amdgcn_mfma_f32_16x16x4f32(-xr, xi, acc_im, 0, 0, 0);
        };
        int t = 0;
        constexpr int UN = 4;                          // 4 wave-iterations (64 samples) of loads in flight
        for (; t + 16 * UN <= g.K; t += 16 * UN) {
            float2 x[UN][4];
#pragma unroll
            for (int u = 0; u < UN; u++) {
                if constexpr (VEC2) {
                    // default cache policy on purpose: the two loads of a block share 128-byte lines and
                    // want to meet in L1 (non-temporal: 169 vs 120 us)
                    const float4 v0 = *reinterpret_cast<const float4 *>(p + t + 16 * u);
                    const float4 v1 = *reinterpret_cast<const float4 *>(p + t + 16 * u + 8);
                    x[u][0] = make_float2(v0.x, v0.y); x[u][1] = make_float2(v0.z, v0.w);
                    x[u][2] = make_float2(v1.x, v1.y); x[u][3] = make_float2(v1.z, v1.w);
                } else {
#pragma unroll
                    for (int j = 0; j < 4; j++) x[u][j] = p[t + 16 * u + j];
                }
            }
#pragma unroll
            for (int u = 0; u < UN; u++)
#pragma unroll
                for (int j = 0; j < 4; j++) step(live ? x[u][j].x : 0.f, live ? x[u][j].y : 0.f);
        }
        for (; t < g.K; t += 16) {                      // remainder, sample by sample with bounds checks
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int sidx = t + 4 * grp + j;
                float2 v = make_float2(0.f, 0.f);
                if (live && sidx < g.K) v = src[(size_t)snap * (size_t)g.S + sidx];
                step(v.x, v.y);
            }
        }
        float2 *item = g.out + (size_t)snap * g.n_ch * g.n_ch;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int a = 4 * grp + r;
            if (a < g.n_ch && live) {
                float2 v = make_float2(__fmul_rn(acc_re[r], g.inv_k), __fmul_rn(acc_im[r], g.inv_k));
                if (g.gain) {
                    const float2 w = g.gain[a + ch * g.n_ch];
                    v = make_float2(fmaf(w.x, v.x, -w.y * v.y), fmaf(w.x, v.y, w.y * v.x));
                }
                item[a + (size_t)ch * g.n_ch] = v;
            }
        }
    }
}

// R <- 0.5 R + (0.5/K) conj(R[N-1-a, N-1-b]), pairwise in place (element e with N^2-1-e).
__global__ void cov_fb_kernel(float2 *out, int nn, long long n_items, float fb_hk)
{
    const int half = (nn + 1) / 2;
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= n_items * half) return;
    const long long item = gid / half;
    const int e = (int)(gid - item * half), e2 = nn - 1 - e;
    float2 *p = out + item * nn;
    const float2 u = p[e], v = p[e2];
    float2 nu, nv;
    nu.x = __fadd_rn(__fmul_rn(0.5f, u.x), __fmul_rn(fb_hk, v.x));
    nu.y = __fadd_rn(__fmul_rn(0.5f, u.y), __fmul_rn(fb_hk, -v.y));
    nv.x = __fadd_rn(__fmul_rn(0.5f, v.x), __fmul_rn(fb_hk, u.x));
    nv.y = __fadd_rn(__fmul_rn(0.5f, v.y), __fmul_rn(fb_hk, -u.y));
    p[e] = nu;
    if (e2 != e) p[e2] = nv;
}

// Waves per CU one launch may occupy (0 = one wave per snapshot, no cap); larger launches grid-stride.  At N = 4,
// 16 (= one wave per snapshot at the benchmark batch: 4096 snapshots on 256 CUs) measured against 8 with the round-1
// kernels: -0.6 us on the kernel alone (23.2 vs 23.8 us) and -1.2 us per 4-stream pipeline step (25.6 vs 27.1).
static int cov_waves_per_cu(int n_ch)
{
    // N <= 4: one wave per snapshot at the benchmark batch; wider arrays hold 2-4x the registers per wave and measured
    // better with 8 (N = 6: 54.4 vs 57.3 us, 4-stream step 56.8 vs 63.0 us)
    const int v = DOA_LAB_ENV_INT("DOA_COV_WAVES_PER_CU", -1);
    return v >= 0 ? v : (n_ch <= 4 ? 16 : 8);
}

// the read-once two-kernel path (see cov_piece_kernel)
template <int TN> static void launch_pieces(const CovArgs &g, hipStream_t st)
{
    const int waves_per_block = 4;
    int blocks = (g.n_steps + waves_per_block - 1) / waves_per_block;
    if (cov_waves_per_cu(TN) > 0) {
        const int cap = cu_count() * cov_waves_per_cu(TN) / waves_per_block;
        if (blocks > cap) blocks = cap;
    }
    constexpr int UN = (TN <= 6) ? 4 : 2;
    hipLaunchKernelGGL((cov_piece_kernel<TN, UN, true>), dim3(blocks), dim3(waves_per_block * kWave), 0, st, g);
    const long long total = (long long)g.n_out * g.n_ch * g.n_ch;
    hipLaunchKernelGGL(cov_combine_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, g);
}

template <int TN> static void launch_wave(const CovArgs &g, bool vec2, hipStream_t st)
{
    if (g.pieces) { launch_pieces<TN>(g, st); return; }
    const int waves_per_block = 4;
    int blocks = (g.n_out + waves_per_block - 1) / waves_per_block;
    if (cov_waves_per_cu(TN) > 0) {
        const int cap = cu_count() * cov_waves_per_cu(TN) / waves_per_block;
        if (blocks > cap) blocks = cap;
    }
    dim3 grid(blocks), block(waves_per_block * kWave);
    // UN*TN 16-byte loads in flight per wave (UN = 1, 2, 8 measured within 3 % of UN = 4 at N = 4; N = 6: 44 us with
    // UN = 4 against 55 with 2; N = 7: 68 us with 2 against 74 with 1)
    constexpr int UN = (TN <= 6) ? 4 : 2;
    if (!vec2) {
        hipLaunchKernelGGL((cov_wave_kernel<TN, false, 1, false>), grid, block, 0, st, g);
        return;
    }
    // read-once stream: non-temporal loads (+19 % on MI355X against the default cache policy)
    hipLaunchKernelGGL((cov_wave_kernel<TN, true, UN, true>), grid, block, 0, st, g);
}

// Launches K1 on `st`.  d_in: N device pointers.  Returns DOA_OK / error.
// bytes of piece-sum workspace launch_autocorrelate wants for n_out windows (0: the shape does not
// take the read-once path)
size_t autocorrelate_workspace_bytes(int N, int K, int ovl, int n_out)
{
    const int S = K - ovl;
    if (ovl <= 0 || N > 8 || n_out <= 0 || (S & 1) || ((K % S) & 1)) return 0;
    return (size_t)(n_out + K / S) * 2 * N * N * sizeof(float2);
}

int launch_autocorrelate(int N, int K, int ovl, int avg, int n_out, const void *const *d_in, void *d_out,
                         hipStream_t st, const void *d_gain_outer, void *d_workspace)
{
    if (n_out <= 0) return DOA_OK;
    CovArgs g;
    memset(&g, 0, sizeof g);
    bool vec2 = ((K - ovl) % 2 == 0);
    for (int k = 0; k < N; k++) {
        g.in[k] = static_cast<const float2 *>(d_in[k]);
        if (!d_in[k]) { set_error("autocorrelate: input stream %d is NULL", k); return DOA_ERR_INVALID_ARG; }
        if (reinterpret_cast<uintptr_t>(d_in[k]) % 8) { set_error("autocorrelate: input stream %d is not 8-byte aligned", k); return DOA_ERR_INVALID_ARG; }
        if (reinterpret_cast<uintptr_t>(d_in[k]) % 16) vec2 = false;
    }
    for (int k = N; k < DOA_MAX_ANT_ELE; k++) g.in[k] = g.in[0];
    g.out = static_cast<float2 *>(d_out);
    g.n_ch = N; g.K = K; g.S = K - ovl; g.n_out = n_out; g.avg = avg;
    g.inv_k = (float)(1.0 / K);
    g.fb_hk = (float)(0.5 / K);
    g.gain = static_cast<const float2 *>(d_gain_outer);
    if (d_workspace && vec2 && autocorrelate_workspace_bytes(N, K, ovl, n_out) > 0) {
        g.pieces = static_cast<float2 *>(d_workspace);
        g.q = K / g.S; g.r = K % g.S; g.n_steps = n_out + g.q;
    }
    switch (N) {
    case 1: launch_wave<1>(g, vec2, st); break;
    case 2: launch_wave<2>(g, vec2, st); break;
    case 3: launch_wave<3>(g, vec2, st); break;
    case 4: launch_wave<4>(g, vec2, st); break;
    case 5: launch_wave<5>(g, vec2, st); break;
    case 6: launch_wave<6>(g, vec2, st); break;
    case 7: launch_wave<7>(g, vec2, st); break;
    case 8: launch_wave<8>(g, vec2, st); break;
    default: {
        int blocks = (n_out + 3) / 4;
        const int mfma_wpc = DOA_LAB_ENV_INT("DOA_COV_MFMA_WAVES_PER_CU", 16);
        if (blocks > cu_count() * mfma_wpc / 4) blocks = cu_count() * mfma_wpc / 4;        // <= 16 waves per CU, grid-stride beyond
        if (vec2) hipLaunchKernelGGL(cov_mfma_kernel<true>, dim3(blocks), dim3(256), 0, st, g);
        else      hipLaunchKernelGGL(cov_mfma_kernel<false>, dim3(blocks), dim3(256), 0, st, g);
        if (avg == 1) {
            const int nn = N * N, half = (nn + 1) / 2;
            const long long total = (long long)n_out * half;
            hipLaunchKernelGGL(cov_fb_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, g.out, nn,
                               (long long)n_out, g.fb_hk);
        }
    }
    }
    DOA_HIP_TRY(hipGetLastError());
    return DOA_OK;
}

}  // namespace doa

// ---------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------
struct doa_autocorrelate {
    int inputs, snapshot, overlap, avg;
    int device;
    hipStream_t stream = nullptr;
    doa::DevBuf d_in, d_out, d_gain, d_work;
    bool has_gain = false;
};

extern "C" {

doa_autocorrelate_t *doa_autocorrelate_create(int inputs, int snapshot_size, int overlap_size, int avg_method)
{
    doa::clear_error();
    // grc/doa_autocorrelate.xml:41-43 checks; the C++ ctor of the reference does not validate, so a
    // violating call would read/write out of bounds there — here it fails in create.
    if (inputs <= 0 || snapshot_size <= 0 || overlap_size < 0 || overlap_size >= snapshot_size) {
        doa::set_error("autocorrelate: need inputs > 0, snapshot_size > 0, 0 <= overlap_size < snapshot_size "
                       "(got %d, %d, %d)", inputs, snapshot_size, overlap_size);
        return nullptr;
    }
    if (inputs > DOA_MAX_ANT_ELE) {
        doa::set_error("autocorrelate: inputs=%d exceeds DOA_MAX_ANT_ELE=%d", inputs, DOA_MAX_ANT_ELE);
        return nullptr;
    }
    int dev = 0;
    if (doa::ensure_device(&dev) != DOA_OK) return nullptr;
    auto *h = new (std::nothrow) doa_autocorrelate();
    if (!h) { doa::set_error("out of memory"); return nullptr; }
    h->inputs = inputs; h->snapshot = snapshot_size; h->overlap = overlap_size; h->avg = avg_method;
    h->device = dev;
    if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) {
        doa::set_error("autocorrelate: hipStreamCreate failed");
        delete h;
        return nullptr;
    }
    return h;
}

void doa_autocorrelate_destroy(doa_autocorrelate_t *h)
{
    if (!h) return;
    h->d_in.release();
    h->d_out.release();
    h->d_gain.release();
    h->d_work.release();
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

int doa_autocorrelate_fuse_antenna_correction(doa_autocorrelate_t *h, const float *gains_re_im)
{
    doa::clear_error();
    if (!h) return DOA_ERR_INVALID_ARG;
    if (!gains_re_im) { h->has_gain = false; return DOA_OK; }
    const int N = h->inputs;
    float2 w[DOA_MAX_ANT_ELE * DOA_MAX_ANT_ELE];
    for (int b = 0; b < N; b++)
        for (int a = 0; a < N; a++) {
            const float ar = gains_re_im[2 * a], ai = gains_re_im[2 * a + 1], br = gains_re_im[2 * b], bi = gains_re_im[2 * b + 1];
            w[a + b * N] = make_float2(ar * br + ai * bi, ai * br - ar * bi);   // g_a conj(g_b)
        }
    int rc = h->d_gain.reserve(sizeof(float2) * N * N);
    if (rc != DOA_OK) return rc;
    DOA_HIP_TRY(hipMemcpy(h->d_gain.p, w, sizeof(float2) * N * N, hipMemcpyHostToDevice));
    h->has_gain = true;
    return DOA_OK;
}

int doa_autocorrelate_history(const doa_autocorrelate_t *h) { return h ? h->overlap + 1 : DOA_ERR_INVALID_ARG; }

int doa_autocorrelate_forecast(const doa_autocorrelate_t *h, int noutput_items)
{
    if (!h || noutput_items < 0) return DOA_ERR_INVALID_ARG;
    return (h->snapshot - h->overlap) * noutput_items;
}

long long doa_autocorrelate_input_span(const doa_autocorrelate_t *h, int noutput_items)
{
    if (!h || noutput_items <= 0) return 0;
    return (long long)(noutput_items - 1) * (h->snapshot - h->overlap) + h->snapshot;
}

int doa_autocorrelate_work_dev(doa_autocorrelate_t *h, int noutput_items, const void *const *d_input_items,
                               void *d_output_items0, void *hip_stream)
{
    doa::clear_error();
    if (!h || noutput_items < 0 || !d_input_items || (!d_output_items0 && noutput_items > 0)) {
        doa::set_error("autocorrelate_work_dev: bad arguments");
        return DOA_ERR_INVALID_ARG;
    }
    if (int brc = doa::bind_device(h->device); brc != DOA_OK) return brc;
    // piece-sum workspace of the read-once path (grow-only; growing frees the old buffer, which waits for the device)
    const size_t ws = doa::autocorrelate_workspace_bytes(h->inputs, h->snapshot, h->overlap, noutput_items);
    int rc = ws ? h->d_work.reserve(ws) : DOA_OK;
    if (rc != DOA_OK) return rc;
    rc = doa::launch_autocorrelate(h->inputs, h->snapshot, h->overlap, h->avg, noutput_items, d_input_items,
                                   d_output_items0, static_cast<hipStream_t>(hip_stream),
                                   h->has_gain ? h->d_gain.p : nullptr, ws ? h->d_work.p : nullptr);
    return rc == DOA_OK ? noutput_items : rc;
}

int doa_autocorrelate_work(doa_autocorrelate_t *h, int noutput_items, const void *const *input_items,
                           void *output_items0)
{
    doa::clear_error();
    if (!h || noutput_items < 0 || !input_items || (!output_items0 && noutput_items > 0)) {
        doa::set_error("autocorrelate_work: bad arguments");
        return DOA_ERR_INVALID_ARG;
    }
    if (noutput_items == 0) return 0;
    if (int brc = doa::bind_device(h->device); brc != DOA_OK) return brc;
    const int N = h->inputs;
    const size_t span = (size_t)doa_autocorrelate_input_span(h, noutput_items);
    // distance between the device copies of the streams: 16-B aligned, staggered against the 8 KiB aliasing period
    const size_t span_al = doa::stream_stride_bytes(span * sizeof(float2)) / sizeof(float2);
    int rc = h->d_in.reserve(span_al * N * sizeof(float2));
    if (rc != DOA_OK) return rc;
    const size_t out_bytes = (size_t)noutput_items * N * N * sizeof(float2);
    rc = h->d_out.reserve(out_bytes);
    if (rc != DOA_OK) return rc;
    const void *d_ptrs[DOA_MAX_ANT_ELE];
    for (int k = 0; k < N; k++) {
        if (!input_items[k]) { doa::set_error("autocorrelate_work: input_items[%d] is NULL", k); return DOA_ERR_INVALID_ARG; }
        float2 *dst = h->d_in.as<float2>() + k * span_al;
        DOA_HIP_TRY(hipMemcpyAsync(dst, input_items[k], span * sizeof(float2), hipMemcpyHostToDevice, h->stream));
        d_ptrs[k] = dst;
    }
    const size_t ws = doa::autocorrelate_workspace_bytes(N, h->snapshot, h->overlap, noutput_items);
    if (ws) rc = h->d_work.reserve(ws);
    if (rc != DOA_OK) return rc;
    rc = doa::launch_autocorrelate(N, h->snapshot, h->overlap, h->avg, noutput_items, d_ptrs, h->d_out.p, h->stream,
                                   h->has_gain ? h->d_gain.p : nullptr, ws ? h->d_work.p : nullptr);
    if (rc != DOA_OK) return rc;
    DOA_HIP_TRY(hipMemcpyAsync(output_items0, h->d_out.p, out_bytes, hipMemcpyDeviceToHost, h->stream));
    DOA_HIP_TRY(hipStreamSynchronize(h->stream));
    return noutput_items;
}

}  // extern "C"
