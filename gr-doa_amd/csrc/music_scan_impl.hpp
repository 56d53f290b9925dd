// music_scan_impl.hpp — K4 (spectrum scan, optionally with K5 fused) kernels and their launcher for one
// compiled polynomial size; included only by music_scan_inst.hip.
#pragma once
#include "music_scan.hpp"
#include "peak_device.hpp"

#include <climits>
#include <cmath>
#include <cstdlib>
#include <type_traits>

namespace doa {

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

// The same arithmetic for U angles at once, the U Horner chains advanced step by step side by side (U independent
// dependency chains in flight: what a kernel with few waves per SIMD needs to keep the FP64 pipe busy).
template <int N, typename T, int U>
__device__ __forceinline__ void null_spectrum_multi(const T (&c)[2 * N], const T (&zr)[U], const T (&zi)[U], float (&q)[U])
{
    if constexpr (N == 1) {
#pragma unroll
        for (int u = 0; u < U; u++) q[u] = (float)c[0];
    } else {
        T hr[U], hi[U];
#pragma unroll
        for (int u = 0; u < U; u++) { hr[u] = c[2 * (N - 1) - 1]; hi[u] = c[2 * (N - 1)]; }
#pragma unroll
        for (int l = N - 2; l >= 1; l--) {
            // one Horner step of all U chains: the 2U inner products first, then the 2U outer ones, so that no instruction
            // depends on one of the 2U - 1 before it.  (Left alone, instruction selection puts the chains back one after the
            // other to save registers -- a scheduling barrier does not bind side-effect-free arithmetic; an empty asm that
            // "modifies" all running values does.)
            static_assert(U == 4, "the ordering fences below name their operands");
            T ir[U], ii[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                ir[u] = fma(-hi[u], zi[u], c[2 * l - 1]);
                ii[u] = fma(hi[u], zr[u], c[2 * l]);
            }
            asm volatile("" : "+v"(ir[0]), "+v"(ii[0]), "+v"(ir[1]), "+v"(ii[1]), "+v"(ir[2]), "+v"(ii[2]), "+v"(ir[3]), "+v"(ii[3]));
#pragma unroll
            for (int u = 0; u < U; u++) {
                const T tr = fma(hr[u], zr[u], ir[u]);
                const T ti = fma(hr[u], zi[u], ii[u]);
                hr[u] = tr; hi[u] = ti;
            }
            asm volatile("" : "+v"(hr[0]), "+v"(hi[0]), "+v"(hr[1]), "+v"(hi[1]), "+v"(hr[2]), "+v"(hi[2]), "+v"(hr[3]), "+v"(hi[3]));
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const T re = fma(hr[u], zr[u], -hi[u] * zi[u]);
            q[u] = (float)fma((T)2, re, c[0]);
        }
    }
}

// ---- normalisation to the maximum: 10*log10(out/max(out)), out = 1/Q (reference :140-142) ------------------
// The reference forms out = 1.0/Q (double division stored to float: correctly rounded) and divides by the
// maximum in float; an angle is at exactly 0 dB iff its ROUNDED reciprocal equals the largest one (x/x == 1),
// and find_local_max then takes the first such angle (find_local_max_impl.h:53-56).  The kernels below keep
// exactly that tie set: v_rcp_f32 / v_log_f32 do the bulk, but every angle that could tie with the maximum goes
// through IEEE divisions, and 10*log10(r) for r within 1e-5 of 1 is 4.3429448 (r - 1) (r - 1 is exact there;
// the hardware log2 loses it), so no angle other than a true tie can come out as 0 dB or above.
constexpr float kDbPerLog2 = 3.0102999566398120f;       // 10 log10(2)
constexpr float kDbPerUnit = 4.3429448190325183f;       // 10 / ln(10)

__device__ __forceinline__ float db_from_ratio(float out, float mx, float inv_mx)
{
    // general form: out, mx are correctly rounded reciprocals, out <= mx (NaN/inf rows: inf/inf stays NaN as
    // in the reference)
    if (out == mx && mx != INFINITY) return kDbPerLog2 * __log2f(1.0f);
    const float approx = out * inv_mx;
    float db = kDbPerLog2 * __log2f(fminf(approx, 1.0f));
    if (approx > 0.99999f) db = kDbPerUnit * (out / mx - 1.0f);      // IEEE division: < 1 since out < mx
    return db;
}

// Lean form, from Q itself and its item minimum mn (a normal positive float): max(out) = RN(1/mn) =: mx, and
// because RN(1/q) is monotone in q the tie set {q : RN(1/q) == mx} is an interval [mn, q_hi] of floats.  Two floats
// 1 ulp apart have reciprocals at least half an ulp (of the result) apart, so at most TWO consecutive floats can share a
// rounded reciprocal (a third would need 1/q to sit exactly on a rounding boundary twice: q a power of two twice):
// q_hi is mn or mn + 1 ulp.  Whether mn + 1 ulp ties is decided exactly without a second division:
//     RN(1/q) == mx  <=>  |1/q - mx| <= ulp(mx)/2  <=>  |mx q - 1| <= (ulp(mx)/2) q,
// where r = fma(mx, q, -1) is exact (mx q is within 2^-22 of 1, so the difference has fewer than 24 significant
// bits) and (ulp(mx)/2) q is a power-of-two multiple of q.  (Equality would put 1/q exactly on the boundary, which
// needs q to be a power of two; then 1/q is itself a float and the comparison is strict anyway.)
// Per angle the whole rule is then ONE compare, tie = (q <= q_hi), and
//     dB = tie ? 0 : -10 log10(2) log2(q inv_up),      inv_up = v_rcp_f32(mn) (1 + 2^-22):
// the reciprocal is biased UP by two ulp so that every non-tied angle (q >= mn (1 + 2^-23)) has q inv_up > 1 whatever
// v_rcp_f32 and the product round to, i.e. a strictly negative dB without a clamp; the price is a common -1e-6 dB on
// the whole row, a quarter of v_log_f32's own error at -30 dB.
struct LeanNorm {
    float inv_up, q_hi;
    __device__ __forceinline__ explicit LeanNorm(float mn) : inv_up(__builtin_amdgcn_rcpf(mn) * 1.00000023841857910f), q_hi(mn)
    {
        const float mx = 1.0f / mn;                                        // IEEE: RN(1/mn)
        const float q1 = __int_as_float(__float_as_int(mn) + 1);           // mn + 1 ulp
        const float half_ulp = __int_as_float((__float_as_int(mx) & 0x7f800000) - (24 << 23));   // 2^(e - 24)
        const float r = fmaf(mx, q1, -1.0f);
        q_hi = (fabsf(r) < half_ulp * q1) ? q1 : mn;
    }
    // dB of a non-tied angle (meaningless, slightly negative, for a tied one)
    __device__ __forceinline__ float db_fast(float q) const { return -kDbPerLog2 * __log2f(q * inv_up); }
    __device__ __forceinline__ float db(float q, bool &tie) const
    {
        tie = (q <= q_hi);
        return tie ? 0.0f : db_fast(q);
    }
};
// (mx = 1/mn must be a normal float with ulp(mx)/2 representable: mn between 1e-30 and 1e30 -- the null spectrum of
// a pre-scaled projector is O(1); anything else takes the general path)
__device__ __forceinline__ bool lean_norm_ok(float mn) { return (mn >= 1e-30f) && (mn <= 1e30f); }

// Fast path: P % 4 == 0 and P <= 256*CH.  One wave per item, grid-stride over items so that the
// z table (4*CH angles per lane) is loaded once per wave and stays in registers.  With PEAK the
// find_local_max step (K5) runs on the dB values while they are still in registers, so the spectrum
// is written once and never read back.
template <int N, int CH, typename T, bool HAS_Q, bool PEAK, bool NT = false, bool ZREG = true>
__global__ __launch_bounds__(256) void music_scan_kernel(const T *__restrict__ coef, const T *__restrict__ ztab,
                                                         float *__restrict__ spec, float *__restrict__ qout, int P,
                                                         int n_items, const float *__restrict__ xaxis,
                                                         float *__restrict__ pk_val, float *__restrict__ pk_loc, int M,
                                                         int n_ant)
{
    // N is the compiled polynomial size (>= n_ant, the array's element count): records are 2*n_ant
    // values long and the missing high-order coefficients are zero, which leaves Q unchanged
    const int rec = 2 * n_ant;
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(blockIdx.x * (blockDim.x / kWave) + (threadIdx.x / kWave));
    const int n_waves = gridDim.x * (blockDim.x / kWave);

    // the x axis of the fused peak pick lives in LDS: the arg-max -> location lookup at the end of
    // every item is then a ~100-cycle ds_read instead of a dependent global load the wave has to
    // sit out (measured: 30 % of the wave's cycles were spent in that wait)
    __shared__ float xs[PEAK ? 256 * CH : 1];
    if constexpr (PEAK) {
        for (int i = threadIdx.x; i < 256 * CH; i += blockDim.x) xs[i] = (i < P) ? xaxis[i] : 0.f;
        __syncthreads();
    }
    // ZREG: the z table of this lane's 4*CH angles stays in registers across items.  Long spectra in
    // double (CH > 4: more than 128 VGPRs of table) re-read it from L2 per item instead, which is
    // noise next to their (N-1)-step double Horner.
    constexpr int ZCH = ZREG ? CH : 1;
    T zr[ZCH][4], zi[ZCH][4];
    if constexpr (ZREG) {
#pragma unroll
        for (int j = 0; j < CH; j++) {
            const int i0 = 4 * lane + 256 * j;
#pragma unroll
            for (int e = 0; e < 4; e++) {
                if (i0 < P) { zr[j][e] = ztab[2 * (i0 + e)]; zi[j][e] = ztab[2 * (i0 + e) + 1]; }
                else { zr[j][e] = 1; zi[j][e] = 0; }
            }
        }
    }
    // coefficient records arrive through scalar loads (wave-uniform address); the next item's record
    // is requested before this item's arithmetic so its latency hides behind it
    // (records of more than 64 dwords -- N > 8 in double -- are not double-buffered: two of them do not fit
    // the scalar register file and end up in VGPRs, which the long-spectrum variants cannot spare)
    constexpr bool PREFETCH = (2 * N * sizeof(T) <= 256);
    T c[2 * N], c_next[PREFETCH ? 2 * N : 1];
    if constexpr (PREFETCH) {
        if (wave < n_items) {
#pragma unroll
            for (int k = 0; k < 2 * N; k++) c_next[k] = (k < rec - 1) ? coef[(size_t)wave * rec + k] : (T)0;
        }
    }
    // One item: Q at this lane's 4*CH angles, 1/Q, wave maximum, dB, stores, optional peak pick.
    // FULL (P == 256*CH, the usual power-of-two lengths) drops every bounds predicate.
    auto do_item = [&](int item, auto full_tag) {
        constexpr bool FULL = decltype(full_tag)::value;
        float out[CH][4];
        float mx = -INFINITY;
        // !ZREG: the table loads do not depend on the item, so the optimiser would hoist them out of the
        // item loop and rebuild the register-resident table this variant exists to avoid (432-512 VGPRs at
        // CH = 16): launder the pointer once per item
        const T *zt = ztab;
        if constexpr (!ZREG) asm volatile("" : "+s"(zt));
#pragma unroll
        for (int j = 0; j < CH; j++) {
            const bool live = FULL || ((4 * lane + 256 * j) < P);
            if constexpr (!ZREG) {
                const int i0 = 4 * lane + 256 * j;
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    if (live) { zr[0][e] = zt[2 * (i0 + e)]; zi[0][e] = zt[2 * (i0 + e) + 1]; }
                    else { zr[0][e] = 1; zi[0][e] = 0; }
                }
            }
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const float q = (float)null_spectrum<N, T>(c, zr[ZREG ? j : 0][e], zi[ZREG ? j : 0][e]);
                if constexpr (HAS_Q) {
                    if (live) qout[(size_t)item * P + 4 * lane + 256 * j + e] = q;
                }
                out[j][e] = 1.0f / q;                           // 1.0/Q  (:140), correctly rounded as there
                mx = live ? fmaxf(mx, out[j][e]) : mx;
            }
            // table re-read per chunk: keep the scheduler from hoisting all CH chunks' loads to the top
            // (16 chunks x 8 doubles of table = 256 VGPRs, i.e. one wave per SIMD and spills)
            if constexpr (!ZREG) __builtin_amdgcn_sched_barrier(0);
        }
        mx = wave_allreduce_max(mx);
        float *row = spec + (size_t)item * P;
        // common case, wave-uniform: a finite positive maximum.  Then every ratio out/max is <= 1, the
        // maximum maps to exactly 0 dB, and for num_max_vals == 1 the arg-max (arma index_max: first
        // occurrence of the largest value, NaNs never win) is simply the first position whose dB is 0.
        const bool regular = (mx > 0.0f) && (mx < INFINITY);
        if (regular) {
            const float inv_mx = __builtin_amdgcn_rcpf(mx);
            int first_zero = INT_MAX;
#pragma unroll
            for (int j = CH - 1; j >= 0; j--) {
                const int i0 = 4 * lane + 256 * j;
#pragma unroll
                for (int e = 3; e >= 0; e--) {
                    const float db = db_from_ratio(out[j][e], mx, inv_mx);
                    out[j][e] = db;
                    if constexpr (PEAK) {
                        if (FULL || i0 < P) first_zero = (db == 0.0f) ? (i0 + e) : first_zero;
                    }
                }
                if (FULL || i0 < P)
                    store_f4<NT>(reinterpret_cast<float4 *>(row + i0), make_float4(out[j][0], out[j][1], out[j][2], out[j][3]));
            }
            if constexpr (PEAK) {
                if (M == 1) {
                    const int pos = wave_allreduce_min_int(first_zero);
                    if (lane == 0) {
                        pk_val[(size_t)item] = 0.0f;                 // the maximum is exactly 0 dB
                        pk_loc[(size_t)item] = xs[pos];
                    }
                    return;
                }
            }
        } else {
            const float inv_mx = __builtin_amdgcn_rcpf(mx);
#pragma unroll
            for (int j = 0; j < CH; j++) {
                const int i0 = 4 * lane + 256 * j;
#pragma unroll
                for (int e = 0; e < 4; e++) out[j][e] = db_from_ratio(out[j][e], mx, inv_mx);
                if (FULL || i0 < P)
                    store_f4<NT>(reinterpret_cast<float4 *>(row + i0), make_float4(out[j][0], out[j][1], out[j][2], out[j][3]));
            }
        }
        if constexpr (PEAK) peak_pick<CH>(out, lane, P, M, xs, pk_val + (size_t)item * M, pk_loc + (size_t)item * M);
    };
    const bool full = (P == 256 * CH);
    for (int item = wave; item < n_items; item += n_waves) {
        if constexpr (PREFETCH) {
#pragma unroll
            for (int k = 0; k < 2 * N; k++) c[k] = c_next[k];
            const int nxt = item + n_waves;
            if (nxt < n_items) {
#pragma unroll
                for (int k = 0; k < 2 * N; k++) c_next[k] = (k < rec - 1) ? coef[(size_t)nxt * rec + k] : (T)0;
            }
        } else {
#pragma unroll
            for (int k = 0; k < 2 * N; k++) c[k] = (k < rec - 1) ? coef[(size_t)item * rec + k] : (T)0;
        }
        if (full) do_item(item, std::true_type{});
        else do_item(item, std::false_type{});
    }
}

// N <= 4 in double: Q(psi) = u0 + 2 sum_l (a_l cos(l psi) - b_l sin(l psi)), u_l = a_l + j b_l, rewritten
// with cos 2x = 2c^2-1, cos 3x = 4c^3-3c, sin 2x = 2sc, sin 3x = s(4c^2-1) (c = cos psi, s = sin psi) as
//     Q = A(c) + s B(c),   A = (u0-2a2) + (2a1-6a3) c + 4a2 c^2 + 8a3 c^3,   B = (2b3-2b1) - 4b2 c - 8b3 c^2
// : 6 fused multiply-adds per angle instead of the 11 of the complex Horner form (the per-item
// coefficient transform is wave-uniform).  Degree 3 keeps the Chebyshev -> monomial change of basis
// harmless (|coefficients| grow by <= 8); it is used for the double path only, where its rounding
// (~1e-15 of the largest term) is far below the 1e-7 the float output resolves.
template <int N, typename T> struct ChebQ {
    T a0, a1, a2, a3, b0, b1, b2;
    __device__ __forceinline__ explicit ChebQ(const T (&c)[2 * N])
    {
        const T u0 = c[0];
        const T x1 = (N > 1) ? c[1] : (T)0, y1 = (N > 1) ? c[2] : (T)0;
        const T x2 = (N > 2) ? c[3] : (T)0, y2 = (N > 2) ? c[4] : (T)0;
        const T x3 = (N > 3) ? c[5] : (T)0, y3 = (N > 3) ? c[6] : (T)0;
        a0 = u0 - 2 * x2; a1 = 2 * x1 - 6 * x3; a2 = 4 * x2; a3 = 8 * x3;
        b0 = 2 * y3 - 2 * y1; b1 = -4 * y2; b2 = -8 * y3;
    }
    // from the record the EVD kernel already wrote in this basis (launch_music_evd, d_cheb): [a0 a1 a2 a3 b0 b1 b2 0]
    struct Pre {};
    __device__ __forceinline__ ChebQ(Pre, const T (&r)[8]) : a0(r[0]), a1(r[1]), a2(r[2]), a3(r[3]), b0(r[4]), b1(r[5]), b2(r[6]) {}
    __device__ __forceinline__ T operator()(T cs, T sn) const
    {
        const T A = fma(fma(fma(a3, cs, a2), cs, a1), cs, a0);
        const T B = fma(fma(b2, cs, b1), cs, b0);
        return fma(sn, B, A);
    }
};

// The benchmark shape of K4+K5, lean: P == 256*CH exactly (no bounds predicates), compiled polynomial size ==
// the array size, coefficients wave-uniform.  lean_scan_item is one item of it.  Nothing generic is compiled in, which keeps it at ~100 VGPRs (4 waves per SIMD) where
// the general kernel needs 200+:
//   * records: for N <= 4 in double the EVD kernel has already written the polynomial in the basis this kernel evaluates
//     (launch_music_evd's d_cheb, 8 doubles per item): the change of basis is per-item work and used to be redone by every
//     wave here (14 FP64 instructions and 10 VGPRs per item); otherwise the u_l record itself;
//   * pass 1 keeps Q itself, the minimum of every 4-angle group of a lane, and the wave minimum (no reciprocal per angle);
//   * pass 2 is dB = -10 log10(2) log2(Q (1/Qmin)): ONE transcendental per angle;
//   * the tie rule (LeanNorm) costs ONE compare per lane and 256-angle chunk on ordinary rows: a lane can hold a tied angle
//     in a chunk only if that chunk's lane minimum is <= q_hi; the per-angle compare / select (exactly 0 dB on the tied
//     angles) and the scalar position search (s_ff1, s_min: find_local_max's answer for num_max_vals = 1 is the first angle
//     holding the maximum) run only in chunks where some lane reports one -- one chunk of four on ordinary rows.  Every
//     non-tied angle is strictly negative by construction (LeanNorm::inv_up), so skipping the select elsewhere changes no bit;
//   * num_max_vals > 1 (MULTI, the flowgraph's two sources): the dB values stay in registers and go through the
//     general peak_pick.
// Items whose minimum of Q is not a normal positive float (zero, negative or non-finite null spectrum:
// non-finite input, in practice) take a slow rolled path that follows the general kernel's semantics literally.
template <int N, typename T> struct LeanRecord {
    static constexpr bool kPre = (N <= 4 && sizeof(T) == 8);       // Chebyshev-form records from the EVD kernel
    static constexpr int kLen = kPre ? kChebRecord : 2 * N;
};
// Q at table entry (cs, sn) from a record of either form
template <int N, typename T> struct LeanQ {
    static constexpr int RL = LeanRecord<N, T>::kLen;
    const T (&c)[RL];
    __device__ __forceinline__ explicit LeanQ(const T (&c_)[RL]) : c(c_) {}
    __device__ __forceinline__ T operator()(T cs, T sn) const
    {
        if constexpr (LeanRecord<N, T>::kPre) {
            const T A = fma(fma(fma(c[3], cs, c[2]), cs, c[1]), cs, c[0]);
            const T B = fma(fma(c[6], cs, c[5]), cs, c[4]);
            return fma(sn, B, A);
        } else {
            return null_spectrum<N, T>(c, cs, sn);
        }
    }
};

template <int N, int CH, typename T, bool MULTI, bool PEAKS>
__device__ __forceinline__ void lean_scan_item_irregular(const LeanQ<N, T> &Q, const T *__restrict__ ztab,
                                                         float *__restrict__ row, const float *__restrict__ xs,
                                                         float *__restrict__ pk_val_item, float *__restrict__ pk_loc_item,
                                                         int M, int lane);

// PEAKS = false: spectrum only (the stand-alone MUSIC_lin_array block: same arithmetic, hence the same bits, as the
// pipeline's kernel); a template parameter rather than nullable pointers, which cost the hot path 30 VGPRs.
// STORE = false (with PEAKS): the angles-only mode of the pipeline -- nobody wants the spectrum (the fused block with only
// its angle port connected), so the row is neither converted to dB (num_max_vals == 1: the answer is the first angle
// whose rounded reciprocal ties with the maximum, and its value is 0 dB by construction) nor written; rows on the
// irregular path use `row` as scratch.
// ABL (lab builds only, make LAB=1): 1 = everything but the row stores, 2 = the row stores only (results invalid) -- the
// two ablations behind the "what bounds this kernel" numbers in DESIGN.md
template <int N, int CH, typename T, bool MULTI, bool PEAKS = true, bool STORE = true, int ABL = 0>
__device__ __forceinline__ void lean_scan_item(const T (&c)[LeanRecord<N, T>::kLen], const T (&zr)[CH][4], const T (&zi)[CH][4],
                                               const T *__restrict__ ztab, float *__restrict__ row,
                                               const float *__restrict__ xs, float *__restrict__ pk_val_item,
                                               float *__restrict__ pk_loc_item, int M, int lane, float *lds_row = nullptr)
{
    constexpr int P = 256 * CH;
    const LeanQ<N, T> Q(c);
    if constexpr (ABL == 2) {
#pragma unroll
        for (int j = 0; j < CH; j++)
            store_f4<true>(reinterpret_cast<float4 *>(row + 4 * lane + 256 * j), make_float4((float)lane, (float)j, -1.f, -2.f));
        if (lane == 0) { pk_val_item[0] = 0.0f; pk_loc_item[0] = xs[lane]; }
        return;                 // (nothing here depends on the item's record)
    }
    // pass 1: the null spectrum itself (no reciprocal), the minimum of each lane's 4-angle groups, the item minimum
    float qf[CH][4], cm[CH];
#pragma unroll
    for (int j = 0; j < CH; j++) {
#pragma unroll
        for (int e = 0; e < 4; e++) qf[j][e] = (float)Q(zr[j][e], zi[j][e]);
        cm[j] = fminf(fminf(qf[j][0], qf[j][1]), fminf(qf[j][2], qf[j][3]));
    }
    float mn = cm[0];
#pragma unroll
    for (int j = 1; j < CH; j++) mn = fminf(mn, cm[j]);
    mn = wave_allreduce_min(mn);
    if (lean_norm_ok(mn)) {
        const LeanNorm nrm(mn);
        if constexpr (MULTI) {
            // num_max_vals > 1: the dB row goes to HBM and, for the peak pick, into this wave's LDS row (position p at word
            // p + p / BS, BS = P / 64 positions per lane), where find_local_max runs in its lane-blocked form
            // (peak_device.hpp: every lane walks its own BS positions; 14.7 -> 10.6 us per 4096 items at P = 1024, M = 2 against
            // the register-resident peak_pick<CH> this replaced)
            constexpr int BS = P / 64, LOG_BS = (BS == 4) ? 2 : (BS == 8) ? 3 : 4;
#pragma unroll
            for (int j = 0; j < CH; j++) {
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    bool tie;
                    qf[j][e] = nrm.db(qf[j][e], tie);
                }
                if constexpr (STORE)
                    store_f4<true>(reinterpret_cast<float4 *>(row + 4 * lane + 256 * j),
                                   make_float4(qf[j][0], qf[j][1], qf[j][2], qf[j][3]));
                if constexpr (PEAKS) {
                    const int p0 = 4 * lane + 256 * j;                 // 4 consecutive positions never straddle a block
                    float *d = lds_row + p0 + (p0 >> LOG_BS);
                    d[0] = qf[j][0]; d[1] = qf[j][1]; d[2] = qf[j][2]; d[3] = qf[j][3];
                }
            }
            if constexpr (PEAKS) {
                struct PaddedRow {
                    const float *r, *mine;
                    __device__ __forceinline__ float operator()(int p) const { return r[p + (p >> LOG_BS)]; }
                    __device__ __forceinline__ float blk(int i) const { return mine[i + (i >> LOG_BS)]; }
                };
                // written position by position, read block by block (same wave: LDS operations of one wave complete in
                // order, the fences only keep the compiler from moving them)
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                peak_pick_stream<true, BS>(PaddedRow{lds_row, lds_row + (BS + 1) * lane}, P, M, xs, pk_val_item, pk_loc_item, lane);
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            }
        } else {
            int pos = INT_MAX;
            float keep = 0.f;
#pragma unroll
            for (int j = 0; j < CH; j++) {
                typedef float v2f __attribute__((ext_vector_type(2)));
                float db[4];
                if constexpr (STORE) {
                    v2f a = {qf[j][0], qf[j][1]}, b = {qf[j][2], qf[j][3]};
                    a *= nrm.inv_up; b *= nrm.inv_up;                            // v_pk_mul_f32
                    a = v2f{__log2f(a.x), __log2f(a.y)}; b = v2f{__log2f(b.x), __log2f(b.y)};
                    a *= -kDbPerLog2; b *= -kDbPerLog2;                          // v_pk_mul_f32
                    db[0] = a.x; db[1] = a.y; db[2] = b.x; db[3] = b.y;
                }
                // lanes that hold a tied angle in this chunk (wave-uniform mask, zero for most chunks)
                const unsigned long long tied_lanes = __builtin_amdgcn_ballot_w64(cm[j] <= nrm.q_hi);
                if (tied_lanes != 0ull) {
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        // the tie compare lands in an SGPR pair: the position search is scalar work, and the same mask drives
                        // the select that puts exactly 0 dB on the tied angles (written as `tie ? 0.0f : d` the compiler issues
                        // a second, inverted compare per angle)
                        const unsigned long long at_max = __builtin_amdgcn_ballot_w64(qf[j][e] <= nrm.q_hi);
                        const int cand = at_max ? (4 * (int)__builtin_ctzll(at_max) + 256 * j + e) : INT_MAX;
                        pos = min(pos, cand);
                        if constexpr (STORE) asm("v_cndmask_b32_e64 %0, %1, 0, %2" : "=v"(db[e]) : "v"(db[e]), "s"(at_max));
                    }
                }
                if constexpr (STORE && ABL == 1) keep += (db[0] + db[1]) + (db[2] + db[3]);
                else if constexpr (STORE)
                    store_f4<true>(reinterpret_cast<float4 *>(row + 4 * lane + 256 * j), make_float4(db[0], db[1], db[2], db[3]));
            }
            if constexpr (ABL == 1) { if (keep == 12345.678f) row[lane] = keep; }      // keeps the arithmetic alive; never true
            // (the minimum itself always ties, so pos is a valid angle; the clamp only keeps a broken invariant from
            // turning into a wild address)
            if constexpr (PEAKS) {
                if (lane == 0) { pk_val_item[0] = 0.0f; pk_loc_item[0] = xs[min(pos, P - 1)]; }
            }
        }
    } else {
        lean_scan_item_irregular<N, CH, T, MULTI, PEAKS>(Q, ztab, row, xs, pk_val_item, pk_loc_item, M, lane);
    }
}

// rare (a row whose minimum of Q is not a usable positive float: non-finite input, in practice): follow the general
// semantics (db_from_ratio, arma index_max / the general peak pick) in ROLLED loops that recompute Q from the table --
// nothing of this path may cost the hot path a register (unrolled, its sixteen IEEE divisions took the kernel from 112
// to 152 VGPRs, i.e. from four to three waves per SIMD next to the covariance kernel's 120-VGPR waves: 2.5 us per
// pipeline step)
template <int N, int CH, typename T, bool MULTI, bool PEAKS>
__device__ __forceinline__ void lean_scan_item_irregular(const LeanQ<N, T> &Q, const T *__restrict__ ztab,
                                                         float *__restrict__ row, const float *__restrict__ xs,
                                                         float *__restrict__ pk_val_item, float *__restrict__ pk_loc_item,
                                                         int M, int lane)
{
    constexpr int P = 256 * CH;
    auto q_at = [&](int i) { return (float)Q((T)ztab[2 * i], (T)ztab[2 * i + 1]); };
    float mx = -INFINITY;
#pragma unroll 1
    for (int k = 0; k < 4 * CH; k++) mx = fmaxf(mx, 1.0f / q_at(4 * lane + 256 * (k >> 2) + (k & 3)));
    mx = wave_allreduce_max(mx);
    const float inv_mx = __builtin_amdgcn_rcpf(mx);
    float bv = 0.f;
    int bi = INT_MAX;
    float v0 = 0.f;
#pragma unroll 1
    for (int k = 0; k < 4 * CH; k++) {
        const int i = 4 * lane + 256 * (k >> 2) + (k & 3);
        const float db = db_from_ratio(1.0f / q_at(i), mx, inv_mx);
        row[i] = db;
        if (i == 0) v0 = db;
        if (db > -INFINITY && cand_better(db, i, bv, bi)) { bv = db; bi = i; }
    }
    if constexpr (!PEAKS) return;
    if constexpr (MULTI) {
        // the general peak pick on the row just written (same wave: its stores are visible to its loads behind the fence)
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "agent");
        peak_pick_stream([&](int p) { return __builtin_nontemporal_load(row + p); }, P, M, xs, pk_val_item, pk_loc_item, lane);
        return;
    }
    wave_argbest(bv, bi);
    v0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v0), 0));
    if (lane == 0) {
        pk_val_item[0] = (bi == INT_MAX) ? v0 : bv;
        pk_loc_item[0] = xs[(bi == INT_MAX) ? 0 : bi];
    }
}

// this lane's 4*CH table entries z_i = (cos psi_i, sin psi_i), i = 4*lane + 256*j + e, into registers
template <int CH, typename T>
__device__ __forceinline__ void lean_load_table(const T *__restrict__ ztab, int lane, T (&zr)[CH][4], T (&zi)[CH][4])
{
#pragma unroll
    for (int j = 0; j < CH; j++)
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const int i = 4 * lane + 256 * j + e;
            zr[j][e] = ztab[2 * i]; zi[j][e] = ztab[2 * i + 1];
        }
    // the table must have ARRIVED before the item loop: otherwise the compiler's wait for it sits inside the loop, where
    // (gfx9 has one counter for loads and stores) it also waits for the previous item's four row stores, i.e. every wave
    // would drain its stores before the second half of every pass 1
#pragma unroll
    for (int j = 0; j < CH; j++)
#pragma unroll
        for (int e = 0; e < 4; e++) asm volatile("" :: "v"(zr[j][e]), "v"(zi[j][e]));
}

template <int N, int CH, typename T, bool MULTI = false, bool PEAKS = true, bool STORE = true, int ABL = 0>
__global__ __launch_bounds__(256) void music_scan_peak1_kernel(const T *__restrict__ coef, const T *__restrict__ ztab,
                                                               float *__restrict__ spec, int n_items,
                                                               const float *__restrict__ xaxis, float *__restrict__ pk_val,
                                                               float *__restrict__ pk_loc, int M)
{
    constexpr int P = 256 * CH;
    constexpr int RL = LeanRecord<N, T>::kLen;
    __shared__ float xs[P];
    __shared__ float prow[(MULTI && PEAKS) ? 4 : 1][(MULTI && PEAKS) ? P + 64 + 4 : 1];     // peak-pick rows, one per wave
    if constexpr (PEAKS) {
        for (int i = threadIdx.x; i < P; i += blockDim.x) xs[i] = xaxis[i];
        __syncthreads();
    }
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(blockIdx.x * (blockDim.x / kWave) + (threadIdx.x / kWave));
    const int n_waves = gridDim.x * (blockDim.x / kWave);
    float *lds_row = (MULTI && PEAKS) ? prow[threadIdx.x / kWave] : nullptr;
    T zr[CH][4], zi[CH][4];
    lean_load_table<CH, T>(ztab, lane, zr, zi);
    // coefficient records arrive through scalar loads; the next item's record is requested before this item's
    // arithmetic so that its latency hides behind it.  (Measured and not kept, all within +-2 % of this form per 262144 items:
    // requests two items ahead; the request and its s_waitcnt placed by hand at the two ends of the item -- the compiler puts
    // the wait behind pass 1 --; sixteen consecutive items per wave with the records staged through LDS; the last chunk of the
    // table in LDS for five waves per SIMD.  DESIGN.md section 3.)
    T c[RL], c_next[RL];
    if (wave < n_items) {
#pragma unroll
        for (int k = 0; k < RL; k++) c_next[k] = coef[(size_t)wave * RL + k];
    }
    for (int item = wave; item < n_items; item += n_waves) {
#pragma unroll
        for (int i = 0; i < RL; i++) c[i] = c_next[i];
        const int nxt = item + n_waves;
        if (nxt < n_items) {
#pragma unroll
            for (int i = 0; i < RL; i++) c_next[i] = coef[(size_t)nxt * RL + i];
        }
        lean_scan_item<N, CH, T, MULTI, PEAKS, STORE, ABL>(c, zr, zi, ztab, spec + (size_t)item * P, xs, pk_val + (size_t)item * M,
                                                          pk_loc + (size_t)item * M, M, lane, lds_row);
    }
}

// Long spectra (P % 4 == 0, any length): one wave per item, 4 consecutive angles per lane per 256-angle
// chunk, chunks in a ROLLED loop, Q parked as float in the output row between the two passes (minimum first,
// then dB in place), so that nothing of the row lives in registers: ~100 VGPRs where the unrolled CH = 16 kernel needs 430-511 (one wave per
// SIMD).  The table is re-read from L2 per chunk; coefficients sit in SGPRs.  Same arithmetic as the lean
// kernel (one transcendental per angle); rows whose minimum of Q is not a finite positive number follow the
// general semantics (db_from_ratio) in a third form of the second pass.
template <int N, typename T>
__global__ __launch_bounds__(256) void music_scan_stream_kernel(const T *__restrict__ coef, const T *__restrict__ ztab,
                                                                float *__restrict__ spec, int P, int n_items, int n_ant)
{
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(blockIdx.x * (blockDim.x / kWave) + (threadIdx.x / kWave));
    const int n_waves = gridDim.x * (blockDim.x / kWave);
    const int rec = 2 * n_ant;
    const int n_chunks = (P + 255) / 256;
    for (int item = wave; item < n_items; item += n_waves) {
        T c[2 * N];
#pragma unroll
        for (int k = 0; k < 2 * N; k++) c[k] = (k < rec - 1) ? coef[(size_t)item * rec + k] : (T)0;
        auto q4 = [&](int j, float (&q)[4]) -> bool {
            const int i0 = 4 * lane + 256 * j;
            const bool live = i0 < P;
            if (live) {
                const T *zp = ztab + 2 * (size_t)i0;
#pragma unroll
                for (int e = 0; e < 4; e++) q[e] = (float)null_spectrum<N, T>(c, zp[2 * e], zp[2 * e + 1]);
            }
            return live;
        };
        // pass 1: Q as float parked in the output row itself (this lane re-reads exactly the addresses it wrote),
        // and its minimum; pass 2 turns the row into dB in place -- the (N-1)-step double Horner runs once per angle
        float *row = spec + (size_t)item * P;
        float mn = INFINITY;
#pragma unroll 1
        for (int j = 0; j < n_chunks; j++) {
            float q[4];
            if (q4(j, q)) {
                mn = fminf(fminf(mn, fminf(q[0], q[1])), fminf(q[2], q[3]));
                *reinterpret_cast<float4 *>(row + 4 * lane + 256 * j) = make_float4(q[0], q[1], q[2], q[3]);
            }
        }
        mn = wave_allreduce_min(mn);
        auto reload = [&](int j, float (&q)[4]) -> bool {
            const int i0 = 4 * lane + 256 * j;
            if (i0 >= P) return false;
            const float4 t = *reinterpret_cast<const float4 *>(row + i0);
            q[0] = t.x; q[1] = t.y; q[2] = t.z; q[3] = t.w;
            return true;
        };
        if (lean_norm_ok(mn)) {
            const LeanNorm nrm(mn);
#pragma unroll 1
            for (int j = 0; j < n_chunks; j++) {
                float q[4];
                if (reload(j, q)) {
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        bool tie;
                        q[e] = nrm.db(q[e], tie);
                    }
                    store_f4<true>(reinterpret_cast<float4 *>(row + 4 * lane + 256 * j), make_float4(q[0], q[1], q[2], q[3]));
                }
            }
        } else {
            float mx = -INFINITY;
#pragma unroll 1
            for (int j = 0; j < n_chunks; j++) {
                float q[4];
                if (reload(j, q)) {
#pragma unroll
                    for (int e = 0; e < 4; e++) mx = fmaxf(mx, 1.0f / q[e]);
                }
            }
            mx = wave_allreduce_max(mx);
            const float inv_mx = __builtin_amdgcn_rcpf(mx);
#pragma unroll 1
            for (int j = 0; j < n_chunks; j++) {
                float q[4];
                if (reload(j, q)) {
#pragma unroll
                    for (int e = 0; e < 4; e++) q[e] = db_from_ratio(1.0f / q[e], mx, inv_mx);
                    store_f4<true>(reinterpret_cast<float4 *>(row + 4 * lane + 256 * j), make_float4(q[0], q[1], q[2], q[3]));
                }
            }
        }
    }
}

// Long spectra with the peak pick fused (2048 < P <= 4096, P % 64 == 0, double): one wave per item, one angle per
// lane per step (position p = 64 g + lane), the item's row staged in LDS (16 KiB per wave, 64 KiB per workgroup:
// two workgroups per CU).  Pass 1 leaves Q as float in the LDS row and finds its minimum; pass 2 turns the row into
// dB in place and writes it to HBM ONCE; find_local_max then runs on the LDS row (peak_pick_stream, the same masks-in-
// SGPRs code as the stand-alone K5).  The two-launch form this replaces (music_scan_stream_kernel parks Q in the
// output row, then find_local_max_stream_kernel reads the row back) moves the 4P-byte row through HBM four times:
// write, read, write, read.  Rolled loops: ~100 VGPRs where a register-resident row plus peak_pick<16> needs 511.
template <int N, typename T, int ABL = 0>
__global__ __launch_bounds__(256) void music_scan_peak_long_kernel(const T *__restrict__ coef, const T *__restrict__ ztab,
                                                                   float *__restrict__ spec, int P, int n_items, int n_ant,
                                                                   const float *__restrict__ xaxis, float *__restrict__ pk_val,
                                                                   float *__restrict__ pk_loc, int M)
{
    constexpr int PMAX = 4096;
    // 66 624 B of rows + the records below = 66-68 KiB of static LDS per workgroup: two workgroups (8 waves) per CU.  That is more
    // than the 64 KiB a workgroup gets on the CDNA parts before gfx950 (160 KiB per CU here): this library is written for gfx950
    // only, and a build for another target must fail here rather than in the linker's resource check.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "music_scan_peak_long_kernel needs more than 64 KiB of LDS per workgroup: gfx950 (MI355X) only"
#endif
    // Position p lives at word p + (p >> 6): the passes below touch
    // the row as 64 g + lane, the peak pick as 64 lane + i -- both conflict-free with one pad word per 64
    // (Round 4: two waves per item sharing one row -- four waves per SIMD instead of two, three workgroup barriers per item --
    // was built and is NOT faster: 65.0 against 63.2 us per 4096 items at N = 16, P = 4096, 45.7 against 40.7 at N = 8;
    // tools/lab/scan_long_two_waves_variant.diff.txt, profiles/r04_lab_long_scan_two_waves.txt.  Occupancy is not what holds it.)
    __shared__ float rows[4][PMAX + PMAX / 64 + 4];      // (+4: the peak pick may read position P itself)
    const int lane = threadIdx.x & (kWave - 1);
    const int wib = threadIdx.x / kWave;
    const int wave = __builtin_amdgcn_readfirstlane(blockIdx.x * (blockDim.x / kWave) + wib);
    const int n_waves = gridDim.x * (blockDim.x / kWave);
    const int rec = 2 * n_ant;
    const int G = P >> 6;
    float *lrow = rows[wib];
    // The record is wave-uniform, but it must NOT travel through scalar registers: 2N - 1 doubles are 62 SGPRs at N = 16, the
    // compiler spills them to lanes of a VGPR and reads every coefficient back with v_readlane inside the Horner steps (84 us
    // per 4096 items of 4096 angles).  Lanes 0..rec-2 fetch the record with ONE coalesced load (zero beyond it: the
    // polynomial is compiled for N >= n_ant), park it in LDS, and every lane reads all of it back into 2(2N - 1) VGPRs;
    // two waves per SIMD (the LDS rows decide that) leave 256 VGPRs per wave.
    __shared__ T recs[4][2 * N];
    auto fetch_record = [&](int it) -> T { return (it < n_items && lane < rec - 1) ? coef[(size_t)it * rec + lane] : (T)0; };
    T rec_next = fetch_record(wave);
    for (int item = wave; item < n_items; item += n_waves) {
        T c[2 * N];
        if (lane < 2 * N) recs[wib][lane] = rec_next;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
        for (int k = 0; k < 2 * N; k++) c[k] = recs[wib][k];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        rec_next = fetch_record(item + n_waves);
        // pass 1: Q (float) into the LDS row, and its minimum.  UZ angles per lane go through their Horner chains together
        // (independent chains: the FP64 pipe is busy with two waves per SIMD), the table entries of the NEXT UZ steps are
        // requested before the arithmetic of these starts.
        float mn = INFINITY;
        constexpr int UZ = 4;
        T zr[UZ], zi[UZ], nr[UZ], ni[UZ];
        auto fetch = [&](int g0, T (&a)[UZ], T (&b)[UZ]) {
#pragma unroll
            for (int u = 0; u < UZ; u++) {
                const int i = min(64 * (g0 + u) + lane, P - 1);
                a[u] = ztab[2 * (size_t)i]; b[u] = ztab[2 * (size_t)i + 1];
            }
        };
        fetch(0, nr, ni);
        for (int g0 = 0; g0 < G; g0 += UZ) {
#pragma unroll
            for (int u = 0; u < UZ; u++) { zr[u] = nr[u]; zi[u] = ni[u]; }
            if (g0 + UZ < G) fetch(g0 + UZ, nr, ni);
            float q[UZ];
            if constexpr (ABL & 1) {                      // lab: no Horner
#pragma unroll
                for (int u = 0; u < UZ; u++) q[u] = (float)(zr[u] + c[0]) + 2.0f;
            } else null_spectrum_multi<N, T, UZ>(c, zr, zi, q);
#pragma unroll
            for (int u = 0; u < UZ; u++) {
                if (g0 + u < G) {
                    lrow[65 * (g0 + u) + lane] = q[u];
                    mn = fminf(mn, q[u]);
                }
            }
        }
        mn = wave_allreduce_min(mn);
        float *grow = spec + (size_t)item * P;
        // pass 2: dB in place (each lane rewrites exactly the positions it wrote), one HBM write of the row
        if constexpr (ABL & 2) { if (mn == 12345.678f) grow[lane] = mn; }
        else if (lean_norm_ok(mn)) {
            const LeanNorm nrm(mn);
#pragma unroll 4
            for (int g = 0; g < G; g++) {
                bool tie;
                const float d = nrm.db(lrow[65 * g + lane], tie);
                lrow[65 * g + lane] = d;
                __builtin_nontemporal_store(d, grow + 64 * g + lane);
            }
        } else {
            float mx = -INFINITY;
#pragma unroll 4
            for (int g = 0; g < G; g++) mx = fmaxf(mx, 1.0f / lrow[65 * g + lane]);
            mx = wave_allreduce_max(mx);
            const float inv_mx = __builtin_amdgcn_rcpf(mx);
#pragma unroll 4
            for (int g = 0; g < G; g++) {
                const float d = db_from_ratio(1.0f / lrow[65 * g + lane], mx, inv_mx);
                lrow[65 * g + lane] = d;
                __builtin_nontemporal_store(d, grow + 64 * g + lane);
            }
        }
        // the row was written lane by lane; the peak pick reads it across lanes (same wave: LDS operations of one wave
        // complete in order, the fence only keeps the compiler from moving them)
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        if (!(ABL & 4) && M > 0) {                   // M == 0: spectrum only (the stand-alone MUSIC_lin_array block)
        struct PaddedRow {
            const float *r, *mine;                         // mine = this lane's block
            __device__ __forceinline__ float operator()(int p) const { return r[p + (p >> 6)]; }
            __device__ __forceinline__ float blk(int i) const { return mine[i + (i >> 6)]; }
        };
        peak_pick_stream<true>(PaddedRow{lrow, lrow + 65 * lane}, P, M, xaxis, pk_val + (size_t)item * M, pk_loc + (size_t)item * M, lane);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
}

// Any P: one wave per item, one angle per lane per step, two passes (max, then write).
template <int N, typename T>
__global__ __launch_bounds__(256) void music_scan_generic_kernel(const T *__restrict__ coef, const T *__restrict__ ztab,
                                                                 float *__restrict__ spec, float *__restrict__ qout, int P,
                                                                 int n_items, int n_ant)
{
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(blockIdx.x * (blockDim.x / kWave) + (threadIdx.x / kWave));
    const int n_waves = gridDim.x * (blockDim.x / kWave);
    const int rec = 2 * n_ant;
    for (int item = wave; item < n_items; item += n_waves) {
        const T *co = coef + (size_t)item * rec;
        T c[2 * N];
#pragma unroll
        for (int k = 0; k < 2 * N; k++) c[k] = (k < rec - 1) ? co[k] : (T)0;
        float mx = -INFINITY;
        for (int i = lane; i < P; i += kWave) {
            const float q = (float)null_spectrum<N, T>(c, ztab[2 * i], ztab[2 * i + 1]);
            if (qout) qout[(size_t)item * P + i] = q;
            mx = fmaxf(mx, 1.0f / q);
        }
        mx = wave_allreduce_max(mx);
        const float inv_mx = __builtin_amdgcn_rcpf(mx);
        for (int i = lane; i < P; i += kWave) {
            const float o = 1.0f / (float)null_spectrum<N, T>(c, ztab[2 * i], ztab[2 * i + 1]);
            spec[(size_t)item * P + i] = db_from_ratio(o, mx, inv_mx);
        }
    }
}


template <int N, int CH, typename T, bool ZREG = true>
static void launch_scan_fast(dim3 grid, dim3 block, hipStream_t st, const T *co, const T *z, float *sp, float *q, int P,
                             int n_items, const ScanPeakArgs &pk, int n_ant)
{
    // three variants: diagnostics (Q out), plain, fused with the peak pick; spectra are write-once ->
    // non-temporal stores in the two production variants
    if (q)
        hipLaunchKernelGGL((music_scan_kernel<N, CH, T, true, false, false, ZREG>), grid, block, 0, st, co, z, sp, q, P,
                           n_items, nullptr, nullptr, nullptr, 0, n_ant);
    else if (pk.val)
        hipLaunchKernelGGL((music_scan_kernel<N, CH, T, false, true, true, ZREG>), grid, block, 0, st, co, z, sp, q, P,
                           n_items, pk.xaxis, pk.val, pk.loc, pk.M, n_ant);
    else
        hipLaunchKernelGGL((music_scan_kernel<N, CH, T, false, false, true, ZREG>), grid, block, 0, st, co, z, sp, q, P,
                           n_items, nullptr, nullptr, nullptr, 0, n_ant);
}

// returns true when the fused peak pick ran (fast path only)
template <int N, typename T>
static bool launch_scan_nt(const T *co, const T *z, int P, int n_items, void *d_spec, void *d_q, const ScanPeakArgs &pk,
                           int n_ant, hipStream_t st)
{
    float *sp = (float *)d_spec, *q = (float *)d_q;
    const int waves_per_block = 4;
    const bool aligned = (P % 4 == 0) && (reinterpret_cast<uintptr_t>(d_spec) % 16 == 0);
    // two items per wave at the benchmark batch: the z table is loaded once per wave and the next
    // item's coefficient record is prefetched behind the current item's arithmetic
    int blocks = (n_items + waves_per_block - 1) / waves_per_block;
    const int wpc = DOA_LAB_ENV_INT("DOA_SCAN_WAVES_PER_CU", 8);
    const int max_blocks = cu_count() * wpc / waves_per_block;
    if (blocks > max_blocks) blocks = max_blocks;
    dim3 grid(blocks), block(waves_per_block * kWave);
    // the lean benchmark-shape kernel (see music_scan_peak1_kernel); also without a peak pick -- the stand-alone
    // MUSIC_lin_array block -- so that block and pipeline produce the same spectrum bit for bit: same kernel, same
    // arithmetic.  For N <= 4 in double it reads the pre-transformed records (pk.cheb), which the caller must supply.
    constexpr bool kPre = LeanRecord<N, T>::kPre;
    if (aligned && !q && n_ant == N && (P == 256 || P == 512 || P == 1024) && (!kPre || pk.cheb)) {
        const T *rec = kPre ? static_cast<const T *>(pk.cheb) : co;
        // lab: workgroup size in waves, 1..4 (the kernels are bounded to 256 threads and their per-wave LDS rows to four)
        const int lwpb_env = DOA_LAB_ENV_INT("DOA_SCAN_WPB", waves_per_block);
        const int lwpb = lwpb_env < 1 ? 1 : (lwpb_env > 4 ? 4 : lwpb_env);
        const dim3 lblock(lwpb * kWave);
        int lb = (n_items + lwpb - 1) / lwpb;
        // Waves per CU: 12 up to a few items per wave (the benchmark batch: 2048 waves with two items each; one item per wave
        // makes every wave pay the 16 KiB table load for 4 KiB of output), 16 -- all the 104-VGPR kernel can hold -- beyond:
        // at large batches the kernel sits between its two ablations (rows stored without arithmetic, arithmetic without
        // stores: DESIGN.md section 3) and a fourth wave per SIMD is worth 225-229 against 232-240 us per 262144 items.
        const int per_wave16 = n_items / (cu_count() * 16);
        const int lwpc_env = DOA_LAB_ENV_INT("DOA_SCAN_LEAN_WAVES_PER_CU", 0);
        const int lwpc = lwpc_env > 0 ? lwpc_env : (per_wave16 >= 4 ? 16 : 12);
        const int cap = cu_count() * lwpc / lwpb;
        if (lb > cap && !DOA_LAB_ENV_INT("DOA_SCAN_NOTRIM", 0)) {
            // every wave takes the same number of items (4096 items on a cap of 3072 waves would be one round of 3072 and a
            // second of 1024 with two thirds of the chip idle: 2048 waves with two items each instead)
            const int cap_waves = cap * lwpb;
            const int per_wave = (n_items + cap_waves - 1) / cap_waves;
            const int waves = (n_items + per_wave - 1) / per_wave;
            lb = (waves + lwpb - 1) / lwpb;
        } else if (lb > cap) lb = cap;
        dim3 lgrid(lb);
#define DOA_LEAN_LAUNCH(CH_, MULTI_, PEAKS_, STORE_)                                                               \
    hipLaunchKernelGGL((music_scan_peak1_kernel<N, CH_, T, MULTI_, PEAKS_, STORE_>), lgrid, lblock, 0, st, rec, z, sp,  \
                       n_items, pk.xaxis, pk.val, pk.loc, pk.M)
#define DOA_LEAN_CH(MULTI_, PEAKS_, STORE_)                                                                        \
    do {                                                                                                           \
        if (P == 256) DOA_LEAN_LAUNCH(1, MULTI_, PEAKS_, STORE_);                                                  \
        else if (P == 512) DOA_LEAN_LAUNCH(2, MULTI_, PEAKS_, STORE_);                                             \
        else DOA_LEAN_LAUNCH(4, MULTI_, PEAKS_, STORE_);                                                           \
    } while (0)
#ifdef DOA_LAB
        if constexpr (N == 4 && sizeof(T) == 8) {
            // ablations of the graded kernel (results invalid): DOA_SCAN_ABLATE=1 no row stores, =2 row stores only
            const int ablate = DOA_LAB_ENV_INT("DOA_SCAN_ABLATE", 0);
            if (ablate && P == 1024 && pk.val && pk.store && pk.M == 1) {
                if (ablate == 1)
                    hipLaunchKernelGGL((music_scan_peak1_kernel<4, 4, T, false, true, true, 1>), lgrid, lblock, 0, st, rec, z, sp, n_items,
                                       pk.xaxis, pk.val, pk.loc, pk.M);
                else
                    hipLaunchKernelGGL((music_scan_peak1_kernel<4, 4, T, false, true, true, 2>), lgrid, lblock, 0, st, rec, z, sp, n_items,
                                       pk.xaxis, pk.val, pk.loc, pk.M);
                return true;
            }
        }
#endif
        if (!pk.val) DOA_LEAN_CH(false, false, true);
        else if (!pk.store) {                                   // angles only (sp is scratch for irregular rows)
            if (pk.M == 1) DOA_LEAN_CH(false, true, false);
            else DOA_LEAN_CH(true, true, false);
        }
        else if (pk.M == 1) DOA_LEAN_CH(false, true, true);
        else DOA_LEAN_CH(true, true, true);
#undef DOA_LEAN_CH
#undef DOA_LEAN_LAUNCH
        return pk.val != nullptr;
    }
    // long spectra without diagnostics: P > 2048 in double.  peak_pick<16> on a register-resident row makes the fused fast
    // kernel a 511-VGPR, one-wave-per-SIMD kernel (135 us per 4096 items at N = 16); instead:
    if (aligned && P > 2048 && sizeof(T) == 8 && !q) {
        // with the peak pick wanted (the pipeline) and P a multiple of 64 up to 4096: scan + K5 in one launch, the row staged in
        // LDS and written once (the two-launch form it replaced moved the row through HBM four times)
        // (also without the peak pick -- the stand-alone MUSIC_lin_array block: same kernel, same bits, 56 against the 62 us
        // per 4096 items of the two-pass kernel below that parks Q in the output row)
        if ((!pk.val || pk.M >= 1) && P % 64 == 0 && P <= 4096) {
            const int M_pick = pk.val ? pk.M : 0;
            int fb = (n_items + waves_per_block - 1) / waves_per_block;
            if (fb > cu_count() * 2) fb = cu_count() * 2;                 // 64 KiB of LDS per workgroup: two per CU
#ifdef DOA_LAB
            if constexpr (N == 16) {
                const int abl = DOA_LAB_ENV_INT("DOA_SCAN_LONG_ABLATE", 0);
#define DOA_LONG_ABL(A_) if (abl == A_) { hipLaunchKernelGGL((music_scan_peak_long_kernel<N, T, A_>), dim3(fb), block, 0, st, co, z, sp, P, n_items, n_ant, pk.xaxis, pk.val, pk.loc, M_pick); return pk.val != nullptr; }
                DOA_LONG_ABL(1) DOA_LONG_ABL(2) DOA_LONG_ABL(4) DOA_LONG_ABL(6) DOA_LONG_ABL(7) DOA_LONG_ABL(3) DOA_LONG_ABL(5)
#undef DOA_LONG_ABL
            }
#endif
            hipLaunchKernelGGL((music_scan_peak_long_kernel<N, T>), dim3(fb), block, 0, st, co, z, sp, P, n_items, n_ant,
                               pk.xaxis, pk.val, pk.loc, M_pick);
            return pk.val != nullptr;
        }
        // lengths the fused kernel does not take: the rolled two-pass kernel; the peak pick, if any, is left to the caller
        // (returns false)
        int sb = (n_items + waves_per_block - 1) / waves_per_block;
        if (sb > cu_count() * 16 / waves_per_block) sb = cu_count() * 16 / waves_per_block;
        hipLaunchKernelGGL((music_scan_stream_kernel<N, T>), dim3(sb), block, 0, st, co, z, sp, P, n_items, n_ant);
        return false;
    }
    if (aligned && P <= 4096) {
        constexpr bool ZBIG = (sizeof(T) == 4);          // float tables fit the register file up to P = 4096
        if (P <= 256) launch_scan_fast<N, 1, T>(grid, block, st, co, z, sp, q, P, n_items, pk, n_ant);
        else if (P <= 512) launch_scan_fast<N, 2, T>(grid, block, st, co, z, sp, q, P, n_items, pk, n_ant);
        else if (P <= 1024) launch_scan_fast<N, 4, T>(grid, block, st, co, z, sp, q, P, n_items, pk, n_ant);
        else if (P <= 2048) launch_scan_fast<N, 8, T, ZBIG>(grid, block, st, co, z, sp, q, P, n_items, pk, n_ant);
        else launch_scan_fast<N, 16, T, ZBIG>(grid, block, st, co, z, sp, q, P, n_items, pk, n_ant);
        return pk.val != nullptr && q == nullptr;
    }
    hipLaunchKernelGGL((music_scan_generic_kernel<N, T>), grid, block, 0, st, co, z, sp, q, P, n_items, n_ant);
    return false;
}

template <int N> bool launch_scan_n(const MusicTables &t, int bits, int n_items, const void *d_coef, void *d_spec,
                                           void *d_q, const ScanPeakArgs &pk, hipStream_t st)
{
    if (bits == 32)
        return launch_scan_nt<N, float>((const float *)d_coef, t.d_z.as<float>(), t.P, n_items, d_spec, d_q, pk, t.N, st);
    return launch_scan_nt<N, double>((const double *)d_coef, t.d_zd.as<double>(), t.P, n_items, d_spec, d_q, pk, t.N, st);
}


}  // namespace doa
