// music_scan_lab.hpp — experimental variants of the lean K4+K5 kernel (N = 4, P = 1024, double, num_max_vals = 1, spectrum
// stored), selected at run time by DOA_SCAN_VARIANT (bit mask) for same-box A/B runs.  Included by music_scan_impl.hpp
// only when the library is built with -DDOA_LAB; nothing here is reachable in the default build.
//   bit 0  PIN    table registers forced to arrive before the item loop (no in-loop s_waitcnt on the previous stores)
//   bit 1  RARE   tie handling off the per-angle path: one compare per (lane, chunk) on the chunk's lane minimum; the
//                 per-angle compare/select and the scalar position search run only in chunks that hold a tie
//   bit 2  DPPMIN wave minimum as six v_min_f32_dpp instead of mov_dpp + canonicalise + min per step
//   bit 3  ROT    wave w takes item k n_waves + (w + 37 k) % n_waves in turn k (rows written at one moment stay one
//                 contiguous window; a wave's successive rows are no longer a power-of-two distance apart)
//   bit 4  NOSTORE / bit 5 NOCOMPUTE: ablations (results invalid): everything but the stores / stores of constants only
//   bit 6  CLOCK  (with a stamp buffer) in-kernel clock
#pragma once

namespace doa {

__device__ __forceinline__ float wave_allreduce_min_dpp(float v)
{
    // VALU write -> DPP read of the same VGPR needs 2 wait states (the assembler inserts nothing inside inline asm)
    asm volatile("s_nop 1\n\t"
                 "v_min_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\t"
                 "v_min_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\t"
                 "v_min_f32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\t"
                 "v_min_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\t"
                 "v_min_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                 "s_nop 1\n\t"
                 "v_min_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
                 "s_nop 1"
                 : "+v"(v));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

template <int N, int CH, typename T, int VAR>
__device__ __forceinline__ void lean_scan_item_lab(const T (&c)[2 * N], const T (&zr)[CH][4], const T (&zi)[CH][4],
                                                   const T *__restrict__ ztab, float *__restrict__ row,
                                                   const float *__restrict__ xs, float *__restrict__ pk_val_item,
                                                   float *__restrict__ pk_loc_item, int lane)
{
    constexpr int P = 256 * CH;
    constexpr bool RARE = (VAR & 2) != 0, DPPMIN = (VAR & 4) != 0, NOSTORE = (VAR & 16) != 0, NOCOMPUTE = (VAR & 32) != 0;
    if constexpr (NOCOMPUTE) {
        // stores only: the row pattern of the real kernel, constants as data
#pragma unroll
        for (int j = 0; j < CH; j++)
            store_f4<true>(reinterpret_cast<float4 *>(row + 4 * lane + 256 * j), make_float4((float)c[0], (float)j, -1.f, -2.f));
        if (lane == 0) { pk_val_item[0] = 0.0f; pk_loc_item[0] = xs[lane]; }
        return;
    }
    float qf[CH][4], cm[CH];
    const ChebQ<N, T> Q(c);
#pragma unroll
    for (int j = 0; j < CH; j++) {
#pragma unroll
        for (int e = 0; e < 4; e++) qf[j][e] = (float)Q(zr[j][e], zi[j][e]);
        cm[j] = fminf(fminf(qf[j][0], qf[j][1]), fminf(qf[j][2], qf[j][3]));
    }
    float mn = cm[0];
#pragma unroll
    for (int j = 1; j < CH; j++) mn = fminf(mn, cm[j]);
    mn = DPPMIN ? wave_allreduce_min_dpp(mn) : wave_allreduce_min(mn);
    if (lean_norm_ok(mn)) {
        const LeanNorm nrm(mn);
        int pos = INT_MAX;
        float acc = 0.f;
#pragma unroll
        for (int j = 0; j < CH; j++) {
            typedef float v2f __attribute__((ext_vector_type(2)));
            float db[4];
            {
                v2f a = {qf[j][0], qf[j][1]}, b = {qf[j][2], qf[j][3]};
                a *= nrm.inv_up; b *= nrm.inv_up;
                a = v2f{__log2f(a.x), __log2f(a.y)}; b = v2f{__log2f(b.x), __log2f(b.y)};
                a *= -kDbPerLog2; b *= -kDbPerLog2;
                db[0] = a.x; db[1] = a.y; db[2] = b.x; db[3] = b.y;
            }
            if constexpr (RARE) {
                // lanes of this chunk that hold at least one tied angle (wave-uniform mask; zero for three chunks of four
                // on ordinary rows): only then the per-angle rule
                const unsigned long long lanes = __builtin_amdgcn_ballot_w64(cm[j] <= nrm.q_hi);
                if (lanes != 0ull) {
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        const unsigned long long at_max = __builtin_amdgcn_ballot_w64(qf[j][e] <= nrm.q_hi);
                        const int cand = at_max ? (4 * (int)__builtin_ctzll(at_max) + 256 * j + e) : INT_MAX;
                        pos = min(pos, cand);
                        asm("v_cndmask_b32_e64 %0, %1, 0, %2" : "=v"(db[e]) : "v"(db[e]), "s"(at_max));
                    }
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const unsigned long long at_max = __builtin_amdgcn_ballot_w64(qf[j][e] <= nrm.q_hi);
                    const int cand = at_max ? (4 * (int)__builtin_ctzll(at_max) + 256 * j + e) : INT_MAX;
                    pos = min(pos, cand);
                    asm("v_cndmask_b32_e64 %0, %1, 0, %2" : "=v"(db[e]) : "v"(db[e]), "s"(at_max));
                }
            }
            if constexpr (NOSTORE) acc += db[0] + db[1] + db[2] + db[3];
            else store_f4<true>(reinterpret_cast<float4 *>(row + 4 * lane + 256 * j), make_float4(db[0], db[1], db[2], db[3]));
        }
        if constexpr (NOSTORE) { if (acc == 12345.678f) row[lane] = acc; }      // keeps the arithmetic alive, never true
        if (lane == 0) { pk_val_item[0] = 0.0f; pk_loc_item[0] = xs[min(pos, P - 1)]; }
    } else {
        lean_scan_item_irregular<N, CH, T, false, true>(c, qf, ztab, row, xs, pk_val_item, pk_loc_item, 1, lane);
    }
}

template <int N, int CH, typename T, int VAR>
__global__ __launch_bounds__(256) void music_scan_peak1_lab_kernel(const T *__restrict__ coef, const T *__restrict__ ztab,
                                                                   float *__restrict__ spec, int n_items,
                                                                   const float *__restrict__ xaxis, float *__restrict__ pk_val,
                                                                   float *__restrict__ pk_loc, unsigned long long *__restrict__ stamps)
{
    constexpr int P = 256 * CH;
    constexpr bool PIN = (VAR & 1) != 0, ROT = (VAR & 8) != 0, CLOCK = (VAR & 64) != 0;
    __shared__ float xs[P];
    for (int i = threadIdx.x; i < P; i += blockDim.x) xs[i] = xaxis[i];
    __syncthreads();
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(blockIdx.x * (blockDim.x / kWave) + (threadIdx.x / kWave));
    const int n_waves = gridDim.x * (blockDim.x / kWave);
    T zr[CH][4], zi[CH][4];
    lean_load_table<CH, T>(ztab, lane, zr, zi);
    if constexpr (PIN) {
#pragma unroll
        for (int j = 0; j < CH; j++)
#pragma unroll
            for (int e = 0; e < 4; e++) asm volatile("" :: "v"(zr[j][e]), "v"(zi[j][e]));
    }
    unsigned long long t0 = 0, r0 = 0;
    if constexpr (CLOCK) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
    auto item_of = [&](int k) {
        const int base = k * n_waves;
        return ROT ? base + (int)((unsigned)(wave + 37 * k) % (unsigned)n_waves) : base + wave;
    };
    const int turns = (n_items + n_waves - 1) / n_waves;
    T c[2 * N], c_next[2 * N];
    {
        const int it0 = item_of(0);
        if (it0 < n_items) {
#pragma unroll
            for (int k = 0; k < 2 * N; k++) c_next[k] = coef[(size_t)it0 * (2 * N) + k];
        }
    }
    for (int k = 0; k < turns; k++) {
        const int item = item_of(k);
#pragma unroll
        for (int i = 0; i < 2 * N; i++) c[i] = c_next[i];
        const int nxt = item_of(k + 1);
        if (k + 1 < turns && nxt < n_items) {
#pragma unroll
            for (int i = 0; i < 2 * N; i++) c_next[i] = coef[(size_t)nxt * (2 * N) + i];
        }
        if (item < n_items)
            lean_scan_item_lab<N, CH, T, VAR>(c, zr, zi, ztab, spec + (size_t)item * P, xs, pk_val + (size_t)item, pk_loc + (size_t)item, lane);
    }
    if constexpr (CLOCK) {
        const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
        if (lane == 0 && stamps) { stamps[2 * wave] = t1 - t0; stamps[2 * wave + 1] = r1 - r0; }
    }
}

}  // namespace doa
