// scan_proxy.hip -- lab (round 4): the lean scan kernel's two halves -- 16 angles per lane of the N = 4 double polynomial
// (6 FMA64, a conversion, a min, two multiplies and a log2 per angle) and the four 1 KiB row stores per item -- reduced to
// the minimum that reproduces its time, so that store SCHEDULES can be swept in one process:
//   ORDER  0 chunks 0..3 in order; 1 first chunk rotated by the wave index; 2 rotated by the wave's turn
//   PIPE   0 each chunk stored right after its dB values exist (the shipped schedule);
//          1 the row held in registers and stored during the NEXT item's pass 1, one chunk after each pass-1 chunk;
//          2 same, spread over pass 1 and pass 2 (after chunks 1 and 3 of each);
//          3 all four stores back to back at the end of the item
//   MAP    0 item = wave + k n_waves; 1 contiguous run of items per wave; 2 per-XCD ticket (row = 8 t + xcc);
//          3 one global ticket counter; 4 per-workgroup contiguous chunk of 64 rows taken from a global counter, rows of the
//          chunk handed to the workgroup's waves through an LDS counter
//          5 XCD-contiguous: the waves of XCD x (blockIdx & 7) stride over rows [x n/8, (x+1) n/8) only;
//          6 one 16-wave workgroup per CU, rows through an LDS counter, row = t * workgroups + blockIdx;  7 same, row = blockIdx * per + t
//          8 static, one 16-wave workgroup per CU: row = wg + workgroups * (wave_in_wg + 16 k) (a CU's rows 1 MiB apart, the fill pattern)
//          9 as 6 with the next ticket taken before the current item is computed
//   ARITH / STORE switch the halves off (STORE 2 = plain instead of nt stores); ARITH 2 = no arithmetic but a pseudo-random
//   s_sleep per item of about the arithmetic's duration (random phases without issue pressure)
//   PK     0 no peak records, 1 two plain 4-byte stores per item from lane 0 (the shipped kernel), 2 the same as nt stores
// Not product code; results are not checked for meaning (the table and the records are synthetic).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <type_traits>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));
#ifdef FP32
typedef float real_t;      // the same instruction stream in float: separates issue time from what FP64 costs otherwise (power, clocks)
#else
typedef double real_t;
#endif
#ifndef XTRA
#define XTRA 0          // extra FMA64 per angle: the shipped kernel's arithmetic alone takes 143 us per 262144 rows, this proxy's 106 at XTRA = 0
#endif

__device__ __forceinline__ float wave_min(float v)
{
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v = fminf(v, __shfl_xor(v, o));
    return v;
}

template <int STORE> __device__ __forceinline__ void st4(float *p, const float (&d)[4])
{
    f4 t = {d[0], d[1], d[2], d[3]};
    if constexpr (STORE == 1) __builtin_nontemporal_store(t, reinterpret_cast<f4 *>(p));
    if constexpr (STORE == 2) *reinterpret_cast<f4 *>(p) = t;
}

struct Params {
    const double *ztab;      // [1024][2]
    const double *coef;      // [n_items][8]
    float *spec;             // [n_items][1024]
    float *pk;               // [n_items][2]
    unsigned *counters;      // 16 words, zeroed before the launch
    int n_items;
    int coop_stride;         // MAP 10: rows of one workgroup's waves are this many rows apart
};

// one item for a wave whose first chunk is R (compile time, so that register arrays keep static indices)
template <int ARITH, int STORE, int PIPE, int R, int PK = 1, int COOP = 0>
__device__ __forceinline__ void item_body(const Params &p, int item, int lane, const real_t (&zr)[4][4], const real_t (&zi)[4][4],
                                          float (&held)[4][4], float *&held_row)
{
    real_t c[8];
    if constexpr (ARITH == 3 || ARITH == 4) {           // 3: no record loads at all (stores only); 4: arithmetic on a constant record
#pragma unroll
        for (int k = 0; k < 8; k++) c[k] = (real_t)(1.0 + 0.125 * k + 1e-9 * item);
    } else {
#ifdef SCALAR_REC
        const double *__restrict__ rec = (const double *)__builtin_assume_aligned(p.coef, 64);
        const int uitem = __builtin_amdgcn_readfirstlane(item);
#pragma unroll
        for (int k = 0; k < 8; k++) c[k] = (real_t)__builtin_nontemporal_load(rec + (size_t)uitem * 8 + k);
#else
#pragma unroll
        for (int k = 0; k < 8; k++) c[k] = (real_t)p.coef[(size_t)item * 8 + k];
#endif
    }
    float *row = p.spec + (size_t)item * 1024;
    float qf[4][4], cm[4];
    if constexpr (COOP == 3) __builtin_amdgcn_s_barrier();
    // pass 1
#pragma unroll
    for (int jj = 0; jj < 4; jj++) {
        const int j = (jj + R) & 3;
        if constexpr (ARITH == 1 || ARITH == 4) {
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const real_t cs = zr[j][e], sn = zi[j][e];
                const real_t A = fma(fma(fma(c[3], cs, c[2]), cs, c[1]), cs, c[0]);
                real_t B = fma(fma(c[6], cs, c[5]), cs, c[4]);
#pragma unroll
                for (int x = 0; x < XTRA; x++) B = fma(B, sn, c[7]);
                qf[j][e] = (float)fma(sn, B, A);
            }
        } else {
#pragma unroll
            for (int e = 0; e < 4; e++) qf[j][e] = (float)c[e] + lane;
        }
        cm[j] = fminf(fminf(qf[j][0], qf[j][1]), fminf(qf[j][2], qf[j][3]));
        if constexpr (COOP == 4) { if (jj == 0) __builtin_amdgcn_s_barrier(); }
        if constexpr (PIPE == 1 && STORE) { if (held_row) st4<STORE>(held_row + 4 * lane + 256 * j, held[j]); }
        if constexpr (PIPE == 2 && STORE) { if ((jj & 1) && held_row) { const int h = (((jj >> 1)) + R) & 3; st4<STORE>(held_row + 4 * lane + 256 * h, held[h]); } }
    }
    float mn = fminf(fminf(cm[0], cm[1]), fminf(cm[2], cm[3]));
    if constexpr (ARITH == 1 || ARITH == 4) mn = wave_min(mn);
    if constexpr (ARITH == 2) {
        // 0..127 x 64 x 8 / 16... cycles: s_sleep 7 = 448 cycles, 0..15 of them, mean 3360 cycles
        unsigned h = (unsigned)item * 2654435761u; h ^= h >> 15;
        const int n = (int)(h & 15u);
        for (int i = 0; i < n; i++) __builtin_amdgcn_s_sleep(7);
    }
    const float inv = (ARITH == 1 || ARITH == 4) ? __builtin_amdgcn_rcpf(mn) * 1.0000002f : 1.0f;
    if constexpr (COOP == 1 || COOP == 4) __builtin_amdgcn_s_barrier();            // the workgroup's waves enter pass 2 (and its stores) together
    // pass 2
#pragma unroll
    for (int jj = 0; jj < 4; jj++) {
        const int j = (jj + R) & 3;
        float db[4];
#pragma unroll
        for (int e = 0; e < 4; e++) db[e] = (ARITH == 1 || ARITH == 4) ? -3.0103f * __log2f(qf[j][e] * inv) : qf[j][e];
        if constexpr (PIPE == 2 && STORE) { if ((jj & 1) && held_row) { const int h = (2 + (jj >> 1) + R) & 3; st4<STORE>(held_row + 4 * lane + 256 * h, held[h]); } }
        if constexpr (PIPE == 0) { if constexpr (STORE) st4<STORE>(row + 4 * lane + 256 * j, db); else if (db[0] == 12345.678f) row[lane] = db[1]; }
        else {
#pragma unroll
            for (int e = 0; e < 4; e++) qf[j][e] = db[e];
        }
    }
    if constexpr (PIPE == 3) {
        if constexpr (COOP == 2) __syncthreads();
#pragma unroll
        for (int jj = 0; jj < 4; jj++) {
            const int j = (jj + R) & 3;
            if constexpr (STORE) st4<STORE>(row + 4 * lane + 256 * j, qf[j]); else if (qf[j][0] == 12345.678f) row[lane] = qf[j][1];
        }
    }
    if constexpr (PIPE == 1 || PIPE == 2 || PIPE == 4) {
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int e = 0; e < 4; e++) held[j][e] = qf[j][e];
        held_row = row;
        if constexpr (!STORE) { if (qf[0][0] == 12345.678f) row[lane] = qf[1][1]; }
    }
    if constexpr (PK == 1) { if (lane == 0) { p.pk[(size_t)item] = 0.f; p.pk[(size_t)p.n_items + item] = mn; } }
    if constexpr (PK == 2) { if (lane == 0) { __builtin_nontemporal_store(0.f, p.pk + (size_t)item); __builtin_nontemporal_store(mn, p.pk + (size_t)p.n_items + item); } }
    if constexpr (PK == 0) { if (mn == 12345.678f) p.pk[item] = mn; }
}

template <int ARITH, int STORE, int ORDER, int PIPE, int MAP, int PK = 1, int COOP = 0>
__global__ __launch_bounds__((MAP >= 6) ? 1024 : 256) __attribute__((amdgpu_waves_per_eu(4, 4))) void proxy_kernel(Params p)
{
    const int lane = threadIdx.x & 63;
    const int wpb = blockDim.x >> 6;
    const int wib = threadIdx.x >> 6;
    const int wave = __builtin_amdgcn_readfirstlane(blockIdx.x * wpb + wib);
    const int n_waves = gridDim.x * wpb;
    real_t zr[4][4], zi[4][4];
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const int i = 4 * lane + 256 * j + e;
            zr[j][e] = (real_t)p.ztab[2 * i]; zi[j][e] = (real_t)p.ztab[2 * i + 1];
        }
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
        for (int e = 0; e < 4; e++) asm volatile("" :: "v"(zr[j][e]), "v"(zi[j][e]));
    float held[4][4];
    float *held_row = nullptr;
    __shared__ unsigned s_next;
    int turn = 0;
    auto run_item = [&](int item) {
        const int r = (ORDER == 0) ? 0 : (ORDER == 1) ? (wave & 3) : ((wave + turn) & 3);
        if (ORDER == 0 || r == 0) item_body<ARITH, STORE, PIPE, 0, PK>(p, item, lane, zr, zi, held, held_row);
        else if (r == 1) item_body<ARITH, STORE, PIPE, 1>(p, item, lane, zr, zi, held, held_row);
        else if (r == 2) item_body<ARITH, STORE, PIPE, 2>(p, item, lane, zr, zi, held, held_row);
        else item_body<ARITH, STORE, PIPE, 3>(p, item, lane, zr, zi, held, held_row);
        turn++;
    };
    if constexpr (MAP == 0 && ORDER != 0) {
        // the rotation is decided once per wave, outside the item loop (four copies of the loop), so that each copy keeps
        // static register indices; ORDER 2 walks the four rotations in an unrolled group of four turns
        auto loop = [&](auto r0_tag) {
            constexpr int R0 = decltype(r0_tag)::value;
            if constexpr (ORDER == 1) {
                for (int item = wave; item < p.n_items; item += n_waves) item_body<ARITH, STORE, PIPE, R0>(p, item, lane, zr, zi, held, held_row);
            } else {
                int item = wave;
                for (; item + 3 * n_waves < p.n_items; item += 4 * n_waves) {
                    item_body<ARITH, STORE, PIPE, R0>(p, item, lane, zr, zi, held, held_row);
                    item_body<ARITH, STORE, PIPE, (R0 + 1) & 3>(p, item + n_waves, lane, zr, zi, held, held_row);
                    item_body<ARITH, STORE, PIPE, (R0 + 2) & 3>(p, item + 2 * n_waves, lane, zr, zi, held, held_row);
                    item_body<ARITH, STORE, PIPE, (R0 + 3) & 3>(p, item + 3 * n_waves, lane, zr, zi, held, held_row);
                }
                for (; item < p.n_items; item += n_waves) item_body<ARITH, STORE, PIPE, R0>(p, item, lane, zr, zi, held, held_row);
            }
        };
        const int r0 = wave & 3;
        if (r0 == 0) loop(std::integral_constant<int, 0>{});
        else if (r0 == 1) loop(std::integral_constant<int, 1>{});
        else if (r0 == 2) loop(std::integral_constant<int, 2>{});
        else loop(std::integral_constant<int, 3>{});
    } else if constexpr (MAP == 0) {
        for (int item = wave; item < p.n_items; item += n_waves) run_item(item);
    } else if constexpr (MAP == 1) {
        const int per = (p.n_items + n_waves - 1) / n_waves;
        const int last = min(p.n_items, (wave + 1) * per);
        for (int item = wave * per; item < last; item++) run_item(item);
    } else if constexpr (MAP == 2) {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(xcc));
        xcc &= 7;
        auto take = [&]() -> unsigned {
            unsigned t = 0;
            if (lane == 0) t = __hip_atomic_fetch_add(p.counters + xcc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return __builtin_amdgcn_readfirstlane(t);
        };
        unsigned nxt = take();
        while (true) {
            const unsigned item = 8u * nxt + xcc;
            if (item >= (unsigned)p.n_items) break;
            nxt = take();
            run_item((int)item);
        }
    } else if constexpr (MAP == 3) {
        auto take = [&]() -> unsigned {
            unsigned t = 0;
            if (lane == 0) t = __hip_atomic_fetch_add(p.counters, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return __builtin_amdgcn_readfirstlane(t);
        };
        unsigned nxt = take();
        while (true) {
            const unsigned item = nxt;
            if (item >= (unsigned)p.n_items) break;
            nxt = take();
            run_item((int)item);
        }
    } else if constexpr (MAP == 5) {
        const int xcd = blockIdx.x & 7;
        const int local = (blockIdx.x >> 3) * wpb + wib;
        const int per_xcd = n_waves >> 3;                 // (grids are multiples of 8 workgroups here)
        const int region = p.n_items >> 3;
        for (int i = local; i < region; i += per_xcd) run_item(xcd * region + i);
    } else if constexpr (MAP == 10) {
        // the waves of a workgroup take rows coop_stride apart inside an aligned block of coop_stride * wpb rows that
        // coop_stride consecutive workgroups share (n_items must be a multiple of the wave count: barriers inside)
        const int st = p.coop_stride;
        const int blk = blockIdx.x / st, o = blockIdx.x - blk * st;
        const int first = blk * st * wpb + o + st * wib;
        // COOP 4: the two halves of the workgroup run half an item apart (two barriers per item, the second half one barrier
        // behind): at every barrier one half starts its stores while the other is inside pass 1
        const bool late = (COOP == 4) && (wib >= wpb / 2);
        if (late) __builtin_amdgcn_s_barrier();
        for (int item = first; item < p.n_items; item += n_waves) item_body<ARITH, STORE, PIPE, 0, PK, COOP>(p, item, lane, zr, zi, held, held_row);
        if (COOP == 4 && !late) __builtin_amdgcn_s_barrier();
    } else if constexpr (MAP == 12) {
        // pairs: a wave computes rows r and r + stride one after the other, keeps both in registers and writes them together,
        // chunk-major (the same KiB of both rows back to back); no synchronisation between waves
        const int st = p.coop_stride;
        const int n_pairs = p.n_items / 2;
        for (int pr = wave; pr < n_pairs; pr += n_waves) {
            const int blk = pr / st, o = pr - blk * st;
            const int r0 = blk * 2 * st + o;
            float first_row[4][4];
            item_body<ARITH, STORE, 4, 0, PK, 0>(p, r0, lane, zr, zi, held, held_row);
#pragma unroll
            for (int j = 0; j < 4; j++)
#pragma unroll
                for (int e = 0; e < 4; e++) first_row[j][e] = held[j][e];
            item_body<ARITH, STORE, 4, 0, PK, 0>(p, r0 + st, lane, zr, zi, held, held_row);
            if constexpr (STORE != 0) {
                float *ra = p.spec + (size_t)r0 * 1024, *rb = p.spec + (size_t)(r0 + st) * 1024;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    st4<STORE>(ra + 4 * lane + 256 * j, first_row[j]);
                    st4<STORE>(rb + 4 * lane + 256 * j, held[j]);
                }
            }
        }
        held_row = nullptr;
    } else if constexpr (MAP == 8) {
        const int per = (p.n_items + gridDim.x - 1) / gridDim.x;
        for (int t = wib; t < per; t += wpb) {
            const long item = (long)t * gridDim.x + blockIdx.x;
            if (item < p.n_items) run_item((int)item);
        }
    } else if constexpr (MAP == 9) {
        if (threadIdx.x == 0) s_next = 0;
        __syncthreads();
        const int per = (p.n_items + gridDim.x - 1) / gridDim.x;
        auto take = [&]() -> unsigned {
            unsigned t = 0;
            if (lane == 0) t = atomicAdd(&s_next, 1u);
            return __builtin_amdgcn_readfirstlane(t);
        };
        unsigned cur = take();
        int guard = 0;
        while (cur < (unsigned)per) {
            const unsigned nxt = take();
            if (++guard > 3000) { if (lane == 0) atomicAdd(p.counters + 8, 1u); break; }
            const long item = (long)cur * gridDim.x + blockIdx.x;
            if (item < p.n_items) run_item((int)item);
            cur = nxt;
        }
        if (lane == 0) atomicMax(p.counters + 11, (unsigned)guard);
    } else if constexpr (MAP == 6 || MAP == 7) {
        if (threadIdx.x == 0) s_next = 0;
        __syncthreads();
        const int per = (p.n_items + gridDim.x - 1) / gridDim.x;
        int guard = 0;
        unsigned last = 0;
        while (true) {
            unsigned t = 0;
            if (lane == 0) t = atomicAdd(&s_next, 1u);
            t = __builtin_amdgcn_readfirstlane(t);
            if (t >= (unsigned)per) break;
            if (++guard > 3000) {                      // diagnosis of the round-4 hang: never more than `per` turns per wave
                if (lane == 0) { atomicAdd(p.counters + 8, 1u); atomicMax(p.counters + 9, t); atomicMax(p.counters + 10, last); }
                break;
            }
            last = t;
            const long item = (MAP == 6) ? (long)t * gridDim.x + blockIdx.x : (long)blockIdx.x * per + t;
            if (item < p.n_items) run_item((int)item);
        }
        if (lane == 0) atomicMax(p.counters + 11, (unsigned)guard);
    }
    if constexpr ((PIPE == 1 || PIPE == 2) && STORE != 0) {
        if (held_row) {
#pragma unroll
            for (int j = 0; j < 4; j++) st4<STORE>(held_row + 4 * lane + 256 * j, held[j]);
        }
    }
}

// stores only, random phases: a wave writes G rows back to back (rows base + i * stride, i < G), then sleeps a pseudo-random
// time of about G items of arithmetic.  Which (G, stride) brings the rate back to the lockstep sweep's says which rows share
// DRAM pages / what the write path wants to see close together in time.  CHUNK_MAJOR: all G rows' first KiB, then the second...
template <int G, bool CHUNK_MAJOR, bool SLEEP>
__global__ __launch_bounds__(256) void group_store_kernel(float *__restrict__ out, int n_items, int stride)
{
    const int lane = threadIdx.x & 63;
    const int wpb = blockDim.x >> 6;
    const int wave = __builtin_amdgcn_readfirstlane(blockIdx.x * wpb + (threadIdx.x >> 6));
    const int n_waves = gridDim.x * wpb;
    const int n_groups = n_items / G;
    for (int g = wave; g < n_groups; g += n_waves) {
        const int blk = g / stride, off = g - blk * stride;
        const size_t base = (size_t)blk * stride * G + off;
        if constexpr (SLEEP) {
            unsigned h = (unsigned)g * 2654435761u; h ^= h >> 15;
            const int n = (int)(h & 15u) * G;
            for (int i = 0; i < n; i++) __builtin_amdgcn_s_sleep(7);
        }
        if constexpr (CHUNK_MAJOR) {
#pragma unroll
            for (int j = 0; j < 4; j++)
#pragma unroll
                for (int i = 0; i < G; i++) {
                    f4 t = {(float)g, (float)i, (float)j, (float)lane};
                    __builtin_nontemporal_store(t, reinterpret_cast<f4 *>(out + (base + (size_t)i * stride) * 1024 + 4 * lane + 256 * j));
                }
        } else {
#pragma unroll
            for (int i = 0; i < G; i++)
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    f4 t = {(float)g, (float)i, (float)j, (float)lane};
                    __builtin_nontemporal_store(t, reinterpret_cast<f4 *>(out + (base + (size_t)i * stride) * 1024 + 4 * lane + 256 * j));
                }
        }
    }
}

// Producer / consumer inside one 16-wave workgroup per CU: NC computing waves put their dB rows into an LDS generation buffer
// (double-buffered), NW = 16 - NC writer waves -- one or more per SIMD -- drain the previous generation to HBM meanwhile: chunk-major
// over the generation's rows, which are `stride` rows apart (the page-sharing rows written together, by waves that do nothing
// else and may block at their stores as long as they like).  One workgroup barrier per generation.
template <int ARITH, int STORE, int NC>
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(4, 4))) void pc_kernel(Params p)
{
    constexpr int NW = 16 - NC;
    __shared__ f4 buf[2][NC][256];                      // NC x 4 KiB per generation
    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int st = p.coop_stride;
    const int blk = blockIdx.x / st, off = blockIdx.x - blk * st;
    const int per_gen = gridDim.x * NC;
    const int n_gen = (p.n_items + per_gen - 1) / per_gen;
    const int first = blk * st * NC + off;              // + st * row_in_generation + g * per_gen
    if (wib < NC) {
        double zr[4][4], zi[4][4];
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const int i = 4 * lane + 256 * j + e;
                zr[j][e] = p.ztab[2 * i]; zi[j][e] = p.ztab[2 * i + 1];
            }
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int e = 0; e < 4; e++) asm volatile("" :: "v"(zr[j][e]), "v"(zi[j][e]));
        for (int g = 0; g < n_gen; g++) {
            const int item = first + st * wib + g * per_gen;
            if (item < p.n_items) {
                double c[8];
#pragma unroll
                for (int k = 0; k < 8; k++) c[k] = p.coef[(size_t)item * 8 + k];
                float qf[4][4], cm[4];
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    if constexpr (ARITH) {
#pragma unroll
                        for (int e = 0; e < 4; e++) {
                            const double cs = zr[j][e], sn = zi[j][e];
                            const double A = fma(fma(fma(c[3], cs, c[2]), cs, c[1]), cs, c[0]);
                            double B = fma(fma(c[6], cs, c[5]), cs, c[4]);
#pragma unroll
                            for (int x = 0; x < XTRA; x++) B = fma(B, sn, c[7]);
                            qf[j][e] = (float)fma(sn, B, A);
                        }
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; e++) qf[j][e] = (float)c[e] + lane;
                    }
                    cm[j] = fminf(fminf(qf[j][0], qf[j][1]), fminf(qf[j][2], qf[j][3]));
                }
                float mn = fminf(fminf(cm[0], cm[1]), fminf(cm[2], cm[3]));
                if constexpr (ARITH) mn = wave_min(mn);
                const float inv = ARITH ? __builtin_amdgcn_rcpf(mn) * 1.0000002f : 1.0f;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    f4 d;
                    d.x = ARITH ? -3.0103f * __log2f(qf[j][0] * inv) : qf[j][0];
                    d.y = ARITH ? -3.0103f * __log2f(qf[j][1] * inv) : qf[j][1];
                    d.z = ARITH ? -3.0103f * __log2f(qf[j][2] * inv) : qf[j][2];
                    d.w = ARITH ? -3.0103f * __log2f(qf[j][3] * inv) : qf[j][3];
                    buf[g & 1][wib][lane + 64 * j] = d;
                }
                if (lane == 0) { p.pk[(size_t)item] = 0.f; p.pk[(size_t)p.n_items + item] = mn; }
            }
            __syncthreads();
        }
    } else {
        const int w = wib - NC;
        for (int g = 0; g < n_gen; g++) {
            __syncthreads();
            if constexpr (STORE) {
                // chunk-major: this writer takes the 1 KiB pieces q = w, w + NW, ... of the generation, piece q = chunk (q / NC)
                // of row (q % NC): consecutive pieces are the same KiB of rows `stride` apart
#pragma unroll 4
                for (int q = w; q < 4 * NC; q += NW) {
                    const int j = q / NC, r = q - j * NC;
                    const int item = first + st * r + g * per_gen;
                    if (item < p.n_items) {
                        const f4 d = buf[g & 1][r][lane + 64 * j];
                        __builtin_nontemporal_store(d, reinterpret_cast<f4 *>(p.spec + (size_t)item * 1024) + lane + 64 * j);
                    }
                }
            }
        }
    }
}

struct Ctx { Params p[2]; unsigned *counters; };

template <int ARITH, int STORE, int NC> double run_pc(Ctx &cx, int blocks)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int r = 0; r < 2; r++) hipLaunchKernelGGL((pc_kernel<ARITH, STORE, NC>), dim3(blocks), dim3(1024), 0, 0, cx.p[r & 1]);
    CK(hipDeviceSynchronize());
    const int reps = 8;
    CK(hipEventRecord(e0));
    for (int r = 0; r < reps; r++) hipLaunchKernelGGL((pc_kernel<ARITH, STORE, NC>), dim3(blocks), dim3(1024), 0, 0, cx.p[r & 1]);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    return ms * 1e3 / reps;
}

template <int G, bool CM, bool SLEEP> double run_group(Ctx &cx, int wpb, int blocks, int stride)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int r = 0; r < 2; r++) hipLaunchKernelGGL((group_store_kernel<G, CM, SLEEP>), dim3(blocks), dim3(64 * wpb), 0, 0, cx.p[r & 1].spec, cx.p[0].n_items, stride);
    CK(hipDeviceSynchronize());
    const int reps = 6;
    CK(hipEventRecord(e0));
    for (int r = 0; r < reps; r++) hipLaunchKernelGGL((group_store_kernel<G, CM, SLEEP>), dim3(blocks), dim3(64 * wpb), 0, 0, cx.p[r & 1].spec, cx.p[0].n_items, stride);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    return ms * 1e3 / reps;
}

template <int ARITH, int STORE, int ORDER, int PIPE, int MAP, int PK = 1, int COOP = 0> double run(Ctx &cx, int wpb, int blocks)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto launch = [&](int r) {
        if (MAP == 2 || MAP == 3) CK(hipMemsetAsync(cx.counters, 0, 64, 0));
        hipLaunchKernelGGL((proxy_kernel<ARITH, STORE, ORDER, PIPE, MAP, PK, COOP>), dim3(blocks), dim3(64 * wpb), 0, 0, cx.p[r & 1]);
    };
    for (int r = 0; r < 2; r++) launch(r);
    CK(hipDeviceSynchronize());
    const int reps = 8;
    CK(hipEventRecord(e0));
    for (int r = 0; r < reps; r++) launch(r);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    return ms * 1e3 / reps;
}

#include <chrono>
static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static double t_start;
#define ROWG(label, GEOS, A, S, O, PI, M, PKK) do { \
    printf("[%6.1fs] %-46s", now_s() - t_start, label); fflush(stdout); \
    for (auto g : GEOS) { const int blocks = g.blocks > 0 ? g.blocks : cus * g.wpc / g.wpb; printf(" %7.1f", run<A, S, O, PI, M, PKK>(cx, g.wpb, blocks)); fflush(stdout); } \
    printf("\n"); } while (0)
#define ROW(label, A, S, O, PI, M) ROWG(label, geos, A, S, O, PI, M, 1)

__global__ void fill_kernel(f4 *out, size_t n16)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n16; i += stride) {
        f4 t = {1.f, 2.f, 3.f, (float)i};
        __builtin_nontemporal_store(t, out + i);
    }
}

int main(int argc, char **argv)
{
    t_start = now_s();
    const int n_items = argc > 1 ? atoi(argv[1]) : 262144;
    const char *which = argc > 2 ? argv[2] : "all";
    auto want = [&](const char *g) { return !strcmp(which, "all") || strstr(which, g) != nullptr; };
    hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
    const int cus = pr.multiProcessorCount;
    Ctx cx;
    std::vector<double> z(2048), co((size_t)n_items * 8);
    for (int i = 0; i < 1024; i++) { const double th = 3.141592653589793 * i / 1024; const double psi = -3.141592653589793 * cos(th); z[2 * i] = cos(psi); z[2 * i + 1] = sin(psi); }
    for (int i = 0; i < n_items; i++) {
        double *c = &co[(size_t)i * 8];
        c[0] = 3.0 + (i % 7) * 0.01; c[1] = 0.3; c[2] = 0.2; c[3] = 0.05; c[4] = 0.1; c[5] = 0.07; c[6] = 0.02; c[7] = 0;
    }
    double *dz, *dc; float *pk;
    CK(hipMalloc(&dz, z.size() * 8)); CK(hipMalloc(&dc, co.size() * 8)); CK(hipMalloc(&pk, (size_t)n_items * 8));
    CK(hipMemcpy(dz, z.data(), z.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dc, co.data(), co.size() * 8, hipMemcpyHostToDevice));
    CK(hipMalloc(&cx.counters, 64));
    for (int i = 0; i < 2; i++) {
        float *sp; CK(hipMalloc(&sp, (size_t)n_items * 4096));
        cx.p[i] = Params{dz, dc, sp, pk, cx.counters, n_items, 1};
    }
    printf("[%6.1fs] set up: %d rows of 4 KiB, %d CUs; us per launch\n", now_s() - t_start, n_items, cus);
    {   // the box's own write ceiling, same process
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        const size_t bytes = (size_t)n_items * 4096;
        for (int r = 0; r < 2; r++) CK(hipMemsetAsync(cx.p[r].spec, 0, bytes, 0));
        CK(hipEventRecord(e0));
        for (int r = 0; r < 8; r++) CK(hipMemsetAsync(cx.p[r & 1].spec, 0, bytes, 0));
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("hipMemsetAsync: %.1f us\n", ms * 1e3 / 8);
        for (int wpc : {4, 8, 16}) {
            CK(hipEventRecord(e0));
            for (int r = 0; r < 8; r++) hipLaunchKernelGGL(fill_kernel, dim3(cus * wpc / 4), dim3(256), 0, 0, (f4 *)cx.p[r & 1].spec, bytes / 16);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1));
            printf("linear nt fill, %2d waves per CU: %.1f us\n", wpc, ms * 1e3 / 8);
        }
    }
    struct Geo { int wpb, wpc, blocks; };
    std::vector<Geo> geos = {{4, 16, 0}, {4, 12, 0}, {1, 16, 0}, {1, 12, 0}, {2, 16, 0}, {4, 0, 1023}, {4, 0, 1025}, {4, 20, 0}};
    std::vector<Geo> geos2 = {{4, 16, 0}, {4, 12, 0}, {1, 16, 0}, {1, 12, 0}, {4, 8, 0}, {1, 8, 0}};
    std::vector<Geo> geo_cu = {{16, 16, 0}};
    if (want("base") || want("order") || want("pipe") || want("map"))
        printf("columns (waves per workgroup x waves per CU): 4x16 4x12 1x16 1x12 2x16 4x1023wg 4x1025wg 4x20\n");
    if (want("base")) {
        ROW("arith only", 1, 0, 0, 0, 0);
        ROW("stores only (nt)", 0, 1, 0, 0, 0);
        ROW("stores only (plain)", 0, 2, 0, 0, 0);
        ROW("both: shipped schedule", 1, 1, 0, 0, 0);
        ROW("both: plain stores", 1, 2, 0, 0, 0);
    }
    if (want("order")) {
        ROW("stores only, first chunk by wave", 0, 1, 1, 0, 0);
        ROW("stores only, first chunk by wave+turn", 0, 1, 2, 0, 0);
        ROW("both, first chunk by wave", 1, 1, 1, 0, 0);
        ROW("both, first chunk by wave+turn", 1, 1, 2, 0, 0);
    }
    if (want("pipe")) {
        ROW("arith only, row held (pipe 1)", 1, 0, 0, 1, 0);
        ROW("both, stores in next pass 1 (pipe 1)", 1, 1, 0, 1, 0);
        ROW("both, stores over next item (pipe 2)", 1, 1, 0, 2, 0);
        ROW("both, four stores at the end (pipe 3)", 1, 1, 0, 3, 0);
        ROW("both, pipe 1 + first chunk by wave", 1, 1, 1, 1, 0);
        ROW("both, pipe 2 + first chunk by wave+turn", 1, 1, 2, 2, 0);
    }
    if (want("map")) {
        ROW("stores only, run of items per wave", 0, 1, 0, 0, 1);
        ROW("both, run of items per wave", 1, 1, 0, 0, 1);
    }
    auto dbg = [&]() {
        unsigned h[16]; CK(hipMemcpy(h, cx.counters, 64, hipMemcpyDeviceToHost));
        printf("      guard trips %u, ticket at trip %u, last good ticket %u, most turns of a wave %u\n", h[8], h[9], h[10], h[11]);
        CK(hipMemset(cx.counters, 0, 64));
    };
    if (want("r3")) {
        CK(hipMemset(cx.counters, 0, 64));
        ROWG("stores only, row = t*256 + cu, no peak records", geo_cu, 0, 1, 0, 0, 6, 0); dbg();
        ROWG("both, row = t*256 + cu, no peak records", geo_cu, 1, 1, 0, 0, 6, 0); dbg();
        ROWG("both, row = t*256 + cu, peak records nt", geo_cu, 1, 1, 0, 0, 6, 2); dbg();
        ROWG("stores only, row = t*256 + cu, peak records plain", geo_cu, 0, 1, 0, 0, 6, 1); dbg();
        ROWG("both, row = t*256 + cu, peak records plain", geo_cu, 1, 1, 0, 0, 6, 1); dbg();
        ROWG("arith only, row = t*256 + cu", geo_cu, 1, 0, 0, 0, 6, 0); dbg();
        ROWG("stores only, row = cu*1024 + t, no peak records", geo_cu, 0, 1, 0, 0, 7, 0); dbg();
        ROWG("both, row = cu*1024 + t, no peak records", geo_cu, 1, 1, 0, 0, 7, 0); dbg();
        ROWG("both, row = cu*1024 + t, peak records plain", geo_cu, 1, 1, 0, 0, 7, 1); dbg();
    }
    if (want("r4")) {
        std::vector<Geo> g416 = {{4, 16, 0}}, g112 = {{1, 12, 0}}, g116 = {{1, 16, 0}}, g816 = {{8, 16, 0}}, g1616 = {{16, 16, 0}};
        CK(hipMemset(cx.counters, 0, 64));
        for (int rep = 0; rep < 3; rep++) {
            printf("-- repetition %d\n", rep);
            ROWG("stores only, static 4x16", g416, 0, 1, 0, 0, 0, 1);
            ROWG("stores only, static 1x12", g112, 0, 1, 0, 0, 0, 1);
            ROWG("stores only, static 16x16 (item = wave + k n)", g1616, 0, 1, 0, 0, 0, 1);
            ROWG("stores only, CU-strided static (map 8)", g1616, 0, 1, 0, 0, 8, 1);
            ROWG("stores only, CU-strided LDS ticket (map 6)", g1616, 0, 1, 0, 0, 6, 1);
            ROWG("both, static 4x16", g416, 1, 1, 0, 0, 0, 1);
            ROWG("both, static 1x12", g112, 1, 1, 0, 0, 0, 1);
            ROWG("both, static 1x16", g116, 1, 1, 0, 0, 0, 1);
            ROWG("both, static 8x16", g816, 1, 1, 0, 0, 0, 1);
            ROWG("both, static 16x16 (item = wave + k n)", g1616, 1, 1, 0, 0, 0, 1);
            ROWG("both, CU-strided static (map 8)", g1616, 1, 1, 0, 0, 8, 1);
            ROWG("both, CU-strided LDS ticket (map 6)", g1616, 1, 1, 0, 0, 6, 1);
            ROWG("both, CU-strided LDS ticket, prefetched (map 9)", g1616, 1, 1, 0, 0, 9, 1);
            ROWG("both, CU-linear LDS ticket (map 7)", g1616, 1, 1, 0, 0, 7, 1);
            ROWG("arith only, static 4x16", g416, 1, 0, 0, 0, 0, 1);
            ROWG("arith only, CU-strided static (map 8)", g1616, 1, 0, 0, 0, 8, 1);
            ROWG("arith only, LDS ticket prefetched (map 9)", g1616, 1, 0, 0, 0, 9, 1);
            dbg();
        }
    }
    if (want("r5")) {
        std::vector<Geo> g = {{4, 16, 0}, {1, 12, 0}, {4, 12, 0}, {1, 8, 0}};
        printf("columns: 4x16 1x12 4x12 1x8\n");
        for (int rep = 0; rep < 2; rep++) {
            printf("-- repetition %d\n", rep);
            ROWG("stores only, no record loads, no peak records", g, 3, 1, 0, 0, 0, 0);
            ROWG("stores only, record loads, no peak records", g, 0, 1, 0, 0, 0, 0);
            ROWG("stores only, no record loads, peak records", g, 3, 1, 0, 0, 0, 1);
            ROWG("stores only, record loads, peak records", g, 0, 1, 0, 0, 0, 1);
            ROWG("both, constant record, no peak records", g, 4, 1, 0, 0, 0, 0);
            ROWG("both, record loads, no peak records", g, 1, 1, 0, 0, 0, 0);
            ROWG("both, constant record, peak records", g, 4, 1, 0, 0, 0, 1);
            ROWG("both, record loads, peak records", g, 1, 1, 0, 0, 0, 1);
        }
    }
    if (want("r6")) {
        const int strides[] = {1, 2, 4, 8, 16, 32, 64, 128, 256, 512, 1024, 4096};
        printf("groups of G rows written back to back by one wave, rows of a group `stride` rows apart; random sleeps between groups\n");
        printf("columns: stride (rows of 4 KiB) = 1 2 4 8 16 32 64 128 256 512 1024 4096\n");
#define GROW(label, G_, CM_, SL_, wpb_, wpc_) do { printf("%-52s", label); for (int st : strides) { printf(" %6.1f", run_group<G_, CM_, SL_>(cx, wpb_, cus * wpc_ / wpb_, st)); fflush(stdout); } printf("\n"); } while (0)
        GROW("G=1 1x12 no sleep (lockstep sweep)", 1, false, false, 1, 12);
        GROW("G=1 1x12 sleep", 1, false, true, 1, 12);
        GROW("G=2 1x12 sleep", 2, false, true, 1, 12);
        GROW("G=4 1x12 sleep", 4, false, true, 1, 12);
        GROW("G=8 1x12 sleep", 8, false, true, 1, 12);
        GROW("G=16 1x12 sleep", 16, false, true, 1, 12);
        GROW("G=4 1x12 sleep, chunk-major", 4, true, true, 1, 12);
        GROW("G=1 4x16 sleep", 1, false, true, 4, 16);
        GROW("G=2 4x16 sleep", 2, false, true, 4, 16);
        GROW("G=4 4x16 sleep", 4, false, true, 4, 16);
        GROW("G=8 4x16 sleep", 8, false, true, 4, 16);
        GROW("G=4 4x16 sleep, chunk-major", 4, true, true, 4, 16);
        GROW("G=4 1x12 no sleep", 4, false, false, 1, 12);
        GROW("G=4 4x16 no sleep", 4, false, false, 4, 16);
    }
    if (want("r7")) {
        const int strides[] = {1, 2, 4, 8, 16, 32, 64};
        printf("workgroup-cooperative rows (map 10): the waves of a workgroup write rows `stride` apart; columns: stride = 1 2 4 8 16 32 64\n");
#define CROW(label, A_, PI_, CO_, wpb_) do { printf("%-56s", label); for (int st : strides) { cx.p[0].coop_stride = cx.p[1].coop_stride = st; \
            printf(" %6.1f", run<A_, 1, 0, PI_, 10, 1, CO_>(cx, wpb_, cus * 16 / wpb_)); fflush(stdout); } printf("\n"); } while (0)
        for (int rep = 0; rep < 2; rep++) {
            printf("-- repetition %d\n", rep);
            ROWG("both, static 4x16 (shipped mapping)", (std::vector<Geo>{{4, 16, 0}}), 1, 1, 0, 0, 0, 1);
            ROWG("both, static 1x12", (std::vector<Geo>{{1, 12, 0}}), 1, 1, 0, 0, 0, 1);
            CROW("both 4x16, no barrier", 1, 0, 0, 4);
            CROW("both 4x16, barrier before pass 2", 1, 0, 1, 4);
            CROW("both 4x16, stores at the end, barrier before them", 1, 3, 2, 4);
            CROW("both 8x16, no barrier", 1, 0, 0, 8);
            CROW("both 8x16, barrier before pass 2", 1, 0, 1, 8);
            CROW("both 16x16, barrier before pass 2", 1, 0, 1, 16);
            CROW("both 2x16, barrier before pass 2", 1, 0, 1, 2);
            CROW("stores only 4x16, barrier", 0, 0, 1, 4);
        }
    }
    if (want("r8")) {
        const int strides[] = {4, 8, 16};
        printf("XTRA = %d extra FMA64 per angle; workgroup-cooperative rows (map 10), columns: stride = 4 8 16\n", XTRA);
#define CROW8(label, A_, S_, PI_, CO_, wpb_) do { printf("%-64s", label); for (int st : strides) { cx.p[0].coop_stride = cx.p[1].coop_stride = st; \
            printf(" %6.1f", run<A_, S_, 0, PI_, 10, 1, CO_>(cx, wpb_, cus * 16 / wpb_)); fflush(stdout); } printf("\n"); } while (0)
        for (int rep = 0; rep < 2; rep++) {
            printf("-- repetition %d\n", rep);
            ROWG("both, static 4x16 (shipped mapping)", (std::vector<Geo>{{4, 16, 0}}), 1, 1, 0, 0, 0, 1);
            ROWG("arith only, static 4x16", (std::vector<Geo>{{4, 16, 0}}), 1, 0, 0, 0, 0, 1);
            CROW8("arith only 16x16, barrier before pass 2", 1, 0, 0, 1, 16);
            CROW8("stores only 16x16, barrier before pass 2", 0, 1, 0, 1, 16);
            CROW8("both 16x16, barrier before pass 2", 1, 1, 0, 1, 16);
            CROW8("both 16x16, row held, stores in next pass 1, barrier at item start", 1, 1, 1, 3, 16);
            CROW8("both 16x16, row held, stores over next item, barrier at item start", 1, 1, 2, 3, 16);
            CROW8("both 16x16, row held, stores in next pass 1, barrier before pass 2", 1, 1, 1, 1, 16);
            CROW8("both 16x16, four stores at item end behind a barrier", 1, 1, 3, 2, 16);
            CROW8("both 8x16, row held, stores in next pass 1, barrier at item start", 1, 1, 1, 3, 8);
            CROW8("both 16x16, row held, stores in next pass 1, no barrier", 1, 1, 1, 0, 16);
        }
    }
    if (want("r9")) {
        const int strides[] = {1, 4, 8, 16, 32};
        printf("XTRA = %d; producer/consumer in one 16-wave workgroup per CU (NC computing + 16-NC writer waves); columns: stride = 1 4 8 16 32\n", XTRA);
#define PROW(label, A_, S_, NC_) do { printf("%-60s", label); for (int st : strides) { cx.p[0].coop_stride = cx.p[1].coop_stride = st; \
            printf(" %6.1f", run_pc<A_, S_, NC_>(cx, cus / st * st)); fflush(stdout); } printf("\n"); } while (0)
        for (int rep = 0; rep < 2; rep++) {
            printf("-- repetition %d\n", rep);
            ROWG("both, static 4x16 (shipped mapping)", (std::vector<Geo>{{4, 16, 0}}), 1, 1, 0, 0, 0, 1);
            ROWG("arith only, static 4x16", (std::vector<Geo>{{4, 16, 0}}), 1, 0, 0, 0, 0, 1);
            ROWG("arith only, static 4x12", (std::vector<Geo>{{4, 12, 0}}), 1, 0, 0, 0, 0, 1);
            PROW("12 + 4: arith only (rows into LDS, no HBM stores)", 1, 0, 12);
            PROW("12 + 4: stores only", 0, 1, 12);
            PROW("12 + 4: both", 1, 1, 12);
            PROW("14 + 2: both", 1, 1, 14);
            PROW("8 + 8: both", 1, 1, 8);
            PROW("15 + 1: both", 1, 1, 15);
        }
    }
    if (want("r10")) {
        const int strides[] = {4, 8, 16};
        printf("XTRA = %d; workgroup-cooperative rows (map 10), columns: stride = 4 8 16\n", XTRA);
        for (int rep = 0; rep < 2; rep++) {
            printf("-- repetition %d\n", rep);
            ROWG("both, static 4x16 (shipped mapping)", (std::vector<Geo>{{4, 16, 0}}), 1, 1, 0, 0, 0, 1);
            ROWG("both, static 1x12", (std::vector<Geo>{{1, 12, 0}}), 1, 1, 0, 0, 0, 1);
            CROW8("both 4x16, barrier before pass 2", 1, 1, 0, 1, 4);
            CROW8("both 8x16, barrier before pass 2", 1, 1, 0, 1, 8);
            CROW8("both 16x16, barrier before pass 2", 1, 1, 0, 1, 16);
            CROW8("both 16x16, halves in anti-phase (2 barriers per item)", 1, 1, 0, 4, 16);
            CROW8("both 8x16, halves in anti-phase", 1, 1, 0, 4, 8);
            CROW8("arith only 16x16, halves in anti-phase", 1, 0, 0, 4, 16);
            CROW8("stores only 16x16, halves in anti-phase", 0, 1, 0, 4, 16);
        }
    }
    if (want("r11")) {
        const int strides[] = {1, 4, 8, 16, 32};
        printf("XTRA = %d; pairs of rows per wave (map 12), no barrier; columns: stride = 1 4 8 16 32\n", XTRA);
#define PAIRROW(label, A_, S_, wpb_, wpc_) do { printf("%-50s", label); for (int st : strides) { cx.p[0].coop_stride = cx.p[1].coop_stride = st; \
            printf(" %6.1f", run<A_, S_, 0, 0, 12, 1, 0>(cx, wpb_, cus * wpc_ / wpb_)); fflush(stdout); } printf("\n"); } while (0)
        for (int rep = 0; rep < 2; rep++) {
            printf("-- repetition %d\n", rep);
            ROWG("both, static 4x16 (shipped mapping)", (std::vector<Geo>{{4, 16, 0}}), 1, 1, 0, 0, 0, 1);
            ROWG("both, static 1x12", (std::vector<Geo>{{1, 12, 0}}), 1, 1, 0, 0, 0, 1);
            ROWG("both, static 4x12", (std::vector<Geo>{{4, 12, 0}}), 1, 1, 0, 0, 0, 1);
            PAIRROW("pairs, both 4x16", 1, 1, 4, 16);
            PAIRROW("pairs, both 1x16", 1, 1, 1, 16);
            PAIRROW("pairs, both 4x12", 1, 1, 4, 12);
            PAIRROW("pairs, both 1x12", 1, 1, 1, 12);
            PAIRROW("pairs, arith only 4x16", 1, 0, 4, 16);
            PAIRROW("pairs, stores only 4x16", 0, 1, 4, 16);
        }
    }
    if (want("r12")) {
#ifdef FP32
        printf("arithmetic in FLOAT, XTRA = %d\n", XTRA);
#else
        printf("arithmetic in DOUBLE, XTRA = %d\n", XTRA);
#endif
        const int strides[] = {8};
        for (int rep = 0; rep < 3; rep++) {
            printf("-- repetition %d\n", rep);
            ROWG("arith only, static 4x16", (std::vector<Geo>{{4, 16, 0}}), 1, 0, 0, 0, 0, 1);
            ROWG("stores only, static 4x16", (std::vector<Geo>{{4, 16, 0}}), 0, 1, 0, 0, 0, 1);
            ROWG("both, static 4x16 (shipped mapping)", (std::vector<Geo>{{4, 16, 0}}), 1, 1, 0, 0, 0, 1);
            PAIRROW("pairs stride 8, arith only 4x16", 1, 0, 4, 16);
            PAIRROW("pairs stride 8, stores only 4x16", 0, 1, 4, 16);
            PAIRROW("pairs stride 8, both 4x16", 1, 1, 4, 16);
            CROW8("coop 16x16 stride 8, barrier before pass 2, both", 1, 1, 0, 1, 16);
        }
    }
    if (want("ticket")) {        // (3 ms per launch: ~11.5 ns per returning atomic on one line, whatever the address count)
        ROW("arith only, per-XCD ticket", 1, 0, 0, 0, 2);
        ROW("arith only, global ticket", 1, 0, 0, 0, 3);
    }
    if (want("r2")) {
        printf("columns (waves per workgroup x waves per CU): 4x16 4x12 1x16 1x12 4x8 1x8\n");
        ROWG("stores only, peak records plain", geos2, 0, 1, 0, 0, 0, 1);
        ROWG("stores only, no peak records", geos2, 0, 1, 0, 0, 0, 0);
        ROWG("stores only, peak records nt", geos2, 0, 1, 0, 0, 0, 2);
        ROWG("random sleep + stores, peak records plain", geos2, 2, 1, 0, 0, 0, 1);
        ROWG("random sleep + stores, no peak records", geos2, 2, 1, 0, 0, 0, 0);
        ROWG("random sleep only", geos2, 2, 0, 0, 0, 0, 0);
        ROWG("both, peak records plain", geos2, 1, 1, 0, 0, 0, 1);
        ROWG("both, no peak records", geos2, 1, 1, 0, 0, 0, 0);
        ROWG("both, peak records nt", geos2, 1, 1, 0, 0, 0, 2);
        ROWG("stores only, XCD-contiguous, no peak records", geos2, 0, 1, 0, 0, 5, 0);
        ROWG("both, XCD-contiguous, no peak records", geos2, 1, 1, 0, 0, 5, 0);
        ROWG("both, XCD-contiguous, peak records plain", geos2, 1, 1, 0, 0, 5, 1);
        printf("one 16-wave workgroup per CU, rows through an LDS counter:\n");
        ROWG("stores only, row = t*256 + cu, no peak records", geo_cu, 0, 1, 0, 0, 6, 0);
        ROWG("both, row = t*256 + cu, no peak records", geo_cu, 1, 1, 0, 0, 6, 0);
        ROWG("both, row = t*256 + cu, peak records plain", geo_cu, 1, 1, 0, 0, 6, 1);
        ROWG("arith only, row = t*256 + cu", geo_cu, 1, 0, 0, 0, 6, 0);
        ROWG("stores only, row = cu*1024 + t, no peak records", geo_cu, 0, 1, 0, 0, 7, 0);
        ROWG("both, row = cu*1024 + t, no peak records", geo_cu, 1, 1, 0, 0, 7, 0);
    }
    printf("[%6.1fs] done\n", now_s() - t_start);
    return 0;
}
