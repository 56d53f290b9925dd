// peak_device.hpp — device-side top-M local-maximum pick on a vector held in registers.
// Shared by the standalone find_local_max kernel (K5) and the fused scan+peak kernel (K4+K5).
// Algorithm and reference citations: see find_local_max.hip.
#pragma once
#include "common.hpp"

#include <climits>

namespace doa {

// (value, index) ordering used for ranking: larger value first, then lower index; idx == INT_MAX
// marks "no candidate".
__device__ __forceinline__ bool cand_better(float v, int i, float bv, int bi)
{
    if (i == INT_MAX) return false;
    if (bi == INT_MAX) return true;
    return (v > bv) || (v == bv && i < bi);
}
template <int CTRL, int ROW_MASK> __device__ __forceinline__ void argbest_step(float &v, int &i)
{
    // lanes of rows masked off receive their own (v, i) back, which never beats itself
    const float ov = dpp_move<CTRL, ROW_MASK>(v, v);
    const int oi = __builtin_amdgcn_update_dpp(i, i, CTRL, ROW_MASK, 0xF, false);
    if (cand_better(ov, oi, v, i)) { v = ov; i = oi; }
}
// wave-wide best (value, index) pair, returned wave-uniform; DPP only (no LDS traffic)
__device__ __forceinline__ void wave_argbest(float &v, int &i)
{
    argbest_step<0xB1, 0xF>(v, i);
    argbest_step<0x4E, 0xF>(v, i);
    argbest_step<0x124, 0xF>(v, i);
    argbest_step<0x128, 0xF>(v, i);
    argbest_step<0x142, 0xA>(v, i);
    argbest_step<0x143, 0xC>(v, i);
    v = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
    i = __builtin_amdgcn_readlane(i, 63);
}
__device__ __forceinline__ int wave_sum_int(int x)
{
    x += __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xF, 0xF, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xF, 0xF, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x124, 0xF, 0xF, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x128, 0xF, 0xF, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xA, 0xF, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xC, 0xF, false);
    return __builtin_amdgcn_readlane(x, 63);
}

// v[j][e] = element 256*j + 4*lane + e of the vector (positions >= L hold anything).  Writes the
// item's M peak values (rank order) and M locations (descending) — lanes 0..M-1 do the stores.
template <int CH>
__device__ __forceinline__ void peak_pick(const float (&v)[CH][4], int lane, int L, int M,
                                          const float *__restrict__ xaxis, float *__restrict__ out_val_item,
                                          float *__restrict__ out_loc_item)
{
    if (M == 1) {
        // find_one_local_peak_indx = arma index_max (op_max::direct_max): best starts at -inf and is
        // replaced only by a strictly greater element -> first occurrence of the maximum; NaNs never
        // win; nothing above -inf -> index 0
        float bv = 0.f;
        int bi = INT_MAX;
#pragma unroll
        for (int j = 0; j < CH; j++)
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const int p = 256 * j + 4 * lane + e;
                if (p < L && v[j][e] > -INFINITY && cand_better(v[j][e], p, bv, bi)) { bv = v[j][e]; bi = p; }
            }
        wave_argbest(bv, bi);
        const float v0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v[0][0]), 0));
        if (lane == 0) {
            const int idx = (bi == INT_MAX) ? 0 : bi;
            out_val_item[0] = (bi == INT_MAX) ? v0 : bv;
            out_loc_item[0] = xaxis[idx];
        }
        return;
    }
    int sel_idx = INT_MAX;      // lane r keeps the r-th ranked index
    float sel_val = 0.f;
    int n_valid = 0, best_list_pos = 0;

    if (M > 1) {
        // --- sign of the first-order difference; positions past the last difference act as +1 ---
        int s[CH][4];
        bool flat_here = false;
#pragma unroll
        for (int j = 0; j < CH; j++) {
            float nxt_first = __shfl_down(v[j][0], 1, kWave);                 // lane+1, e=0, same chunk
            const float wrap = (j + 1 < CH) ? __shfl(v[(j + 1 < CH) ? j + 1 : j][0], 0, kWave) : 0.f;
            if (lane == kWave - 1) nxt_first = wrap;                          // lane 0 of the next chunk
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const int p = 256 * j + 4 * lane + e;
                const float nx = (e < 3) ? v[j][e + 1] : nxt_first;
                if (p < L - 1) {
                    const float d = nx - v[j][e];
                    s[j][e] = (d > 0.f) ? 1 : ((d < 0.f) ? -1 : 0);
                    flat_here |= (s[j][e] == 0);
                } else {
                    s[j][e] = 1;
                }
            }
        }
        // --- flats: suffix scan in position order (chunk, lane, e), last chunk first ---
        if (__any(flat_here)) {
            int chunk_carry = 1;  // sign taken by a flat that runs to the end of the vector
#pragma unroll
            for (int j = CH - 1; j >= 0; j--) {
                int f = 0;
#pragma unroll
                for (int e = 3; e >= 0; e--) f = (s[j][e] != 0) ? s[j][e] : f;   // first non-zero of this lane
                const unsigned long long nz = __ballot(f != 0), pos = __ballot(f > 0);
                const unsigned long long above = (lane == kWave - 1) ? 0ull : (nz >> (lane + 1));
                int carry = chunk_carry;
                if (above) {
                    const int src = lane + 1 + __builtin_ctzll(above);
                    carry = ((pos >> src) & 1ull) ? 1 : -1;
                }
#pragma unroll
                for (int e = 3; e >= 0; e--) {
                    if (s[j][e] == 0) s[j][e] = carry;
                    carry = s[j][e];
                }
                chunk_carry = __shfl(s[j][0], 0, kWave);
            }
        }
        // --- peaks: s[p-1] == +1 and s[p] == -1 ---
        unsigned long long pk = 0;   // bit (4*j+e)
#pragma unroll
        for (int j = 0; j < CH; j++) {
            int prev_last = __shfl_up(s[j][3], 1, kWave);                      // lane-1, e=3
            const int wrap = (j > 0) ? __shfl(s[(j > 0) ? j - 1 : 0][3], kWave - 1, kWave) : 0;
            if (lane == 0) prev_last = wrap;
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const int p = 256 * j + 4 * lane + e;
                const int sp = (e > 0) ? s[j][e - 1] : prev_last;
                if (p >= 1 && p <= L - 2 && sp == 1 && s[j][e] == -1) pk |= 1ull << (4 * j + e);
            }
        }
        n_valid = wave_sum_int(__popcll(pk));
        // --- top-M by value (descending), ties -> lowest index ---
        const int rounds = (n_valid < M) ? n_valid : M;
        for (int r = 0; r < rounds; r++) {
            float bv = 0.f;
            int bi = INT_MAX;
#pragma unroll
            for (int j = 0; j < CH; j++)
#pragma unroll
                for (int e = 0; e < 4; e++)
                    if ((pk >> (4 * j + e)) & 1ull) {
                        const int p = 256 * j + 4 * lane + e;
                        if (cand_better(v[j][e], p, bv, bi)) { bv = v[j][e]; bi = p; }
                    }
            wave_argbest(bv, bi);
            if (r == 0) {
                // position of the best peak inside the ascending peak list (= #peaks before it)
                int before = 0;
#pragma unroll
                for (int j = 0; j < CH; j++)
#pragma unroll
                    for (int e = 0; e < 4; e++)
                        if (((pk >> (4 * j + e)) & 1ull) && (256 * j + 4 * lane + e) < bi) before++;
                best_list_pos = wave_sum_int(before);
            }
            // owner drops the winner from its candidate set
            if (bi != INT_MAX && ((bi & 255) >> 2) == lane) pk &= ~(1ull << (4 * (bi >> 8) + (bi & 3)));
            if (lane == r) { sel_idx = bi; sel_val = bv; }
        }
    }

    // --- fill slots / M == 1: indices that are not ranked peaks ---
    int fill_idx = 0;
    if (M == 1 || n_valid == 0) {
        // arma index_max (op_max::direct_max): best starts at -inf, an element replaces it only if
        // strictly greater -> first occurrence of the maximum, NaNs never win, nothing > -inf -> 0
        float bv = 0.f;
        int bi = INT_MAX;
#pragma unroll
        for (int j = 0; j < CH; j++)
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const int p = 256 * j + 4 * lane + e;
                if (p < L && v[j][e] > -INFINITY && cand_better(v[j][e], p, bv, bi)) { bv = v[j][e]; bi = p; }
            }
        wave_argbest(bv, bi);
        fill_idx = (bi == INT_MAX) ? 0 : bi;
    } else {
        fill_idx = best_list_pos;   // reference quirk: list position used as a vector index (:153,160)
    }
    // value at fill_idx, fetched from the owning lane's registers (wave-uniform index)
    float fill_val;
    {
        float mine = 0.f;
#pragma unroll
        for (int j = 0; j < CH; j++)
#pragma unroll
            for (int e = 0; e < 4; e++)
                if ((fill_idx >> 8) == j && (fill_idx & 3) == e) mine = v[j][e];
        fill_val = __shfl(mine, (fill_idx & 255) >> 2, kWave);
    }
    if (lane < M) {
        int idx;
        float val;
        if (M > 1 && lane < n_valid) { idx = sel_idx; val = sel_val; }
        else { idx = fill_idx; val = fill_val; }
        const float x = xaxis[idx];
        out_val_item[lane] = val;
        // descending sort of the M locations: rank = number of entries that must precede this one
        int rank = 0;
        for (int k = 0; k < M; k++) {
            const float xk = __shfl(x, k, kWave);
            rank += (xk > x || (xk == x && k < lane)) ? 1 : 0;
        }
        out_loc_item[rank] = x;
    }
}

// peak_pick_stream: find_local_max on a vector the caller serves through v_at(p) (global memory for the stand-alone
// K5, an LDS row for the long-spectrum scan kernel that fuses K5), one wave per vector, 64 consecutive positions
// per step (position p = 64 k + lane), nothing of the vector kept in registers.  The three-way sign of every
// first difference is two compares whose results are wave masks in SGPR pairs, in position order, so the
// reference's steps become scalar bit arithmetic:
//   * flats (sign 0; a NaN difference counts as one) take the sign of the next non-zero difference to their
//     right: groups are walked from the last to the first, the sign at the start of the group to the right is
//     a carried bit, and inside a group the fill is a 6-step Kogge-Stone propagation on the 64-bit mask
//     (only groups that contain a flat pay for it);
//   * lane k keeps the resolved "negative" mask of group k; all peak masks are then one vector expression,
//     peak[k] = neg[k] & ~((neg[k] << 1) | (neg[k-1] >> 63));
//   * the top-M are M rounds of (every lane scans the set bits of its own group, re-reading those values;
//     wave arg-max); the fill rule, index_max and the output ordering are those of peak_pick.
// Any 1 <= L <= 4096.  ~1000 instructions and ~40 VGPRs per vector at L = 4096, where the register-resident CH = 16
// kernel executes ~6500 with 230 VGPRs.
// BLOCKED = true: the same masks built the other way round -- lane k walks ITS OWN 64 positions 64k .. 64k+63 serially (65
// reads, a handful of vector instructions per position, no cross-lane traffic), flats inside the block are filled by the
// same Kogge-Stone steps on the lane's own 64-bit masks, and the sign a block's trailing flats inherit -- that of the first
// non-flat difference in any block to the right -- comes from two ballots (which blocks have a non-flat difference, and the
// sign of their first one).  For callers whose v_at(64 k + i) is conflict-free across k (the long-spectrum scan kernel
// pads its LDS row by one word per 64): 64 steps of a scalar dependency chain with a cross-lane shuffle, a readlane and
// two ballots each become 64 independent lanes of straight-line code.
// BS (blocked form only): positions per lane, a power of two from 4 to 64 with L <= 64 BS -- 64 for vectors of up to 4096
// values, L / 64 for the short spectra of the lean scan kernels (all 64 lanes busy, BS steps each).
template <bool BLOCKED = false, int BS = 64, class Fetch>
__device__ __forceinline__ void peak_pick_stream(Fetch v_at, int L, int M, const float *__restrict__ xaxis,
                                                 float *__restrict__ out_val_item, float *__restrict__ out_loc_item, int lane)
{
    static_assert(BLOCKED || BS == 64, "the mask form works in groups of 64");
    static_assert(BS == 4 || BS == 8 || BS == 16 || BS == 32 || BS == 64, "block size");
    constexpr int LOG_BS = (BS == 4) ? 2 : (BS == 8) ? 3 : (BS == 16) ? 4 : (BS == 32) ? 5 : 6;
    constexpr unsigned long long kBlockMask = (BS == 64) ? ~0ull : ((1ull << (BS & 63)) - 1ull);
    const int G = (L + BS - 1) >> LOG_BS;                // blocks of BS positions (<= 64)

    // arma index_max (op_max::direct_max): first occurrence of the maximum; NaN and -inf never win; none -> 0
    float mv = 0.f;
    int mi = INT_MAX;
    unsigned long long neg_mine = 0ull;                  // lane k: resolved "sign == -1" mask of block k
    if constexpr (BLOCKED) {
        // contract of this form: v_at.blk(i) = the value at position BS lane + i, readable for 0 <= i <= BS in every lane whose
        // block starts inside the vector (what it returns at positions >= L is not used).  The maximum is not tracked here:
        // it is looked for below only if the answer needs it.  Each compare lands in VCC and is shifted into the lane's
        // mask word as the carry-in of w = w + w + carry (v_addc_co_u32): two vector instructions per position and mask.
        const bool live = BS * lane < L;
        constexpr int W = (BS + 31) / 32, WB = (BS < 32) ? BS : 32;          // mask words per lane, bits per word
        unsigned neg_w[2] = {0u, 0u}, pos_w[2] = {(unsigned)kBlockMask, (unsigned)(kBlockMask >> 32)};
        if (live) {
            float cur = v_at.blk(BS);
            // descending walk, so that the first bit shifted in ends up as the top bit of its word
#pragma unroll
            for (int h = W - 1; h >= 0; h--) {
                unsigned nw = 0u, pw = 0u;
#pragma unroll
                for (int ii = WB - 1; ii >= 0; ii--) {
                    const float v = v_at.blk(32 * h + ii);              // cur = the value one position to the right
                    asm("v_cmp_lt_f32 vcc, %2, %3\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc\n\t"
                        "v_cmp_gt_f32 vcc, %2, %3\n\tv_addc_co_u32 %1, vcc, %1, %1, vcc"
                        : "+v"(nw), "+v"(pw) : "v"(cur), "v"(v) : "vcc");
                    cur = v;
                }
                neg_w[h] = nw; pos_w[h] = pw;
            }
        }
        unsigned long long neg = ((unsigned long long)neg_w[1] << 32) | neg_w[0];
        unsigned long long pos = ((unsigned long long)pos_w[1] << 32) | pos_w[0];
        // positions from the last one of the vector on have no difference: +1 (what the compares saw there is garbage)
        const int t = L - 1 - BS * lane;
        if (live && t < BS) {
            const unsigned long long beyond = (~0ull << t) & kBlockMask;
            neg &= ~beyond; pos |= beyond;
        }
        const unsigned long long nonflat = neg | pos;
        unsigned long long res = neg;
        if (__builtin_amdgcn_ballot_w64((~nonflat & kBlockMask) != 0ull) != 0ull) {    // some flat somewhere in the vector
            // resolved sign at the first position of the block to the right = sign of the first non-flat difference at or
            // beyond it (blocks past the end are all +1, so one always exists up to the last lane)
            const bool any_nf = nonflat != 0ull;
            const bool first_neg = any_nf && ((neg >> (any_nf ? __builtin_ctzll(nonflat) : 0)) & 1ull) != 0ull;
            const unsigned long long has = __builtin_amdgcn_ballot_w64(any_nf);
            const unsigned long long sgn = __builtin_amdgcn_ballot_w64(first_neg);
            const unsigned long long right = (lane < 63) ? (has >> (lane + 1)) : 0ull;
            unsigned long long carry_neg = 0ull;
            if (right != 0ull) carry_neg = (sgn >> (lane + 1 + (int)__builtin_ctzll(right))) & 1ull;
            unsigned long long prop = ~nonflat & kBlockMask;
            res |= prop & (carry_neg << (BS - 1));
#pragma unroll
            for (int sh = 1; sh < BS; sh <<= 1) {
                res |= (res >> sh) & prop;
                prop &= prop >> sh;
            }
        }
        neg_mine = res;
    } else {
        unsigned long long carry_neg = 0ull;             // resolved sign at the first position of the group to the right (0: +1)
        float above_first = 0.f;                         // value at the first position of the group to the right
        constexpr int U = 8;                             // groups per batch: U loads in flight per lane before any is consumed
        for (int kb = ((G - 1) / U) * U; kb >= 0; kb -= U) {
            float vb[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int p = 64 * (kb + u) + lane;
                vb[u] = (p < L) ? v_at(p) : 0.f;
            }
#pragma unroll
            for (int u = U - 1; u >= 0; u--) {
                const int k = kb + u;
                if (k >= G) continue;                    // wave-uniform
                const int p = 64 * k + lane;
                const bool inb = p < L, has_diff = p + 1 < L;
                const float v = vb[u];
                // next value: lane+1 of this group; lane 63: the first value of the group to the right
                float nx = __shfl_down(v, 1, kWave);
                if (lane == kWave - 1) nx = above_first;
                above_first = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
                // descending walk: ">=" lets the lower position win a tie, i.e. the first occurrence overall
                if (inb && v > -INFINITY && (mi == INT_MAX || v >= mv)) { mv = v; mi = p; }
                const unsigned long long neg = __builtin_amdgcn_ballot_w64(has_diff && nx < v);
                const unsigned long long pos = __builtin_amdgcn_ballot_w64(has_diff ? (nx > v) : true);   // past the last difference: +1
                const unsigned long long flat = ~(neg | pos);
                unsigned long long res = neg;
                if (flat != 0ull) {
                    unsigned long long prop = flat;
                    res |= prop & (carry_neg << 63);
#pragma unroll
                    for (int sh = 1; sh < 64; sh <<= 1) {
                        res |= (res >> sh) & prop;
                        prop &= prop >> sh;
                    }
                }
                carry_neg = res & 1ull;
                if (lane == k) neg_mine = res;
            }
        }
    }
    if constexpr (!BLOCKED) wave_argbest(mv, mi);
    int idx_max = (mi == INT_MAX) ? 0 : mi;

    int sel_idx = INT_MAX;
    float sel_val = 0.f;
    int n_valid = 0, best_list_pos = 0;
    if (M > 1) {
        // peaks: s[p-1] == +1 and s[p] == -1, 1 <= p <= L-2 (positions >= L-1 carry +1, so they never qualify)
        const unsigned long long below = __shfl_up(neg_mine, 1, kWave);
        unsigned long long pk = neg_mine & ~((neg_mine << 1) | ((lane > 0) ? (below >> (BS - 1)) : 0ull)) & kBlockMask;
        if (lane == 0) pk &= ~1ull;
        const unsigned long long pk_all = pk;
        n_valid = wave_sum_int(__builtin_popcountll(pk));
        const int rounds = (n_valid < M) ? n_valid : M;
        for (int r = 0; r < rounds; r++) {
            float bv = 0.f;
            int bi = INT_MAX;
            for (unsigned long long m = pk; m != 0ull; m &= m - 1ull) {
                const int pp = BS * lane + (int)__builtin_ctzll(m);
                const float val = v_at(pp);
                if (cand_better(val, pp, bv, bi)) { bv = val; bi = pp; }
            }
            wave_argbest(bv, bi);
            if (r == 0) {
                // position of the best peak inside the ascending peak list (= #peaks before it)
                const int rel = bi - BS * lane;
                const unsigned long long lower = (rel >= BS) ? ~0ull : ((rel <= 0) ? 0ull : ((1ull << rel) - 1ull));
                best_list_pos = wave_sum_int(__builtin_popcountll(pk_all & lower));
            }
            if (bi != INT_MAX && (bi >> LOG_BS) == lane) pk &= ~(1ull << (bi & (BS - 1)));
            if (lane == r) { sel_idx = bi; sel_val = bv; }
        }
    }
    if constexpr (BLOCKED) {
        if (M == 1 || n_valid == 0) {                    // wave-uniform: only now is index_max part of the answer
            if (lane < G) {
                // ascending walk of the own block: ">" keeps the first occurrence, -inf and NaN never pass it
                float best = -INFINITY;
                const int lim = L - BS * lane;                          // positions of this block inside the vector
#pragma unroll 8
                for (int i = 0; i < BS; i++) {
                    const float v = v_at.blk(i);
                    if (i < lim && v > best) { best = v; mi = BS * lane + i; }
                }
                mv = best;
            }
            wave_argbest(mv, mi);
            idx_max = (mi == INT_MAX) ? 0 : mi;
        }
    }
    const int fill_idx = (M == 1 || n_valid == 0) ? idx_max : best_list_pos;   // reference quirk for 0 < n_valid < M (:153,160)
    const float fill_val = v_at(fill_idx);
    if (lane < M) {
        int idx;
        float val;
        if (M > 1 && lane < n_valid) { idx = sel_idx; val = sel_val; }
        else { idx = fill_idx; val = fill_val; }
        const float x = xaxis[idx];
        out_val_item[lane] = val;
        int rank = 0;                                    // descending sort of the M locations
        for (int k = 0; k < M; k++) {
            const float xk = __shfl(x, k, kWave);
            rank += (xk > x || (xk == x && k < lane)) ? 1 : 0;
        }
        out_loc_item[rank] = x;
    }
}

}  // namespace doa
