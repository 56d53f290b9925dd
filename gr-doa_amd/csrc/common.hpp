// common.hpp — shared host-side plumbing of libdoa_hip.so (error reporting, HIP call checks,
// small device-buffer helper, wave-level primitives used by every kernel file).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>

#include "../../include/doa_hip.h"
#include "../../include/doa_hip_test.h"

namespace doa {

// ---- error reporting ------------------------------------------------------------------------
void set_error(const char *fmt, ...) __attribute__((format(printf, 1, 2)));
void clear_error();

#define DOA_HIP_TRY(expr)                                                                       \
    do {                                                                                        \
        hipError_t _e = (expr);                                                                 \
        if (_e != hipSuccess) {                                                                 \
            ::doa::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__,   \
                             __LINE__);                                                         \
            return (_e == hipErrorNoDevice || _e == hipErrorInvalidDevice) ? DOA_ERR_NO_DEVICE  \
                                                                           : DOA_ERR_HIP;       \
        }                                                                                       \
    } while (0)

// Binds the calling thread to a usable HIP device (the current one); DOA_ERR_NO_DEVICE if none.  First call per device: populates
// the runtime's hardware-queue pool (prime_hw_queues, common.hip).
int ensure_device(int *device_out);
// Makes `device` (the one the handle was created on) current for the calling thread if it is not.
int bind_device(int device);
// Compute units of the current device (cached); the grid caps of the streaming kernels are multiples of it.
int cu_count();

// ---- grow-only device / pinned-host buffers ----------------------------------------------------
struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    int reserve(size_t bytes);  // DOA_OK or error; keeps contents only if no growth was needed
    void release();
    template <class T> T *as() const { return static_cast<T *>(p); }
};
struct PinnedBuf {
    void *p = nullptr;
    size_t cap = 0;
    int reserve(size_t bytes);
    void release();
    template <class T> T *as() const { return static_cast<T *>(p); }
};

int internal_precision_bits();  // 32 or 64 (doa_set_internal_precision)

// Launch-shape knobs for same-box A/B runs exist only in lab builds (make LAB=1); the default build has none of them
// compiled in and this is the constant `dflt`.
#ifdef DOA_LAB
#define DOA_LAB_ENV_INT(name, dflt) ([] { static const int v_ = [] { const char *e_ = getenv(name); return e_ ? atoi(e_) : (dflt); }(); return v_; }())
#else
#define DOA_LAB_ENV_INT(name, dflt) (dflt)
#endif

// Distance between the device copies of consecutive input streams (doa_stream_stride_bytes): the stream size rounded up
// to 8 KiB plus 4.5 KiB, so that stream k starts 4.5 k KiB further into the 8 KiB period than stream 0 -- sixteen
// distinct multiples of 512 B for sixteen streams (9 is odd).  See include/doa_hip.h for the measurement.
inline size_t stream_stride_bytes(size_t stream_bytes) { return ((stream_bytes + 8191) & ~(size_t)8191) + 4608; }

// ---- device-side wave primitives (wave = 64 lanes on gfx950) -----------------------------------
constexpr int kWave = 64;

__device__ __forceinline__ int wave_allreduce_min_int(int v)
{
    auto step = [](int x, int y) { return x < y ? x : y; };
    v = step(v, __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xF, 0xF, false));
    v = step(v, __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xF, 0xF, false));
    v = step(v, __builtin_amdgcn_update_dpp(v, v, 0x124, 0xF, 0xF, false));
    v = step(v, __builtin_amdgcn_update_dpp(v, v, 0x128, 0xF, 0xF, false));
    v = step(v, __builtin_amdgcn_update_dpp(v, v, 0x142, 0xA, 0xF, false));
    v = step(v, __builtin_amdgcn_update_dpp(v, v, 0x143, 0xC, 0xF, false));
    return __builtin_amdgcn_readlane(v, 63);
}

// streamed-once data: non-temporal 16-byte accesses (read-once input samples, write-once spectra)
typedef float doa_f32x4 __attribute__((ext_vector_type(4)));
template <bool NT> __device__ __forceinline__ float4 load_f4(const float4 *p)
{
    if constexpr (NT) {
        const doa_f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const doa_f32x4 *>(p));
        return make_float4(v.x, v.y, v.z, v.w);
    } else {
        return *p;
    }
}
template <bool NT> __device__ __forceinline__ void store_f4(float4 *p, float4 v)
{
    if constexpr (NT) {
        doa_f32x4 t = {v.x, v.y, v.z, v.w};
        __builtin_nontemporal_store(t, reinterpret_cast<doa_f32x4 *>(p));
    } else {
        *p = v;
    }
}

// Cross-lane moves without LDS: DPP row operations (gfx9 encodings) + one v_readlane.
//   quad_perm [1,0,3,2] = 0xB1, quad_perm [2,3,0,1] = 0x4E, row_ror:4 = 0x124, row_ror:8 = 0x128,
//   row_bcast:15 = 0x142 (lane 15 of each row -> the next row), row_bcast:31 = 0x143.
template <int CTRL, int ROW_MASK> __device__ __forceinline__ float dpp_move(float old, float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), CTRL, ROW_MASK, 0xF, false));
}
// Sum over the 64 lanes, returned wave-uniform (the total forms in row 3 and is read from lane 63).
__device__ __forceinline__ float wave_allreduce_sum(float v)
{
    v += dpp_move<0xB1, 0xF>(0.f, v);
    v += dpp_move<0x4E, 0xF>(0.f, v);
    v += dpp_move<0x124, 0xF>(0.f, v);
    v += dpp_move<0x128, 0xF>(0.f, v);       // every lane: sum of its row of 16
    v += dpp_move<0x142, 0xA>(0.f, v);       // rows 1,3 += rows 0,2
    v += dpp_move<0x143, 0xC>(0.f, v);       // rows 2,3 += row 1 (= rows 0+1)
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ float wave_allreduce_max(float v)
{
    v = fmaxf(v, dpp_move<0xB1, 0xF>(v, v));
    v = fmaxf(v, dpp_move<0x4E, 0xF>(v, v));
    v = fmaxf(v, dpp_move<0x124, 0xF>(v, v));
    v = fmaxf(v, dpp_move<0x128, 0xF>(v, v));
    v = fmaxf(v, dpp_move<0x142, 0xA>(v, v));
    v = fmaxf(v, dpp_move<0x143, 0xC>(v, v));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
// (six v_min_f32 with the DPP control on the instruction itself: written with update_dpp + fminf the compiler issues a move,
// a canonicalising max and the min per step.  v_min_f32 returns the other operand for a NaN one, like fminf.  A DPP read of
// a VGPR the previous VALU instruction wrote needs two wait states, which nothing inserts inside inline asm.)
__device__ __forceinline__ float wave_allreduce_min(float v)
{
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

}  // namespace doa
