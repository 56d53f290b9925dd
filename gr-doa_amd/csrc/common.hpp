// common.hpp — shared host-side plumbing of libdoa_hip.so (error reporting, HIP call checks,
// small device-buffer helper, wave-level primitives used by every kernel file).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <new>

#include "../../include/doa_hip.h"

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

// Binds the calling thread to a usable HIP device (the current one); DOA_ERR_NO_DEVICE if none.
int ensure_device(int *device_out);

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

// ---- device-side wave primitives (wave = 64 lanes on gfx950) -----------------------------------
constexpr int kWave = 64;

__device__ __forceinline__ float wave_allreduce_sum(float v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, kWave);
    return v;
}
__device__ __forceinline__ float wave_allreduce_max(float v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m, kWave));
    return v;
}

}  // namespace doa
