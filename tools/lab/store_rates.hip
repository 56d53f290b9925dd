// store_rates.hip -- lab: what the write path gives for the scan kernel's output pattern (rows of 4 KiB written by one
// wave as 4 x 1 KiB float4 stores), by store flavour, waves per CU, item-to-wave assignment and drain policy.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// FLAV: 0 plain, 1 nontemporal, 2 sc1 (write-through)   DRAIN: s_waitcnt vmcnt(0) before every row   BLOCKED: contiguous item ranges per wave
template <int FLAV, bool DRAIN, bool BLOCKED, int DW>
__global__ __launch_bounds__(256) void store_kernel(float *__restrict__ out, int n_items, float v)
{
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int n_waves = gridDim.x * 4;
    const int per = (n_items + n_waves - 1) / n_waves;
    const int first = BLOCKED ? wave * per : wave;
    const int last = BLOCKED ? min(n_items, first + per) : n_items;
    const int step = BLOCKED ? 1 : n_waves;
    for (int item = first; item < last; item += step) {
        float *row = out + (size_t)item * 1024;
        if constexpr (DRAIN) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if constexpr (DW == 4) {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                float4 val = make_float4(v + j, v, v, v + lane);
                float4 *p = reinterpret_cast<float4 *>(row + 4 * lane + 256 * j);
                if constexpr (FLAV == 0) *p = val;
                if constexpr (FLAV == 1) { typedef float f4 __attribute__((ext_vector_type(4))); f4 t = {val.x, val.y, val.z, val.w}; __builtin_nontemporal_store(t, reinterpret_cast<f4 *>(p)); }
                if constexpr (FLAV == 2) { typedef float f4 __attribute__((ext_vector_type(4))); f4 t = {val.x, val.y, val.z, val.w}; asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(t) : "memory"); }
            }
        } else {
#pragma unroll
            for (int j = 0; j < 16; j++) {
                float *p = row + lane + 64 * j;
                if constexpr (FLAV == 0) *p = v + j;
                if constexpr (FLAV == 1) __builtin_nontemporal_store(v + j, p);
            }
        }
    }
}

// rows again, but wave w takes row k * n_waves + (w + k * ROT) % n_waves in its k-th turn: the rows being written at one
// moment are still a contiguous window, while one wave's successive rows no longer sit a power-of-two distance apart
template <int FLAV>
__global__ __launch_bounds__(256) void store_rot_kernel(float *__restrict__ out, int n_items, float v, int rot)
{
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int n_waves = gridDim.x * 4;
    int k = 0;
    for (int base = 0; base < n_items; base += n_waves, k++) {
        const int item = base + (wave + k * rot) % n_waves;
        if (item >= n_items) continue;
        float *row = out + (size_t)item * 1024;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            typedef float f4 __attribute__((ext_vector_type(4)));
            f4 t = {v + j, v, v, v + lane};
            f4 *p = reinterpret_cast<f4 *>(row + 4 * lane + 256 * j);
            if constexpr (FLAV == 0) *p = t; else __builtin_nontemporal_store(t, p);
        }
    }
}
// a wave takes 16 consecutive rows in 16 consecutive turns (item = (k >> 4) 16 n_waves + 16 w + (k & 15))
template <int FLAV>
__global__ __launch_bounds__(256) void store_blk16_kernel(float *__restrict__ out, int n_items, float v)
{
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int n_waves = gridDim.x * 4;
    const int turns = 16 * ((n_items + 16 * n_waves - 1) / (16 * n_waves));
    for (int k = 0; k < turns; k++) {
        const int item = (k >> 4) * (16 * n_waves) + 16 * wave + (k & 15);
        if (item >= n_items) continue;
        float *row = out + (size_t)item * 1024;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            typedef float f4 __attribute__((ext_vector_type(4)));
            f4 t = {v + j, v, v, v + lane};
            f4 *p = reinterpret_cast<f4 *>(row + 4 * lane + 256 * j);
            if constexpr (FLAV == 0) *p = t; else __builtin_nontemporal_store(t, p);
        }
    }
}
// what a fill kernel does: thread t writes 16 B at t, t + T, t + 2T, ... (every wave instruction 1 KiB contiguous, the whole
// grid a contiguous moving window)
template <int FLAV>
__global__ __launch_bounds__(256) void store_linear_kernel(float *__restrict__ out, size_t n_f4, float v)
{
    typedef float f4 __attribute__((ext_vector_type(4)));
    f4 *o = reinterpret_cast<f4 *>(out);
    const size_t T = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_f4; i += T) {
        f4 t = {v, v, v, v};
        if constexpr (FLAV == 0) o[i] = t; else __builtin_nontemporal_store(t, o + i);
    }
}

template <typename F> void time_it(const char *name, int wpc, int n_items, F launch)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int r = 0; r < 2; r++) launch(r);
    CK(hipDeviceSynchronize());
    const int reps = 10;
    CK(hipEventRecord(e0));
    for (int r = 0; r < reps; r++) launch(r);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / reps;
    printf("%-44s waves/CU %2d: %7.1f us  %5.2f TB/s\n", name, wpc, us, (double)n_items * 4096 / us / 1e6);
}

template <int FLAV, bool DRAIN, bool BLOCKED, int DW = 4>
void run(const char *name, float *bufs[2], int n_items, int cus)
{
    for (int wpc : {4, 8, 12, 16, 24, 32}) {
        const int blocks = cus * wpc / 4;
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int r = 0; r < 2; r++) hipLaunchKernelGGL((store_kernel<FLAV, DRAIN, BLOCKED, DW>), dim3(blocks), dim3(256), 0, 0, bufs[r & 1], n_items, 1.0f);
        CK(hipDeviceSynchronize());
        const int reps = 10;
        CK(hipEventRecord(e0));
        for (int r = 0; r < reps; r++) hipLaunchKernelGGL((store_kernel<FLAV, DRAIN, BLOCKED, DW>), dim3(blocks), dim3(256), 0, 0, bufs[r & 1], n_items, 1.0f);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1e3 / reps;
        printf("%-44s waves/CU %2d: %7.1f us  %5.2f TB/s\n", name, wpc, us, (double)n_items * 4096 / us / 1e6);
    }
}

int main(int argc, char **argv)
{
    const int n_items = argc > 1 ? atoi(argv[1]) : 262144;
    hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
    const int cus = pr.multiProcessorCount;
    float *bufs[2];
    for (int i = 0; i < 2; i++) CK(hipMalloc(&bufs[i], (size_t)n_items * 4096));
    printf("%d rows of 4 KiB (%.1f MB per launch)\n", n_items, n_items * 4096.0 / 1e6);
    {   // reference: the runtime's own fill
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipMemsetAsync(bufs[0], 0, (size_t)n_items * 4096, 0)); CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int r = 0; r < 10; r++) CK(hipMemsetAsync(bufs[r & 1], 0, (size_t)n_items * 4096, 0));
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("hipMemsetAsync: %.1f us  %.2f TB/s\n", ms * 100, (double)n_items * 4096 / (ms * 100) / 1e6);
    }
    for (int wpc : {4, 8, 12, 16, 32}) {
        time_it("linear fill, nt", wpc, n_items, [&](int r) { hipLaunchKernelGGL(store_linear_kernel<1>, dim3(cus * wpc / 4), dim3(256), 0, 0, bufs[r & 1], (size_t)n_items * 256, 1.0f); });
        time_it("linear fill, plain", wpc, n_items, [&](int r) { hipLaunchKernelGGL(store_linear_kernel<0>, dim3(cus * wpc / 4), dim3(256), 0, 0, bufs[r & 1], (size_t)n_items * 256, 1.0f); });
    }
    for (int wpc : {4, 8, 12, 16, 20}) {
        time_it("nt rows, 16 consecutive per wave", wpc, n_items, [&](int r) { hipLaunchKernelGGL(store_blk16_kernel<1>, dim3(cus * wpc / 4), dim3(256), 0, 0, bufs[r & 1], n_items, 1.0f); });
        time_it("plain rows, 16 consecutive per wave", wpc, n_items, [&](int r) { hipLaunchKernelGGL(store_blk16_kernel<0>, dim3(cus * wpc / 4), dim3(256), 0, 0, bufs[r & 1], n_items, 1.0f); });
    }
    for (int rot : {0})
        for (int wpc : {6, 8, 10, 12, 14, 16, 20}) {
            char nm[64]; snprintf(nm, sizeof nm, "nt rows, rotation %d", rot);
            time_it(nm, wpc, n_items, [&](int r) { hipLaunchKernelGGL(store_rot_kernel<1>, dim3(cus * wpc / 4), dim3(256), 0, 0, bufs[r & 1], n_items, 1.0f, rot); });
        }
    if (argc > 2) return 0;
    run<1, false, false>("nt dwordx4, strided items", bufs, n_items, cus);
    run<1, true, false>("nt dwordx4, strided items, drain per row", bufs, n_items, cus);
    run<0, false, false>("plain dwordx4, strided items", bufs, n_items, cus);
    run<2, false, false>("sc1 dwordx4, strided items", bufs, n_items, cus);
    run<1, false, true>("nt dwordx4, blocked items", bufs, n_items, cus);
    run<0, false, true>("plain dwordx4, blocked items", bufs, n_items, cus);
    run<1, false, false, 1>("nt dword (16 x 256 B), strided items", bufs, n_items, cus);
    run<0, false, false, 1>("plain dword (16 x 256 B), strided items", bufs, n_items, cus);
    return 0;
}
