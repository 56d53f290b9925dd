// store_geometry.hip -- lab: is the dependence of the row-store rate on "waves per CU" (store_rates.hip: 5.3 / 5.96 / 5.0 / 5.8
// TB/s at 8 / 12 / 16 / 20) a matter of waves per CU, of workgroups per CU, of the workgroup size, or of the total number of
// waves (= the distance between one wave's successive rows)?  Rows of 4 KiB, one wave per row, 4 x 1 KiB nt stores, item =
// wave + k * n_waves; the grid is given as (waves per workgroup, total workgroups).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// SPIN: dependent FMA chain per row (a stand-in for the scan kernel's arithmetic between two rows of stores)
template <int SPIN>
__global__ __launch_bounds__(512) void rows_kernel(float *__restrict__ out, int n_items, float v)
{
    const int lane = threadIdx.x & 63;
    const int wpb = blockDim.x >> 6;
    const int wave = blockIdx.x * wpb + (threadIdx.x >> 6);
    const int n_waves = gridDim.x * wpb;
    typedef float f4 __attribute__((ext_vector_type(4)));
    for (int item = wave; item < n_items; item += n_waves) {
        float *row = out + (size_t)item * 1024;
        float a = v + lane;
        if constexpr (SPIN > 0) {
#pragma unroll 8
            for (int s = 0; s < SPIN; s++) a = __builtin_fmaf(a, 1.0000001f, 1e-9f);
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            f4 t = {a + j, v, v, a};
            __builtin_nontemporal_store(t, reinterpret_cast<f4 *>(row + 4 * lane + 256 * j));
        }
    }
}

template <int SPIN> double run(float *bufs[2], int n_items, int wpb, int blocks)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int r = 0; r < 2; r++) hipLaunchKernelGGL(rows_kernel<SPIN>, dim3(blocks), dim3(64 * wpb), 0, 0, bufs[r & 1], n_items, 1.0f);
    CK(hipDeviceSynchronize());
    const int reps = 10;
    CK(hipEventRecord(e0));
    for (int r = 0; r < reps; r++) hipLaunchKernelGGL(rows_kernel<SPIN>, dim3(blocks), dim3(64 * wpb), 0, 0, bufs[r & 1], n_items, 1.0f);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    return ms * 1e3 / reps;
}

int main(int argc, char **argv)
{
    const int n_items = argc > 1 ? atoi(argv[1]) : 262144;
    hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
    const int cus = pr.multiProcessorCount;
    float *bufs[2];
    for (int i = 0; i < 2; i++) CK(hipMalloc(&bufs[i], (size_t)n_items * 4096));
    printf("%d rows of 4 KiB, %d CUs\n", n_items, cus);
    printf("-- waves per workgroup x workgroups per CU (no arithmetic)\n");
    for (int wpb : {1, 2, 3, 4, 5, 6, 8})
        for (int wpc : {8, 10, 12, 14, 15, 16, 18, 20, 24}) {
            if (wpc % wpb) continue;
            const int blocks = cus * wpc / wpb;
            const double us = run<0>(bufs, n_items, wpb, blocks);
            printf("wpb %d  wg/CU %2d  waves/CU %2d  waves %5d: %7.1f us  %5.2f TB/s\n", wpb, wpc / wpb, wpc, blocks * wpb, us, (double)n_items * 4096 / us / 1e6);
        }
    printf("-- total waves not a multiple of the CU count (4 waves per workgroup)\n");
    for (int blocks : {700, 750, 768, 800, 900, 960, 1000, 1023, 1024, 1025, 1050, 1100, 1152, 1200, 1280})
        printf("workgroups %4d  waves %5d (%.2f per CU): %7.1f us\n", blocks, blocks * 4, blocks * 4.0 / cus, run<0>(bufs, n_items, 4, blocks));
    printf("-- with a dependent chain of 600 FMAs per row (4 waves per workgroup and 3, 5)\n");
    for (int wpb : {3, 4, 5})
        for (int wpc : {12, 15, 16, 20}) {
            if (wpc % wpb) continue;
            const int blocks = cus * wpc / wpb;
            const double us = run<600>(bufs, n_items, wpb, blocks);
            printf("wpb %d  waves/CU %2d: %7.1f us\n", wpb, wpc, us);
        }
    return 0;
}
