// event_cost.hip -- lab: what a fork/join of L internal streams around a caller stream costs on this runtime.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ void spin(float *p, int iters)
{
    float v = p[threadIdx.x];
    for (int i = 0; i < iters; i++) v = v * 1.0000001f + 1e-7f;
    p[threadIdx.x] = v;
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main()
{
    float *d; CK(hipMalloc(&d, 4096 * 64));
    const int L = 4;
    hipStream_t caller, lane[L];
    CK(hipStreamCreateWithFlags(&caller, hipStreamNonBlocking));
    for (auto &s : lane) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t fork_ev, join_ev[L];
    for (unsigned flags : {(unsigned)hipEventDisableTiming, (unsigned)hipEventDefault}) {
        CK(hipEventCreateWithFlags(&fork_ev, flags));
        for (auto &e : join_ev) CK(hipEventCreateWithFlags(&e, flags));
        for (int per_lane : {1, 5, 20}) {
            for (int iters : {100, 20000}) {                 // ~1 us and ~25+ us kernels
                auto call = [&]() {
                    CK(hipEventRecord(fork_ev, caller));
                    for (int l = 0; l < L; l++) CK(hipStreamWaitEvent(lane[l], fork_ev, 0));
                    for (int k = 0; k < per_lane; k++)
                        for (int l = 0; l < L; l++) hipLaunchKernelGGL(spin, dim3(64), dim3(64), 0, lane[l], d + 4096 * l, iters);
                    for (int l = 0; l < L; l++) { CK(hipEventRecord(join_ev[l], lane[l])); CK(hipStreamWaitEvent(caller, join_ev[l], 0)); }
                };
                auto plain = [&]() {     // the same kernels with no events at all (caller-side streams, host join)
                    for (int k = 0; k < per_lane; k++)
                        for (int l = 0; l < L; l++) hipLaunchKernelGGL(spin, dim3(64), dim3(64), 0, lane[l], d + 4096 * l, iters);
                };
                for (int mode = 0; mode < 3; mode++) {
                    // mode 0: plain + sync every lane on the host; mode 1: fork/join, sync caller; mode 2: 10 fork/join calls back to back, one sync
                    const int reps = 30;
                    for (int w = 0; w < 3; w++) { call(); CK(hipStreamSynchronize(caller)); }
                    CK(hipDeviceSynchronize());
                    double t0 = now();
                    if (mode == 0) for (int r = 0; r < reps; r++) { plain(); for (auto s : lane) CK(hipStreamSynchronize(s)); }
                    if (mode == 1) for (int r = 0; r < reps; r++) { call(); CK(hipStreamSynchronize(caller)); }
                    if (mode == 2) { for (int r = 0; r < reps; r++) call(); CK(hipStreamSynchronize(caller)); }
                    double dt = (now() - t0) / reps * 1e6;
                    printf("flags %u  kernels/lane %2d  iters %5d  %-28s %8.1f us per call\n", flags, per_lane, iters,
                           mode == 0 ? "plain + host sync of 4 lanes" : mode == 1 ? "fork/join + sync caller" : "fork/join x30, one sync", dt);
                }
            }
        }
    }
    return 0;
}
