// lane_streams.hip — streams for the overlap lanes of the pipeline handles (pipeline_lanes.hpp) that have been SEEN to run
// their kernels side by side.
//
// Why this exists (measured, round 4: profiles/r04_lab_lane_queue_sharing.txt).  The HIP runtime maps every stream onto one
// of a few hardware queues (four by default) when the stream is created, taking a queue with the fewest streams on it and
// breaking ties arbitrarily; kernels of streams that share a queue run one after the other.  With one foreign stream alive
// in the process -- an idle one is enough -- the fourth lane of a handle can therefore land on the queue of another LANE
// instead of on the idle stream's queue: rocprofv3 then shows the four lanes on three queues (80 / 40 / 40 dispatches) and
// the simulation flowgraph's step takes 44 us instead of 37.5.  Nothing in the HIP API tells which queue a stream got, so
// the lanes are probed: a candidate stream and the lanes accepted so far each run a one-wave kernel that sleeps for
// kProbeUs and stamps its start and end with the device's wall clock; the candidate is kept if its interval overlaps every
// other, otherwise it is set aside -- alive, so that the runtime counts it and hands the next candidate another queue -- and
// released when the lanes are complete.  Costs one short launch per lane and try, once per handle.
//
// A probe that cannot decide (the device is saturated by the caller's own work, so the probe kernels are delayed) costs
// retries, never correctness: after kMaxTries candidates a lane takes what it has.  DOA_HIP_NO_LANE_PROBE=1 in the
// environment switches the probing off (plain hipStreamCreateWithFlags, as before round 4).
#include "pipeline_lanes.hpp"

#include <atomic>
#include <vector>

namespace doa {

namespace {

constexpr int kProbeUs = 150;          // long against the few us between the launches of one probe round
constexpr int kMaxTries = 6;           // candidates per lane before it takes what it has

std::atomic<int> g_last_verified{-1};  // lanes of the last created set that passed the probe (-1: none created yet)
std::atomic<int> g_last_set_aside{0};

__global__ void lane_probe_kernel(unsigned long long *stamp, unsigned long long ticks)
{
    const unsigned long long t0 = wall_clock64();
    unsigned long long t = t0;
    // the wall clock advances whatever this wave does; the iteration bound is a second exit should it not (~1 us per turn)
    for (int guard = 0; guard < (1 << 16) && t - t0 < ticks; guard++) {
        __builtin_amdgcn_s_sleep(32);
        t = wall_clock64();
    }
    if (threadIdx.x == 0) {
        stamp[0] = t0;
        stamp[1] = t;
    }
}

struct Probe {
    DevBuf dev;
    unsigned long long ticks = 0;
    int init()
    {
        int dev_id = 0, khz = 0;
        DOA_HIP_TRY(hipGetDevice(&dev_id));
        if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev_id) != hipSuccess || khz <= 0) {
            (void)hipGetLastError();
            khz = 100000;                              // gfx9: 100 MHz
        }
        ticks = (unsigned long long)khz * kProbeUs / 1000;
        return dev.reserve(2 * sizeof(unsigned long long) * (PipeLanes::kMaxLanes + 1));
    }
    // 1: `cand` ran beside every stream of `acc`; 0: it did not; < 0: error
    int runs_beside(const std::vector<hipStream_t> &acc, hipStream_t cand)
    {
        const int n = (int)acc.size() + 1;
        unsigned long long *d = dev.as<unsigned long long>();
        // a stream's first dispatch can cost the host far more than a later one: spend it before the round that is judged
        hipLaunchKernelGGL(lane_probe_kernel, dim3(1), dim3(kWave), 0, cand, d + 2 * (n - 1), 0ull);
        DOA_HIP_TRY(hipStreamSynchronize(cand));
        for (int i = 0; i < n; i++)
            hipLaunchKernelGGL(lane_probe_kernel, dim3(1), dim3(kWave), 0, i + 1 < n ? acc[i] : cand, d + 2 * i, ticks);
        DOA_HIP_TRY(hipGetLastError());
        for (int i = 0; i < n; i++) DOA_HIP_TRY(hipStreamSynchronize(i + 1 < n ? acc[i] : cand));
        unsigned long long h[2 * (PipeLanes::kMaxLanes + 1)];
        DOA_HIP_TRY(hipMemcpy(h, d, 2 * sizeof(unsigned long long) * n, hipMemcpyDeviceToHost));
        const unsigned long long c0 = h[2 * (n - 1)], c1 = h[2 * (n - 1) + 1];
        for (int i = 0; i + 1 < n; i++) {
            const unsigned long long lo = h[2 * i] > c0 ? h[2 * i] : c0, hi = h[2 * i + 1] < c1 ? h[2 * i + 1] : c1;
            if (hi <= lo || hi - lo < ticks / 4) return 0;           // side by side for less than a quarter of the sleep
        }
        return 1;
    }
};

}  // namespace

int lane_streams_last_verified() { return g_last_verified.load(); }
int lane_streams_last_set_aside() { return g_last_set_aside.load(); }

int create_lane_streams(const hipStream_t *have, int n_have, hipStream_t *out, int n_new)
{
    static const bool off = [] { const char *e = getenv("DOA_HIP_NO_LANE_PROBE"); return e && *e && *e != '0'; }();
    std::vector<hipStream_t> acc(have, have + n_have), aside;
    Probe probe;
    bool probing = !off && n_have + n_new > 1 && n_have + n_new <= PipeLanes::kMaxLanes;
    if (probing && probe.init() != DOA_OK) probing = false;          // a set-up failure only costs the probing
    int verified = 0, rc = DOA_OK, made = 0;
    for (int k = 0; k < n_new && rc == DOA_OK; k++) {
        hipStream_t got = nullptr;
        for (int t = 0; t < kMaxTries && !got && rc == DOA_OK; t++) {
            hipStream_t c = nullptr;
            const hipError_t e = hipStreamCreateWithFlags(&c, hipStreamNonBlocking);
            if (e != hipSuccess) {
                set_error("lane stream: hipStreamCreateWithFlags failed: %s", hipGetErrorString(e));
                rc = DOA_ERR_HIP;
                break;
            }
            if (!probing || acc.empty()) { got = c; break; }
            const int beside = probe.runs_beside(acc, c);
            if (beside > 0) { got = c; verified++; }
            else if (beside == 0) aside.push_back(c);
            else { got = c; probing = false; }                       // the probe itself failed: stop probing, keep the stream
        }
        if (rc != DOA_OK) break;
        if (!got) {                                                  // no candidate ran beside the others (more lanes than
            got = aside.back();                                      // hardware queues, or a saturated device): take the
            aside.pop_back();                                        // last one and stop looking
            probing = false;
        }
        acc.push_back(got);
        out[made++] = got;
    }
    for (hipStream_t s : aside) (void)hipStreamDestroy(s);
    if (rc != DOA_OK) {
        for (int k = 0; k < made; k++) (void)hipStreamDestroy(out[k]);
        probe.dev.release();
        return rc;
    }
    probe.dev.release();
    // lanes of the set known to run side by side: the ones handed in, the first new one when there was nothing to compare it
    // with, and every later one that passed
    g_last_verified.store(off ? -1 : n_have + verified + (n_have == 0 && n_new > 0 ? 1 : 0));
    g_last_set_aside.store((int)aside.size());
    return DOA_OK;
}

}  // namespace doa

extern "C" {
int doa_hip_lane_streams_verified_debug(void) { return doa::lane_streams_last_verified(); }
int doa_hip_lane_streams_set_aside_debug(void) { return doa::lane_streams_last_set_aside(); }
}
