// pipeline_lanes.hpp — the overlap lanes shared by the fused pipeline handles (music_pipeline: pipeline.hip; root_pipeline:
// root_pipeline.hip): a HIP stream plus a private set of workspace buffers each, batches of one call spread over them in
// rotation, one fork and one join per call when the call is attached to a caller stream, none in the detached form.
// The reference has nothing of the kind: GNU Radio runs its blocks on one thread each, which is where its overlap comes from.
#pragma once
#include "common.hpp"

#include <string>

namespace doa {

// n_new streams that were seen to run their kernels beside each other and beside the n_have given ones (lane_streams.hip)
int create_lane_streams(const hipStream_t *have, int n_have, hipStream_t *out, int n_new);

// the two copy / compute streams of the host-buffer entries, created on first use: the second is probed against the first
inline int ensure_stream_pair(hipStream_t (&st)[2])
{
    for (int i = 0; i < 2; i++) {
        if (st[i]) continue;
        if (const int rc = create_lane_streams(&st[1 - i], st[1 - i] ? 1 : 0, &st[i], 1); rc != DOA_OK) return rc;
    }
    return DOA_OK;
}

struct PipeLane {
    hipStream_t st = nullptr;
    bool own_stream = true;         // false: adopted from the caller (doa_*_pipeline_set_lane_streams)
    hipEvent_t done = nullptr;
    static constexpr int kBufs = 6;
    DevBuf buf[kBufs];              // what they hold is the owning pipeline's business
};

struct PipeLanes {
    static constexpr int kMaxLanes = 8;
    PipeLane lanes[kMaxLanes];
    int n_lanes = 4;
    int next_lane = 0;              // lanes keep rotating across calls
    hipEvent_t fork_ev = nullptr;
    int fail_batch = -1;            // test aid (doa_*_pipeline_inject_failure): one-shot, cleared by every call

    void release()
    {
        for (auto &l : lanes) {
            for (auto &b : l.buf) b.release();
            if (l.done) (void)hipEventDestroy(l.done);
            if (l.st && l.own_stream) (void)hipStreamDestroy(l.st);
            l = PipeLane();
        }
        if (fork_ev) (void)hipEventDestroy(fork_ev);
        fork_ev = nullptr;
    }
    int set_count(int n)
    {
        if (n < 1 || n > kMaxLanes) return DOA_ERR_INVALID_ARG;
        n_lanes = n;
        next_lane = 0;
        return DOA_OK;
    }
    int adopt(int n, void *const *hip_streams)
    {
        if (n < 1 || n > kMaxLanes || !hip_streams) return DOA_ERR_INVALID_ARG;
        for (int l = 0; l < n; l++) {
            auto &ln = lanes[l];
            if (ln.st) { (void)hipStreamSynchronize(ln.st); if (ln.own_stream) (void)hipStreamDestroy(ln.st); }
            ln.st = static_cast<hipStream_t>(hip_streams[l]);
            ln.own_stream = false;
        }
        n_lanes = n;
        next_lane = 0;
        return DOA_OK;
    }
    int synchronize()
    {
        for (auto &l : lanes)
            if (l.st) DOA_HIP_TRY(hipStreamSynchronize(l.st));
        return DOA_OK;
    }
    bool idle() const
    {
        for (auto &l : lanes)
            if (l.st && hipStreamQuery(l.st) != hipSuccess) return false;
        return true;
    }

    // Batch b of the call runs launch(b, lane) on lane (next_lane + b) % n_lanes, in order on that lane.  `prepare(lane)` sizes
    // a lane's buffers (first use / growth); `launch` returns < 0 on failure.  Attached to a stream, the call forks (every lane
    // waits for what the caller's stream holds now: one event) and joins (the caller's stream waits for every lane: one event
    // per lane) ONCE, whatever n_batches is: cross-stream events cost tens of microseconds on this runtime and an event per
    // batch degrades the lanes to the serial rate -- hence also the detached form (caller == DETACHED: no
    // event at all, the caller joins with synchronize()).
    // Error contract: whatever fails, from the first lane operation on, the join is still enqueued and -- on an error return --
    // every lane the call used has been synchronised: "nothing of a failed call is still running", the detached form included.
    template <class Prepare, class Launch>
    int run_batches(const char *what, int n_batches, void *hip_stream, Prepare prepare, Launch launch)
    {
        const bool detached = (hip_stream == DOA_STREAM_DETACHED);
        hipStream_t caller = detached ? nullptr : static_cast<hipStream_t>(hip_stream);
        const int fail_at = fail_batch;
        fail_batch = -1;                                   // one-shot: armed for THIS call only, wherever it ends
        const int L = n_lanes;
        // set-up that cannot leave work behind: failures here return at once
        if (!detached && !fork_ev) DOA_HIP_TRY(hipEventCreateWithFlags(&fork_ev, hipEventDisableTiming));
        {
            hipStream_t have[kMaxLanes], fresh[kMaxLanes];
            int n_have = 0, n_new = 0;
            for (int l = 0; l < L; l++)
                if (lanes[l].st) have[n_have++] = lanes[l].st;
            if (n_have < L) {
                if (const int rc = create_lane_streams(have, n_have, fresh, L - n_have); rc != DOA_OK) return rc;
                for (int l = 0; l < L; l++)
                    if (!lanes[l].st) { lanes[l].st = fresh[n_new++]; lanes[l].own_stream = true; }
            }
        }
        for (int l = 0; l < L; l++) {
            auto &ln = lanes[l];
            if (!detached && !ln.done) DOA_HIP_TRY(hipEventCreateWithFlags(&ln.done, hipEventDisableTiming));
            if (const int rc = prepare(ln); rc != DOA_OK) return rc;
        }
        const int lane0 = next_lane % L;
        next_lane = (lane0 + n_batches) % L;
        const int used = n_batches < L ? n_batches : L;
        int rc = DOA_OK;
        auto fail = [&](hipError_t e, const char *step) {
            if (e != hipSuccess && rc >= 0) { set_error("%s: %s failed: %s", what, step, hipGetErrorString(e)); rc = DOA_ERR_HIP; }
        };
        // fork: from here on every exit goes through the join and, on failure, the lane synchronisation below
        if (!detached) {
            fail(hipEventRecord(fork_ev, caller), "fork record");
            for (int u = 0; u < used && rc >= 0; u++) fail(hipStreamWaitEvent(lanes[(lane0 + u) % L].st, fork_ev, 0), "fork wait");
        }
        for (int b = 0; b < n_batches && rc >= 0; b++) {
            if (fail_at == b) {
                set_error("%s: injected failure in batch %d", what, b);
                rc = DOA_ERR_HIP;
                break;
            }
            const int r = launch(b, lanes[(lane0 + b) % L]);
            if (r < 0) rc = r;
        }
        for (int u = 0; u < used && !detached; u++) {
            auto &ln = lanes[(lane0 + u) % L];
            const hipError_t e1 = hipEventRecord(ln.done, ln.st);
            fail(e1 == hipSuccess ? hipStreamWaitEvent(caller, ln.done, 0) : e1, "join");
        }
        if (rc < 0) {
            const std::string msg = doa_last_error();
            for (int u = 0; u < used; u++) (void)hipStreamSynchronize(lanes[(lane0 + u) % L].st);
            set_error("%s", msg.c_str());
        }
        return rc;
    }
};

}  // namespace doa
