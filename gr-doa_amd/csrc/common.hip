// common.hip — error state, device binding and buffer helpers of libdoa_hip.so.
#include "common.hpp"

#include <atomic>
#include <mutex>

namespace doa {

static thread_local char g_err[512] = "";
static std::atomic<int> g_prec_bits{64};

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}
void clear_error() { g_err[0] = '\0'; }

void prime_hw_queues(int dev);

int ensure_device(int *device_out)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        set_error("no HIP device available (%s); libdoa_hip has no CPU fallback",
                  e != hipSuccess ? hipGetErrorString(e) : "device count is 0");
        (void)hipGetLastError();
        return DOA_ERR_NO_DEVICE;
    }
    int dev = 0;
    DOA_HIP_TRY(hipGetDevice(&dev));
    if (device_out) *device_out = dev;
    prime_hw_queues(dev);
    return DOA_OK;
}

// The HIP runtime creates its hardware queues lazily, one per newly USED stream until its pool (4 by default) is full, and
// maps later streams onto the existing ones.  Streams that caused a queue to be created overlap their kernels measurably
// worse than streams mapped onto a populated pool: four pipeline chains on the first four streams of a process run at
// 32.6-32.9 us per 4096-snapshot step, on any later four at 26.0-26.8 (tools/lab/lanes_sweep.py with PRE_STREAMS = 0 / 3+,
// same box, whoever creates the streams).  So the pool is populated once per device, before any handle creates a stream:
// four throw-away streams, one trivial operation each; they stay alive for the life of the process (64 bytes of device
// memory) -- whether the hardware queues would outlive their streams is not documented, so the streams are kept.
// This is an effect on the HOST PROCESS (include/doa_hip.h and INTEGRATION.md say so); DOA_HIP_NO_QUEUE_PRIMING=1 in the
// environment switches it off.  One thread primes a device; others creating handles on it meanwhile wait for it (the point
// is that no lane stream exists before the pool is populated), and a failure is not remembered as success.
void prime_hw_queues(int dev)
{
    static const bool off = [] { const char *e = getenv("DOA_HIP_NO_QUEUE_PRIMING"); return e && *e && *e != '0'; }();
    if (off || dev < 0 || dev >= 64) return;
    static std::mutex mtx;
    static bool primed[64] = {};
    std::lock_guard<std::mutex> lock(mtx);
    if (primed[dev]) return;
    void *scratch = nullptr;
    if (hipMalloc(&scratch, 64) != hipSuccess) { (void)hipGetLastError(); return; }
    int made = 0;
    for (int i = 0; i < 4; i++) {
        hipStream_t s = nullptr;
        if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); break; }
        if (hipMemsetAsync(scratch, 0, 64, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) { (void)hipGetLastError(); break; }
        made++;
    }
    primed[dev] = (made == 4);
}

int cu_count()
{
    // compute units of the current device (MI355X: 256), cached per device; launch caps scale with it
    static std::atomic<int> cached[16];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return 256;
    int v = cached[dev].load(std::memory_order_relaxed);
    if (v > 0) return v;
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    cached[dev].store(n, std::memory_order_relaxed);
    return n;
}

int bind_device(int device)
{
    int cur = -1;
    DOA_HIP_TRY(hipGetDevice(&cur));
    if (cur != device) DOA_HIP_TRY(hipSetDevice(device));
    return DOA_OK;
}

int DevBuf::reserve(size_t bytes)
{
    if (bytes <= cap) return DOA_OK;
    if (p) {
        DOA_HIP_TRY(hipFree(p));
        p = nullptr;
        cap = 0;
    }
    size_t want = bytes + bytes / 4;  // amortise growth
    if (want < 4096) want = 4096;
    DOA_HIP_TRY(hipMalloc(&p, want));
    cap = want;
    return DOA_OK;
}
void DevBuf::release()
{
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
}
int PinnedBuf::reserve(size_t bytes)
{
    if (bytes <= cap) return DOA_OK;
    if (p) {
        DOA_HIP_TRY(hipHostFree(p));
        p = nullptr;
        cap = 0;
    }
    size_t want = bytes + bytes / 4;
    if (want < 4096) want = 4096;
    DOA_HIP_TRY(hipHostMalloc(&p, want, hipHostMallocDefault));
    cap = want;
    return DOA_OK;
}
void PinnedBuf::release()
{
    if (p) (void)hipHostFree(p);
    p = nullptr;
    cap = 0;
}

int internal_precision_bits() { return g_prec_bits.load(); }

}  // namespace doa

extern "C" {

const char *doa_last_error(void) { return doa::g_err; }
int doa_hip_abi_version(void) { return 1; }

int doa_hip_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

size_t doa_stream_stride_bytes(size_t stream_bytes) { return doa::stream_stride_bytes(stream_bytes); }

int doa_set_internal_precision(int bits)
{
    if (bits != 32 && bits != 64) {
        doa::set_error("doa_set_internal_precision: bits must be 32 or 64 (got %d)", bits);
        return DOA_ERR_INVALID_ARG;
    }
    doa::g_prec_bits.store(bits);
    return DOA_OK;
}
int doa_get_internal_precision(void) { return doa::g_prec_bits.load(); }

}  // extern "C"
