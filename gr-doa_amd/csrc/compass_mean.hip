// compass_mean.hip — the consumer side of the hot path: the per-call average the
// reference's `compass` sink displays, taken on the device so that the per-snapshot angles need not
// cross PCIe at all.
//
// Replaces, as a unit, blocks.vector_to_streams(gr.sizeof_float, num_targets)
// (apps/run_MUSIC_lin_array_simulation.py:199, connections :236-239: port 1 of find_local_max, item =
// num_targets floats, de-interleaved into num_targets float streams) followed by one doa.compass per
// stream, whose work() is `self.next_angle = numpy.mean(input_items[0])` over the items of the call
// (python/compass.py:134-136).  The Qt dial/LCD of the compass is GUI and out of scope.
//
//   next_angle[m] = mean_i in[i * num_streams + m],  i = 0 .. ninput_items-1
//
// numpy.mean of a float32 array accumulates pairwise in float32; here the sum is carried in double
// and rounded once, so the result is the correctly rounded mean (numpy's differs from it by <= a few
// ulp).  ninput_items == 0 gives NaN, as numpy.mean of an empty array does.
#include "kernels.hpp"

#include <cmath>

namespace doa {

__global__ __launch_bounds__(256) void compass_mean_kernel(const float *__restrict__ in, float *__restrict__ out, int n, int M)
{
    __shared__ double part[256];
    const int m = blockIdx.x;
    double acc = 0.0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) acc += (double)in[(size_t)i * M + m];
    part[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {          // fixed tree: the result does not depend on scheduling
        if ((int)threadIdx.x < s) part[threadIdx.x] += part[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[m] = (n > 0) ? (float)(part[0] / (double)n) : nanf("");
}

}  // namespace doa

struct doa_compass_mean {
    int M = 0;
    int device = 0;
    hipStream_t stream = nullptr;
    doa::DevBuf d_in, d_out;
};

extern "C" {

doa_compass_mean_t *doa_compass_mean_create(int num_streams)
{
    doa::clear_error();
    if (num_streams <= 0 || num_streams > DOA_MAX_PEAKS) {
        doa::set_error("compass_mean: num_streams=%d outside 1..%d", num_streams, DOA_MAX_PEAKS);
        return nullptr;
    }
    int dev = 0;
    if (doa::ensure_device(&dev) != DOA_OK) return nullptr;
    auto *h = new (std::nothrow) doa_compass_mean();
    if (!h) { doa::set_error("out of memory"); return nullptr; }
    h->M = num_streams; h->device = dev;
    if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) {
        doa::set_error("compass_mean: hipStreamCreate failed");
        delete h;
        return nullptr;
    }
    return h;
}

void doa_compass_mean_destroy(doa_compass_mean_t *h)
{
    if (!h) return;
    h->d_in.release(); h->d_out.release();
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

int doa_compass_mean_work_dev(doa_compass_mean_t *h, int ninput_items, const void *d_input_items0, float *d_next_angle,
                              void *hip_stream)
{
    doa::clear_error();
    if (!h || ninput_items < 0 || !d_next_angle || (ninput_items > 0 && !d_input_items0)) {
        doa::set_error("compass_mean_work_dev: bad arguments");
        return DOA_ERR_INVALID_ARG;
    }
    if (int brc = doa::bind_device(h->device); brc != DOA_OK) return brc;
    hipLaunchKernelGGL(doa::compass_mean_kernel, dim3(h->M), dim3(256), 0, static_cast<hipStream_t>(hip_stream),
                       static_cast<const float *>(d_input_items0), d_next_angle, ninput_items, h->M);
    DOA_HIP_TRY(hipGetLastError());
    return ninput_items;          // "consume all inputs" (python/compass.py:143-144)
}

int doa_compass_mean_work(doa_compass_mean_t *h, int ninput_items, const void *input_items0, float *next_angle)
{
    doa::clear_error();
    if (!h || ninput_items < 0 || !next_angle || (ninput_items > 0 && !input_items0)) {
        doa::set_error("compass_mean_work: bad arguments");
        return DOA_ERR_INVALID_ARG;
    }
    if (int brc = doa::bind_device(h->device); brc != DOA_OK) return brc;
    const size_t bytes = (size_t)ninput_items * h->M * sizeof(float);
    int rc = h->d_in.reserve(bytes ? bytes : sizeof(float));
    if (rc == DOA_OK) rc = h->d_out.reserve(h->M * sizeof(float));
    if (rc != DOA_OK) return rc;
    if (bytes) DOA_HIP_TRY(hipMemcpyAsync(h->d_in.p, input_items0, bytes, hipMemcpyHostToDevice, h->stream));
    rc = doa_compass_mean_work_dev(h, ninput_items, h->d_in.p, h->d_out.as<float>(), h->stream);
    if (rc < 0) return rc;
    DOA_HIP_TRY(hipMemcpyAsync(next_angle, h->d_out.p, h->M * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    DOA_HIP_TRY(hipStreamSynchronize(h->stream));
    return ninput_items;
}

}  // extern "C"
