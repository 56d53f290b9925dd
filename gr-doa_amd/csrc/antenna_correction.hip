// antenna_correction.hip — per-stream complex gain (the step in front of autocorrelate in every
// calibrated gr-doa flowgraph), standalone and fused into K1.
//
// Replaces gr::doa::antenna_correction (reference lib/antenna_correction_impl.cc:47-99):
//   ctor  :56-73  config file with one "gain phase" pair per line ->
//                 g_k = complex(1.0/gain, 0) * exp(complex(0, -phase)); missing file, too many or too
//                 few lines -> std::invalid_argument (here: create fails with the same message text)
//   work  :85-99  out_k[i] = g_k * in_k[i] for every stream k
//
// Standalone kernel: pure streaming (8 B in, 8 B out per sample), 16-byte accesses, grid-stride.
// Fused form (doa_autocorrelate_fuse_antenna_correction): since (g_a x_a)(g_b x_b)^* = g_a conj(g_b)
// x_a x_b^*, the correction is applied to the N x N covariance in K1's epilogue
// (R[a,b] *= g_a conj(g_b), before the forward-backward step) and the corrected streams are never
// written to or re-read from HBM — that removes 16 B/sample/stream of traffic, twice what K1 itself
// moves.  The two forms agree to fp32 rounding (~1e-7 relative).
#include "kernels.hpp"

#include <cmath>
#include <complex>
#include <fstream>
#include <string>
#include <vector>

namespace doa {

struct GainArgs {
    const float2 *in[DOA_MAX_ANT_ELE];
    float2 *out[DOA_MAX_ANT_ELE];
    float2 g[DOA_MAX_ANT_ELE];
    long long n;     // samples per stream
    int n_ch;
};

__device__ __forceinline__ float2 cmul_nofma(float2 g, float2 x)
{
    // std::complex<float> product without contraction: (gr*xr - gi*xi, gr*xi + gi*xr)
    return make_float2(__fsub_rn(__fmul_rn(g.x, x.x), __fmul_rn(g.y, x.y)),
                       __fadd_rn(__fmul_rn(g.x, x.y), __fmul_rn(g.y, x.x)));
}

template <bool VEC2> __global__ __launch_bounds__(256) void antenna_correction_kernel(GainArgs a)
{
    const int k = blockIdx.y;
    const float2 g = a.g[k];
    const float2 *in = a.in[k];
    float2 *out = a.out[k];
    const long long stride = (long long)gridDim.x * blockDim.x;
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if constexpr (VEC2) {
        const long long npair = a.n >> 1;
        for (; i < npair; i += stride) {
            const float4 v = load_f4<true>(reinterpret_cast<const float4 *>(in) + i);
            const float2 y0 = cmul_nofma(g, make_float2(v.x, v.y)), y1 = cmul_nofma(g, make_float2(v.z, v.w));
            store_f4<true>(reinterpret_cast<float4 *>(out) + i, make_float4(y0.x, y0.y, y1.x, y1.y));
        }
        if ((a.n & 1) && blockIdx.x == 0 && threadIdx.x == 0) out[a.n - 1] = cmul_nofma(g, in[a.n - 1]);
    } else {
        for (; i < a.n; i += stride) out[i] = cmul_nofma(g, in[i]);
    }
}

int launch_antenna_correction(int N, const float *gains_re_im, long long n, const void *const *d_in, void *const *d_out,
                              hipStream_t st)
{
    if (n <= 0) return DOA_OK;
    GainArgs a;
    memset(&a, 0, sizeof a);
    bool vec2 = true;
    for (int k = 0; k < N; k++) {
        if (!d_in[k] || !d_out[k]) { set_error("antenna_correction: stream %d is NULL", k); return DOA_ERR_INVALID_ARG; }
        a.in[k] = static_cast<const float2 *>(d_in[k]);
        a.out[k] = static_cast<float2 *>(d_out[k]);
        a.g[k] = make_float2(gains_re_im[2 * k], gains_re_im[2 * k + 1]);
        if ((reinterpret_cast<uintptr_t>(d_in[k]) | reinterpret_cast<uintptr_t>(d_out[k])) % 16) vec2 = false;
    }
    a.n = n; a.n_ch = N;
    long long work = vec2 ? (n + 1) / 2 : n;
    int bx = (int)((work + 255) / 256);
    if (bx > 2048) bx = 2048;
    dim3 grid(bx, N), block(256);
    if (vec2) hipLaunchKernelGGL(antenna_correction_kernel<true>, grid, block, 0, st, a);
    else      hipLaunchKernelGGL(antenna_correction_kernel<false>, grid, block, 0, st, a);
    DOA_HIP_TRY(hipGetLastError());
    return DOA_OK;
}

}  // namespace doa

struct doa_antenna_correction {
    int N = 0;
    int device = 0;
    float gains[2 * DOA_MAX_ANT_ELE] = {0};
    hipStream_t stream = nullptr;
    doa::DevBuf d_in, d_out;
};

extern "C" {

doa_antenna_correction_t *doa_antenna_correction_create(int num_ant_ele, const char *config_filename)
{
    doa::clear_error();
    if (num_ant_ele <= 0 || num_ant_ele > DOA_MAX_ANT_ELE) {
        doa::set_error("antenna_correction: num_ant_ele=%d outside 1..%d", num_ant_ele, DOA_MAX_ANT_ELE);
        return nullptr;
    }
    if (!config_filename) { doa::set_error("antenna_correction: config_filename is NULL"); return nullptr; }
    // lib/antenna_correction_impl.cc:56-73 — same parsing (operator>> on floats), same messages
    std::ifstream infile(config_filename);
    if (!infile.good()) { doa::set_error("Cannot find configuration file."); return nullptr; }
    std::vector<std::complex<float>> g;
    float GainEst, PhaseEst;
    while (infile >> GainEst >> PhaseEst) {
        if ((int)g.size() >= num_ant_ele) { doa::set_error("Configuration file has too many inputs."); return nullptr; }
        g.push_back(std::complex<float>((float)(1.0 / GainEst), 0) * std::exp(std::complex<float>(0, -PhaseEst)));
    }
    if ((int)g.size() != num_ant_ele) { doa::set_error("Configuration file does not have enough inputs."); return nullptr; }
    int dev = 0;
    if (doa::ensure_device(&dev) != DOA_OK) return nullptr;
    auto *h = new (std::nothrow) doa_antenna_correction();
    if (!h) { doa::set_error("out of memory"); return nullptr; }
    h->N = num_ant_ele; h->device = dev;
    for (int k = 0; k < num_ant_ele; k++) { h->gains[2 * k] = g[k].real(); h->gains[2 * k + 1] = g[k].imag(); }
    if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) {
        doa::set_error("antenna_correction: hipStreamCreate failed");
        delete h;
        return nullptr;
    }
    return h;
}

doa_antenna_correction_t *doa_antenna_correction_create_gains(int num_ant_ele, const float *gains_re_im)
{
    doa::clear_error();
    if (num_ant_ele <= 0 || num_ant_ele > DOA_MAX_ANT_ELE) {
        doa::set_error("antenna_correction: num_ant_ele=%d outside 1..%d", num_ant_ele, DOA_MAX_ANT_ELE);
        return nullptr;
    }
    if (!gains_re_im) { doa::set_error("antenna_correction: gains_re_im is NULL"); return nullptr; }
    int dev = 0;
    if (doa::ensure_device(&dev) != DOA_OK) return nullptr;
    auto *h = new (std::nothrow) doa_antenna_correction();
    if (!h) { doa::set_error("out of memory"); return nullptr; }
    h->N = num_ant_ele; h->device = dev;
    memcpy(h->gains, gains_re_im, sizeof(float) * 2 * num_ant_ele);
    if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) {
        doa::set_error("antenna_correction: hipStreamCreate failed");
        delete h;
        return nullptr;
    }
    return h;
}

void doa_antenna_correction_destroy(doa_antenna_correction_t *h)
{
    if (!h) return;
    h->d_in.release(); h->d_out.release();
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

int doa_antenna_correction_gains(const doa_antenna_correction_t *h, float *gains_re_im)
{
    if (!h || !gains_re_im) return DOA_ERR_INVALID_ARG;
    memcpy(gains_re_im, h->gains, sizeof(float) * 2 * h->N);
    return h->N;
}

int doa_antenna_correction_work_dev(doa_antenna_correction_t *h, int noutput_items, const void *const *d_input_items,
                                    void *const *d_output_items, void *hip_stream)
{
    doa::clear_error();
    if (!h || noutput_items < 0 || !d_input_items || !d_output_items) {
        doa::set_error("antenna_correction_work_dev: bad arguments");
        return DOA_ERR_INVALID_ARG;
    }
    if (int brc = doa::bind_device(h->device); brc != DOA_OK) return brc;
    int rc = doa::launch_antenna_correction(h->N, h->gains, noutput_items, d_input_items, d_output_items,
                                            static_cast<hipStream_t>(hip_stream));
    return rc == DOA_OK ? noutput_items : rc;
}

int doa_antenna_correction_work(doa_antenna_correction_t *h, int noutput_items, const void *const *input_items,
                                void *const *output_items)
{
    doa::clear_error();
    if (!h || noutput_items < 0 || !input_items || !output_items) {
        doa::set_error("antenna_correction_work: bad arguments");
        return DOA_ERR_INVALID_ARG;
    }
    if (noutput_items == 0) return 0;
    if (int brc = doa::bind_device(h->device); brc != DOA_OK) return brc;
    const size_t n = (size_t)noutput_items, n_al = (n + 1) & ~(size_t)1;
    int rc = h->d_in.reserve(n_al * h->N * sizeof(float2));
    if (rc == DOA_OK) rc = h->d_out.reserve(n_al * h->N * sizeof(float2));
    if (rc != DOA_OK) return rc;
    const void *di[DOA_MAX_ANT_ELE];
    void *dout[DOA_MAX_ANT_ELE];
    for (int k = 0; k < h->N; k++) {
        if (!input_items[k] || !output_items[k]) { doa::set_error("antenna_correction_work: port %d is NULL", k); return DOA_ERR_INVALID_ARG; }
        di[k] = h->d_in.as<float2>() + k * n_al;
        dout[k] = h->d_out.as<float2>() + k * n_al;
        DOA_HIP_TRY(hipMemcpyAsync(const_cast<void *>(di[k]), input_items[k], n * sizeof(float2), hipMemcpyHostToDevice, h->stream));
    }
    rc = doa::launch_antenna_correction(h->N, h->gains, noutput_items, di, dout, h->stream);
    if (rc != DOA_OK) return rc;
    for (int k = 0; k < h->N; k++)
        DOA_HIP_TRY(hipMemcpyAsync(output_items[k], dout[k], n * sizeof(float2), hipMemcpyDeviceToHost, h->stream));
    DOA_HIP_TRY(hipStreamSynchronize(h->stream));
    return noutput_items;
}

}  // extern "C"
