// sim_source.hip — the signal front end of the reference's simulation flowgraphs as one device-side
// generator, so a whole simulation runs without a host-side sample path.
//
// Replaces, as a unit, the GNU Radio core blocks wired in
// apps/run_MUSIC_lin_array_simulation.py:66-74 (array manifold matrix from ant_locs / theta) and
// :204-210 (analog.sig_source_c per source + analog.noise_source_c(GR_GAUSSIAN, ampl) per source ->
// blocks.add -> blocks.multiply_matrix_cc(array_manifold_matrix)):
//
//   s_m[t] = tone_ampl_m * exp(j 2 pi f_m t) + src_noise_m * (g + j g')          (per source m)
//   x_n[t] = sum_m A[n][m] s_m[t] + ant_sigma * (g + j g') / sqrt(2)             (per antenna n)
//   A[n][m] = exp(-j 2 pi cos(theta_m) loc_n),  loc_n = d * ((N-1)/2 - n)         (:70-73)
//
// (the per-antenna term is the SNR model of examples/@wpi_twinrx_doa_testbench/music_test_input_gen.m
// :97-107 and of SURVEY §8d's benchmark workload; the flowgraphs themselves use ant_sigma = 0).
// g are standard normals from a counter-based generator — Philox4x32-10 (Salmon et al., SC'11), key =
// seed, counter = (sample pair index, noise stream id) -> 4 words -> two Box-Muller pairs = the
// complex samples t = 2i and 2i+1 of that noise stream — so any sample range of any stream can be
// produced independently (seek, sharding across GPUs) and a numpy restatement reproduces the integers
// exactly.  GNU Radio's own generator (gr::random, a fixed-point NCO for the tone) is not reproduced:
// the streams are statistically, not sample-for-sample, those of the reference ("parity unpinned").
//
// The kernel is write-bound in principle (8 N bytes per sample); at small N the Philox rounds make it
// ALU-bound instead, which is irrelevant for a setup step.  One thread produces two consecutive samples
// of every stream (16-byte stores, 1 KiB contiguous per wave instruction per stream).
#include "kernels.hpp"

#include <cmath>

namespace doa {

struct SimArgs {
    float2 *out[DOA_MAX_ANT_ELE];
    float2 A[DOA_MAX_ANT_ELE * DOA_MAX_PEAKS];     // [n * M + m]
    double freq[DOA_MAX_PEAKS];
    float ampl[DOA_MAX_PEAKS];
    float src_noise[DOA_MAX_PEAKS];
    float ant_sigma;
    unsigned key0, key1;
    long long t0;      // absolute index of out[.][0] (even)
    long long n;       // samples per stream
    int N, M;
};

__device__ __forceinline__ void philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1,
                                              unsigned (&r)[4])
{
#pragma unroll
    for (int round = 0; round < 10; round++) {
        const unsigned hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const unsigned hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const unsigned n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    r[0] = c0; r[1] = c1; r[2] = c2; r[3] = c3;
}

// two standard normals from two 32-bit words: u = ((w >> 9) + 0.5) * 2^-23 in (0, 1), exactly representable
__device__ __forceinline__ float2 box_muller(unsigned w0, unsigned w1)
{
    const float u1 = ((float)(w0 >> 9) + 0.5f) * 1.1920928955078125e-7f;
    const float u2 = ((float)(w1 >> 9) + 0.5f) * 1.1920928955078125e-7f;
    const float r = sqrtf(-2.0f * logf(u1));
    float s, c;
    sincospif(2.0f * u2, &s, &c);
    return make_float2(r * c, r * s);
}

// complex samples t = 2i, 2i+1 of noise stream `sid`
__device__ __forceinline__ void noise_pair(long long pair, unsigned sid, unsigned k0, unsigned k1, float2 &g0, float2 &g1)
{
    unsigned r[4];
    philox4x32_10((unsigned)(pair & 0xFFFFFFFFll), (unsigned)((unsigned long long)pair >> 32), sid, 0u, k0, k1, r);
    g0 = box_muller(r[0], r[1]);
    g1 = box_muller(r[2], r[3]);
}

__device__ __forceinline__ float2 tone(double f, long long t, float ampl)
{
    const double cyc = f * (double)t;
    const float frac = (float)(cyc - floor(cyc));
    float s, c;
    sincospif(2.0f * frac, &s, &c);
    return make_float2(ampl * c, ampl * s);
}

__global__ __launch_bounds__(256) void sim_source_kernel(SimArgs a)
{
    const long long n_pairs = (a.n + 1) >> 1;
    const long long stride = (long long)gridDim.x * blockDim.x;
    const long long pair0 = a.t0 >> 1;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n_pairs; i += stride) {
        const long long pair = pair0 + i, t = 2 * pair;
        float2 s0[DOA_MAX_PEAKS], s1[DOA_MAX_PEAKS];
        for (int m = 0; m < a.M; m++) {
            s0[m] = tone(a.freq[m], t, a.ampl[m]);
            s1[m] = tone(a.freq[m], t + 1, a.ampl[m]);
            if (a.src_noise[m] != 0.0f) {
                float2 g0, g1;
                noise_pair(pair, (unsigned)m, a.key0, a.key1, g0, g1);
                s0[m].x += a.src_noise[m] * g0.x; s0[m].y += a.src_noise[m] * g0.y;
                s1[m].x += a.src_noise[m] * g1.x; s1[m].y += a.src_noise[m] * g1.y;
            }
        }
        const float sg = a.ant_sigma * 0.70710678118654752f;
        for (int k = 0; k < a.N; k++) {
            float2 x0 = make_float2(0.f, 0.f), x1 = x0;
            for (int m = 0; m < a.M; m++) {
                const float2 w = a.A[k * a.M + m];
                x0.x += w.x * s0[m].x - w.y * s0[m].y; x0.y += w.x * s0[m].y + w.y * s0[m].x;
                x1.x += w.x * s1[m].x - w.y * s1[m].y; x1.y += w.x * s1[m].y + w.y * s1[m].x;
            }
            if (sg != 0.0f) {
                float2 g0, g1;
                noise_pair(pair, (unsigned)(a.M + k), a.key0, a.key1, g0, g1);
                x0.x += sg * g0.x; x0.y += sg * g0.y;
                x1.x += sg * g1.x; x1.y += sg * g1.y;
            }
            if (2 * i + 1 < a.n) store_f4<true>(reinterpret_cast<float4 *>(a.out[k]) + i, make_float4(x0.x, x0.y, x1.x, x1.y));
            else a.out[k][2 * i] = x0;
        }
    }
}

}  // namespace doa

struct doa_sim_source {
    doa::SimArgs a;
    int device = 0;
    long long pos = 0;      // next sample index
    hipStream_t stream = nullptr;
    doa::DevBuf d_out;
};

extern "C" {

doa_sim_source_t *doa_sim_source_create(int num_ant_ele, int num_sources, float norm_spacing, const float *theta_deg,
                                        const double *tone_freq, const float *tone_ampl, const float *source_noise_ampl,
                                        float antenna_noise_sigma, unsigned long long seed)
{
    doa::clear_error();
    if (num_ant_ele <= 0 || num_ant_ele > DOA_MAX_ANT_ELE || num_sources <= 0 || num_sources > DOA_MAX_PEAKS ||
        !theta_deg || !tone_freq || !(norm_spacing > 0.0f) || !(antenna_noise_sigma >= 0.0f)) {
        doa::set_error("sim_source: bad arguments (num_ant_ele=%d num_sources=%d norm_spacing=%g)", num_ant_ele,
                       num_sources, (double)norm_spacing);
        return nullptr;
    }
    int dev = 0;
    if (doa::ensure_device(&dev) != DOA_OK) return nullptr;
    auto *h = new (std::nothrow) doa_sim_source();
    if (!h) { doa::set_error("out of memory"); return nullptr; }
    memset(&h->a, 0, sizeof h->a);
    h->device = dev;
    h->a.N = num_ant_ele; h->a.M = num_sources;
    h->a.ant_sigma = antenna_noise_sigma;
    h->a.key0 = (unsigned)(seed & 0xFFFFFFFFull); h->a.key1 = (unsigned)(seed >> 32);
    const double pi = 3.14159265358979323846;
    for (int m = 0; m < num_sources; m++) {
        h->a.freq[m] = tone_freq[m];
        h->a.ampl[m] = tone_ampl ? tone_ampl[m] : 1.0f;
        h->a.src_noise[m] = source_noise_ampl ? source_noise_ampl[m] : 0.0f;
        const double c = cos((double)theta_deg[m] * pi / 180.0);
        for (int k = 0; k < num_ant_ele; k++) {
            const double loc = (double)norm_spacing * ((num_ant_ele - 1) / 2.0 - k);
            const double ph = -2.0 * pi * c * loc;
            h->a.A[k * num_sources + m] = make_float2((float)cos(ph), (float)sin(ph));
        }
    }
    if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) {
        doa::set_error("sim_source: hipStreamCreate failed");
        delete h;
        return nullptr;
    }
    return h;
}

void doa_sim_source_destroy(doa_sim_source_t *h)
{
    if (!h) return;
    h->d_out.release();
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

int doa_sim_source_seek(doa_sim_source_t *h, long long sample_index)
{
    doa::clear_error();
    if (!h || sample_index < 0 || (sample_index & 1)) {
        doa::set_error("sim_source_seek: the position must be a non-negative even sample index");
        return DOA_ERR_INVALID_ARG;
    }
    h->pos = sample_index;
    return DOA_OK;
}

long long doa_sim_source_tell(const doa_sim_source_t *h) { return h ? h->pos : -1; }

static int sim_launch(doa_sim_source_t *h, int n, void *const *d_out, hipStream_t st)
{
    doa::SimArgs a = h->a;
    for (int k = 0; k < a.N; k++) {
        if (!d_out[k]) { doa::set_error("sim_source: output stream %d is NULL", k); return DOA_ERR_INVALID_ARG; }
        if (reinterpret_cast<uintptr_t>(d_out[k]) % 16) {
            doa::set_error("sim_source: output stream %d is not 16-byte aligned", k);
            return DOA_ERR_INVALID_ARG;
        }
        a.out[k] = static_cast<float2 *>(d_out[k]);
    }
    a.t0 = h->pos; a.n = n;
    const long long pairs = ((long long)n + 1) / 2;
    long long bx = (pairs + 255) / 256;
    if (bx > 4096) bx = 4096;
    hipLaunchKernelGGL(doa::sim_source_kernel, dim3((unsigned)bx), dim3(256), 0, st, a);
    DOA_HIP_TRY(hipGetLastError());
    return DOA_OK;
}

// Positions stay even between calls (a Philox block covers a sample pair): every call but the last of a
// run must ask for an even number of samples.
int doa_sim_source_work_dev(doa_sim_source_t *h, int noutput_items, void *const *d_output_items, void *hip_stream)
{
    doa::clear_error();
    if (!h || noutput_items < 0 || !d_output_items) { doa::set_error("sim_source_work_dev: bad arguments"); return DOA_ERR_INVALID_ARG; }
    if (h->pos & 1) { doa::set_error("sim_source: the previous call produced an odd number of samples; seek first"); return DOA_ERR_INVALID_ARG; }
    if (noutput_items == 0) return 0;
    if (int brc = doa::bind_device(h->device); brc != DOA_OK) return brc;
    int rc = sim_launch(h, noutput_items, d_output_items, static_cast<hipStream_t>(hip_stream));
    if (rc != DOA_OK) return rc;
    h->pos += noutput_items;
    return noutput_items;
}

int doa_sim_source_work(doa_sim_source_t *h, int noutput_items, void *const *output_items)
{
    doa::clear_error();
    if (!h || noutput_items < 0 || !output_items) { doa::set_error("sim_source_work: bad arguments"); return DOA_ERR_INVALID_ARG; }
    if (h->pos & 1) { doa::set_error("sim_source: the previous call produced an odd number of samples; seek first"); return DOA_ERR_INVALID_ARG; }
    if (noutput_items == 0) return 0;
    if (int brc = doa::bind_device(h->device); brc != DOA_OK) return brc;
    const size_t n = (size_t)noutput_items, n_al = (n + 1) & ~(size_t)1;
    int rc = h->d_out.reserve(n_al * h->a.N * sizeof(float2));
    if (rc != DOA_OK) return rc;
    void *d[DOA_MAX_ANT_ELE];
    for (int k = 0; k < h->a.N; k++) {
        if (!output_items[k]) { doa::set_error("sim_source_work: output stream %d is NULL", k); return DOA_ERR_INVALID_ARG; }
        d[k] = h->d_out.as<float2>() + k * n_al;
    }
    rc = sim_launch(h, noutput_items, d, h->stream);
    if (rc != DOA_OK) return rc;
    for (int k = 0; k < h->a.N; k++)
        DOA_HIP_TRY(hipMemcpyAsync(output_items[k], d[k], n * sizeof(float2), hipMemcpyDeviceToHost, h->stream));
    DOA_HIP_TRY(hipStreamSynchronize(h->stream));
    h->pos += noutput_items;
    return noutput_items;
}

}  // extern "C"
