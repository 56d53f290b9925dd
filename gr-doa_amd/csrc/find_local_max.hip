// find_local_max.hip — K5: top-M local maxima of a float vector and their x-axis locations.
//
// Replaces gr::doa::find_local_max (reference lib/find_local_max_impl.cc:47-194, _impl.h:53-58):
//   s = sign(diff(v)); zeros ("flats") take, right to left, the sign of their right neighbour
//   (>= 0 -> +1, else -1; a flat at the very end becomes +1) (:89-107); peaks are the i with
//   s[i-1] = +1, s[i] = -1 (:114; end points never qualify); peaks are ranked by value, descending
//   (:135-137); fewer than M peaks -> the remaining slots repeat an index (:145-163, including the
//   reference's use of the best peak's *position in the peak list* as that index); M == 1 is the
//   global arg-max (_impl.h:53-56).  Port 0 = v[pk] in rank order, port 1 = sort(x_axis[pk],
//   "descend") (:186-188) on the float-accumulated x axis (:60-69).
// Integer/compare logic only: results are bit-identical to the oracle for identical inputs (ties
// between equal peak values are ordered lowest index first; the reference leaves them unspecified).
//
// Kernel shape: one wave per vector, the vector in registers as CH float4 per lane (position
// p = 256 j + 4 lane + e, i.e. 1 KiB contiguous per load instruction).  Neighbour signs cross
// lanes by shuffles; flats (rare) are resolved by a ballot-based suffix scan in position order;
// the top-M are M rounds of a wave arg-max.  Longer vectors (up to 4096), lengths that are not a multiple of
// 4 and unaligned buffers take find_local_max_blocked_kernel (the vector staged once in a padded LDS row, lane k
// walks positions 64k..64k+63; peak_device.hpp: peak_pick_stream<true>); at most 64 elements of such a vector:
// find_local_max_stream_kernel (sign masks in SGPRs, nothing of the vector in registers); beyond 4096 elements a
// one-thread-per-vector fallback walks the reference's steps literally.
#include "kernels.hpp"
#include "peak_device.hpp"

#include <vector>

namespace doa {

int PeakTables::build(int num_max_vals, int vector_len, float x_min_, float x_max_)
{
    M = num_max_vals; L = vector_len; x_min = x_min_; x_max = x_max_;
    std::vector<float> x(L);
    x[0] = x_min;
    float x_prev = x_min;
    const float x_range = x_max - x_min;
    for (int ii = 1; ii < L; ii++) {           // lib/find_local_max_impl.cc:60-69, all in float
        float v = x_prev + x_range / L;
        x_prev = v;
        x[ii] = v;
    }
    int rc = d_x.reserve(sizeof(float) * (size_t)L);
    if (rc != DOA_OK) return rc;
    DOA_HIP_TRY(hipMemcpy(d_x.p, x.data(), sizeof(float) * (size_t)L, hipMemcpyHostToDevice));
    return DOA_OK;
}

template <int CH>
__global__ __launch_bounds__(256) void find_local_max_kernel(const float *__restrict__ in, const float *__restrict__ xaxis,
                                                             float *__restrict__ out_val, float *__restrict__ out_loc,
                                                             int L, int M, int n_items)
{
    const int lane = threadIdx.x & (kWave - 1);
    const int item = __builtin_amdgcn_readfirstlane(blockIdx.x * (blockDim.x / kWave) + (threadIdx.x / kWave));
    if (item >= n_items) return;
    const float *v_in = in + (size_t)item * L;

    float v[CH][4];
#pragma unroll
    for (int j = 0; j < CH; j++) {
        const int p0 = 256 * j + 4 * lane;
        if (p0 < L) {   // L % 4 == 0: a float4 is entirely inside or outside
            const float4 t = load_f4<true>(reinterpret_cast<const float4 *>(v_in + p0));
            v[j][0] = t.x; v[j][1] = t.y; v[j][2] = t.z; v[j][3] = t.w;
        } else {
            v[j][0] = v[j][1] = v[j][2] = v[j][3] = 0.f;
        }
    }
    peak_pick<CH>(v, lane, L, M, xaxis, out_val + (size_t)item * M, out_loc + (size_t)item * M);
}

// Long or oddly sized vectors (any 1 <= L <= 4096, any alignment): peak_pick_stream (peak_device.hpp) straight
// out of global memory.
__global__ __launch_bounds__(256) void find_local_max_stream_kernel(const float *__restrict__ in, const float *__restrict__ xaxis,
                                                                    float *__restrict__ out_val, float *__restrict__ out_loc,
                                                                    int L, int M, int n_items)
{
    const int lane = threadIdx.x & (kWave - 1);
    const int item = __builtin_amdgcn_readfirstlane(blockIdx.x * (blockDim.x / kWave) + (threadIdx.x / kWave));
    if (item >= n_items) return;
    const float *v_in = in + (size_t)item * L;
    peak_pick_stream([&](int p) { return v_in[p]; }, L, M, xaxis, out_val + (size_t)item * M, out_loc + (size_t)item * M, lane);
}

// The same through LDS (any 64 < L <= 4096, any alignment): the vector is fetched once with coalesced loads into a row
// padded by one word per 64 (position p at word p + (p >> 6)), and the peak pick runs in its lane-blocked form: lane k walks
// positions 64k .. 64k+63 of the row, conflict-free.  Per 4096 vectors of 4096 values: 16.8-17.7 us against the 23.0-24.3 of the
// streaming mask kernel on MUSIC spectra, 29 against 125 on vectors with a peak every few positions.
// BS = positions per lane: 64 for vectors of up to 4096 values, 16 up to 1024 (all lanes busy).
template <int BS>
__global__ __launch_bounds__(256) void find_local_max_blocked_kernel(const float *__restrict__ in, const float *__restrict__ xaxis,
                                                                     float *__restrict__ out_val, float *__restrict__ out_loc,
                                                                     int L, int M, int n_items, int vec4)
{
    constexpr int LMAX = 64 * BS;
    constexpr int LOG_BS = (BS == 16) ? 4 : 6;
    static_assert(BS == 16 || BS == 64, "block size");
    __shared__ float rows[4][LMAX + 64 + 4];
    const int lane = threadIdx.x & (kWave - 1);
    const int wib = threadIdx.x / kWave;
    const int wave = __builtin_amdgcn_readfirstlane(blockIdx.x * (blockDim.x / kWave) + wib);
    const int n_waves = gridDim.x * (blockDim.x / kWave);
    float *lrow = rows[wib];
    struct PaddedRow {
        const float *r, *mine;                             // mine = this lane's block
        __device__ __forceinline__ float operator()(int p) const { return r[p + (p >> LOG_BS)]; }
        __device__ __forceinline__ float blk(int i) const { return mine[i + (i >> LOG_BS)]; }
    };
    if (vec4) {
        // L % 4 == 0 and 16-byte aligned rows: 1 KiB per load instruction, and the NEXT vector of this wave is fetched into
        // registers while the peak pick of the current one runs (two LDS-limited waves per SIMD hide no HBM latency)
        const int n4 = L >> 2;
        float4 pre[LMAX / 256];
        auto fetch = [&](int it) {
            const float4 *src = reinterpret_cast<const float4 *>(in + (size_t)it * L);
#pragma unroll
            for (int j = 0; j < LMAX / 256; j++) {
                const int q = lane + kWave * j;
                if (q < n4) pre[j] = load_f4<true>(src + q);
            }
        };
        if (wave < n_items) fetch(wave);
        for (int item = wave; item < n_items; item += n_waves) {
#pragma unroll
            for (int j = 0; j < LMAX / 256; j++) {
                const int q = lane + kWave * j;
                if (q < n4) {
                    float *d = lrow + 4 * q + (q >> (LOG_BS - 2));    // 4 consecutive positions never straddle a block
                    d[0] = pre[j].x; d[1] = pre[j].y; d[2] = pre[j].z; d[3] = pre[j].w;
                }
            }
            // written position by position, read block by block (same wave: LDS operations of one wave complete in order,
            // the fences only keep the compiler from moving them)
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            if (item + n_waves < n_items) fetch(item + n_waves);
            peak_pick_stream<true, BS>(PaddedRow{lrow, lrow + (BS + 1) * lane}, L, M, xaxis, out_val + (size_t)item * M, out_loc + (size_t)item * M, lane);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        }
        return;
    }
    for (int item = wave; item < n_items; item += n_waves) {
        const float *v_in = in + (size_t)item * L;
#pragma unroll 4
        for (int p = lane; p < L; p += kWave) lrow[p + (p >> LOG_BS)] = __builtin_nontemporal_load(v_in + p);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        peak_pick_stream<true, BS>(PaddedRow{lrow, lrow + (BS + 1) * lane}, L, M, xaxis, out_val + (size_t)item * M, out_loc + (size_t)item * M, lane);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
}

// Literal walk of the reference's steps, one thread per vector (any L >= 1).  `scratch` holds L
// signed bytes per item.
__global__ void find_local_max_serial_kernel(const float *__restrict__ in, const float *__restrict__ xaxis,
                                             float *__restrict__ out_val, float *__restrict__ out_loc,
                                             signed char *__restrict__ scratch, int L, int M, int n_items)
{
    const int item = blockIdx.x * blockDim.x + threadIdx.x;
    if (item >= n_items) return;
    const float *v = in + (size_t)item * L;
    signed char *s = scratch + (size_t)item * L;
    int pkidx[DOA_MAX_PEAKS];
    auto argmax_first = [&]() {   // arma index_max semantics, see the wave kernel
        int k = 0;
        float best = -INFINITY;
        for (int i = 0; i < L; i++) if (v[i] > best) { best = v[i]; k = i; }
        return k;
    };
    if (M == 1 || L < 3) {
        const int k = argmax_first();
        for (int j = 0; j < M; j++) pkidx[j] = k;
    } else {
        const int Ls = L - 1;
        for (int i = 0; i < Ls; i++) { const float d = v[i + 1] - v[i]; s[i] = (d > 0.f) ? 1 : ((d < 0.f) ? -1 : 0); }
        for (int i = Ls - 1; i >= 0; i--)
            if (s[i] == 0) { const int nx = (i + 1 < Ls - 1) ? i + 1 : Ls - 1; s[i] = (s[nx] >= 0) ? 1 : -1; }
        int n_valid = 0;
        for (int i = 1; i < Ls; i++) if (s[i - 1] == 1 && s[i] == -1) n_valid++;
        const int rounds = n_valid < M ? n_valid : M;
        int best_list_pos = 0;
        for (int r = 0; r < rounds; r++) {
            int bi = -1;
            for (int i = 1; i < Ls; i++)
                if (s[i - 1] == 1 && s[i] == -1) {
                    bool used = false;
                    for (int u = 0; u < r; u++) used |= (pkidx[u] == i);
                    if (!used && (bi < 0 || v[i] > v[bi])) bi = i;
                }
            pkidx[r] = bi;
            if (r == 0)
                for (int i = 1; i < bi; i++) if (s[i - 1] == 1 && s[i] == -1) best_list_pos++;
        }
        const int fill = (n_valid == 0) ? argmax_first() : best_list_pos;
        for (int j = rounds; j < M; j++) pkidx[j] = fill;
    }
    float loc[DOA_MAX_PEAKS];
    for (int j = 0; j < M; j++) { out_val[(size_t)item * M + j] = v[pkidx[j]]; loc[j] = xaxis[pkidx[j]]; }
    for (int a = 1; a < M; a++) {
        const float t = loc[a];
        int b = a - 1;
        while (b >= 0 && loc[b] < t) { loc[b + 1] = loc[b]; b--; }
        loc[b + 1] = t;
    }
    for (int j = 0; j < M; j++) out_loc[(size_t)item * M + j] = loc[j];
}



int launch_find_local_max(const PeakTables &t, int n_items, const void *d_in, void *d_max, void *d_argmax,
                          hipStream_t st)
{
    if (n_items <= 0) return DOA_OK;
    const int L = t.L, M = t.M;
    const float *in = (const float *)d_in;
    const float *x = t.d_x.as<float>();
    float *ov = (float *)d_max, *ol = (float *)d_argmax;
    if (L < 1 || L > 4096) {
        set_error("find_local_max: vector_len=%d needs the serial path (call through a handle)", L);
        return DOA_ERR_UNSUPPORTED;
    }
    dim3 block(256), grid((n_items + 3) / 4);
    // short, 16-byte aligned vectors of a multiple-of-4 length: the vector lives in registers (CH float4 per
    // lane); everything else up to 4096 elements: the LDS-staged blocked kernel (46.5 -> 23.7 -> 17 us per 4096 vectors of
    // 4096: register-resident CH = 16 kernel, streaming mask kernel, this one)
    const bool reg_ok = (L % 4 == 0) && L >= 4 && (reinterpret_cast<uintptr_t>(d_in) % 16 == 0);
    const bool use_reg = reg_ok && L <= 1024;
    const int k5_lab = DOA_LAB_ENV_INT("DOA_K5_STREAM", 0);      // lab: 1 = streaming mask kernel, 2 = register kernel for M > 1 too
    if (L > 1024 && k5_lab != 1) {
        // 65 KiB of LDS per 4-wave workgroup: two per CU
        int bb = (n_items + 3) / 4;
        if (bb > cu_count() * 2) bb = cu_count() * 2;
        hipLaunchKernelGGL(find_local_max_blocked_kernel<64>, dim3(bb), block, 0, st, in, x, ov, ol, L, M, n_items, reg_ok ? 1 : 0);
    }
    else if (L > 64 && (!use_reg || (M > 1 && k5_lab != 2)) && k5_lab != 1) {
        // up to 1024 values: 16 positions per lane; also for aligned vectors when more than one peak is wanted (the
        // register kernel's general peak pick costs more than the trip through LDS)
        int bb = (n_items + 3) / 4;
        if (bb > cu_count() * 4) bb = cu_count() * 4;
        hipLaunchKernelGGL(find_local_max_blocked_kernel<16>, dim3(bb), block, 0, st, in, x, ov, ol, L, M, n_items, reg_ok ? 1 : 0);
    }
    else if (!use_reg)  hipLaunchKernelGGL(find_local_max_stream_kernel, grid, block, 0, st, in, x, ov, ol, L, M, n_items);
    else if (L <= 256)  hipLaunchKernelGGL(find_local_max_kernel<1>, grid, block, 0, st, in, x, ov, ol, L, M, n_items);
    else if (L <= 512)  hipLaunchKernelGGL(find_local_max_kernel<2>, grid, block, 0, st, in, x, ov, ol, L, M, n_items);
    else                hipLaunchKernelGGL(find_local_max_kernel<4>, grid, block, 0, st, in, x, ov, ol, L, M, n_items);
    DOA_HIP_TRY(hipGetLastError());
    return DOA_OK;
}

int launch_find_local_max_serial(const PeakTables &t, int n_items, const void *d_in, void *d_max, void *d_argmax,
                                 void *d_scratch, hipStream_t st)
{
    if (n_items <= 0) return DOA_OK;
    hipLaunchKernelGGL(find_local_max_serial_kernel, dim3((n_items + 63) / 64), dim3(64), 0, st, (const float *)d_in,
                       t.d_x.as<float>(), (float *)d_max, (float *)d_argmax, (signed char *)d_scratch, t.L, t.M,
                       n_items);
    DOA_HIP_TRY(hipGetLastError());
    return DOA_OK;
}

bool find_local_max_fast_ok(int L, const void *d_in)
{
    (void)d_in;                                          // the streaming kernel takes any alignment
    return L >= 1 && L <= 4096;
}

}  // namespace doa

// ---------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------
struct doa_find_local_max {
    doa::PeakTables tab;
    int device = 0;
    hipStream_t stream = nullptr;
    doa::DevBuf d_in, d_out0, d_out1, d_scratch;
};

extern "C" {

doa_find_local_max_t *doa_find_local_max_create(int num_max_vals, int vector_len, float x_min, float x_max)
{
    doa::clear_error();
    // grc/doa_find_local_max.xml:33-35
    if (num_max_vals <= 0 || vector_len <= 0 || !(x_max > x_min)) {
        doa::set_error("find_local_max: need num_max_vals > 0, vector_len > 0, x_max > x_min (got %d, %d, %g, %g)",
                       num_max_vals, vector_len, (double)x_min, (double)x_max);
        return nullptr;
    }
    if (num_max_vals > DOA_MAX_PEAKS) {
        doa::set_error("find_local_max: num_max_vals=%d exceeds DOA_MAX_PEAKS=%d", num_max_vals, DOA_MAX_PEAKS);
        return nullptr;
    }
    int dev = 0;
    if (doa::ensure_device(&dev) != DOA_OK) return nullptr;
    auto *h = new (std::nothrow) doa_find_local_max();
    if (!h) { doa::set_error("out of memory"); return nullptr; }
    h->device = dev;
    if (h->tab.build(num_max_vals, vector_len, x_min, x_max) != DOA_OK ||
        hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) {
        if (!*doa_last_error()) doa::set_error("find_local_max: device setup failed");
        doa_find_local_max_destroy(h);
        return nullptr;
    }
    return h;
}

void doa_find_local_max_destroy(doa_find_local_max_t *h)
{
    if (!h) return;
    h->tab.release();
    h->d_in.release(); h->d_out0.release(); h->d_out1.release(); h->d_scratch.release();
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

int doa_find_local_max_work_dev(doa_find_local_max_t *h, int noutput_items, const void *d_input_items0,
                                void *d_output_items0, void *d_output_items1, void *hip_stream)
{
    doa::clear_error();
    if (!h || noutput_items < 0 || (noutput_items > 0 && (!d_input_items0 || !d_output_items0 || !d_output_items1))) {
        doa::set_error("find_local_max_work_dev: bad arguments");
        return DOA_ERR_INVALID_ARG;
    }
    if (noutput_items == 0) return 0;
    if (int brc = doa::bind_device(h->device); brc != DOA_OK) return brc;
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    int rc;
    if (doa::find_local_max_fast_ok(h->tab.L, d_input_items0)) {
        rc = doa::launch_find_local_max(h->tab, noutput_items, d_input_items0, d_output_items0, d_output_items1, st);
    } else {
        rc = h->d_scratch.reserve((size_t)noutput_items * h->tab.L);
        if (rc == DOA_OK)
            rc = doa::launch_find_local_max_serial(h->tab, noutput_items, d_input_items0, d_output_items0,
                                                   d_output_items1, h->d_scratch.p, st);
    }
    return rc == DOA_OK ? noutput_items : rc;
}

int doa_find_local_max_work(doa_find_local_max_t *h, int noutput_items, const void *input_items0, void *output_items0,
                            void *output_items1)
{
    doa::clear_error();
    if (!h || noutput_items < 0 || (noutput_items > 0 && (!input_items0 || !output_items0 || !output_items1))) {
        doa::set_error("find_local_max_work: bad arguments");
        return DOA_ERR_INVALID_ARG;
    }
    if (noutput_items == 0) return 0;
    if (int brc = doa::bind_device(h->device); brc != DOA_OK) return brc;
    const size_t in_bytes = (size_t)noutput_items * h->tab.L * sizeof(float);
    const size_t out_bytes = (size_t)noutput_items * h->tab.M * sizeof(float);
    int rc = h->d_in.reserve(in_bytes);
    if (rc == DOA_OK) rc = h->d_out0.reserve(out_bytes);
    if (rc == DOA_OK) rc = h->d_out1.reserve(out_bytes);
    if (rc != DOA_OK) return rc;
    DOA_HIP_TRY(hipMemcpyAsync(h->d_in.p, input_items0, in_bytes, hipMemcpyHostToDevice, h->stream));
    rc = doa_find_local_max_work_dev(h, noutput_items, h->d_in.p, h->d_out0.p, h->d_out1.p, h->stream);
    if (rc < 0) return rc;
    DOA_HIP_TRY(hipMemcpyAsync(output_items0, h->d_out0.p, out_bytes, hipMemcpyDeviceToHost, h->stream));
    DOA_HIP_TRY(hipMemcpyAsync(output_items1, h->d_out1.p, out_bytes, hipMemcpyDeviceToHost, h->stream));
    DOA_HIP_TRY(hipStreamSynchronize(h->stream));
    return noutput_items;
}

}  // extern "C"
