// valu_rates.hip -- lab: issue cost of the vector instructions the scan / Jacobi kernels are made of, per SIMD, at
// 1..4 resident waves per SIMD, plus the in-kernel clock (s_memtime / s_memrealtime).  Not product code.
//   hipcc -O3 --offload-arch=gfx950 tools/lab/valu_rates.hip -o gpurun_out/valu_rates && gpurun_out/valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int ITERS = 2048;

// 8 independent chains, 2 rounds per iteration = 16 instructions per iteration
#define BODY8(INS)  INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7)

template <int OP> __global__ __launch_bounds__(256) void rate_kernel(unsigned long long *stamps, float *sink, float seed)
{
    double d[8]; float f[8]; float g[8];
    typedef float v2f __attribute__((ext_vector_type(2)));
    v2f p[8];
    for (int i = 0; i < 8; i++) { d[i] = seed + i + threadIdx.x * 1e-3; f[i] = seed + i; g[i] = seed * 0.5f + i; p[i] = v2f{f[i], g[i]}; }
    double dm = 1.0000001, da = 1e-9;
    float fm = 1.0000001f, fa = 1e-9f;
    v2f pm = {fm, fm}, pa = {fa, fa};
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
#pragma unroll 1
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int r = 0; r < 2; r++) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if constexpr (OP == 0) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[i]) : "v"(dm), "v"(da));
                if constexpr (OP == 1) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[i]) : "v"(fm), "v"(fa));
                if constexpr (OP == 2) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pm), "v"(pa));
                if constexpr (OP == 3) asm volatile("v_log_f32 %0, %0" : "+v"(f[i]));
                if constexpr (OP == 4) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f[i]) : "v"(d[i]));
                if constexpr (OP == 5) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(f[i]) : "v"(g[i]), "v"(fm));
                if constexpr (OP == 6) asm volatile("v_cmp_le_f32 vcc, %1, %2\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(f[i]) : "v"(g[i]), "v"(fm) : "vcc");
                if constexpr (OP == 7) asm volatile("v_mov_b32_dpp %0, %1 row_ror:4 row_mask:0xf bank_mask:0xf" : "+v"(f[i]) : "v"(g[i]));
                if constexpr (OP == 8) asm volatile("v_rcp_f32 %0, %0" : "+v"(f[i]));
                if constexpr (OP == 9) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"(dm));
                if constexpr (OP == 10) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(da));
                if constexpr (OP == 11) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pm));
                if constexpr (OP == 12) asm volatile("v_min_f32 %0, %0, %1" : "+v"(f[i]) : "v"(g[i]));
                if constexpr (OP == 13) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[i]) : "v"(f[i]));
                if constexpr (OP == 14) asm volatile("v_rsq_f64 %0, %0" : "+v"(d[i]));
                if constexpr (OP == 15) asm volatile("v_rcp_f64 %0, %0" : "+v"(d[i]));
                if constexpr (OP == 16) asm volatile("v_sqrt_f32 %0, %0" : "+v"(f[i]));
                if constexpr (OP == 17) asm volatile("v_rsq_f32 %0, %0" : "+v"(f[i]));
                // mixed: one fma_f64 + one fma_f32 (do the two rates add or overlap?)
                if constexpr (OP == 18) asm volatile("v_fma_f64 %0, %0, %2, %3\n\tv_fma_f32 %1, %1, %4, %5" : "+v"(d[i]), "+v"(f[i]) : "v"(dm), "v"(da), "v"(fm), "v"(fa));
                if constexpr (OP == 19) asm volatile("ds_bpermute_b32 %0, %1, %0\n\ts_waitcnt lgkmcnt(0)" : "+v"(f[i]) : "v"(g[i]));
                if constexpr (OP == 20) asm volatile("s_nop 0");
                if constexpr (OP == 21) asm volatile("s_add_u32 s20, s20, 1" ::: "s20");
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    float acc = 0;
    for (int i = 0; i < 8; i++) acc += (float)d[i] + f[i] + g[i] + p[i].x + p[i].y;
    if (acc == 12345.678f) sink[0] = acc;
    if ((threadIdx.x & 63) == 0) {
        const int w = blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
        stamps[2 * w] = t1 - t0; stamps[2 * w + 1] = r1 - r0;
    }
}

template <int OP> void run(const char *name, int cus, unsigned long long *d_st, float *d_sink, int instr_per_slot = 1)
{
    for (int wps = 1; wps <= 4; wps++) {
        const int blocks = cus * wps;     // 256-thread blocks: one wave per SIMD each
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        hipLaunchKernelGGL(rate_kernel<OP>, dim3(blocks), dim3(256), 0, 0, d_st, d_sink, 1.0f);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        const int reps = 5;
        for (int r = 0; r < reps; r++) hipLaunchKernelGGL(rate_kernel<OP>, dim3(blocks), dim3(256), 0, 0, d_st, d_sink, 1.0f);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<unsigned long long> st(2 * blocks * 4);
        CK(hipMemcpy(st.data(), d_st, st.size() * 8, hipMemcpyDeviceToHost));
        // median over waves of cycles and of clock
        std::vector<double> cyc, clk;
        for (int w = 0; w < blocks * 4; w++) { cyc.push_back((double)st[2 * w]); clk.push_back(st[2 * w] / (st[2 * w + 1] * 10e-9) * 1e-9); }
        std::sort(cyc.begin(), cyc.end()); std::sort(clk.begin(), clk.end());
        const double c = cyc[cyc.size() / 2], ghz = clk[clk.size() / 2];
        const double n_ins = (double)ITERS * 16 * instr_per_slot;
        printf("%-22s waves/SIMD %d: wave cycles/instr %6.2f  -> SIMD cycles/instr %5.2f   clock %.2f GHz   (launch %.1f us)\n", name, wps,
               c / n_ins, c / n_ins / wps, ghz, ms * 1e3 / reps);
    }
}

int main()
{
    hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
    const int cus = pr.multiProcessorCount;
    printf("device %s, %d CUs\n", pr.name, cus);
    unsigned long long *d_st; float *d_sink;
    CK(hipMalloc(&d_st, 2 * 8 * cus * 4 * 4 * 2)); CK(hipMalloc(&d_sink, 64));
    run<0>("v_fma_f64", cus, d_st, d_sink);
    run<9>("v_mul_f64", cus, d_st, d_sink);
    run<10>("v_add_f64", cus, d_st, d_sink);
    run<1>("v_fma_f32", cus, d_st, d_sink);
    run<2>("v_pk_fma_f32", cus, d_st, d_sink);
    run<11>("v_pk_mul_f32", cus, d_st, d_sink);
    run<3>("v_log_f32", cus, d_st, d_sink);
    run<8>("v_rcp_f32", cus, d_st, d_sink);
    run<16>("v_sqrt_f32", cus, d_st, d_sink);
    run<17>("v_rsq_f32", cus, d_st, d_sink);
    run<14>("v_rsq_f64", cus, d_st, d_sink);
    run<15>("v_rcp_f64", cus, d_st, d_sink);
    run<4>("v_cvt_f32_f64", cus, d_st, d_sink);
    run<13>("v_cvt_f64_f32", cus, d_st, d_sink);
    run<5>("v_min3_f32", cus, d_st, d_sink);
    run<12>("v_min_f32", cus, d_st, d_sink);
    run<6>("v_cmp+v_cndmask (2)", cus, d_st, d_sink, 2);
    run<7>("v_mov_b32_dpp", cus, d_st, d_sink);
    run<18>("fma_f64+fma_f32 (2)", cus, d_st, d_sink, 2);
    run<19>("ds_bpermute+wait", cus, d_st, d_sink);
    run<20>("s_nop 0", cus, d_st, d_sink);
    run<21>("s_add_u32", cus, d_st, d_sink);
    return 0;
}
