// Issue-cost microbenchmark for gfx950: cycles per wave64 instruction of the kinds the block-sum scan uses,
// for 1 and 2 waves per SIMD (dependent chain per wave).  hipcc --offload-arch=gfx950 -O3 -o microbench_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("hip error %s\n", hipGetErrorString(e_)); return 1; } } while (0)
constexpr int N = 1 << 17;
template <int K> __global__ void k(double *out, double seed)
{
    double a = seed + threadIdx.x, b = seed * 0.5, c = 1.0000001;
    float f = static_cast<float>(a), g = 1.0001f, f2 = f + 1, f3 = f + 2, f4 = f + 3;
    double a2 = a + 1, a3 = a + 2, a4 = a + 3;
    int i = threadIdx.x + 1;
    long long t0 = clock64();
#pragma unroll 16
    for (int it = 0; it < N; ++it) {
        if (K == 0) a = fma(a, c, b);
        if (K == 1) a = a + b;
        if (K == 2) a = a * c;
        if (K == 3) { a = static_cast<double>(i); i += static_cast<int>(a) & 1; }
        if (K == 4) { f = static_cast<float>(a); a += f; }
        if (K == 5) f = __builtin_amdgcn_logf(f) + 3.0f;
        if (K == 6) f = __builtin_amdgcn_rcpf(f) + 1.0f;
        if (K == 7) f = fmaf(f, g, 0.5f);
        if (K == 8) i = __builtin_amdgcn_update_dpp(i, i, 0x138, 0xf, 0xf, false) + 1;
        if (K == 9) { long long p = static_cast<long long>(i) * i; i = static_cast<int>(p >> 7) | 1; }
        if (K == 10) { f = static_cast<float>(i); i += static_cast<int>(f) & 3; }
        if (K == 11) { f = fmaf(f, g, 0.5f); f2 = fmaf(f2, g, 0.25f); f3 = fmaf(f3, g, 0.125f); f4 = fmaf(f4, g, 0.75f); }
        if (K == 12) { a = fma(a, c, b); a2 = fma(a2, c, b); a3 = fma(a3, c, b); a4 = fma(a4, c, b); }
        if (K == 13) { f = __builtin_amdgcn_logf(f) + 3.0f; f2 = __builtin_amdgcn_logf(f2) + 3.0f; f3 = __builtin_amdgcn_logf(f3) + 3.0f; f4 = __builtin_amdgcn_logf(f4) + 3.0f; }
    }
    long long t1 = clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + f + i + f2 + f3 + f4 + a2 + a3 + a4;
    if (threadIdx.x == 0) out[gridDim.x * blockDim.x + blockIdx.x] = static_cast<double>(t1 - t0);
}
template <int K> int run(const char *name, int instr_per_iter)
{
    double *d; const int G = 256 * 16;             // up to 8 waves per CU
    CHK(hipMalloc(&d, sizeof(double) * (G * 64 + G)));
    for (int waves_per_cu : {4, 8, 16}) {
        int grid = 256 * waves_per_cu;
        hipLaunchKernelGGL(k<K>, dim3(grid), dim3(64), 0, 0, d, 1.5);
        CHK(hipDeviceSynchronize());
        hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
        CHK(hipEventRecord(e0)); hipLaunchKernelGGL(k<K>, dim3(grid), dim3(64), 0, 0, d, 1.5); CHK(hipEventRecord(e1));
        CHK(hipDeviceSynchronize());
        float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
        double clk; CHK(hipMemcpy(&clk, d + grid * 64, sizeof(double), hipMemcpyDeviceToHost));
        printf("%-28s waves/CU %d: %.1f ns per iteration per wave (%d instr): kernel %.3f ms, s_memtime delta %.0f\n", name, waves_per_cu,
               ms * 1e6 / N, instr_per_iter, ms, clk);
    }
    CHK(hipFree(d));
    return 0;
}
int main()
{
    run<0>("v_fma_f64", 1); run<1>("v_add_f64", 1); run<2>("v_mul_f64", 1); run<3>("cvt_f64_i32 + and + add", 4);
    run<4>("cvt_f32_f64 + cvt + add_f64", 3); run<5>("v_log_f32 + add", 2); run<6>("v_rcp_f32 + add", 2); run<7>("v_fma_f32", 1);
    run<8>("dpp wave_shr + add", 2); run<9>("mul i64 + shift + or", 4); run<10>("cvt_f32_i32 + cvt + and + add", 4);
    run<11>("4 indep v_fma_f32", 4); run<12>("4 indep v_fma_f64", 4); run<13>("4 indep (log + add)", 8);
    return 0;
}
