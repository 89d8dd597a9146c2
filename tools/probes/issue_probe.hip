// Issue-rate probe for one SIMD of gfx950: how many cycles does an instruction cost when 1 / 2 / 4 waves share a SIMD,
// and do scalar and vector instructions of DIFFERENT waves issue side by side?  (diagnostic, not part of the library)
//   hipcc --offload-arch=gfx950 -O3 tools/probes/issue_probe.hip -o /tmp/issue_probe && /tmp/issue_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)

// 64 vector instructions per iteration, four independent chains (no dependent-issue stalls)
__global__ void k_valu(float *out, int iters)
{
    float a = threadIdx.x, b = a + 1, c = a + 2, d = a + 3;
    for (int i = 0; i < iters; ++i)
        asm volatile(REP16("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3\n")
                     : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : : "scc");
    out[blockIdx.x * 64 + threadIdx.x] = a + b + c + d;
}
// one dependent chain: the latency of a lone wave's back-to-back dependent instructions
__global__ void k_valu_dep(float *out, int iters)
{
    float a = threadIdx.x;
    for (int i = 0; i < iters; ++i) asm volatile(REP64("v_fma_f32 %0, %0, %0, %0\n") : "+v"(a) : : "scc");
    out[blockIdx.x * 64 + threadIdx.x] = a;
}
__global__ void k_salu(float *out, int iters)
{
    int a = blockIdx.x, b = a + 1, c = a + 2, d = a + 3;
    for (int i = 0; i < iters; ++i)
        asm volatile(REP16("s_add_u32 %0, %0, %0\n s_add_u32 %1, %1, %1\n s_add_u32 %2, %2, %2\n s_add_u32 %3, %3, %3\n")
                     : "+s"(a), "+s"(b), "+s"(c), "+s"(d) : : "scc");
    out[blockIdx.x * 64 + threadIdx.x] = a + b + c + d;
}
// 64 vector + 64 scalar per iteration, alternating
__global__ void k_mix(float *out, int iters)
{
    float a = threadIdx.x, b = a + 1;
    int s = blockIdx.x, t = s + 1;
    for (int i = 0; i < iters; ++i)
        asm volatile(REP16("v_fma_f32 %0, %0, %0, %0\n s_add_u32 %2, %2, %2\n v_fma_f32 %1, %1, %1, %1\n s_add_u32 %3, %3, %3\n"
                           "v_fma_f32 %0, %0, %0, %0\n s_add_u32 %2, %2, %2\n v_fma_f32 %1, %1, %1, %1\n s_add_u32 %3, %3, %3\n")
                     : "+v"(a), "+v"(b), "+s"(s), "+s"(t) : : "scc");
    out[blockIdx.x * 64 + threadIdx.x] = a + b + s + t;
}
// like k_mix, but the scalar work is branches: 64 vector instructions and 16 (not taken) branches
__global__ void k_branch(float *out, int iters)
{
    float a = threadIdx.x, b = a + 1, c = a + 2, d = a + 3;
    for (int i = 0; i < iters; ++i)
        asm volatile(REP16("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3\n"
                           "s_cmp_eq_u32 0, 1\n s_cbranch_scc1 0\n")
                     : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : : "scc");
    out[blockIdx.x * 64 + threadIdx.x] = a + b + c + d;
}
// transcendental and fp64 mixes: 64 instructions per iteration
__global__ void k_trans(float *out, int iters)
{
    float a = threadIdx.x + 1.0f, b = a + 1, c = a + 2, d = a + 3;
    for (int i = 0; i < iters; ++i)
        asm volatile(REP16("v_log_f32 %0, %0\n v_rcp_f32 %1, %1\n v_log_f32 %2, %2\n v_rcp_f32 %3, %3\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : : "scc");
    out[blockIdx.x * 64 + threadIdx.x] = a + b + c + d;
}
__global__ void k_f64(float *out, int iters)
{
    double a = threadIdx.x + 1.0, b = a + 1, c = a + 2, d = a + 3;
    for (int i = 0; i < iters; ++i)
        asm volatile(REP16("v_fma_f64 %0, %0, %0, %0\n v_fma_f64 %1, %1, %1, %1\n v_fma_f64 %2, %2, %2, %2\n v_fma_f64 %3, %3, %3, %3\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : : "scc");
    out[blockIdx.x * 64 + threadIdx.x] = static_cast<float>(a + b + c + d);
}
__global__ void k_cvt(float *out, int iters)
{
    double a = threadIdx.x + 1.0, b = a + 1; float c = 1, d = 2;
    for (int i = 0; i < iters; ++i)
        asm volatile(REP16("v_cvt_f32_f64 %2, %0\n v_cvt_f32_f64 %3, %1\n v_cvt_f64_f32 %0, %2\n v_cvt_f64_f32 %1, %3\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : : "scc");
    out[blockIdx.x * 64 + threadIdx.x] = static_cast<float>(a + b) + c + d;
}
__global__ void k_dpp(float *out, int iters)
{
    int a = threadIdx.x, b = a + 1, c = a + 2, d = a + 3;
    for (int i = 0; i < iters; ++i)
        asm volatile(REP16("v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %1, %1, %1 row_shr:2 row_mask:0xf bank_mask:0xf\n"
                           "v_mov_b32_dpp %2, %2 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %3 row_bcast:15 row_mask:0xa bank_mask:0xf\n")
                     : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : : "scc");
    out[blockIdx.x * 64 + threadIdx.x] = a + b + c + d;
}
__global__ void k_readlane(float *out, int iters)
{
    int a = threadIdx.x, b = a + 1; int s = 0, t = 0;
    for (int i = 0; i < iters; ++i)
        asm volatile(REP16("v_readlane_b32 %2, %0, 3\n v_readlane_b32 %3, %1, 5\n v_add_u32 %0, %2, %0\n v_add_u32 %1, %3, %1\n") : "+v"(a), "+v"(b), "+s"(s), "+s"(t) : : "scc");
    out[blockIdx.x * 64 + threadIdx.x] = a + b + s + t;
}

template <typename K> void run(const char *name, K k, int per_iter, float *out, int n_cu, double mhz)
{
    const int iters = 20000;
    for (int w : {1, 2, 4, 8}) {
        const int grid = n_cu * 4 * w;
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k, dim3(grid), dim3(64), 0, 0, out, 200);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(grid), dim3(64), 0, 0, out, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        const double cyc = ms * 1e-3 * mhz * 1e6;
        printf("%-12s %d waves/SIMD: %7.3f ms  %6.2f cycles per instruction per wave, %6.2f per SIMD (at %.0f MHz)\n", name, w, ms,
               cyc / (double(iters) * per_iter), cyc / (double(iters) * per_iter * w), mhz);
    }
}
int main()
{
    setvbuf(stdout, nullptr, _IONBF, 0);
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int n_cu = p.multiProcessorCount; const double mhz = p.clockRate / 1000.0;
    printf("%s: %d CUs, %.0f MHz\n", p.name, n_cu, mhz);
    float *out; hipMalloc(&out, size_t(n_cu) * 4 * 8 * 64 * sizeof(float));
    run("valu x4", k_valu, 64, out, n_cu, mhz);
    run("valu dep", k_valu_dep, 64, out, n_cu, mhz);
    run("salu", k_salu, 64, out, n_cu, mhz);
    run("valu+salu", k_mix, 128, out, n_cu, mhz);
    run("valu+branch", k_branch, 96, out, n_cu, mhz);
    run("log/rcp", k_trans, 64, out, n_cu, mhz);
    run("fma_f64", k_f64, 64, out, n_cu, mhz);
    run("cvt f64", k_cvt, 64, out, n_cu, mhz);
    run("dpp", k_dpp, 64, out, n_cu, mhz);
    run("readlane", k_readlane, 64, out, n_cu, mhz);
    return 0;
}
