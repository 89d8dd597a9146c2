// Residency probe (round 6): waves that HOLD registers and do nothing -- what do the scan kernels lose when K0-like waves sit on
// the SIMDs' register files?  squat_start(waves_per_simd, regs) launches 256 x 4 x waves_per_simd single-wave workgroups whose
// kernel is compiled for `regs` vector registers (128 / 192 / 256: one instance each) and sleeps until squat_stop() raises a flag
// in pinned host memory.  Built by: hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/probes/squatter.hip -o tools/probes/libsquatter.so
#include <hip/hip_runtime.h>
#include <cstdio>

template <int REGS> __device__ __forceinline__ void hold();
template <> __device__ __forceinline__ void hold<64>() { asm volatile("" ::: "v63"); }
template <> __device__ __forceinline__ void hold<96>() { asm volatile("" ::: "v95"); }
template <> __device__ __forceinline__ void hold<128>() { asm volatile("" ::: "v127"); }
template <> __device__ __forceinline__ void hold<192>() { asm volatile("" ::: "v191"); }
template <> __device__ __forceinline__ void hold<256>() { asm volatile("" ::: "v255"); }

template <int REGS>
__global__ __launch_bounds__(64) void squat_kernel(const volatile int *flag, unsigned long long *alive)
{
    hold<REGS>();
    if (threadIdx.x == 0) atomicAdd(alive, 1ull);
    for (int spin = 0; spin < 200000; ++spin) {             // (a guard: ~3 s -- the first version waited 400 s for a flag it never saw)
        // (the flag lives in DEVICE memory: the first version polled pinned host memory from 1 024+ waves and choked the host link --
        //  the calls' own uploads and result copies crawled, 0.11 -> 4 ms per step, which had nothing to do with registers)
        if (__hip_atomic_load(const_cast<const int *>(flag), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
#pragma unroll
        for (int z = 0; z < 8; ++z) __builtin_amdgcn_s_sleep(127);
    }
    hold<REGS>();
}

static int *g_flag = nullptr;          // device memory
static hipStream_t g_stop_stream = nullptr;
static unsigned long long *g_alive = nullptr;
static hipStream_t g_stream = nullptr;

extern "C" int squat_start(int waves_per_simd, int regs)
{
    if (!g_flag) {
        if (hipMalloc(reinterpret_cast<void **>(&g_flag), sizeof(int)) != hipSuccess) return -1;
        if (hipStreamCreateWithFlags(&g_stop_stream, hipStreamNonBlocking) != hipSuccess) return -1;
        if (hipMalloc(reinterpret_cast<void **>(&g_alive), sizeof(unsigned long long)) != hipSuccess) return -1;
        if (hipStreamCreateWithFlags(&g_stream, hipStreamNonBlocking) != hipSuccess) return -1;
    }
    (void)hipMemsetAsync(g_flag, 0, sizeof(int), g_stream);
    (void)hipMemsetAsync(g_alive, 0, sizeof(unsigned long long), g_stream);
    hipDeviceProp_t p;
    (void)hipGetDeviceProperties(&p, 0);
    const unsigned grid = static_cast<unsigned>(p.multiProcessorCount) * 4u * static_cast<unsigned>(waves_per_simd);
    int *dflag = g_flag;
    if (grid == 0) return 0;
    switch (regs) {
    case 64: hipLaunchKernelGGL(squat_kernel<64>, dim3(grid), dim3(64), 0, g_stream, dflag, g_alive); break;
    case 96: hipLaunchKernelGGL(squat_kernel<96>, dim3(grid), dim3(64), 0, g_stream, dflag, g_alive); break;
    case 128: hipLaunchKernelGGL(squat_kernel<128>, dim3(grid), dim3(64), 0, g_stream, dflag, g_alive); break;
    case 192: hipLaunchKernelGGL(squat_kernel<192>, dim3(grid), dim3(64), 0, g_stream, dflag, g_alive); break;
    case 256: hipLaunchKernelGGL(squat_kernel<256>, dim3(grid), dim3(64), 0, g_stream, dflag, g_alive); break;
    default: return -3;
    }
    return hipGetLastError() == hipSuccess ? static_cast<int>(grid) : -4;
}

// how many squatter waves have started so far (they start as slots free up)
extern "C" long long squat_alive(void)
{
    unsigned long long v = 0;
    if (!g_alive || hipMemcpyAsync(&v, g_alive, sizeof v, hipMemcpyDeviceToHost, g_stop_stream) != hipSuccess || hipStreamSynchronize(g_stop_stream) != hipSuccess) return -1;
    return static_cast<long long>(v);
}

extern "C" int squat_stop(void)
{
    if (!g_flag) return 0;
    static const int one = 1;
    if (hipMemcpyAsync(g_flag, &one, sizeof(int), hipMemcpyHostToDevice, g_stop_stream) != hipSuccess) return -1;
    if (hipStreamSynchronize(g_stop_stream) != hipSuccess) return -1;
    return hipStreamSynchronize(g_stream) == hipSuccess ? 0 : -1;
}
