// Diagnostic micro-benchmark (not product code): cost of staging one 10k-sample window into LDS
// with different load shapes, cold (HBM) vs warm (L2), as seen by one workgroup and by a full grid.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int NT, int MODE>
__global__ __launch_bounds__(NT) void stage(const float *x, long stride, int n, int reps, long long *cyc, float *sink)
{
    extern __shared__ int ys[];
    const float *p = x + (long)blockIdx.x * stride;
    long long t0 = clock64(), acc = 0;
    float s = 0;
    for (int r = 0; r < reps; ++r) {
        const float *q = p + (long)r * (n / 2);          // overlapping windows like the spine
        if (MODE == 0) {                                  // dword, all loads first
            constexpr int U = 10240 / NT;
            float raw[U];
#pragma unroll
            for (int u = 0; u < U; ++u) raw[u] = q[min((int)threadIdx.x + u * NT, n - 1)];
#pragma unroll
            for (int u = 0; u < U; ++u) if (threadIdx.x + u * NT < n) ys[threadIdx.x + u * NT] = (int)raw[u];
        } else if (MODE == 1) {                           // dwordx4 (requires 16B alignment of q)
            constexpr int U = 10240 / NT / 4;
            float4 raw[U];
#pragma unroll
            for (int u = 0; u < U; ++u) raw[u] = reinterpret_cast<const float4 *>(q)[min((int)threadIdx.x + u * NT, n / 4 - 1)];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                int i = (threadIdx.x + u * NT) * 4;
                if (i < n) { ys[i] = (int)raw[u].x; ys[i + 1] = (int)raw[u].y; ys[i + 2] = (int)raw[u].z; ys[i + 3] = (int)raw[u].w; }
            }
        } else {                                          // naive loop
            for (int i = threadIdx.x; i < n; i += NT) ys[i] = (int)q[i];
        }
        __syncthreads();
        s += (float)ys[(threadIdx.x * 7) % n];
        __syncthreads();
    }
    acc = clock64() - t0;
    if (threadIdx.x == 0) cyc[blockIdx.x] = acc;
    if (s == 12345.f) sink[0] = s;
}

template <int NT, int MODE> int run(const float *d, long total, int grid, int reps, const char *name)
{
    long long *dc; float *sink;
    CK(hipMalloc(&dc, grid * sizeof(long long))); CK(hipMalloc(&sink, 4));
    const int n = 10000;
    long stride = total / grid;
    for (int pass = 0; pass < 2; ++pass) {
        hipLaunchKernelGGL((stage<NT, MODE>), dim3(grid), dim3(NT), 10240 * 4, 0, d, stride, n, reps, dc, sink);
        CK(hipDeviceSynchronize());
        std::vector<long long> h(grid);
        CK(hipMemcpy(h.data(), dc, grid * sizeof(long long), hipMemcpyDeviceToHost));
        double m = 0; for (auto v : h) m += v; m /= grid;
        printf("%-28s NT=%4d grid=%4d pass=%d: %8.0f cycles/window\n", name, NT, grid, pass, m / reps);
    }
    CK(hipFree(dc)); CK(hipFree(sink));
    return 0;
}

int main()
{
    const long total = 100000000;
    float *d; CK(hipMalloc(&d, total * sizeof(float)));
    CK(hipMemset(d, 0, total * sizeof(float)));
    for (int grid : {1, 256, 625}) {
        run<512, 2>(d, total, grid, 30, "naive loop");
        run<512, 0>(d, total, grid, 30, "dword batched");
        run<512, 1>(d, total, grid, 30, "dwordx4 batched");
        run<1024, 0>(d, total, grid, 30, "dword batched");
        run<256, 1>(d, total, grid, 30, "dwordx4 batched");
    }
    return 0;
}
