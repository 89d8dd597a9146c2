// round 5: is v_cvt_pknorm_i16_f32(y * (1/32767)) == y for every integer |y| <= 32767 (and saturating beyond)?  K0 would pack the
// counts of fp32 samples with it.   hipcc --offload-arch=gfx950 -O3 tools/probes/pknorm_probe.hip -o tools/probes/pknorm_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int lo, int n, int *bad, int *first_bad)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int y = lo + i;
    const float c = 1.0f / 32767.0f;
    // as K0 would form it: t = x * inv_q is the count as a float, m its centre; (t - m) * c in one fma: t * c - m * c
    for (int m = -3; m <= 3; m += 3) {
        const float t = static_cast<float>(y + 5000 * m), mf = static_cast<float>(5000 * m);
        const float a = fmaf(t, c, -(mf * c));
        const float b = (t - mf) * c;
        typedef short v2s __attribute__((ext_vector_type(2)));
        const v2s pa = __builtin_amdgcn_cvt_pknorm_i16(a, b);
        const short ra = pa.x, rb = pa.y;
        const int want = y > 32767 ? 32767 : (y < -32767 ? -32767 : y);
        if (rb != want) { atomicAdd(&bad[0], 1); atomicMin(&first_bad[0], y < 0 ? -y : y); }
        if (ra != want) { atomicAdd(&bad[1], 1); atomicMin(&first_bad[1], y < 0 ? -y : y); }
    }
}
int main()
{
    int *bad, *fb; hipMalloc(&bad, 8); hipMalloc(&fb, 8);
    int z[2] = {0, 0}, big[2] = {1 << 30, 1 << 30};
    hipMemcpy(bad, z, 8, hipMemcpyHostToDevice); hipMemcpy(fb, big, 8, hipMemcpyHostToDevice);
    const int lo = -70000, n = 140001;
    hipLaunchKernelGGL(k, dim3((n + 255) / 256), dim3(256), 0, 0, lo, n, bad, fb);
    hipDeviceSynchronize();
    hipMemcpy(z, bad, 8, hipMemcpyDeviceToHost); hipMemcpy(big, fb, 8, hipMemcpyDeviceToHost);
    printf("(t - m) * c: %d mismatches (smallest |y| %d); fma(t, c, -m c): %d mismatches (smallest |y| %d)\n", z[0], big[0], z[1], big[1]);
    return 0;
}
