// round 5: do 16-byte global loads from 2-byte-aligned addresses return the right bytes on this part (events cut out of an int16
// file trace start at any sample), and what do they cost?   hipcc --offload-arch=gfx950 -O3 tools/probes/unaligned_probe.hip -o /tmp/unaligned_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
__global__ void k(const char *p, int off, long long n16, int4 *out, unsigned long long *sum)
{
    long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    unsigned long long s = 0;
    for (; i < n16; i += (long long)gridDim.x * blockDim.x) {
        const int4 v = *reinterpret_cast<const int4 *>(p + off + 16 * i);
        if (out) out[i] = v;
        s += (unsigned)v.x + (unsigned)v.y + (unsigned)v.z + (unsigned)v.w;
    }
    if (s == 0x123456789abcdefull) atomicAdd(sum, s);
}
int main()
{
    const long long n16 = 1 << 24;                       // 256 MB
    std::vector<short> h((n16 + 2) * 8);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (short)(i * 2654435761u >> 13);
    char *d; int4 *o; unsigned long long *s;
    hipMalloc(&d, h.size() * 2); hipMalloc(&o, n16 * 16); hipMalloc(&s, 8);
    hipMemcpy(d, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    std::vector<int4> back(n16);
    for (int off : {0, 2, 4, 6, 8, 10, 14}) {
        hipLaunchKernelGGL(k, dim3(4096), dim3(256), 0, 0, d, off, n16, o, s);
        if (hipDeviceSynchronize() != hipSuccess) { printf("offset %d: FAULT %s\n", off, hipGetErrorString(hipGetLastError())); return 1; }
        hipMemcpy(back.data(), o, n16 * 16, hipMemcpyDeviceToHost);
        const bool ok = std::memcmp(back.data(), reinterpret_cast<const char *>(h.data()) + off, n16 * 16) == 0;
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        hipEventRecord(a, 0);
        for (int r = 0; r < 10; ++r) hipLaunchKernelGGL(k, dim3(4096), dim3(256), 0, 0, d, off, n16, (int4 *)nullptr, s);
        hipEventRecord(b, 0); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("offset %2d bytes: %s, read-only pass %.1f us = %.2f TB/s\n", off, ok ? "bytes equal" : "WRONG BYTES", ms * 100, n16 * 16 / (ms / 10 * 1e-3) / 1e12);
    }
    return 0;
}
