// Issue behaviour of the block-sum scan's row body in isolation (registers only, no memory): does a second wave on a
// SIMD add throughput?  hipcc --offload-arch=gfx950 -O3 -o microbench_row microbench_row.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("hip error %s\n", hipGetErrorString(e_)); return 1; } } while (0)
typedef float f2 __attribute__((ext_vector_type(2)));
constexpr int N = 1 << 15;
__device__ __forceinline__ int below(int x) { return __builtin_amdgcn_update_dpp(x, x, 0x138, 0xf, 0xf, false); }
__device__ __forceinline__ float belowf(float x) { return __int_as_float(below(__float_as_int(x))); }
struct Top2 { float b, s; int i; };
__device__ __forceinline__ void top2_push(Top2 &t, float g, int i)
{
    t.s = __builtin_amdgcn_fmed3f(t.b, g, t.s); t.i = g > t.b ? i : t.i; t.b = fmaxf(t.b, g);
}
template <int VARIANT>
__global__ __launch_bounds__(64) void k(double *out, int seed, float Tprune)
{
    const int lane = threadIdx.x;
    int a1 = seed + lane, J = 1000 + 8 * lane, qcount = 0;
    double a2 = 1.0e6 + 977.0 * lane, nld = 800.0 + 8 * lane;
    float nlf = 800.0f + 8 * lane;
    const double T1d = 5.0e6, T2 = 9.0e9, dn = 10000.0;
    const float nf = 10000.0f, vfloor = 0.01f, LOG2E = 1.4426950408889634f;
    const f2 cc = {3.0f, 3.0f};
    const int ps = 0, n = 10000, cand_lo = 100;
    const unsigned crange = 9800;
    Top2 top = {-INFINITY, -INFINITY, -1};
    unsigned flag = 0;
    for (int it = 0; it < N; ++it) {
        a1 += 3 + (it & 3); a2 += 4099.0;
        const int nl = J - ps;
        const double a1d = static_cast<double>(a1), b1d = T1d - a1d;
        const double DL = fma(nld, a2, -(a1d * a1d));
        const double DR = fma(dn - nld, T2 - a2, -(b1d * b1d));
        const float nrf = nf - nlf;
        const f2 D = {static_cast<float>(DL), static_cast<float>(DR)};
        const f2 nv = {nlf, nrf};
        const f2 rr = {__builtin_amdgcn_rcpf(nlf), __builtin_amdgcn_rcpf(nrf)};
        const f2 u = D * rr * rr;
        const f2 lgu = {__builtin_amdgcn_logf(u.x), __builtin_amdgcn_logf(u.y)};
        const f2 lg = lgu - cc;
        const f2 tt = nv * lg;
        const float g = -(tt.x + tt.y);
        const bool valid = static_cast<unsigned>(nl - 1) < static_cast<unsigned>(n - 1);
        const bool okL = valid && u.x >= vfloor, okR = valid && u.y >= vfloor;
        const bool inr = static_cast<unsigned>(J - cand_lo) <= crange && (lane != 0 || it == 0);
        const float ge = (inr && okL && okR) ? g : -INFINITY;
        top2_push(top, ge, nl);
        flag |= static_cast<unsigned>(inr && !(okL && okR));
        if (VARIANT >= 1) {
            const float aL = belowf(lg.x), rlb = belowf(rr.x);
            const bool pokL = below(static_cast<int>(okL)) != 0;
            const bool blk = lane >= 1 && static_cast<unsigned>(J - 1 - cand_lo) <= crange + 6u;
            const float nl0f = nlf - 8.0f, nlef = nlf - 1.0f, nr0f = nrf + 8.0f, nref = nrf + 1.0f;
            const float bR = lg.y, rre = rr.y;
            const float cR0 = bR - 8.0f * LOG2E * rre, cR1 = bR - LOG2E * rre, cL1 = aL - 7.0f * LOG2E * rlb;
            const float h0 = -fmaf(nl0f, aL, nr0f * cR0), h1 = -fmaf(nlef, cL1, nref * cR1);
            const bool pruned = pokL && okR && nl >= 9 && fmaxf(h0, h1) < Tprune;
            const bool keep = blk && !pruned;
            if (VARIANT >= 2) {
                const unsigned long long km = __ballot(keep);
                if (km) qcount += __popcll(km);
            } else flag |= keep;
        }
        J += 504; nlf += 504.0f; nld += 504.0;
        if (J > 9000) { J -= 8000; nlf -= 8000.0f; nld -= 8000.0; }
    }
    out[blockIdx.x * 64 + lane] = top.b + top.s + top.i + flag + qcount + a2;
}
template <int V> int run(const char *name)
{
    double *d; CHK(hipMalloc(&d, sizeof(double) * 64 * 4096));
    for (int waves_per_cu : {1, 4, 8, 16}) {
        int grid = 256 * waves_per_cu;
        hipLaunchKernelGGL(k<V>, dim3(grid), dim3(64), 0, 0, d, 7, -1.0e30f);
        CHK(hipDeviceSynchronize());
        hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
        CHK(hipEventRecord(e0)); hipLaunchKernelGGL(k<V>, dim3(grid), dim3(64), 0, 0, d, 7, -1.0e30f); CHK(hipEventRecord(e1));
        CHK(hipDeviceSynchronize());
        float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-34s waves/CU %2d: %.1f ns per boundary per wave, %.2f boundaries/us/CU\n", name, waves_per_cu, ms * 1e6 / N,
               waves_per_cu * N / (ms * 1e3));
    }
    CHK(hipFree(d));
    return 0;
}
int main() { run<0>("eval + top2"); run<1>("eval + top2 + dpp + bound"); run<2>("... + ballot/branch"); return 0; }
