// Round 6 probe: the bench's headline job -- T contexts on one GPU, each segmenting its own 1e8-sample fp32 trace, K steps --
// driven by T NATIVE threads through the C ABI, no Python in the loop.  Against `python bench.py` on the same box it says what
// the interpreter (the GIL between sixteen host threads, ctypes marshalling, torch's stream sync) costs the pool's step.
//   hipcc -O2 -std=c++17 tools/probes/pool_native.cpp -Iinclude -Lpypore_amd -lporeseg -Wl,-rpath,$PWD/pypore_amd -lpthread -o /tmp/pool_native
//   /tmp/pool_native [T=16] [K=100] [W=32] [n=100000000]
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>
#include "poreseg.h"

static uint64_t splitmix64(uint64_t z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

int main(int argc, char **argv)
{
    const int T = argc > 1 ? std::atoi(argv[1]) : 16, K = argc > 2 ? std::atoi(argv[2]) : 100, Wm = argc > 3 ? std::atoi(argv[3]) : 32;
    const int64_t n = argc > 4 ? std::atoll(argv[4]) : 100000000;
    const int32_t levels[5] = {1600, 1344, 1760, 1216, 1504};
    ps_split_params p = {};
    p.min_width = 100; p.max_width = 1000000; p.window_width = 10000;
    p.prior_segments_per_second = 10.0; p.sampling_freq = 1e5;
    double mg = 0;
    if (ps_min_gain(&p, &mg)) { std::fprintf(stderr, "ps_min_gain failed\n"); return 1; }
    std::vector<ps_ctx *> ctx(T);
    std::vector<void *> trace(T), out(T);
    const int64_t cap = n / p.min_width + 1;
    for (int t = 0; t < T; ++t) {
        if (ps_create(0, nullptr, &ctx[t])) { std::fprintf(stderr, "ps_create failed\n"); return 1; }
        ps_set_option(ctx[t], "shared_device", T);
        hipMalloc(&trace[t], n * 4);
        hipMalloc(&out[t], cap * 4);
        const uint64_t seed = 2024 + 1000ull * t;
        std::vector<int64_t> ends;
        std::vector<int32_t> lv;
        int64_t acc = 0;
        for (uint64_t k = 1; acc < n; ++k) {                  // synth.dwell_table: d_k = lo + splitmix64((seed ^ X) + k G) % (hi - lo)
            const uint64_t d = splitmix64((seed ^ 0xD1B54A32D192ED03ull) + k * 0x9E3779B97F4A7C15ull) % 19000 + 1000;
            acc += static_cast<int64_t>(d);
            ends.push_back(acc);
            lv.push_back(levels[(k - 1) % 5]);
        }
        if (ps_synth_trace(ctx[t], trace[t], PS_DTYPE_F32, n, seed, ends.data(), lv.data(), static_cast<int64_t>(ends.size()))) {
            std::fprintf(stderr, "ps_synth_trace: %s\n", ps_last_error(ctx[t])); return 1;
        }
        ps_synchronize(ctx[t]);
    }
    const ps_sample_format fmt = {PS_DTYPE_F32, 0, 1.0 / 32};
    const int64_t ev_off[2] = {0, n};
    std::vector<int64_t> nb(T, 0);
    auto run = [&](int steps) {
        std::atomic<int> bad{0};
        std::vector<std::thread> th;
        for (int t = 0; t < T; ++t)
            th.emplace_back([&, t] {
                int64_t boff[2];
                for (int k = t; k < steps; k += T) {          // job k on context k % T, like engine.StreamPool.run
                    if (ps_segment_batch(ctx[t], trace[t], &fmt, ev_off, 1, &p, static_cast<int32_t *>(out[t]), cap, boff, nullptr)) {
                        std::fprintf(stderr, "ps_segment_batch: %s\n", ps_last_error(ctx[t]));
                        ++bad;
                        return;
                    }
                    nb[t] = boff[1];
                }
            });
        for (auto &x : th) x.join();
        return bad.load();
    };
    if (run(Wm)) return 1;
    hipDeviceSynchronize();
    for (int rep = 0; rep < 3; ++rep) {
        const auto t0 = std::chrono::steady_clock::now();
        if (run(K)) return 1;
        hipDeviceSynchronize();
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        std::printf("native pool: T %d, %d steps: %.4f ms per step (%.1f %% of the HBM-read roofline), boundaries of context 0: %lld\n", T, K,
                    ms / K, 100.0 * 4.0 * n / (ms / K * 1e-3) / 8e12, static_cast<long long>(nb[0]));
    }
    for (int t = 0; t < T; ++t) { ps_destroy(ctx[t]); hipFree(trace[t]); hipFree(out[t]); }
    return 0;
}
