// capi_ergo12.cpp -- the C ABI of include/evrep.h driven from plain C++ with the HIP runtime only
// (no Python, no torch): B synthetic windows -> evrep_bin_events -> evrep_optimized -> FNV-1a hash
// of the (B, H, W, 12) float64 result.  tests/test_gpu_capi_native.py checks the hash against the
// Python engine and the CPU oracle on the same events.
//
//   hipcc -std=c++17 -I include examples/capi_ergo12.cpp -L event_representation_study_amd -levrep \
//         -Wl,-rpath,$PWD/event_representation_study_amd -o examples/capi_ergo12
//   examples/capi_ergo12 [B] [events_per_window] [H] [W]
#include <hip/hip_runtime.h>

#include <cinttypes>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "evrep.h"

#define HIP_OK(call)                                                                              \
    do {                                                                                          \
        hipError_t e_ = (call);                                                                   \
        if (e_ != hipSuccess) {                                                                   \
            std::fprintf(stderr, "%s failed: %s\n", #call, hipGetErrorString(e_));               \
            return 2;                                                                             \
        }                                                                                         \
    } while (0)
#define EVREP_DO(call)                                                                            \
    do {                                                                                          \
        int rc_ = (call);                                                                         \
        if (rc_ != EVREP_OK) {                                                                    \
            std::fprintf(stderr, "%s -> %d (%s)\n", #call, rc_, evrep_last_hip_error());         \
            return 3;                                                                             \
        }                                                                                         \
    } while (0)

static uint64_t splitmix64(uint64_t &s) {
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

int main(int argc, char **argv) {
    const int B = argc > 1 ? std::atoi(argv[1]) : 4;
    const int64_t N = argc > 2 ? std::atoll(argv[2]) : 20000;
    const int H = argc > 3 ? std::atoi(argv[3]) : 240;
    const int W = argc > 4 ? std::atoi(argv[4]) : 304;

    // window b: x, y uniform, p in {-1,+1}, t a non-decreasing walk (steps 0..3 us) -- all from splitmix64(seed = b+1)
    std::vector<int32_t> ev((size_t)B * N * 4);
    std::vector<int64_t> off(B + 1);
    for (int b = 0; b < B; ++b) {
        uint64_t s = (uint64_t)b + 1;
        int32_t t = 0;
        off[b] = (int64_t)b * N;
        for (int64_t i = 0; i < N; ++i) {
            const uint64_t r = splitmix64(s);
            int32_t *e = &ev[((size_t)b * N + i) * 4];
            e[0] = (int32_t)((r & 0xffffffffull) % (uint64_t)W);
            e[1] = (int32_t)(((r >> 32) & 0xfffffffull) % (uint64_t)H);
            t += (int32_t)((r >> 60) & 3u);
            e[2] = t;
            e[3] = ((r >> 62) & 1u) ? 1 : -1;
        }
    }
    off[B] = (int64_t)B * N;

    evrep_plan plan;
    EVREP_DO(evrep_plan_init(&plan, B, H, W, (int64_t)B * N, N));
    const size_t out_bytes = (size_t)B * H * W * 12 * sizeof(double);
    void *d_ev = nullptr, *d_off = nullptr, *d_ws = nullptr, *d_out = nullptr;
    hipStream_t stream;
    HIP_OK(hipStreamCreate(&stream));
    HIP_OK(hipMalloc(&d_ev, ev.size() * sizeof(int32_t) + 16));
    HIP_OK(hipMalloc(&d_off, off.size() * sizeof(int64_t)));
    HIP_OK(hipMalloc(&d_ws, evrep_workspace_bytes(&plan)));
    HIP_OK(hipMalloc(&d_out, out_bytes));
    HIP_OK(hipMemcpyAsync(d_ev, ev.data(), ev.size() * sizeof(int32_t), hipMemcpyHostToDevice, stream));
    HIP_OK(hipMemcpyAsync(d_off, off.data(), off.size() * sizeof(int64_t), hipMemcpyHostToDevice, stream));

    EVREP_DO(evrep_bin_events(&plan, (const int32_t *)d_ev, (const int64_t *)d_off, d_ws, stream));
    EVREP_DO(evrep_optimized(&plan, (const int32_t *)d_ev, (const int64_t *)d_off, d_ws, 1.0, EVREP_F64, d_out, stream));
    std::vector<uint32_t> status(B);
    EVREP_DO(evrep_read_status(&plan, d_ws, status.data(), stream));

    std::vector<double> out((size_t)B * H * W * 12);
    HIP_OK(hipMemcpyAsync(out.data(), d_out, out_bytes, hipMemcpyDeviceToHost, stream));
    HIP_OK(hipStreamSynchronize(stream));

    uint64_t h = 0xcbf29ce484222325ull;  // FNV-1a over the raw bytes
    const unsigned char *p = reinterpret_cast<const unsigned char *>(out.data());
    for (size_t i = 0; i < out_bytes; ++i) { h ^= p[i]; h *= 0x100000001b3ull; }
    uint32_t st = 0;
    for (int b = 0; b < B; ++b) st |= status[b];
    std::printf("{\"B\": %d, \"N\": %" PRId64 ", \"H\": %d, \"W\": %d, \"abi\": %d, \"status_or\": %u, \"fnv1a64\": \"%016" PRIx64 "\"}\n",
                B, N, H, W, evrep_abi_version(), st, h);
    (void)hipFree(d_out); (void)hipFree(d_ws); (void)hipFree(d_off); (void)hipFree(d_ev);
    (void)hipStreamDestroy(stream);
    return 0;
}
