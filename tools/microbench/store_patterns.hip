// Microbenchmark: how fast can one-wave workgroups stream 12 KiB tiles to HBM, and which ingredient
// of the builder skeleton (LDS allocation, LDS round trip, dependent scalar load) costs what?
// build: hipcc --offload-arch=gfx950 -O3 -o store_patterns store_patterns.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

constexpr int kTileBytes = 12288;

template <int MODE, int TPB>
__global__ __launch_bounds__(TPB) void k_store(float4 *__restrict__ out, const unsigned *__restrict__ offs, int ntiles) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int tile = blockIdx.x * (TPB / 64) + wave;
    if (tile >= ntiles) return;
    float4 *dst = out + (size_t)tile * (kTileBytes / 16);
    float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    if (MODE & 4) {  // dependent scalar load before anything else (like chunk_off)
        unsigned a = offs[tile], b = offs[tile + 1];
        if (a != b) z.x = 1.0f;
    }
    if (MODE & 1) {  // LDS round trip
        float4 *t = reinterpret_cast<float4 *>(smem) + wave * (kTileBytes / 16);
#pragma unroll
        for (int q = 0; q < 12; ++q) t[lane + 64 * q] = z;
        __builtin_amdgcn_wave_barrier();
        float4 v[12];
#pragma unroll
        for (int q = 0; q < 12; ++q) v[q] = t[lane + 64 * q];
#pragma unroll
        for (int q = 0; q < 12; ++q) dst[lane + 64 * q] = v[q];
    } else {
#pragma unroll
        for (int q = 0; q < 12; ++q) dst[lane + 64 * q] = z;
    }
}

template <int MODE, int TPB>
float run(float4 *out, const unsigned *offs, int ntiles, size_t lds, int iters) {
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    const int wpb = TPB / 64;
    const int grid = (ntiles + wpb - 1) / wpb;
    for (int i = 0; i < 3; ++i) k_store<MODE, TPB><<<grid, TPB, lds>>>(out, offs, ntiles);
    CHECK(hipEventRecord(a));
    for (int i = 0; i < iters; ++i) k_store<MODE, TPB><<<grid, TPB, lds>>>(out, offs, ntiles);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms; CHECK(hipEventElapsedTime(&ms, a, b));
    return ms / iters;
}

int main() {
    const int ntiles = 76800;
    const size_t bytes = (size_t)ntiles * kTileBytes;
    float4 *out; unsigned *offs;
    CHECK(hipMalloc(&out, bytes));
    CHECK(hipMalloc(&offs, (ntiles + 1) * 4));
    CHECK(hipMemset(offs, 0, (ntiles + 1) * 4));
    struct R { const char *name; float ms; };
    std::vector<R> r;
    r.push_back({"direct, 64 thr, no LDS", run<0, 64>(out, offs, ntiles, 0, 20)});
    r.push_back({"direct, 64 thr, 16.5K LDS alloc", run<0, 64>(out, offs, ntiles, 16896, 20)});
    r.push_back({"direct, 64 thr, 8K LDS alloc", run<0, 64>(out, offs, ntiles, 8192, 20)});
    r.push_back({"LDS round trip, 64 thr, 16.5K", run<1, 64>(out, offs, ntiles, 16896, 20)});
    r.push_back({"dep. scalar load + direct, 64 thr, 16.5K", run<4, 64>(out, offs, ntiles, 16896, 20)});
    r.push_back({"dep. scalar load + LDS, 64 thr, 16.5K", run<5, 64>(out, offs, ntiles, 16896, 20)});
    r.push_back({"direct, 256 thr, no LDS", run<0, 256>(out, offs, ntiles, 0, 20)});
    r.push_back({"LDS round trip, 256 thr, 4x12K", run<1, 256>(out, offs, ntiles, 4 * 12288, 20)});
    r.push_back({"dep + LDS, 256 thr, 4x12K", run<5, 256>(out, offs, ntiles, 4 * 12288, 20)});
    for (auto &x : r) printf("%-45s %8.1f us  %6.2f TB/s\n", x.name, x.ms * 1e3, bytes / (x.ms * 1e-3) / 1e12);
    return 0;
}
