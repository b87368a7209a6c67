// Microbenchmark 7: 12 KiB one-wave tiles (XCD-contiguous eighths, the builder's write pattern) as a function of
// RESIDENT WAVES PER CU (forced through the dynamic LDS size) and of a per-wave delay before the stores
// (s_sleep, standing in for the builder's load + reduce phase).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int NV, int SLEEP>
__global__ __launch_bounds__(64) void k(float4 *__restrict__ out, int n) {
    extern __shared__ float4 lds[];
    const int i = blockIdx.x;
    const int t = (i % 8) * (n / 8) + i / 8;
    if (SLEEP) {
#pragma unroll
        for (int s = 0; s < SLEEP; ++s) __builtin_amdgcn_s_sleep(127);   // ~127*64 cycles each
    }
    float4 z = make_float4(1.f, 2.f, 3.f, 4.f);
    if (out == nullptr) z = lds[threadIdx.x];   // keep the LDS allocation alive
    float4 *b = out + (size_t)t * NV * 64 + threadIdx.x;
#pragma unroll
    for (int q = 0; q < NV; ++q) b[q * 64] = z;
}

template <int NV, int SLEEP>
float once(float4 *out, size_t bytes, size_t lds) {
    const int n = (int)(bytes / ((size_t)NV * 1024));
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    k<NV, SLEEP><<<n, 64, lds>>>(out, n);
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipEventRecord(a));
        for (int i = 0; i < 10; ++i) k<NV, SLEEP><<<n, 64, lds>>>(out, n);
        CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
        float ms; CHECK(hipEventElapsedTime(&ms, a, b));
        if (ms / 10 < best) best = ms / 10;
    }
    return best;
}

int main() {
    const size_t bytes = (size_t)76800 * 12288;
    float4 *out; CHECK(hipMalloc(&out, bytes + (1 << 20)));
    const size_t lds[] = {0, 5120, 6656, 8192, 10240, 13312, 20480, 40960};
    for (size_t l : lds) {
        const int waves = l ? (int)(163840 / l) : 32;
        const float t0 = once<12, 0>(out, bytes, l), t1 = once<12, 1>(out, bytes, l), t3 = once<12, 3>(out, bytes, l);
        printf("LDS %6zu B (<= %2d waves/CU): no delay %6.1f us %.2f TB/s | 3.4 us delay %6.1f us %.2f TB/s | 10 us delay %6.1f us %.2f TB/s\n", l,
               waves > 32 ? 32 : waves, t0 * 1e3, bytes / (t0 * 1e-3) / 1e12, t1 * 1e3, bytes / (t1 * 1e-3) / 1e12, t3 * 1e3, bytes / (t3 * 1e-3) / 1e12);
    }
    return 0;
}
