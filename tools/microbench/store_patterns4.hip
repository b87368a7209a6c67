// Microbenchmark 4: HBM write bandwidth vs contiguous KiB per wave (one-wave workgroups), round-robin repeats.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int NV>
__global__ __launch_bounds__(64) void k(float4 *__restrict__ out) {
    const float4 z = make_float4(1.f, 2.f, 3.f, 4.f);
    float4 *b = out + (size_t)blockIdx.x * NV * 64 + threadIdx.x;
#pragma unroll
    for (int q = 0; q < NV; ++q) b[q * 64] = z;
}

template <int NV>
float once(float4 *out, size_t bytes) {
    const int grid = (int)(bytes / ((size_t)NV * 1024));
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    k<NV><<<grid, 64>>>(out);
    CHECK(hipEventRecord(a));
    for (int i = 0; i < 10; ++i) k<NV><<<grid, 64>>>(out);
    CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
    float ms; CHECK(hipEventElapsedTime(&ms, a, b));
    return ms / 10;
}

int main() {
    const size_t bytes = (size_t)76800 * 12288;
    float4 *out; CHECK(hipMalloc(&out, bytes + (1 << 20)));
    const int sizes[9] = {1, 2, 3, 4, 6, 8, 12, 16, 24};
    float best[9]; for (int i = 0; i < 9; ++i) best[i] = 1e9f;
    float all[9][4];
    for (int rep = 0; rep < 4; ++rep) {
        float t[9] = {once<1>(out, bytes), once<2>(out, bytes), once<3>(out, bytes), once<4>(out, bytes), once<6>(out, bytes),
                      once<8>(out, bytes), once<12>(out, bytes), once<16>(out, bytes), once<24>(out, bytes)};
        for (int i = 0; i < 9; ++i) { all[i][rep] = t[i]; if (t[i] < best[i]) best[i] = t[i]; }
    }
    for (int i = 0; i < 9; ++i)
        printf("%2d KiB/wave: %6.1f %6.1f %6.1f %6.1f us   best %.2f TB/s\n", sizes[i], all[i][0] * 1e3, all[i][1] * 1e3,
               all[i][2] * 1e3, all[i][3] * 1e3, bytes / (best[i] * 1e-3) / 1e12);
    return 0;
}
