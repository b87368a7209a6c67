// Microbenchmark 5: 12 KiB tiles per one-wave workgroup; tile -> workgroup mapping variants.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// MODE 0: tile = i                      (dispatch order)
// MODE 1: tile = (i%8)*(n/8) + i/8      (XCD x sweeps its own contiguous eighth)
// MODE 2: tile = (i%8)*(n/8) + i/8, n/8 rounded so each eighth starts 32 KiB-aligned ... same as 1 here
// MODE 3: tile = (i%4)*(n/4) + i/4      (pairs of XCDs share a quarter)
// MODE 4: tile = (i%16)*(n/16) + i/16
template <int MODE, int NV>
__global__ __launch_bounds__(64) void k(float4 *__restrict__ out, int n) {
    const int i = blockIdx.x;
    int t = i;
    if (MODE == 1) t = (i % 8) * (n / 8) + i / 8;
    if (MODE == 3) t = (i % 4) * (n / 4) + i / 4;
    if (MODE == 4) t = (i % 16) * (n / 16) + i / 16;
    const float4 z = make_float4(1.f, 2.f, 3.f, 4.f);
    float4 *b = out + (size_t)t * NV * 64 + threadIdx.x;
#pragma unroll
    for (int q = 0; q < NV; ++q) b[q * 64] = z;
}

template <int MODE, int NV>
float once(float4 *out, size_t bytes) {
    const int n = (int)(bytes / ((size_t)NV * 1024));
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    k<MODE, NV><<<n, 64>>>(out, n);
    CHECK(hipEventRecord(a));
    for (int i = 0; i < 10; ++i) k<MODE, NV><<<n, 64>>>(out, n);
    CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
    float ms; CHECK(hipEventElapsedTime(&ms, a, b));
    return ms / 10;
}

int main() {
    const size_t bytes = (size_t)76800 * 12288;
    float4 *out; CHECK(hipMalloc(&out, bytes + (1 << 20)));
    const char *names[8] = {"12K dispatch order", "12K XCD-eighths", "12K quarters", "12K sixteenths",
                            "4K dispatch order", "4K XCD-eighths", "6K XCD-eighths", "24K XCD-eighths"};
    float best[8]; for (int i = 0; i < 8; ++i) best[i] = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        float t[8] = {once<0, 12>(out, bytes), once<1, 12>(out, bytes), once<3, 12>(out, bytes), once<4, 12>(out, bytes),
                      once<0, 4>(out, bytes), once<1, 4>(out, bytes), once<1, 6>(out, bytes), once<1, 24>(out, bytes)};
        for (int i = 0; i < 8; ++i) if (t[i] < best[i]) best[i] = t[i];
    }
    for (int i = 0; i < 8; ++i) printf("%-22s best %6.1f us  %.2f TB/s\n", names[i], best[i] * 1e3, bytes / (best[i] * 1e-3) / 1e12);
    return 0;
}
