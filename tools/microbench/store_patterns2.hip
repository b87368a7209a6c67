// Microbenchmark 2: bytes per wave / store flavour vs achieved HBM write bandwidth.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// each wave writes NV consecutive 1 KiB vectors; MODE 0 plain, 1 nontemporal, 2 wave-interleaved inside a 256-thread block
template <int NV, int MODE, int TPB>
__global__ __launch_bounds__(TPB) void k(float4 *__restrict__ out, size_t nvec_total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int WPB = TPB / 64;
    const float4 z = make_float4(1.f, 2.f, 3.f, 4.f);
    if (MODE == 2) {
        // block covers WPB*NV KiB; store q of wave w goes to vector (q*WPB + w)
        float4 *base = out + (size_t)blockIdx.x * WPB * NV * 64;
#pragma unroll
        for (int q = 0; q < NV; ++q) base[(size_t)(q * WPB + wave) * 64 + lane] = z;
    } else {
        float4 *base = out + ((size_t)blockIdx.x * WPB + wave) * NV * 64;
#pragma unroll
        for (int q = 0; q < NV; ++q) {
            if (MODE == 1) {
                float *p = reinterpret_cast<float *>(&base[q * 64 + lane]);
                __builtin_nontemporal_store(z.x, p); __builtin_nontemporal_store(z.y, p + 1);
                __builtin_nontemporal_store(z.z, p + 2); __builtin_nontemporal_store(z.w, p + 3);
            } else base[q * 64 + lane] = z;
        }
    }
}

template <int NV, int MODE, int TPB>
void run(const char *name, float4 *out, size_t bytes) {
    const size_t per_block = (size_t)(TPB / 64) * NV * 1024;
    const int grid = (int)(bytes / per_block);
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) k<NV, MODE, TPB><<<grid, TPB>>>(out, bytes / 16);
    CHECK(hipEventRecord(a));
    for (int i = 0; i < 20; ++i) k<NV, MODE, TPB><<<grid, TPB>>>(out, bytes / 16);
    CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
    float ms; CHECK(hipEventElapsedTime(&ms, a, b)); ms /= 20;
    printf("%-44s %8.1f us  %6.2f TB/s\n", name, ms * 1e3, bytes / (ms * 1e-3) / 1e12);
}

int main() {
    const size_t bytes = (size_t)76800 * 12288;  // 943.7 MB, the config-2 float64 output
    float4 *out; CHECK(hipMalloc(&out, bytes + (1 << 20)));
    run<1, 0, 64>("1 KiB/wave, 64 thr", out, bytes);
    run<1, 0, 256>("1 KiB/wave, 256 thr (torch-fill shape)", out, bytes);
    run<4, 0, 64>("4 KiB/wave, 64 thr", out, bytes);
    run<4, 2, 256>("4 KiB/wave interleaved, 256 thr", out, bytes);
    run<12, 0, 64>("12 KiB/wave, 64 thr", out, bytes);
    run<12, 1, 64>("12 KiB/wave, 64 thr, nontemporal", out, bytes);
    run<12, 2, 256>("12 KiB/wave interleaved, 256 thr", out, bytes);
    run<12, 0, 256>("12 KiB/wave, 256 thr", out, bytes);
    run<3, 0, 64>("3 KiB/wave, 64 thr", out, bytes);
    run<6, 0, 64>("6 KiB/wave, 64 thr", out, bytes);
    run<6, 1, 64>("6 KiB/wave, 64 thr, nontemporal", out, bytes);
    run<24, 0, 64>("24 KiB/wave, 64 thr", out, bytes);
    return 0;
}
