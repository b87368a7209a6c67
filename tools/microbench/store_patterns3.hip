// Microbenchmark 3: does HBM write bandwidth depend on which XCD writes which 4 KiB block?
// Workgroup i runs on XCD i % 8 (observed dispatch order).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// MODE 0: wave i writes 4 KiB block i                       (block % 8 == XCD)
// MODE 1: wave i writes 4 KiB block i ^ 1                   (block % 8 != XCD, still a bijection)
// MODE 2: wave i writes 4 KiB block (i/8) + (i%8)*nb/8      (each XCD sweeps its own contiguous eighth)
// MODE 3: wave i writes three 4 KiB blocks with block % 8 == i % 8: g = 24*(i/8) + i%8 + {0,8,16}
// MODE 4: wave i writes the 12 KiB tile i (three consecutive blocks)              [the builder today]
// MODE 5: wave i writes 12 KiB = blocks {3i, 3i+1, 3i+2} but as q-interleaved stores
// MODE 6: wave i writes 8 KiB tile i ; MODE 7: 16 KiB tile i ; MODE 8: 2 KiB tile i
template <int MODE>
__global__ __launch_bounds__(64) void k(float4 *__restrict__ out, int nwaves, int nblocks) {
    const int lane = threadIdx.x, i = blockIdx.x;
    const float4 z = make_float4(1.f, 2.f, 3.f, 4.f);
    auto put4k = [&](size_t g) {
        float4 *b = out + g * 256;
#pragma unroll
        for (int q = 0; q < 4; ++q) b[q * 64 + lane] = z;
    };
    if (MODE == 0) put4k(i);
    if (MODE == 1) put4k(i ^ 1);
    if (MODE == 2) put4k((size_t)(i / 8) + (size_t)(i % 8) * (nblocks / 8));
    if (MODE == 3) { const size_t g = (size_t)24 * (i / 8) + (i % 8); put4k(g); put4k(g + 8); put4k(g + 16); }
    if (MODE == 4) { put4k((size_t)3 * i); put4k((size_t)3 * i + 1); put4k((size_t)3 * i + 2); }
    if (MODE == 6) { put4k((size_t)2 * i); put4k((size_t)2 * i + 1); }
    if (MODE == 7) { for (int q = 0; q < 4; ++q) put4k((size_t)4 * i + q); }
    if (MODE == 8) { float4 *b = out + (size_t)i * 128; b[lane] = z; b[64 + lane] = z; }
}

template <int MODE>
void run(const char *name, float4 *out, size_t bytes, size_t per_wave) {
    const int nwaves = (int)(bytes / per_wave);
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) k<MODE><<<nwaves, 64>>>(out, nwaves, (int)(bytes / 4096));
    CHECK(hipEventRecord(a));
    for (int i = 0; i < 20; ++i) k<MODE><<<nwaves, 64>>>(out, nwaves, (int)(bytes / 4096));
    CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
    float ms; CHECK(hipEventElapsedTime(&ms, a, b)); ms /= 20;
    printf("%-62s %8.1f us  %6.2f TB/s\n", name, ms * 1e3, bytes / (ms * 1e-3) / 1e12);
}

int main() {
    const size_t bytes = (size_t)76800 * 12288;  // 943.7 MB = 230400 blocks of 4 KiB
    float4 *out; CHECK(hipMalloc(&out, bytes + (1 << 20)));
    run<0>("4 KiB/wave, block i            (block%8 == XCD)", out, bytes, 4096);
    run<1>("4 KiB/wave, block i^1          (block%8 != XCD)", out, bytes, 4096);
    run<2>("4 KiB/wave, XCD sweeps its own eighth", out, bytes, 4096);
    run<3>("12 KiB/wave = 3 blocks, all with block%8 == XCD", out, bytes, 12288);
    run<4>("12 KiB/wave = 3 consecutive blocks (builder today)", out, bytes, 12288);
    run<6>("8 KiB/wave", out, bytes, 8192);
    run<7>("16 KiB/wave", out, bytes, 16384);
    run<8>("2 KiB/wave", out, bytes, 2048);
    return 0;
}
