// Microbenchmark 11 (round 3): PACED stores.  placement_patterns3 showed that what decides the time of a 900 MiB write on a
// "slow" placement is not which wave owns which 4 KiB block but how much store traffic the resident waves OFFER: a
// kernel whose waves pause ~3.4 us before storing (offered load ~ the memory's service rate) writes the tensor in
// 131 us on every placement, the same kernel with shorter pauses collapses to 155-170 us on a slow one.
// Here: one wave per 12 KiB unit (the builder's footprint, 19 waves per CU), every wave HOLDS its slot for at least
// L x 10 ns (s_memrealtime, 100 MHz), so the offered load is 19 x 12 KiB x 256 CUs / L whatever the front end took.
//   hold-before : wait until L has passed, then store (few waves in the store phase at any time)
//   hold-after  : store, then wait until L has passed
//   jitter      : a pseudo-random front end of 0 .. 3.4 us first (the builder's units differ)
//   two-part    : the 12 KiB written as two 6 KiB halves with ~0.4 us between them (the builder's part tiles)
// hipcc --offload-arch=gfx950 -O3 -o placement_patterns4 placement_patterns4.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

typedef float nt4 __attribute__((ext_vector_type(4)));

__device__ inline void hold_until(unsigned long long t0, int ticks) {
    while ((long long)(wall_clock64() - t0) < (long long)ticks) __builtin_amdgcn_s_sleep(4);
}

// MODE bit 0: hold AFTER the stores instead of before; bit 1: jittered front end; bit 2: two halves
template <int MODE>
__global__ __launch_bounds__(64) void k_paced(nt4 *__restrict__ out, int n, int ticks) {
    extern __shared__ float4 lds[];
    const unsigned long long t0 = wall_clock64();
    const int i = blockIdx.x;
    const int t = (i % 8) * (n / 8) + i / 8;
    nt4 z = {1.f, 2.f, 3.f, 4.f};
    if (out == nullptr) { const float4 l = lds[threadIdx.x]; z.x = l.x; }
    nt4 *b = out + (size_t)t * 768 + threadIdx.x;
    if (MODE & 2) {
        const unsigned h = (unsigned)i * 2654435761u;
        hold_until(t0, (int)((h >> 16) % 340u));   // 0 .. 3.4 us
    }
    if (!(MODE & 1)) hold_until(t0, ticks);
    if (MODE & 4) {
#pragma unroll
        for (int q = 0; q < 6; ++q) __builtin_nontemporal_store(z, b + q * 64);
        __builtin_amdgcn_s_sleep(16);
#pragma unroll
        for (int q = 6; q < 12; ++q) __builtin_nontemporal_store(z, b + q * 64);
    } else {
#pragma unroll
        for (int q = 0; q < 12; ++q) __builtin_nontemporal_store(z, b + q * 64);
    }
    if (MODE & 1) hold_until(t0, ticks);
}

template <typename F>
static float timed(F launch) {
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    launch(); launch();
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipEventRecord(a));
        for (int i = 0; i < 10; ++i) launch();
        CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
        float ms; CHECK(hipEventElapsedTime(&ms, a, b));
        if (ms / 10 < best) best = ms / 10;
    }
    CHECK(hipEventDestroy(a)); CHECK(hipEventDestroy(b));
    return best * 1e3f;
}

int main(int argc, char **argv) {
    const int nbuf = argc > 1 ? atoi(argv[1]) : 8;
    const int set = argc > 2 ? atoi(argv[2]) : 0;
    const int n = 76800;
    const size_t bytes = (size_t)n * 12288;
    const size_t lds = 8320;   // 19 waves per CU
    void **bufs = (void **)malloc(sizeof(void *) * nbuf);
    for (int k = 0; k < nbuf; ++k) CHECK(hipMalloc(&bufs[k], bytes));
    int L[10] = {0, 600, 700, 750, 780, 800, 820, 850, 900, 1000};
    if (argc > 3) for (int j = 0; j < 10; ++j) L[j] = atoi(argv[3]) + j * (argc > 4 ? atoi(argv[4]) : 10);
    const size_t ldsv = argc > 5 ? (size_t)atoi(argv[5]) : lds;
    if (set == 0) printf("# hold-before, plain front end; columns = L in 10 ns ticks (offered TB/s = 5977 / L)\n");
    if (set == 1) printf("# hold-before, jittered front end + two-part stores\n");
    if (set == 2) printf("# hold-AFTER the stores, jittered front end\n");
    printf("%-16s", "buffer");
    for (int j = 0; j < 10; ++j) printf(" %7d", L[j]);
    printf("\n");
    for (int k = 0; k < nbuf; ++k) {
        nt4 *o = (nt4 *)bufs[k];
        printf("%-16p", bufs[k]);
        for (int j = 0; j < 10; ++j) {
            float r;
            if (set == 0) r = timed([&] { k_paced<0><<<n, 64, ldsv>>>(o, n, L[j]); });
            else if (set == 1) r = timed([&] { k_paced<2 | 4><<<n, 64, ldsv>>>(o, n, L[j]); });
            else r = timed([&] { k_paced<1 | 2><<<n, 64, ldsv>>>(o, n, L[j]); });
            printf(" %7.1f", r);
        }
        printf("\n");
    }
    return 0;
}
