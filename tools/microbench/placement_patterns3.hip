// Microbenchmark 10 (round 3): follow-up of placement_patterns2 -- how the 3-wave-per-unit store pattern depends on the
// producer's time, and multi-unit workgroups in which no wave idles.  All kernels write 76 800 units of 12 KiB.
//   l@S      : 3-wave group per unit through a 12 KiB LDS tile, wave 0 pauses S x 64 cycles before the barrier
//   p3@S     : the same, but all three waves pause S / 3 (a front end split over the waves)
//   wg2l     : 2-wave group per unit, 6 KiB per wave
//   u3own    : 3-wave group, 3 units: every wave pauses (its own front end), barrier, wave w stores unit w (12 KiB)
//   u3blk    : ... wave w stores block w of unit 0, then of unit 1, then of unit 2
//   u3blkb   : ... with a barrier between the three rounds
// hipcc --offload-arch=gfx950 -O3 -o placement_patterns3 placement_patterns3.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

typedef float nt4 __attribute__((ext_vector_type(4)));

template <int NV>
__global__ __launch_bounds__(64) void k_tilev(nt4 *__restrict__ out, int n) {
    extern __shared__ float4 lds[];
    const int i = blockIdx.x;
    const int t = (i % 8) * (n / 8) + i / 8;
    nt4 z = {1.f, 2.f, 3.f, 4.f};
    if (out == nullptr) { const float4 l = lds[threadIdx.x]; z.x = l.x; }
    nt4 *b = out + (size_t)t * NV * 64 + threadIdx.x;
#pragma unroll
    for (int q = 0; q < NV; ++q) __builtin_nontemporal_store(z, b + q * 64);
}

__device__ inline void pause(int s) {  // s x 64 cycles
    for (; s >= 127; s -= 127) __builtin_amdgcn_s_sleep(127);
    if (s >= 64) { __builtin_amdgcn_s_sleep(64); s -= 64; }
    if (s >= 32) { __builtin_amdgcn_s_sleep(32); s -= 32; }
    if (s >= 16) { __builtin_amdgcn_s_sleep(16); s -= 16; }
    if (s >= 8) { __builtin_amdgcn_s_sleep(8); s -= 8; }
}

// NW waves per unit; ALLP: every wave pauses S / NW instead of wave 0 pausing S; TEMP: temporal stores
template <int NW, int ALLP, int TEMP>
__global__ __launch_bounds__(NW * 64) void k_wg(nt4 *__restrict__ out, int n, int S) {
    extern __shared__ float4 lds[];
    const int i = blockIdx.x, w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int g = (i % 8) * (n / 8) + i / 8;
    constexpr int PER = 12 / NW;   // KiB per wave
    nt4 *b = out + (size_t)g * 768 + w * PER * 64 + lane;
    float4 *mine = lds + w * PER * 64 + lane;
    if (w == 0) {
#pragma unroll
        for (int q = 0; q < 12; ++q) lds[q * 64 + lane] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (ALLP) pause(S / NW); else if (w == 0) pause(S);
    if (w == 0) lds[(lane * 37) % 768] = make_float4(1.f, 2.f, 3.f, 4.f);
    __syncthreads();
#pragma unroll
    for (int q = 0; q < PER; ++q) {
        const float4 l = mine[q * 64];
        const nt4 v = {l.x, l.y, l.z, l.w};
        if (TEMP) b[q * 64] = v; else __builtin_nontemporal_store(v, b + q * 64);
    }
}

// 3 waves, 3 units.  HOW 0: wave w stores unit w; 1: block w of each unit in turn; 2: the same with barriers between
template <int HOW>
__global__ __launch_bounds__(192) void k_u3(nt4 *__restrict__ out, int n, int S) {
    extern __shared__ float4 lds[];
    const int i = blockIdx.x, w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int g = (i % 8) * (n / 8) + i / 8;   // n = groups
    nt4 *base = out + (size_t)g * 3 * 768 + lane;
    nt4 z = {1.f, 2.f, 3.f, 4.f};
    if (out == nullptr) { const float4 l = lds[threadIdx.x]; z.x = l.x; }
    pause(S);
    __syncthreads();
    if (HOW == 0) {
#pragma unroll
        for (int q = 0; q < 12; ++q) __builtin_nontemporal_store(z, base + w * 768 + q * 64);
    } else {
#pragma unroll
        for (int u = 0; u < 3; ++u) {
#pragma unroll
            for (int q = 0; q < 4; ++q) __builtin_nontemporal_store(z, base + u * 768 + (w * 4 + q) * 64);
            if (HOW == 2 && u < 2) __syncthreads();
        }
    }
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
    const size_t lds10 = 14 * 1024;
    void **bufs = (void **)malloc(sizeof(void *) * nbuf);
    for (int k = 0; k < nbuf; ++k) CHECK(hipMalloc(&bufs[k], bytes));
    if (set == 0)
        printf("%-16s %7s %7s %7s %7s %7s %7s %7s %7s %7s %7s %7s\n", "buffer", "tile12", "l@0", "l@32", "l@64", "l@96", "l@127", "l@160",
               "l@190", "l@254", "lt@127", "lt@190");
    else
        printf("%-16s %7s %7s %7s %7s %7s %7s %7s %7s %7s %7s %7s\n", "buffer", "tile12", "p3@127", "p3@190", "p3@254", "wg2l@127", "wg2l@64",
               "u3own", "u3blk", "u3blkb", "u3blk@6", "u3blkb@6");
    for (int k = 0; k < nbuf; ++k) {
        nt4 *o = (nt4 *)bufs[k];
        const float a = timed([&] { k_tilev<12><<<n, 64, 8320>>>(o, n); });
        if (set == 0) {
            float r[10];
            const int S[8] = {0, 32, 64, 96, 127, 160, 190, 254};
            for (int j = 0; j < 8; ++j) r[j] = timed([&] { k_wg<3, 0, 0><<<n, 192, lds10>>>(o, n, S[j]); });
            r[8] = timed([&] { k_wg<3, 0, 1><<<n, 192, lds10>>>(o, n, 127); });
            r[9] = timed([&] { k_wg<3, 0, 1><<<n, 192, lds10>>>(o, n, 190); });
            printf("%-16p %7.1f", bufs[k], a);
            for (int j = 0; j < 10; ++j) printf(" %7.1f", r[j]);
            printf("\n");
        } else {
            const float p1 = timed([&] { k_wg<3, 1, 0><<<n, 192, lds10>>>(o, n, 127); });
            const float p2 = timed([&] { k_wg<3, 1, 0><<<n, 192, lds10>>>(o, n, 190); });
            const float p3 = timed([&] { k_wg<3, 1, 0><<<n, 192, lds10>>>(o, n, 254); });
            const float w2 = timed([&] { k_wg<2, 0, 0><<<n, 128, lds10>>>(o, n, 127); });
            const float w2b = timed([&] { k_wg<2, 0, 0><<<n, 128, lds10>>>(o, n, 64); });
            // 3-unit groups: 25 KiB of LDS (sparse value lists) -> 6 groups = 18 units per CU
            const float u0 = timed([&] { k_u3<0><<<n / 3, 192, 25 * 1024>>>(o, n / 3, 127); });
            const float u1 = timed([&] { k_u3<1><<<n / 3, 192, 25 * 1024>>>(o, n / 3, 127); });
            const float u2 = timed([&] { k_u3<2><<<n / 3, 192, 25 * 1024>>>(o, n / 3, 127); });
            // a slower front end (~5 us)
            const float u1s = timed([&] { k_u3<1><<<n / 3, 192, 25 * 1024>>>(o, n / 3, 190); });
            const float u2s = timed([&] { k_u3<2><<<n / 3, 192, 25 * 1024>>>(o, n / 3, 190); });
            printf("%-16p %7.1f %7.1f %7.1f %7.1f %7.1f %7.1f %7.1f %7.1f %7.1f %7.1f %7.1f\n", bufs[k], a, p1, p2, p3, w2, w2b, u0, u1, u2,
                   u1s, u2s);
        }
    }
    return 0;
}
