// Microbenchmark 8 (round 2): which WRITE PATTERNS are sensitive to where a 900 MiB tensor lies?
// NBUF allocations alive at once; every pattern is timed into every one of them.
//   fill      : linear sweep, 256-thread blocks, 16 B per lane, 4 stores per thread (what a torch fill does)
//   tile12    : one wave per 12 KiB tile (the builder's footprint), XCD-contiguous eighths, non-temporal stores,
//               LDS sized for 19 waves per CU
//   tile12t   : the same with ordinary (temporal) stores
//   tile12lin : one wave per 12 KiB tile, linear unit order (no XCD mapping)
//   tile12d   : tile12 with a ~3.4 us delay before the stores (the builder's load + reduce phase)
//   tile6x2   : tile12 written as two 6 KiB halves with a ~1.7 us delay before each (the part tiles)
// hipcc --offload-arch=gfx950 -O3 -o placement_patterns placement_patterns.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

typedef float nt4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k_fill(float4 *__restrict__ out, size_t nvec) {
    const size_t base = (size_t)blockIdx.x * 1024 + threadIdx.x;
    const float4 z = make_float4(1.f, 2.f, 3.f, 4.f);
#pragma unroll
    for (int q = 0; q < 4; ++q) if (base + q * 256 < nvec) out[base + q * 256] = z;
}

// MODE bit 0: XCD mapping, bit 1: non-temporal, bit 2: delay before, bit 3: two halves with a delay before each
template <int MODE>
__global__ __launch_bounds__(64) void k_tile(nt4 *__restrict__ out, int n) {
    extern __shared__ float4 lds[];
    const int i = blockIdx.x;
    const int t = (MODE & 1) ? (i % 8) * (n / 8) + i / 8 : i;
    nt4 z = {1.f, 2.f, 3.f, 4.f};
    if (out == nullptr) { const float4 l = lds[threadIdx.x]; z.x = l.x; }
    nt4 *b = out + (size_t)t * 12 * 64 + threadIdx.x;
    if (MODE & 4) __builtin_amdgcn_s_sleep(127);
    if (MODE & 8) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            __builtin_amdgcn_s_sleep(64);
#pragma unroll
            for (int q = 0; q < 6; ++q) {
                if (MODE & 2) __builtin_nontemporal_store(z, b + (h * 6 + q) * 64); else b[(h * 6 + q) * 64] = z;
            }
        }
    } else {
#pragma unroll
        for (int q = 0; q < 12; ++q) {
            if (MODE & 2) __builtin_nontemporal_store(z, b + q * 64); else b[q * 64] = z;
        }
    }
}

// 256-thread workgroups: four waves, each its own 12 KiB tile (adjacent tiles), XCD-contiguous eighths of workgroups
__global__ __launch_bounds__(256) void k_tile4(nt4 *__restrict__ out, int ngroups) {
    extern __shared__ float4 lds[];
    const int i = blockIdx.x, w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int t = ((i % 8) * (ngroups / 8) + i / 8) * 4 + w;
    nt4 z = {1.f, 2.f, 3.f, 4.f};
    if (out == nullptr) { const float4 l = lds[threadIdx.x]; z.x = l.x; }
    nt4 *b = out + (size_t)t * 12 * 64 + lane;
#pragma unroll
    for (int q = 0; q < 12; ++q) __builtin_nontemporal_store(z, b + q * 64);
}
// 256-thread workgroups writing 48 KiB with the four waves interleaved KiB by KiB (a fill's footprint), XCD eighths
template <int NT>
__global__ __launch_bounds__(256) void k_inter(nt4 *__restrict__ out, int ngroups) {
    extern __shared__ float4 lds[];
    const int i = blockIdx.x;
    const int t = (i % 8) * (ngroups / 8) + i / 8;
    nt4 z = {1.f, 2.f, 3.f, 4.f};
    if (out == nullptr) { const float4 l = lds[threadIdx.x]; z.x = l.x; }
    nt4 *b = out + (size_t)t * 48 * 64 + threadIdx.x;
#pragma unroll
    for (int q = 0; q < 12; ++q) { if (NT) __builtin_nontemporal_store(z, b + q * 256); else b[q * 256] = z; }
}
// one wave per tile of NV KiB, XCD eighths, non-temporal
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

// one wave per 12 KiB tile written as 12 / NB bursts of NB KiB with a short sleep before each; PERM: the KiB pieces in a
// scattered order (piece (q * 5) % 12) instead of ascending
template <int NB, int PERM>
__global__ __launch_bounds__(64) void k_burst(nt4 *__restrict__ out, int n) {
    extern __shared__ float4 lds[];
    const int i = blockIdx.x;
    const int t = (i % 8) * (n / 8) + i / 8;
    nt4 z = {1.f, 2.f, 3.f, 4.f};
    if (out == nullptr) { const float4 l = lds[threadIdx.x]; z.x = l.x; }
    nt4 *b = out + (size_t)t * 12 * 64 + threadIdx.x;
#pragma unroll
    for (int h = 0; h < 12 / NB; ++h) {
        if (NB < 12) __builtin_amdgcn_s_sleep(127 * NB / 12 + 1);
#pragma unroll
        for (int q = 0; q < NB; ++q) {
            const int piece = PERM ? ((h * NB + q) * 5) % 12 : h * NB + q;
            __builtin_nontemporal_store(z, b + piece * 64);
        }
    }
}

// one wave per 12 KiB tile, its three 4 KiB blocks written in an order rotated by the tile index (ROT) / by the CU-local
// wave order is not controllable; BLK4: 4-wave workgroups, wave w writes the 4 KiB blocks w, w + 4, w + 8 of the group's 48 KiB
template <int BLK4>
__global__ __launch_bounds__(256) void k_rot(nt4 *__restrict__ out, int n) {
    extern __shared__ float4 lds[];
    nt4 z = {1.f, 2.f, 3.f, 4.f};
    if (out == nullptr) { const float4 l = lds[threadIdx.x]; z.x = l.x; }
    if (BLK4) {
        const int i = blockIdx.x, w = threadIdx.x >> 6, lane = threadIdx.x & 63;
        const int g = (i % 8) * (n / 8) + i / 8;            // n = workgroups
        nt4 *b = out + (size_t)g * 48 * 64 + lane;
#pragma unroll
        for (int h = 0; h < 3; ++h)
#pragma unroll
            for (int q = 0; q < 4; ++q) __builtin_nontemporal_store(z, b + ((w + 4 * h) * 4 + q) * 64);
    } else {
        const int i = blockIdx.x;
        const int t = (i % 8) * (n / 8) + i / 8;
        nt4 *b = out + (size_t)t * 12 * 64 + threadIdx.x;
        const int r = (i / 8) % 3;
#pragma unroll
        for (int h = 0; h < 3; ++h) {
            const int blk = (h + r) % 3;
#pragma unroll
            for (int q = 0; q < 4; ++q) __builtin_nontemporal_store(z, b + (blk * 4 + q) * 64);
        }
    }
}

// one wave per 12 KiB tile; TOUCH: the first lanes LOAD one word from each of the tile's three 4 KiB pages when the wave
// starts (translation prefetch), then the wave pauses SLEEP x ~0.03 us (the builder's front end), then stores
template <int TOUCH, int SLEEP>
__global__ __launch_bounds__(64) void k_touch(nt4 *__restrict__ out, int n, int *__restrict__ sink) {
    extern __shared__ float4 lds[];
    const int i = blockIdx.x;
    const int t = (i % 8) * (n / 8) + i / 8;
    nt4 z = {1.f, 2.f, 3.f, 4.f};
    if (out == nullptr) { const float4 l = lds[threadIdx.x]; z.x = l.x; }
    nt4 *b = out + (size_t)t * 12 * 64 + threadIdx.x;
    int got = 0;
    if (TOUCH && threadIdx.x < 3) got = __builtin_nontemporal_load(reinterpret_cast<const int *>(out + (size_t)t * 12 * 64 + threadIdx.x * 256));
    if (SLEEP) __builtin_amdgcn_s_sleep(SLEEP);
    if (TOUCH && got == 0x7fffffff) sink[0] = got;   // keeps the load alive; practically never true
#pragma unroll
    for (int q = 0; q < 12; ++q) __builtin_nontemporal_store(z, b + q * 64);
}

// one wave per 12 KiB tile, stores with an explicit cache policy (gfx942 / gfx950 modifiers)
#define ASM_STORE(MOD) asm volatile("global_store_dwordx4 %0, %1, off " MOD :: "v"(p), "v"(z) : "memory")
template <int POL>
__global__ __launch_bounds__(64) void k_pol(nt4 *__restrict__ out, int n) {
    extern __shared__ float4 lds[];
    const int i = blockIdx.x;
    const int t = (i % 8) * (n / 8) + i / 8;
    nt4 z = {1.f, 2.f, 3.f, 4.f};
    if (out == nullptr) { const float4 l = lds[threadIdx.x]; z.x = l.x; }
    nt4 *b = out + (size_t)t * 12 * 64 + threadIdx.x;
#pragma unroll
    for (int q = 0; q < 12; ++q) {
        nt4 *p = b + q * 64;
        if (POL == 0) ASM_STORE("");
        else if (POL == 1) ASM_STORE("nt");
        else if (POL == 2) ASM_STORE("sc0");
        else if (POL == 3) ASM_STORE("sc1");
        else if (POL == 4) ASM_STORE("sc0 sc1");
        else if (POL == 5) ASM_STORE("sc0 nt");
        else if (POL == 6) ASM_STORE("sc1 nt");
        else ASM_STORE("sc0 sc1 nt");
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
    const int nbuf = argc > 1 ? atoi(argv[1]) : 12;
    const int n = 76800;                       // 12 KiB tiles: 32 x 480 x 640 x 12 float64
    const size_t bytes = (size_t)n * 12288;
    const size_t lds = 8320;                   // 19 waves per CU, as the float64 12-channel builder
    const bool more = argc > 2 && argv[2][0] == 'm';
    const bool burst = argc > 2 && argv[2][0] == 'b';
    const bool sizes = argc > 2 && argv[2][0] == 's';
    const bool rot = argc > 2 && argv[2][0] == 'r';
    const bool touch = argc > 2 && argv[2][0] == 't';
    const bool pol = argc > 2 && argv[2][0] == 'p';
    if (pol) printf("%-14s %8s %8s %8s %8s %8s %8s %8s %8s\n", "buffer", "plain", "nt", "sc0", "sc1", "sc0sc1", "sc0nt", "sc1nt", "sc0sc1nt");
    int *sink; CHECK(hipMalloc(&sink, 64));
    if (touch) printf("%-14s %8s %8s %8s %8s %8s\n", "buffer", "tile12", "sleep", "touch", "touch+sl", "tile4");
    if (rot) printf("%-14s %8s %8s %8s %8s\n", "buffer", "tile12", "tile4", "rot3", "blk4");
    if (sizes) printf("%-14s %8s %8s %8s %8s %8s %8s %8s %8s\n", "buffer", "t1", "t2", "t3", "t4", "t6", "t8", "t12", "t6@19w");
    if (burst) printf("%-14s %8s %8s %8s %8s %8s %8s %8s\n", "buffer", "memset", "tile12", "b4x3", "b3x4", "b2x6", "b1x12", "perm12");
    if (more) printf("%-14s %8s %8s %8s %8s %8s %8s %8s %8s %8s\n", "buffer", "memset", "tile12", "tile12x4", "inter48", "inter48t", "tile4", "tile24", "t12occ32", "t12occ10");
    else if (!burst && !sizes && !rot && !touch && !pol) printf("%-14s %8s %8s %8s %8s %8s %8s\n", "buffer", "fill", "tile12", "tile12t", "tile12lin", "tile12d", "tile6x2");
    void **bufs = (void **)malloc(sizeof(void *) * nbuf);
    for (int k = 0; k < nbuf; ++k) CHECK(hipMalloc(&bufs[k], bytes));
    for (int k = 0; k < nbuf; ++k) {
        nt4 *o = (nt4 *)bufs[k];
        const size_t nvec = bytes / 16;
        if (pol) {
            const float p0 = timed([&] { k_pol<0><<<n, 64, lds>>>(o, n); });
            const float p1 = timed([&] { k_pol<1><<<n, 64, lds>>>(o, n); });
            const float p2 = timed([&] { k_pol<2><<<n, 64, lds>>>(o, n); });
            const float p3 = timed([&] { k_pol<3><<<n, 64, lds>>>(o, n); });
            const float p4 = timed([&] { k_pol<4><<<n, 64, lds>>>(o, n); });
            const float p5 = timed([&] { k_pol<5><<<n, 64, lds>>>(o, n); });
            const float p6 = timed([&] { k_pol<6><<<n, 64, lds>>>(o, n); });
            const float p7 = timed([&] { k_pol<7><<<n, 64, lds>>>(o, n); });
            printf("%p %8.1f %8.1f %8.1f %8.1f %8.1f %8.1f %8.1f %8.1f\n", bufs[k], p0, p1, p2, p3, p4, p5, p6, p7);
            continue;
        }
        if (touch) {
            const float a = timed([&] { k_tile<1 | 2><<<n, 64, lds>>>(o, n); });
            const float sl = timed([&] { k_touch<0, 100><<<n, 64, lds>>>(o, n, sink); });
            const float tc = timed([&] { k_touch<1, 0><<<n, 64, lds>>>(o, n, sink); });
            const float ts = timed([&] { k_touch<1, 100><<<n, 64, lds>>>(o, n, sink); });
            const float t4 = timed([&] { k_tilev<4><<<n * 3, 64, 0>>>(o, n * 3); });
            printf("%p %8.1f %8.1f %8.1f %8.1f %8.1f\n", bufs[k], a, sl, tc, ts, t4);
            continue;
        }
        if (rot) {
            const float a = timed([&] { k_tile<1 | 2><<<n, 64, lds>>>(o, n); });
            const float t4 = timed([&] { k_tilev<4><<<n * 3, 64, 0>>>(o, n * 3); });
            const float r3 = timed([&] { k_rot<0><<<n, 64, lds>>>(o, n); });
            const float b4 = timed([&] { k_rot<1><<<n / 4, 256, 4 * lds>>>(o, n / 4); });
            printf("%p %8.1f %8.1f %8.1f %8.1f\n", bufs[k], a, t4, r3, b4);
            continue;
        }
        if (sizes) {   // LDS per wave scaled with the tile so that bytes in flight per CU stay those of 19 x 12 KiB, up to the 32-wave limit
            const float t1 = timed([&] { k_tilev<1><<<n * 12, 64, 0>>>(o, n * 12); });
            const float t2 = timed([&] { k_tilev<2><<<n * 6, 64, 0>>>(o, n * 6); });
            const float t3 = timed([&] { k_tilev<3><<<n * 4, 64, 0>>>(o, n * 4); });
            const float t4 = timed([&] { k_tilev<4><<<n * 3, 64, 0>>>(o, n * 3); });
            const float t6 = timed([&] { k_tilev<6><<<n * 2, 64, 0>>>(o, n * 2); });
            const float t8 = timed([&] { k_tilev<8><<<n * 3 / 2, 64, lds * 2 / 3>>>(o, n * 3 / 2); });
            const float t12 = timed([&] { k_tilev<12><<<n, 64, lds>>>(o, n); });
            const float t6b = timed([&] { k_tilev<6><<<n * 2, 64, lds>>>(o, n * 2); });
            printf("%p %8.1f %8.1f %8.1f %8.1f %8.1f %8.1f %8.1f %8.1f\n", bufs[k], t1, t2, t3, t4, t6, t8, t12, t6b);
            continue;
        }
        if (burst) {
            const float m = timed([&] { CHECK(hipMemsetAsync(o, 0, bytes)); });
            const float a = timed([&] { k_tile<1 | 2><<<n, 64, lds>>>(o, n); });
            const float b4 = timed([&] { k_burst<4, 0><<<n, 64, lds>>>(o, n); });
            const float b3 = timed([&] { k_burst<3, 0><<<n, 64, lds>>>(o, n); });
            const float b2 = timed([&] { k_burst<2, 0><<<n, 64, lds>>>(o, n); });
            const float b1 = timed([&] { k_burst<1, 0><<<n, 64, lds>>>(o, n); });
            const float pm = timed([&] { k_burst<12, 1><<<n, 64, lds>>>(o, n); });
            printf("%p %8.1f %8.1f %8.1f %8.1f %8.1f %8.1f %8.1f\n", bufs[k], m, a, b4, b3, b2, b1, pm);
            continue;
        }
        if (more) {
            const float m = timed([&] { CHECK(hipMemsetAsync(o, 0, bytes)); });
            const float a = timed([&] { k_tile<1 | 2><<<n, 64, lds>>>(o, n); });
            const float b4 = timed([&] { k_tile4<<<n / 4, 256, 4 * lds>>>(o, n / 4); });
            const float i1 = timed([&] { k_inter<1><<<n / 4, 256, 4 * lds>>>(o, n / 4); });
            const float i0 = timed([&] { k_inter<0><<<n / 4, 256, 4 * lds>>>(o, n / 4); });
            const float t4 = timed([&] { k_tilev<4><<<n * 3, 64, lds / 3>>>(o, n * 3); });
            const float t24 = timed([&] { k_tilev<24><<<n / 2, 64, lds>>>(o, n / 2); });
            const float o32 = timed([&] { k_tile<1 | 2><<<n, 64, 0>>>(o, n); });
            const float o10 = timed([&] { k_tile<1 | 2><<<n, 64, 16000>>>(o, n); });
            printf("%p %8.1f %8.1f %8.1f %8.1f %8.1f %8.1f %8.1f %8.1f %8.1f\n", bufs[k], m, a, b4, i1, i0, t4, t24, o32, o10);
            continue;
        }
        const float f = timed([&] { k_fill<<<(unsigned)((nvec + 1023) / 1024), 256>>>((float4 *)o, nvec); });
        const float a = timed([&] { k_tile<1 | 2><<<n, 64, lds>>>(o, n); });
        const float b = timed([&] { k_tile<1><<<n, 64, lds>>>(o, n); });
        const float c = timed([&] { k_tile<2><<<n, 64, lds>>>(o, n); });
        const float d = timed([&] { k_tile<1 | 2 | 4><<<n, 64, lds>>>(o, n); });
        const float e = timed([&] { k_tile<1 | 2 | 8><<<n, 64, lds>>>(o, n); });
        printf("%p %8.1f %8.1f %8.1f %8.1f %8.1f %8.1f\n", bufs[k], f, a, b, c, d, e);
    }
    return 0;
}
