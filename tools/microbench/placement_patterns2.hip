// Microbenchmark 9 (round 3): write patterns in which every WAVE owns exactly one aligned 4 KiB block, while the
// builder's work unit (front end + reduce) stays one 12 KiB / 128-pixel unit.  Round 2 (placement_patterns.hip) found
// that one wave per aligned 4 KiB block is indifferent to where a 900 MiB tensor lies physically and that every further
// block the same wave writes costs ~5 us on a slow placement.  Here: multi-wave workgroups per unit with a simulated
// producer phase, and allocations built with the HIP virtual-memory API.
//   memset  : hipMemsetAsync
//   tile12  : one wave per 12 KiB tile, 19 waves / CU (round 2's builder footprint)
//   tile4   : one wave per 4 KiB tile, non-temporal            tile4t: temporal stores
//   tile4d  : tile4 with a ~3.4 us pause before the stores (a front end per 4 KiB sub-unit)
//   wg3     : 3-wave workgroup per 12 KiB unit, wave w stores block w straight from registers, LDS for 10 groups / CU
//   wg3d    : wg3, wave 0 pauses ~3.4 us (the front end) before the group's barrier
//   wg3l    : wg3d with the data going through the group's 12 KiB LDS tile (zero fill by all waves, barrier, wave 0
//             pauses, barrier, every wave reads its block from LDS and stores it)
//   wg3lt   : wg3l with temporal stores
//   wg6l    : 6-wave workgroup per two units (24 KiB), producers = waves 0 and 3
//   wg4p    : 4-wave workgroup: wave 0 only produces (pause), waves 1..3 store one block each
// hipcc --offload-arch=gfx950 -O3 -o placement_patterns2 placement_patterns2.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

typedef float nt4 __attribute__((ext_vector_type(4)));

template <int NV, int NT, int SLEEP>
__global__ __launch_bounds__(64) void k_tilev(nt4 *__restrict__ out, int n) {
    extern __shared__ float4 lds[];
    const int i = blockIdx.x;
    const int t = (i % 8) * (n / 8) + i / 8;
    nt4 z = {1.f, 2.f, 3.f, 4.f};
    if (out == nullptr) { const float4 l = lds[threadIdx.x]; z.x = l.x; }
    nt4 *b = out + (size_t)t * NV * 64 + threadIdx.x;
    if (SLEEP) __builtin_amdgcn_s_sleep(SLEEP);
#pragma unroll
    for (int q = 0; q < NV; ++q) { if (NT) __builtin_nontemporal_store(z, b + q * 64); else b[q * 64] = z; }
}

// NW-wave workgroup per NW / 3 units; wave w owns the 4 KiB block w of the group's NW * 4 KiB.
// MODE bit 0: producer pause in waves 0, 3, ...; bit 1: through LDS; bit 2: temporal stores; bit 3: linear order (no XCD map)
template <int NW, int MODE>
__global__ __launch_bounds__(NW * 64) void k_wg(nt4 *__restrict__ out, int n) {
    extern __shared__ float4 lds[];
    const int i = blockIdx.x, w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int g = (MODE & 8) ? i : (i % 8) * (n / 8) + i / 8;
    nt4 *b = out + ((size_t)g * NW + w) * 256 + lane;
    nt4 v[4];
    if (MODE & 2) {
        float4 *mine = lds + w * 256 + lane;
#pragma unroll
        for (int q = 0; q < 4; ++q) mine[q * 64] = make_float4(0.f, 0.f, 0.f, 0.f);
        __syncthreads();
        if ((MODE & 1) && (w % 3) == 0) {
            __builtin_amdgcn_s_sleep(127);
            lds[(lane * 37) % (NW * 256)] = make_float4(1.f, 2.f, 3.f, 4.f);   // the reduced pixels land somewhere in the tile
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; ++q) { const float4 l = mine[q * 64]; v[q] = nt4{l.x, l.y, l.z, l.w}; }
    } else {
        nt4 z = {1.f, 2.f, 3.f, 4.f};
        if (out == nullptr) { const float4 l = lds[threadIdx.x]; z.x = l.x; }
        if (MODE & 1) {
            if ((w % 3) == 0) __builtin_amdgcn_s_sleep(127);
            __syncthreads();
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = z;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) { if (MODE & 4) b[q * 64] = v[q]; else __builtin_nontemporal_store(v[q], b + q * 64); }
}

// 4-wave workgroup per unit: wave 0 produces only, waves 1..3 store block w - 1 (through LDS)
template <int NT>
__global__ __launch_bounds__(256) void k_wg4p(nt4 *__restrict__ out, int n) {
    extern __shared__ float4 lds[];
    const int i = blockIdx.x, w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int g = (i % 8) * (n / 8) + i / 8;
    if (w > 0) {
        float4 *mine = lds + (w - 1) * 256 + lane;
#pragma unroll
        for (int q = 0; q < 4; ++q) mine[q * 64] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();
    if (w == 0) {
        __builtin_amdgcn_s_sleep(127);
        lds[(lane * 37) % 768] = make_float4(1.f, 2.f, 3.f, 4.f);
    }
    __syncthreads();
    if (w == 0) return;
    float4 *mine = lds + (w - 1) * 256 + lane;
    nt4 *b = out + ((size_t)g * 3 + (w - 1)) * 256 + lane;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float4 l = mine[q * 64];
        const nt4 v = {l.x, l.y, l.z, l.w};
        if (NT) __builtin_nontemporal_store(v, b + q * 64); else b[q * 64] = v;
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

static void *vmm_alloc(size_t bytes, int dev) {
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = dev;
    size_t gran = 0;
    if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended) != hipSuccess || gran == 0) return nullptr;
    const size_t sz = (bytes + gran - 1) / gran * gran;
    hipMemGenericAllocationHandle_t h;
    if (hipMemCreate(&h, sz, &prop, 0) != hipSuccess) return nullptr;
    void *p = nullptr;
    if (hipMemAddressReserve(&p, sz, gran, nullptr, 0) != hipSuccess) return nullptr;
    if (hipMemMap(p, sz, 0, h, 0) != hipSuccess) return nullptr;
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    if (hipMemSetAccess(p, sz, &acc, 1) != hipSuccess) return nullptr;
    static bool said = false;
    if (!said) { printf("# VMM granularity %zu, mapped %zu bytes per buffer\n", gran, sz); said = true; }
    return p;
}

int main(int argc, char **argv) {
    const int nbuf = argc > 1 ? atoi(argv[1]) : 12;
    const int nvmm = argc > 2 ? atoi(argv[2]) : 4;
    const int set = argc > 3 ? atoi(argv[3]) : 0;
    const int n = 76800;                       // 12 KiB units: 32 x 480 x 640 x 12 float64
    const size_t bytes = (size_t)n * 12288;
    const size_t lds19 = 8320, lds10 = 14 * 1024;
    void **bufs = (void **)malloc(sizeof(void *) * (nbuf + nvmm));
    for (int k = 0; k < nbuf; ++k) CHECK(hipMalloc(&bufs[k], bytes));
    int got_vmm = 0;
    for (int k = 0; k < nvmm; ++k) { void *p = vmm_alloc(bytes, 0); if (!p) break; bufs[nbuf + got_vmm++] = p; }
    if (set == 0)
        printf("%-16s %7s %7s %7s %7s %7s %7s %7s %7s %7s %7s %7s %7s\n", "buffer", "memset", "tile12", "tile4", "tile4t", "tile4d",
               "wg3", "wg3d", "wg3l", "wg3lt", "wg6l", "wg4p", "wg4pt");
    else
        printf("%-16s %7s %7s %7s %7s %7s %7s %7s %7s\n", "buffer", "tile12", "wg3l@10", "wg3l@8", "wg3l@6", "wg3l@5", "wg3llin", "wg6l@5", "wg3l0");
    for (int k = 0; k < nbuf + got_vmm; ++k) {
        nt4 *o = (nt4 *)bufs[k];
        char name[32];
        snprintf(name, sizeof(name), "%s%p", k >= nbuf ? "V" : "", bufs[k]);
        if (set == 0) {
            const float m = timed([&] { CHECK(hipMemsetAsync(o, 0, bytes)); });
            const float a = timed([&] { k_tilev<12, 1, 0><<<n, 64, lds19>>>(o, n); });
            const float t4 = timed([&] { k_tilev<4, 1, 0><<<n * 3, 64, 0>>>(o, n * 3); });
            const float t4t = timed([&] { k_tilev<4, 0, 0><<<n * 3, 64, 0>>>(o, n * 3); });
            const float t4d = timed([&] { k_tilev<4, 1, 127><<<n * 3, 64, 0>>>(o, n * 3); });
            const float w3 = timed([&] { k_wg<3, 0><<<n, 192, lds10>>>(o, n); });
            const float w3d = timed([&] { k_wg<3, 1><<<n, 192, lds10>>>(o, n); });
            const float w3l = timed([&] { k_wg<3, 3><<<n, 192, lds10>>>(o, n); });
            const float w3lt = timed([&] { k_wg<3, 7><<<n, 192, lds10>>>(o, n); });
            const float w6l = timed([&] { k_wg<6, 3><<<n / 2, 384, 2 * lds10>>>(o, n / 2); });
            const float w4p = timed([&] { k_wg4p<1><<<n, 256, lds10>>>(o, n); });
            const float w4pt = timed([&] { k_wg4p<0><<<n, 256, lds10>>>(o, n); });
            printf("%-16s %7.1f %7.1f %7.1f %7.1f %7.1f %7.1f %7.1f %7.1f %7.1f %7.1f %7.1f %7.1f\n", name, m, a, t4, t4t, t4d, w3, w3d,
                   w3l, w3lt, w6l, w4p, w4pt);
        } else {
            // occupancy series of the LDS-staged 3-wave group: 10 / 8 / 6 / 5 groups per CU
            const float a = timed([&] { k_tilev<12, 1, 0><<<n, 64, lds19>>>(o, n); });
            const float o10 = timed([&] { k_wg<3, 3><<<n, 192, 14 * 1024>>>(o, n); });
            const float o8 = timed([&] { k_wg<3, 3><<<n, 192, 19 * 1024>>>(o, n); });
            const float o6 = timed([&] { k_wg<3, 3><<<n, 192, 26 * 1024>>>(o, n); });
            const float o5 = timed([&] { k_wg<3, 3><<<n, 192, 31 * 1024>>>(o, n); });
            const float lin = timed([&] { k_wg<3, 3 | 8><<<n, 192, 14 * 1024>>>(o, n); });
            const float w6 = timed([&] { k_wg<6, 3><<<n / 2, 384, 31 * 1024>>>(o, n / 2); });
            const float w0 = timed([&] { k_wg<3, 2><<<n, 192, 14 * 1024>>>(o, n); });
            printf("%-16s %7.1f %7.1f %7.1f %7.1f %7.1f %7.1f %7.1f %7.1f\n", name, a, o10, o8, o6, o5, lin, w6, w0);
        }
    }
    return 0;
}
