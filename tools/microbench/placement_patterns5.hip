// Microbenchmark 12 (round 3): closed-loop control of the store phase instead of open-loop pacing.  Workgroups of NW
// independent waves (one 12 KiB unit each, no barrier between them); the waves of a workgroup take turns in their store
// phase through a ticket counter in LDS, so at most CAP of them are storing at a time -- the number of concurrent store
// bursts per CU is bounded whatever the memory system's state, without a tuned hold time.
//   free        : no control (= tile12 in NW-wave workgroups)
//   issue       : the turn is passed on when the burst has been ISSUED
//   done        : the turn is passed on when the burst has COMPLETED (s_waitcnt vmcnt(0))
// Every wave first pauses S x 64 cycles (its front end), jittered +-25 % by the unit index.
// hipcc --offload-arch=gfx950 -O3 -o placement_patterns5 placement_patterns5.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

typedef float nt4 __attribute__((ext_vector_type(4)));

__device__ inline void pause(int s) {  // s x 64 cycles
    for (; s >= 127; s -= 127) __builtin_amdgcn_s_sleep(127);
    if (s >= 64) { __builtin_amdgcn_s_sleep(64); s -= 64; }
    if (s >= 32) { __builtin_amdgcn_s_sleep(32); s -= 32; }
    if (s >= 16) { __builtin_amdgcn_s_sleep(16); s -= 16; }
    if (s >= 8) { __builtin_amdgcn_s_sleep(8); s -= 8; }
}

// MODE 0 free, 1 pass the turn after issue, 2 after completion.  CAP = waves of the group allowed in the store phase.
template <int NW, int MODE, int CAP>
__global__ __launch_bounds__(NW * 64) void k_turns(nt4 *__restrict__ out, int ngroups, int S) {
    extern __shared__ int lds[];
    volatile int *next = lds, *serving = lds + 1;
    const int i = blockIdx.x, w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int g = (i % 8) * (ngroups / 8) + i / 8;
    const int unit = g * NW + w;
    if (MODE) {
        if (threadIdx.x == 0) { lds[0] = 0; lds[1] = 0; }
        __syncthreads();
    }
    nt4 z = {1.f, 2.f, 3.f, 4.f};
    if (out == nullptr) z.x = (float)lds[threadIdx.x + 2];
    const unsigned h = (unsigned)unit * 2654435761u;
    pause(S * 3 / 4 + (int)((h >> 16) % (unsigned)(S / 2 + 1)));
    nt4 *b = out + (size_t)unit * 768 + lane;
    if (MODE) {
        int my = 0;
        if (lane == 0) {
            my = atomicAdd((int *)next, 1);
            while (my - *serving >= CAP) __builtin_amdgcn_s_sleep(2);
        }
        __builtin_amdgcn_wave_barrier();
    }
#pragma unroll
    for (int q = 0; q < 12; ++q) __builtin_nontemporal_store(z, b + q * 64);
    if (MODE == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (MODE && lane == 0) atomicAdd((int *)serving, 1);
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
    const int nbuf = argc > 1 ? atoi(argv[1]) : 6;
    const int S = argc > 2 ? atoi(argv[2]) : 127;     // front end, x 64 cycles (127 = 3.4 us)
    const int n = 76800;
    const size_t bytes = (size_t)n * 12288;
    const size_t per_wave = 8320;                     // LDS of a builder wave
    void **bufs = (void **)malloc(sizeof(void *) * nbuf);
    for (int k = 0; k < nbuf; ++k) CHECK(hipMalloc(&bufs[k], bytes));
    printf("# front end %d x 64 cycles; columns: group size / mode / cap\n", S);
    printf("%-16s %7s %7s %7s %7s %7s %7s %7s %7s %7s %7s %7s %7s\n", "buffer", "1/free", "4/free", "4/iss/1", "4/don/1", "3/iss/1", "3/don/1",
           "6/iss/1", "6/don/1", "6/iss/2", "6/don/2", "8/iss/2", "8/don/2");
    for (int k = 0; k < nbuf; ++k) {
        nt4 *o = (nt4 *)bufs[k];
        float r[12];
        r[0] = timed([&] { k_turns<1, 0, 1><<<n, 64, per_wave>>>(o, n, S); });
        r[1] = timed([&] { k_turns<4, 0, 1><<<n / 4, 256, 4 * per_wave>>>(o, n / 4, S); });
        r[2] = timed([&] { k_turns<4, 1, 1><<<n / 4, 256, 4 * per_wave>>>(o, n / 4, S); });
        r[3] = timed([&] { k_turns<4, 2, 1><<<n / 4, 256, 4 * per_wave>>>(o, n / 4, S); });
        r[4] = timed([&] { k_turns<3, 1, 1><<<n / 3, 192, 3 * per_wave>>>(o, n / 3, S); });
        r[5] = timed([&] { k_turns<3, 2, 1><<<n / 3, 192, 3 * per_wave>>>(o, n / 3, S); });
        r[6] = timed([&] { k_turns<6, 1, 1><<<n / 6, 384, 6 * per_wave>>>(o, n / 6, S); });
        r[7] = timed([&] { k_turns<6, 2, 1><<<n / 6, 384, 6 * per_wave>>>(o, n / 6, S); });
        r[8] = timed([&] { k_turns<6, 1, 2><<<n / 6, 384, 6 * per_wave>>>(o, n / 6, S); });
        r[9] = timed([&] { k_turns<6, 2, 2><<<n / 6, 384, 6 * per_wave>>>(o, n / 6, S); });
        r[10] = timed([&] { k_turns<8, 1, 2><<<n / 8, 512, 8 * per_wave>>>(o, n / 8, S); });
        r[11] = timed([&] { k_turns<8, 2, 2><<<n / 8, 512, 8 * per_wave>>>(o, n / 8, S); });
        printf("%-16p", bufs[k]);
        for (int j = 0; j < 12; ++j) printf(" %7.1f", r[j]);
        printf("\n");
    }
    return 0;
}
