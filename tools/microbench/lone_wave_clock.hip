// What a LONE wave gets: the shader clock (s_memtime ticks per 10 ns of s_memrealtime) and the time of a chain of dependent
// VALU / LDS instructions, with 1 wave on the chip and with every CU busy.  (r06: the tail of a clustered builder launch is a few
// lone waves; NOTES.md prices their instructions at ~8 cycles each.)
//   hipcc --offload-arch=gfx950 -O2 -o tools/microbench/lone_wave_clock tools/microbench/lone_wave_clock.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void k(int iters, unsigned long long *out, int report_block) {
    __shared__ unsigned int lds[256];
    lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    const unsigned long long c0 = clock64(), w0 = wall_clock64();
    float a = (float)threadIdx.x;
    unsigned int idx = threadIdx.x;
    // phase 1: dependent float adds (one VALU instruction per step)
    for (int i = 0; i < iters; ++i) { a = a * 1.0000001f + 1.0f; }
    const unsigned long long c1 = clock64(), w1 = wall_clock64();
    // phase 2: dependent LDS round trips (pointer chase)
    for (int i = 0; i < iters; ++i) { idx = lds[idx & 255u]; }
    const unsigned long long c2 = clock64(), w2 = wall_clock64();
    // phase 3: dependent returning LDS atomics
    for (int i = 0; i < iters; ++i) { idx = atomicMin(&lds[(idx + i) & 255u], idx) & 255u; }
    const unsigned long long c3 = clock64(), w3 = wall_clock64();
    if (blockIdx.x == report_block && threadIdx.x == 0) {
        out[0] = c1 - c0; out[1] = w1 - w0; out[2] = c2 - c1; out[3] = w2 - w1; out[4] = c3 - c2; out[5] = w3 - w2;
        out[6] = (unsigned long long)(a != 0.0f) + idx;
    }
}
int main() {
    unsigned long long *d, h[8];
    hipMalloc(&d, 64);
    const int iters = 20000;
    for (int blocks : {1, 256, 256 * 8, 256 * 32}) {
        for (int rep = 0; rep < 3; ++rep) k<<<blocks, 64>>>(iters, d, blocks - 1);
        hipDeviceSynchronize();
        hipMemcpy(h, d, 56, hipMemcpyDeviceToHost);
        printf("%6d one-wave workgroups: shader clock %.0f MHz | dependent v_fma: %.1f ns = %.1f cycles | LDS chase: %.1f ns = %.1f cycles | returning LDS atomic: %.1f ns = %.1f cycles\n",
               blocks, 100.0 * h[0] / h[1], 10.0 * h[1] / iters, (double)h[0] / iters, 10.0 * h[3] / iters, (double)h[2] / iters, 10.0 * h[5] / iters, (double)h[4] / iters);
    }
    return 0;
}
