// Microbenchmark: issue cadence of v_mfma_f32_32x32x2_f32 as a function of (independent accumulators per wave,
// waves per SIMD).  Build: hipcc --offload-arch=gfx950 -O3 mfma_f32_chain.hip -o mfma_f32_chain
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
using f32x16 = __attribute__((ext_vector_type(16))) float;

template <int NACC>
__global__ __launch_bounds__(256) void k(float *out, int iters, long long *cyc) {
    f32x16 acc[NACC];
    for (int a = 0; a < NACC; ++a)
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.0f;
    float x = 1.0f + threadIdx.x * 1e-9f, y = 0.5f;
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[a], 0, 0, 0);
    }
    long long t1 = __builtin_readcyclecounter();
    float s = 0.0f;
    for (int a = 0; a < NACC; ++a)
        for (int r = 0; r < 16; ++r) s += acc[a][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int NACC>
void run(int wgs_per_cu) {
    const int iters = 2000, blocks = 256 * wgs_per_cu;
    float *out; long long *cyc;
    hipMalloc(&out, sizeof(float) * blocks * 256);
    hipMalloc(&cyc, 8);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<NACC><<<blocks, 256>>>(out, iters, cyc);
    hipDeviceSynchronize();
    hipEventRecord(a);
    k<NACC><<<blocks, 256>>>(out, iters, cyc);
    hipEventRecord(b);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, a, b);
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double n_mfma_per_simd = (double)iters * NACC * wgs_per_cu;   // one wave of each WG per SIMD
    const double flops = (double)blocks * 4 * iters * NACC * 4096.0;
    printf("acc/wave %d  waves/SIMD %d : %.1f us, %.1f TF, wave-cycles per MFMA (s_memtime of wave 0) %.1f, us per MFMA per SIMD %.4f\n",
           NACC, wgs_per_cu, ms * 1e3, flops / (ms * 1e-3) / 1e12, (double)c / (iters * NACC), ms * 1e3 / n_mfma_per_simd);
    hipFree(out); hipFree(cyc);
}

int main() {
    run<1>(1); run<2>(1); run<4>(1);
    run<1>(2); run<2>(2); run<4>(2);
    run<1>(3); run<2>(3);
    return 0;
}
