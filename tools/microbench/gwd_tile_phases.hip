// Where a k_gwd_tiles workgroup spends its time: the library's tile body (included, not copied) run with wall-clock marks
// around its phases, plus variants of the body under test.  Random augmented clouds (the values only steer the
// exponentials, whose rate is data independent).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I include -I event_representation_study_amd/csrc \
//        tools/microbench/gwd_tile_phases.hip -o tools/microbench/gwd_tile_phases
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "evrep_gwd.hip"

using namespace evrep;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// variant 0: the library body.  1: loads only (no compute).  2: compute only (LDS garbage, no loads).  3: 2 without the
// exponentials.  4: 2 without the MFMA chains.  5: the MFMA chains alone.
template <int NSS, int NST, int VAR>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(5))) void k_var(GwdTileArgs P, long long *marks) {
    extern __shared__ float lds[];
    __shared__ double red[kWaves];
    const int tid = threadIdx.x, tile = blockIdx.x;
    const long long t0 = wall_clock64();
    if (VAR == 0) {
        gwd_tile_body<NSS, NST>(P, tile, lds, tid);
    } else {
        constexpr int KPS = 2 * NSS, KPT = 2 * NST;
        int t = tile, bi = 0;
        const int T = P.T;
        while (t >= T - bi) { t -= T - bi; ++bi; }
        const int bj = bi + t;
        const int64_t i0 = (int64_t)bi * kTile, j0 = (int64_t)bj * kTile;
        constexpr int ROWS = 2 * KPS + 2 * KPT, NV = ROWS * (kTile / 4), NIT = (NV + kThreads - 1) / kThreads;
        if (VAR == 1) {
            float4 v[NIT];
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int e = tid + it * kThreads, row = e / (kTile / 4), c = (e % (kTile / 4)) * 4;
                v[it] = make_float4(0, 0, 0, 0);
                if (e < NV) {
                    const bool s_row = row < 2 * KPS;
                    const int k = s_row ? (row < KPS ? row : row - KPS) : (row < 2 * KPS + KPT ? row - 2 * KPS : row - 2 * KPS - KPT);
                    const bool a_form = s_row ? row < KPS : row < 2 * KPS + KPT;
                    const float *base = s_row ? (a_form ? P.YsA : P.YsB) : (a_form ? P.YtA : P.YtB);
                    v[it] = *reinterpret_cast<const float4 *>(base + (int64_t)k * (s_row ? P.npad : P.mpad) + (a_form ? i0 : j0) + c);
                }
            }
#pragma unroll
            for (int it = 0; it < NIT; ++it) { const int e = tid + it * kThreads; if (e < NV) *reinterpret_cast<float4 *>(lds + 4 * e) = v[it]; }
            __syncthreads();
            if (tid == 0) P.partial[tile] = lds[tile & 127];
        } else {
            const int lane = tid & 63, wave = tid >> 6, r0 = wave * 32;
            float *As = lds, *Bs = lds + KPS * kTile, *At = lds + 2 * KPS * kTile, *Bt = At + KPT * kTile;
            float as[NSS], at[NST];
            gwd_load_strip<NSS>(As, r0, lane, as);
            gwd_load_strip<NST>(At, r0, lane, at);
            float sum = 0.0f;
#pragma unroll 1
            for (int cb = 0; cb < 4; ++cb) {
                float bs[NSS], bt[NST];
                f32x16 es, et;
                gwd_load_strip<NSS>(Bs, cb * 32, lane, bs);
                gwd_load_strip<NST>(Bt, cb * 32, lane, bt);
                if (VAR != 4) { es = gwd_block<NSS>(as, bs); et = gwd_block<NST>(at, bt); }
                else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) { es[r] = as[r % NSS] * bs[(r + 1) % NSS] - (float)r; et[r] = at[r % NST] * bt[(r + 3) % NST] - (float)r; }
                }
                if (VAR == 2 || VAR == 4) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) sum += fabsf(__builtin_amdgcn_exp2f(es[r]) - __builtin_amdgcn_exp2f(et[r]));
                } else if (VAR == 3) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) sum += fabsf(es[r] - et[r]);
                } else {
                    sum += es[0] + et[0];
                }
            }
            double d = (double)sum;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) d += __shfl_xor(d, o, 64);
            if (lane == 0) red[wave] = d;
            __syncthreads();
            if (tid == 0) P.partial[tile] = red[0] + red[1] + red[2] + red[3];
        }
    }
    const long long t1 = wall_clock64();
    if (tid == 0 && marks) marks[tile] = t1 - t0;
}

// Compute-loop candidates (LDS garbage in, speed only): NORMV = the two norm-carrying MFMA steps replaced by VALU adds
// into the accumulators (9 instead of 11 MFMA per block); NPOLY of a block's 32 exponentials as a degree-5 polynomial
// on the plain VALU (which overlaps the matrix pipe; v_exp_f32 does not).
__device__ inline float exp2_poly(float x) {
    const float r = __builtin_rintf(x), f = x - r;
    float p = 1.535336188319500e-4f;
    p = __builtin_fmaf(p, f, 1.339887440266574e-3f);
    p = __builtin_fmaf(p, f, 9.618437357674640e-3f);
    p = __builtin_fmaf(p, f, 5.550332471162809e-2f);
    p = __builtin_fmaf(p, f, 2.402264791363012e-1f);
    p = __builtin_fmaf(p, f, 6.931472028550421e-1f);
    p = __builtin_fmaf(p, f, 1.0f);
    return __builtin_ldexpf(p, (int)r);
}
template <int NSS, int NST, bool NORMV, int NPOLY>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(5))) void k_cand(GwdTileArgs P, long long *marks) {
    extern __shared__ float lds[];
    __shared__ double red[kWaves];
    const int tid = threadIdx.x, tile = blockIdx.x;
    const long long t0 = wall_clock64();
    constexpr int KPS = 2 * NSS, KPT = 2 * NST;
    constexpr int MS = NORMV ? NSS - 1 : NSS, MT = NORMV ? NST - 1 : NST;
    const int lane = tid & 63, wave = tid >> 6, r0 = wave * 32;
    float *As = lds, *Bs = lds + KPS * kTile, *At = lds + 2 * KPS * kTile, *Bt = At + KPT * kTile;
    float as[MS], at[MT];
    gwd_load_strip<MS>(As, r0, lane, as);
    gwd_load_strip<MT>(At, r0, lane, at);
    float ns[16], nt[16];
    if (NORMV) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = r0 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            ns[r] = As[(KPS - 2) * kTile + row]; nt[r] = At[(KPT - 2) * kTile + row];
        }
    }
    float sum = 0.0f;
#pragma unroll 1
    for (int cb = 0; cb < 4; ++cb) {
        float bs[MS], bt[MT];
        f32x16 es, et;
        gwd_load_strip<MS>(Bs, cb * 32, lane, bs);
        gwd_load_strip<MT>(Bt, cb * 32, lane, bt);
        if (NORMV) {
            const float cs = Bs[(KPS - 1) * kTile + cb * 32 + (lane & 31)], ct = Bt[(KPT - 1) * kTile + cb * 32 + (lane & 31)];
#pragma unroll
            for (int r = 0; r < 16; ++r) { es[r] = ns[r] + cs; et[r] = nt[r] + ct; }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) { es[r] = 0.0f; et[r] = 0.0f; }
        }
#pragma unroll
        for (int s_ = 0; s_ < MS; ++s_) es = __builtin_amdgcn_mfma_f32_32x32x2f32(as[s_], bs[s_], es, 0, 0, 0);
#pragma unroll
        for (int s_ = 0; s_ < MT; ++s_) et = __builtin_amdgcn_mfma_f32_32x32x2f32(at[s_], bt[s_], et, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float a = (2 * r < NPOLY) ? exp2_poly(es[r]) : __builtin_amdgcn_exp2f(es[r]);
            const float b = (2 * r + 1 < NPOLY) ? exp2_poly(et[r]) : __builtin_amdgcn_exp2f(et[r]);
            sum += fabsf(a - b);
        }
    }
    double d = (double)sum;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) d += __shfl_xor(d, o, 64);
    if (lane == 0) red[wave] = d;
    __syncthreads();
    if (tid == 0) P.partial[tile] = red[0] + red[1] + red[2] + red[3];
    const long long t1 = wall_clock64();
    if (tid == 0 && marks) marks[tile] = t1 - t0;
}

// The bf16 x 3 split form of the same exponent matrix (each float32 coordinate = hi + mid + lo in bfloat16; the six
// largest cross terms per dimension laid along K): MS / MT v_mfma_f32_32x32x16_bf16 per block.  Speed only (garbage in).
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
template <int MS, int MT, bool EXP>
__global__ __launch_bounds__(kThreads) void k_bf(GwdTileArgs P, long long *marks) {
    extern __shared__ float lds[];   // B operands: [MS + MT][2 k-halves][128 points] x 16 bytes
    __shared__ double red[kWaves];
    const int tid = threadIdx.x, tile = blockIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long t0 = wall_clock64();
    const bf16x8 *B = reinterpret_cast<const bf16x8 *>(lds);
    bf16x8 as[MS], at[MT];
#pragma unroll
    for (int s_ = 0; s_ < MS; ++s_) as[s_] = B[(s_ * 2 + (lane >> 5)) * kTile + wave * 32 + (lane & 31)];
#pragma unroll
    for (int s_ = 0; s_ < MT; ++s_) at[s_] = B[((MS + s_) * 2 + (lane >> 5)) * kTile + wave * 32 + (lane & 31)];
    float sum = 0.0f;
#pragma unroll 1
    for (int cb = 0; cb < 4; ++cb) {
        f32x16 es, et;
#pragma unroll
        for (int r = 0; r < 16; ++r) { es[r] = 0.0f; et[r] = 0.0f; }
#pragma unroll
        for (int s_ = 0; s_ < MS; ++s_)
            es = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as[s_], B[(s_ * 2 + (lane >> 5)) * kTile + cb * 32 + (lane & 31)], es, 0, 0, 0);
#pragma unroll
        for (int s_ = 0; s_ < MT; ++s_)
            et = __builtin_amdgcn_mfma_f32_32x32x16_bf16(at[s_], B[((MS + s_) * 2 + (lane >> 5)) * kTile + cb * 32 + (lane & 31)], et, 0, 0, 0);
        if (EXP) {
#pragma unroll
            for (int r = 0; r < 16; ++r) sum += fabsf(__builtin_amdgcn_exp2f(es[r]) - __builtin_amdgcn_exp2f(et[r]));
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) sum += fabsf(es[r] - et[r]);
        }
    }
    double d = (double)sum;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) d += __shfl_xor(d, o, 64);
    if (lane == 0) red[wave] = d;
    __syncthreads();
    if (tid == 0) P.partial[tile] = red[0] + red[1] + red[2] + red[3];
    const long long t1 = wall_clock64();
    if (tid == 0 && marks) marks[tile] = t1 - t0;
}

int main(int argc, char **argv) {
    const int64_t n = 12500, m = 14400;
    const int64_t npad = gwd_pad_tile(n), mpad = gwd_pad_tile(m);
    constexpr int NSS = 3, NST = 8;
    const size_t sfl = (size_t)2 * NSS * npad, tfl = (size_t)2 * NST * mpad;
    std::vector<float> h(sfl > tfl ? sfl : tfl);
    float *YsA, *YsB, *YtA, *YtB; double *partial; long long *marks;
    CK(hipMalloc(&YsA, sfl * 4)); CK(hipMalloc(&YsB, sfl * 4)); CK(hipMalloc(&YtA, tfl * 4)); CK(hipMalloc(&YtB, tfl * 4));
    srand(5);
    for (float **p : {&YsA, &YsB}) { for (size_t i = 0; i < sfl; ++i) h[i] = -(rand() % 1000) * 1e-3f; CK(hipMemcpy(*p, h.data(), sfl * 4, hipMemcpyHostToDevice)); }
    for (float **p : {&YtA, &YtB}) { for (size_t i = 0; i < tfl; ++i) h[i] = -(rand() % 1000) * 1e-3f; CK(hipMemcpy(*p, h.data(), tfl * 4, hipMemcpyHostToDevice)); }
    GwdTileArgs P;
    P.YsA = YsA; P.YsB = YsB; P.YtA = YtA; P.YtB = YtB; P.n = n; P.m = m; P.npad = npad; P.mpad = mpad;
    P.T = (int)(mpad / kTile); P.ntiles = P.T * (P.T + 1) / 2;
    CK(hipMalloc(&partial, (size_t)P.ntiles * 4 * 8)); CK(hipMalloc(&marks, P.ntiles * 8));
    P.partial = partial;
    size_t lds = (size_t)(2 * 2 * NSS + 2 * 2 * NST) * kTile * 4;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    std::vector<long long> hm(P.ntiles);
    auto report = [&](const char *name, auto kern) {
        CK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        int occ = 0;
        CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, kThreads, lds));
        for (int i = 0; i < 3; ++i) kern<<<P.ntiles, kThreads, lds>>>(P, marks);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(a));
        for (int i = 0; i < 10; ++i) kern<<<P.ntiles, kThreads, lds>>>(P, marks);
        CK(hipEventRecord(b));
        CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        CK(hipMemcpy(hm.data(), marks, P.ntiles * 8, hipMemcpyDeviceToHost));
        double s = 0; for (long long v : hm) s += (double)v;
        printf("%-28s %7.1f us per launch, %d workgroups/CU, workgroup lifetime avg %.2f us (x%d tiles / (256 CUs x occ) = %.1f us)\n", name,
               ms * 100.0, occ, s / P.ntiles * 0.01, P.ntiles, s / P.ntiles * 0.01 * P.ntiles / (256.0 * occ));
    };
    report("library body", k_var<NSS, NST, 0>);
    report("loads + barrier only", k_var<NSS, NST, 1>);
    report("compute only", k_var<NSS, NST, 2>);
    report("compute, no exp", k_var<NSS, NST, 3>);
    report("compute, no MFMA", k_var<NSS, NST, 4>);
    report("compute, MFMA only", k_var<NSS, NST, 5>);
    report("cand: 11 MFMA, 0 poly", k_cand<NSS, NST, false, 0>);
    report("cand: 9 MFMA + norm adds", k_cand<NSS, NST, true, 0>);
    report("cand: 11 MFMA, 8 poly", k_cand<NSS, NST, false, 8>);
    report("cand: 11 MFMA, 16 poly", k_cand<NSS, NST, false, 16>);
    report("cand: 11 MFMA, 24 poly", k_cand<NSS, NST, false, 24>);
    report("cand: 11 MFMA, 32 poly", k_cand<NSS, NST, false, 32>);
    report("cand: 9 MFMA, 16 poly", k_cand<NSS, NST, true, 16>);
    report("cand: 9 MFMA, 24 poly", k_cand<NSS, NST, true, 24>);
    {   // the library's split-form kernel itself (operands: small random bfloat16 patterns)
        GwdTileArgs Q = P;
        uint16_t *z[4];
        const size_t sb = (size_t)2 * 2 * npad * 8, tb = (size_t)2 * 6 * mpad * 8;   // bfloat16 elements per form
        std::vector<uint16_t> hz(tb);
        for (int f = 0; f < 4; ++f) {
            const size_t cnt = f < 2 ? sb : tb;
            for (size_t i = 0; i < cnt; ++i) hz[i] = (uint16_t)(0x3c00 + (rand() & 0xff) + ((rand() & 1) << 15));   // |x| ~ 0.008
            CK(hipMalloc(&z[f], cnt * 2));
            CK(hipMemcpy(z[f], hz.data(), cnt * 2, hipMemcpyHostToDevice));
        }
        Q.YsA = reinterpret_cast<float *>(z[0]); Q.YsB = reinterpret_cast<float *>(z[1]);
        Q.YtA = reinterpret_cast<float *>(z[2]); Q.YtB = reinterpret_cast<float *>(z[3]);
        double *part4; CK(hipMalloc(&part4, (size_t)P.ntiles * 4 * 8));
        Q.partial = part4;
        auto kern = k_gwd_tiles_split<2, 6>;
        const size_t l2 = (size_t)2 * (2 + 6) * kTile * 16;
        int occ = 0;
        CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, kThreads, l2));
        for (int i = 0; i < 3; ++i) kern<<<P.ntiles, kThreads, l2>>>(Q);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(a));
        for (int i = 0; i < 10; ++i) kern<<<P.ntiles, kThreads, l2>>>(Q);
        CK(hipEventRecord(b));
        CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        printf("k_gwd_tiles_split<2, 6>      %7.1f us per launch, %d workgroups/CU (occupancy query)\n", ms * 100.0, occ);
    }
    lds = (size_t)(2 + 6) * 2 * kTile * 16;
    report("bf16x3 compute (2 + 6 MFMA)", k_bf<2, 6, true>);
    report("bf16x3 compute, no exp", k_bf<2, 6, false>);
    return 0;
}
