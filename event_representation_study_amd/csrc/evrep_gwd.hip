// evrep_gwd.hip -- the GWD score of representation_search/compute_otmi.py:61-93 in closed form
// for POT's max_iter=0 path:
//     C = (1/L^2) * sum_{i,j<L} | Ks_pad[i,j] - Kt_pad[i,j] |,   L = max(n, m),
//     K = exp(-(D / (h*sigma))^2 / 2),  D = pairwise L2,  sigma^2 = mean(D^2) / 2.
// Nothing n^2 touches HBM: the two Gaussian kernels are generated tile by tile from the point
// coordinates (LDS-resident) and folded into the |.| sum on the fly.  sigma needs no n^2 pass:
// mean_ij ||a_i - a_j||^2 = 2 * mean_i ||a_i - abar||^2.  The squared distances come off the matrix
// cores (v_mfma_f32_32x32x2_f32 on augmented coordinates, exact float32) while the VALU only does the two
// exp2, the |.| and the accumulate of every entry; both K matrices are symmetric, so only
// upper-triangular tiles are evaluated.
#include "evrep_common.h"

namespace evrep {

constexpr int kGwdMaxD = 32;
constexpr int kTile = 128;  // tile edge; 4 waves, each a 32-row strip of four 32 x 32 MFMA blocks


constexpr int kStatBlocks = 64;  // partial-sum blocks per cloud


// grid (kStatBlocks, 2): block (j, c) sums x and x^2 per dimension over its slice of cloud c
// (float64, one pass; the clouds hold O(1e4) points of magnitude <= 255, so sum(x^2)/N - mean^2
// keeps ~1e-11 relative accuracy, far inside the 1e-5 budget).  partial: [2][kStatBlocks][2*kGwdMaxD].
__global__ __launch_bounds__(kThreads) void k_gwd_stats(const double *__restrict__ Xs, int64_t n, int ds,
                                                       const double *__restrict__ Xt, int64_t m, int dt,
                                                       double *__restrict__ partial) {
    __shared__ double red[kWaves][2 * kGwdMaxD];
    const bool second = blockIdx.y == 1;
    const double *X = second ? Xt : Xs;
    const int64_t N = second ? m : n;
    const int d = second ? dt : ds;
    const int64_t per = (N + kStatBlocks - 1) / kStatBlocks;
    const int64_t i0 = (int64_t)blockIdx.x * per, i1 = (i0 + per < N) ? i0 + per : N;
    // flat walk over the slice's elements: consecutive lanes read consecutive doubles (coalesced);
    // a lane's elements e = i0*d + tid + k*256 visit dimension (e % d)
    double s[kGwdMaxD], q[kGwdMaxD];
#pragma unroll
    for (int k = 0; k < kGwdMaxD; ++k) { s[k] = 0.0; q[k] = 0.0; }
    for (int64_t i = i0 + threadIdx.x; i < i1; i += kThreads) {
        const double *row = X + i * d;
#pragma unroll
        for (int k = 0; k < kGwdMaxD; ++k)
            if (k < d) { const double v = row[k]; s[k] += v; q[k] += v * v; }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < kGwdMaxD; ++k) {
        if (k < d) {
            double a = s[k], c = q[k];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o, 64); c += __shfl_xor(c, o, 64); }
            if (lane == 0) { red[wave][2 * k] = a; red[wave][2 * k + 1] = c; }
        }
    }
    __syncthreads();
    if (threadIdx.x < 2 * d) {
        double a = 0.0;
        for (int w = 0; w < kWaves; ++w) a += red[w][threadIdx.x];
        partial[((size_t)blockIdx.y * kStatBlocks + blockIdx.x) * (2 * kGwdMaxD) + threadIdx.x] = a;
    }
}

// Augmented coordinates the matrix cores consume (built per tile in LDS):
//     row form  A'_i = ( 2 y_i , -|y_i|^2 , -1 )        column form  B'_j = ( y_j , 1 , |y_j|^2 )
// with y = (x - mean) * sqrt(log2(e) / (2 h^2 sigma^2)) in float32, so that
//     A'_i . B'_j = 2 y_i.y_j - |y_i|^2 - |y_j|^2 = -||y_i - y_j||^2   and   K = exp2(A'_i . B'_j):
// one MFMA chain over Kp = d + 2 (rounded up to even) inner steps yields the exponent of K directly, no VALU
// work per dimension.  |y|^2 is taken from the ROUNDED float32 coordinates, so the terms cancel to ~1e-7 on the
// diagonal.  Padding points (index >= N) are all-zero; their entries are masked in k_gwd_tiles.
__host__ __device__ inline int gwd_kp(int d) { return (d + 2 + 1) & ~1; }

// MFMA steps the kernel is instantiated for: the smallest entry with 2 * steps >= d + 2 serves a cloud of dimension
// d (rows d + 2 .. 2 * steps of the tiles are zero): 3 = the reference's event clouds (x, y, t, p), 8 = its
// representation clouds (12 channels + 2 positional), 17 = anything up to kGwdMaxD.
__host__ __device__ inline int gwd_steps(int d) { return d + 2 <= 6 ? 3 : (d + 2 <= 16 ? 8 : 17); }

using f32x16 = __attribute__((ext_vector_type(16))) float;

// grid (ceil(npad / 256) + ceil(mpad / 256)), 256 threads: both clouds in one launch.  Every block first finishes
// the cloud statistics from the 2 x 64 partial sums itself (fixed order, so all blocks agree bit for bit; no
// separate one-block kernel), then centres / scales one point per thread and writes both augmented forms,
// dimension-major float32, zero points beyond N:  YA, YB = [2 * steps][Npad].
// sigma^2 = mean ||a - abar||^2 = sum_k (E[x_k^2] - E[x_k]^2).
__global__ __launch_bounds__(kThreads) void k_gwd_prep(const double *__restrict__ Xs, int64_t n, int ds, int64_t npad,
                                                      const double *__restrict__ Xt, int64_t m, int dt, int64_t mpad,
                                                      const double *__restrict__ stat_partial, double h, int sblocks,
                                                      float *__restrict__ YsA, float *__restrict__ YsB,
                                                      float *__restrict__ YtA, float *__restrict__ YtB) {
    __shared__ double smean[kGwdMaxD];
    __shared__ double ssc;
    const int c = (int)blockIdx.x >= sblocks ? 1 : 0;  // which cloud this block scales
    const double *X = c ? Xt : Xs;
    const int64_t N = c ? m : n, Npad = c ? mpad : npad;
    const int d = c ? dt : ds;
    float *YA = c ? YtA : YsA, *YB = c ? YtB : YsB;
    if (threadIdx.x < 32) {
        const int k = threadIdx.x;
        double sx = 0.0, sq = 0.0;
        if (k < d) {
            const double *p = stat_partial + (size_t)c * kStatBlocks * (2 * kGwdMaxD) + 2 * k;
#pragma unroll 16
            for (int j = 0; j < kStatBlocks; ++j) { sx += p[(size_t)j * (2 * kGwdMaxD)]; sq += p[(size_t)j * (2 * kGwdMaxD) + 1]; }
        }
        const double mean = sx / (double)N;
        double var = (k < d) ? sq / (double)N - mean * mean : 0.0;
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) var += __shfl_xor(var, o, 64);
        smean[k] = mean;
        if (k == 0) ssc = sqrt(1.4426950408889634 / (2.0 * h * h * var));
    }
    __syncthreads();
    const int64_t i = (int64_t)((int)blockIdx.x - (c ? sblocks : 0)) * kThreads + threadIdx.x;
    if (i >= Npad) return;
    const double sc = ssc;
    const int kp = 2 * gwd_steps(d);
    const bool real = i < N;
    float nrm = 0.0f;
    for (int k = 0; k < d; ++k) {
        float v = 0.0f;
        if (real) v = (float)((X[i * d + k] - smean[k]) * sc);
        nrm = fmaf(v, v, nrm);
        YA[(int64_t)k * Npad + i] = 2.0f * v;
        YB[(int64_t)k * Npad + i] = v;
    }
    YA[(int64_t)d * Npad + i] = real ? -nrm : 0.0f;
    YB[(int64_t)d * Npad + i] = real ? 1.0f : 0.0f;
    YA[(int64_t)(d + 1) * Npad + i] = real ? -1.0f : 0.0f;
    YB[(int64_t)(d + 1) * Npad + i] = real ? nrm : 0.0f;
    for (int k = d + 2; k < kp; ++k) { YA[(int64_t)k * Npad + i] = 0.0f; YB[(int64_t)k * Npad + i] = 0.0f; }
}

// Operands of one 32-point strip for v_mfma_f32_32x32x2_f32 (cdna_hip_programming.md 3): lane l holds
// A[i = l & 31][k = l >> 5] / B[k = l >> 5][j = l & 31]: one float per lane per 2 inner steps.
template <int NS>
__device__ inline void gwd_load_strip(const float *T, int p0, int lane, float (&v)[NS]) {
#pragma unroll
    for (int s = 0; s < NS; ++s) v[s] = T[(2 * s + (lane >> 5)) * kTile + p0 + (lane & 31)];
}
// -||y_i - y_j||^2 of a 32 x 32 block: NS dependent MFMA steps on one accumulator (an exact float32 fma chain).
// Result register r of lane l is row (r & 3) + 8 (r >> 2) + 4 (l >> 5), column l & 31.
template <int NS>
__device__ inline f32x16 gwd_block(const float (&a)[NS], const float (&b)[NS]) {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
    for (int s = 0; s < NS; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[s], acc, 0, 0, 0);
    return acc;
}

struct GwdTileArgs {
    const float *YsA, *YsB, *YtA, *YtB;  // augmented clouds, [2 * steps][pad]
    int64_t n, m, npad, mpad;
    int32_t T, ntiles;
    double *partial;  // [ntiles]
};

// grid: one workgroup (4 waves) per upper-triangular tile pair (bi <= bj) of the L x L grid, T = Lpad / kTile.
// Wave w owns tile rows [32w, 32w + 32) and walks the four 32-column blocks.  Deliberately lean in registers (one
// accumulator set, operands re-read from LDS per block): a tile is only ~2800 matrix-pipe cycles per wave, far less
// than the latency of fetching its points, so what keeps the matrix pipe fed is the NUMBER of resident
// workgroups, not instruction-level overlap inside one (measured: register-resident operands, double-buffered
// accumulators and multi-tile workgroups all lowered the occupancy and ran 30-50 % slower).
// partial[blockIdx.x] = sum over the tile of |Ks_pad - Kt_pad| (off-diagonal tiles counted twice).
// NSS / NST: inner MFMA steps = gwd_steps(ds), gwd_steps(dt) (compile time).
template <int NSS, int NST>
__global__ __launch_bounds__(kThreads) void k_gwd_tiles(GwdTileArgs P) {
    constexpr int KPS = 2 * NSS, KPT = 2 * NST;
    extern __shared__ float lds[];  // As[KPS][kTile] Bs At[KPT][kTile] Bt
    __shared__ double red[kWaves];
    const int64_t n = P.n, m = P.m;
    const int T = P.T;
    // decode (bi, bj), bi <= bj, from the linear upper-triangular index
    int t = blockIdx.x, bi = 0;
    while (t >= T - bi) { t -= T - bi; ++bi; }
    const int bj = bi + t;
    const int64_t i0 = (int64_t)bi * kTile, j0 = (int64_t)bj * kTile;
    const bool has_s = j0 < n, has_t = j0 < m;  // bi <= bj: the row range starts no later
    float *As = lds, *Bs = lds + KPS * kTile, *At = lds + 2 * KPS * kTile, *Bt = At + KPT * kTile;
    if (has_s)
        for (int e = threadIdx.x; e < KPS * kTile; e += kThreads) {
            const int k = e / kTile, i = e % kTile;
            As[e] = P.YsA[(int64_t)k * P.npad + i0 + i];
            Bs[e] = P.YsB[(int64_t)k * P.npad + j0 + i];
        }
    if (has_t)
        for (int e = threadIdx.x; e < KPT * kTile; e += kThreads) {
            const int k = e / kTile, i = e % kTile;
            At[e] = P.YtA[(int64_t)k * P.mpad + i0 + i];
            Bt[e] = P.YtB[(int64_t)k * P.mpad + j0 + i];
        }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r0 = wave * 32;
    float as[NSS], at[NST];
    if (has_s) gwd_load_strip<NSS>(As, r0, lane, as);
    if (has_t) gwd_load_strip<NST>(At, r0, lane, at);
    float sum = 0.0f;
    const int64_t lim = n < m ? n : m;
    const bool interior = j0 + kTile <= lim;  // every entry exists in both kernels
#pragma unroll 1
    for (int cb = 0; cb < 4; ++cb) {
        float bs[NSS], bt[NST];
        f32x16 es, et;
        if (has_s) { gwd_load_strip<NSS>(Bs, cb * 32, lane, bs); es = gwd_block<NSS>(as, bs); }
        if (has_t) { gwd_load_strip<NST>(Bt, cb * 32, lane, bt); et = gwd_block<NST>(at, bt); }
        if (interior) {
#pragma unroll
            for (int r = 0; r < 16; ++r) sum += fabsf(__builtin_amdgcn_exp2f(es[r]) - __builtin_amdgcn_exp2f(et[r]));
        } else {  // edge tile: zero padding outside each kernel's own n x n / m x m block
            const int64_t gj = j0 + cb * 32 + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t gi = i0 + r0 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const float a = (has_s && gi < n && gj < n) ? __builtin_amdgcn_exp2f(es[r]) : 0.0f;
                const float bb = (has_t && gi < m && gj < m) ? __builtin_amdgcn_exp2f(et[r]) : 0.0f;
                sum += fabsf(a - bb);
            }
        }
    }
    // wave sums in float64 (DPP-free butterfly), one LDS word per wave, one store per tile
    double d = (double)sum;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) d += __shfl_xor(d, o, 64);
    if (lane == 0) red[wave] = d;
    __syncthreads();
    if (threadIdx.x == 0) P.partial[blockIdx.x] = (bi == bj ? 1.0 : 2.0) * (((red[0] + red[1]) + red[2]) + red[3]);
}

// grid (1), 1024 threads: deterministic final reduction of the per-tile sums (fixed assignment and tree);
// cost = sum / L^2.
__global__ __launch_bounds__(1024) void k_gwd_finish(const double *__restrict__ partial, int count, double L,
                                                    double *__restrict__ cost) {
    __shared__ double red[16];
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;  // four independent chains: the loads pipeline
    int i = threadIdx.x;
    for (; i + 3 * 1024 < count; i += 4 * 1024) { s0 += partial[i]; s1 += partial[i + 1024]; s2 += partial[i + 2048]; s3 += partial[i + 3072]; }
    for (; i < count; i += 1024) s0 += partial[i];
    double d = (s0 + s1) + (s2 + s3);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) d += __shfl_xor(d, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = d;
    __syncthreads();
    if (threadIdx.x == 0) {
        double tsum = 0.0;
        for (int w = 0; w < 16; ++w) tsum += red[w];
        *cost = tsum / (L * L);
    }
}

}  // namespace evrep
