// evrep_gwd.hip -- the GWD score of representation_search/compute_otmi.py:61-93 in closed form
// for POT's max_iter=0 path:
//     C = (1/L^2) * sum_{i,j<L} | Ks_pad[i,j] - Kt_pad[i,j] |,   L = max(n, m),
//     K = exp(-(D / (h*sigma))^2 / 2),  D = pairwise L2,  sigma^2 = mean(D^2) / 2.
// Nothing n^2 touches HBM: the two Gaussian kernels are generated tile by tile from the point
// coordinates (LDS-resident) and folded into the |.| sum on the fly.  sigma needs no n^2 pass:
// mean_ij ||a_i - a_j||^2 = 2 * mean_i ||a_i - abar||^2.  The squared distances come off the matrix
// cores (augmented coordinates; v_mfma_f32_32x32x16_bf16 on exact three-way bfloat16 splits of the float32 operands for
// clouds of <= 15 dimensions, v_mfma_f32_32x32x2_f32 beyond) while the VALU only does the two exp2, the |.| and the
// accumulate of every entry; both K matrices are symmetric, so only upper-triangular tiles are evaluated.
#include "evrep_common.h"

namespace evrep {

constexpr int kGwdMaxD = 32;
constexpr int kTile = 128;  // tile edge; 4 waves, each a 32-row strip of four 32 x 32 MFMA blocks


constexpr int kStatBlocks = 64;  // partial-sum blocks per cloud


// Block `blk` of kStatBlocks sums x and x^2 per dimension over its slice of one cloud
// (float64, one pass; the clouds hold O(1e4) points of magnitude <= 255, so sum(x^2)/N - mean^2
// keeps ~1e-11 relative accuracy, far inside the 1e-5 budget).  dst: this cloud's [kStatBlocks][2*kGwdMaxD].
__device__ __forceinline__ void gwd_stats_body(const double *__restrict__ X, int64_t N, int d, int blk, double *__restrict__ dst,
                                      double (*red)[2 * kGwdMaxD]) {
    const int64_t per = (N + kStatBlocks - 1) / kStatBlocks;
    const int64_t i0 = (int64_t)blk * per, i1 = (i0 + per < N) ? i0 + per : N;
    double s[kGwdMaxD], q[kGwdMaxD];
#pragma unroll
    for (int k = 0; k < kGwdMaxD; ++k) { s[k] = 0.0; q[k] = 0.0; }
    for (int64_t i = i0 + threadIdx.x; i < i1; i += kThreads) {
        const double *row = X + i * d;
#pragma unroll
        for (int k = 0; k < kGwdMaxD; ++k)
            if (k < d) { const double v = gload_f64(row + k); s[k] += v; q[k] += v * v; }
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
        gstore_f64(dst + (size_t)blk * (2 * kGwdMaxD) + threadIdx.x, a);
    }
}

// grid (kStatBlocks, 2): block (j, c) = slice j of cloud c.  partial: [2][kStatBlocks][2*kGwdMaxD].
__global__ __launch_bounds__(kThreads) void k_gwd_stats(const double *__restrict__ Xs, int64_t n, int ds,
                                                       const double *__restrict__ Xt, int64_t m, int dt,
                                                       double *__restrict__ partial) {
    __shared__ double red[kWaves][2 * kGwdMaxD];
    const bool second = blockIdx.y == 1;
    gwd_stats_body(second ? Xt : Xs, second ? m : n, second ? dt : ds, (int)blockIdx.x,
                   partial + (size_t)blockIdx.y * kStatBlocks * (2 * kGwdMaxD), red);
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

// ---- The split form (r03): the same exponent matrix off the bfloat16 matrix pipe, 16x the float32 rate.
// The float32 MFMA chain and the kernel's v_exp_f32 do not overlap on a SIMD (tools/microbench/gwd_tile_phases.hip: 31 us of
// MFMA + 15 us of exponentials = 46 us per 12.5k x 14.4k pair, each alone hides behind the other's absence), so the
// MFMA time itself had to shrink.  Every float32 operand is split EXACTLY into three bfloat16 terms, x = hi + mid + lo
// (8 + 8 + 8 mantissa bits), and a product a * b is replaced by its six largest cross terms
//     ah bh + ah bm + am bh + am bm + ah bl + al bh            (dropped: am bl + al bm + al bl < 2^-23 |a b|),
// laid side by side along the inner dimension of v_mfma_f32_32x32x16_bf16 (exact products, float32 accumulation).  The two
// norm-carrying steps (-|y_i|^2 * 1, -1 * |y_j|^2) need three terms each.  6 d + 6 slots: d = 4 -> 2 MFMA steps (it was
// 3 float32 steps of twice the cycles), d = 14 -> 6 (it was 8).  Measured against the float64 oracle the cost keeps its
// 5e-9-class relative error (tests/test_gpu_gwd*.py; a numpy model of both chains: 2.4e-9 float32, 3.7e-9 split).
// Used for clouds of up to kGwdSplitMaxD dimensions; wider ones keep the float32 chain.
constexpr int kGwdSplitMaxD = 15;
__host__ __device__ inline int gwd_split_steps(int d) { return 6 * d + 6 <= 32 ? 2 : 6; }   // of 16 slots; d <= 4 / d <= 15
__host__ __device__ inline bool gwd_use_split(int ds, int dt) { return ds <= kGwdSplitMaxD && dt <= kGwdSplitMaxD; }
// bytes per point and form of the scaled cloud a tile kernel reads
__host__ __device__ inline size_t gwd_form_bytes(int d, bool split) {
    return split ? (size_t)gwd_split_steps(d) * 32 : (size_t)2 * gwd_steps(d) * sizeof(float);
}
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

__device__ inline uint32_t gwd_bf16_bits(float x) {   // round to nearest even; the operands are finite
    const uint32_t u = __float_as_uint(x);
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}
// term t (0 hi, 1 mid, 2 lo) of the exact three-way split of x, as bfloat16 bits
__device__ inline uint32_t gwd_split_term(float x, int t) {
    const uint32_t h = gwd_bf16_bits(x);
    if (t == 0) return h;
    const float r1 = x - __uint_as_float(h << 16);          // exact
    const uint32_t m = gwd_bf16_bits(r1);
    if (t == 1) return m;
    return gwd_bf16_bits(r1 - __uint_as_float(m << 16));    // exact difference, then the third rounding
}
// Chunk C (slots 8 C .. 8 C + 7) of one point, both forms.  xrow = its float64 coordinates, fin = the cloud's means and
// scale.  Dimension k = sigma / 6 takes the term pattern (h h m m h l) x (h m h m l h); slots 6 d .. 6 d + 2 carry
// (-nrm terms) x 1, slots 6 d + 3 .. 6 d + 5 carry -1 x (nrm terms), nrm = the squared norm of the ROUNDED float32
// coordinates; anything behind is zero.  C is a compile-time value: a chunk reads the two or three coordinates its
// slots belong to, and only the chunks that hold norm slots (wave-uniform test) read the whole row.
template <int C>
__device__ inline void gwd_split_chunk(const double *__restrict__ xrow, const double *__restrict__ fin, double sc, int d,
                                       bool real, uint4 &za, uint4 &zb) {
    constexpr int K0 = (8 * C) / 6, K1 = (8 * C + 7) / 6;   // dimensions the slots cover
    float v[K1 - K0 + 1];
#pragma unroll
    for (int k = K0; k <= K1; ++k) v[k - K0] = (real && k < d) ? (float)((gload_f64(xrow + k) - gload_f64(fin + k)) * sc) : 0.0f;
    float nrm = 0.0f;
    if (8 * C + 7 >= 6 * d && real) {   // this chunk holds norm slots
#pragma unroll
        for (int k = 0; k < kGwdSplitMaxD; ++k)
            if (k < d) { const float u = (float)((gload_f64(xrow + k) - gload_f64(fin + k)) * sc); nrm = fmaf(u, u, nrm); }
    }
    uint32_t a[8], b[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int sigma = 8 * C + e;
        const int k = sigma / 6, t = sigma % 6;   // compile-time after unrolling
        uint32_t av = 0u, bv = 0u;
        if (real) {
            if (sigma < 6 * d) {
                const int ta = (t == 2 || t == 3) ? 1 : (t == 5 ? 2 : 0), tb = (t == 1 || t == 3) ? 1 : (t == 4 ? 2 : 0);
                av = gwd_split_term(2.0f * v[k - K0], ta);
                bv = gwd_split_term(v[k - K0], tb);
            } else if (sigma < 6 * d + 6) {
                const int j = sigma - 6 * d;   // 0 .. 5, wave-uniform
                if (j < 3) { av = gwd_split_term(-nrm, j); bv = 0x3f80u; }   // x 1.0
                else { av = 0xbf80u; bv = gwd_split_term(nrm, j - 3); }      // -1.0 x
            }
        }
        a[e] = av; b[e] = bv;
    }
    za = make_uint4(a[0] | a[1] << 16, a[2] | a[3] << 16, a[4] | a[5] << 16, a[6] | a[7] << 16);
    zb = make_uint4(b[0] | b[1] << 16, b[2] | b[3] << 16, b[4] | b[5] << 16, b[6] | b[7] << 16);
}

// The cloud statistics from the 64 partial sums, once per cloud (r03: every block of the scaling pass used to redo this
// chain of 128 dependent loads): fin[k] = mean of dimension k, fin[kGwdMaxD] = sqrt(log2(e) / (2 h^2 sigma^2)).
// sigma^2 = mean ||a - abar||^2 = sum_k (E[x_k^2] - E[x_k]^2).  One wave.
__device__ __forceinline__ void gwd_stats_finish_body(const double *__restrict__ stat_cloud, int64_t N, int d, double h,
                                             double *__restrict__ fin) {
    const int k = threadIdx.x;
    if (k >= 32) return;
    double sx = 0.0, sq = 0.0;
    if (k < d) {
        const double *p = stat_cloud + 2 * k;
#pragma unroll 16
        for (int j = 0; j < kStatBlocks; ++j) { sx += gload_f64(p + (size_t)j * (2 * kGwdMaxD)); sq += gload_f64(p + (size_t)j * (2 * kGwdMaxD) + 1); }
    }
    const double mean = sx / (double)N;
    double var = (k < d) ? sq / (double)N - mean * mean : 0.0;
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) var += __shfl_xor(var, o, 64);
    gstore_f64(fin + k, mean);
    if (k == 0) gstore_f64(fin + kGwdMaxD, sqrt(1.4426950408889634 / (2.0 * h * h * var)));
}
constexpr int kGwdFin = kGwdMaxD + 1;   // doubles per cloud behind the partial sums
// grid (2), 64 threads
__global__ __launch_bounds__(64) void k_gwd_stats_finish(const double *__restrict__ stat_partial, int64_t n, int ds, int64_t m, int dt,
                                                        double h, double *__restrict__ fin) {
    const int c = blockIdx.x;
    gwd_stats_finish_body(stat_partial + (size_t)c * kStatBlocks * (2 * kGwdMaxD), c ? m : n, c ? dt : ds, h, fin + c * kGwdFin);
}

// One block of the scaling pass of ONE cloud (blk = its index among the cloud's ceil(Npad / 256) blocks): centres /
// scales one point per thread (fin = the cloud's means and scale, gwd_stats_finish_body) and writes both forms, zero
// points beyond N.  float32 form: dimension-major YA, YB = [2 * steps][Npad].  Split form: chunk `sub` of the point
// (the launch spreads the chunks over blockIdx.z), ZA, ZB = [2 * split_steps][Npad] x 8 bfloat16, one 16-byte store each.
__device__ __forceinline__ void gwd_prep_body(const double *__restrict__ X, int64_t N, int d, int64_t Npad, int blk,
                                     const double *__restrict__ fin, bool split, int sub, int nsub,
                                     float *__restrict__ YA, float *__restrict__ YB) {
    const int64_t i = (int64_t)blk * kThreads + threadIdx.x;
    if (i >= Npad) return;
    const double sc = gload_f64(fin + kGwdMaxD);
    const int kp = 2 * gwd_steps(d);
    const bool real = i < N;
    if (split) {
        const double *xrow = X + (real ? i : 0) * d;
        const int nchunks = 2 * gwd_split_steps(d);
        for (int c = sub; c < nchunks; c += nsub) {   // nsub = gridDim.z: how far the launch spreads a point's chunks
            uint4 za = make_uint4(0u, 0u, 0u, 0u), zb = za;
            switch (c) {   // wave-uniform
#define GWD_CHUNK(C) case C: gwd_split_chunk<C>(xrow, fin, sc, d, real, za, zb); break;
                GWD_CHUNK(0) GWD_CHUNK(1) GWD_CHUNK(2) GWD_CHUNK(3) GWD_CHUNK(4) GWD_CHUNK(5)
                GWD_CHUNK(6) GWD_CHUNK(7) GWD_CHUNK(8) GWD_CHUNK(9) GWD_CHUNK(10) GWD_CHUNK(11)
#undef GWD_CHUNK
                default: break;
            }
            gstore16(reinterpret_cast<uint4 *>(YA) + (int64_t)c * Npad + i, za);
            gstore16(reinterpret_cast<uint4 *>(YB) + (int64_t)c * Npad + i, zb);
        }
        return;
    }
    if (sub != 0) return;
    float nrm = 0.0f;
    for (int k = 0; k < d; ++k) {
        float v = 0.0f;
        if (real) v = (float)((gload_f64(X + i * d + k) - gload_f64(fin + k)) * sc);
        nrm = fmaf(v, v, nrm);
        gstore_f32(YA + (int64_t)k * Npad + i, 2.0f * v);
        gstore_f32(YB + (int64_t)k * Npad + i, v);
    }
    gstore_f32(YA + (int64_t)d * Npad + i, real ? -nrm : 0.0f);
    gstore_f32(YB + (int64_t)d * Npad + i, real ? 1.0f : 0.0f);
    gstore_f32(YA + (int64_t)(d + 1) * Npad + i, real ? -1.0f : 0.0f);
    gstore_f32(YB + (int64_t)(d + 1) * Npad + i, real ? nrm : 0.0f);
    for (int k = d + 2; k < kp; ++k) { gstore_f32(YA + (int64_t)k * Npad + i, 0.0f); gstore_f32(YB + (int64_t)k * Npad + i, 0.0f); }
}

// grid (ceil(npad / 256) + ceil(mpad / 256), 1, Z), 256 threads: both clouds in one launch; Z = the chunks per point of the
// split form (1 for the float32 form).  fin: [2][kGwdFin].
__global__ __launch_bounds__(kThreads) void k_gwd_prep(const double *__restrict__ Xs, int64_t n, int ds, int64_t npad,
                                                      const double *__restrict__ Xt, int64_t m, int dt, int64_t mpad,
                                                      const double *__restrict__ fin, int sblocks,
                                                      float *__restrict__ YsA, float *__restrict__ YsB,
                                                      float *__restrict__ YtA, float *__restrict__ YtB) {
    const int c = (int)blockIdx.x >= sblocks ? 1 : 0;  // which cloud this block scales
    gwd_prep_body(c ? Xt : Xs, c ? m : n, c ? dt : ds, c ? mpad : npad, (int)blockIdx.x - (c ? sblocks : 0),
                  fin + c * kGwdFin, gwd_use_split(ds, dt), (int)blockIdx.z, (int)gridDim.z, c ? YtA : YsA, c ? YtB : YsB);
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
    double *partial;  // [ntiles][kWaves]: one sum per wave of a tile's workgroup
};

// One workgroup (4 waves) per upper-triangular tile pair (bi <= bj) of the L x L grid, T = Lpad / kTile.
// Wave w owns tile rows [32w, 32w + 32) and walks the four 32-column blocks.  Deliberately lean in registers (one
// accumulator set, operands re-read from LDS per block): a tile is only ~2800 matrix-pipe cycles per wave, so what
// keeps the matrix pipe fed is the NUMBER of resident workgroups, not instruction-level overlap inside one (measured:
// register-resident operands, double-buffered accumulators and multi-tile workgroups all lowered the occupancy and ran
// 30-50 % slower).
// partial[tile][wave] = the sum over the wave's 32 rows of the tile of |Ks_pad - Kt_pad| (off-diagonal tiles counted twice).
// NSS / NST: inner MFMA steps = gwd_steps(ds), gwd_steps(dt) (compile time).
// waves per SIMD = workgroups per CU the tile kernels are compiled for
#define GWD_WAVES(NSS, NST) ((NSS) + (NST) <= 11 ? 5 : 4)
template <int NSS, int NST>
struct GwdTileShape {
    static constexpr int KPS = 2 * NSS, KPT = 2 * NST;
    static constexpr int ROWS = 2 * KPS + 2 * KPT;          // point rows of 128 floats, LDS order As Bs At Bt
    static constexpr int NV = ROWS * (kTile / 4);           // float4 per tile
    static constexpr int NIT = (NV + kThreads - 1) / kThreads;
};

struct GwdTilePos {
    int bi, bj;
};
// (bi, bj), bi <= bj, from the linear upper-triangular index: row bi starts at tile bi T - bi (bi - 1) / 2.  A binary
// search in scalar registers (wave-uniform; the r02 form walked the rows one by one, up to T scalar iterations per tile; a
// closed form with a float square root and 64-bit corrections cost the kernel 7 us).
__device__ inline GwdTilePos gwd_tile_pos(int T, int tile) {
    int lo = 0, hi = T - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        const int start = mid * T - ((mid * (mid - 1)) >> 1);
        if (start <= tile) lo = mid; else hi = mid - 1;
    }
    GwdTilePos q;
    q.bi = lo; q.bj = lo + (tile - (lo * T - ((lo * (lo - 1)) >> 1)));
    return q;
}

// The tile's point rows as float4, EVERY load of a thread issued before anything waits for one (r02 / early r03: eleven
// dependent load -> wait -> LDS write rounds per tile; the rows are padded to whole tiles and 16-byte aligned).
template <int NSS, int NST>
__device__ inline void gwd_tile_fetch(const GwdTileArgs &P, GwdTilePos q, int tid, float4 (&v)[GwdTileShape<NSS, NST>::NIT]) {
    using S = GwdTileShape<NSS, NST>;
    const int64_t i0 = (int64_t)q.bi * kTile, j0 = (int64_t)q.bj * kTile;
    const bool has_s = j0 < P.n, has_t = j0 < P.m;  // bi <= bj: the row range starts no later
    const float *ysa = P.YsA, *ysb = P.YsB, *yta = P.YtA, *ytb = P.YtB;   // values, not members: a select of members
    const int64_t npad = P.npad, mpad = P.mpad;                           // would index the struct in scratch
#pragma unroll
    for (int it = 0; it < S::NIT; ++it) {
        const int e = tid + it * kThreads, row = e / (kTile / 4), c = (e % (kTile / 4)) * 4;
        v[it] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (e < S::NV) {
            const bool s_row = row < 2 * S::KPS;
            const int k = s_row ? (row < S::KPS ? row : row - S::KPS)
                                : (row < 2 * S::KPS + S::KPT ? row - 2 * S::KPS : row - 2 * S::KPS - S::KPT);
            const bool a_form = s_row ? row < S::KPS : row < 2 * S::KPS + S::KPT;
            const float *base = s_row ? (a_form ? ysa : ysb) : (a_form ? yta : ytb);
            const int64_t pad = s_row ? npad : mpad;
            if (s_row ? has_s : has_t)
                v[it] = gload16f(base + (int64_t)k * pad + (a_form ? i0 : j0) + c);
        }
    }
}
template <int NSS, int NST>
__device__ inline void gwd_tile_stage(const float4 (&v)[GwdTileShape<NSS, NST>::NIT], float *lds, int tid) {
    using S = GwdTileShape<NSS, NST>;
#pragma unroll
    for (int it = 0; it < S::NIT; ++it) {
        const int e = tid + it * kThreads;
        if (e < S::NV) *reinterpret_cast<float4 *>(lds + 4 * e) = v[it];
    }
}

// The tile's sums from the staged rows (the caller's barrier stands between the stage and this).
template <int NSS, int NST>
__device__ inline void gwd_tile_compute(const GwdTileArgs &P, int tile, GwdTilePos q, const float *lds, int tid) {
    using S = GwdTileShape<NSS, NST>;
    const int64_t n = P.n, m = P.m;
    const int64_t i0 = (int64_t)q.bi * kTile, j0 = (int64_t)q.bj * kTile;
    const bool has_s = j0 < n, has_t = j0 < m;
    const float *As = lds, *Bs = lds + S::KPS * kTile, *At = lds + 2 * S::KPS * kTile, *Bt = At + S::KPT * kTile;
    const int lane = tid & 63, wave = tid >> 6;
    const int r0 = wave * 32;
    float as[NSS], at[NST];
    if (has_s) gwd_load_strip<NSS>(As, r0, lane, as);
    if (has_t) gwd_load_strip<NST>(At, r0, lane, at);
    float sum = 0.0f;
    const int64_t lim = n < m ? n : m;
    const bool interior = j0 + kTile <= lim;  // every entry exists in both kernels
    // behind the smaller cloud: the tile lies wholly inside the larger kernel's block and wholly in the other's padding
    // (a quarter of the tiles of a 12.5k x 14.4k pair; they went through the masked path until r03)
    const bool one_sided = (has_s != has_t) && j0 + kTile <= (n > m ? n : m);
    // rows / columns of the tile inside each kernel's own block (masked path)
    auto inside = [](int64_t N, int64_t o) -> int { const int64_t v = N - o; return v < 0 ? 0 : (v > kTile ? kTile : (int)v); };
    const int nrow_s = inside(n, i0), ncol_s = inside(n, j0), nrow_t = inside(m, i0), ncol_t = inside(m, j0);
#pragma unroll 1
    for (int cb = 0; cb < 4; ++cb) {
        float bs[NSS], bt[NST];
        f32x16 es, et;
        if (has_s) { gwd_load_strip<NSS>(Bs, cb * 32, lane, bs); es = gwd_block<NSS>(as, bs); }
        if (has_t) { gwd_load_strip<NST>(Bt, cb * 32, lane, bt); et = gwd_block<NST>(at, bt); }
        if (interior) {
#pragma unroll
            for (int r = 0; r < 16; ++r) sum += fabsf(__builtin_amdgcn_exp2f(es[r]) - __builtin_amdgcn_exp2f(et[r]));
        } else if (one_sided) {   // one kernel covers the whole tile, the other is all padding here: K >= 0, no |.|
#pragma unroll
            for (int r = 0; r < 16; ++r) sum += __builtin_amdgcn_exp2f(has_s ? es[r] : et[r]);
        } else {  // edge tile: zero padding outside each kernel's own n x n / m x m block
            // tile-local limits in 32 bits (the 64-bit compares of r02 cost ~70 instructions per tile, hoisted in front of
            // every tile's loop by the compiler)
            // (the lane index is laundered: the masks of this rare path are otherwise formed in front of EVERY tile's loop)
            int ln = lane;
            asm volatile("" : "+v"(ln));
            const int lj = cb * 32 + (ln & 31);
            const bool cs_ok = has_s && lj < ncol_s, ct_ok = has_t && lj < ncol_t;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int li = r0 + (r & 3) + 8 * (r >> 2) + 4 * (ln >> 5);
                const float a = (cs_ok && li < nrow_s) ? __builtin_amdgcn_exp2f(es[r]) : 0.0f;
                const float bb = (ct_ok && li < nrow_t) ? __builtin_amdgcn_exp2f(et[r]) : 0.0f;
                sum += fabsf(a - bb);
            }
        }
    }
    // the wave's sum in float64 (DPP-free butterfly), one store per wave: no LDS word, no closing barrier (r03; the
    // finishing kernel adds four numbers per tile instead of one)
    double d = (double)sum;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) d += __shfl_xor(d, o, 64);
    if (lane == 0) P.partial[(size_t)tile * kWaves + wave] = (q.bi == q.bj ? 1.0 : 2.0) * d;
}

// One tile, start to end (the unit the microbenchmark times).
template <int NSS, int NST>
__device__ inline void gwd_tile_body(const GwdTileArgs &P, int tile, float *lds, int tid) {
    const GwdTilePos q = gwd_tile_pos(P.T, tile);
    float4 v[GwdTileShape<NSS, NST>::NIT];
    gwd_tile_fetch<NSS, NST>(P, q, tid, v);
    gwd_tile_stage<NSS, NST>(v, lds, tid);
    __syncthreads();
    gwd_tile_compute<NSS, NST>(P, tile, q, lds, tid);
}

// grid (ntiles): one workgroup per tile.  (r03 measured the alternatives once more -- a resident grid striding over the
// tile list, with and without the next tile's rows prefetched into registers: 86-99 us against 67 us for this form;
// tools/microbench/gwd_tile_phases.hip.  The hardware's own dispatch of short workgroups balances the SIMDs better than
// any static assignment of tiles to resident workgroups, whose waves wait for each other at two barriers per tile.)
template <int NSS, int NST>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(GWD_WAVES(NSS, NST))))
void k_gwd_tiles(GwdTileArgs P) {
    extern __shared__ float lds[];  // As[KPS][kTile] Bs At[KPT][kTile] Bt
    gwd_tile_body<NSS, NST>(P, (int)blockIdx.x, lds, (int)threadIdx.x);
}

// ---- split-form tiles.  MS / MT = gwd_split_steps(ds), gwd_split_steps(dt).  The column tile's operands are staged in
// LDS ([MS + MT][2 k-halves][128 points] x 16 bytes, conflict-free ds_read_b128); the wave's own 32 rows come straight
// from L2 into registers, one 16-byte load per step (a lane's operand of v_mfma_f32_32x32x16_bf16: point l & 31, slots
// 8 (l >> 5) .. + 8).
template <int MS, int MT>
__device__ inline void gwd_tile_body_split(const GwdTileArgs &P, int tile, uint4 *lds, int tid) {
    constexpr int CH = 2 * (MS + MT);                         // 16-byte chunks per point
    constexpr int NV = CH * kTile, NIT = (NV + kThreads - 1) / kThreads;
    const GwdTilePos q = gwd_tile_pos(P.T, tile);
    const int64_t n = P.n, m = P.m;
    const int64_t i0 = (int64_t)q.bi * kTile, j0 = (int64_t)q.bj * kTile;
    const bool has_s = j0 < n, has_t = j0 < m;
    const uint4 *zsa = reinterpret_cast<const uint4 *>(P.YsA), *zsb = reinterpret_cast<const uint4 *>(P.YsB);
    const uint4 *zta = reinterpret_cast<const uint4 *>(P.YtA), *ztb = reinterpret_cast<const uint4 *>(P.YtB);
    const int64_t npad = P.npad, mpad = P.mpad;
    const int lane = tid & 63, wave = tid >> 6, r0 = wave * 32;
    // Every load of the tile is issued before anything waits: the column tile's chunks (to be staged in LDS), then the
    // wave's own rows (registers) -- the stage waits for the former only, the rows land behind the barrier.  32-bit byte
    // offsets from the wave-uniform bases (a cloud form is < 2^31 bytes: evrep_gwd_padded_l1 bounds the tile count).
    const uint32_t np16 = (uint32_t)npad * 16u, mp16 = (uint32_t)mpad * 16u;
    // thread tid stages point tid & 127 of chunks (tid >> 7) + 2 it: whether a chunk belongs to cloud s or t is a
    // compile-time property of `it`, a load's offset is one addition away from the previous one
    static_assert(kThreads == 2 * kTile && NV == NIT * kThreads, "two chunks per round of the workgroup");
    uint4 v[NIT];
    {
        const uint32_t half = (uint32_t)tid >> 7, pt16 = ((uint32_t)j0 + ((uint32_t)tid & 127u)) * 16u;
        const uint32_t s0 = half * np16 + pt16, t0 = half * mp16 + pt16;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            if (it < MS) v[it] = has_s ? gload16_at(zsb, s0 + (uint32_t)(2 * it) * np16) : make_uint4(0u, 0u, 0u, 0u);
            else v[it] = has_t ? gload16_at(ztb, t0 + (uint32_t)(2 * (it - MS)) * mp16) : make_uint4(0u, 0u, 0u, 0u);
        }
    }
    bf16x8 as[MS], at[MT];
    {
        const uint32_t row16 = ((uint32_t)i0 + (uint32_t)(r0 + (lane & 31))) * 16u;
        const uint32_t kh = (uint32_t)(lane >> 5);
        const uint32_t s0 = kh * np16 + row16, t0 = kh * mp16 + row16;
#pragma unroll
        for (int s_ = 0; s_ < MS; ++s_) {
            const uint4 w = has_s ? gload16_at(zsa, s0 + (uint32_t)(2 * s_) * np16) : make_uint4(0u, 0u, 0u, 0u);
            __builtin_memcpy(&as[s_], &w, 16);
        }
#pragma unroll
        for (int s_ = 0; s_ < MT; ++s_) {
            const uint4 w = has_t ? gload16_at(zta, t0 + (uint32_t)(2 * s_) * mp16) : make_uint4(0u, 0u, 0u, 0u);
            __builtin_memcpy(&at[s_], &w, 16);
        }
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int e = tid + it * kThreads;
        if (e < NV) lds[e] = v[it];
    }
    __syncthreads();
    const bf16x8 *B = reinterpret_cast<const bf16x8 *>(lds);
    float sum = 0.0f;
    const int64_t lim = n < m ? n : m;
    const bool interior = j0 + kTile <= lim;  // every entry exists in both kernels
    // behind the smaller cloud: the tile lies wholly inside the larger kernel's block and wholly in the other's padding
    // (a quarter of the tiles of a 12.5k x 14.4k pair; they went through the masked path until r03)
    const bool one_sided = (has_s != has_t) && j0 + kTile <= (n > m ? n : m);
    // rows / columns of the tile inside each kernel's own block (masked path)
    auto inside = [](int64_t N, int64_t o) -> int { const int64_t v = N - o; return v < 0 ? 0 : (v > kTile ? kTile : (int)v); };
    const int nrow_s = inside(n, i0), ncol_s = inside(n, j0), nrow_t = inside(m, i0), ncol_t = inside(m, j0);
#pragma unroll 1
    for (int cb = 0; cb < 4; ++cb) {
        const int col = cb * 32 + (lane & 31), kh = lane >> 5;
        // (the chains start from the literal zero operand of the first MFMA: no accumulator is initialised by the VALU)
        const f32x16 zero = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
        f32x16 es, et;   // a kernel that is absent from the tile (has_s / has_t false) is never read below
        if (has_s) {
#pragma unroll
            for (int s_ = 0; s_ < MS; ++s_)
                es = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as[s_], B[(2 * s_ + kh) * kTile + col], s_ ? es : zero, 0, 0, 0);
        }
        if (has_t) {
#pragma unroll
            for (int s_ = 0; s_ < MT; ++s_)
                et = __builtin_amdgcn_mfma_f32_32x32x16_bf16(at[s_], B[(2 * (MS + s_) + kh) * kTile + col], s_ ? et : zero, 0, 0, 0);
        }
        if (interior) {
#pragma unroll
            for (int r = 0; r < 16; ++r) sum += fabsf(__builtin_amdgcn_exp2f(es[r]) - __builtin_amdgcn_exp2f(et[r]));
        } else if (one_sided) {   // one kernel covers the whole tile, the other is all padding here: K >= 0, no |.|
#pragma unroll
            for (int r = 0; r < 16; ++r) sum += __builtin_amdgcn_exp2f(has_s ? es[r] : et[r]);
        } else {  // edge tile: zero padding outside each kernel's own n x n / m x m block
            // (the lane index is laundered: the masks of this rare path are otherwise formed in front of EVERY tile's loop)
            int ln = lane;
            asm volatile("" : "+v"(ln));
            const int lcol = cb * 32 + (ln & 31);
            const bool cs_ok = has_s && lcol < ncol_s, ct_ok = has_t && lcol < ncol_t;   // tile-local limits, 32 bits
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int li = r0 + (r & 3) + 8 * (r >> 2) + 4 * (ln >> 5);
                const float a = (cs_ok && li < nrow_s) ? __builtin_amdgcn_exp2f(es[r]) : 0.0f;
                const float bb = (ct_ok && li < nrow_t) ? __builtin_amdgcn_exp2f(et[r]) : 0.0f;
                sum += fabsf(a - bb);
            }
        }
    }
    double d = (double)sum;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) d += __shfl_xor(d, o, 64);
    if (lane == 0) P.partial[(size_t)tile * kWaves + wave] = (q.bi == q.bj ? 1.0 : 2.0) * d;
}

template <int MS, int MT>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(MS + MT <= 8 ? 5 : 4))) void k_gwd_tiles_split(GwdTileArgs P) {
    extern __shared__ uint4 lds4[];
    gwd_tile_body_split<MS, MT>(P, (int)blockIdx.x, lds4, (int)threadIdx.x);
}

// ---------------------------------------------------------------------------------------------- batched solves
// P independent (Xs, Xt) pairs in FOUR launches in all (r03; one solve alone is four launches, three of them tiny:
// 21.6 of 91 us).  The clouds' sizes live on the DEVICE (the harness kernels of evrep_otmi.hip produce them), so
// nothing here depends on a host read-back: k_gwd_batch_setup derives every pair's paddings, tile count and
// scratch pointers; the statistics / scaling grids are sized by the caller's upper bounds (surplus blocks leave at
// once); the tile kernel is a fixed grid of workgroups striding over the concatenated tile list of all pairs.
struct GwdPair {
    GwdTileArgs a;          // a.partial points at this pair's first tile sum
    const double *Xs, *Xt;
    double *stat;           // [2][kStatBlocks][2 * kGwdMaxD] partial sums, then [2][kGwdFin] final statistics
    int64_t tile0;          // first tile of the pair in the concatenated list
};

struct GwdBatchArgs {
    const double *Xs, *Xt;            // bases of the clouds
    const int64_t *xs_row, *xt_row;   // [P] first row of each pair's cloud (NULL: p * n_cap / p * m_cap)
    const int64_t *n, *m;             // [P] DEVICE sizes
    int32_t P, ds, dt;
    int64_t n_cap, m_cap;             // upper bounds: size the per-pair scratch slots
    char *scratch;
    GwdPair *pairs;                   // [P] in scratch
    int64_t *total_tiles;             // [1] in scratch
    double *partial;                  // concatenated tile sums
};

__host__ __device__ inline int64_t gwd_pad_tile(int64_t n) { return (n + kTile - 1) / kTile * kTile; }

struct GwdBatchLayout {
    size_t off_pairs, off_total, off_stat, off_y, off_partial, stat_stride, ys_bytes, yt_bytes, y_stride, partial_stride, bytes;
};
__host__ __device__ inline GwdBatchLayout gwd_batch_layout(int P, int ds, int dt, int64_t n_cap, int64_t m_cap) {
    GwdBatchLayout L;
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const int64_t Lc = n_cap > m_cap ? n_cap : m_cap;
    const int64_t T = gwd_pad_tile(Lc) / kTile;
    size_t o = 0;
    L.off_pairs = o;  o += up((size_t)P * sizeof(GwdPair));
    L.off_total = o;  o += 256;
    L.stat_stride = up(((size_t)2 * kStatBlocks * 2 * kGwdMaxD + 2 * kGwdFin) * sizeof(double));   // partial sums + final statistics
    L.off_stat = o;   o += (size_t)P * L.stat_stride;
    L.ys_bytes = up(gwd_form_bytes(ds, gwd_use_split(ds, dt)) * (size_t)gwd_pad_tile(n_cap));
    L.yt_bytes = up(gwd_form_bytes(dt, gwd_use_split(ds, dt)) * (size_t)gwd_pad_tile(m_cap));
    L.y_stride = 2 * L.ys_bytes + 2 * L.yt_bytes;
    L.off_y = o;      o += (size_t)P * L.y_stride;
    L.partial_stride = (size_t)(T * (T + 1) / 2) * kWaves;
    L.off_partial = o; o += up((size_t)P * L.partial_stride * sizeof(double));
    L.bytes = o;
    return L;
}

// grid (1), 256 threads: the pair table.  A pair whose clouds exceed the caller's bounds (or are empty) gets no tiles;
// its cost is NaN (k_gwd_finish_batch).
__global__ __launch_bounds__(kThreads) void k_gwd_batch_setup(GwdBatchArgs B) {
    __shared__ int64_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    const GwdBatchLayout L = gwd_batch_layout(B.P, B.ds, B.dt, B.n_cap, B.m_cap);
    for (int p0 = 0; p0 < B.P; p0 += kThreads) {
        const int p = p0 + threadIdx.x;
        int64_t nt = 0, n = 0, m = 0;
        int T = 0;
        if (p < B.P) {
            n = B.n[p]; m = B.m[p];
            if (n > 0 && m > 0 && n <= B.n_cap && m <= B.m_cap) {
                const int64_t Lp = n > m ? n : m;
                T = (int)(gwd_pad_tile(Lp) / kTile);
                nt = (int64_t)T * (T + 1) / 2;
            }
        }
        // exclusive scan of nt over the 256 pairs of this round (wave scans + 4 wave totals)
        __shared__ int64_t wtot[kWaves];
        int64_t incl = nt;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int64_t o = __shfl_up(incl, d, 64); if ((threadIdx.x & 63) >= d) incl += o; }
        if ((threadIdx.x & 63) == 63) wtot[threadIdx.x >> 6] = incl;
        __syncthreads();
        int64_t base = carry;
        for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) base += wtot[w];
        const int64_t tile0 = base + incl - nt;
        if (p < B.P) {
            GwdPair q;
            char *y = B.scratch + L.off_y + (size_t)p * L.y_stride;
            q.a.YsA = reinterpret_cast<const float *>(y);
            q.a.YsB = reinterpret_cast<const float *>(y + L.ys_bytes);
            q.a.YtA = reinterpret_cast<const float *>(y + 2 * L.ys_bytes);
            q.a.YtB = reinterpret_cast<const float *>(y + 2 * L.ys_bytes + L.yt_bytes);
            q.a.n = n; q.a.m = m;
            q.a.npad = gwd_pad_tile(n > 0 ? n : 1); q.a.mpad = gwd_pad_tile(m > 0 ? m : 1);
            q.a.T = T; q.a.ntiles = (int32_t)nt;
            q.a.partial = B.partial + tile0 * kWaves;
            q.Xs = B.Xs + (B.xs_row ? B.xs_row[p] : (int64_t)p * B.n_cap) * B.ds;
            q.Xt = B.Xt + (B.xt_row ? B.xt_row[p] : (int64_t)p * B.m_cap) * B.dt;
            q.stat = reinterpret_cast<double *>(B.scratch + L.off_stat + (size_t)p * L.stat_stride);
            q.tile0 = tile0;
            B.pairs[p] = q;
        }
        __syncthreads();
        if (threadIdx.x == kThreads - 1) carry = base + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) *B.total_tiles = carry;
}

// grid (kStatBlocks, 2, P): the statistics of cloud blockIdx.y of pair blockIdx.z
__global__ __launch_bounds__(kThreads) void k_gwd_stats_batch(const GwdPair *__restrict__ pairs, int ds, int dt) {
    __shared__ double red[kWaves][2 * kGwdMaxD];
    const GwdPair &q = pairs[blockIdx.z];
    if (q.a.ntiles == 0) return;
    const bool second = blockIdx.y == 1;
    gwd_stats_body(second ? q.Xt : q.Xs, second ? q.a.m : q.a.n, second ? dt : ds, (int)blockIdx.x,
                   q.stat + (size_t)blockIdx.y * kStatBlocks * (2 * kGwdMaxD), red);
}

// grid (2, P), 64 threads: the final statistics of cloud blockIdx.x of pair blockIdx.y, behind its partial sums
__global__ __launch_bounds__(64) void k_gwd_stats_finish_batch(const GwdPair *__restrict__ pairs, int ds, int dt, double h) {
    const GwdPair &q = pairs[blockIdx.y];
    if (q.a.ntiles == 0) return;
    const int c = blockIdx.x;
    gwd_stats_finish_body(q.stat + (size_t)c * kStatBlocks * (2 * kGwdMaxD), c ? q.a.m : q.a.n, c ? dt : ds, h,
                          q.stat + (size_t)2 * kStatBlocks * (2 * kGwdMaxD) + c * kGwdFin);
}

// grid (sblocks_cap + tblocks_cap, P, Z): the scaling pass of pair blockIdx.y; blocks beyond the pair's own paddings leave
__global__ __launch_bounds__(kThreads) void k_gwd_prep_batch(const GwdPair *__restrict__ pairs, int ds, int dt, int sblocks_cap) {
    const GwdPair &q = pairs[blockIdx.y];
    if (q.a.ntiles == 0) return;
    const int c = (int)blockIdx.x >= sblocks_cap ? 1 : 0;
    const int blk = (int)blockIdx.x - (c ? sblocks_cap : 0);
    const int64_t Npad = c ? q.a.mpad : q.a.npad;
    if ((int64_t)blk * kThreads >= Npad) return;
    gwd_prep_body(c ? q.Xt : q.Xs, c ? q.a.m : q.a.n, c ? dt : ds, Npad, blk,
                  q.stat + (size_t)2 * kStatBlocks * (2 * kGwdMaxD) + c * kGwdFin, gwd_use_split(ds, dt), (int)blockIdx.z, (int)gridDim.z,
                  const_cast<float *>(c ? q.a.YtA : q.a.YsA), const_cast<float *>(c ? q.a.YtB : q.a.YsB));
}

// A pair's table entry as wave-uniform values.  Inside the tile loop (global stores, barriers) the compiler cannot prove
// the table unclobbered, loads it with VECTOR loads and keeps the 20-odd dwords in VGPRs -- every address computation
// of the tile body then runs on the VALU in 64 bits.  readfirstlane puts each dword into an SGPR.
__device__ inline void gwd_load_pair_uniform(const GwdPair *q, GwdTileArgs &a, int64_t &tile0) {
    static_assert(sizeof(GwdTileArgs) % 4 == 0, "GwdTileArgs is a whole number of dwords");
    constexpr int NW = (int)(sizeof(GwdTileArgs) / 4);
    const uint32_t *src = reinterpret_cast<const uint32_t *>(&q->a);
    uint32_t w[NW];
#pragma unroll
    for (int i = 0; i < NW; ++i) w[i] = (uint32_t)__builtin_amdgcn_readfirstlane((int)src[i]);
    __builtin_memcpy(&a, w, sizeof(GwdTileArgs));
    const uint32_t *t0 = reinterpret_cast<const uint32_t *>(&q->tile0);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)t0[0]), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)t0[1]);
    tile0 = (int64_t)(((uint64_t)hi << 32) | lo);
}

// grid (tile_cap, P): workgroup (t, p) evaluates tile t of pair p; tile_cap = the tile count of the caller's size bounds,
// workgroups beyond a pair's own count leave at once.  (The first r03 form -- a resident grid striding over the
// concatenated tile list -- took 82 us per 12.5k x 14.4k pair where the single solve's one-workgroup-per-tile kernel took 67.)
template <int NSS, int NST>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(GWD_WAVES(NSS, NST))))
void k_gwd_tiles_batch(const GwdPair *__restrict__ pairs) {
    extern __shared__ float lds[];
    GwdTileArgs a;
    int64_t tile0;
    gwd_load_pair_uniform(pairs + blockIdx.y, a, tile0);
    if ((int)blockIdx.x >= a.ntiles) return;
    gwd_tile_body<NSS, NST>(a, (int)blockIdx.x, lds, (int)threadIdx.x);
}

template <int MS, int MT>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(MS + MT <= 8 ? 5 : 4)))
void k_gwd_tiles_split_batch(const GwdPair *__restrict__ pairs) {
    extern __shared__ uint4 lds4[];
    GwdTileArgs a;
    int64_t tile0;
    gwd_load_pair_uniform(pairs + blockIdx.y, a, tile0);
    if ((int)blockIdx.x >= a.ntiles) return;
    gwd_tile_body_split<MS, MT>(a, (int)blockIdx.x, lds4, (int)threadIdx.x);
}

// grid (P), 1024 threads: k_gwd_finish of pair blockIdx.x (the same fixed summation order as the single solve, so a
// batched cost equals the single-solve cost bit for bit); NaN for a pair that has no tiles (empty / oversized cloud).
__global__ __launch_bounds__(1024) void k_gwd_finish_batch(const GwdPair *__restrict__ pairs, double *__restrict__ costs) {
    __shared__ double red[16];
    const GwdPair &q = pairs[blockIdx.x];
    const int count = q.a.ntiles * kWaves;
    if (count == 0) { if (threadIdx.x == 0) costs[blockIdx.x] = __longlong_as_double(0x7ff8000000000000ll); return; }
    const double *partial = q.a.partial;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
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
        const double L = (double)(q.a.n > q.a.m ? q.a.n : q.a.m);
        costs[blockIdx.x] = tsum / (L * L);
    }
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
