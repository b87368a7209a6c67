// evrep_gw.hip -- EXTENSION: entropic Gromov-Wasserstein by projected gradient, the solver
// family BASELINE.json's north_star sketches ("pairwise cost tensor + Sinkhorn projections, MFMA for the dense
// C1.T.C2^T contraction").  No live call of the reference computes this: its only true-GW call is the dead-code
// ot.gromov.gromov_wasserstein(Ks, Kt, p, q, "kl_loss") of representation_search/gromov_wasserstein.py:62-69, and
// POT is absent from /root/reference.  What is restated here is POT's published entropic_gromov_wasserstein
// (PGD solver) with fixed iteration counts:
//     T <- p q^T
//     repeat outer_iters:   tens = constC - h1(C1) T h2(C2)^T          (POT tensor_product, init_matrix)
//                           T    = sinkhorn(p, q, 2 tens, epsilon)      (gwggrad = 2 tens; sinkhorn_knopp)
//     gw = sum(tens(T) * T)                                             (gwloss)
// PARITY UNPINNED against POT itself; the checker is oracle/gw_oracle.py (float64 numpy, same recurrences).
//
// The two GEMMs of the tensor product (2 n^2 m + 2 n m^2 flop per outer iteration, 9.7 TFLOP at the reference's
// n = 12 500, m = 14 400) run on the matrix cores: v_mfma_f64_16x16x4_f64 for float64 parity with the oracle, or
// v_mfma_f32_16x16x4_f32 (exact float32, twice the rate) when the caller asks for float32.  constC is rank one
// (a_i + b_j), so it never exists as a matrix; the Gibbs kernel exp(-2 tens / epsilon) and the final loss are one
// streaming pass each over the second GEMM's product (0.6 ms against ~180 ms of GEMM).
// Sinkhorn is HBM-bound (two passes over the n x m kernel per iteration): row sums by one wave per row, column
// sums by row slices + a fixed-order finish (no floating-point atomics: the result is deterministic).
#include "evrep_common.h"

namespace evrep {

using f64x4 = __attribute__((ext_vector_type(4))) double;
using f32x4 = __attribute__((ext_vector_type(4))) float;

// 16 x 16 x 4 MFMA traits: operands are ONE element per lane, A[i = l & 15][k = l >> 4], B[k = l >> 4][j = l & 15]
// for both precisions; the result layouts differ (cdna_hip_programming.md 3: the f64 form does NOT use the f32 map).
template <typename T> struct Mfma16;
template <> struct Mfma16<double> {
    using acc_t = f64x4;
    __device__ static inline acc_t mfma(double a, double b, acc_t c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
    __device__ static inline int row(int lane, int r) { return (lane >> 4) + 4 * r; }
};
template <> struct Mfma16<float> {
    using acc_t = f32x4;
    __device__ static inline acc_t mfma(float a, float b, acc_t c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
    __device__ static inline int row(int lane, int r) { return (lane >> 4) * 4 + r; }
};

constexpr int kGwBM = 128, kGwBN = 128, kGwBK = 16;  // workgroup tile; 4 waves, each a 64 x 64 quadrant
// LDS row padding (elements).  A fragment read takes 16 consecutive rows of k = lane >> 4 and of k + 1 in one 32-lane
// group: with a pitch of 128 + 16 the two k rows land on disjoint halves of the banks for both element sizes
// (float: pitch = 16 mod 32 words; double: 32 mod 64 words); 128 + 4 left them overlapping (2-way conflicts).
constexpr int kGwPad = 16;
// Leading dimension of the solver's own matrices: a row pitch that is a multiple of 512 bytes lets the 128 rows of
// a GEMM tile fall on a few HBM channels (m = 14 400 doubles = 225 x 512 B: the K = 14 400 GEMM ran at 38 TFLOP/s
// against 55 for the K = 12 500 one); such pitches get 64 bytes more.
__host__ __device__ inline int gw_ld(int cols, size_t elem) { return ((size_t)cols * elem) % 512 == 0 ? cols + (int)(64 / elem) : cols; }

enum GwEpilogue { GW_EPI_STORE = 0, GW_EPI_GIBBS = 1, GW_EPI_LOSS = 2 };

template <typename T>
struct GwGemmArgs {
    const T *A;      // [M][K] row-major
    const T *B;      // BT = false: [K][N] row-major;  BT = true: [N][K] row-major (the product uses B^T)
    T *C;            // [M][N] (STORE: A B;  GIBBS: exp(-2 (a_i + b_j - A B) / eps))
    int M, N, K;
    int lda, ldb, ldc;    // leading dimensions (elements) of A, B and of C / Tplan, see gw_ld()
    const T *ai, *bj;     // constC = a_i + b_j (GIBBS, LOSS)
    const T *Tplan;       // [M][N] (LOSS: sum over the tile of (a_i + b_j - A B) * T)
    double inv_eps;       // GIBBS
    double *partial;      // LOSS: one float64 per workgroup
};

// grid (ceil(N / 128), ceil(M / 128)), 256 threads; static LDS 2 x 2 x 16 x (128 + pad) elements (double-buffered).
template <typename T, bool BT, int EPI>
__global__ __launch_bounds__(kThreads) void k_gw_gemm(GwGemmArgs<T> P) {
    using MF = Mfma16<T>;
    using acc_t = typename MF::acc_t;
    constexpr int LD = kGwBM + kGwPad;
    __shared__ __align__(16) T As[2][kGwBK][LD];  // k-major: a fragment read is 16 consecutive rows of one k
    __shared__ __align__(16) T Bs[2][kGwBK][LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * kGwBM, n0 = blockIdx.x * kGwBN;
    const int M = P.M, N = P.N, K = P.K;

    // global -> register staging of one K step, 16 bytes per lane per load (VEC elements).  Operand with K
    // contiguous (A always; B when BT): the 128 x 16 tile is 128 rows of 16 / VEC vectors, consecutive lanes take
    // consecutive vectors of a row.  [K][N] operand: 16 k rows of 128 / VEC vectors, a wave reads whole rows.
    // A vector that is not entirely inside the matrix, or operands whose pitch is not a multiple of VEC, take
    // the element-wise path (zero fill).
    constexpr int VEC = 16 / (int)sizeof(T);
    constexpr int NV = kGwBM * kGwBK / VEC / kThreads;  // vectors per thread and operand: 2 (float), 4 (double)
    struct alignas(16) Vec { T e[VEC]; };
    Vec ra[NV], rb[NV];
    const bool a_vec = (P.lda % VEC) == 0, b_vec = (P.ldb % VEC) == 0;
    auto load_kcontig = [&](const T *X, int ld, bool vec_ok, int rows, int r0, int k0, Vec (&r)[NV]) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int v = tid + i * kThreads;
            const int row = r0 + v / (kGwBK / VEC), kk = k0 + (v % (kGwBK / VEC)) * VEC;
            const T *src = X + (size_t)row * ld + kk;
            if (vec_ok && row < rows && kk + VEC <= K) {
                r[i] = *reinterpret_cast<const Vec *>(src);
            } else {
#pragma unroll
                for (int q = 0; q < VEC; ++q) r[i].e[q] = (row < rows && kk + q < K) ? src[q] : (T)0;
            }
        }
    };
    auto load_ncontig = [&](const T *X, int k0, Vec (&r)[NV]) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int v = tid + i * kThreads;
            const int k = k0 + v / (kGwBN / VEC), col = n0 + (v % (kGwBN / VEC)) * VEC;
            const T *src = X + (size_t)k * P.ldb + col;
            if (b_vec && k < K && col + VEC <= N) {
                r[i] = *reinterpret_cast<const Vec *>(src);
            } else {
#pragma unroll
                for (int q = 0; q < VEC; ++q) r[i].e[q] = (k < K && col + q < N) ? src[q] : (T)0;
            }
        }
    };
    auto stage_kcontig = [&](T (&S)[kGwBK][LD], const Vec (&r)[NV]) {   // transposing: S[k][row]
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int v = tid + i * kThreads;
            const int row = v / (kGwBK / VEC), kq = (v % (kGwBK / VEC)) * VEC;
#pragma unroll
            for (int q = 0; q < VEC; ++q) S[kq + q][row] = r[i].e[q];
        }
    };
    auto stage = [&](int buf) {
        stage_kcontig(As[buf], ra);
        if (BT) {
            stage_kcontig(Bs[buf], rb);
        } else {
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int v = tid + i * kThreads;
                *reinterpret_cast<Vec *>(&Bs[buf][v / (kGwBN / VEC)][(v % (kGwBN / VEC)) * VEC]) = rb[i];
            }
        }
    };
    auto fetch = [&](int k0) {
        load_kcontig(P.A, P.lda, a_vec, M, m0, k0, ra);
        if (BT) load_kcontig(P.B, P.ldb, b_vec, N, n0, k0, rb); else load_ncontig(P.B, k0, rb);
    };

    acc_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][j][r] = (T)0;

    const int nk = (K + kGwBK - 1) / kGwBK;
    fetch(0);
    stage(0);
    __syncthreads();
    for (int ks = 0; ks < nk; ++ks) {
        const int cur = ks & 1;
        if (ks + 1 < nk) fetch((ks + 1) * kGwBK);  // the next step's global loads fly during this step's MFMAs
#pragma unroll
        for (int kk = 0; kk < kGwBK; kk += 4) {
            T a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = As[cur][kk + (lane >> 4)][wm * 64 + i * 16 + (lane & 15)];
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = Bs[cur][kk + (lane >> 4)][wn * 64 + j * 16 + (lane & 15)];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = MF::mfma(a[i], b[j], acc[i][j]);
        }
        if (ks + 1 < nk) stage(cur ^ 1);
        __syncthreads();
    }

    // epilogue: element (row, col) of accumulator tile (i, j), register r
    double lsum = 0.0;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + wm * 64 + i * 16 + MF::row(lane, r);
                const int col = n0 + wn * 64 + j * 16 + (lane & 15);
                if (row < M && col < N) {
                    const T v = acc[i][j][r];
                    if (EPI == GW_EPI_STORE) {
                        P.C[(size_t)row * P.ldc + col] = v;
                    } else {
                        const double tens = ((double)P.ai[row] + (double)P.bj[col]) - (double)v;
                        if (EPI == GW_EPI_GIBBS) P.C[(size_t)row * P.ldc + col] = (T)exp(-2.0 * tens * P.inv_eps);
                        else lsum += tens * (double)P.Tplan[(size_t)row * P.ldc + col];
                    }
                }
            }
    if (EPI == GW_EPI_LOSS) {
        __shared__ double red[kWaves];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) lsum += __shfl_xor(lsum, o, 64);
        if (lane == 0) red[wave] = lsum;
        __syncthreads();
        if (tid == 0) P.partial[blockIdx.y * gridDim.x + blockIdx.x] = ((red[0] + red[1]) + red[2]) + red[3];
    }
}

// loss functions of POT's init_matrix: 0 = square_loss (f1 = a^2, f2 = b^2, h1 = a, h2 = 2 b);
// 1 = kl_loss (f1 = a log a - a, f2 = b, h1 = a, h2 = log(b + 1e-15)).
__device__ inline double gw_f1(double a, int loss) { return loss == 0 ? a * a : (a * log(a + 1e-15) - a); }
__device__ inline double gw_f2(double b, int loss) { return loss == 0 ? b * b : b; }
__device__ inline double gw_h2(double b, int loss) { return loss == 0 ? 2.0 * b : log(b + 1e-15); }

// grid (rows), 64 threads: out_vec[i] = sum_k f(C[i][k]) w[k];  hC = h(C).
// which = 1: (f1, h1 = identity), hC[i][k];  which = 2: (f2, h2), stored TRANSPOSED, hC[k][i] = h2(C2[i][k]): the
// second GEMM of the tensor product multiplies by h2(C2)^T, and with the transpose laid down once here both GEMMs
// of every iteration read their B operand with N contiguous (135.7 -> ~94 ms for the K = 14 400 GEMM in float64).
template <typename T>
__global__ __launch_bounds__(kWave) void k_gw_init(const double *__restrict__ C, int n, const double *__restrict__ w, int loss,
                                                  int which, T *__restrict__ hC, int ld, T *__restrict__ out_vec) {
    const int i = blockIdx.x, lane = threadIdx.x;
    const double *row = C + (size_t)i * n;
    double s = 0.0;
    for (int k = lane; k < n; k += kWave) {
        const double c = row[k];
        s += (which == 1 ? gw_f1(c, loss) : gw_f2(c, loss)) * w[k];
        if (which == 1) hC[(size_t)i * ld + k] = (T)c;
        else hC[(size_t)k * ld + i] = (T)gw_h2(c, loss);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (lane == 0) out_vec[i] = (T)s;
}

// T = p q^T;  u = 1/n, v = 1/m are (re)set per Sinkhorn call by k_gw_fill.
template <typename T>
__global__ void k_gw_outer(const double *__restrict__ p, const double *__restrict__ q, int n, int m, int ld, T *__restrict__ Tp) {
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < (size_t)n * m) Tp[(e / m) * ld + e % m] = (T)(p[e / m] * q[e % m]);
}
__global__ void k_gw_fill(double *__restrict__ x, int n, double v) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) x[i] = v;
}

// Sinkhorn (POT sinkhorn_knopp): v = q / (K^T u);  u = p / (K v).  u, v and all sums in float64.
constexpr int kGwSlices = 64;  // row slices of the column-sum pass
// grid (ceil(m / 256), kGwSlices), 256 threads: part[s][j] = sum over the slice's rows of K[i][j] u[i]
template <typename T>
__global__ __launch_bounds__(kThreads) void k_gw_colsum(const T *__restrict__ Km, const double *__restrict__ u, int n, int m,
                                                       int ld, double *__restrict__ part) {
    const int j = blockIdx.x * kThreads + threadIdx.x;
    const int per = (n + kGwSlices - 1) / kGwSlices;
    const int i0 = blockIdx.y * per, i1 = min(n, i0 + per);
    if (j >= m) return;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;  // four chains: the loads pipeline; fixed order -> deterministic
    int i = i0;
    for (; i + 3 < i1; i += 4) {
        s0 += (double)Km[(size_t)i * ld + j] * u[i];
        s1 += (double)Km[(size_t)(i + 1) * ld + j] * u[i + 1];
        s2 += (double)Km[(size_t)(i + 2) * ld + j] * u[i + 2];
        s3 += (double)Km[(size_t)(i + 3) * ld + j] * u[i + 3];
    }
    for (; i < i1; ++i) s0 += (double)Km[(size_t)i * ld + j] * u[i];
    part[(size_t)blockIdx.y * m + j] = (s0 + s1) + (s2 + s3);
}
__global__ void k_gw_col_finish(const double *__restrict__ part, const double *__restrict__ q, int m, double *__restrict__ v) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= m) return;
    double s = 0.0;
    for (int r = 0; r < kGwSlices; ++r) s += part[(size_t)r * m + j];
    v[j] = q[j] / s;
}
// grid (n), 64 threads: u[i] = p[i] / sum_j K[i][j] v[j]
template <typename T>
__global__ __launch_bounds__(kWave) void k_gw_rowdot(const T *__restrict__ Km, const double *__restrict__ v, const double *__restrict__ p,
                                                    int m, int ld, double *__restrict__ u) {
    const int i = blockIdx.x, lane = threadIdx.x;
    const T *row = Km + (size_t)i * ld;
    double s = 0.0;
    for (int j = lane; j < m; j += kWave) s += (double)row[j] * v[j];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (lane == 0) u[i] = p[i] / s;
}
// T = diag(u) K diag(v)
template <typename T>
__global__ void k_gw_plan(const T *__restrict__ Km, const double *__restrict__ u, const double *__restrict__ v, int n, int m,
                          int ld, T *__restrict__ Tp) {
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < (size_t)n * m) { const size_t a = (e / m) * ld + e % m; Tp[a] = (T)(u[e / m] * (double)Km[a] * v[e % m]); }
}
// The tensor product's epilogues as their own streaming passes: folding exp() / the loss product into the GEMM's
// epilogue cost its main loop an occupancy step (128 accumulator registers + the float64 exp's temporaries: 186
// VGPRs, one workgroup per CU, 145 ms instead of 97 ms for the K = 14 400 GEMM), while one more pass over the
// 1.44 GB product costs 0.6 ms.
// K = exp(-2 (a_i + b_j - X) / eps), in place
template <typename T>
__global__ void k_gw_gibbs(T *__restrict__ X, const T *__restrict__ ai, const T *__restrict__ bj, int n, int m, int ld, double inv_eps) {
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (size_t)n * m) return;
    const size_t a = (e / m) * ld + e % m;
    const double tens = ((double)ai[e / m] + (double)bj[e % m]) - (double)X[a];
    X[a] = (T)exp(-2.0 * tens * inv_eps);
}
// grid (n), 64 threads: rowpart[i] = sum_j (a_i + b_j - X[i][j]) * T[i][j]   (fixed order: deterministic)
template <typename T>
__global__ __launch_bounds__(kWave) void k_gw_lossrows(const T *__restrict__ X, const T *__restrict__ Tp, const T *__restrict__ ai,
                                                      const T *__restrict__ bj, int m, int ld, double *__restrict__ rowpart) {
    const int i = blockIdx.x, lane = threadIdx.x;
    const double a = (double)ai[i];
    double s = 0.0;
    for (int j = lane; j < m; j += kWave) s += ((a + (double)bj[j]) - (double)X[(size_t)i * ld + j]) * (double)Tp[(size_t)i * ld + j];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (lane == 0) rowpart[i] = s;
}
// out_T (float64) = T;  gw = sum of the loss partials
template <typename T>
__global__ void k_gw_export(const T *__restrict__ Tp, int n, int m, int ld, double *__restrict__ out) {
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < (size_t)n * m) out[e] = (double)Tp[(e / m) * ld + e % m];
}
__global__ __launch_bounds__(kThreads) void k_gw_loss_finish(const double *__restrict__ partial, int count, double *__restrict__ gw) {
    __shared__ double red[kThreads];
    double s = 0.0;
    for (int i = threadIdx.x; i < count; i += kThreads) s += partial[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int w = kThreads / 2; w > 0; w >>= 1) {
        if (threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) *gw = red[0];
}

}  // namespace evrep
