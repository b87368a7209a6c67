// evrep_capi_gwd.hip -- the extern "C" surface, part 4: the GWD score, the OTMI clouds, the entropic-GW extension.
#include "evrep_capi_shared.h"

#include "evrep_gwd.hip"
#include "evrep_otmi.hip"
#include "evrep_gw.hip"

using namespace evrep;
using evrep_host::hip_check;


template <int NSS, int NST>
static int gwd_launch_tiles(const GwdTileArgs &P, hipStream_t stream) {
    const size_t lds = (size_t)2 * (2 * NSS + 2 * NST) * kTile * sizeof(float);  // row + column tile of both clouds
    // the widest instantiations need > 64 KB of dynamic LDS.  The opt-in is a property of (function, DEVICE); it is
    // renewed on every launch that needs it instead of being remembered in a process-wide flag (a second device or a
    // second host thread would find the flag set and the attribute missing)
    if (lds > 64 * 1024) {
        int rc = hip_check(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_gwd_tiles<NSS, NST>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), "hipFuncSetAttribute(k_gwd_tiles)");
        if (rc) return rc;
    }
    k_gwd_tiles<NSS, NST><<<P.ntiles, kThreads, lds, stream>>>(P);
    return EVREP_OK;
}

// split-form tiles (clouds of <= kGwdSplitMaxD dimensions): LDS = the column tile's operands
template <int MS, int MT>
static int gwd_launch_tiles_split(const GwdTileArgs &P, hipStream_t stream) {
    const size_t lds = (size_t)2 * (MS + MT) * kTile * 16;
    k_gwd_tiles_split<MS, MT><<<P.ntiles, kThreads, lds, stream>>>(P);
    return EVREP_OK;
}
template <int MS, int MT>
static int gwd_launch_tiles_split_batch(const GwdPair *pairs, int P, int64_t tile_cap, hipStream_t stream) {
    const size_t lds = (size_t)2 * (MS + MT) * kTile * 16;
    k_gwd_tiles_split_batch<MS, MT><<<dim3((unsigned)tile_cap, (unsigned)P), kThreads, lds, stream>>>(pairs);
    return EVREP_OK;
}

template <int NSS, int NST>
static int gwd_launch_tiles_batch(const GwdPair *pairs, int P, int64_t tile_cap, hipStream_t stream) {
    const size_t lds = (size_t)2 * (2 * NSS + 2 * NST) * kTile * sizeof(float);
    if (lds > 64 * 1024) {
        int rc = hip_check(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_gwd_tiles_batch<NSS, NST>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), "hipFuncSetAttribute(k_gwd_tiles_batch)");
        if (rc) return rc;
    }
    k_gwd_tiles_batch<NSS, NST><<<dim3((unsigned)tile_cap, (unsigned)P), kThreads, lds, stream>>>(pairs);
    return EVREP_OK;
}

// ---------------------------------------------------------------------------------------------- entropic GW (F5)
template <typename T>
struct GwScratch {
    T *hC1, *hC2, *ai, *bj, *Tp, *G, *Km;
    double *u, *v, *colpart, *losspart;
    int ldn, ldm;  // leading dimensions of the n-column (hC1) and m-column (hC2^T, T, G, K) matrices
    size_t bytes;
};
template <typename T>
static GwScratch<T> gw_carve(void *scratch, int64_t n, int64_t m) {
    GwScratch<T> w;
    char *p = static_cast<char *>(scratch);
    size_t o = 0;
    auto take = [&](size_t b) { char *r = p ? p + o : nullptr; o += up256(b); return r; };
    w.ldn = gw_ld((int)n, sizeof(T)); w.ldm = gw_ld((int)m, sizeof(T));
    w.hC1 = reinterpret_cast<T *>(take((size_t)n * w.ldn * sizeof(T)));
    w.hC2 = reinterpret_cast<T *>(take((size_t)m * w.ldm * sizeof(T)));
    w.ai = reinterpret_cast<T *>(take((size_t)n * sizeof(T)));
    w.bj = reinterpret_cast<T *>(take((size_t)m * sizeof(T)));
    w.Tp = reinterpret_cast<T *>(take((size_t)n * w.ldm * sizeof(T)));
    w.G = reinterpret_cast<T *>(take((size_t)n * w.ldm * sizeof(T)));
    w.Km = reinterpret_cast<T *>(take((size_t)n * w.ldm * sizeof(T)));
    w.u = reinterpret_cast<double *>(take((size_t)n * sizeof(double)));
    w.v = reinterpret_cast<double *>(take((size_t)m * sizeof(double)));
    w.colpart = reinterpret_cast<double *>(take((size_t)kGwSlices * m * sizeof(double)));
    w.losspart = reinterpret_cast<double *>(take((size_t)n * sizeof(double)));   // one loss partial per row
    w.bytes = o;
    return w;
}

template <typename T>
static int gw_solve(const double *C1, int n, const double *C2, int m, const double *p, const double *q, int loss, double eps,
                    int outer_iters, int sinkhorn_iters, void *scratch, double *T_out, double *gw_out, hipStream_t stream) {
    GwScratch<T> w = gw_carve<T>(scratch, n, m);
    const dim3 ggrid((m + kGwBN - 1) / kGwBN, (n + kGwBM - 1) / kGwBM);
    const size_t nm = (size_t)n * m;
    const unsigned eblocks = (unsigned)((nm + 255) / 256);
    k_gw_init<T><<<n, kWave, 0, stream>>>(C1, n, p, loss, 1, w.hC1, w.ldn, w.ai);
    k_gw_init<T><<<m, kWave, 0, stream>>>(C2, m, q, loss, 2, w.hC2, w.ldm, w.bj);
    k_gw_outer<T><<<eblocks, 256, 0, stream>>>(p, q, n, m, w.ldm, w.Tp);
    LAUNCH_CHECK("k_gw_init");
    GwGemmArgs<T> g1;   // G = hC1 T
    memset(&g1, 0, sizeof(g1));
    g1.A = w.hC1; g1.B = w.Tp; g1.C = w.G; g1.M = n; g1.N = m; g1.K = n; g1.lda = w.ldn; g1.ldb = w.ldm; g1.ldc = w.ldm;
    GwGemmArgs<T> g2;   // exp(-2 (a_i + b_j - G hC2^T) / eps)   or the loss; w.hC2 holds h2(C2)^T, [K = m][N = m]
    memset(&g2, 0, sizeof(g2));
    g2.A = w.G; g2.B = w.hC2; g2.C = w.Km; g2.M = n; g2.N = m; g2.K = m; g2.ai = w.ai; g2.bj = w.bj;
    g2.lda = w.ldm; g2.ldb = w.ldm; g2.ldc = w.ldm;
    g2.Tplan = w.Tp; g2.inv_eps = 1.0 / eps; g2.partial = w.losspart;
    for (int it = 0; it < outer_iters; ++it) {
        k_gw_gemm<T, false, GW_EPI_STORE><<<ggrid, kThreads, 0, stream>>>(g1);
        k_gw_gemm<T, false, GW_EPI_STORE><<<ggrid, kThreads, 0, stream>>>(g2);
        k_gw_gibbs<T><<<eblocks, 256, 0, stream>>>(w.Km, w.ai, w.bj, n, m, w.ldm, 1.0 / eps);
        LAUNCH_CHECK("k_gw_gemm");
        k_gw_fill<<<(n + 255) / 256, 256, 0, stream>>>(w.u, n, 1.0 / n);
        k_gw_fill<<<(m + 255) / 256, 256, 0, stream>>>(w.v, m, 1.0 / m);
        for (int s = 0; s < sinkhorn_iters; ++s) {
            k_gw_colsum<T><<<dim3((m + kThreads - 1) / kThreads, kGwSlices), kThreads, 0, stream>>>(w.Km, w.u, n, m, w.ldm, w.colpart);
            k_gw_col_finish<<<(m + 255) / 256, 256, 0, stream>>>(w.colpart, q, m, w.v);
            k_gw_rowdot<T><<<n, kWave, 0, stream>>>(w.Km, w.v, p, m, w.ldm, w.u);
        }
        k_gw_plan<T><<<eblocks, 256, 0, stream>>>(w.Km, w.u, w.v, n, m, w.ldm, w.Tp);
        LAUNCH_CHECK("sinkhorn");
    }
    k_gw_gemm<T, false, GW_EPI_STORE><<<ggrid, kThreads, 0, stream>>>(g1);
    k_gw_gemm<T, false, GW_EPI_STORE><<<ggrid, kThreads, 0, stream>>>(g2);
    k_gw_lossrows<T><<<n, kWave, 0, stream>>>(w.Km, w.Tp, w.ai, w.bj, m, w.ldm, w.losspart);
    k_gw_loss_finish<<<1, kThreads, 0, stream>>>(w.losspart, n, gw_out);
    if (T_out) k_gw_export<T><<<eblocks, 256, 0, stream>>>(w.Tp, n, m, w.ldm, T_out);
    LAUNCH_CHECK("k_gw_loss");
    return EVREP_OK;
}

extern "C" {

// ---------------------------------------------------------------------------------------------- GWD
static int64_t pad_tile(int64_t n) { return (n + kTile - 1) / kTile * kTile; }
// bytes per point and form of a scaled cloud, whatever its dimension and form: 6 split steps x 32 B > 2 x 17 float32 steps x 4 B
constexpr size_t kGwdFormBytesMax = 192;
static_assert(kGwdFormBytesMax >= 6 * 32 && kGwdFormBytesMax >= 2 * 17 * sizeof(float), "kGwdFormBytesMax");

size_t evrep_gwd_scratch_bytes(int64_t n, int64_t m) {
    if (n <= 0 || m <= 0) return 0;
    const int64_t L = n > m ? n : m;
    const int64_t T = pad_tile(L) / kTile;
    size_t o = 0;
    o += up256(((size_t)2 * kStatBlocks * 2 * kGwdMaxD + 2 * kGwdFin) * sizeof(double));   // statistics: partial sums + final
    o += 2 * up256(kGwdFormBytesMax * (size_t)pad_tile(n));  // scaled cloud s: row + column form
    o += 2 * up256(kGwdFormBytesMax * (size_t)pad_tile(m));  // scaled cloud t
    o += up256((size_t)(T * (T + 1) / 2) * kWaves * sizeof(double));             // per-tile, per-wave sums
    return o;
}

int evrep_gwd_padded_l1(const double *Xs, int64_t n, int32_t ds, const double *Xt, int64_t m, int32_t dt, double h,
                        void *scratch, double *cost, void *stream_) {
    if (!Xs || !Xt || !scratch || !cost || n <= 0 || m <= 0 || ds <= 0 || dt <= 0 || ds > kGwdMaxD || dt > kGwdMaxD)
        return EVREP_EINVAL;
    if (!(h > 0.0) || (reinterpret_cast<uintptr_t>(scratch) & 255u)) return EVREP_EINVAL;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const int64_t L = n > m ? n : m;
    const int T = (int)(pad_tile(L) / kTile);
    if ((int64_t)T * (T + 1) / 2 * kWaves > 0x7fffffff) return EVREP_EINVAL;
    char *p = static_cast<char *>(scratch);
    double *stat_partial = reinterpret_cast<double *>(p); p += up256(((size_t)2 * kStatBlocks * 2 * kGwdMaxD + 2 * kGwdFin) * sizeof(double));
    double *fin = stat_partial + (size_t)2 * kStatBlocks * 2 * kGwdMaxD;
    const int64_t npad = pad_tile(n), mpad = pad_tile(m);
    const size_t sbytes = up256(kGwdFormBytesMax * (size_t)npad);
    const size_t tbytes = up256(kGwdFormBytesMax * (size_t)mpad);
    float *YsA = reinterpret_cast<float *>(p); p += sbytes;
    float *YsB = reinterpret_cast<float *>(p); p += sbytes;
    float *YtA = reinterpret_cast<float *>(p); p += tbytes;
    float *YtB = reinterpret_cast<float *>(p); p += tbytes;
    double *partial = reinterpret_cast<double *>(p);
    // four launches per solve: the clouds' partial sums; one prep launch for both clouds (every block finishes
    // the statistics itself); the tiles; the final sum
    k_gwd_stats<<<dim3(kStatBlocks, 2), kThreads, 0, stream>>>(Xs, n, ds, Xt, m, dt, stat_partial);
    LAUNCH_CHECK("k_gwd_stats");
    const int sblocks = (int)((npad + kThreads - 1) / kThreads), tblocks = (int)((mpad + kThreads - 1) / kThreads);
    k_gwd_stats_finish<<<2, 64, 0, stream>>>(stat_partial, n, ds, m, dt, h, fin);
    LAUNCH_CHECK("k_gwd_stats_finish");
    // the split form spreads a point's chunks over blockIdx.z
    const int prep_z = gwd_use_split(ds, dt) ? 2 * gwd_split_steps(ds > dt ? ds : dt) : 1;
    k_gwd_prep<<<dim3(sblocks + tblocks, 1, prep_z), kThreads, 0, stream>>>(Xs, n, ds, npad, Xt, m, dt, mpad, fin, sblocks, YsA, YsB, YtA, YtB);
    LAUNCH_CHECK("k_gwd_prep");
    GwdTileArgs P;
    P.YsA = YsA; P.YsB = YsB; P.YtA = YtA; P.YtB = YtB; P.n = n; P.m = m; P.npad = npad; P.mpad = mpad;
    P.T = T; P.ntiles = T * (T + 1) / 2; P.partial = partial;
    int rc = EVREP_OK;
    const int ss = gwd_steps(ds), st = gwd_steps(dt);
    if (gwd_use_split(ds, dt)) {
        const int ms = gwd_split_steps(ds), mt = gwd_split_steps(dt);
        if (ms == 2 && mt == 2) rc = gwd_launch_tiles_split<2, 2>(P, stream);
        else if (ms == 2) rc = gwd_launch_tiles_split<2, 6>(P, stream);
        else if (mt == 2) rc = gwd_launch_tiles_split<6, 2>(P, stream);
        else rc = gwd_launch_tiles_split<6, 6>(P, stream);
    } else
#define GWD_CASE(A, B) if (ss == A && st == B) rc = gwd_launch_tiles<A, B>(P, stream)
    GWD_CASE(3, 3); else GWD_CASE(3, 8); else GWD_CASE(3, 17); else GWD_CASE(8, 3); else GWD_CASE(8, 8);
    else GWD_CASE(8, 17); else GWD_CASE(17, 3); else GWD_CASE(17, 8); else GWD_CASE(17, 17);
#undef GWD_CASE
    if (rc) return rc;
    LAUNCH_CHECK("k_gwd_tiles");
    k_gwd_finish<<<1, 1024, 0, stream>>>(partial, P.ntiles * kWaves, (double)L, cost);
    LAUNCH_CHECK("k_gwd_finish");
    return EVREP_OK;
}

size_t evrep_gwd_batch_scratch_bytes(int32_t P, int32_t ds, int32_t dt, int64_t n_cap, int64_t m_cap) {
    if (P <= 0 || n_cap <= 0 || m_cap <= 0 || ds <= 0 || dt <= 0 || ds > kGwdMaxD || dt > kGwdMaxD) return 0;
    return gwd_batch_layout(P, ds, dt, n_cap, m_cap).bytes;
}

int evrep_gwd_padded_l1_batch(int32_t P, const double *Xs, const int64_t *xs_row, const int64_t *n, int32_t ds,
                              const double *Xt, const int64_t *xt_row, const int64_t *m, int32_t dt, int64_t n_cap,
                              int64_t m_cap, double h, void *scratch, double *costs, void *stream_) {
    if (P <= 0 || P > 65535 || !Xs || !Xt || !n || !m || !scratch || !costs) return EVREP_EINVAL;
    if (ds <= 0 || dt <= 0 || ds > kGwdMaxD || dt > kGwdMaxD || n_cap <= 0 || m_cap <= 0 || !(h > 0.0)) return EVREP_EINVAL;
    if (reinterpret_cast<uintptr_t>(scratch) & 255u) return EVREP_EINVAL;
    const int64_t Lc = n_cap > m_cap ? n_cap : m_cap;
    const int64_t Tc = pad_tile(Lc) / kTile;
    if (Tc * (Tc + 1) / 2 * kWaves > 0x7fffffff) return EVREP_EINVAL;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const GwdBatchLayout L = gwd_batch_layout(P, ds, dt, n_cap, m_cap);
    GwdBatchArgs B;
    B.Xs = Xs; B.Xt = Xt; B.xs_row = xs_row; B.xt_row = xt_row; B.n = n; B.m = m; B.P = P; B.ds = ds; B.dt = dt;
    B.n_cap = n_cap; B.m_cap = m_cap;
    B.scratch = static_cast<char *>(scratch);
    B.pairs = reinterpret_cast<GwdPair *>(B.scratch + L.off_pairs);
    B.total_tiles = reinterpret_cast<int64_t *>(B.scratch + L.off_total);
    B.partial = reinterpret_cast<double *>(B.scratch + L.off_partial);
    // four launches for ALL pairs: the pair table; the clouds' partial sums; the scaling pass; the tiles (a fixed grid
    // striding over the concatenated tile list, so no size has to be known on the host); the final sums
    k_gwd_batch_setup<<<1, kThreads, 0, stream>>>(B);
    LAUNCH_CHECK("k_gwd_batch_setup");
    k_gwd_stats_batch<<<dim3(kStatBlocks, 2, P), kThreads, 0, stream>>>(B.pairs, ds, dt);
    LAUNCH_CHECK("k_gwd_stats_batch");
    const int sblocks = (int)((pad_tile(n_cap) + kThreads - 1) / kThreads), tblocks = (int)((pad_tile(m_cap) + kThreads - 1) / kThreads);
    k_gwd_stats_finish_batch<<<dim3(2, P), 64, 0, stream>>>(B.pairs, ds, dt, h);
    LAUNCH_CHECK("k_gwd_stats_finish_batch");
    // a point's chunks over blockIdx.z: all of them for a few pairs (the launch is latency-bound), three slices for many
    // (183 000 tiny blocks cost more to dispatch than they work: 144 pairs 5.81 -> 5.64 ms)
    int prep_z = gwd_use_split(ds, dt) ? 2 * gwd_split_steps(ds > dt ? ds : dt) : 1;
    if (P >= 8 && prep_z > 3) prep_z = 3;
    k_gwd_prep_batch<<<dim3(sblocks + tblocks, P, prep_z), kThreads, 0, stream>>>(B.pairs, ds, dt, sblocks);
    LAUNCH_CHECK("k_gwd_prep_batch");
    int rc = EVREP_OK;
    const int ss = gwd_steps(ds), st = gwd_steps(dt);
    if (gwd_use_split(ds, dt)) {
        const int ms = gwd_split_steps(ds), mt = gwd_split_steps(dt);
        const int64_t cap = Tc * (Tc + 1) / 2;
        if (ms == 2 && mt == 2) rc = gwd_launch_tiles_split_batch<2, 2>(B.pairs, P, cap, stream);
        else if (ms == 2) rc = gwd_launch_tiles_split_batch<2, 6>(B.pairs, P, cap, stream);
        else if (mt == 2) rc = gwd_launch_tiles_split_batch<6, 2>(B.pairs, P, cap, stream);
        else rc = gwd_launch_tiles_split_batch<6, 6>(B.pairs, P, cap, stream);
    } else
#define GWD_CASE(A, Bq) if (ss == A && st == Bq) rc = gwd_launch_tiles_batch<A, Bq>(B.pairs, P, Tc * (Tc + 1) / 2, stream)
    GWD_CASE(3, 3); else GWD_CASE(3, 8); else GWD_CASE(3, 17); else GWD_CASE(8, 3); else GWD_CASE(8, 8);
    else GWD_CASE(8, 17); else GWD_CASE(17, 3); else GWD_CASE(17, 8); else GWD_CASE(17, 17);
#undef GWD_CASE
    if (rc) return rc;
    LAUNCH_CHECK("k_gwd_tiles_batch");
    k_gwd_finish_batch<<<P, 1024, 0, stream>>>(B.pairs, costs);
    LAUNCH_CHECK("k_gwd_finish_batch");
    return EVREP_OK;
}

size_t evrep_otmi_scratch_bytes(int32_t count) {
    if (count <= 0) return 0;
    const size_t a = otmi_ev_scratch_bytes(count), b = otmi_rep_scratch_bytes(count);
    return up256(a > b ? a : b);
}

int evrep_otmi_event_clouds(const int32_t *events, const int64_t *offsets, int32_t B, int32_t height, int32_t width,
                            int64_t cap, double *Xs, int64_t *n_out, int32_t *quad_out, void *scratch, void *stream_) {
    if (!events || !offsets || !Xs || !n_out || !quad_out || !scratch || B <= 0 || B > 65535 || height <= 1 || width <= 1 || cap <= 0)
        return EVREP_EINVAL;
    if ((reinterpret_cast<uintptr_t>(events) & 15u) || (reinterpret_cast<uintptr_t>(scratch) & 15u)) return EVREP_EINVAL;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const int4 *ev = reinterpret_cast<const int4 *>(events);
    const OtmiEvScratch w = otmi_ev_scratch(scratch, B);
    const dim3 grid(kOtmiEvSlices, B);
    k_otmi_ev_stats<<<grid, kOtmiThreads, 0, stream>>>(ev, offsets, height, width, w);
    LAUNCH_CHECK("k_otmi_ev_stats");
    k_otmi_ev_plan<<<B, 64, 0, stream>>>(w, quad_out);
    LAUNCH_CHECK("k_otmi_ev_plan");
    k_otmi_ev_rows<false><<<grid, kOtmiThreads, 0, stream>>>(ev, offsets, height, width, cap, w, Xs, n_out);
    LAUNCH_CHECK("k_otmi_ev_rows<count>");
    k_otmi_ev_rows<true><<<grid, kOtmiThreads, 0, stream>>>(ev, offsets, height, width, cap, w, Xs, n_out);
    LAUNCH_CHECK("k_otmi_ev_rows<write>");
    return EVREP_OK;
}

int evrep_otmi_rep_clouds(const void *rep, int32_t rep_dtype, int32_t items, int32_t B, int32_t S, int32_t C,
                          const int32_t *quad, int64_t m_cap, double *Xt, int64_t *m_out, void *scratch, void *stream_) {
    if (!rep || !quad || !Xt || !m_out || !scratch || items <= 0 || items > 65535 || B <= 0 || S < 4 || C <= 0 || C + 2 > kGwdMaxD || m_cap <= 0)
        return EVREP_EINVAL;
    if (rep_dtype != EVREP_F64 && rep_dtype != EVREP_F32) return EVREP_EINVAL;
    if (reinterpret_cast<uintptr_t>(scratch) & 15u) return EVREP_EINVAL;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    uint32_t *cnt = static_cast<uint32_t *>(scratch);
    const dim3 grid(kOtmiRepSlices, 3, items);
#define OTMI_REP(T)                                                                                                          \
    do {                                                                                                                     \
        k_otmi_rep<T, false><<<grid, kOtmiThreads, 0, stream>>>(static_cast<const T *>(rep), B, S, C, quad, m_cap, cnt, Xt, m_out); \
        k_otmi_rep<T, true><<<grid, kOtmiThreads, 0, stream>>>(static_cast<const T *>(rep), B, S, C, quad, m_cap, cnt, Xt, m_out);  \
    } while (0)
    if (rep_dtype == EVREP_F64) OTMI_REP(double); else OTMI_REP(float);
#undef OTMI_REP
    LAUNCH_CHECK("k_otmi_rep");
    return EVREP_OK;
}

size_t evrep_gw_scratch_bytes(int64_t n, int64_t m, int32_t precision) {
    if (n <= 0 || m <= 0) return 0;
    return precision == EVREP_F32 ? gw_carve<float>(nullptr, n, m).bytes : gw_carve<double>(nullptr, n, m).bytes;
}

int evrep_entropic_gw(const double *C1, int64_t n, const double *C2, int64_t m, const double *p, const double *q,
                      int32_t loss, double epsilon, int32_t outer_iters, int32_t sinkhorn_iters, int32_t precision,
                      void *scratch, double *T_out, double *gw_out, void *stream_) {
    if (!C1 || !C2 || !p || !q || !scratch || !gw_out || n <= 0 || m <= 0 || n > 46340 || m > 46340) return EVREP_EINVAL;
    if ((loss != 0 && loss != 1) || !(epsilon > 0.0) || outer_iters < 0 || sinkhorn_iters < 1) return EVREP_EINVAL;
    if ((precision != EVREP_F64 && precision != EVREP_F32) || (reinterpret_cast<uintptr_t>(scratch) & 255u)) return EVREP_EINVAL;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (precision == EVREP_F32)
        return gw_solve<float>(C1, (int)n, C2, (int)m, p, q, loss, epsilon, outer_iters, sinkhorn_iters, scratch, T_out, gw_out, stream);
    return gw_solve<double>(C1, (int)n, C2, (int)m, p, q, loss, epsilon, outer_iters, sinkhorn_iters, scratch, T_out, gw_out, stream);
}

}  // extern "C"
