// evrep_capi.hip -- the extern "C" surface declared in include/evrep.h: argument checks, workspace
// carving and kernel launches.  No allocation, no global state besides the thread-local last HIP error string, no
// environment variable read.
#include <stddef.h>
#include <stdlib.h>
#include <string.h>

#include "evrep_common.h"

// kernels (defined in the sibling translation units; everything is compiled into one .so)
#include "evrep_bin.hip"
#include "evrep_builders.hip"
#include "evrep_gwd.hip"
#include "evrep_otmi.hip"
#include "evrep_gw.hip"

using namespace evrep;

static thread_local char g_last_error[256] = "";

static int hip_check(hipError_t e, const char *what) {
    if (e == hipSuccess) return EVREP_OK;
    snprintf(g_last_error, sizeof(g_last_error), "%s: %s", what, hipGetErrorString(e));
    return EVREP_EHIP;
}
#define LAUNCH_CHECK(what)                                  \
    do {                                                    \
        int rc_ = hip_check(hipGetLastError(), what);       \
        if (rc_ != EVREP_OK) return rc_;                    \
    } while (0)

static size_t up256(size_t v) { return (v + 255) & ~(size_t)255; }
constexpr double kKeySortedMaxPerUnit = 220.0;   // average records per builder unit up to which the key-sorted pass is chosen (r04: 110 -> 220, the warm path of the builders beats the per-key column sort for a single builder per binning pass up to 500 000 events on 640x480: bin + build 159 vs 163 us for ERGO-12, 91 vs 115 us for EventStack)
constexpr double kDeepStageMinPerUnit = 110.0;   // classic passes: windows denser than this stage 256 records per unit (stage_classic)

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

int evrep_abi_version(void) { return EVREP_ABI_VERSION; }
const char *evrep_last_hip_error(void) { return g_last_error; }

int evrep_plan_init(evrep_plan *plan, int32_t B, int32_t H, int32_t W, int64_t total_events,
                    int64_t max_events_per_window) {
    return evrep_plan_init_ex(plan, B, H, W, total_events, max_events_per_window, 0u);
}

int evrep_plan_set_pacing(evrep_plan *plan, int32_t ticks) {
    if (!plan || plan->abi_version != EVREP_ABI_VERSION || ticks < -1 || ticks > 100000) return EVREP_EINVAL;
    plan->pacing = ticks;
    return EVREP_OK;
}

int evrep_plan_init_ex(evrep_plan *plan, int32_t B, int32_t H, int32_t W, int64_t total_events,
                       int64_t max_events_per_window, uint32_t flags) {
    const bool f_classic = flags & EVREP_PLAN_NO_KEY_PASS, f_three = flags & EVREP_PLAN_THREE_KERNEL;
    const bool f_force_ks = flags & EVREP_PLAN_FORCE_KEY_SORTED, f_big = flags & EVREP_PLAN_BIG_BLOCKS;
    if (!plan || B <= 0 || H <= 0 || W <= 0 || H > EVREP_MAX_DIM || W > EVREP_MAX_DIM) return EVREP_EINVAL;
    if (total_events < 0 || max_events_per_window < 0 || max_events_per_window > total_events) return EVREP_EINVAL;
    if (total_events >= (int64_t)1 << 31 || (int64_t)H * W >= (int64_t)1 << 30) return EVREP_EINVAL;
    // launch geometry: windows ride gridDim.z / .y (<= 65535) and work-unit ids are 32-bit
    if (B > 65535 || (int64_t)B * H * ((W + kChunkPx - 1) / kChunkPx) >= (int64_t)1 << 31) return EVREP_EINVAL;
    memset(plan, 0, sizeof(*plan));
    plan->abi_version = EVREP_ABI_VERSION;
    plan->B = B; plan->H = H; plan->W = W;
    plan->total_events = total_events;
    plan->max_events_per_window = max_events_per_window;
    plan->flags = (int32_t)flags;
    plan->pacing = -1;
    int64_t chunk = 2048;
    int64_t nblk = (max_events_per_window + chunk - 1) / chunk;
    if (nblk > 128) {
        chunk = ((max_events_per_window + 127) / 128 + 255) / 256 * 256;
        nblk = (max_events_per_window + chunk - 1) / chunk;
    }
    // the two-kernel pass (k_block_rowsort + k_col_sort_runs): windows of <= 64 blocks of 8192 events on sensors
    // whose per-wave row counters fit next to the 128 KB record stage in one workgroup's LDS (H <= ~900)
    const bool two_kernel = max_events_per_window <= (int64_t)kCsMaxRuns * kBsChunk &&
                            block_rowsort_lds_bytes(H) + 1024 <= 160 * 1024 && !f_three;
    if (two_kernel) {
        chunk = kBsChunk;
        nblk = (max_events_per_window + chunk - 1) / chunk;
    }
    if (nblk < 1) nblk = 1;
    plan->chunk = (int32_t)chunk;
    plan->nblk = (int32_t)nblk;
    plan->nchunk = (W + kChunkPx - 1) / kChunkPx;
    plan->reserved = two_kernel ? 1 : 0;
    // the key-sorted pass (k_block_keysort alone; the builder waves finish the order): windows of up to ~110 records per
    // builder unit on average, on sensors whose key table fits next to the record stage.  Up to ~30 per unit practically
    // every unit is ordered in one 64-lane batch; up to 128 records a unit is ordered inside LDS in two register batches
    // (the reference's own Gen1 shape, 304x240 x 50 000 events, holds ~69 per unit: binning 59.5 -> 27.9 us for 32
    // windows, every builder + 8-11 us, bin + build 15-27 % shorter; 640x480 x 150 000 / 250 000 events likewise,
    // profiles/r03/sweep_mid_density.txt); beyond, the units spill and the per-key column sort (pass 3) wins
    const int64_t NK = (int64_t)H * plan->nchunk;
    const bool key_sorted = two_kernel && max_events_per_window <= (int64_t)kBsMaxBlocks * kBsChunk && NK < 65535 &&
                            block_keysort_lds_bytes((int)NK, 4096, kBsChunk) + 1024 <= 160 * 1024 &&
                            ((double)max_events_per_window <= kKeySortedMaxPerUnit * (double)NK || f_force_ks) && !f_classic && !f_three;
    // dense windows on the same sensors: k_block_keysort + the column sort run per KEY (k_col_sort_runs, by_key),
    // then the classic builders on the pixel-sorted stream
    const bool key_dense = !key_sorted && two_kernel && NK < 65535 &&
                           block_keysort_lds_bytes((int)NK, 4096, kBsChunk) + 1024 <= 160 * 1024 &&
                           !f_classic && !f_three;
    size_t table_words = (size_t)B * nblk * (H + 1);
    if (key_dense) {
        plan->reserved = 3;
        table_words = (size_t)B * nblk * ((size_t)NK + 1);
    }
    if (key_sorted) {
        plan->reserved = 2;
        // windows of <= 16 x 4096 events: 4096-event blocks (the builder waves still find a record's run by the
        // 16-step readlane chain), twice the workgroups of the 8192-event blocks
        if (max_events_per_window <= (int64_t)kBsChainBlocks * 4096 && !f_big) {
            chunk = 4096;
            nblk = (max_events_per_window + chunk - 1) / chunk;
            if (nblk < 1) nblk = 1;
            plan->chunk = (int32_t)chunk;
            plan->nblk = (int32_t)nblk;
        }
        table_words = (size_t)B * nblk * ((size_t)NK + 1);
    }
    size_t o = 0;
    plan->off_meta = o;    o += up256((size_t)B * sizeof(WindowMeta));
    plan->off_table = o;   o += up256(table_words * sizeof(uint32_t));
    plan->off_stats = o;   o += up256((size_t)B * nblk * sizeof(BlockStats));
    plan->off_rowoff = o;  o += up256((size_t)B * (H + 1) * sizeof(uint32_t));
    plan->off_chunkoff = o; o += up256((size_t)B * H * (plan->nchunk + 1) * sizeof(uint32_t));
    plan->off_sorted1 = o; o += up256((size_t)(total_events + 1) * sizeof(Rec));
    plan->off_sorted2 = o; o += up256((size_t)(total_events + 1) * sizeof(Rec));
    plan->off_cuts = o;    o += up256((size_t)B * sizeof(TsCuts));
    // the hot list (evrep_builders.hip, run_units): count, exit ticket, then one id per unit of more than 64 records
    plan->off_scratch = o; o += up256(4 * ((size_t)kHotHdrWords + 2 * (size_t)kHotLists * hot_sublist_cap((uint32_t)(kHotParts * ((size_t)total_events / 65 + 1)))));
    plan->workspace_bytes = o;
    return EVREP_OK;
}

size_t evrep_workspace_bytes(const evrep_plan *plan) { return plan ? plan->workspace_bytes : 0; }

static int check_common(const evrep_plan *plan, const void *events, const void *offsets, const void *ws) {
    if (!plan || plan->abi_version != EVREP_ABI_VERSION || !offsets || !ws) return EVREP_EINVAL;
    if (plan->total_events > 0 && !events) return EVREP_EINVAL;
    if ((reinterpret_cast<uintptr_t>(events) & 15u) || (reinterpret_cast<uintptr_t>(ws) & 255u)) return EVREP_EINVAL;
    return EVREP_OK;
}

#define WS(type, field) reinterpret_cast<type *>(static_cast<char *>(workspace) + plan->field)
#define CWS(type, field) reinterpret_cast<const type *>(static_cast<const char *>(workspace) + plan->field)

// The pixel-sorted stream + chunk offsets + WindowMeta from the runs of k_block_keysort: the column sort, one wave per key.
static int column_sort_keys(const evrep_plan *plan, const int32_t *events, const int64_t *offsets, void *workspace, hipStream_t stream) {
    const int NK = plan->H * plan->nchunk;
    k_col_sort_runs<<<dim3((NK + kCsWaves - 1) / kCsWaves, plan->B), kCsWaves * kWave,
                                      (size_t)kCsWaves * col_sort_wave_words(kChunkPx) * 4, stream>>>(
        reinterpret_cast<const int4 *>(events), CWS(Rec, off_sorted1), offsets, CWS(uint32_t, off_table), CWS(BlockStats, off_stats), plan->H, plan->W, plan->nblk,
        plan->nchunk, plan->nchunk, plan->chunk == 4096 ? 12 : 13, 1, WS(Rec, off_sorted2), WS(uint32_t, off_chunkoff),
        WS(WindowMeta, off_meta));
    LAUNCH_CHECK("k_col_sort_runs");
    return EVREP_OK;
}

int evrep_bin_events(const evrep_plan *plan, const int32_t *events, const int64_t *offsets, void *workspace,
                     void *stream_) {
    int rc = check_common(plan, events, offsets, workspace);
    if (rc) return rc;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const int B = plan->B, H = plan->H, W = plan->W, chunk = plan->chunk, nblk = plan->nblk;
    const int4 *ev = reinterpret_cast<const int4 *>(events);
    WindowMeta *meta = WS(WindowMeta, off_meta);
    uint32_t *table = WS(uint32_t, off_table);
    uint32_t *row_off = WS(uint32_t, off_rowoff);
    Rec *s1 = WS(Rec, off_sorted1);
    Rec *s2 = WS(Rec, off_sorted2);
    BlockStats *stats = WS(BlockStats, off_stats);
    const unsigned xgrid = 8u * (unsigned)((B + 7) / 8) * (unsigned)nblk;  // XCD-aware 1-D grid, see decode_window_block
    uint32_t *hot = WS(uint32_t, off_scratch);   // the builders' hot lists start empty (k_block_keysort clears them itself)
    const_cast<evrep_plan *>(plan)->flags &= ~((int32_t)1 << 30);
    if (plan->reserved != 2 && plan->reserved != 3) {
        int rc2 = hip_check(hipMemsetAsync(hot, 0, (size_t)kHotHdrWords * 4, stream), "hipMemsetAsync(hot list)");
        if (rc2) return rc2;
    }
    if (plan->reserved == 2 || plan->reserved == 3) {
        if ((chunk != kBsChunk && chunk != 4096) || nblk > (plan->reserved == 2 ? kBsMaxBlocks : kCsMaxRuns)) return EVREP_EINVAL;
        const int NK = H * plan->nchunk;
        // > 64 KB of dynamic LDS has to be opted into per (function, device): renewed on every launch that needs it (no
        // process-wide "already done" flag: a plan may run on any device, from any host thread)
        auto opt_in = [&](const void *fn, size_t lds) -> int {
            if (lds <= 64 * 1024) return EVREP_OK;
            return hip_check(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024),
                             "hipFuncSetAttribute(k_block_keysort)");
        };
        if (chunk == 4096) {
            // the whole block in one round where that keeps the workgroup under 53 KB (three per CU; 8-byte records: 32 KB of
            // stage), else a 2048-record stage in two rounds if THAT does, else one round
            const int cap = block_keysort_lds_bytes(NK, 4096, 4096) <= 52 * 1024 ? 4096
                            : (block_keysort_lds_bytes(NK, 2048, 4096) <= 52 * 1024 ? 2048 : 4096);
            // 1024 threads x 4 events per lane (r03; r02: 512 x 8): the same 4096-event block with twice the lanes -- the
            // per-lane chains of the kernel (returning LDS atomics, group walks, scan) are half as long: 19.4 -> 18.6 us
            if (int rc2 = opt_in(reinterpret_cast<const void *>(&k_block_keysort<1024, 4>), block_keysort_lds_bytes(NK, cap, 4096))) return rc2;
            k_block_keysort<1024, 4><<<xgrid, 1024, block_keysort_lds_bytes(NK, cap, 4096), stream>>>(
                ev, offsets, B, H, W, plan->nchunk, nblk, cap, table, stats, s1, reinterpret_cast<int64_t *>(row_off), hot);
        } else {
            // two workgroups per CU (<= 79 KB each) beat one with a one-round stage: the kernel is a chain of barrier-separated
            // latencies, a second resident workgroup fills them
            const int cap = block_keysort_lds_bytes(NK, 4096, kBsChunk) <= 79 * 1024 ? 4096
                            : (block_keysort_lds_bytes(NK, kBsChunk, kBsChunk) + 1024 <= 160 * 1024 ? kBsChunk : 4096);
            if (int rc2 = opt_in(reinterpret_cast<const void *>(&k_block_keysort<1024>), block_keysort_lds_bytes(NK, cap, kBsChunk))) return rc2;
            k_block_keysort<1024><<<xgrid, kBsThreads, block_keysort_lds_bytes(NK, cap, kBsChunk), stream>>>(
                ev, offsets, B, H, W, plan->nchunk, nblk, cap, table, stats, s1, reinterpret_cast<int64_t *>(row_off), hot);
        }
        LAUNCH_CHECK("k_block_keysort");
        if (plan->reserved == 3) return column_sort_keys(plan, events, offsets, workspace, stream);
        return EVREP_OK;
    }
    if (plan->reserved == 1) {
        if (chunk != kBsChunk || nblk > kCsMaxRuns) return EVREP_EINVAL;
        const size_t lds = block_rowsort_lds_bytes(H);
        if (lds > 64 * 1024) {  // per (function, device) opt-in, renewed per launch (see k_block_keysort above)
            int rc2 = hip_check(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_block_rowsort),
                                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024),
                                "hipFuncSetAttribute(k_block_rowsort)");
            if (rc2) return rc2;
        }
        k_block_rowsort<<<xgrid, kBsThreads, lds, stream>>>(ev, offsets, B, H, W, nblk, table, stats, s1);
        LAUNCH_CHECK("k_block_rowsort");
        if (BS_DEBUG & 15) return EVREP_OK;  // timing experiments: the run table may be garbage
        constexpr int rows_per_wg = kCsRowWaves;
        k_col_sort_runs<<<dim3((H + rows_per_wg - 1) / rows_per_wg, B), kCsRowWaves * kWave, (size_t)kCsRowWaves * col_sort_wave_words(W) * 4, stream>>>(
            ev, s1, offsets, table, stats, H, W, nblk, plan->nchunk, 1, 13, 0, s2, WS(uint32_t, off_chunkoff), meta);
        LAUNCH_CHECK("k_col_sort_runs");
        return EVREP_OK;
    }
    k_row_hist<<<xgrid, kBinThreads, (size_t)H * 4, stream>>>(ev, offsets, B, H, W, chunk, nblk, table, stats);
    LAUNCH_CHECK("k_row_hist");
    if (chunk <= kStageRecs && fused_scatter_lds_bytes(H) <= 65536 && !(plan->flags & EVREP_PLAN_NO_FUSED_SCATTER)) {
        // scan fused into the scatter (one kernel and one launch boundary fewer)
        k_row_scatter_fused<<<xgrid, kBinThreads, fused_scatter_lds_bytes(H), stream>>>(ev, offsets, B, H, W, chunk, nblk, table,
                                                                                     stats, row_off, meta, s1);
        LAUNCH_CHECK("k_row_scatter_fused");
    } else {
        k_row_scan<<<B, kBinThreads, (size_t)H * 4, stream>>>(offsets, H, chunk, nblk, table, row_off, stats, meta);
        LAUNCH_CHECK("k_row_scan");
        k_row_scatter<<<xgrid, kBinThreads, (size_t)kBinWaves * H * 4, stream>>>(ev, offsets, B, H, W, chunk, nblk, table, row_off, s1);
        LAUNCH_CHECK("k_row_scatter");
    }
    k_col_sort<<<dim3(H, B), kWave, (size_t)W * 4, stream>>>(s1, row_off, H, W, plan->nchunk, s2, WS(uint32_t, off_chunkoff));
    LAUNCH_CHECK("k_col_sort");
    return EVREP_OK;
}

#define BUILDER_GRID dim3(plan->nchunk, plan->H, plan->B)

// what the builders read of the binning pass (see BinView in evrep_builders.hip)
static BinView bin_view(const evrep_plan *plan, const int32_t *events, void *workspace) {
    BinView bv;
    bv.ev = reinterpret_cast<const int4 *>(events);
    bv.fused = plan->reserved == 2 ? 1 : 0;
    bv.sorted = bv.fused ? CWS(Rec, off_sorted1) : CWS(Rec, off_sorted2);
    bv.chunk_off = CWS(uint32_t, off_chunkoff);
    bv.table = CWS(uint32_t, off_table);
    bv.stats = CWS(BlockStats, off_stats);
    bv.meta = CWS(WindowMeta, off_meta);
    bv.spill = WS(Rec, off_sorted2);
    bv.nblk = plan->nblk;
    bv.hot = WS(uint32_t, off_scratch);
    bv.hot_cap = (uint32_t)(kHotParts * (plan->total_events / 65 + 1));
    bv.hot_sel = (plan->flags >> 30) & 1;
    bv.chunk_shift = plan->chunk == 4096 ? 12 : 13;  // only read after the key-sorted pass
#ifdef EVREP_TIMING
    // 8 slots per builder wave: behind the 8-byte records of the key-sorted pass (the upper half of sorted1 is idle; sorted2 is
    // the spill stream of the warm / hot units), in sorted1 under the classic passes
    bv.dbg = reinterpret_cast<unsigned long long *>(static_cast<char *>(workspace) + plan->off_sorted1 +
                                                    (bv.fused ? up256((size_t)plan->total_events * 8) : 0));
#endif
    return bv;
}

// After the key-sorted pass: the pixel-sorted stream + chunk offsets + WindowMeta, for the consumers that walk
// them directly (k_voxel_subpixel).  The column sort of the two-kernel pass, reading a row's runs chunk by chunk.
static int ensure_column_sorted(const evrep_plan *plan, const int32_t *events, const int64_t *offsets, void *workspace, hipStream_t stream) {
    if (plan->reserved != 2) return EVREP_OK;
    return column_sort_keys(plan, events, offsets, workspace, stream);
}

// The unit of one builder wave.  span = 128-pixel chunks it takes: 2 for small pixels (float32 x 12, float64 x 5 ...) on
// sparse windows (<= 30 records per chunk on average, so a 256-pixel unit still fits the one-lane-per-non-empty-pixel
// fast path), else 1.  stage = records its LDS stage holds: 64 for one-chunk units, 128 for wider ones (they hold ~65
// records on the sparse windows they are chosen for).
static UnitCfg unit_cfg(const evrep_plan *plan, size_t pixel_bytes, int extra_chunks = 0, bool wide_part = false, bool deep_stage = true) {
    UnitCfg uc;
    const double per_chunk = (double)plan->max_events_per_window / ((double)plan->H * plan->nchunk);
    uc.span = (pixel_bytes * kChunkPx > 8192 || plan->nchunk < 2) ? 1 : (per_chunk <= 30.0 ? 2 : 1);  // a 128-pixel chunk of >= 8 KB stays alone
    // denser units (the reference's own Gen1 shape, 304x240 x 50 000 events: ~69 records per unit) are ordered inside LDS
    // in two register batches: a 128-record stage; the dense windows of the classic passes stage 256 (stage_classic)
    // (deep_stage = false: EventStack only reads a segment's last records, TimeSurface measured slower with it)
    uc.stage = (deep_stage && plan->reserved != 2 && per_chunk > kDeepStageMinPerUnit) ? 256 : ((uc.span + extra_chunks > 1 || per_chunk > 28.0 || (plan->flags & 128)) ? 128 : 64);
    uc.partpx = (wide_part && per_chunk <= 30.0) ? 2 * kPartPx : kPartPx;  // sparse windows only: dense ones lose 5 % with it
    uc.hold = plan->pacing > 0 ? plan->pacing : 0;  // automatic pacing is decided per launch (auto_hold)
    // a short tail chunk (<= 64 of 128 pixels: Gen1's 304-pixel rows end in 48) rides with the row's last unit (UnitCfg::merge);
    // TORE's units live in the OUTPUT frame and keep their own geometry (extra_chunks)
    const int tail = plan->W % kChunkPx;
    // (an experiment switch, off by default: measured at the Gen1 shape the 176-pixel unit loses the sparse emit -- ~95 records
    //  in ~80 pixels -- and its three-part tile sequence costs more than the 48-pixel tail unit it saves: ERGO-12 68 -> 82 us)
    uc.merge = (extra_chunks == 0 && tail != 0 && tail <= kChunkPx / 2 && plan->nchunk >= 2 && (plan->flags & EVREP_PLAN_X_TAIL_MERGE)) ? 1 : 0;
    return uc;
}
#define SPAN_GRID(span) dim3(units_per_row(plan->nchunk, (span), uc.merge), plan->H, plan->B)
// The builder calls on a plan alternate between the two hot lists (run_units): bit 30 of plan->flags is the library's own.
// (A plan drives ONE workspace between two binning passes; evrep_bin_events clears both lists and the bit.)
static void hot_flip(const evrep_plan *plan) { const_cast<evrep_plan *>(plan)->flags ^= (int32_t)1 << 30; }
// the hot launch behind every builder launch (run_units): the same unit numbering (span), a stage of kHotStage records, no pacing
// (only launched after the key-sorted pass: the main launches of the classic passes defer nothing)
static UnitCfg hot_cfg(UnitCfg uc) { uc.stage = kHotStage; uc.hold = 0; return uc; }

// Automatic store pacing (plan->pacing == -1) of a builder instance whose launch is bound by its HBM writes on sparse
// windows.  The waves resident on a CU offer U x unit_bytes every wave lifetime; on most placements of a ~1 GB output tensor
// MI355X serves ~5.7 TB/s when that offer exceeds ~7 TB/s (19 waves x 12 KiB ready after ~7 us: 8.5 TB/s), and
// 6.4-6.9 TB/s when it stays just below (NOTES.md 3.2, tools/experiments/pacing.py: the float64 12-channel builder takes
// 168 us unpaced, 148 us held at 7.0 us, 155 us held at 7.5 us -- a cliff on the short side, a slope on the long side, so
// the hold sits 2 % beyond the knee).  The hold scales with the bytes the CU's resident waves own.
static int auto_hold(const evrep_plan *plan, const void *kernel, size_t lds_bytes, int span, size_t pixel_bytes, int merge) {
    const double per_chunk = (double)plan->max_events_per_window / ((double)plan->H * plan->nchunk);
    if (per_chunk > 30.0) return 0;   // dense windows are bound by their segment walks, not by their stores
    int waves = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&waves, kernel, kWave, lds_bytes) != hipSuccess || waves <= 0) {
        (void)hipGetLastError();
        return 0;
    }
    const int nunit = units_per_row(plan->nchunk, span, merge);
    const double unit_bytes = (double)plan->W * (double)pixel_bytes / (double)nunit;   // a row's bytes over its units
    const double ticks = 690.0 * ((double)waves * unit_bytes) / (19.0 * 12288.0);
    return ticks < 50.0 ? 0 : (int)(ticks + 0.5);
}

int evrep_mdes_sbt_windows(const int32_t *events, const int64_t *offsets, int32_t B, int32_t H, int32_t W, int32_t *bounds,
                           uint32_t *flags, void *stream_) {
    if (!events || !offsets || !bounds || !flags || B <= 0 || B > 65535 || H <= 0 || W <= 0) return EVREP_EINVAL;
    if (reinterpret_cast<uintptr_t>(events) & 15u) return EVREP_EINVAL;
    k_mdes_sbt_windows<<<B, 1024, 0, static_cast<hipStream_t>(stream_)>>>(reinterpret_cast<const int4 *>(events), offsets, H, W, bounds, flags);
    LAUNCH_CHECK("k_mdes_sbt_windows");
    return EVREP_OK;
}

int evrep_mdes(const evrep_plan *plan, const int32_t *events, const int64_t *offsets, void *workspace, int32_t C,
               const int32_t *window, const int32_t *func, const int32_t *agg, double scale, int32_t out_dtype,
               void *out, void *stream_) {
    return evrep_mdes_ex(plan, events, offsets, workspace, C, window, func, agg, scale, out_dtype, out, nullptr, nullptr, stream_);
}

int evrep_mdes_ex(const evrep_plan *plan, const int32_t *events, const int64_t *offsets, void *workspace, int32_t C,
                  const int32_t *window, const int32_t *func, const int32_t *agg, double scale, int32_t out_dtype,
                  void *out, const int32_t *bounds, const uint32_t *flags, void *stream_) {
    int rc = check_common(plan, events, offsets, workspace);
    if (rc) return rc;
    if (C <= 0 || C > EVREP_MAX_CHANNELS || !window || !func || !agg || !out) return EVREP_EINVAL;
    if (out_dtype != EVREP_F64 && out_dtype != EVREP_F32) return EVREP_EINVAL;
    if ((bounds == nullptr) != (flags == nullptr)) return EVREP_EINVAL;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    MdesParams P;
    memset(&P, 0, sizeof(P));
    P.C = C;
    P.bounds = bounds;
    P.wflags = flags;
    for (int c = 0; c < C; ++c) { P.win[c] = window[c]; P.func[c] = func[c]; P.agg[c] = agg[c]; }
    // the ERGO-12 triples get the kernel instance with compile-time descriptors
    bool ergo = C == Ergo12Table::kC && bounds == nullptr;
    for (int c = 0; ergo && c < C; ++c)
        ergo = window[c] == Ergo12Table::kWin[c] && func[c] == Ergo12Table::kFunc[c] && agg[c] == Ergo12Table::kAgg[c];
    UnitCfg uc = unit_cfg(plan, (size_t)C * (out_dtype == EVREP_F64 ? 8 : 4));
    if ((plan->flags & EVREP_PLAN_X_SPAN2) && plan->nchunk >= 2) { uc.span = 2; uc.stage = 128; }
    const int span = uc.span;
    const bool pace_auto = plan->pacing < 0 && out_dtype == EVREP_F64 && C * 8 >= 64;   // the store-bound instances
    bool hot_launch = false;
#define MDES_LAUNCH(T, DESC)                                                                                          \
    do {                                                                                                              \
        const size_t lds_ = chunk_lds_bytes(C, sizeof(T), (span + uc.merge) * kChunkPx, uc.stage, uc.partpx);                       \
        if (pace_auto) uc.hold = auto_hold(plan, reinterpret_cast<const void *>(&k_mdes<T, DESC>), lds_, span, (size_t)C * sizeof(T), uc.merge); \
        k_mdes<T, DESC><<<SPAN_GRID(span), kWave, lds_, stream>>>(bin_view(plan, events, workspace), offsets, P, plan->H, plan->W,   \
                                                                  plan->nchunk, uc, scale, static_cast<T *>(out));           \
        /* the float64 ERGO-12 instance defers nothing (its split path, mdes_unit): no hot launch behind it */          \
        hot_launch = plan->reserved == 2 && !(MdesIsErgo12<DESC>::value && sizeof(T) == 8);                             \
        if (hot_launch) k_mdes<T, DESC, true><<<kHotGrid, kWave, chunk_lds_bytes(C, sizeof(T), (span + uc.merge) * kChunkPx, kHotStage, uc.partpx), stream>>>(  \
            bin_view(plan, events, workspace), offsets, P, plan->H, plan->W, plan->nchunk, hot_cfg(uc), scale, static_cast<T *>(out)); \
    } while (0)
#define MDES_RUNTIME(T)                                     \
    do {                                                    \
        if (C <= 4) MDES_LAUNCH(T, RuntimeDesc<4>);         \
        else if (C <= 8) MDES_LAUNCH(T, RuntimeDesc<8>);    \
        else if (C <= 12) MDES_LAUNCH(T, RuntimeDesc<12>);  \
        else MDES_LAUNCH(T, RuntimeDesc<16>);               \
    } while (0)
    if (out_dtype == EVREP_F64) {
        if (ergo) MDES_LAUNCH(double, StaticDesc<Ergo12Table>); else MDES_RUNTIME(double);
    } else {
        if (ergo) MDES_LAUNCH(float, StaticDesc<Ergo12Table>); else MDES_RUNTIME(float);
    }
#undef MDES_RUNTIME
#undef MDES_LAUNCH
    if (hot_launch) hot_flip(plan);
    LAUNCH_CHECK("k_mdes");
    return EVREP_OK;
}

int evrep_optimized(const evrep_plan *plan, const int32_t *events, const int64_t *offsets, void *workspace,
                    double scale, int32_t out_dtype, void *out, void *stream) {
    int32_t win[12], func[12], agg[12];
    for (int c = 0; c < 12; ++c) { win[c] = Ergo12Table::kWin[c]; func[c] = Ergo12Table::kFunc[c]; agg[c] = Ergo12Table::kAgg[c]; }
    return evrep_mdes(plan, events, offsets, workspace, 12, win, func, agg, scale, out_dtype, out, stream);
}

int evrep_event_stack(const evrep_plan *plan, const int32_t *events, const int64_t *offsets, void *workspace,
                      int32_t stack_size, int32_t premap, float scale, float *out, void *stream_) {
    int rc = check_common(plan, events, offsets, workspace);
    if (rc) return rc;
    if (stack_size <= 0 || stack_size > EVREP_MAX_CHANNELS || !out) return EVREP_EINVAL;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const UnitCfg uc = unit_cfg(plan, (size_t)stack_size * 4, 0, true, false);  // float32 pixels of <= 64 B: 128-pixel part tiles (see UnitCfg)
    const int span = uc.span;
    // EventStack reads the last record of a pixel only: a unit beyond the record stage keeps one (rank, polarity) word per pixel
    // (unit_records, LAST) whenever its pixels fit the hot stage -- then the main launch defers nothing and there is no hot
    // launch (and no flip of the hot lists: the current one stays empty)
    const bool last_fits = (size_t)(span + uc.merge) * kChunkPx * sizeof(Rec) <=
                           align16((size_t)uc.partpx * stack_size * 4) + (size_t)uc.stage * sizeof(Rec);
    const bool hot_launch = plan->reserved == 2 && !last_fits;
#define ES_LAUNCH(CM)                                                                                              \
    do {                                                                                                           \
    k_event_stack<CM><<<SPAN_GRID(span), kWave, chunk_lds_bytes(stack_size, 4, (span + uc.merge) * kChunkPx, uc.stage, uc.partpx), stream>>>(   \
        bin_view(plan, events, workspace), offsets, plan->H, plan->W, plan->nchunk, uc, stack_size, premap, scale, out);          \
    if (hot_launch) k_event_stack<CM, true><<<kHotGrid, kWave, chunk_lds_bytes(stack_size, 4, (span + uc.merge) * kChunkPx, kHotStage, uc.partpx), stream>>>(  \
        bin_view(plan, events, workspace), offsets, plan->H, plan->W, plan->nchunk, hot_cfg(uc), stack_size, premap, scale, out); \
    } while (0)
    if (stack_size <= 8) ES_LAUNCH(8); else if (stack_size <= 12) ES_LAUNCH(12); else ES_LAUNCH(16);
#undef ES_LAUNCH
    if (hot_launch) hot_flip(plan);
    LAUNCH_CHECK("k_event_stack");
    return EVREP_OK;
}

int evrep_time_surface(const evrep_plan *plan, const int32_t *events, const int64_t *offsets, void *workspace,
                       int32_t slices, const int32_t *indices, double tau, int32_t premap, double scale,
                       int32_t out_dtype, void *out, void *stream_) {
    return evrep_time_surface_ftime(plan, events, offsets, workspace, slices, indices, nullptr, tau, premap, scale, out_dtype, out, stream_);
}

int evrep_time_surface_ftime(const evrep_plan *plan, const int32_t *events, const int64_t *offsets, void *workspace,
                             int32_t slices, const int32_t *indices, const double *tf, double tau, int32_t premap, double scale,
                             int32_t out_dtype, void *out, void *stream_) {
    int rc = check_common(plan, events, offsets, workspace);
    if (rc) return rc;
    if (tf && !indices) return EVREP_EINVAL;   // the dispatcher's cut search is defined on its integer timestamps
    if (slices <= 0 || slices > kMaxSlices || !out || !(tau > 0.0)) return EVREP_EINVAL;
    if (out_dtype != EVREP_F64 && out_dtype != EVREP_F32) return EVREP_EINVAL;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    TsCuts *cuts = WS(TsCuts, off_cuts);
    k_ts_cuts<<<plan->B, 64, 0, stream>>>(reinterpret_cast<const int4 *>(events), offsets, slices, indices, tau, scale, cuts, tf);
    LAUNCH_CHECK("k_ts_cuts");
    // windows whose units are practically all fully staged (<= 128 records: everything the key-sorted pass is chosen for, r03;
    // r02: <= 30 records per unit on average): the kernel with the factorised exponentials compiled in -- a wave uses them
    // when ITS unit is fully staged, whatever the binning pass (Gen1 shape 88 -> 80 us)
    const bool ts_fact = (double)plan->max_events_per_window <= kDeepStageMinPerUnit * (double)plan->H * plan->nchunk;
    bool hot_launch = false;
    if (out_dtype == EVREP_F64) {
        const UnitCfg uc = unit_cfg(plan, (size_t)1 << 20, 0, false, false);  // one-chunk units whatever the slice count
#define TS_LAUNCH_F(T, CM, F, GRID, SEG)                                                                             \
    k_time_surface<T, CM, F><<<GRID, kWave, chunk_lds_bytes(2 * slices, sizeof(T), SEG, uc.stage), stream>>>(            \
        bin_view(plan, events, workspace), offsets, cuts, plan->H, plan->W, plan->nchunk, uc, slices, tau, premap, scale, tf,   \
        static_cast<T *>(out))
    // the hot launch always takes its exponentials per slice: a unit beyond the stage does so under every binning pass
    // (no hot launch, and no flip of the hot lists, when every unit beyond the record stage can be VISITED instead of ordered:
    // 2 * slices words per pixel of the unit fit the part tile -- unit_records, Visit: the float64 surfaces)
#define TS_LAUNCH(T, CM, GRID, SEG)                                                                                  \
    do {                                                                                                             \
        hot_launch = plan->reserved == 2 &&                                                                          \
                     (size_t)(SEG) * 2 * slices * 4 > align16((size_t)kPartPx * 2 * slices * sizeof(T));                          \
        if (ts_fact) TS_LAUNCH_F(T, CM, true, GRID, SEG); else TS_LAUNCH_F(T, CM, false, GRID, SEG);                  \
        if (hot_launch) k_time_surface<T, CM, false, true><<<kHotGrid, kWave, chunk_lds_bytes(2 * slices, sizeof(T), SEG, kHotStage), stream>>>( \
            bin_view(plan, events, workspace), offsets, cuts, plan->H, plan->W, plan->nchunk, hot_cfg(uc), slices, tau, premap,  \
            scale, tf, static_cast<T *>(out));                                                                           \
    } while (0)
        if (slices <= 6) TS_LAUNCH(double, 12, SPAN_GRID(1), (1 + uc.merge) * kChunkPx); else TS_LAUNCH(double, 16, SPAN_GRID(1), (1 + uc.merge) * kChunkPx);
    } else {
        const UnitCfg uc = unit_cfg(plan, (size_t)2 * slices * 4, 0, false, false);
        const int span = uc.span;
        if (slices <= 6) TS_LAUNCH(float, 12, SPAN_GRID(span), (span + uc.merge) * kChunkPx); else TS_LAUNCH(float, 16, SPAN_GRID(span), (span + uc.merge) * kChunkPx);
    }
#undef TS_LAUNCH
#undef TS_LAUNCH_F
    if (hot_launch) hot_flip(plan);
    LAUNCH_CHECK("k_time_surface");
    return EVREP_OK;
}

int evrep_tore(const evrep_plan *plan, const int32_t *events, const int64_t *offsets, void *workspace, int32_t k,
               int32_t frame_mode, const int32_t *sample_times, float scale, float *out, void *stream_) {
    return evrep_tore_ftime(plan, events, offsets, workspace, k, frame_mode, sample_times, nullptr, nullptr, scale, out, stream_);
}

int evrep_tore_ftime(const evrep_plan *plan, const int32_t *events, const int64_t *offsets, void *workspace, int32_t k,
                     int32_t frame_mode, const int32_t *sample_times, const double *tf, const double *sample_times_f,
                     float scale, float *out, void *stream_) {
    int rc = check_common(plan, events, offsets, workspace);
    if (rc) return rc;
    if (k <= 0 || k > kMaxToreK || frame_mode < 0 || frame_mode > 2 || !out) return EVREP_EINVAL;
    if (sample_times_f && !tf) return EVREP_EINVAL;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const UnitCfg uc = unit_cfg(plan, (size_t)2 * k * 4, 1);   // the shifted frame straddles one more chunk
    const int span = uc.span;
#define TORE_LAUNCH(CM)                                                                                             \
    do {                                                                                                            \
    k_tore<CM><<<SPAN_GRID(span), kWave, chunk_lds_bytes(2 * k, 4, (span + 1) * kChunkPx, uc.stage), stream>>>(          \
        reinterpret_cast<const int4 *>(events), bin_view(plan, events, workspace), offsets, sample_times, tf, sample_times_f,    \
        plan->H, plan->W, plan->nchunk, uc, k, frame_mode, scale, out);                                             \
    if (plan->reserved == 2) k_tore<CM, true><<<kHotGrid, kWave, chunk_lds_bytes(2 * k, 4, (span + 1) * kChunkPx, kHotStage), stream>>>(     \
        reinterpret_cast<const int4 *>(events), bin_view(plan, events, workspace), offsets, sample_times, tf, sample_times_f,    \
        plan->H, plan->W, plan->nchunk, hot_cfg(uc), k, frame_mode, scale, out);                                    \
    } while (0)
    if (k <= 6) TORE_LAUNCH(12); else TORE_LAUNCH(16);
#undef TORE_LAUNCH
    hot_flip(plan);
    LAUNCH_CHECK("k_tore");
    return EVREP_OK;
}

int evrep_voxel(const evrep_plan *plan, const int32_t *events, const int64_t *offsets, void *workspace, int32_t bins,
                int32_t mode, double scale, double *out, void *stream_) {
    return evrep_voxel_range(plan, events, offsets, workspace, bins, mode, scale, nullptr, out, stream_);
}

static int voxel_launch(const evrep_plan *plan, const int32_t *events, const int64_t *offsets, void *workspace,
                        int32_t bins, int32_t mode, double scale, const int64_t *t_range, const double *tnorm, double *out, void *stream_);

int evrep_voxel_range(const evrep_plan *plan, const int32_t *events, const int64_t *offsets, void *workspace,
                      int32_t bins, int32_t mode, double scale, const int64_t *t_range, double *out, void *stream_) {
    return voxel_launch(plan, events, offsets, workspace, bins, mode, scale, t_range, nullptr, out, stream_);
}

int evrep_voxel_tnorm(const evrep_plan *plan, const int32_t *events, const int64_t *offsets, void *workspace,
                      const double *tnorm, int32_t bins, double scale, double *out, void *stream_) {
    if (plan && plan->total_events > 0 && !tnorm) return EVREP_EINVAL;
    return voxel_launch(plan, events, offsets, workspace, bins, 0, scale, nullptr, tnorm, out, stream_);
}

static int voxel_launch(const evrep_plan *plan, const int32_t *events, const int64_t *offsets, void *workspace,
                        int32_t bins, int32_t mode, double scale, const int64_t *t_range, const double *tnorm, double *out, void *stream_) {
    int rc = check_common(plan, events, offsets, workspace);
    if (rc) return rc;
    if (bins <= 0 || bins > EVREP_MAX_CHANNELS || mode < 0 || mode > 2 || !out) return EVREP_EINVAL;
    if (t_range && mode != 2) return EVREP_EINVAL;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const UnitCfg uc = unit_cfg(plan, (size_t)bins * 8);
    const int span = uc.span;
#define VOXEL_LAUNCH(CM)                                                                                         \
    do {                                                                                                         \
    k_voxel<CM><<<SPAN_GRID(span), kWave, chunk_lds_bytes(bins, 8, (span + uc.merge) * kChunkPx, uc.stage), stream>>>(              \
        reinterpret_cast<const int4 *>(events), bin_view(plan, events, workspace), offsets, plan->H, plan->W, plan->nchunk, uc, \
        bins, mode, scale, t_range, tnorm, out);                                                                 \
    if (plan->reserved == 2) k_voxel<CM, true><<<kHotGrid, kWave, chunk_lds_bytes(bins, 8, (span + uc.merge) * kChunkPx, kHotStage), stream>>>(         \
        reinterpret_cast<const int4 *>(events), bin_view(plan, events, workspace), offsets, plan->H, plan->W, plan->nchunk,   \
        hot_cfg(uc), bins, mode, scale, t_range, tnorm, out);                                                    \
    } while (0)
    if (bins <= 8) VOXEL_LAUNCH(8); else VOXEL_LAUNCH(16);
#undef VOXEL_LAUNCH
    hot_flip(plan);
    LAUNCH_CHECK("k_voxel");
    return EVREP_OK;
}

int evrep_voxel_subpixel(const evrep_plan *plan, const int32_t *events, const int64_t *offsets, void *workspace,
                         const double *xy, int32_t bins, const int64_t *t_range, float *out, void *stream_) {
    int rc = check_common(plan, events, offsets, workspace);
    if (rc) return rc;
    if (bins <= 0 || bins > EVREP_MAX_CHANNELS || !out || (plan->total_events > 0 && !xy)) return EVREP_EINVAL;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const dim3 grid((unsigned)(((size_t)plan->H * plan->W + kThreads - 1) / kThreads), (unsigned)plan->B);
    rc = ensure_column_sorted(plan, events, offsets, workspace, stream);
    if (rc) return rc;
    k_voxel_subpixel<<<grid, kThreads, 0, stream>>>(reinterpret_cast<const int4 *>(events), CWS(Rec, off_sorted2),
                                                   CWS(uint32_t, off_chunkoff), offsets, xy, plan->H, plan->W, plan->nchunk,
                                                   bins, t_range, out);
    LAUNCH_CHECK("k_voxel_subpixel");
    return EVREP_OK;
}

int evrep_polstats(const evrep_plan *plan, const int32_t *events, const int64_t *offsets, void *workspace,
                   const double *tnorm, int32_t C, const int32_t *pol, const int32_t *stat, double tau, float *out,
                   void *stream_) {
    int rc = check_common(plan, events, offsets, workspace);
    if (rc) return rc;
    if (C <= 0 || C > EVREP_MAX_CHANNELS || !pol || !stat || !out) return EVREP_EINVAL;
    if (plan->total_events > 0 && !tnorm) return EVREP_EINVAL;
    PolStatParams P;
    memset(&P, 0, sizeof(P));
    P.C = C;
    P.tau = tau;
    for (int c = 0; c < C; ++c) {
        if (pol[c] < EVREP_PS_ANY || pol[c] > EVREP_PS_NEG || stat[c] < EVREP_PS_COUNT || stat[c] > EVREP_PS_SIGNED) return EVREP_EINVAL;
        if (stat[c] == EVREP_PS_EXP && !(tau > 0.0)) return EVREP_EINVAL;
        P.pol[c] = pol[c];
        P.stat[c] = stat[c];
    }
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const UnitCfg uc = unit_cfg(plan, (size_t)C * 4, 0, true);  // float32 pixels of <= 64 B: 128-pixel part tiles (see UnitCfg)
    const int span = uc.span;
#define PS_LAUNCH(CM)                                                                                                 \
    do {                                                                                                              \
        k_polstats<CM><<<SPAN_GRID(span), kWave, chunk_lds_bytes(C, 4, (span + uc.merge) * kChunkPx, uc.stage, uc.partpx), stream>>>(   \
            bin_view(plan, events, workspace), offsets, tnorm, P, plan->H, plan->W, plan->nchunk, uc, out);            \
        if (plan->reserved == 2) k_polstats<CM, true><<<kHotGrid, kWave, chunk_lds_bytes(C, 4, (span + uc.merge) * kChunkPx, kHotStage, uc.partpx), stream>>>(   \
            bin_view(plan, events, workspace), offsets, tnorm, P, plan->H, plan->W, plan->nchunk, hot_cfg(uc), out);   \
    } while (0)
    if (C <= 8) PS_LAUNCH(8); else PS_LAUNCH(16);
#undef PS_LAUNCH
    hot_flip(plan);
    LAUNCH_CHECK("k_polstats");
    return EVREP_OK;
}

int evrep_est_voxel(const evrep_plan *plan, const int32_t *events, const int64_t *offsets, void *workspace,
                    const float *tnorm, int32_t C, const double *segments, int32_t nseg, const uint32_t *buckets,
                    int32_t nbucket, double lo, double hi, float *out, void *stream_) {
    int rc = check_common(plan, events, offsets, workspace);
    if (rc) return rc;
    if (C < 2 || C > kEstMaxBins || !segments || nseg < 1 || !buckets || nbucket < 1 || !(hi > lo) || !out) return EVREP_EINVAL;
    if (plan->total_events > 0 && !tnorm) return EVREP_EINVAL;
    EstParams P;
    memset(&P, 0, sizeof(P));
    P.C = C; P.nseg = nseg; P.nbucket = nbucket;
    P.lo = lo; P.inv_width = (double)nbucket / (hi - lo);
    for (int i = 0; i < C; ++i) P.shift[i] = (float)((double)i / (double)(C - 1));
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const UnitCfg uc = unit_cfg(plan, (size_t)2 * C * 4);
    const int span = uc.span;
    k_est<false><<<SPAN_GRID(span), kWave, chunk_lds_bytes(2 * C, 4, (span + uc.merge) * kChunkPx, uc.stage), stream>>>(
        bin_view(plan, events, workspace), offsets, tnorm, segments, buckets, P, plan->H, plan->W,
        plan->nchunk, uc, out);
    if (plan->reserved == 2) k_est<true><<<kHotGrid, kWave, chunk_lds_bytes(2 * C, 4, (span + uc.merge) * kChunkPx, kHotStage), stream>>>(
        bin_view(plan, events, workspace), offsets, tnorm, segments, buckets, P, plan->H, plan->W,
        plan->nchunk, hot_cfg(uc), out);
    hot_flip(plan);
    LAUNCH_CHECK("k_est");
    return EVREP_OK;
}

// Both read-backs are ONE strided copy of a field of every window's 64-byte WindowMeta (2-D copy: row = window).
static int read_meta_field(const evrep_plan *plan, const void *workspace, size_t field_off, size_t field_bytes,
                           void *dst, hipStream_t stream) {
    const char *meta = static_cast<const char *>(workspace) + plan->off_meta;
    if (plan->reserved == 2) {  // the key-sorted pass leaves the block statistics unmerged: merge them now
        char *ws = const_cast<char *>(static_cast<const char *>(workspace));
        k_window_meta<<<plan->B, kWave, 0, stream>>>(reinterpret_cast<const int64_t *>(ws + plan->off_rowoff),
                                                     reinterpret_cast<const BlockStats *>(ws + plan->off_stats), plan->nblk, plan->chunk == 4096 ? 12 : 13,
                                                     reinterpret_cast<WindowMeta *>(ws + plan->off_meta));
        int rc0 = hip_check(hipGetLastError(), "k_window_meta");
        if (rc0) return rc0;
    }
    int rc = hip_check(hipMemcpy2DAsync(dst, field_bytes, meta + field_off, sizeof(WindowMeta), field_bytes,
                                        (size_t)plan->B, hipMemcpyDeviceToHost, stream), "hipMemcpy2DAsync(meta)");
    if (rc) return rc;
    return hip_check(hipStreamSynchronize(stream), "hipStreamSynchronize");
}

int evrep_copy_window_meta_async(const evrep_plan *plan, const void *workspace, void *meta_out, void *stream_) {
    if (!plan || plan->abi_version != EVREP_ABI_VERSION || !workspace || !meta_out) return EVREP_EINVAL;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const char *meta = static_cast<const char *>(workspace) + plan->off_meta;
    if (plan->reserved == 2) {  // the key-sorted pass leaves the block statistics unmerged: merge them now
        char *ws = const_cast<char *>(static_cast<const char *>(workspace));
        k_window_meta<<<plan->B, kWave, 0, stream>>>(reinterpret_cast<const int64_t *>(ws + plan->off_rowoff),
                                                     reinterpret_cast<const BlockStats *>(ws + plan->off_stats), plan->nblk, plan->chunk == 4096 ? 12 : 13,
                                                     reinterpret_cast<WindowMeta *>(ws + plan->off_meta));
        LAUNCH_CHECK("k_window_meta");
    }
    return hip_check(hipMemcpyAsync(meta_out, meta, (size_t)plan->B * sizeof(WindowMeta), hipMemcpyDeviceToHost, stream),
                     "hipMemcpyAsync(meta)");
}

int evrep_read_status(const evrep_plan *plan, const void *workspace, uint32_t *status, void *stream_) {
    if (!plan || !workspace || !status) return EVREP_EINVAL;
    return read_meta_field(plan, workspace, offsetof(WindowMeta, status), sizeof(uint32_t), status,
                           static_cast<hipStream_t>(stream_));
}

int evrep_read_bbox(const evrep_plan *plan, const void *workspace, int32_t *bbox, void *stream_) {
    if (!plan || !workspace || !bbox) return EVREP_EINVAL;
    static_assert(offsetof(WindowMeta, ymax) - offsetof(WindowMeta, xmin) == 12, "xmin, xmax, ymin, ymax are contiguous");
    int rc = read_meta_field(plan, workspace, offsetof(WindowMeta, xmin), 4 * sizeof(int32_t), bbox,
                             static_cast<hipStream_t>(stream_));
    if (rc) return rc;
    for (int b = 0; b < plan->B; ++b) {  // stored xmin, xmax, ymin, ymax -> reported xmin, ymin, xmax, ymax
        const int32_t t = bbox[4 * b + 1];
        bbox[4 * b + 1] = bbox[4 * b + 2];
        bbox[4 * b + 2] = t;
    }
    return EVREP_OK;
}

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

int evrep_resize_taps(const void *in, int32_t in_dtype, int32_t B, int32_t H, int32_t W, int32_t C, int32_t Ho, int32_t Wo,
                      int32_t T, const int32_t *ystart, const int32_t *ycount, const double *ywt, const int32_t *xstart,
                      const int32_t *xcount, const double *xwt, double scale, int32_t out_dtype, void *out, void *stream_) {
    if (!in || !out || !ystart || !ycount || !ywt || !xstart || !xcount || !xwt) return EVREP_EINVAL;
    if (B <= 0 || B > 65535 || H <= 0 || W <= 0 || C <= 0 || Ho <= 0 || Wo <= 0 || T <= 0) return EVREP_EINVAL;
    if ((in_dtype != EVREP_F64 && in_dtype != EVREP_F32) || (out_dtype != EVREP_F64 && out_dtype != EVREP_F32)) return EVREP_EINVAL;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    ResizeTaps tp;
    tp.ystart = ystart; tp.ycount = ycount; tp.xstart = xstart; tp.xcount = xcount; tp.ywt = ywt; tp.xwt = xwt; tp.T = T;
    const size_t per = (size_t)Ho * Wo * C;
    const dim3 grid((unsigned)((per + kThreads - 1) / kThreads), (unsigned)B);
#define RESIZE_LAUNCH(IN, OUT) \
    k_resize_taps<IN, OUT><<<grid, kThreads, 0, stream>>>(static_cast<const IN *>(in), H, W, C, tp, Ho, Wo, scale, static_cast<OUT *>(out))
    if (in_dtype == EVREP_F64) { if (out_dtype == EVREP_F64) RESIZE_LAUNCH(double, double); else RESIZE_LAUNCH(double, float); }
    else { if (out_dtype == EVREP_F64) RESIZE_LAUNCH(float, double); else RESIZE_LAUNCH(float, float); }
#undef RESIZE_LAUNCH
    LAUNCH_CHECK("k_resize_taps");
    return EVREP_OK;
}

int evrep_probe_store(void *out, size_t bytes, void *stream_) {
    if (!out || (reinterpret_cast<uintptr_t>(out) & 15u)) return EVREP_EINVAL;
    const size_t tiles = bytes / 12288;
    if (tiles == 0 || tiles > 0x7fffffffu) return EVREP_EINVAL;
    k_store_probe<<<(unsigned)tiles, kWave, 8320, static_cast<hipStream_t>(stream_)>>>(static_cast<float *>(out), (int)tiles);
    LAUNCH_CHECK("k_store_probe");
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
