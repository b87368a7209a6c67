// evrep_capi.hip -- the extern "C" surface declared in include/evrep.h, part 1: plans, the binning pass, the read-backs
// (argument checks, workspace carving and kernel launches).  The builders are in evrep_capi_mdes.hip /
// evrep_capi_builders.hip, the Gromov-Wasserstein side in evrep_capi_gwd.hip; evrep_capi_shared.h is what they share.
#define EVREP_TU_CORE 1
#include "evrep_capi_shared.h"

// kernels
#include "evrep_bin.hip"

using namespace evrep;

static thread_local char g_last_error[256] = "";

namespace evrep_host {
char *last_error_buf() { return g_last_error; }
int hip_check(hipError_t e, const char *what) {
    if (e == hipSuccess) return EVREP_OK;
    snprintf(g_last_error, sizeof(g_last_error), "%s: %s", what, hipGetErrorString(e));
    return EVREP_EHIP;
}
// The pixel-sorted stream + chunk offsets + WindowMeta from the runs of k_block_keysort: the column sort, one wave per key.
int column_sort_keys(const evrep_plan *plan, const int32_t *events, const int64_t *offsets, void *workspace, hipStream_t stream) {
    const int NK = plan->H * plan->nchunk;
    k_col_sort_runs<<<dim3((NK + kCsWaves - 1) / kCsWaves, plan->B), kCsWaves * kWave,
                                      (size_t)kCsWaves * col_sort_wave_words(kChunkPx) * 4, stream>>>(
        reinterpret_cast<const int4 *>(events), CWS(Rec, off_sorted1), offsets, CWS(uint32_t, off_table), CWS(BlockStats, off_stats), plan->H, plan->W, plan->nblk,
        plan->nchunk, plan->nchunk, plan->chunk == 4096 ? 12 : 13, 1, WS(Rec, off_sorted2), WS(uint32_t, off_chunkoff),
        WS(WindowMeta, off_meta));
    LAUNCH_CHECK("k_col_sort_runs");
    return EVREP_OK;
}
}  // namespace evrep_host
using evrep_host::hip_check;
using evrep_host::column_sort_keys;

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
    // (r06: up to kCsMaxRuns = 128 block runs -- windows of up to 1 048 576 events: the STREAM builders gather two runs per lane;
    //  the ordered builders, one run per lane, take such a window through the per-key column sort, on demand: ensure_pixel_stream)
    const bool key_sorted = two_kernel && max_events_per_window <= (int64_t)kCsMaxRuns * kBsChunk && NK < 65535 &&
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
    plan->off_cuts = o;    o += up256((size_t)B * kTsCutsBytes);
    // the hot list (evrep_common.h; evrep_builders.hip, run_units): per sublist a count and an exit ticket, then the item slots
    plan->off_scratch = o; o += up256(4 * ((size_t)kHotHdrWords + (size_t)kHotLists * hot_sublist_cap(hot_items_total(total_events))));
    plan->workspace_bytes = o;
    return EVREP_OK;
}

size_t evrep_workspace_bytes(const evrep_plan *plan) { return plan ? plan->workspace_bytes : 0; }

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
    uint32_t *hot = WS(uint32_t, off_scratch);   // the builders' hot list starts empty (k_block_keysort clears it itself)
    if (plan->reserved != 2 && plan->reserved != 3) {
        int rc2 = hip_check(hipMemsetAsync(hot, 0, (size_t)kHotHdrWords * 4, stream), "hipMemsetAsync(hot list)");
        if (rc2) return rc2;
    }
    if (plan->reserved == 2 || plan->reserved == 3) {
        if ((chunk != kBsChunk && chunk != 4096) || nblk > kCsMaxRuns) return EVREP_EINVAL;
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

}  // extern "C"
