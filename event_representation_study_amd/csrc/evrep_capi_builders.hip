// evrep_capi_builders.hip -- the extern "C" surface, part 3: EventStack, TimeSurface, TORE, the voxel grids, the n_imagenet
// accumulators, EST, the resize taps and the store probe.
#define EVREP_TU_BUILDERS 1
#include <cstdlib>
#include "evrep_capi_builders.h"

// LDS padding of a stream launch on SPARSE windows (<= 30 records per unit on average).  EventStack's and TORE's streams are bound by
// their HBM stores there, and two workgroups fewer per CU leave them FASTER -- the store-pacing effect of DESIGN.md 3.2, found again
// when a TORE experiment added 1 KB of LDS.  Library variants alternated (tools/experiments/lib_ab.sh; pad 0 / 1 / 2 / 4 KB, us):
// EventStack 640x480 x 50 000 events 85.9 / 81.9 / 83.5 / 80.1, 1 Mpx 64.0 / 59.1 / 62.6 / 63.3, 1 Mpx edges 64.8 / 62.8 / 62.3 / 61.4;
// TORE 77.9 / 77.7 / 84.3 / 90.4, 60.8 / 58.8 / 62.5 / 68.4, 640x480 edges 84.3 / 74.3 / 80.0 / 86.5.  The voxel grid and the
// accumulators (latency-bound: they want every wave they can get) lose 8-30 % with any padding: none.
static inline size_t stream_pad(const evrep_plan *plan, size_t bytes) {
    const double per_chunk = (double)plan->max_events_per_window / ((double)plan->H * plan->nchunk);
    return per_chunk <= 30.0 ? bytes : 0;
}

extern "C" {

int evrep_event_stack(const evrep_plan *plan, const int32_t *events, const int64_t *offsets, void *workspace,
                      int32_t stack_size, int32_t premap, float scale, float *out, void *stream_) {
    int rc = check_common(plan, events, offsets, workspace);
    if (rc) return rc;
    if (stack_size <= 0 || stack_size > EVREP_MAX_CHANNELS || !out) return EVREP_EINVAL;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (plan->reserved == 2 && !(plan->flags & EVREP_PLAN_X_ESTACK_ORDERED) && plan->W <= 512 * 8) {
        // after the key-sorted pass: the streaming form (k_event_stack_stream) -- one launch, every unit, no hot list
        UnitCfg us = unit_cfg(plan, (size_t)stack_size * 4, 0, true, false);
        us.span = 1; us.merge = 0; us.hold = 0;
        unit_cfg_geometry(us, plan);
        const UnitCfg &uc = us;
        constexpr int kRB = 4;
#define ESS_LAUNCH(CM) k_event_stack_stream<CM, kRB><<<SPAN_GRID(1), kWave, event_stack_stream_lds_bytes(stack_size, kChunkPx, kRB) + stream_pad(plan, 1024), stream>>>( \
            bin_view(plan, events, workspace, true), offsets, plan->H, plan->W, plan->nchunk, us, stack_size, premap, scale, out)
        if (stack_size <= 8) ESS_LAUNCH(8); else if (stack_size <= 12) ESS_LAUNCH(12); else ESS_LAUNCH(16);
#undef ESS_LAUNCH
        LAUNCH_CHECK("k_event_stack_stream");
        return EVREP_OK;
    }
    if (int rc3 = ensure_pixel_stream(plan, events, offsets, workspace, stream)) return rc3;
    const UnitCfg uc = unit_cfg(plan, (size_t)stack_size * 4, 0, true, false);  // float32 pixels of <= 64 B: 128-pixel part tiles (see UnitCfg)
    const int span = uc.span;
    // EventStack reads the last record of a pixel only: a unit beyond the record stage keeps one (rank, polarity) word per pixel
    // (unit_records, LAST) whenever its pixels fit the hot stage -- then the main launch defers nothing and there is no hot
    // launch (and no flip of the hot lists: the current one stays empty)
    const bool last_fits = (size_t)(span + uc.merge) * kChunkPx * sizeof(Rec) <=
                           align16((size_t)uc.partpx * stack_size * 4) + (size_t)uc.stage * sizeof(Rec);
    const bool hot_launch = ks_pass(plan) && !last_fits;
#define ES_LAUNCH(CM)                                                                                              \
    do {                                                                                                           \
    k_event_stack<CM><<<SPAN_GRID(span), kWave, chunk_lds_bytes(stack_size, 4, (span + uc.merge) * kChunkPx, uc.stage, uc.partpx), stream>>>(   \
        bin_view(plan, events, workspace), offsets, plan->H, plan->W, plan->nchunk, uc, stack_size, premap, scale, out);          \
    if (hot_launch) k_event_stack<CM, true><<<kHotGrid, kWave, chunk_lds_bytes(stack_size, 4, (span + uc.merge) * kChunkPx, kHotStage, uc.partpx), stream>>>(  \
        bin_view(plan, events, workspace), offsets, plan->H, plan->W, plan->nchunk, hot_cfg(uc), stack_size, premap, scale, out); \
    } while (0)
    if (stack_size <= 8) ES_LAUNCH(8); else if (stack_size <= 12) ES_LAUNCH(12); else ES_LAUNCH(16);
#undef ES_LAUNCH
    LAUNCH_CHECK("k_event_stack");
    return EVREP_OK;
}

// where the time surface's stream beats its ordered builder (measured, r06, float64, build launch in us, ordered / stream): Gen1
// shape 79.8 / 67.1 (circle 75.8 / 73.8, edges 97.3 / 70.0), 8 x 500 000 events at 640x480 124.0 / 79.3; sparse windows lose --
// 640x480 x 50 000 events (21 records per unit) 177 / 218, 1 Mpx x 200 000 141 / 167 (12 KB of 64-bit words to zero and to scan
// per unit, 11 waves per CU), so they keep k_time_surface, which is near the store roof there
static bool ts_stream_wins(const evrep_plan *plan, int32_t out_dtype) {
    const double per_chunk = (double)plan->max_events_per_window / ((double)plan->H * plan->nchunk);
    (void)out_dtype;
    return per_chunk > 28.0;
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
    if (plan->reserved == 2 && tf == nullptr && !(premap & 2) && ((plan->flags & EVREP_PLAN_X_TS_STREAM) || ts_stream_wins(plan, out_dtype)) && !(plan->flags & EVREP_PLAN_X_TS_ORDERED) && plan->W <= 512 * 8) {
        // after the key-sorted pass: the streaming form (k_time_surface_stream) -- one launch, every unit, no hot list
        UnitCfg us = unit_cfg(plan, (size_t)1 << 20, 0, false, false);
        us.span = 1; us.merge = 0; us.hold = 0;
        unit_cfg_geometry(us, plan);
        const UnitCfg &uc = us;
        constexpr int kRB = 4;
#define TSS_LAUNCH(T, CM) k_time_surface_stream<T, CM, kRB><<<SPAN_GRID(1), kWave, time_surface_stream_lds_bytes(slices, kChunkPx, sizeof(T), kRB), stream>>>( \
            bin_view(plan, events, workspace, true), offsets, cuts, plan->H, plan->W, plan->nchunk, us, slices, tau, premap, scale, static_cast<T *>(out))
        if (out_dtype == EVREP_F64) { if (slices <= 6) TSS_LAUNCH(double, 12); else TSS_LAUNCH(double, 16); }
        else { if (slices <= 6) TSS_LAUNCH(float, 12); else TSS_LAUNCH(float, 16); }
#undef TSS_LAUNCH
        LAUNCH_CHECK("k_time_surface_stream");
        return EVREP_OK;
    }
    if (int rc3 = ensure_pixel_stream(plan, events, offsets, workspace, stream)) return rc3;
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
        hot_launch = ks_pass(plan) &&                                                                                \
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
    if (plan->reserved == 2 && tf == nullptr && scale >= 0.0f && !(plan->flags & EVREP_PLAN_X_TORE_ORDERED) && plan->W <= 512 * 8) {
        // after the key-sorted pass, integer times: the streaming form (k_tore_stream) -- one launch, every unit and frame, no hot list
        UnitCfg us = unit_cfg(plan, (size_t)2 * k * 4, 1);
        us.span = 1; us.merge = 0; us.hold = 0;
        unit_cfg_geometry(us, plan);
        const UnitCfg &uc = us;
        constexpr int kRB = 4;
        k_tore_stream<kRB><<<SPAN_GRID(1), kWave, tore_stream_lds_bytes(k, kChunkPx, kRB) + stream_pad(plan, 1024), stream>>>(
            reinterpret_cast<const int4 *>(events), bin_view(plan, events, workspace, true), offsets, sample_times, plan->H, plan->W,
            plan->nchunk, us, k, frame_mode, scale, out);
        LAUNCH_CHECK("k_tore_stream");
        return EVREP_OK;
    }
    if (int rc3 = ensure_pixel_stream(plan, events, offsets, workspace, stream)) return rc3;
    const UnitCfg uc = unit_cfg(plan, (size_t)2 * k * 4, 1);   // the shifted frame straddles one more chunk
    const int span = uc.span;
    // dense windows: the main launch runs the order-free cascade itself (k_tore, SM), as k_polstats does
    const double per_chunk_t = (double)plan->max_events_per_window / ((double)plan->H * plan->nchunk);
    const bool sweep_main = ks_pass(plan) && span == 1 && per_chunk_t > 150.0 && !(plan->flags & 4096) && tf == nullptr;
    UnitCfg um = uc;
    if (sweep_main) {
        const size_t need = (size_t)kChunkPx * 2 * k * 4 + 1024, have = align16((size_t)kPartPx * 2 * k * 4) + align16((size_t)EVREP_MAX_CHANNELS * 4);
        const int st = (int)((need > have ? need - have : 0) + 15) / 16;
        if (um.stage < st) um.stage = (st + 63) & ~63;
    }
#define TORE_LAUNCH(CM)                                                                                             \
    do {                                                                                                            \
    if (sweep_main) {                                                                                               \
        k_tore<CM, false, true><<<SPAN_GRID(span), kWave, chunk_lds_bytes(2 * k, 4, (span + 1) * kChunkPx, um.stage), stream>>>(   \
            reinterpret_cast<const int4 *>(events), bin_view(plan, events, workspace), offsets, sample_times, tf, sample_times_f, \
            plan->H, plan->W, plan->nchunk, um, k, frame_mode, scale, out);                                         \
    } else {                                                                                                        \
    k_tore<CM><<<SPAN_GRID(span), kWave, chunk_lds_bytes(2 * k, 4, (span + 1) * kChunkPx, uc.stage), stream>>>(          \
        reinterpret_cast<const int4 *>(events), bin_view(plan, events, workspace), offsets, sample_times, tf, sample_times_f,    \
        plan->H, plan->W, plan->nchunk, uc, k, frame_mode, scale, out);                                             \
    }                                                                                                               \
    UnitCfg hc = hot_cfg(uc);                                                                                       \
    if (uc.xflags & 6) hc.stage = hot_sweep_stage((size_t)(span + uc.merge) * kChunkPx * 2 * k * 4, 512, (size_t)kPartPx * 2 * k * 4);   /* whole units by the order-free sweep: room for their words */ \
    /* (none behind the sweeping main launch: it defers nothing -- units with a shifted frame or unsorted timestamps are emitted from their slot by the main wave) */ \
    if (ks_pass(plan) && !sweep_main) k_tore<CM, true><<<kHotGrid, kWave, chunk_lds_bytes(2 * k, 4, (span + 1) * kChunkPx, hc.stage), stream>>>(     \
        reinterpret_cast<const int4 *>(events), bin_view(plan, events, workspace), offsets, sample_times, tf, sample_times_f,    \
        plan->H, plan->W, plan->nchunk, hc, k, frame_mode, scale, out);                                    \
    } while (0)
    if (k <= 6) TORE_LAUNCH(12); else TORE_LAUNCH(16);
#undef TORE_LAUNCH
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
    if (plan->reserved == 2 && !(plan->flags & EVREP_PLAN_X_VOXEL_ORDERED) && plan->W <= 512 * 8) {
        // after the key-sorted pass: the streaming form (k_voxel_stream) -- one launch, every unit, no hot list
        UnitCfg us = unit_cfg(plan, (size_t)bins * 8);
        us.span = 1; us.merge = 0; us.hold = 0;
        // (the stream has no record stage: the field carries the burst threshold) a unit of more records than this AND of more than three
        // times its window's average is handed to k_voxel_hot (grids of up to kVhMaxBins bins: its per-wave cell counters must fit LDS)
        const bool hot = bins <= kVhMaxBins;
        // (measured, r06, build in us, stream only / threshold 768 with 16 waves per unit: Gen1 circle 148 / 126, 640x480 circle 132 / 118,
        //  1 Mpx circle 233 / 155; the edge streams, whose largest units hold ~1 000 records at 4-5 per batch and pixel, 80 / 93 and 93 / 91:
        //  a unit of that size is cheaper inside the main launch's tail than in a launch behind it; a threshold of 1 536 loses on the circles --
        //  640x480 155 us: the units of 768-1 536 records then make the main launch's tail AND the hot launch still runs behind it)
#ifndef EVREP_VOXEL_HOT_MIN
#define EVREP_VOXEL_HOT_MIN 768
#endif
#ifndef EVREP_VOXEL_RB
#define EVREP_VOXEL_RB 4
#endif
        us.stage = hot ? EVREP_VOXEL_HOT_MIN : 0x7fffffff;
        unit_cfg_geometry(us, plan);
        const UnitCfg &uc = us;
        constexpr int kRB = EVREP_VOXEL_RB;
        k_voxel_stream<kRB><<<SPAN_GRID(1), kWave, voxel_stream_lds_bytes(bins, kChunkPx, kRB), stream>>>(
            reinterpret_cast<const int4 *>(events), bin_view(plan, events, workspace, true), offsets, plan->H, plan->W, plan->nchunk, us,
            bins, mode, scale, t_range, tnorm, out);
        if (hot) {
            const size_t lds = voxel_hot_lds_bytes(bins, kChunkPx);
            if (lds > 64 * 1024) {   // per (function, device) opt-in, renewed per launch (see k_block_keysort)
                if (hipFuncSetAttribute(reinterpret_cast<const void *>(&k_voxel_hot), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024) != hipSuccess) {
                    (void)hipGetLastError();
                    return EVREP_EHIP;
                }
            }
            k_voxel_hot<<<kVhGrid, kVhThreads, lds, stream>>>(
                reinterpret_cast<const int4 *>(events), bin_view(plan, events, workspace, true), offsets, plan->H, plan->W, plan->nchunk, us,
                bins, mode, scale, t_range, tnorm, out);
        }
        LAUNCH_CHECK("k_voxel_stream");
        return EVREP_OK;
    }
    if (int rc3 = ensure_pixel_stream(plan, events, offsets, workspace, stream)) return rc3;
    const UnitCfg uc = unit_cfg(plan, (size_t)bins * 8);
    const int span = uc.span;
#define VOXEL_LAUNCH(CM)                                                                                         \
    do {                                                                                                         \
    /* + the lanes' own rows of running sums: 64 x bins float64 behind the carve (k_voxel's reduce) */              \
    k_voxel<CM><<<SPAN_GRID(span), kWave, chunk_lds_bytes(bins, 8, (span + uc.merge) * kChunkPx, uc.stage) + (size_t)kWave * bins * 8, stream>>>(              \
        reinterpret_cast<const int4 *>(events), bin_view(plan, events, workspace), offsets, plan->H, plan->W, plan->nchunk, uc, \
        bins, mode, scale, t_range, tnorm, out);                                                                 \
    if (ks_pass(plan)) k_voxel<CM, true><<<kHotGrid, kWave, chunk_lds_bytes(bins, 8, (span + uc.merge) * kChunkPx, kHotStage) + (size_t)kWave * bins * 8, stream>>>(         \
        reinterpret_cast<const int4 *>(events), bin_view(plan, events, workspace), offsets, plan->H, plan->W, plan->nchunk,   \
        hot_cfg(uc), bins, mode, scale, t_range, tnorm, out);                                                    \
    } while (0)
    if (bins <= 8) VOXEL_LAUNCH(8); else VOXEL_LAUNCH(16);
#undef VOXEL_LAUNCH
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
    UnitCfg uc = unit_cfg(plan, (size_t)C * 4, 0, true);  // float32 pixels of <= 64 B: 128-pixel part tiles (see UnitCfg)
    const int span = uc.span;
    // every statistic is order-free: two-chunk units (sparse windows) go to the hot launch whole as well -- measured, r05b: 1 Mpx
    // circle windows 230 -> 129 us, the other clustered streams within 3 % (TORE, whose sweep is heavier and cannot be sliced: 124
    // -> 142 us on the config 2 circle, so its two-chunk units stay)
    if (span == 2) uc.xflags |= 2;
    if (plan->reserved == 2 && !(plan->flags & EVREP_PLAN_X_POLSTATS_ORDERED) && plan->W <= 512 * 8) {
        // after the key-sorted pass: the streaming form (k_polstats_stream) -- one launch, every unit, no hot list
        UnitCfg us = uc;
        us.span = 1; us.merge = 0; us.hold = 0;
        unit_cfg_geometry(us, plan);
        const UnitCfg &uc = us;
        constexpr int kRB = 4;
        bool any_exp = false;
        for (int c = 0; c < C; ++c) any_exp = any_exp || stat[c] == EVREP_PS_EXP;
        const bool k32 = !any_exp && C <= 8;   // float32 extremes: exact for every statistic but EXP (k_polstats_stream)
#define PSS_LAUNCH(CM, K32) k_polstats_stream<CM, kRB, K32><<<SPAN_GRID(1), kWave, polstats_stream_lds_bytes(kChunkPx, kRB, K32), stream>>>( \
                bin_view(plan, events, workspace, true), offsets, tnorm, P, plan->H, plan->W, plan->nchunk, us, out)
        if (k32) PSS_LAUNCH(8, true); else if (C <= 8) PSS_LAUNCH(8, false); else PSS_LAUNCH(16, false);
#undef PSS_LAUNCH
        LAUNCH_CHECK("k_polstats_stream");
        return EVREP_OK;
    }
    if (int rc3 = ensure_pixel_stream(plan, events, offsets, workspace, stream)) return rc3;
    // dense windows (every unit beyond the record stage): the main launch sweeps order-free itself (k_polstats, SM) with a stage
    // that holds the unit's fourteen words per pixel; nothing is deferred and there is no hot launch
    const double per_chunk_ps = (double)plan->max_events_per_window / ((double)plan->H * plan->nchunk);
    const bool sweep_main = ks_pass(plan) && span == 1 && per_chunk_ps > 150.0 && !(plan->flags & 4096);   // (measured: 8 x 500 000 events at 640x480 74 -> 63 us; at 250 000 -- 104 per unit, most of them inside the stage -- 52 -> 58: the instance's budget is the sweep's); 4096: EVREP_X_NO_SWEEP_MAIN
    if (sweep_main) {
        const size_t need = (size_t)kChunkPx * 14 * 4 + 1024, have = align16((size_t)uc.partpx * C * 4) + align16((size_t)EVREP_MAX_CHANNELS * 4);
        const int st = (int)((need > have ? need - have : 0) + 15) / 16;
        if (uc.stage < st) uc.stage = (st + 63) & ~63;
    }
#define PS_LAUNCH(CM)                                                                                                 \
    do {                                                                                                              \
        if (sweep_main) {                                                                                             \
            k_polstats<CM, false, true><<<SPAN_GRID(span), kWave, chunk_lds_bytes(C, 4, (span + uc.merge) * kChunkPx, uc.stage, uc.partpx), stream>>>(   \
                bin_view(plan, events, workspace), offsets, tnorm, P, plan->H, plan->W, plan->nchunk, uc, out);        \
            break;                                                                                                    \
        }                                                                                                             \
        k_polstats<CM><<<SPAN_GRID(span), kWave, chunk_lds_bytes(C, 4, (span + uc.merge) * kChunkPx, uc.stage, uc.partpx), stream>>>(   \
            bin_view(plan, events, workspace), offsets, tnorm, P, plan->H, plan->W, plan->nchunk, uc, out);            \
        /* one-chunk units of sparse windows go to the hot launch whole (order-free sweep there): a larger stage for their words */ \
        UnitCfg hc = hot_cfg(uc);                                                                                     \
        if (uc.xflags & 2) hc.stage = hot_sweep_stage((size_t)(span + uc.merge) * kChunkPx * 14 * 4, 512, (size_t)uc.partpx * C * 4);   \
        if (ks_pass(plan)) k_polstats<CM, true><<<kHotGrid, kWave, chunk_lds_bytes(C, 4, (span + uc.merge) * kChunkPx, hc.stage, uc.partpx), stream>>>(   \
            bin_view(plan, events, workspace), offsets, tnorm, P, plan->H, plan->W, plan->nchunk, hc, out);   \
    } while (0)
    if (C <= 8) PS_LAUNCH(8); else PS_LAUNCH(16);
#undef PS_LAUNCH
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
    if (int rc3 = ensure_pixel_stream(plan, events, offsets, workspace, stream)) return rc3;
    const UnitCfg uc = unit_cfg(plan, (size_t)2 * C * 4);
    const int span = uc.span;
    k_est<false><<<SPAN_GRID(span), kWave, chunk_lds_bytes(2 * C, 4, (span + uc.merge) * kChunkPx, uc.stage), stream>>>(
        bin_view(plan, events, workspace), offsets, tnorm, segments, buckets, P, plan->H, plan->W,
        plan->nchunk, uc, out);
    if (ks_pass(plan)) k_est<true><<<kHotGrid, kWave, chunk_lds_bytes(2 * C, 4, (span + uc.merge) * kChunkPx, kHotStage), stream>>>(
        bin_view(plan, events, workspace), offsets, tnorm, segments, buckets, P, plan->H, plan->W,
        plan->nchunk, hot_cfg(uc), out);
    LAUNCH_CHECK("k_est");
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

}  // extern "C"
