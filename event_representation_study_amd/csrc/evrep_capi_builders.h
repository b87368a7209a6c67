// evrep_capi_builders.h -- host-side launch helpers shared by the builder translation units (evrep_capi_mdes.hip,
// evrep_capi_builders.hip): the view of the binning pass, the unit configuration, store pacing.
#pragma once
#include "evrep_capi_shared.h"
#include "evrep_bin.hip"
#include "evrep_builders.hip"

using namespace evrep;

#define BUILDER_GRID dim3(plan->nchunk, plan->H, plan->B)

// The key-sorted pass leaves block RUNS (plan->reserved == 2).  The ORDERED builders gather a unit's records with one lane per run:
// up to kBsMaxBlocks = 64 runs (ks_pass); the STREAM builders (r06) take two runs per lane: up to kCsMaxRuns = 128 -- windows of up
// to 1 048 576 events.  A window of more than 64 runs reaches an ordered builder through the per-key column sort, run on demand in
// front of that builder's launch (ensure_pixel_stream): the builder then reads the pixel-sorted stream as after the classic passes.
static inline bool ks_pass(const evrep_plan *plan) { return plan->reserved == 2 && plan->nblk <= kBsMaxBlocks; }

// what the builders read of the binning pass (see BinView in evrep_builders.hip); runs: the view of a stream builder
static inline BinView bin_view(const evrep_plan *plan, const int32_t *events, void *workspace, bool runs = false) {
    BinView bv;
    bv.ev = reinterpret_cast<const int4 *>(events);
    bv.fused = (runs ? plan->reserved == 2 : ks_pass(plan)) ? 1 : 0;
    bv.sorted = bv.fused ? CWS(Rec, off_sorted1) : CWS(Rec, off_sorted2);
    bv.chunk_off = CWS(uint32_t, off_chunkoff);
    bv.table = CWS(uint32_t, off_table);
    bv.stats = CWS(BlockStats, off_stats);
    bv.meta = CWS(WindowMeta, off_meta);
    bv.spill = WS(Rec, off_sorted2);
    bv.nblk = plan->nblk;
    bv.hot = WS(uint32_t, off_scratch);
    bv.hot_cap = hot_items_total(plan->total_events);
    bv.stats_rw = WS(BlockStats, off_stats);
    bv.placed_pool = reinterpret_cast<double *>(static_cast<char *>(workspace) + plan->off_sorted1 + align16((size_t)plan->total_events * 8));   // the key-sorted pass moves 8-byte records: the upper half of sorted1 (8 bytes per event, fully) is idle
    bv.chunk_shift = plan->chunk == 4096 ? 12 : 13;  // only read after the key-sorted pass
#ifdef EVREP_TIMING
    // 8 slots per builder wave: behind the 8-byte records of the key-sorted pass (the upper half of sorted1 is idle; sorted2 is
    // the spill stream of the warm / hot units), in sorted1 under the classic passes
    bv.dbg = reinterpret_cast<unsigned long long *>(static_cast<char *>(workspace) + plan->off_sorted1 +
                                                    (bv.fused ? up256((size_t)plan->total_events * 8) : 0));
    bv.dbg_wave = nullptr;
#endif
    return bv;
}

// After the key-sorted pass: the pixel-sorted stream + chunk offsets + WindowMeta, for the consumers that walk
// them directly (k_voxel_subpixel).  The column sort of the two-kernel pass, reading a row's runs chunk by chunk.
static inline int ensure_column_sorted(const evrep_plan *plan, const int32_t *events, const int64_t *offsets, void *workspace, hipStream_t stream) {
    if (plan->reserved != 2) return EVREP_OK;
    return evrep_host::column_sort_keys(plan, events, offsets, workspace, stream);
}
// in front of an ORDERED builder's launches: a key-sorted window of more than 64 runs gets its pixel-sorted stream now
static inline int ensure_pixel_stream(const evrep_plan *plan, const int32_t *events, const int64_t *offsets, void *workspace, hipStream_t stream) {
    if (plan->reserved != 2 || ks_pass(plan)) return EVREP_OK;
    return evrep_host::column_sort_keys(plan, events, offsets, workspace, stream);
}

// The unit of one builder wave.  span = 128-pixel chunks it takes: 2 for small pixels (float32 x 12, float64 x 5 ...) on
// sparse windows (<= 30 records per chunk on average, so a 256-pixel unit still fits the one-lane-per-non-empty-pixel
// fast path), else 1.  stage = records its LDS stage holds: 64 for one-chunk units, 128 for wider ones (they hold ~65
// records on the sparse windows they are chosen for).
// the division-free unit decode of the builder waves (UnitCfg::nunit ...): after every change of span / merge
static inline void unit_cfg_geometry(UnitCfg &uc, const evrep_plan *plan) {
    uc.nunit = units_per_row(plan->nchunk, uc.span, uc.merge);
    fastdiv_make((uint32_t)uc.nunit, uc.nunit_m, uc.nunit_sh);
    fastdiv_make((uint32_t)plan->H, uc.h_m, uc.h_sh);
}
static inline UnitCfg unit_cfg(const evrep_plan *plan, size_t pixel_bytes, int extra_chunks = 0, bool wide_part = false, bool deep_stage = true) {
    UnitCfg uc;
    const double per_chunk = (double)plan->max_events_per_window / ((double)plan->H * plan->nchunk);
    uc.span = (pixel_bytes * kChunkPx > 8192 || plan->nchunk < 2) ? 1 : (per_chunk <= 30.0 ? 2 : 1);  // a 128-pixel chunk of >= 8 KB stays alone
    // denser units (the reference's own Gen1 shape, 304x240 x 50 000 events: ~69 records per unit) are ordered inside LDS
    // in two register batches: a 128-record stage; the dense windows of the classic passes stage 256 (stage_classic)
    // (deep_stage = false: EventStack only reads a segment's last records, TimeSurface measured slower with it)
    uc.stage = (deep_stage && !ks_pass(plan) && per_chunk > kDeepStageMinPerUnit) ? 256 : ((uc.span + extra_chunks > 1 || per_chunk > 28.0 || (plan->flags & 128)) ? 128 : 64);
    if ((plan->flags & 512) && ks_pass(plan) && uc.span == 1) uc.stage = 64;   // experiment (EVREP_X_STAGE64): units of > 64 records leave the two-batch path
    uc.partpx = (wide_part && per_chunk <= 30.0) ? 2 * kPartPx : kPartPx;  // sparse windows only: dense ones lose 5 % with it
    uc.hold = plan->pacing > 0 ? plan->pacing : 0;  // automatic pacing is decided per launch (auto_hold)
    // a short tail chunk (<= 64 of 128 pixels: Gen1's 304-pixel rows end in 48) rides with the row's last unit (UnitCfg::merge);
    // TORE's units live in the OUTPUT frame and keep their own geometry (extra_chunks)
    const int tail = plan->W % kChunkPx;
    // (an experiment switch, off by default: measured at the Gen1 shape the 176-pixel unit loses the sparse emit -- ~95 records
    //  in ~80 pixels -- and its three-part tile sequence costs more than the 48-pixel tail unit it saves: ERGO-12 68 -> 82 us)
    uc.merge = (extra_chunks == 0 && tail != 0 && tail <= kChunkPx / 2 && plan->nchunk >= 2 && (plan->flags & EVREP_PLAN_X_TAIL_MERGE)) ? 1 : 0;
    uc.xflags = ((uc.span == 1 && (per_chunk <= 90.0 || (plan->flags & 2048))) || ((plan->flags & 1024) && uc.span > 1)) ? 2 : 0;   // see UnitCfg::xflags (2048: EVREP_X_HANDOVER_DENSE, experiment)
    if (uc.span > 1 && !(uc.xflags & 2) && !(plan->flags & 8192)) uc.xflags |= 4;   // monsters only (8192: EVREP_X_NO_MONSTER_HANDOVER, experiment)
    unit_cfg_geometry(uc, plan);
    return uc;
}
#define SPAN_GRID(span) dim3(units_per_row(plan->nchunk, (span), uc.merge), plan->H, plan->B)
// records of the LDS stage of a hot launch whose waves take whole units by an order-free sweep: tile + background + stage have to hold
// the unit's words (+ `list_bytes` of kept records); never less than the ordinary hot stage (the ordered hot pieces use it)
static inline int hot_sweep_stage(size_t words_bytes, size_t list_bytes, size_t tile_bytes) {
    const size_t have = align16(tile_bytes) + align16((size_t)EVREP_MAX_CHANNELS * 8), need = words_bytes + list_bytes;
    const int st = need > have ? (int)((need - have + 15) / 16) : 0;
    const int r = (st + 63) & ~63;
    return r > kHotStage ? r : kHotStage;
}
// the hot launch behind a builder launch (run_units): the same unit numbering (span), a stage of kHotStage records, no pacing
// (only launched after the key-sorted pass: the main launches of the classic passes defer nothing)
static inline UnitCfg hot_cfg(UnitCfg uc) { uc.stage = kHotStage; uc.hold = 0; return uc; }

// Automatic store pacing (plan->pacing == -1) of a builder instance whose launch is bound by its HBM writes on sparse
// windows.  The waves resident on a CU offer U x unit_bytes every wave lifetime; on most placements of a ~1 GB output tensor
// MI355X serves ~5.7 TB/s when that offer exceeds ~7 TB/s (19 waves x 12 KiB ready after ~7 us: 8.5 TB/s), and
// 6.4-6.9 TB/s when it stays just below (NOTES.md 3.2, tools/experiments/pacing.py: the float64 12-channel builder takes
// 168 us unpaced, 148 us held at 7.0 us, 155 us held at 7.5 us -- a cliff on the short side, a slope on the long side, so
// the hold sits 2 % beyond the knee).  The hold scales with the bytes the CU's resident waves own.
static inline int auto_hold(const evrep_plan *plan, const void *kernel, size_t lds_bytes, int span, size_t pixel_bytes, int merge) {
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
