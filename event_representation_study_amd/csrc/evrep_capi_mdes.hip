// evrep_capi_mdes.hip -- the extern "C" surface, part 2: MixedDensityEventStack / Operations / ERGO-12 (k_mdes).
#define EVREP_TU_MDES 1
#include <cstdlib>
#include "evrep_capi_builders.h"

extern "C" {

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
    if (ergo && plan->reserved == 2 && !(plan->flags & EVREP_PLAN_X_MDES_ORDERED) && plan->W <= 512 * 8) {
        // after the key-sorted pass: ERGO-12 as a stream (k_mdes_stream) -- one launch, every unit, no hot list -- where it wins
        // (measured, r06, build launch(es) in us, ordered / stream; the stream's fixed cost per wave -- 10 KB of state to zero,
        // twelve values per non-empty pixel from scratch, 12-14 KB of LDS = 11-13 waves per CU -- loses on sparse uniform windows,
        // where the ordered builder's sparse emit is at the store roof):
        //   float32: Gen1 shape 55.6 / 56.5 uniform, 78 / 78 circle, 97 / 67 edges; 8 x 500 000 events 82 / 71;
        //            640x480 x 50 000 (21 records per unit) 82 / 147, 1 Mpx x 200 000 77 / 116 (circle 222 / 129)
        //   float64: Gen1 shape 64.8 / 67.5, circle 73 / 86, edges 86 / 77; 8 x 500 000 events 99 / 85; 640x480 x 50 000 152 / 188
        const double per_chunk_ms = (double)plan->max_events_per_window / ((double)plan->H * plan->nchunk);
        const bool use = (plan->flags & EVREP_PLAN_X_MDES_STREAM) || (out_dtype == EVREP_F32 ? per_chunk_ms > 28.0 : per_chunk_ms > 100.0);
        if (use) {
            UnitCfg us = unit_cfg(plan, (size_t)C * (out_dtype == EVREP_F64 ? 8 : 4));
            us.span = 1; us.merge = 0; us.hold = 0;
            unit_cfg_geometry(us, plan);
            const UnitCfg &uc = us;
            constexpr int kRB = 4;
            if (out_dtype == EVREP_F64)
                k_mdes_stream<double, kRB><<<SPAN_GRID(1), kWave, mdes_stream_lds_bytes(kChunkPx, 8, kRB), stream>>>(
                    bin_view(plan, events, workspace, true), offsets, plan->H, plan->W, plan->nchunk, us, scale, static_cast<double *>(out));
            else
                k_mdes_stream<float, kRB><<<SPAN_GRID(1), kWave, mdes_stream_lds_bytes(kChunkPx, 4, kRB), stream>>>(
                    bin_view(plan, events, workspace, true), offsets, plan->H, plan->W, plan->nchunk, us, scale, static_cast<float *>(out));
            LAUNCH_CHECK("k_mdes_stream");
            return EVREP_OK;
        }
    }
    if (int rc3 = ensure_pixel_stream(plan, events, offsets, workspace, stream)) return rc3;
    UnitCfg uc = unit_cfg(plan, (size_t)C * (out_dtype == EVREP_F64 ? 8 : 4));
    if ((plan->flags & EVREP_PLAN_X_SPAN2) && plan->nchunk >= 2) { uc.span = 2; uc.stage = 128; unit_cfg_geometry(uc, plan); }
    const int span = uc.span;
    const bool pace_auto = plan->pacing < 0 && out_dtype == EVREP_F64 && C * 8 >= 64;   // the store-bound instances
#define MDES_LAUNCH(T, DESC)                                                                                          \
    do {                                                                                                              \
        const size_t lds_ = chunk_lds_bytes(C, sizeof(T), (span + uc.merge) * kChunkPx, uc.stage, uc.partpx);                       \
        /* the float64 ERGO-12 instance defers nothing (its split path, mdes_unit): no hot launch behind it; the float32 one   \
           hands hot units to its hot launch whole (Split, IN_HOT): a stage of kHotSplitStage records there -- and, r06, units of \
           >= kErgoCoopMin records to a cooperative launch of sixteen waves per unit (k_mdes_coop; UnitCfg::xflags bit 8) */       \
        const bool hot_launch = ks_pass(plan) && !(MdesIsErgo12<DESC>::value && sizeof(T) == 8);                                       \
        const bool coop = hot_launch && MdesIsErgo12<DESC>::value && sizeof(T) == 4 && !(plan->flags & EVREP_PLAN_X_MDES_NO_COOP);      \
        if (coop) uc.xflags |= 8;                                                                                                       \
        if (pace_auto) uc.hold = auto_hold(plan, reinterpret_cast<const void *>(&k_mdes<T, DESC>), lds_, span, (size_t)C * sizeof(T), uc.merge); \
        k_mdes<T, DESC><<<SPAN_GRID(span), kWave, lds_, stream>>>(bin_view(plan, events, workspace), offsets, P, plan->H, plan->W,   \
                                                                  plan->nchunk, uc, scale, static_cast<T *>(out));           \
        UnitCfg hc = hot_cfg(uc);                                                                                             \
        if (MdesIsErgo12<DESC>::value) hc.stage = hot_sweep_stage((size_t)(span + uc.merge) * kChunkPx * 7 * 4, 4096, (size_t)uc.partpx * C * sizeof(T));                                                             \
        if (coop) {                                                                                                                     \
            const size_t cl = mdes_coop_lds_bytes((span + uc.merge) * kChunkPx);                                                        \
            if (hipFuncSetAttribute(reinterpret_cast<const void *>(&k_mdes_coop), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024) != hipSuccess) { (void)hipGetLastError(); return EVREP_EHIP; } \
            k_mdes_coop<<<kMcGrid, kMcThreads, cl, stream>>>(bin_view(plan, events, workspace), offsets, plan->H, plan->W, plan->nchunk, uc, scale, reinterpret_cast<float *>(out)); \
        }                                                                                                                               \
        if (hot_launch) k_mdes<T, DESC, true><<<kHotGrid, kWave, chunk_lds_bytes(C, sizeof(T), (span + uc.merge) * kChunkPx, hc.stage, uc.partpx), stream>>>(  \
            bin_view(plan, events, workspace), offsets, P, plan->H, plan->W, plan->nchunk, hc, scale, static_cast<T *>(out)); \
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
    LAUNCH_CHECK("k_mdes");
    return EVREP_OK;
}

int evrep_optimized(const evrep_plan *plan, const int32_t *events, const int64_t *offsets, void *workspace,
                    double scale, int32_t out_dtype, void *out, void *stream) {
    int32_t win[12], func[12], agg[12];
    for (int c = 0; c < 12; ++c) { win[c] = Ergo12Table::kWin[c]; func[c] = Ergo12Table::kFunc[c]; agg[c] = Ergo12Table::kAgg[c]; }
    return evrep_mdes(plan, events, offsets, workspace, 12, win, func, agg, scale, out_dtype, out, stream);
}

}  // extern "C"
