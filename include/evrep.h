/*
 * evrep.h -- C ABI of libevrep.so, the MI355X (gfx950) event-representation engine.
 *
 * The reference (uzh-rpg/event_representation_study) has no FFI: its boundary is a set of
 * importable Python names.  Each entry point below replaces the hot
 * loop behind one of those names; the Python host mirror under
 * event_representation_study_amd/representations/ binds them with ctypes (INTEGRATION.md shows
 * the stub a reference maintainer would add).  Citations are relative to the reference tree.
 *
 * Conventions
 *   - plain C types only; every pointer marked DEVICE is a HIP device pointer, `stream` is a
 *     hipStream_t passed as void* (NULL = the null stream);
 *   - no global state (nothing is remembered between calls, no environment variable is read, any number of host
 *     threads may drive any number of devices and streams), no allocation: the caller owns a workspace of
 *     evrep_workspace_bytes();
 *   - every call is asynchronous on `stream` and returns an EVREP_* status (launch errors
 *     included); data-dependent failures the reference reports as Python exceptions are
 *     recorded per window in the workspace and read back with evrep_read_status();
 *   - events are int32 rows [x, y, t, p] (the '<i4' structured array the reference adapters
 *     build, ev-YOLOv6/yolov6/data/gen1_2yolo.py:567-571), time-sorted, B windows concatenated,
 *     window b = rows [offsets[b], offsets[b+1]);
 *   - outputs are dense, channel-last (B, H, W, C), written exactly once (zero fill fused).
 */
#ifndef EVREP_H_
#define EVREP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EVREP_ABI_VERSION 3

/* return codes */
#define EVREP_OK 0
#define EVREP_EINVAL 1      /* bad argument (sizes, NULL pointers, unsupported C / k / bins) */
#define EVREP_EWORKSPACE 2  /* workspace too small */
#define EVREP_EHIP 3        /* a HIP launch failed; see evrep_last_hip_error() */
#define EVREP_ENOTBINNED 4  /* builder called on a plan that has not been binned */

/* per-window status bits (evrep_read_status) */
#define EVREP_ST_EMPTY 1u      /* window has no events (reference: ValueError on t.min()) */
#define EVREP_ST_OOB 2u        /* some x + y*W outside [0, H*W) (reference: IndexError in put / zero channel in MDES) */
#define EVREP_ST_UNSORTED 4u   /* timestamps not ascending: evrep_mdes / evrep_optimized, evrep_event_stack, evrep_time_surface (premap bit 1) and evrep_tore work in array order as the reference does; the other builders' tensors are undefined */
#define EVREP_ST_FLAT_TIME 8u  /* t[-1] == t[0] (reference divides by zero) */
#define EVREP_ST_HOT_OVERFLOW 16u /* a builder could not queue a unit of a clustered window (the workspace's hot list was full): pixels of the window are unwritten */

/* MDES function / aggregation codes (representation_search/operations.py:42-87, :16-34) */
enum evrep_func { EVREP_F_TIMESTAMP = 0, EVREP_F_POLARITY, EVREP_F_COUNT, EVREP_F_TIMESTAMP_POS,
                  EVREP_F_TIMESTAMP_NEG, EVREP_F_COUNT_POS, EVREP_F_COUNT_NEG };
enum evrep_agg { EVREP_A_SUM = 0, EVREP_A_MEAN, EVREP_A_MAX, EVREP_A_VARIANCE };
enum evrep_dtype { EVREP_F64 = 0, EVREP_F32 = 1 };

#define EVREP_MAX_CHANNELS 16  /* MDES channels per launch (larger stacks: several launches) */
#define EVREP_MAX_DIM 4096     /* H, W */

/* Host-side description of one batch; filled by evrep_plan_init, read-only afterwards. */
typedef struct evrep_plan {
    int32_t abi_version;
    int32_t B, H, W;
    int64_t total_events;          /* offsets[B] */
    int64_t max_events_per_window; /* upper bound used to size grids; no host sync needed */
    int32_t chunk, nblk;           /* row-partition geometry (derived) */
    int32_t nchunk, reserved;      /* 128-pixel column chunks per row (derived); reserved = the binning pass chosen:
                                      2 = key-sorted (k_block_keysort alone; the builder waves finish the order by pixel),
                                      3 = k_block_keysort + the column sort per (row, chunk) key (dense windows),
                                      1 = k_block_rowsort + the column sort per row, 0 = the three-kernel pass */
    int32_t flags;                 /* EVREP_PLAN_* bits the plan was made with */
    int32_t pacing;                /* store pacing of the wide float64 builders (evrep_plan_set_pacing): -1 = automatic,
                                      0 = off, > 0 = every builder wave starts its stores no earlier than this many
                                      10 ns ticks after it started */
    size_t off_meta, off_table, off_stats, off_rowoff, off_chunkoff, off_sorted1, off_sorted2, off_cuts, off_scratch;
    size_t workspace_bytes;
} evrep_plan;

/* evrep_plan_init_ex flags: which binning passes the plan may choose from (A/B timing and the cross-pass parity
 * tests; every pass produces the same tensors bit for bit -- but for the time surface, whose stream builder (after the
 * key-sorted pass, dense windows) takes one exponential per event where the ordered builder takes one per slice: the two
 * forms agree to 1e-13 relative), and tuning / A/B knobs. */
#define EVREP_PLAN_NO_KEY_PASS 1u       /* keep passes 2 and 3 (k_block_keysort) out of the choice */
#define EVREP_PLAN_THREE_KERNEL 2u      /* the round-1 three-kernel pass (0) only */
#define EVREP_PLAN_FORCE_KEY_SORTED 4u  /* pass 2 also for windows denser than it is chosen for */
#define EVREP_PLAN_BIG_BLOCKS 8u        /* key-sorted pass: 8192-event blocks also for short windows */
#define EVREP_PLAN_NO_FUSED_SCATTER 16u /* three-kernel pass: separate scan and scatter kernels */
#define EVREP_PLAN_X_SPAN2 64u          /* experiment: float64 MDES units of two 128-pixel chunks (NOTES.md 8) */
#define EVREP_PLAN_X_TAIL_MERGE 256u    /* experiment: a row's last unit also takes a short tail chunk (NOTES.md r04: slower) */
#define EVREP_PLAN_X_POLSTATS_ORDERED 65536u /* A/B: the n_imagenet accumulators by k_polstats also where r06 streams them (k_polstats_stream) */
#define EVREP_PLAN_X_ESTACK_ORDERED 131072u /* A/B: EventStack by k_event_stack also where r06 streams it (k_event_stack_stream) */
#define EVREP_PLAN_X_MDES_NO_COOP 4194304u /* A/B: the ordered float32 ERGO-12's big hot units by time slices of one-wave workgroups (r05) instead of k_mdes_coop */
#define EVREP_PLAN_X_TS_STREAM 1048576u   /* tests / A/B: the time surface by k_time_surface_stream at every density */
#define EVREP_PLAN_X_TS_ORDERED 2097152u  /* A/B: the time surface by k_time_surface also where r06 streams it */
#define EVREP_PLAN_X_MDES_STREAM 524288u  /* tests / A/B: ERGO-12 by k_mdes_stream at every density (default: where it is measured faster) */
#define EVREP_PLAN_X_MDES_ORDERED 262144u /* A/B: ERGO-12 by k_mdes also where r06 streams it (k_mdes_stream) */
#define EVREP_PLAN_X_TORE_ORDERED 32768u  /* A/B: TORE by the ordered / handed-over paths (k_tore) also where r06 streams it (k_tore_stream) */
#define EVREP_PLAN_X_VOXEL_ORDERED 16384u /* A/B: the voxel grid by the ordered paths (k_voxel) also after the key-sorted pass,
                                             where r06 streams it (k_voxel_stream); same tensors bit for bit */

int evrep_abi_version(void);
const char *evrep_last_hip_error(void);

/* Fill `plan` for B windows of an H x W sensor holding total_events events, at most
 * max_events_per_window in any one window. */
int evrep_plan_init(evrep_plan *plan, int32_t B, int32_t H, int32_t W, int64_t total_events,
                    int64_t max_events_per_window);
/* The same with EVREP_PLAN_* flags (evrep_plan_init = flags 0).  The library itself reads no environment variable:
 * the Python binding translates EVREP_BIN_CLASSIC / EVREP_BIN_THREE_KERNEL / EVREP_BIN_KEY_SORTED / ... into flags. */
int evrep_plan_init_ex(evrep_plan *plan, int32_t B, int32_t H, int32_t W, int64_t total_events,
                       int64_t max_events_per_window, uint32_t flags);
/* Store pacing of the builders whose launch is bound by HBM writes (NOTES.md 3.2, Store pacing): ticks = -1 automatic (what
 * evrep_plan_init sets), 0 off, > 0 explicit hold in 10 ns ticks.  Results never depend on it. */
int evrep_plan_set_pacing(evrep_plan *plan, int32_t ticks);
size_t evrep_workspace_bytes(const evrep_plan *plan);

/* The (y,x) binning pass every builder consumes: a stable two-level partition of each window's
 * events by pixel id x + y*W (row partition across workgroups, then column partition inside one
 * workgroup per row), plus per-window statistics (t range, bounding box, MDES polarity flags).
 * Replaces the per-builder `index = y*W + x` scatter of event_stack.py:123-125,
 * operations.py:40, time_surface.py:67, tore.py:23-47.
 * events DEVICE int32 [total,4]; offsets DEVICE int64 [B+1]; workspace DEVICE. */
int evrep_bin_events(const evrep_plan *plan, const int32_t *events, const int64_t *offsets,
                     void *workspace, void *stream);

/* MixedDensityEventStack.stack (representation_search/mixed_density_event_stack.py:25-151) with
 * Operations.exec/run (operations.py:15-89) for C <= EVREP_MAX_CHANNELS channels:
 * window[c] in 0..6 (anything else = the reference's failed channel -> zeros), func[c], agg[c].
 * out DEVICE (B,H,W,C) of out_dtype, each value multiplied by `scale` (1.0, or 255.0 as
 * gen1_transforms.py:31 does). */
int evrep_mdes(const evrep_plan *plan, const int32_t *events, const int64_t *offsets, void *workspace,
               int32_t C, const int32_t *window, const int32_t *func, const int32_t *agg, double scale,
               int32_t out_dtype, void *out, void *stream);

/* The "SBT" stacking of MixedDensityEventStack (mixed_density_event_stack.py:76-107): EIGHT windows cut by the normalised
 * time t_s instead of by event count -- w0 all, w1..w3 i/3 <= t_s <= (i+1)/3 (inclusive both ends), w4..w7 t_s <= 1/2, 1/4,
 * 1/8, 1/16.  evrep_mdes_sbt_windows forms them for B windows (timestamps ascending, as every builder requires) as rank
 * ranges: bounds DEVICE int32 [B][8][2] {lo, hi}, flags DEVICE uint32 [B][2] (which windows hold p == -1 / out-of-frame
 * events).  evrep_mdes_ex = evrep_mdes with window[c] in 0..7 over those windows; bounds == flags == NULL: evrep_mdes. */
int evrep_mdes_sbt_windows(const int32_t *events, const int64_t *offsets, int32_t B, int32_t H, int32_t W, int32_t *bounds,
                           uint32_t *flags, void *stream);
int evrep_mdes_ex(const evrep_plan *plan, const int32_t *events, const int64_t *offsets, void *workspace, int32_t C,
                  const int32_t *window, const int32_t *func, const int32_t *agg, double scale, int32_t out_dtype,
                  void *out, const int32_t *bounds, const uint32_t *flags, void *stream);

/* get_optimized_representation (optimized_representation.py:86-134): the ERGO-12 triples. */
int evrep_optimized(const evrep_plan *plan, const int32_t *events, const int64_t *offsets, void *workspace,
                    double scale, int32_t out_dtype, void *out, void *stream);

/* EventStack.make_stack + post_stack (event_stack.py:45-131) of one half (past, or the reversed future) per
 * window: level k = polarity of the last event at each pixel among events[off_k:].
 * out DEVICE (B,H,W,S) float32; premap 1 applies p -> (p+1)//2 first (gen1_transforms.py:34) and the value is
 * int8(2p - 1) (event_stack.py:18); premap 0: p is {0,1}; premap 2: the p column already holds the int8 value
 * (EventStack.pre_stack builds both halves on the host side: past as is, future reversed and negated, :21-41). */
int evrep_event_stack(const evrep_plan *plan, const int32_t *events, const int64_t *offsets, void *workspace,
                      int32_t stack_size, int32_t premap, float scale, float *out, void *stream);

/* ToTimesurface.__call__ (time_surface.py:25-74): S <= 8 surfaces sampled at event indices.
 * indices == NULL: the cuts gen1_transforms.py:79-81 computes, searchsorted(t_norm, 1..S);
 * otherwise DEVICE int32 [B,S], the `indices` argument of ToTimesurface.__call__.
 * premap bit 0 applies p -> int8((p+1)/2) first (gen1_transforms.py:70-72); bit 1 (premap 2 / 3): the timestamps are NOT
 * ascending -- the scan of time_surface.py:66-74 runs in ARRAY order whatever the timestamps, and so do the kernels; the bit
 * only keeps them from factorising the exponentials around a reference time (a memory timestamp may then lie far BEHIND a
 * cut's).  The caller must hand `indices` itself for such a window: searchsorted on an unsorted array is the caller's numpy's.
 * out DEVICE (B,H,W,2S) float64/float32, channel c = 2s+p. */
int evrep_time_surface(const evrep_plan *plan, const int32_t *events, const int64_t *offsets, void *workspace,
                       int32_t slices, const int32_t *indices, double tau, int32_t premap, double scale,
                       int32_t out_dtype, void *out, void *stream);

/* The same with float64 timestamps (time_surface.py:66-74 is dtype-agnostic): tf DEVICE double [total_events], one time per
 * event, indexed like `events` (whose own t column then only orders them); `indices` must be given.  tf == NULL:
 * evrep_time_surface. */
int evrep_time_surface_ftime(const evrep_plan *plan, const int32_t *events, const int64_t *offsets, void *workspace,
                             int32_t slices, const int32_t *indices, const double *tf, double tau, int32_t premap,
                             double scale, int32_t out_dtype, void *out, void *stream);

/* events2ToreFeature (tore.py:6-83) for one sample time per window, k <= 8.
 * sample_times == NULL: T = t[-1] (gen1_transforms.py:63); otherwise DEVICE int32 [B].
 * frame_mode 0: the events' bounding box, origin-shifted (gen1_transforms.py:61-64); the window's
 *               output is a compact (Hbb,Wbb,2k) array at out + b*H*W*2k, bbox via evrep_read_bbox;
 * frame_mode 1: full (H,W) frame, origin-shifted by (xmin,ymin) (n_imagenet .../imagenet.py:1095-1103);
 * frame_mode 2: full (H,W) frame, no shift (x, y used as 0-based pixel coordinates).
 * Timestamps that are not ascending (EVREP_ST_UNSORTED): array order, each event replacing the (pixel, polarity) k-vector v by the
 * sorted [dt] + v[:k-1] -- what np.partition yields there on numpy >= 2.0 / AVX2+ hosts (tore.py:22-25).
 * out DEVICE float32. */
int evrep_tore(const evrep_plan *plan, const int32_t *events, const int64_t *offsets, void *workspace,
               int32_t k, int32_t frame_mode, const int32_t *sample_times, float scale, float *out, void *stream);

/* The same with FLOAT64 timestamps (n_imagenet hands seconds as '<f8', imagenet.py:1002-1006,1093-1103; tore.py itself
 * computes currentSampleTime - ts in whatever dtype it is given): tf DEVICE double [total_events] = every event's
 * time, indexed like `events` (their t column is then only used for the sortedness check); sample_times_f DEVICE
 * double [B] or NULL (= tf of the window's last event).  tf == NULL: exactly evrep_tore. */
int evrep_tore_ftime(const evrep_plan *plan, const int32_t *events, const int64_t *offsets, void *workspace, int32_t k,
                     int32_t frame_mode, const int32_t *sample_times, const double *tf, const double *sample_times_f,
                     float scale, float *out, void *stream);

/* compute_repr (representation_search/gromov_wasserstein.py:72-82) with t normalised as :96;
 * mode 0.  mode 1 = tonic.transforms.ToVoxelGrid as gen1_transforms.py:22-25 consumes it
 * (restated from tonic's published algorithm; parity unpinned).  mode 2 = ev-licious
 * events_to_voxel_grid, integer-pixel path (ev-licious/src/evlicious/tools/utils.py:52-108), before
 * its optional normalisation.  out DEVICE (B,H,W,bins) float64. */
int evrep_voxel(const evrep_plan *plan, const int32_t *events, const int64_t *offsets, void *workspace,
                int32_t bins, int32_t mode, double scale, double *out, void *stream);
/* The same with an explicit time range per window (mode 2 only): t_range DEVICE int64 [B,2] = the t0_us, t1_us
 * arguments of events_to_voxel_grid (utils.py:52,60-63), in the units of the events' t column; NULL = t[0], t[-1].
 * Events outside the range fall outside the bins and are dropped, except that the bin index is truncated toward
 * zero as the reference's astype("int32") does (:67), so up to one bin before t0_us still counts into bin 0. */
int evrep_voxel_range(const evrep_plan *plan, const int32_t *events, const int64_t *offsets, void *workspace,
                      int32_t bins, int32_t mode, double scale, const int64_t *t_range, double *out, void *stream);

/* ev-licious events_to_voxel_grid for SUB-PIXEL event coordinates (Events.divider > 1; utils.py:86-102): the
 * bilinear-in-x/y draw of every event into its four surrounding pixels, float32 accumulation in the reference's
 * order.  `events` carries the TRUNCATED coordinates (x.astype("int32"), :92-93), xy DEVICE double [total,2] the
 * original (x, y) of every event (indexed like `events`); t_range as evrep_voxel_range.
 * out DEVICE float32 (B,H,W,bins), before the optional normalisation. */
/* compute_repr(x, y, t, p, width, height, bins) itself (representation_search/gromov_wasserstein.py:72-82): the caller hands
 * its own normalised time per event -- tnorm DEVICE double [total_events], indexed like `events`, used exactly as the
 * reference uses `t`: b = (bins - 1) * t, blim in {int(b), int(b) + 1}, grid[y, x, blim] += (1 - |blim - b|) * p, lower
 * bin for every event, then the upper bin.  The events' own t column only orders them.  out DEVICE double (B,H,W,bins). */
int evrep_voxel_tnorm(const evrep_plan *plan, const int32_t *events, const int64_t *offsets, void *workspace,
                      const double *tnorm, int32_t bins, double scale, double *out, void *stream);

int evrep_voxel_subpixel(const evrep_plan *plan, const int32_t *events, const int64_t *offsets, void *workspace,
                         const double *xy, int32_t bins, const int64_t *t_range, float *out, void *stream);

/* n_imagenet's per-polarity accumulators (n_imagenet/real_cnn_model/data/imagenet.py:169-511,841-871:
 * reshape_then_acc, _acc_time, _acc_count, _acc_count_pol, _acc_count_only, _acc_all, _flat, _flat_pol,
 * _acc_exp, _acc_time_pol, _acc_intensity) as ONE builder: channel c = stat[c] of the events of polarity
 * class pol[c] at each pixel.  pol: EVREP_PS_ANY (every event), EVREP_PS_POS (p > 0), EVREP_PS_NEG (p < 0).
 * stat: COUNT = torch.bincount (:187-189); TMAX / TMIN = torch_scatter.scatter_max / scatter_min of the
 * normalised time, 0 where empty (:203-206,236-239); FLAG = 1 where any event (:405-406); EXP =
 * exp(-(1 - TMAX)/tau) over the WHOLE frame, empty pixels included (:461-465); SIGNED = count(p>0) -
 * count(p<0) (:866).  tnorm DEVICE double [total_events] = (t - t[0]) / (t[-1] - t[0]) per window in
 * float64 (:180-181,198-199), indexed like `events`; the events' own t column is not used.
 * The window-level normalisations (count / count.max() :190-191, min-max of the intensity :867) are
 * the caller's.  out DEVICE (B,H,W,C) float32, C <= 16. */
#define EVREP_PS_ANY 0
#define EVREP_PS_POS 1
#define EVREP_PS_NEG 2
#define EVREP_PS_COUNT 0
#define EVREP_PS_TMAX 1
#define EVREP_PS_TMIN 2
#define EVREP_PS_FLAG 3
#define EVREP_PS_EXP 4
#define EVREP_PS_SIGNED 5
int evrep_polstats(const evrep_plan *plan, const int32_t *events, const int64_t *offsets, void *workspace,
                   const double *tnorm, int32_t C, const int32_t *pol, const int32_t *stat, double tau, float *out,
                   void *stream);

/* EST quantisation layer, forward only (ev-YOLOv6/yolov6/models/learned_repr.py:143-179): channel p*C + i of a
 * pixel = sum over its events of t_n * f(t_n - i/(C-1)), float32, events of a pixel added in time order (what
 * vox.put_(idx, values, accumulate=True) does on the CPU, :173).  f = the layer's value MLP (:9-43), a scalar
 * function of a scalar with LeakyReLU activations, i.e. EXACTLY piecewise linear: the caller passes it as
 * nseg segments {u_next, a, c} (f(u) = a*u + c for u below u_next, ascending, DEVICE double [nseg][3]) plus a
 * uniform bucket index over [lo, hi] (DEVICE uint32 [nbucket]: first segment that can contain the bucket's left
 * edge) -- built on the host from the MLP weights (event_representation_study_amd/est.py).
 * tnorm DEVICE float [total_events] = t / t.max() per window (:159-160), indexed like `events`; p > 0 selects
 * the second half of the channels (the layer expects p in {0, 1}).  out DEVICE (B,H,W,2C) float32, C <= 8. */
int evrep_est_voxel(const evrep_plan *plan, const int32_t *events, const int64_t *offsets, void *workspace,
                    const float *tnorm, int32_t C, const double *segments, int32_t nseg, const uint32_t *buckets,
                    int32_t nbucket, double lo, double hi, float *out, void *stream);

/* Synchronous read-backs (they synchronise `stream`). status: HOST uint32 [B];
 * bbox: HOST int32 [B,4] = xmin, ymin, xmax, ymax of each window's in-frame events. */
int evrep_read_status(const evrep_plan *plan, const void *workspace, uint32_t *status, void *stream);
/* The same statistics WITHOUT a synchronisation: B records of 64 bytes {int32 tmin, tmax, xmin, xmax, ymin, ymax; uint32
 * neg_flags, oob_flags, status; int32 n_valid; 24 bytes reserved} are copied to meta_out (HOST, ideally pinned) behind the
 * work queued on `stream`; they are valid once the caller has synchronised the stream -- so a per-sample wrapper needs ONE
 * synchronisation for the status and the result together. */
int evrep_copy_window_meta_async(const evrep_plan *plan, const void *workspace, void *meta_out, void *stream);
int evrep_read_bbox(const evrep_plan *plan, const void *workspace, int32_t *bbox, void *stream);

/* Placement probe (no reference counterpart): writes zeros over `bytes` of `out` with the write footprint of the float64
 * 12-channel builder and nothing else.  On MI355X the physical placement of a ~1 GB output tensor decides up to 25 % of a
 * builder launch (NOTES.md section 8); a producer that allocates its output ring once can time this call into a few
 * candidate allocations and keep the fastest (engine.probe_output_placement does).  out DEVICE, 16-byte aligned. */
int evrep_probe_store(void *out, size_t bytes, void *stream);

/* OTMI(Xs, Xt, h).solve()[1] (representation_search/compute_otmi.py:61-93) in closed form for
 * POT's max_iter=0 path: mean over the LxL zero-padded grid of |Ks - Kt|.
 * Xs DEVICE double [n,ds], Xt DEVICE double [m,dt] (ds, dt <= 32); scratch DEVICE of
 * evrep_gwd_scratch_bytes(n, m), 256-byte aligned; cost DEVICE double [1].
 * Arithmetic: float64 statistics; the pairwise exponents in float32 accuracy on the matrix cores -- clouds of <= 15
 * dimensions (both) as exact three-way bfloat16 splits of the float32 operands (v_mfma_f32_32x32x16_bf16), wider ones as
 * float32 MFMA chains; exponentials in float32; sums in float64.  Measured 5e-9 relative against the float64 value, the
 * reference's budget is 1e-5. */
size_t evrep_gwd_scratch_bytes(int64_t n, int64_t m);
int evrep_gwd_padded_l1(const double *Xs, int64_t n, int32_t ds, const double *Xt, int64_t m, int32_t dt,
                        double h, void *scratch, double *cost, void *stream);

/* P solves at once: costs[p] = evrep_gwd_padded_l1 of the pair p, bit for bit, in six launches for ALL pairs (a single
 * solve is five launches, four of them tiny).  The clouds' sizes are read on the DEVICE, so pairs produced by
 * evrep_otmi_event_clouds / evrep_otmi_rep_clouds are scored without a host read-back.
 * Xs DEVICE double: pair p's source cloud = rows [xs_row[p], xs_row[p] + n[p]) of ds columns (xs_row NULL: p * n_cap);
 * likewise Xt / xt_row / m / dt.  xs_row, n, xt_row, m DEVICE int64 [P].  n_cap, m_cap: upper bounds of n[p], m[p]
 * (they size the scratch slots; a pair beyond them, or with an empty cloud, costs NaN).  All pairs share ds, dt, h.
 * scratch DEVICE of evrep_gwd_batch_scratch_bytes; costs DEVICE double [P]. */
size_t evrep_gwd_batch_scratch_bytes(int32_t P, int32_t ds, int32_t dt, int64_t n_cap, int64_t m_cap);
int evrep_gwd_padded_l1_batch(int32_t P, const double *Xs, const int64_t *xs_row, const int64_t *n, int32_t ds,
                              const double *Xt, const int64_t *xt_row, const int64_t *m, int32_t dt, int64_t n_cap,
                              int64_t m_cap, double h, void *scratch, double *costs, void *stream);

/* The point clouds of the harness otmi(events, rep, height, width, rep_size)
 * (representation_search/compute_otmi.py:96-211), built on the device in the reference's own order.
 * evrep_otmi_event_clouds: events DEVICE int32 [total,4] + offsets DEVICE int64 [B+1] (B windows) -> for window b the
 *   three scored sensor quadrants (the most populated one is skipped, :134-135; quadrants 2-4 re-origined, :140-147;
 *   float32 x / ((W-1)//2), y / ((H-1)//2), (t - t0) / (t1 - t0), (p - pmin) / (pmax - pmin), rows with
 *   x < (W-1)//2 and y < (H-1)//2 kept, :164-173): Xs DEVICE double [B][3][cap][4], n_out DEVICE int64 [B][3],
 *   quad_out DEVICE int32 [B][3] (which quadrant each slot holds).  cap >= the longest window.
 * evrep_otmi_rep_clouds: rep DEVICE (items, S, S, C) letterboxed representations, item i belongs to window i % B ->
 *   for slot k the cut of quadrant quad[i % B][k] (:150-155,177-179), two positional channels (:181-198), rows with
 *   sum |feat| > 0 (:200-202): Xt DEVICE double [items][3][m_cap][C+2], m_out DEVICE int64 [items][3];
 *   m_cap >= (S - S/2 + 1)^2.
 * Both cut their input into slices, one workgroup each, that meet through `scratch`: DEVICE, 16-byte aligned, of
 * evrep_otmi_scratch_bytes(B) / evrep_otmi_scratch_bytes(items) bytes (no initialisation needed). */
size_t evrep_otmi_scratch_bytes(int32_t windows_or_items);
int evrep_otmi_event_clouds(const int32_t *events, const int64_t *offsets, int32_t B, int32_t height, int32_t width,
                            int64_t cap, double *Xs, int64_t *n_out, int32_t *quad_out, void *scratch, void *stream);
int evrep_otmi_rep_clouds(const void *rep, int32_t rep_dtype, int32_t items, int32_t B, int32_t S, int32_t C,
                          const int32_t *quad, int64_t m_cap, double *Xt, int64_t *m_out, void *scratch, void *stream);

/* Per-channel resize of a channel-last representation (B,H,W,C) -> (B,Ho,Wo,C), what resize_image /
 * resize_image_process do with cv2.resize per channel before a representation is stored or scored
 * (ev-YOLOv6/yolov6/data/gen4/precompute_reps.py:179-260,424; gen1_2yolo.py:230-265).  The interpolation is given as
 * separable tap tables: output row oy = sum_{t < ycount[oy]} ywt[oy*T + t] * (source row ystart[oy] + t), likewise
 * for columns -- built on the host from OpenCV's published INTER_AREA / INTER_LINEAR table construction
 * (event_representation_study_amd/gwd_pipeline.py; parity unpinned against cv2, which is absent).
 * in DEVICE float64/float32 (in_dtype); tap tables DEVICE; out DEVICE float64/float32 (out_dtype: the reference stores
 * float32, precompute_reps.py:434), every value times `scale`. */
int evrep_resize_taps(const void *in, int32_t in_dtype, int32_t B, int32_t H, int32_t W, int32_t C, int32_t Ho, int32_t Wo,
                      int32_t T, const int32_t *ystart, const int32_t *ycount, const double *ywt, const int32_t *xstart,
                      const int32_t *xcount, const double *xwt, double scale, int32_t out_dtype, void *out, void *stream);

/* EXTENSION (no live call of the reference computes this): entropic Gromov-Wasserstein by
 * projected gradient, restated from POT's published ot.gromov.entropic_gromov_wasserstein (init_matrix,
 * tensor_product, gwggrad, gwloss, sinkhorn_knopp) with FIXED iteration counts instead of its tolerance tests --
 * the solver family of the reference's dead-code call ot.gromov.gromov_wasserstein(Ks, Kt, p, q, "kl_loss")
 * (representation_search/gromov_wasserstein.py:62-69).  PARITY UNPINNED against POT (absent); checked against
 * oracle/gw_oracle.py.
 * C1 DEVICE double [n,n], C2 DEVICE double [m,m], p DEVICE double [n], q DEVICE double [m];
 * loss 0 = "square_loss", 1 = "kl_loss"; precision EVREP_F64 (v_mfma_f64_16x16x4_f64) or EVREP_F32
 * (v_mfma_f32_16x16x4_f32; plan and Gibbs kernel in float32, Sinkhorn scalings in float64);
 * scratch DEVICE of evrep_gw_scratch_bytes(n, m, precision); T_out DEVICE double [n,m] or NULL; gw_out DEVICE double [1]. */
size_t evrep_gw_scratch_bytes(int64_t n, int64_t m, int32_t precision);
int evrep_entropic_gw(const double *C1, int64_t n, const double *C2, int64_t m, const double *p, const double *q,
                      int32_t loss, double epsilon, int32_t outer_iters, int32_t sinkhorn_iters, int32_t precision,
                      void *scratch, double *T_out, double *gw_out, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* EVREP_H_ */
