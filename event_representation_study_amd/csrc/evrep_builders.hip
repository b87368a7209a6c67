// evrep_builders.hip -- the event->dense-tensor builders as per-pixel segmented reductions over
// the binned stream (evrep_bin.hip), gfx950.
//
// Work unit = one wavefront (a 64-thread workgroup) owning a chunk of kChunkPx = 128 consecutive
// pixels of one sensor row of one window (two chunks for narrow float32 outputs on sparse windows).  The wave
//   1. gets the unit's records, grouped by pixel and time-ordered inside a pixel (unit_front): after the key-sorted
//      binning pass it gathers them itself from the window's block runs and groups them in LDS (unit_records);
//      after the classic passes they are one coalesced 16 B/lane load of the pixel-sorted stream; the builder's
//      digest (a per-event division, exponential or logarithm) is applied one record per lane (emit_chunk),
//   2. fills its private LDS part tile (kPartPx pixels x C, output layout) with the channel
//      background (zero, or the builder's empty-pixel value),
//   3. lists the non-empty pixels of the chunk (segment heads of the pixel-sorted records, ballot
//      prefix) -- one lane per NON-EMPTY pixel, so VALU work scales with events, not pixels,
//   4. reduces each segment IN TIME ORDER (float64 sums round exactly as the reference's
//      sequential scatter does) into the lane's registers,
//   5. emits the chunk as part tiles through the ONE part-size LDS tile: lanes whose pixel
//      lies in a later part keep their values in registers while the earlier part is streamed
//      out with 16-byte-per-lane coalesced non-temporal stores.  Halving the tile takes the float64 12-channel
//      builder from 9 to 19 resident waves per CU while a wave still moves 12 KB, so almost twice the
//      store bytes are in flight per CU (202 -> 168 us for the headline kernel).
// No block barrier exists (one wave per workgroup; wave_phase() only orders LDS phases), every
// output element is written exactly once, and HBM traffic per window = 16 B per event (binned
// record) + sizeof(out) per element.
//
// XCD awareness (chunk_unit): workgroup i runs on XCD i % 8 (observed dispatch order); XCD x is
// given the x-th contiguous eighth of the linear (window, row, chunk) order, so every XCD streams one
// sequential region of the output instead of interleaving 12 KiB tiles with the other seven
// (6.9 vs 6.2 TB/s of raw HBM writes for 12 KiB one-wave tiles, tools/microbench/store_patterns5.hip).
#include "evrep_common.h"

namespace evrep {

__host__ __device__ inline size_t align16(size_t v) { return (v + 15) & ~(size_t)15; }

// waves per SIMD asked of the compiler for the main launches (1 = no request).  r05c: EventStack (84-93 -> 73-79 VGPRs) and the voxel
// grid of <= 8 bins (83 -> 80) fit six waves WITHOUT scratch: Gen1 35.1 -> 33.0 / 60.0 -> 58.0 us, 640x480 x 250 000 events 34.5 -> 32.8 /
// 65.6 -> 62.6; TORE (93 -> 75) gains nothing from it
#ifndef EVREP_ES_WAVES
#define EVREP_ES_WAVES 6
#endif
#ifndef EVREP_TORE_WAVES
#define EVREP_TORE_WAVES 1
#endif
#ifndef EVREP_VOXEL_WAVES
#define EVREP_VOXEL_WAVES 6
#endif
#ifndef EVREP_PARTS
#define EVREP_PARTS 2
#endif
#ifndef EVREP_DEFER_MULT
#define EVREP_DEFER_MULT 1   // a main launch defers a unit with a 64-pixel part of more than this many hot stages of records
#endif
#ifndef EVREP_XCD_MAP
#define EVREP_XCD_MAP 1
#endif
#ifndef EVREP_NT_STORES
#define EVREP_NT_STORES 1  // non-temporal output stores (the tensor is written once and never re-read by the step): -3 us on
                           // the ERGO-12 launch and -3 us on the next binning pass, whose loads find less of L2 evicted (r02)
#endif
#ifndef EVREP_SPARSE_EMIT
#define EVREP_SPARSE_EMIT 1  // units of <= 64 records leave the wave through sparse_store (one burst) instead of the part tiles
#endif
constexpr int kParts = EVREP_PARTS;
constexpr int kPartPx = kChunkPx / kParts;  // pixels per part tile
constexpr int kEvStage = 64;                // records staged in LDS; denser chunks read the rest from HBM/L2
// The stage size is chosen per launch (unit_cfg() in evrep_capi.hip): 64 records for one-chunk units, 128 for wider ones
// (two-chunk float32 units and TORE's shifted frame hold ~65 records on the sparse windows they are chosen for: the
// key-sorted front end orders them inside LDS).
struct UnitCfg {
    int span;    // 128-pixel chunks per unit
    int stage;   // records the wave's LDS stage holds (a multiple of 64)
    int partpx;  // pixels of one part tile: 64 (kPartPx), or 128 for narrow pixels, whose 64-pixel tiles are too small to
                 // pay for their own fill / store / phase sequence (n_imagenet accumulators 67 -> 62 us, EventStack 87 -> 81)
    int hold;    // store pacing: the wave starts its stores no earlier than `hold` x 10 ns after it started (0 = off)
    // Unit id -> (window, row, unit of the row) without integer divisions (r05b: the front of every builder wave ran four of
    // them -- ~35 dependent scalar / reciprocal instructions each -- before its first load could be issued): the host hands the
    // units per row and two multiply-shift reciprocals over (fastdiv_make; exact for ids < 2^31, which evrep_plan_init ensures)
    int nunit;
    uint32_t nunit_m, nunit_sh, h_m, h_sh;
    int xflags;  // bit 2 (value 4): two-chunk units of sparse windows: only units of >= kHotSubMin records are handed over (r05b: a monster unit's
                 // ordering is the main launch's tail -- 200 us on the 1 Mpx circle -- while the mid-size hot units of a 640x480 window are
                 // better off ordered beside the store-bound waves);
                 // bit 1: builders with a hot-launch split sweep (float32 ERGO-12) hand units beyond the record stage over whole -- set by
                 // the host for one-chunk units of windows whose AVERAGE unit fits the stage (hot units are the exception: on dense
                 // windows every unit would go, and the hot launch is the slower place: 8 x 500 000 events 82 -> 134 us, measured);
                 // EVREP_PLAN_X_HANDOVER2 (experiment) also sets it for two-chunk units
    int merge;   // 1: the row's last unit also takes the short tail chunk of a sensor whose width is not a multiple of 128 (r04:
                 // Gen1's 304-pixel rows are 128 + 128 + 48 -- a third of the units were 48-pixel tails with a full unit's
                 // fixed cost; now a row is two units, 128 and 176 pixels)
};
// floor(n / d) for 0 <= n < 2^31 as mulhi(n, m) >> sh (Granlund-Montgomery: l = ceil(log2 d), m = floor(2^(31 + l) / d) + 1,
// sh = l - 1; m == 0 stands for d == 1)
__host__ inline void fastdiv_make(uint32_t d, uint32_t &m, uint32_t &sh) {
    m = 0; sh = 0;
    if (d <= 1) return;
    uint32_t l = 0;
    while ((1ull << l) < d) ++l;
    m = (uint32_t)(((1ull << (31 + l)) / d) + 1ull);
    sh = l - 1;
}
__device__ inline uint32_t fastdiv(uint32_t n, uint32_t m, uint32_t sh) { return m ? (__umulhi(n, m) >> sh) : n; }
// units per sensor row / the chunks of unit `ur` of a row
__host__ __device__ inline int units_per_row(int nchunk, int span, int merge) {
    return merge ? ((nchunk - 1 + span - 1) / span > 0 ? (nchunk - 1 + span - 1) / span : 1) : (nchunk + span - 1) / span;
}
// float32 builders on sparse windows take two consecutive 128-pixel chunks per wave (the same 12 KB per
// wave as a float64 builder: 108 -> 90 us for EventStack at 640x480x32); chosen on the host from the
// average record count per chunk, see unit_cfg() in evrep_capi.hip.

// LDS carve of one builder wave.  HOT_: the wave belongs to a HOT launch (run_units): it may run the paths of units that
// do not fit the stage (the spill sort of unit_records, emit_rounds); a main launch defers such units instead.
template <typename OutT, bool HOT_ = false>
struct WaveLds {
    static constexpr bool kHot = HOT_;
    OutT *tile;   // kPartPx * C elements, output layout (pixel-major, channel-minor)
    OutT *bg;     // EVREP_MAX_CHANNELS background values (the empty-pixel value of every channel)
    uint2 *segs;  // (pixel offset inside the chunk, first record index); entry nseg = sentinel
    Rec *evbuf;   // `nstage` records of the unit (the classic front end fills the first kEvStage)
    int segcap;   // capacity of segs (pixels a unit can hold: span * kChunkPx, one more chunk for TORE's shift)
    int nstage;
    int partpx;   // pixels of the part tile
    // the HOT stage (r04, emit_rounds): a unit of more records than `nstage` stages them, piece by piece, across the tile
    // AND the record stage (`bigcap` records; 448 for the float64 12-channel builder).  The background vector lies between
    // the two and stays what it is (TimeSurface reads it inside its walks): slot j >= bigsplit skips it.
    Rec *big;
    int bigcap, bigsplit, bigskip;
    __device__ inline Rec *big_at(uint32_t j) const { return big + j + (j >= (uint32_t)bigsplit ? (uint32_t)bigskip : 0u); }
    // Store pacing (r03).  The resident waves of a store-bound builder OFFER more write traffic than HBM serves (19 waves
    // per CU x 12 KiB each, ready ~3.4 us after they start: 13 TB/s).  On most physical placements of a ~1 GB output
    // tensor that oversubscription collapses the write rate to 5.6-5.9 TB/s (tools/microbench/placement_patterns{3,4}.hip:
    // 168 us for the 944 MB of the headline tensor, against 135 us on the "fast" placements).  A wave that holds its first
    // store until `hold` x 10 ns after its start keeps the offered load at the service rate -- few waves are in their
    // store phase at any time -- and the same write takes 137-140 us wherever the tensor lies.  The clock is
    // s_memrealtime (100 MHz, constant).  Results never depend on it.
    long long t0;
    int hold;
#ifdef EVREP_TIMING  // experiment builds only (tools/experiments/phase_times.py): per-phase wave times, summed into dbg[]
    unsigned long long *dbg;
    __device__ inline void mark(int idx) const {   // dbg = this wave's own 8 slots: no contention
        if (threadIdx.x == 0 && dbg) dbg[idx] = (unsigned long long)((long long)wall_clock64() - t0);
    }
#elif defined(EVREP_STOP_AFTER)  // experiment builds only: the wave ends at mark EVREP_STOP_AFTER (instruction counts per phase)
    __device__ inline void mark(int idx) const { if (idx == EVREP_STOP_AFTER) __builtin_amdgcn_endpgm(); }
#else
    __device__ inline void mark(int) const {}
#endif
    __device__ inline void arm(int hold_) {
        hold = hold_;
#ifdef EVREP_TIMING
        t0 = (long long)wall_clock64();
#else
        t0 = hold_ > 0 ? (long long)wall_clock64() : 0ll;
#endif
    }
    __device__ inline void pace() const {
        if (hold > 0) while ((long long)wall_clock64() - t0 < (long long)hold) __builtin_amdgcn_s_sleep(4);
    }
    __device__ WaveLds(unsigned char *smem, int C, int segcap_, int nstage_, int partpx_ = kPartPx)
        : segcap(segcap_), nstage(nstage_), partpx(partpx_), t0(0), hold(0) {
#ifdef EVREP_TIMING
        dbg = nullptr;
#endif
        size_t o = 0;
        tile = reinterpret_cast<OutT *>(smem + o);  o += align16((size_t)partpx_ * C * sizeof(OutT));
        bigsplit = (int)(o / sizeof(Rec));
        bg = reinterpret_cast<OutT *>(smem + o);    o += align16((size_t)EVREP_MAX_CHANNELS * sizeof(OutT));
        bigskip = (int)(o / sizeof(Rec)) - bigsplit;
        evbuf = reinterpret_cast<Rec *>(smem + o);  o += (size_t)nstage_ * sizeof(Rec);
        segs = reinterpret_cast<uint2 *>(smem + o);
        big = reinterpret_cast<Rec *>(smem);
        bigcap = (int)(o / sizeof(Rec)) - bigskip;
    }
};

// segcap = pixels one unit can touch: span * kChunkPx (+ kChunkPx for TORE's shifted frame)
__host__ __device__ inline size_t chunk_lds_bytes(int C, size_t elem, int segcap, int nstage, int partpx = kPartPx) {
    return align16((size_t)partpx * C * elem) + align16((size_t)EVREP_MAX_CHANNELS * elem) +
           (size_t)nstage * sizeof(Rec) + align16((size_t)(segcap + 1) * sizeof(uint2));
}

// Ordering point between LDS phases of a ONE-WAVE workgroup.  LDS operations of a wave execute in
// program order, so cross-lane producer/consumer phases only need the compiler to keep them in
// order; unlike __syncthreads() nothing is drained (no s_waitcnt vmcnt(0), no s_barrier).
__device__ inline void wave_phase() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <typename OutT>
__device__ inline void tile_fill(OutT *tile, int npix, int C, const OutT *bg) {
    constexpr int V = 16 / (int)sizeof(OutT);
    float4 *t4 = reinterpret_cast<float4 *>(tile);
    if (!bg) {
        const int nvec = (npix * C + V - 1) / V;  // the tile is padded to a multiple of 16 bytes
        for (int v = threadIdx.x; v < nvec; v += kWave) t4[v] = make_float4(0.f, 0.f, 0.f, 0.f);
    } else if ((C % V) == 0) {
        // the pixel is a whole number of 16-byte vectors: the background repeats every vpp vectors
        const int vpp = C / V;
        const float4 *b4 = reinterpret_cast<const float4 *>(bg);
        const int nvec = npix * vpp;
        int q = threadIdx.x % vpp;
        const int step = kWave % vpp;
        for (int v = threadIdx.x; v < nvec; v += kWave) {
            t4[v] = b4[q];
            q += step;
            if (q >= vpp) q -= vpp;
        }
    } else {
        for (int e = threadIdx.x; e < npix * C; e += kWave) tile[e] = bg[e % C];
    }
}

// Stream `count` tile elements to global memory, 16 B per lane (1 KiB per wave-instruction) when
// the destination is aligned; four LDS reads are in flight before their four stores issue.
template <typename OutT>
__device__ inline void tile_store(const OutT *tile, int count, OutT *__restrict__ dst) {
    constexpr int V = 16 / (int)sizeof(OutT);
    if ((reinterpret_cast<uintptr_t>(dst) & 15u) == 0) {
        const int nvec = count / V;
        int v = threadIdx.x;
#if EVREP_NT_STORES
        typedef float nt4 __attribute__((ext_vector_type(4)));
        const nt4 *n4 = reinterpret_cast<const nt4 *>(tile);
        nt4 *o4 = reinterpret_cast<nt4 *>(dst);
        for (; v + 3 * kWave < nvec; v += 4 * kWave) {
            const nt4 a = n4[v], b = n4[v + kWave], c = n4[v + 2 * kWave], d = n4[v + 3 * kWave];
            __builtin_nontemporal_store(a, o4 + v); __builtin_nontemporal_store(b, o4 + v + kWave);
            __builtin_nontemporal_store(c, o4 + v + 2 * kWave); __builtin_nontemporal_store(d, o4 + v + 3 * kWave);
        }
        for (; v < nvec; v += kWave) __builtin_nontemporal_store(n4[v], o4 + v);
#else
        const float4 *s4 = reinterpret_cast<const float4 *>(tile);
        float4 *d4 = reinterpret_cast<float4 *>(dst);
        for (; v + 3 * kWave < nvec; v += 4 * kWave) {
            const float4 a = s4[v], b = s4[v + kWave], c = s4[v + 2 * kWave], d = s4[v + 3 * kWave];
            d4[v] = a; d4[v + kWave] = b; d4[v + 2 * kWave] = c; d4[v + 3 * kWave] = d;
        }
        for (; v < nvec; v += kWave) d4[v] = s4[v];
#endif
        for (int e = nvec * V + threadIdx.x; e < count; e += kWave) dst[e] = tile[e];
    } else {
        for (int e = threadIdx.x; e < count; e += kWave) dst[e] = tile[e];
    }
}

// A stream builder sweeps a unit of more than 64 * RB records RUN BY RUN (a batch = up to 64 consecutive records of one block run: a
// handful of scalar instructions per batch) when its runs hold this many records on average, else 64 consecutive records of the
// UNIT per batch (every lane finds its record's run itself: full batches)
// (measured, r06: `tools/experiments/lib_ab.sh`).  The order-free streams -- a batch is a few LDS atomics -- want the cheap addresses
// (never run by run: EventStack / TORE / accumulators on circle streams +15-50 %); the builders whose batches run ELECTION rounds for
// their ordered float64 sums want full batches -- the voxel grid above all (two passes, every record elected): 12 -> 48 records per
// run takes 1 Mpx edges from 92 to 73 us, 1 Mpx circle 155 -> 132, Gen1 circle / edges 125 / 90 -> 120 / 87.
#ifndef EVREP_STREAM_BYRUN
#define EVREP_STREAM_BYRUN 12
#endif
#ifndef EVREP_STREAM_G   // batches of a big unit a stream wave keeps in flight
#define EVREP_STREAM_G 4
#endif
#ifndef EVREP_ORDERED_BYRUN   // (the ordered builders' sweeps of a unit beyond the record stage: unit_records)
#define EVREP_ORDERED_BYRUN 48
#endif
#ifndef EVREP_VOXEL_BYRUN
#define EVREP_VOXEL_BYRUN 48
#endif
#ifndef EVREP_MDES_STREAM_BYRUN
#define EVREP_MDES_STREAM_BYRUN 48
#endif
// Linear work-unit id of this workgroup among `total` units (see the XCD note in the file header).
__device__ inline int chunk_unit(int total) {
    const int lin = (int)(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z));
#if EVREP_XCD_MAP
    if ((total & 7) == 0) {
        const int per = total >> 3, xcd = lin & 7;
#ifdef EVREP_XCD_ROT
        int idx = (lin >> 3) + xcd * (EVREP_XCD_ROT);
        idx %= per;
#else
        const int idx = lin >> 3;
#endif
        return xcd * per + idx;
    }
#endif
    return lin;
}

struct ChunkGeom {
    int b, row, c0, npix;
    uint32_t cs, ce;
};

// What a builder reads of the binning pass.  Classic (fused == 0): the pixel-sorted stream + its chunk offsets.
// Key-sorted (fused == 1, k_block_keysort): the window's block runs, ordered by (row, 128-pixel chunk) only, and
// their offset tables; the builder wave gathers its unit's records and orders them by pixel itself (unit_records).
struct BinView {
    const int4 *ev;             // the caller's events (key-sorted pass: only read for escaped polarity values, see Rec8)
    const Rec *sorted;          // classic: sorted2 (16-byte records); key-sorted: sorted1 = the block runs, 8-byte records (Rec8)
    const uint32_t *chunk_off;  // classic only
    const uint32_t *table;      // key-sorted: [B][nblk][H * nchunk + 1]
    const BlockStats *stats;    // key-sorted: [B][nblk]
    const WindowMeta *meta;     // classic: [B]
    Rec *spill;                 // key-sorted: sorted2, where a unit of more than kEvStage records is laid out
    int nblk, fused, chunk_shift;  // events per block run = 1 << chunk_shift (a runtime 64-bit division costs ~130 scalar instructions)
    // Hot units (run_units): [l * 16] = the item count of sublist l, [(kHotLists + l) * 16] = its exit ticket,
    // [kHotHdrWords + l * hot_sublist_cap(hot_cap) ...] = its items.  A main launch appends the pieces of the units that do not
    // fit its stage; the hot launch behind it works them off and leaves the list empty.
    uint32_t *hot;
    uint32_t hot_cap;
    double *placed_pool;        // key-sorted: the idle upper half of sorted1 (8 bytes per event of the batch): where a sliced hot unit's kept records are ordered
    BlockStats *stats_rw;       // key-sorted: a main wave that cannot defer a unit (every sublist full) reports it in the window's status
#ifdef EVREP_TIMING
    unsigned long long *dbg;
    unsigned long long *dbg_wave;   // a stream kernel points it at its wave's slots 0-3: stream_unit_records leaves a big unit's sweep times there
#endif
};

// The records of one unit, pixel-sorted: r0 = record `lane`; records [0, nstaged) are in the wave's evbuf, later
// ones are read from sorted[cs + j].
struct UnitRecs {
    const Rec *sorted;
    uint32_t cs, ce;
    int nstaged;
    Rec r0;
    // key-sorted single-batch units (r03): the front end has already grouped the records by pixel and listed the segments
    // (w.segs holds nseg entries + the sentinel); record `lane` (r0) belongs at stage position `pos` -- emit_chunk stages
    // its DIGEST there directly.  nseg < 0: not grouped (the segment heads are found from the staged pixel ids).
    int nseg, pos;
    // the unit holds more records than the wave's stage and was handed to the builder's HOT launch (run_units): nothing to emit
    bool deferred;
    int part;   // hot launch: the 64-pixel part of the unit this wave emits (run_units)
    uint32_t pst, pen;   // hot launch, per lane: the segment of pixel part * 64 + lane in the unit's spill slot
    bool hot_lds;        // hot launch: the part's records lie RAW in the hot stage already (not in the slot)
    int dpx, npixu;      // main launch, a unit in the spill slot: output pixel o = unit pixel o + dpx; w.segs holds every pixel's END
    int sub;             // split path, hot launch: > 0 = the unit was swept in `sub` time slices; this wave, the last to finish, emits it
};

// inclusive scan over the 64 lanes on the VALU (DPP row shifts + row broadcasts), no LDS crossbar
__device__ inline uint32_t wave_incl_scan(uint32_t v) {
    int x = (int)v;
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false);  // row_shr:1
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false);  // row_shr:2
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false);  // row_shr:4
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false);  // row_shr:8
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);  // row_bcast:15 -> rows 1, 3
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);  // row_bcast:31 -> rows 2, 3
    return (uint32_t)x;
}

__device__ inline uint32_t wave_incl_max_scan(uint32_t v) {
    int x = (int)v;  // values are small non-negative run indices: signed max is fine, identity 0
    x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false));
    x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false));
    x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false));
    x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false));
    x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false));
    x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false));
    return (uint32_t)x;
}

// Hot units (r04).  Every builder is TWO launches of the same body: the MAIN launch (one wave per unit, the grid of the frame)
// emits every unit whose records fit its stage and appends the others' ids to the hot list; the HOT launch behind it (a fixed
// grid of one-wave workgroups striding over the list, a stage of kHotStage records, registers to spare) emits those.  One
// kernel would have to carry the hot paths' registers (the spill sort's batches, emit_rounds' ring) through every wave of the
// frame: the float64 12-channel builder went from 75 to 115 VGPRs -- an occupancy step of the headline launch -- when they were
// inlined into it.  On uniform windows the list is empty and the hot launch ends at once.
// The list is kHotLists SUBLISTS with a counter of its own each, 64 bytes apart: a clustered batch defers ten thousand units,
// and device-scope atomics on ONE address retire at ~12 ns each (120 us of a 150 us launch, measured); a unit goes to the
// sublist its id hashes to, a full sublist sends it to the next.  Sublist l is worked off by the kHotGrid / kHotLists hot
// workgroups l, l + 64, ...; each of them takes an EXIT TICKET of the sublist when it is through (r05: one atomic per workgroup
// on 64 addresses, and only on sublists that hold anything), and the one that takes the last ticket clears the sublist -- so the
// list is empty again when the launch ends, whatever the next call on the workspace is: the state is the workspace's (r04 kept
// two lists and a selector bit in the caller's plan that every builder call flipped), the plan is read-only, one plan may
// drive any number of workspaces, and a captured graph of builder calls replays.
// A hot item is ONE piece of a unit (a 64-pixel part or a quarter of one, below).  The unit's main wave has laid the unit out,
// pixel-sorted, in its spill slot before deferring it (r04c); the hot wave finds its pixels' records there by binary search,
// stages and walks them with one lane per pixel.
// A hot item's PIECE code: 0..7 = the 64-pixel part p; 8 + 4 p + s = the s-th 16-pixel quarter of part p.  A part of more
// records than a hot wave's stage holds is deferred as four quarters (r04b): its walks -- a chain as long as the part's longest
// pixel -- split four ways, and a quarter's records mostly fit the stage (LDS walks) where the part's did not (walks through
// the register ring).  item = uid * kHotCodes + code.
__device__ inline int piece_px0(int code) { return code < kHotParts ? code * kWave : ((code - kHotParts) >> 2) * kWave + ((code - kHotParts) & 3) * (kWave / 4); }
__device__ inline int piece_npx(int code) { return code < kHotParts ? kWave : kWave / 4; }
template <bool HOT, typename Body>
__device__ inline void run_units(const BinView &bv, Body body) {
    if constexpr (!HOT) {
        body(chunk_unit((int)(gridDim.x * gridDim.y * gridDim.z)), -1);
    } else {
        const uint32_t l = blockIdx.x % kHotLists, capl = hot_sublist_cap(bv.hot_cap);
        const uint32_t nraw = (uint32_t)__builtin_amdgcn_readfirstlane((int)bv.hot[l * 16]);
        if (nraw == 0u) return;   // (every workgroup of the sublist reads the same count: nothing clears it before all are through)
        const uint32_t n = min(nraw, capl);
        const uint32_t *items = bv.hot + kHotHdrWords + (size_t)l * capl;
        for (uint32_t it = blockIdx.x / kHotLists; it < n; it += gridDim.x / kHotLists) {
            const int item = __builtin_amdgcn_readfirstlane((int)items[it]);
            if (item >= 0) body(item / kHotCodes, item % kHotCodes);   // (< 0: the unused tail of a sublist that filled up)
            wave_phase();
        }
        if (threadIdx.x == 0) {
            const uint32_t ticket = atomicAdd(&bv.hot[(kHotLists + l) * 16], 1u);
            if (ticket + 1u == gridDim.x / kHotLists) {   // the sublist's last workgroup: every other one has read its items
                bv.hot[(kHotLists + l) * 16] = 0u;
                bv.hot[l * 16] = 0u;
            }
        }
    }
}
// main launch: the `nparts` 64-pixel parts of unit `uid` go to the hot list, the parts of `splitmask` as four quarters each.
// False: every sublist is full (the workspace's list holds several times what evrep_plan_init's bound says a batch can defer).
__device__ inline bool defer_unit(const BinView &bv, int uid, int nparts, uint32_t splitmask) {
    const uint32_t capl = hot_sublist_cap(bv.hot_cap);
    // lane 4 p + s: quarter s of part p (a part that is not split: s == 0 stands for the whole part)
    const int lane = threadIdx.x, p = lane >> 2, sq = lane & 3;
    const bool split = (splitmask >> p) & 1u;
    const bool active = p < nparts && (split || sq == 0);
    const uint64_t am = __ballot(active);
    const uint32_t nitems = (uint32_t)__popcll(am), pos = (uint32_t)__popcll(am & ((1ull << lane) - 1ull));
    const uint32_t code = split ? (uint32_t)(kHotParts + 4 * p + sq) : (uint32_t)p;
    uint32_t l = ((uint32_t)uid * 0x9E3779B1u) >> 26;
    for (int tries = 0; tries < kHotLists; ++tries, l = (l + 1) % kHotLists) {   // (wave-uniform)
        uint32_t at = 0;
        if (threadIdx.x == 0) at = atomicAdd(&bv.hot[l * 16], nitems);
        at = (uint32_t)__builtin_amdgcn_readfirstlane((int)at);
        uint32_t *items = bv.hot + kHotHdrWords + (size_t)l * capl;
        if (at + nitems <= capl) {
            if (active) items[at + pos] = (uint32_t)uid * (uint32_t)kHotCodes + code;
            return true;
        }
        if (active && at + pos < capl) items[at + pos] = 0xffffffffu;   // the sublist is full
    }
    return false;
}

// main launch: unit `uid` goes to the hot list untouched (Split::in_hot): `count` items, codes code0 .. code0 + count - 1
__device__ inline bool defer_items(const BinView &bv, int uid, uint32_t code0, uint32_t count) {
    const uint32_t capl = hot_sublist_cap(bv.hot_cap);
    uint32_t l = ((uint32_t)uid * 0x9E3779B1u) >> 26;
    for (int tries = 0; tries < kHotLists; ++tries, l = (l + 1) % kHotLists) {   // (wave-uniform)
        uint32_t at = 0;
        if (threadIdx.x == 0) at = atomicAdd(&bv.hot[l * 16], count);
        at = (uint32_t)__builtin_amdgcn_readfirstlane((int)at);
        uint32_t *items = bv.hot + kHotHdrWords + (size_t)l * capl;
        if (at + count <= capl) {
            if (threadIdx.x < count) items[at + threadIdx.x] = (uint32_t)uid * (uint32_t)kHotCodes + code0 + threadIdx.x;
            return true;
        }
        if (threadIdx.x < count && at + threadIdx.x < capl) items[at + threadIdx.x] = 0xffffffffu;   // the sublist is full
    }
    return false;
}

// grid (ceil(nchunk/span), H, B): unit -> (window, sensor row, `span` consecutive 128-pixel chunks).
// span = 2 gives float32 builders the same 12 KB per wave as float64 ones.
__device__ inline ChunkGeom unit_geom(int H, int W, int nchunk, const UnitCfg &uc, int &chunk, int &nch, int u) {
    ChunkGeom g;
    const int nunit = uc.nunit;   // = units_per_row(nchunk, uc.span, uc.merge), see UnitCfg
    const uint32_t urow = fastdiv((uint32_t)u, uc.nunit_m, uc.nunit_sh);   // u / nunit = window * H + row
    const int ur = u - (int)urow * nunit;
    chunk = ur * uc.span;
    nch = (uc.merge && ur == nunit - 1) ? nchunk - chunk : min(uc.span, nchunk - chunk);
    g.b = (int)fastdiv(urow, uc.h_m, uc.h_sh);
    g.row = (int)urow - g.b * H;
    g.c0 = chunk * kChunkPx;
    g.npix = min(nch * kChunkPx, W - g.c0);
    g.cs = 0; g.ce = 0;
    return g;
}
__device__ inline ChunkGeom unit_geom(int H, int W, int nchunk, const UnitCfg &uc, int &chunk, int u) {
    int nch;
    return unit_geom(H, W, nchunk, uc, chunk, nch, u);
}

// One batch (<= 64 records, record `lane` in r): the records only have to be GROUPED by pixel, in time order inside a
// pixel (the lanes already are in time order: runs are visited in block = time order and are time-ordered inside a key).
// r03: a leader election per pixel in LDS instead of a 7-bit ballot match (56 VALU instructions): in round k the lowest
// unassigned lane of every pixel wins `slot[pixel]` (ds_min), takes rank k and clears the slot; a pixel is done after as
// many rounds as it holds records -- one or two on sparse windows.  The winners keep the pixel's record count and first
// lane in `info[pixel]`; the groups' places in the stage are the prefix sums of the counts over the first lanes; the
// segment list emit_core wants falls out of the same numbers, so the segment-head search over the staged records is
// skipped, and the records go to the stage ONCE, digested (emit_chunk), instead of raw -> read back -> digested -> written.
template <typename OutT, bool HOT>
__device__ inline void group_single_batch(const Rec &r, bool v0, uint32_t nrec, int keybase, int npixu, int segbase,
                                          WaveLds<OutT, HOT> &w, UnitRecs &u) {
    const int lane = threadIdx.x;
    uint32_t *cnt = reinterpret_cast<uint32_t *>(w.segs);
    uint4 *cnt4 = reinterpret_cast<uint4 *>(cnt);
    const uint32_t px = v0 ? (uint32_t)(r.x - keybase) : 0u;
    uint32_t *slot = cnt;                       // [npixu]
    unsigned char *info = reinterpret_cast<unsigned char *>(cnt + npixu);   // [npixu][4]: count, first lane, place
    for (int v = lane; v * 4 < npixu; v += kWave) cnt4[v] = make_uint4(~0u, ~0u, ~0u, ~0u);
    wave_phase();
    bool un = v0;
    uint32_t rk = 0;
    for (uint32_t round = 0; __any(un); ++round) {
        if (un) atomicMin(&slot[px], (uint32_t)lane);
        wave_phase();
        const bool lead = un && slot[px] == (uint32_t)lane;
        wave_phase();
        if (lead) {
            slot[px] = ~0u;
            rk = round;
            info[4 * px] = (unsigned char)(round + 1);
            if (round == 0) info[4 * px + 1] = (unsigned char)lane;
            un = false;
        }
        wave_phase();
    }
    const bool first = v0 && rk == 0u;
    const uint32_t size = first ? (uint32_t)info[4 * px] : 0u;
    const uint32_t goff = wave_incl_scan(size) - size;
    if (first) info[4 * px + 2] = (unsigned char)goff;
    const uint64_t fm = __ballot(first);
    const int nseg = __popcll(fm);
    const int gidx = __popcll(fm & ((1ull << lane) - 1ull));
    wave_phase();
    u.pos = v0 ? (int)((uint32_t)info[4 * px + 2] + rk) : lane;
    wave_phase();     // info is read: the segment list may take its place
    if (first) w.segs[gidx] = make_uint2((uint32_t)(r.x - segbase), goff);   // relative to the builder's own origin
    if (lane == 0) w.segs[nseg] = make_uint2(0u, nrec);
    u.ce = nrec;
    u.nstaged = (int)nrec;
    u.nseg = nseg;
    u.r0 = r;
}

// Key-sorted pass: the records of keys [klo, khi) of window b (consecutive chunks of one sensor row; pixel id of the
// first chunk's first pixel = keybase), gathered from the window's block runs and ordered by pixel, stably (the
// runs are visited in block = time order and are time-ordered inside a key, so equal pixels stay in time order).
//   * run k contributes table[k][klo] .. table[k][khi]; record j of the unit lies in the run whose exclusive
//     length prefix covers j (a chain of conditional sums over <= 16 runs, an LDS search above);
//   * <= 64 records (practically every one-chunk unit of a sparse window, for which this pass is chosen): ONE load per
//     lane, records grouped by pixel with a ballot match (emit_core does not need the groups in pixel order);
//   * <= 128 records in a unit wider than one chunk: two loads per lane, a counting sort over the unit's pixels in
//     the wave's LDS (counters in the not-yet-used segment list), result in evbuf -- in both cases no HBM traffic
//     beyond the one read of the record;
//   * more: the same counting sort in batches of 64, written to the unit's own slot of the spill stream (its
//     position = records of the window with a smaller key = sum over the runs of table[k][klo]: disjoint slots,
//     no atomics, idempotent across builders), then read back like the classic stream.
// c0 = first sensor column of the unit (keybase = row * W + c0): what the 8-byte records are decoded against.
// LAST (r04): the builder only reads the LAST record of a pixel (EventStack: ndarray.put is last-write-wins) -- a unit beyond
// the record stage then needs no order at all: one sweep with an LDS atomicMax per record on (rank, polarity) words, one per
// pixel, and the survivors -- at most one record per pixel -- laid out like a warm unit (emit_warm).  No count sweep, no
// placement, no slot, no deferral, whatever the unit holds.
// Visit (r04): the builder's per-pixel result is an order-free function of the unit's records (TimeSurface: per slice and
// polarity the LAST event at or before the cut) -- a unit beyond the record stage is then not ordered at all: ONE sweep hands
// every record to visit.f(pixel, record, id), which keeps its own words in the part tile (visit.words_per_px per pixel of the
// unit, zeroed here) and whatever it wants of record `id` (its sweep slot, unique, < 64 * batches) in the record stage; the
// kernel emits the unit from them (u.part == -4).  No count sweep, no placement, no slot, no hot launch.
struct NoVisit { static constexpr bool enabled = false; int words_per_px; };
template <typename F>
struct UnitVisit { static constexpr bool enabled = true; F f; int words_per_px; };
template <typename F>
__device__ inline UnitVisit<F> unit_visit(F f, int words_per_px) { return UnitVisit<F>{f, words_per_px}; }

// Split (r05): MOST of the builder's per-pixel result is an order-free function of the unit's records (ERGO-12: ten of twelve
// channels are integer sums, counts, occupancy flags and maxima -- exact in any order), only a FEW records feed accumulators
// that have to run in time order (float64 sums of normalised timestamps over one rank window and polarity class: a sixth of the
// records each).  A unit beyond the record stage is then swept ONCE: split.f(pixel, record, entry) accumulates the record's
// order-free part with LDS atomics on the builder's own words (split.words_per_px per pixel of the unit, at the bottom of the
// wave's LDS, zeroed here) and says whether the record is KEPT for the ordered part; the kept records -- 8 bytes each, `entry`
// -- are compacted, in sweep = time order, into a list behind the words and counted per pixel (w.segs).  The kernel orders the
// list by pixel (a third of the records, out of LDS, no second trip through the block runs), walks it with one lane per pixel
// and emits the unit from its registers (u.part == -5; u.pst = kept records, u.pen = the list's byte offset in the wave's LDS).
// split.begin() is the builder's late set-up (wave-uniform; false: the unit takes the ordered paths); split.done() says whether
// the sweep saw only records it could handle (else the unit takes the ordered paths, in this launch: a builder with a split path
// defers nothing and has no hot launch).  Kept records beyond the list's room go to the unit's slot of the spill stream.
// IN_HOT (r05b): the split sweep runs in the builder's HOT launch instead -- the float32 builders' main waves have neither the LDS
// (half the tile, twice the pixels per unit: the words alone would fill it) nor the registers for it (their budget is the sparse
// paths': inlined, the sweep cost the float32 ERGO-12 main launch an occupancy step and 10-18 % on uniform windows).  A main wave
// hands a unit beyond its record stage to the hot launch AS A WHOLE and at once -- no sweep, no slot (item code kHotWhole;
// st_lane = lane k: the status word of the window's block k, whose kStEscaped bit says that the window holds records the split
// cannot take: its units go the ordered ways) -- and the hot wave, with a stage of kHotSplitStage records and deeper batches,
// sweeps, orders the kept records and emits it.
struct NoSplit { static constexpr bool enabled = false; static constexpr bool in_hot = false; };
// words per pixel of a sliced hot unit in its spill slot: the builder's words, the pixel's kept-record count last; even, so that
// 64-bit words stay 8-byte aligned
__host__ __device__ inline uint32_t split_gwpp(int words_per_px) { return ((uint32_t)words_per_px + 2u) & ~1u; }
struct NoMerge { __device__ inline void operator()(const uint32_t *, uint32_t *) const {} };
struct NoPre { __device__ inline uint2 operator()(const Rec8 &) const { return make_uint2(0u, 0u); } };
template <bool IN_HOT, typename Begin, typename F, typename Done, typename Merge = NoMerge, typename Pre = NoPre, bool SLICEABLE = true>
struct UnitSplit { static constexpr bool enabled = true; static constexpr bool in_hot = IN_HOT; static constexpr bool sliceable = SLICEABLE;   // (time slices need Merge)
                   Begin begin; F f; Done done; int words_per_px; uint32_t st_lane;
                   uint32_t min_rec;   // main launch, IN_HOT: only units of at least this many records are handed over (0: every unit beyond the stage)
                   Merge merge;   // merge(mine, unit): a time slice's words of one pixel into the unit's words in global memory (atomics; sub-waves)
                   Pre pre;       // pre(record) -> 8 bytes the builder wants of the record from global memory (its caller-side time): gathered for every
                                  // batch of a round before the first f() -- all in flight together -- and handed to f as `aux`
                   uint32_t coop_min = 0u;   // main launch, IN_HOT (r06): a unit of at least this many records goes to the hot list as ONE item kHotCoop
                                             // -- for the builder's cooperative launch of several waves per unit -- instead of whole / in time slices (0: never)
};
template <bool IN_HOT = false, typename Begin, typename F, typename Done>
__device__ inline UnitSplit<IN_HOT, Begin, F, Done> unit_split(Begin b, F f, Done d, int words_per_px, uint32_t st_lane = 0u, uint32_t min_rec = 0u) { return UnitSplit<IN_HOT, Begin, F, Done>{b, f, d, words_per_px, st_lane, min_rec, NoMerge(), NoPre()}; }
template <bool IN_HOT, typename Begin, typename F, typename Done, typename Merge>
__device__ inline UnitSplit<IN_HOT, Begin, F, Done, Merge> unit_split_merge(Begin b, F f, Done d, int words_per_px, Merge m) { return UnitSplit<IN_HOT, Begin, F, Done, Merge>{b, f, d, words_per_px, 0u, 0u, m, NoPre()}; }
template <bool IN_HOT, typename Begin, typename F, typename Done>
__device__ inline UnitSplit<IN_HOT, Begin, F, Done, NoMerge, NoPre, false> unit_split_whole(Begin b, F f, Done d, int words_per_px, uint32_t st_lane = 0u, uint32_t min_rec = 0u) { return UnitSplit<IN_HOT, Begin, F, Done, NoMerge, NoPre, false>{b, f, d, words_per_px, st_lane, min_rec, NoMerge(), NoPre()}; }
template <bool IN_HOT, typename Begin, typename F, typename Done, typename Merge, typename Pre>
__device__ inline UnitSplit<IN_HOT, Begin, F, Done, Merge, Pre> unit_split_full(Begin b, F f, Done d, int words_per_px, Merge m, Pre pr) { return UnitSplit<IN_HOT, Begin, F, Done, Merge, Pre>{b, f, d, words_per_px, 0u, 0u, m, pr}; }
#ifndef EVREP_SPLIT_BATCHES
#define EVREP_SPLIT_BATCHES 8
#endif
constexpr int kSplitBatches = EVREP_SPLIT_BATCHES;
// kept records the LDS list holds: 8 bytes each between the builder's words and the pixel counters
__device__ inline uint32_t split_list_cap(uint32_t room, uint32_t wbytes) { return min((room - wbytes) / 8u, (uint32_t)(kSplitBatches * kWave)); }   // kept records are ordered out of registers: at most 8 x 64 of them

// exclusive scan over `npixu` (a multiple of 128) pixel counters in LDS: 16-byte vectors, `per4` consecutive ones per lane
__device__ inline void scan_pixel_counters(uint32_t *cnt, int npixu) {
    const int lane = threadIdx.x;
    uint4 *cnt4 = reinterpret_cast<uint4 *>(cnt);
    const int per4 = npixu / (4 * kWave) + ((npixu % (4 * kWave)) ? 1 : 0);
    uint32_t local = 0;
    for (int k = 0; k < per4; ++k) {
        const int v = lane * per4 + k;
        if (v * 4 < npixu) { const uint4 c = cnt4[v]; local += c.x + c.y + c.z + c.w; }
    }
    uint32_t run = wave_incl_scan(local) - local;
    for (int k = 0; k < per4; ++k) {
        const int v = lane * per4 + k;
        if (v * 4 < npixu) {
            const uint4 c = cnt4[v];
            uint4 o;
            o.x = run; o.y = o.x + c.x; o.z = o.y + c.y; o.w = o.z + c.z;
            run = o.w + c.w;
            cnt4[v] = o;
        }
    }
}

template <typename OutT, bool HOT, bool LAST = false, typename Visit = NoVisit, typename Split = NoSplit>
__device__ inline UnitRecs unit_records(const BinView &bv, const int64_t *__restrict__ off, int b, int NK, int klo, int khi,
                                        int keybase, int npixu, WaveLds<OutT, HOT> &w, int segbase, int c0, int uid,
                                        int npix_out, int part, Visit visit = Visit(), Split split = Split()) {
    const int lane = threadIdx.x;
    UnitRecs u;
    u.sorted = bv.spill; u.cs = 0; u.ce = 0; u.nstaged = kEvStage;
    u.r0 = make_int4(INT32_MIN, 0, 0, 0);
    u.nseg = -1; u.pos = lane; u.deferred = false; u.part = -1; u.hot_lds = false; u.sub = 0;
    if (khi <= klo) return u;
    // the window's extent and the run tables are loaded together (the table address does not depend on the extent;
    // runs beyond the window's block count are masked afterwards): two dependent global latencies, not three
    // (the scalar loads of the extent are issued FIRST: placed behind the table loads the compiler issued them only after
    // waiting for the tables -- a third dependent latency, ~1 us under the builder's own store load)
    const int64_t beg = off[b];
    const int64_t n_win = off[b + 1] - beg;
    uint32_t a = 0, khi_v = 0;
    if (lane < bv.nblk) {
        const uint32_t *tb = bv.table + ((size_t)b * bv.nblk + lane) * ((size_t)NK + 1);
        a = tb[klo];
        khi_v = tb[khi];
    }
    uint32_t len = khi_v - a;
    // (a window longer than the plan's max_events_per_window is the caller's error: its tail was not binned; stay in bounds)
    const int nb = min((int)(((uint32_t)n_win + (1u << bv.chunk_shift) - 1u) >> bv.chunk_shift), bv.nblk);
    if (nb <= 0) return u;
    if (lane >= nb) { a = 0; len = 0; }
    const uint32_t incl = wave_incl_scan(len);
    const uint32_t pre = incl - len;  // lanes >= nb: pre = nrec, never matched
    const uint32_t nrec = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    if (nrec == 0) return u;
    const uint32_t src = (uint32_t)beg + ((uint32_t)lane << bv.chunk_shift) + a - pre;  // record j of the unit, if in run `lane`: src + j
    uint32_t *runs = reinterpret_cast<uint32_t *>(w.evbuf);  // [2][64], only when nb > kBsChainBlocks
    if (nb > kBsChainBlocks) {
        runs[lane] = pre;
        runs[64 + lane] = src;
        wave_phase();
    }
    const Rec8 *__restrict__ s8 = reinterpret_cast<const Rec8 *>(bv.sorted);
    const int4 *evw = bv.ev + beg;
    const int row_base = keybase - c0;
    auto s1_at = [&](uint32_t at) -> Rec { return rec8_unpack(s8[at], row_base, c0, evw); };
    uint32_t *cnt = reinterpret_cast<uint32_t *>(w.segs);  // npixu <= segcap counters: the segment list is built later
    const int nbits = 32 - __builtin_clz((unsigned)npixu - 1u);  // npixu >= 128
    const int per4 = npixu / (4 * kWave) + ((npixu % (4 * kWave)) ? 1 : 0);  // npixu is a multiple of 128: 16-byte vectors per lane
    uint4 *cnt4 = reinterpret_cast<uint4 *>(cnt);
    const int nstage = min(w.nstage, 2 * kEvStage);  // two register batches
    if (!HOT && nrec <= (uint32_t)nstage) {   // (a hot wave always sorts its part out of the unit's records: below)
        // the whole unit is ordered inside LDS: up to two batches of 64 records, held in registers between the count
        // and the placement
        const bool two = nrec > (uint32_t)kWave;  // uniform
        const bool v0 = lane < (int)nrec, v1 = lane + kWave < (int)nrec;
        Rec r = u.r0, r1 = u.r0;
        {
            // the run of record j: every non-empty run writes its index at the position of its first record, an
            // inclusive max-scan over the positions spreads it (a dozen LDS / DPP operations instead of a 3-instruction
            // step per run and record)
            uint32_t *head = cnt + w.segcap;                            // [128], behind the pixel counters
            uint32_t *srcs = reinterpret_cast<uint32_t *>(w.evbuf);     // [64]
            head[lane] = 0u;
            if (two) head[lane + kWave] = 0u;
            srcs[lane] = src;
            wave_phase();
            if (lane < nb && len > 0u && pre < (uint32_t)nstage) head[pre] = (uint32_t)lane;
            wave_phase();
            const uint32_t k0 = wave_incl_max_scan(head[lane]);
            if (v0) r = s1_at(srcs[k0] + (uint32_t)lane);
            if (two) {
                const uint32_t carry = (uint32_t)__builtin_amdgcn_readlane((int)k0, 63);
                const uint32_t k1 = max(carry, wave_incl_max_scan(head[lane + kWave]));
                if (v1) r1 = s1_at(srcs[k1] + (uint32_t)(lane + kWave));
            }
            wave_phase();   // srcs lives in evbuf: read before the placement below writes it
        }
        u.ce = nrec;
        u.nstaged = (int)nrec;
        const uint32_t px = v0 ? (uint32_t)(r.x - keybase) : 0u;
        if (!two) {
            group_single_batch(r, v0, nrec, keybase, npixu, segbase, w, u);
            return u;
        }
        for (int v = lane; v * 4 < npixu; v += kWave) cnt4[v] = make_uint4(0u, 0u, 0u, 0u);
        wave_phase();
        const uint32_t px1 = v1 ? (uint32_t)(r1.x - keybase) : 0u;
        if (v0) atomicAdd(&cnt[px], 1u);
        if (v1) atomicAdd(&cnt[px1], 1u);
        wave_phase();
        // exclusive scan over the pixel counters: 16-byte vectors, `per4` consecutive ones per lane
        {
            uint32_t local = 0;
            for (int k = 0; k < per4; ++k) {
                const int v = lane * per4 + k;
                if (v * 4 < npixu) { const uint4 c = cnt4[v]; local += c.x + c.y + c.z + c.w; }
            }
            uint32_t run = wave_incl_scan(local) - local;
            for (int k = 0; k < per4; ++k) {
                const int v = lane * per4 + k;
                if (v * 4 < npixu) {
                    const uint4 c = cnt4[v];
                    uint4 o;
                    o.x = run; o.y = o.x + c.x; o.z = o.y + c.y; o.w = o.z + c.z;
                    run = o.w + c.w;
                    cnt4[v] = o;
                }
            }
        }
        wave_phase();
        uint32_t rk; bool last;
        wave_match(px, nbits, v0, lane, rk, last);
        uint32_t pos = 0;
        if (v0) { pos = cnt[px] + rk; w.evbuf[pos] = r; }
        wave_phase();   // the second batch goes behind the first inside every pixel
        if (v0 && last) cnt[px] = pos + 1;
        wave_phase();
        wave_match(px1, nbits, v1, lane, rk, last);
        if (v1) w.evbuf[cnt[px1] + rk] = r1;
        wave_phase();
        if (v0) r = w.evbuf[lane];
        u.r0 = r;
        return u;
    }
    // A unit of more records than the stage holds (a HOT unit of a clustered window, r04: a moving edge parallel to the sensor
    // rows puts a thousand records into a handful of units while the window's average is 20): laid out in its slot of the
    // spill stream by a counting sort over the unit's pixels, then staged from there part by part (emit_rounds).  The records
    // are fetched in register batches of kSpillBatch x 64 (8-byte records, every load of a batch in flight together -- r03
    // took one dependent L2 round trip per 64 records, twice); a unit of up to 1024 records is fetched once.
    // A MAIN launch (r04b) sorts the whole unit and emits it part by part from the slot (emit_parts_main) -- such a wave is
    // bound by latency and arithmetic and runs beside the frame's store-bound waves for free, which a separate launch cannot
    // -- unless one of its 64-pixel parts holds more records than the hot stage (tile + record stage): then the unit, sorted
    // into its slot all the same, has its parts appended to the hot list, and the waves of the builder's HOT launch -- which have
    // the registers for the prefetch ring (emit_part) -- each emit ONE piece of it out of the slot.  Four-record batches in a
    // main launch: its register budget is the sparse paths'.
    // the pixels this wave sorts, in pixels of the unit: output pixel o = unit pixel o + (segbase - keybase)
    const int dpx = segbase - keybase;
    const int plo = HOT ? piece_px0(part) + dpx : 0, phi = HOT ? min(piece_px0(part) + piece_npx(part), npix_out) + dpx : npixu;
    u.part = HOT ? part : -2;   // -2: a main launch's unit in its spill slot, -3: in the hot stage (emit_chunk)
#ifndef EVREP_MAIN_SPILL_BATCH
#define EVREP_MAIN_SPILL_BATCH 4
#endif
#ifndef EVREP_HOT_SPILL_BATCH
#define EVREP_HOT_SPILL_BATCH 16   // 8 KB of records in flight per hot wave: a unit of 20 000 records is 20 rounds of latency, not 80
#endif
    constexpr int kSpillBatch = HOT ? EVREP_HOT_SPILL_BATCH : EVREP_MAIN_SPILL_BATCH;   // (a hot wave only sweeps on the split path)
    const uint32_t cs = (uint32_t)beg + (uint32_t)__builtin_amdgcn_readlane((int)wave_incl_scan(a), 63);
    const bool hot_sub = HOT && Split::enabled && Split::in_hot && part >= kHotSub0 && part < kHotSub0 + kHotSubMax;   // wave-uniform
    if (HOT && !(Split::enabled && Split::in_hot && (part == kHotWhole || hot_sub))) {
      if constexpr (HOT) {
        // The unit's MAIN wave has laid its records out in the unit's slot, pixel-sorted (r04c; until then every hot wave
        // sorted its part out of the unit's records itself: two sweeps over ALL of them per piece, ~5 000 of a piece's ~7 000
        // instructions -- and the hot launch is bound by VALU throughput).  Lane l finds the records of its pixel by two
        // 11-step binary searches over the slot's pixel ids: dependent L2 round trips, which the other waves of the SIMD hide.
        const Rec *slot = bv.spill + cs;
        const int px = plo + lane;
        const bool own = px >= 0 && px < phi && px < npixu;
        uint32_t lo0 = 0, hi0 = nrec, lo1 = 0, hi1 = nrec;   // first record with pixel id >= key / >= key + 1
        const int key = keybase + px;
        while (__any(own && (lo0 < hi0 || lo1 < hi1))) {
            if (own && lo0 < hi0) { const uint32_t mid = (lo0 + hi0) >> 1; if (slot[mid].x < key) lo0 = mid + 1; else hi0 = mid; }
            if (own && lo1 < hi1) { const uint32_t mid = (lo1 + hi1) >> 1; if (slot[mid].x <= key) lo1 = mid + 1; else hi1 = mid; }
        }
        w.mark(3);
        u.pst = own ? lo0 : 0u; u.pen = own ? lo1 : 0u;
        u.hot_lds = false;
        u.cs = cs; u.ce = cs + nrec;
        u.nstaged = 0;
        u.dpx = dpx; u.npixu = npixu;
        return u;
      }
    }
    if constexpr (Split::enabled && Split::in_hot && !HOT) {
        // (wave-uniform) a window without escaped polarities: the unit goes to the hot launch's split sweep, whole and untouched
        const uint32_t stw = wave_or(lane < nb ? split.st_lane : 0u);
        // (warm units too -- those that would fit this wave's hot stage: measured, r05b, ordering them here makes the main launch
        //  of a clustered batch 30-40 % longer, which the hot launch they spare does not give back)
        if (!(stw & kStEscaped) && (nrec >= split.min_rec || (split.coop_min != 0u && nrec >= split.coop_min)) && nrec <= 65535u && dpx == 0) {
            // A unit of >= kHotSubMin records is taken in TIME slices (r05b): S hot waves sweep ~1 000 consecutive records each into
            // their own words, merge them into the unit's words in its spill slot (global atomics: sums, flags, maxima), leave their
            // kept records there, and the last one to finish orders the kept records and emits the unit.  One wave's instruction
            // stream bounds a sweep at ~250 instructions per 64 records: a 20 000-record unit of a 1 Mpx circle window took one
            // hot wave 100 us, the whole launch's tail.  This wave clears the slot's header and words (visible at the launch boundary).
            bool ok;
            if (split.coop_min != 0u && nrec >= split.coop_min) {
                ok = defer_items(bv, uid, (uint32_t)kHotCoop, 1u);
            } else if (Split::sliceable && nrec >= kHotSubMin) {
                uint4 *z = reinterpret_cast<uint4 *>(bv.spill + cs);
                const uint32_t nz = (kHotSubHdrBytes + 4u * split_gwpp(split.words_per_px) * (uint32_t)npixu + 15u) / 16u;
                for (uint32_t i = (uint32_t)lane; i < nz; i += kWave) z[i] = make_uint4(0u, 0u, 0u, 0u);
                ok = defer_items(bv, uid, (uint32_t)kHotSub0, hot_sub_count(nrec));
            } else {
                ok = defer_items(bv, uid, (uint32_t)kHotWhole, 1u);
            }
            if (!ok && lane == 0) atomicOr(&bv.stats_rw[(size_t)b * bv.nblk].status, EVREP_ST_HOT_OVERFLOW);
            u.deferred = true;
            u.ce = nrec;
            return u;
        }
    }
    for (int v = lane; v * 4 < npixu; v += kWave) cnt4[v] = make_uint4(0u, 0u, 0u, 0u);
    wave_phase();
    // The sweeps go RUN BY RUN: a batch is up to 64 consecutive records of ONE block run -- address = the run's first record of
    // the unit + lane, a handful of scalar instructions per batch where finding the run of record j of the unit takes a
    // 3-instruction step per run (a 6-step LDS search in windows of more than 16 runs) per record.  kSpillBatch batches are
    // in flight together; runs are visited in block order and a run is time-ordered, so the sweep order is the time order.
    // ... when the runs are long enough to fill batches (a hot unit of a sparse window: a dozen runs of 30-250 records).  A
    // unit whose runs hold a handful of records each (a window of 250 000 events: 31 runs of 4) would load a mostly empty
    // batch per run: there a batch is 64 consecutive records of the UNIT and every lane finds its record's run itself (the
    // readlane chain / LDS search of r03).
    // a hot launch's sub-wave (hot_sub) sweeps records [jlo, jhi) of the unit only: its time slice
    uint32_t jlo = 0, jhi = nrec;
    if (hot_sub) {
        const uint32_t S = hot_sub_count(nrec), qs = hot_sub_quota(nrec, S);
        jlo = min(nrec, (uint32_t)(part - kHotSub0) * qs);
        jhi = min(nrec, jlo + qs);
    }
    const bool by_run = !hot_sub && nrec >= (uint32_t)EVREP_ORDERED_BYRUN * (uint32_t)nb;   // wave-uniform
    const uint32_t run0 = (uint32_t)beg + ((uint32_t)lane << bv.chunk_shift) + a;   // lane k: the unit's first record in run k
    uint32_t *runs2 = cnt + npixu;   // (behind the pixel counters: a main launch places a warm unit over the record stage)
    if (!by_run && nb > kBsChainBlocks) { runs2[lane] = pre; runs2[64 + lane] = src; }
    wave_phase();
    auto src_of = [&](uint32_t j) -> uint32_t {    // the address of record j of the unit in the block runs
        if (nb <= kBsChainBlocks) {
            uint32_t sx = (uint32_t)__builtin_amdgcn_readlane((int)src, 0);
            uint32_t prev = sx;
            for (int k = 1; k < nb; ++k) {
                const uint32_t pk = (uint32_t)__builtin_amdgcn_readlane((int)pre, k);
                const uint32_t sk = (uint32_t)__builtin_amdgcn_readlane((int)src, k);
                sx += (j >= pk) ? sk - prev : 0u;
                prev = sk;
            }
            return sx + j;
        }
        uint32_t lo = 0, hi = (uint32_t)nb;
#pragma unroll
        for (int step = 0; step < 6; ++step) {
            const uint32_t mid = (lo + hi) >> 1;
            const bool go = hi - lo > 1 && runs2[mid] <= j;
            if (hi - lo > 1) { if (go) lo = mid; else hi = mid; }
        }
        return runs2[64 + lo] + j;
    };
    int rk_ = 0;            // the sweep cursor: run, offset inside the run (by run) / record of the unit (wave-uniform)
    uint32_t ro_ = 0;
    uint32_t addr[kSpillBatch], bcnt[kSpillBatch];
    Rec8 q[kSpillBatch];
    auto sweep_begin = [&]() { rk_ = 0; ro_ = by_run ? 0u : jlo; };
    auto skip_empty = [&]() {   // by run: the cursor moves to the next record there is
        uint32_t lk = rk_ < nb ? (uint32_t)__builtin_amdgcn_readlane((int)len, rk_) : 0u;
        while (rk_ < nb && ro_ >= lk) { ++rk_; ro_ = 0; lk = rk_ < nb ? (uint32_t)__builtin_amdgcn_readlane((int)len, rk_) : 0u; }
        return lk;
    };
    auto sweep_done = [&]() -> bool { if (by_run) { skip_empty(); return rk_ >= nb; } return ro_ >= jhi; };
    auto load_batch = [&]() -> bool {   // fills bcnt / q; false: the sweep is over (nothing was filled)
        bool any = false;
        if (by_run) {
#pragma unroll
            for (int sl = 0; sl < kSpillBatch; ++sl) {
                const uint32_t lk = skip_empty();
                bcnt[sl] = 0u; addr[sl] = 0u;
                if (rk_ < nb) {
                    addr[sl] = (uint32_t)__builtin_amdgcn_readlane((int)run0, rk_) + ro_;
                    bcnt[sl] = min(lk - ro_, (uint32_t)kWave);
                    ro_ += kWave;
                    any = true;
                }
            }
#pragma unroll
            for (int sl = 0; sl < kSpillBatch; ++sl) {
                q[sl] = make_uint2(0u, 0u);
                if ((uint32_t)lane < bcnt[sl]) q[sl] = s8[addr[sl] + (uint32_t)lane];
            }
        } else {
#pragma unroll
            for (int sl = 0; sl < kSpillBatch; ++sl) {
                const uint32_t j0 = ro_ + (uint32_t)(sl * kWave);
                bcnt[sl] = j0 < jhi ? min(jhi - j0, (uint32_t)kWave) : 0u;
                q[sl] = make_uint2(0u, 0u);
                if ((uint32_t)lane < bcnt[sl]) q[sl] = s8[src_of(j0 + (uint32_t)lane)];
            }
            any = ro_ < jhi;
            ro_ += (uint32_t)(kSpillBatch * kWave);
        }
        return any;
    };
    auto px_of = [&](const Rec8 &r) -> uint32_t { return ((r.y & 511u) - (uint32_t)c0) & 511u; };   // pixel inside the unit
    if constexpr (Visit::enabled && !HOT) {
        const uint32_t words = (uint32_t)npixu * (uint32_t)visit.words_per_px;
        if (words <= (uint32_t)w.bigsplit * 4u) {   // wave-uniform: the words fit the part tile
            uint4 *t4 = reinterpret_cast<uint4 *>(w.tile);
            for (uint32_t v = (uint32_t)lane; v * 4u < words; v += kWave) t4[v] = make_uint4(0u, 0u, 0u, 0u);
            wave_phase();
            sweep_begin();
            for (uint32_t id0 = 0; load_batch(); id0 += (uint32_t)(kSpillBatch * kWave)) {
#pragma unroll
                for (int sl = 0; sl < kSpillBatch; ++sl)
                    if ((uint32_t)lane < bcnt[sl]) visit.f(px_of(q[sl]), q[sl], id0 + (uint32_t)(sl * kWave + lane));
                if (sweep_done()) break;
            }
            wave_phase();
            u.part = -4;
            u.cs = cs; u.ce = cs + nrec;
            u.nstaged = 0;
            u.dpx = dpx; u.npixu = npixu;
            u.pst = 0; u.pen = 0;
            return u;
        }
    }
    if constexpr (Split::enabled && HOT == Split::in_hot) {
        const uint32_t room = (uint32_t)(reinterpret_cast<const unsigned char *>(w.segs) - reinterpret_cast<const unsigned char *>(w.tile));
        const uint32_t wbytes = ((uint32_t)npixu * (uint32_t)split.words_per_px * 4u + 15u) & ~15u;
        // 16-bit fields in the builder's words: a unit of up to 65 535 records.  (wave-uniform)
        if (wbytes + 64u * 8u <= room && nrec <= 65535u && dpx == 0 && split.begin()) {
            const uint32_t lcap = split_list_cap(room, wbytes);
            uint4 *t4 = reinterpret_cast<uint4 *>(w.tile);
            for (uint32_t v = (uint32_t)lane; v * 16u < wbytes; v += kWave) t4[v] = make_uint4(0u, 0u, 0u, 0u);
            wave_phase();
            // the kept records: the first `lcap` in the list behind the words, later ones (a hot unit) in the unit's own slot of
            // the spill stream -- 16 bytes per record of the unit: the lower half takes the list, the upper half its ordered copy
            uint2 *list = reinterpret_cast<uint2 *>(reinterpret_cast<unsigned char *>(w.tile) + wbytes);
            uint2 *glist = reinterpret_cast<uint2 *>(bv.spill + cs);
            // a sub-wave's kept records all go to ITS region of the unit's slot (behind the header and the unit's words): the
            // compacted kept records of records [jlo, jhi), from offset jlo
            unsigned char *slot = reinterpret_cast<unsigned char *>(bv.spill + cs);
            const uint32_t gwpp = split_gwpp(split.words_per_px);   // a sliced unit's words per pixel in its slot: the builder's, [pad,] the pixel's kept records
            uint2 *sublists = reinterpret_cast<uint2 *>(slot + ((kHotSubHdrBytes + 4u * gwpp * (uint32_t)npixu + 15u) & ~15u));
            if (hot_sub) glist = sublists + jlo;
            const uint32_t lcap_eff = hot_sub ? 0u : lcap;
            uint32_t nk = 0;
            sweep_begin();
            while (load_batch()) {
                uint2 aux[kSpillBatch];
#pragma unroll
                for (int sl = 0; sl < kSpillBatch; ++sl) {
                    aux[sl] = make_uint2(0u, 0u);
                    if ((uint32_t)lane < bcnt[sl]) aux[sl] = split.pre(q[sl]);
                }
#pragma unroll
                for (int sl = 0; sl < kSpillBatch; ++sl) {
                    if (bcnt[sl] == 0u) break;   // uniform
                    bool keep = false;
                    uint2 e = make_uint2(0u, 0u);
                    if ((uint32_t)lane < bcnt[sl]) keep = split.f(px_of(q[sl]), q[sl], e, aux[sl]);
                    const uint64_t km = __ballot(keep);
                    if (km) {
                        const uint32_t at = nk + (uint32_t)__popcll(km & ((1ull << lane) - 1ull));
                        if (keep) {
                            if (at < lcap_eff) list[at] = e; else glist[at] = e;
                            atomicAdd(&cnt[e.y], 1u);
                        }
                        nk += (uint32_t)__popcll(km);
                    }
                }
                if (sweep_done()) break;
            }
            wave_phase();
            if (hot_sub) {
                // merge this slice's words and kept counts into the unit's (8 words per pixel in the slot: the builder's
                // split.words_per_px <= 7, then the pixel's kept records), then take the unit's ticket
                uint32_t *gw = reinterpret_cast<uint32_t *>(slot + kHotSubHdrBytes);
                uint32_t *hdr = reinterpret_cast<uint32_t *>(slot);
                const uint32_t *lw = reinterpret_cast<const uint32_t *>(w.tile);
                for (uint32_t px = (uint32_t)lane; px < (uint32_t)npixu; px += kWave) split.merge(lw + px * (uint32_t)split.words_per_px, gw + px * gwpp);
                for (uint32_t px = (uint32_t)lane; px < (uint32_t)npixu; px += kWave) { const uint32_t kc = cnt[px]; if (kc) atomicAdd(gw + px * gwpp + (gwpp - 1u), kc); }
                const uint32_t S = hot_sub_count(nrec);
                uint32_t ticket = 0;
                __threadfence();   // this wave's list and merges before its ticket
                if (lane == 0) {
                    hdr[1 + (part - kHotSub0)] = nk;
                    __threadfence();
                    ticket = atomicAdd(hdr, 1u);
                }
                ticket = (uint32_t)__builtin_amdgcn_readfirstlane((int)ticket);
                if (ticket + 1u != S) { u.deferred = true; return u; }   // (a later slice's wave emits the unit)
                __threadfence();   // the other slices' words, lists and counts are complete: read them past this CU's L1
                uint32_t *lwm = reinterpret_cast<uint32_t *>(w.tile);
                for (uint32_t px = (uint32_t)lane; px < (uint32_t)npixu; px += kWave) {
                    for (uint32_t k = 0; k < (uint32_t)split.words_per_px; ++k)
                        lwm[px * (uint32_t)split.words_per_px + k] = __hip_atomic_load(gw + px * gwpp + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    cnt[px] = __hip_atomic_load(gw + px * gwpp + (gwpp - 1u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                uint32_t tot = 0;
                if (lane < (int)S) tot = __hip_atomic_load(hdr + 1 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const uint32_t incl = wave_incl_scan(tot);
                nk = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
                wave_phase();
                u.sub = (int)S;
            }
            if (split.done()) {   // wave-uniform
                u.part = -5;
                u.cs = cs; u.ce = cs + nrec;
                u.nstaged = 0;
                u.dpx = dpx; u.npixu = npixu;
                u.pst = nk; u.pen = wbytes;
                return u;
            }
            // the ordered paths after all: they count from zero
            for (int v = lane; v * 4 < npixu; v += kWave) cnt4[v] = make_uint4(0u, 0u, 0u, 0u);
            wave_phase();
        }
        if constexpr (HOT) {   // (cannot happen: the main wave has checked what the sweep checks; a hot wave has no ordered way for a whole unit)
            if (lane == 0) atomicOr(&bv.stats_rw[(size_t)b * bv.nblk].status, EVREP_ST_HOT_OVERFLOW);
            u.deferred = true;
            return u;
        }
    }
    if constexpr (LAST && !HOT) {
        if (npixu <= w.bigcap) {   // wave-uniform: the survivors fit the hot stage (always, for stacks of >= 8 levels)
            sweep_begin();
            while (load_batch()) {
#pragma unroll
                for (int sl = 0; sl < kSpillBatch; ++sl)
                    if ((uint32_t)lane < bcnt[sl]) atomicMax(&cnt[px_of(q[sl])], (((q[sl].y >> 11) + 1u) << 2) | ((q[sl].y >> 9) & 3u));
                if (sweep_done()) break;
            }
            wave_phase();
            uint32_t nlast = 0;
            for (int p0 = 0; p0 < npixu; p0 += kWave) {   // npixu is a multiple of 128
                const int px = p0 + lane;
                const uint32_t v = cnt[px];
                const uint64_t m = __ballot(v != 0u);
                const uint32_t pos = nlast + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
                if (v != 0u) {
                    const int rank = (int)(v >> 2) - 1;
                    const uint32_t p2 = v & 3u;
                    *w.big_at(pos) = make_int4(keybase + px, rank, 0, p2 == 3u ? evw[rank].w : (int)p2 - 1);
                }
                cnt[px] = pos + (v != 0u ? 1u : 0u);   // the pixel's END, as emit_warm reads it
                nlast += (uint32_t)__popcll(m);
            }
            wave_phase();
            u.part = -3;
            u.cs = cs; u.ce = cs + nlast;
            u.nstaged = 0;
            u.dpx = dpx; u.npixu = npixu;
            u.pst = 0; u.pen = 0;
            return u;
        }
    }
    sweep_begin();
    bool resident = false;
    for (int round = 0; load_batch(); ++round) {
#pragma unroll
        for (int sl = 0; sl < kSpillBatch; ++sl)
            if ((uint32_t)lane < bcnt[sl]) atomicAdd(&cnt[px_of(q[sl])], 1u);
        // the whole unit in ONE round: the placement below finds it still in the registers
        if (sweep_done()) { resident = round == 0; break; }
    }
    wave_phase();
    {
        uint32_t local = 0;
        for (int k = 0; k < per4; ++k) {
            const int v = lane * per4 + k;
            if (v * 4 < npixu) { const uint4 c = cnt4[v]; local += c.x + c.y + c.z + c.w; }
        }
        uint32_t run = wave_incl_scan(local) - local;
        for (int k = 0; k < per4; ++k) {
            const int v = lane * per4 + k;
            if (v * 4 < npixu) {
                const uint4 c = cnt4[v];
                uint4 o;
                o.x = run; o.y = o.x + c.x; o.z = o.y + c.y; o.w = o.z + c.z;
                run = o.w + c.w;
                cnt4[v] = o;
            }
        }
    }
    wave_phase();
    // a main launch keeps a unit that fits the hot stage (tile + record stage) in LDS altogether: no slot, no second trip
    const bool in_lds = !HOT && nrec <= (uint32_t)w.bigcap;   // wave-uniform
    bool defer = false;
    uint32_t defer_mask = 0u;
    if (!HOT && !in_lds) {
        bool fits = true;   // wave-uniform: every 64-pixel part of the output fits the hot stage
        uint32_t splitmask = 0u;   // parts beyond a HOT wave's stage (tile + kHotStage records): deferred as quarters
        for (int o0 = 0; o0 < npix_out; o0 += kWave) {
            const int lo = o0 + dpx, hi = min(o0 + kWave, npix_out) + dpx;
            const uint32_t b0 = lo < npixu ? cnt[lo] : nrec, b1 = hi < npixu ? cnt[hi] : nrec;
            fits = fits && b1 - b0 <= (uint32_t)(EVREP_DEFER_MULT * w.bigcap);
            if (b1 - b0 > (uint32_t)(w.bigcap - w.nstage + kHotStage)) splitmask |= 1u << (o0 / kWave);
        }
#ifdef EVREP_NO_QUARTERS
        splitmask = 0u;
#endif
#ifdef EVREP_NO_DEFER
        fits = true;
#endif
        // a builder with a split path has no hot launch behind it: the few units that fall back here (escaped polarity values,
        // more than 65 535 records) are emitted from their slot by this wave, whatever they hold
        if constexpr (Split::enabled && !Split::in_hot) fits = true;
        defer = !fits;   // (deferred AFTER the placement below: the hot waves read the unit from its slot)
        defer_mask = splitmask;
    }
    wave_phase();
    volatile uint32_t *vcnt = cnt;
    if (!resident) sweep_begin();
    for (bool more = resident ? true : load_batch(); more; more = resident ? false : load_batch()) {
#pragma unroll
        for (int sl = 0; sl < kSpillBatch; ++sl) {
            if (bcnt[sl] == 0u) break;   // uniform
            uint32_t px = px_of(q[sl]);
            const bool valid = (uint32_t)lane < bcnt[sl] && (int)px >= plo && (int)px < phi;
            if (!__any(valid)) continue;
            if (!valid) px = 0u;
            uint32_t rk; bool last;
            wave_match(px, nbits, valid, lane, rk, last);
            uint32_t pos = 0;
            if (valid) {
                pos = vcnt[px] + rk;
                const Rec rec = rec8_unpack(q[sl], row_base, c0, evw);
                if (in_lds) *w.big_at(pos) = rec; else bv.spill[cs + pos] = rec;
            }
            __builtin_amdgcn_wave_barrier();
            if (valid && last) vcnt[px] = pos + 1;
            __builtin_amdgcn_wave_barrier();
        }
    }
    wave_phase();
    if (defer) {   // wave-uniform: a part beyond this wave's stage -- the unit's pieces go to the hot launch, which finds them in the slot
        if (!defer_unit(bv, uid, (npix_out + kWave - 1) / kWave, defer_mask) && lane == 0)
            atomicOr(&bv.stats_rw[(size_t)b * bv.nblk].status, EVREP_ST_HOT_OVERFLOW);   // the unit's pixels stay unwritten: the caller is told
        u.deferred = true;
        u.ce = nrec;
        return u;
    }
    u.pst = 0; u.pen = 0;   // (a main launch reads the cursors -- now every pixel's END -- per part)
    if (in_lds) u.part = -3;
    u.hot_lds = false;
    // the wave reads back what its own lanes stored: same CU, same vector L1 -- workgroup-scope release / acquire
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    u.cs = cs;
    u.ce = cs + nrec;
    u.nstaged = 0;   // nothing of it is in the wave's stage: emit_part / emit_parts_main stage it part by part
    u.dpx = dpx; u.npixu = npixu;
    return u;
}

// Classic passes (the stream is pixel-sorted already): records [64, min(nrec, stage)) of the unit go to the wave's LDS
// stage RAW, every load in flight before the first write (r03).  Until then a dense unit read everything behind its first
// 64 records inside the divergent segment walks -- a global load and, for the builders whose digest divides, a float64
// division per step of the longest segment of the wave (70 % of the records of a 640x480 window of 500 000 events).
// emit_chunk digests the staged batches one record per lane once the segment heads are listed.
template <typename OutT, bool HOT>
__device__ inline void stage_classic(UnitRecs &u, WaveLds<OutT, HOT> &w) {
    const int nrec = (int)(u.ce - u.cs), lane = threadIdx.x;
    const int n = min(nrec, w.nstage);
    // wave-uniform.  A 128-record stage (every builder on windows of > 28 records per unit) is only filled when it takes
    // the WHOLE unit -- then the unit is "fully staged" exactly as it is after the key-sorted pass, and builders whose
    // arithmetic depends on that (TimeSurface's factorised exponentials) give the same bits under every binning pass;
    // the 256-record stage of the builders that ask for it (unit_cfg) is filled as far as it goes.
    if (n <= kWave || (w.nstage <= 2 * kEvStage && nrec > w.nstage)) return;
    // stages of up to 256 records: three batches behind the register batch
    const Rec *src = u.sorted + u.cs + lane;
    const bool h1 = kWave + lane < n, h2 = 2 * kWave + lane < n, h3 = 3 * kWave + lane < n;
    Rec t1 = make_int4(0, 0, 0, 0), t2 = t1, t3 = t1;
    if (h1) t1 = src[kWave];
    if (h2) t2 = src[2 * kWave];
    if (h3) t3 = src[3 * kWave];
    if (h1) w.evbuf[kWave + lane] = t1;
    if (h2) w.evbuf[2 * kWave + lane] = t2;
    if (h3) w.evbuf[3 * kWave + lane] = t3;
    u.nstaged = n;
}

// The front end of every tile builder: the unit's geometry and its pixel-sorted records, from either binning pass.
// uid = the unit's id (run_units).  A main launch (HOT false) defers a unit of more records than its stage (u.deferred).
template <typename OutT, bool HOT, bool LAST = false, typename Visit = NoVisit, typename Split = NoSplit>
__device__ inline UnitRecs unit_front(const BinView &bv, const int64_t *__restrict__ off, int H, int W, int nchunk, const UnitCfg &uc,
                                      WaveLds<OutT, HOT> &w, ChunkGeom &g, int uid, int part, Visit visit = Visit(), Split split = Split()) {
    int chunk, nch;
    g = unit_geom(H, W, nchunk, uc, chunk, nch, uid);
    if (bv.fused) {
        const int klo = g.row * nchunk + chunk, khi = klo + nch;
        const UnitRecs u = unit_records<OutT, HOT, LAST, Visit, Split>(bv, off, g.b, H * nchunk, klo, khi, g.row * W + g.c0, w.segcap, w, g.row * W + g.c0, g.c0, uid, g.npix, part, visit, split);
        g.cs = u.cs; g.ce = u.ce;
        return u;
    }
    const uint32_t *co = bv.chunk_off + ((size_t)g.b * H + g.row) * (nchunk + 1);
    g.cs = co[chunk];
    g.ce = co[chunk + nch];
    UnitRecs u;
    u.sorted = bv.sorted; u.cs = g.cs; u.ce = g.ce; u.nstaged = kEvStage;
    u.r0 = make_int4(INT32_MIN, 0, 0, 0);
    u.nseg = -1; u.pos = (int)threadIdx.x; u.deferred = false; u.part = -1; u.hot_lds = false; u.sub = 0;
    if ((int)threadIdx.x < (int)(g.ce - g.cs)) u.r0 = bv.sorted[g.cs + threadIdx.x];
    stage_classic(u, w);
    return u;
}

// The window statistics a builder needs: the classic passes publish WindowMeta; after the key-sorted pass the wave
// merges the window's block statistics itself (one lane per block, DPP reductions of the fields actually used).
// The loads are issued by meta_prefetch() -- at the top of the kernel, together with the unit's table loads, so that
// their latency is not a dependent step of its own -- and consumed by meta_finish().
struct MetaRaw {
    int4 q0, q1, q2;
};
__device__ inline MetaRaw meta_prefetch(const BinView &bv, int b) {
    MetaRaw r;
    r.q0 = make_int4(0, 0, 0, 0); r.q1 = r.q0; r.q2 = r.q0;
    if (bv.fused) {
        if ((int)threadIdx.x < bv.nblk) {   // blocks beyond the window's own are masked by meta_finish
            const int4 *sp = reinterpret_cast<const int4 *>(bv.stats + (size_t)b * bv.nblk + threadIdx.x);
            r.q0 = sp[0]; r.q1 = sp[1]; r.q2 = sp[2];
        }
    } else if (threadIdx.x == 0) {
        const int4 *mp = reinterpret_cast<const int4 *>(bv.meta + b);
        r.q0 = mp[0]; r.q1 = mp[1]; r.q2 = mp[2];
    }
    return r;
}
__device__ inline WindowMeta meta_finish(const BinView &bv, const int64_t *__restrict__ off, int b, const MetaRaw &r) {
    WindowMeta m;
    if (!bv.fused) {   // WindowMeta's first 40 bytes, read by lane 0
        m.tmin = __builtin_amdgcn_readfirstlane(r.q0.x); m.tmax = __builtin_amdgcn_readfirstlane(r.q0.y);
        m.xmin = __builtin_amdgcn_readfirstlane(r.q0.z); m.xmax = __builtin_amdgcn_readfirstlane(r.q0.w);
        m.ymin = __builtin_amdgcn_readfirstlane(r.q1.x); m.ymax = __builtin_amdgcn_readfirstlane(r.q1.y);
        m.neg_flags = (uint32_t)__builtin_amdgcn_readfirstlane(r.q1.z);
        m.oob_flags = (uint32_t)__builtin_amdgcn_readfirstlane(r.q1.w);
        m.status = (uint32_t)__builtin_amdgcn_readfirstlane(r.q2.x); m.n_valid = __builtin_amdgcn_readfirstlane(r.q2.y);
        return m;
    }
    const int lane = threadIdx.x;
    const int64_t n_win = off[b + 1] - off[b];
    const int nb = min((int)(((uint32_t)n_win + (1u << bv.chunk_shift) - 1u) >> bv.chunk_shift), bv.nblk);
    BlockStats st;
    stats_identity(st);
    if (lane < nb) {
        st.tmin = r.q0.x; st.tmax = r.q0.y; st.xmin = r.q0.z; st.xmax = r.q0.w;
        st.ymin = r.q1.x; st.ymax = r.q1.y; st.neg_flags = (uint32_t)r.q1.z; st.oob_flags = (uint32_t)r.q1.w;
        st.status = (uint32_t)r.q2.x; st.n_valid = r.q2.y;
    }
    // The events of a window are normally time-sorted, and the blocks cut the window in index order: the window's first
    // timestamp is block 0's minimum, its last one the last block's maximum -- two readlanes instead of two DPP reductions
    // on every builder wave's critical path.  A window the status word reports as unsorted takes the reductions (below).
    m.tmin = nb > 0 ? __builtin_amdgcn_readlane(st.tmin, 0) : INT32_MAX;
    m.tmax = nb > 0 ? __builtin_amdgcn_readlane(st.tmax, min(nb, kWave) - 1) : INT32_MIN;
    if (nb > kWave) {   // wave-uniform (r06: windows of up to 128 block runs, stream builders only): blocks 64 .. nb - 1, one per lane, merged in
        BlockStats s2;
        stats_identity(s2);
        if (kWave + lane < nb) {
            const int4 *sp = reinterpret_cast<const int4 *>(bv.stats + (size_t)b * bv.nblk + kWave + lane);
            const int4 a0 = sp[0], a1 = sp[1], a2 = sp[2];
            s2.tmin = a0.x; s2.tmax = a0.y; s2.xmin = a0.z; s2.xmax = a0.w;
            s2.ymin = a1.x; s2.ymax = a1.y; s2.neg_flags = (uint32_t)a1.z; s2.oob_flags = (uint32_t)a1.w;
            s2.status = (uint32_t)a2.x; s2.n_valid = a2.y;
        }
        m.tmax = __builtin_amdgcn_readlane(s2.tmax, nb - kWave - 1);
        stats_merge(st, s2);
    }
    m.xmin = wave_min(st.xmin); m.xmax = wave_max(st.xmax);
    m.ymin = wave_min(st.ymin); m.ymax = wave_max(st.ymax);
    m.neg_flags = wave_or(st.neg_flags); m.oob_flags = wave_or(st.oob_flags);
    m.status = wave_or(st.status); m.n_valid = wave_sum(st.n_valid);
    if (m.status & EVREP_ST_UNSORTED) {   // wave-uniform, rare: t.min() / t.max() of a window in any order (MDES accepts it)
        m.tmin = wave_min(st.tmin);
        m.tmax = wave_max(st.tmax);
    }
    return m;
}
__device__ inline WindowMeta window_meta(const BinView &bv, const int64_t *__restrict__ off, int b) {
    return meta_finish(bv, off, b, meta_prefetch(bv, b));
}

// The sparse emit (r03): the unit's <= 64 non-empty pixels sit in a value list in LDS (entry `slot` = the C values of
// one pixel, in the lanes' order; it lives where the part tile would) and map[pixel] names the slot of a pixel.  Empty
// pixels name entry PP, which is the wave's background vector (w.bg lies right behind the tile).  Every 16-byte vector
// of the unit's output is gathered and stored, four wave-instructions at a time: the unit leaves the wave as ONE burst
// of coalesced 1 KiB stores, with no zero fill of a tile and no part sequence in between -- a compact store phase is
// what store pacing needs (WaveLds::pace), and what lets a float64 unit span two chunks.
// Requires whole 16-byte vectors per pixel (C * sizeof(OutT) % 16 == 0) and a 16-byte aligned destination.
// Vector v = 64 q + lane belongs to pixel v / vpp, piece v % vpp: both advance by constants per q.
template <typename OutT>
__device__ inline void sparse_store(const OutT *vlist, const unsigned char *map, int npix, int C, OutT *__restrict__ dst) {
    constexpr int V = 16 / (int)sizeof(OutT);
    typedef float nt4 __attribute__((ext_vector_type(4)));
    const int lane = threadIdx.x;
    const int vpp = C / V;                              // vectors per pixel (<= 8)
    const uint32_t inv = (65536u + (uint32_t)vpp - 1u) / (uint32_t)vpp;  // x / vpp == (x * inv) >> 16 for x < 4096
    const int nvec = npix * vpp;
    const int dq = (int)((64u * inv) >> 16), dr = 64 - dq * vpp;   // 64 = dq * vpp + dr
    const nt4 *vl4 = reinterpret_cast<const nt4 *>(vlist);
    nt4 *o4 = reinterpret_cast<nt4 *>(dst) + lane;
    int pixel = (int)(((uint32_t)lane * inv) >> 16);
    int sub = lane - pixel * vpp;
    auto gather = [&]() -> nt4 {
        const nt4 r = vl4[(int)map[pixel] * vpp + sub];
        sub += dr; pixel += dq;
        if (sub >= vpp) { sub -= vpp; ++pixel; }
        return r;
    };
    int v0 = 0;
    for (; v0 + 4 * kWave <= nvec; v0 += 4 * kWave) {   // whole groups: no lane test
        const nt4 a = gather(), b = gather(), c = gather(), d = gather();
#if EVREP_NT_STORES
        __builtin_nontemporal_store(a, o4 + v0); __builtin_nontemporal_store(b, o4 + v0 + kWave);
        __builtin_nontemporal_store(c, o4 + v0 + 2 * kWave); __builtin_nontemporal_store(d, o4 + v0 + 3 * kWave);
#else
        o4[v0] = a; o4[v0 + kWave] = b; o4[v0 + 2 * kWave] = c; o4[v0 + 3 * kWave] = d;
#endif
    }
    for (; v0 < nvec; v0 += kWave) {
        if (v0 + lane < nvec) {
            const nt4 a = gather();
#if EVREP_NT_STORES
            __builtin_nontemporal_store(a, o4 + v0);
#else
            o4[v0] = a;
#endif
        }
    }
}

// The shared back end of every builder.  `reduce(jb, je, get, vals)` turns one pixel's records
// [jb, je) (time-ordered; get(j) fetches record j of the chunk) into that pixel's C output values;
// pixels without records keep the background.  `r0` = record `lane` of the chunk, loaded by the
// caller before its own independent loads.  Pixel offsets outside [0, npix) (TORE's straddle)
// are ignored.
// `post_heads()` runs once the segment heads are listed (the stage may then be rewritten).
// `get_staged(j)` = get(j) for a unit whose records are all staged in LDS (every unit of <= 64 records): no second source,
// so the segment walks read LDS with ds_read instead of flat loads through a two-address-space pointer.
// `nseg_pre` >= 0: the front end has listed the segments already (UnitRecs::nseg).
template <typename OutT, int CMAX, bool HOT, typename KeyAt, typename RecAt, typename RecStaged, typename PostHeads, typename Reduce>
__device__ inline void emit_core(uint32_t nrec, int nseg_pre, bool all_staged, KeyAt key_at, RecAt get, RecStaged get_staged,
                                 PostHeads post_heads, int key0, int npix, int C, OutT *__restrict__ dst, WaveLds<OutT, HOT> &w,
                                 const OutT *bg, Reduce reduce) {
    const int lane = threadIdx.x;
    const int PP = w.partpx;  // pixels per part tile (wave-uniform)
    constexpr int V = 16 / (int)sizeof(OutT);
    // units of <= 64 NON-EMPTY PIXELS leave through the sparse emit when a pixel is a whole number of 16-byte vectors; the
    // others through the part tiles.  <= 64 records settle it at once; a unit of up to two staged batches (the reference's
    // own Gen1 shape: ~69 records in ~53 pixels) is decided once its segment heads are counted (r03).
    const bool sparse_ok = EVREP_SPARSE_EMIT && (C % V) == 0 && (reinterpret_cast<uintptr_t>(dst) & 15u) == 0 &&
                           (bg == nullptr || bg == w.bg);
    const int sparse_cap = min(kWave, PP);   // the value list lives in the part tile: PP entries, one lane each
    const bool sparse_late = sparse_ok && nrec > (uint32_t)sparse_cap && nrec <= 2u * kWave && all_staged;
    bool sparse = sparse_ok && nrec <= (uint32_t)sparse_cap;
    // a zero tile is filled at once (it overlaps the record load); a background that had to be
    // loaded is filled after the segment heads are listed, when it has arrived behind the records
    if (!sparse && !sparse_late && (!bg || nrec == 0)) tile_fill(w.tile, min(PP, npix), C, bg);
    const uint32_t empty4 = (uint32_t)PP * 0x01010101u;   // four map bytes naming the background entry
    if ((sparse || sparse_late) && !bg && lane * V < EVREP_MAX_CHANNELS)   // a zero background: entry PP = w.bg has to hold it
        reinterpret_cast<float4 *>(w.bg)[lane] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (nrec == 0) {  // empty chunk: the background is streamed for every pixel
        if (sparse) {
            unsigned char *map = reinterpret_cast<unsigned char *>(w.segs);
            for (int i = lane; i * 4 < npix; i += kWave) reinterpret_cast<uint32_t *>(map)[i] = empty4;
            wave_phase();
            w.pace();
            sparse_store(w.tile, map, npix, C, dst);
            return;
        }
        wave_phase();
        w.pace();
        for (int part = 0; part * PP < npix; ++part)
            tile_store(w.tile, min(PP, npix - part * PP) * C, dst + (size_t)part * PP * C);
        return;
    }
    // segment heads = runs of equal pixel id among the sorted records
    int nseg = 0;
    if (nseg_pre >= 0) {
        nseg = nseg_pre;
    } else {
        int carry = INT32_MIN;
        for (uint32_t j0 = 0; j0 < nrec; j0 += kWave) {
            const uint32_t j = j0 + lane;
            const bool valid = j < nrec;
            int key = INT32_MIN;
            if (valid) key = key_at(j);
            int prev = __shfl_up(key, 1, 64);
            if (lane == 0) prev = carry;
            const bool head = valid && key != prev;
            const uint64_t hm = __ballot(head);
            if (head) {
                const int idx = nseg + __popcll(hm & ((1ull << lane) - 1ull));
                if (idx < w.segcap) w.segs[idx] = make_uint2((uint32_t)(key - key0), j);
            }
            nseg += __popcll(hm);
            carry = __shfl(key, 63, 64);
        }
        if (nseg > w.segcap) nseg = w.segcap;  // cannot happen: a unit never holds more distinct pixels
        if (lane == 0) w.segs[nseg] = make_uint2(0u, nrec);
    }
    if (sparse_late) sparse = nseg <= sparse_cap;   // wave-uniform
    if (!sparse && (bg || sparse_late)) tile_fill(w.tile, min(PP, npix), C, bg);
    wave_phase();
    post_heads();
    w.mark(2);

    if (sparse) {
        // one lane per non-empty pixel, reduced once; the values go to the lane's entry of the value list (it lives
        // where the part tile would), the pixel's map byte names the lane (the map takes the place of the segment
        // list, which is spent once every lane holds its segment)
        OutT vals[CMAX];
        int px = -1;
        if (lane < nseg) {
            const uint2 sg = w.segs[lane];
            px = (int)sg.x;
            if (px >= 0 && px < npix) reduce(sg.y, w.segs[lane + 1].y, get_staged, vals); else px = -1;
        }
        w.mark(7);
        wave_phase();
        unsigned char *map = reinterpret_cast<unsigned char *>(w.segs);
        for (int i = lane; i * 4 < npix; i += kWave) reinterpret_cast<uint32_t *>(map)[i] = empty4;
        wave_phase();
        if (px >= 0) {
            map[px] = (unsigned char)lane;
            OutT *mine = w.tile + (size_t)lane * C;
#pragma unroll
            for (int c = 0; c < CMAX; ++c) if (c < C) mine[c] = vals[c];
        }
        wave_phase();
        w.mark(3);
        w.pace();
        w.mark(4);
        sparse_store(w.tile, map, npix, C, dst);
        w.mark(5);
    } else if (nseg <= kWave) {
        // one lane per non-empty pixel, reduced once; pixels of later parts wait in registers
        OutT vals[CMAX];
        int px = -1;
        if (lane < nseg) {
            const uint2 sg = w.segs[lane];
            px = (int)sg.x;
            if (px >= 0 && px < npix) {
                // a fully staged unit walks LDS (ds_read); the two-source form reads through a generic pointer (flat_load)
                if (all_staged) reduce(sg.y, w.segs[lane + 1].y, get_staged, vals); else reduce(sg.y, w.segs[lane + 1].y, get, vals);
            } else px = -1;
        }
        for (int part = 0; part * PP < npix; ++part) {
            const int np = min(PP, npix - part * PP);
            if (part) { wave_phase(); tile_fill(w.tile, np, C, bg); wave_phase(); }
            const int q = px - part * PP;
            if (px >= 0 && q >= 0 && q < PP) {
                OutT *mine = w.tile + (size_t)q * C;
#pragma unroll
                for (int c = 0; c < CMAX; ++c) if (c < C) mine[c] = vals[c];
            }
            wave_phase();
            if (part == 0) w.pace();
            tile_store(w.tile, np * C, dst + (size_t)part * PP * C);
        }
        w.mark(5);
    } else {
        // dense chunk: one pass per part tile, each segment reduced in the pass of its own part.  The segment list is
        // pixel-ordered, so a part owns the contiguous range [sb, se) of it -- at most PP entries, one per lane,
        // every lane of the range busy (r02: striding the whole list and skipping the other part's entries left half
        // of the lanes idle in each of twice as many reduce rounds)
        int sb = 0;
        for (int part = 0; part * PP < npix; ++part) {
            const int np = min(PP, npix - part * PP);
            if (part) { wave_phase(); tile_fill(w.tile, np, C, bg); wave_phase(); }
            int se = sb;
            for (int k0 = sb; k0 < nseg; k0 += kWave) {
                const int k = k0 + lane;
                const int c = __popcll(__ballot(k < nseg && (int)w.segs[k].x < (part + 1) * PP));
                se += c;
                if (c < kWave) break;
            }
            if (part == 0) w.mark(7);
            for (int k = sb + lane; k < se; k += kWave) {
                const uint2 sg = w.segs[k];
                const int q = (int)sg.x - part * PP;
                if (q < 0 || q >= np) continue;
                OutT vals[CMAX];
                if (all_staged) reduce(sg.y, w.segs[k + 1].y, get_staged, vals); else reduce(sg.y, w.segs[k + 1].y, get, vals);
                OutT *mine = w.tile + (size_t)q * C;
#pragma unroll
                for (int c = 0; c < CMAX; ++c) if (c < C) mine[c] = vals[c];
            }
            sb = se;
            wave_phase();
            if (part == 0) { w.mark(3); w.pace(); }
            tile_store(w.tile, np * C, dst + (size_t)part * PP * C);
            if (part == 0) w.mark(4);
        }
        w.mark(5);
    }
}

// Hot launch (r04): ONE 64-pixel part of a unit of more records than a main wave's stage -- a hot unit of a clustered window
// (a moving edge along the sensor rows leaves 25 records per pixel where the window's average is 0.2).  unit_records has
// laid the part's records out, pixel-sorted and time-ordered inside a pixel, in the unit's slot of the spill stream
// (`stream`); lane l owns pixel part * 64 + l, records [st, en).  Until r03 such a unit walked its segments straight from
// global memory inside the frame's own launch -- one dependent L2 round trip (and, for the builders that divide, one float64
// division) per step of the longest segment.  Now
//   * a part whose records fit the hot stage (w.big_at: the part tile + the record stage) has them STAGED -- coalesced
//     16-byte loads, eight in flight -- and digested one record per lane; the walks read LDS; only then is the tile filled,
//     patched and streamed out;
//   * a hotter part is walked from the stream through a four-deep register ring per lane -- the load of record j + 4 is
//     issued when record j is consumed, so a step waits for arithmetic, not for L2 (the walks are sequential per pixel by
//     contract: what bounds such a wave is its longest segment).
template <typename OutT, int CMAX, bool STAGE, typename Digest, typename DigestFly, typename Reduce>
__device__ inline void emit_part(const Rec *__restrict__ stream, uint32_t nrec, uint32_t st, uint32_t en, int part, bool staged_raw,
                                 Digest digest, DigestFly digest_fly, int npix, int C, OutT *__restrict__ dst,
                                 WaveLds<OutT, true> &w, const OutT *bg, Reduce reduce) {
    constexpr bool HOT = true;
    const int lane = threadIdx.x;
    const uint32_t cap = (uint32_t)w.bigcap;
    const int px0 = piece_px0(part);   // `part` = the item's piece code (a 64-pixel part or a 16-pixel quarter of one)
    const int np = min(piece_npx(part), npix - px0);
    if (np <= 0) return;   // (a quarter beyond the row's end)
    const bool mine = en > st && lane < np;
    const uint32_t ra = (uint32_t)wave_min(mine ? (int)st : INT32_MAX), rb = (uint32_t)wave_max(mine ? (int)en : 0);
    auto ld = [&](uint32_t j) -> Rec {
        const uint4 v = gload16(stream + min(j, nrec - 1u));
        return make_int4((int)v.x, (int)v.y, (int)v.z, (int)v.w);
    };
    OutT vals[CMAX];
    if (__any(mine)) {
        if (staged_raw) {   // wave-uniform: unit_records has placed the part in the stage: digested in place, walked from LDS
            for (uint32_t j = (uint32_t)lane; j < rb - ra; j += kWave) { Rec *q = w.big_at(j); *q = digest(*q); }
            wave_phase();
            w.mark(4);
            if (mine) reduce(st - ra, en - ra, [&](uint32_t j) -> Rec { return *w.big_at(j); }, vals);
        } else if (STAGE && rb - ra <= cap) {   // wave-uniform (STAGE false: a builder that reads one record per segment)
            constexpr int kDepth = HOT ? 8 : 4;   // 16-byte loads in flight per lane
            for (uint32_t j0 = ra; j0 < rb; j0 += kDepth * kWave) {
                Rec r[kDepth];
#pragma unroll
                for (int i = 0; i < kDepth; ++i) {
                    const uint32_t j = j0 + (uint32_t)(i * kWave + lane);
                    r[i] = make_int4(0, 0, 0, 0);
                    if (j < rb) r[i] = ld(j);
                }
#pragma unroll
                for (int i = 0; i < kDepth; ++i) {
                    const uint32_t j = j0 + (uint32_t)(i * kWave + lane);
                    if (j < rb) *w.big_at(j - ra) = HOT ? digest(r[i]) : r[i];
                }
            }
            wave_phase();
            if constexpr (!HOT) {   // digested in its own loop: the load batch and the digest's temporaries would add up in a main launch's register budget
                for (uint32_t j = (uint32_t)lane; j < rb - ra; j += kWave) { Rec *q = w.big_at(j); *q = digest(*q); }
                wave_phase();
            }
            w.mark(4);   // timing builds: staged
            if (mine) reduce(st - ra, en - ra, [&](uint32_t j) -> Rec { return *w.big_at(j); }, vals);
        } else if (!HOT) {
            // STAGE false (a main launch only sorts units whose parts fit the stage): one record per segment, from the slot
            if (mine) reduce(st, en, [&](uint32_t j) -> Rec { return digest_fly(stream[j]); }, vals);
        } else if (mine) {
            Rec q0 = make_int4(0, 0, 0, 0), q1 = q0, q2 = q0, q3 = q0;
            uint32_t expect = 0xffffffffu;
            reduce(st, en, [&](uint32_t j) -> Rec {
                if (j != expect) { q0 = ld(j); q1 = ld(j + 1); q2 = ld(j + 2); q3 = ld(j + 3); }   // a walk starts (or restarts: voxel's second pass)
                const Rec e = q0;
                q0 = q1; q1 = q2; q2 = q3;
                q3 = ld(j + 4);
                expect = j + 1;
                return digest_fly(e);
            }, vals);
        }
    }
    wave_phase();   // the walks are done: the stage may become the tile
    tile_fill(w.tile, np, C, bg);
    wave_phase();
    if (mine) {
        OutT *t = w.tile + (size_t)lane * C;
#pragma unroll
        for (int c = 0; c < CMAX; ++c) if (c < C) t[c] = vals[c];
    }
    wave_phase();
    if (part == 0) w.pace();
    tile_store(w.tile, np * C, dst + (size_t)px0 * C);
    w.mark(5);
}

// One pixel's C values straight from a lane's registers: 16 bytes at a time when the pixel is a whole number of 16-byte
// vectors (a wave-instruction then writes 16 bytes of each of 64 consecutive pixels; the C * sizeof / 16 instructions of a
// part complete every line), element by element otherwise.
template <typename OutT, int CMAX>
__device__ inline void store_pixel(OutT *__restrict__ o, const OutT (&vals)[CMAX], int C, bool vec) {
    constexpr int V = 16 / (int)sizeof(OutT);
    if (vec) {
#pragma unroll
        for (int v = 0; v < CMAX / V; ++v) {
            if (v * V < C) {
                uint4 pk;
                if constexpr (sizeof(OutT) == 8) {
                    const double a = (double)vals[2 * v], b = (double)vals[2 * v + 1];
                    pk = make_uint4((uint32_t)__double2loint(a), (uint32_t)__double2hiint(a), (uint32_t)__double2loint(b), (uint32_t)__double2hiint(b));
                } else {
                    pk = make_uint4(__float_as_uint((float)vals[4 * v]), __float_as_uint((float)vals[4 * v + 1]),
                                    __float_as_uint((float)vals[4 * v + 2]), __float_as_uint((float)vals[4 * v + 3]));
                }
                gstore16(o + v * V, pk);
            }
        }
    } else {
#pragma unroll
        for (int c = 0; c < CMAX; ++c) if (c < C) o[c] = vals[c];
    }
}

// Main launch, key-sorted pass, a WARM unit (r04): more records than the record stage, few enough for the hot stage (an edge
// crossing the unit leaves it 100-400 records where the window's average is 20).  unit_records has placed ALL its records,
// pixel-sorted and time-ordered inside a pixel, in the hot stage -- which overlays the part tile; w.segs holds every pixel's
// END.  They are digested one record per lane, then each 64-pixel part is reduced with one lane per PIXEL and leaves the wave
// straight from the lanes' registers: lane l stores the C values of pixel l (the background for an empty pixel), 16 bytes at a
// time -- a wave-instruction writes 16 bytes of each of 64 consecutive pixels, the C * sizeof / 16 instructions of a part
// complete every line.  No tile, so no second home for the records, no spill slot and no trip through it: three dependent
// memory round trips (run tables, records, -- ) where the slot path takes eight, which is what such a wave is made of when it
// runs beside the frame's store-bound waves.
template <typename OutT, int CMAX, typename Digest, typename Reduce>
__device__ inline void emit_warm(const UnitRecs &u, uint32_t nrec, Digest digest, int npix, int C, OutT *__restrict__ dst,
                                 WaveLds<OutT, false> &w, const OutT *bg, Reduce reduce) {
    const int lane = threadIdx.x;
    constexpr int V = 16 / (int)sizeof(OutT);
    for (uint32_t j = (uint32_t)lane; j < nrec; j += kWave) { Rec *q = w.big_at(j); *q = digest(*q); }
    wave_phase();
    w.mark(2);
    const uint32_t *cnt = reinterpret_cast<const uint32_t *>(w.segs);
    const bool vec = (C % V) == 0 && (reinterpret_cast<uintptr_t>(dst) & 15u) == 0;   // wave-uniform
    w.pace();
    for (int p = 0; p * kWave < npix; ++p) {
        const int np = min(kWave, npix - p * kWave);
        const int px = p * kWave + u.dpx + lane;
        uint32_t st = 0, en = 0;
        if (lane < np && px < u.npixu) { en = cnt[px]; st = px ? cnt[px - 1] : 0u; }
        OutT vals[CMAX];
#pragma unroll
        for (int c = 0; c < CMAX; ++c) vals[c] = (bg && c < C) ? bg[c] : (OutT)0;
        if (en > st) reduce(st, en, [&](uint32_t j) -> Rec { return *w.big_at(j); }, vals);
        if (lane < np) store_pixel<OutT, CMAX>(dst + ((size_t)p * kWave + lane) * C, vals, C, vec);
    }
    w.mark(5);
}

// Main launch, key-sorted pass, one 64-pixel part of a unit beyond the hot stage (r04), out of the unit's spill slot
// (`stream`; lane l owns pixel part * 64 + l, records [st, en)).  The part is worked off in PIECES: as many consecutive pixels
// as hold no more records than the hot stage; a piece's records are staged -- coalesced 16-byte loads, four in flight --
// digested one record per lane, reduced from LDS one lane per pixel, and the lanes store their pixels straight from their
// registers (store_pixel), so the stage never has to make room for a tile and nothing outlives a piece.  A single PIXEL of
// more records than the stage is walked from the slot.  (A unit with a part of more than EVREP_DEFER_MULT stages goes to the
// hot launch instead: unit_records.)
template <typename OutT, int CMAX, bool STAGE, typename Digest, typename DigestFly, typename Reduce>
__device__ inline void emit_part_main(const Rec *__restrict__ stream, uint32_t st, uint32_t en, int part, Digest digest,
                                      DigestFly digest_fly, int npix, int C, OutT *__restrict__ dst, WaveLds<OutT, false> &w,
                                      const OutT *bg, Reduce reduce) {
    const int lane = threadIdx.x;
    constexpr int V = 16 / (int)sizeof(OutT);
    const uint32_t cap = (uint32_t)w.bigcap;
    const int np = min(kWave, npix - part * kWave);
    const bool mine = en > st && lane < np;
    const bool vec = (C % V) == 0 && (reinterpret_cast<uintptr_t>(dst) & 15u) == 0;   // wave-uniform
    const uint64_t mm = __ballot(mine);
    OutT *o = dst + ((size_t)part * kWave + lane) * C;
    for (int ls = 0; ls < np;) {   // pieces (wave-uniform)
        const uint64_t m = mm & ~((1ull << ls) - 1ull);
        int le = np;
        bool staged = false;
        uint32_t ra = 0;
        if (m) {
            const int f = __builtin_ctzll(m);
            ra = (uint32_t)__builtin_amdgcn_readlane((int)st, f);
            if (STAGE) {
                const uint64_t bad = __ballot(mine && lane >= f && en - ra > cap);
                if (bad) le = __builtin_ctzll(bad);
                if (le == f) le = f + 1;   // ONE pixel of more records than the stage: walked from the slot
                else {
                    const uint64_t in = m & ((le < 64 ? (1ull << le) : 0ull) - 1ull);
                    const uint32_t rb = (uint32_t)__builtin_amdgcn_readlane((int)en, 63 - __builtin_clzll(in));
                    wave_phase();   // the previous piece's walks have read the stage
                    for (uint32_t j0 = ra; j0 < rb; j0 += 4 * kWave) {
                        uint4 r[4];
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const uint32_t j = j0 + (uint32_t)(i * kWave + lane);
                            r[i] = make_uint4(0u, 0u, 0u, 0u);
                            if (j < rb) r[i] = gload16(stream + j);
                        }
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const uint32_t j = j0 + (uint32_t)(i * kWave + lane);
                            if (j < rb) *w.big_at(j - ra) = make_int4((int)r[i].x, (int)r[i].y, (int)r[i].z, (int)r[i].w);
                        }
                    }
                    wave_phase();
                    for (uint32_t j = (uint32_t)lane; j < rb - ra; j += kWave) { Rec *q = w.big_at(j); *q = digest(*q); }
                    wave_phase();
                    staged = true;
                }
            }
        }
        if (lane >= ls && lane < le) {
            OutT vals[CMAX];
#pragma unroll
            for (int c = 0; c < CMAX; ++c) vals[c] = (bg && c < C) ? bg[c] : (OutT)0;
            if (mine) {
                if (staged) reduce(st - ra, en - ra, [&](uint32_t j) -> Rec { return *w.big_at(j); }, vals);
                else reduce(st, en, [&](uint32_t j) -> Rec { return digest_fly(stream[j]); }, vals);
            }
            store_pixel<OutT, CMAX>(o, vals, C, vec);
        }
        ls = le;
    }
}

// emit_core over a unit's pixel-sorted records.  u.r0 = record `lane`; records [64, u.nstaged) sit in the wave's LDS stage
// (the key-sorted front end put them there), later ones are read from u.sorted[u.cs + j].
// `digest(rec)` = the form in which the builder's reduce wants a record (identity, or with the per-event float64
// division done: MDES' normalised timestamp, the voxel bin position).  reduce() only ever sees digests:
//   * the staged records are digested here, one record per lane, all lanes at once -- not inside the divergent segment
//     walks, where every step of the longest segment costs the whole wave a division;
//   * whatever is not staged is fetched and digested on the fly.
// (Re-staging the record range of every part tile of a dense chunk the same way was measured in round 2: the extra LDS
// round trips cost more than the divergent loads and divisions they replace, for every builder but the voxel grid.)
// `digest_fly(rec)` = what get(j) returns for a record that is NOT staged (read inside a walk): the same digest by
// default; a builder whose digest is expensive and not needed by every reader of a record passes something cheaper and
// tells the two forms apart by the index (staged: j < max(64, u.nstaged)).
template <typename OutT, int CMAX, bool HOT, bool STAGE = true, typename Digest, typename DigestFly, typename Reduce>
__device__ inline void emit_chunk(const UnitRecs &u, Digest digest, DigestFly digest_fly, int key0, int npix, int C,
                                  OutT *__restrict__ dst, WaveLds<OutT, HOT> &w, const OutT *bg, Reduce reduce) {
    const int lane = threadIdx.x;
    const uint32_t nrec = u.ce - u.cs, cs = u.cs, nraw = (uint32_t)u.nstaged;
    const Rec r0 = u.r0;
    const Rec *__restrict__ sorted = u.sorted;
    Rec *evbuf = w.evbuf;
    const uint32_t nst = max((uint32_t)kWave, nraw);  // records [0, nst) are staged
    if constexpr (HOT) {   // one part of a unit beyond a main wave's stage, out of the unit's spill slot
        emit_part<OutT, CMAX, STAGE>(sorted + cs, nrec, u.pst, u.pen, u.part, u.hot_lds, digest, digest_fly, npix, C, dst, w, bg, reduce);
        return;
    } else if (u.part == -3) {   // wave-uniform: a main launch, a WARM unit sorted into the hot stage (over the tile)
        emit_warm<OutT, CMAX>(u, nrec, digest, npix, C, dst, w, bg, reduce);
        return;
    } else if (u.part == -2) {   // wave-uniform: a main launch, a unit it has sorted into its spill slot: part by part
        const uint32_t *cnt = reinterpret_cast<const uint32_t *>(w.segs);   // every pixel's END in the slot
        for (int p = 0; p * kWave < npix; ++p) {
            const int px = p * kWave + u.dpx + lane;
            uint32_t st = 0, en = 0;
            if (lane < min(kWave, npix - p * kWave) && px < u.npixu) { en = cnt[px]; st = px ? cnt[px - 1] : 0u; }
            if (p == 0) w.pace();
            emit_part_main<OutT, CMAX, STAGE>(sorted + cs, st, en, p, digest, digest_fly, npix, C, dst, w, bg, reduce);
        }
        w.mark(5);
        return;
    }
    if (lane < (int)nrec) evbuf[u.nseg >= 0 ? u.pos : lane] = digest(r0);   // grouped units: straight to the record's place
    auto key_at = [&](uint32_t j) -> int { return j < (uint32_t)kWave ? r0.x : (j < nraw ? evbuf[j].x : sorted[cs + j].x); };
    auto get = [&](uint32_t j) -> Rec { return j < nst ? evbuf[j] : digest_fly(sorted[cs + j]); };
    auto get_staged = [&](uint32_t j) -> Rec { return evbuf[j]; };
    auto post_heads = [&]() {
        if (nraw > (uint32_t)kWave) {  // uniform: the later staged batches are still raw (key_at read their pixel ids)
            for (int j = lane + kWave; j < (int)nraw; j += kWave) evbuf[j] = digest(evbuf[j]);
            wave_phase();
        }
    };
    // (main launches of the classic passes: records beyond the stage are read inside the walks, as in r03)
    emit_core<OutT, CMAX, HOT>(nrec, u.nseg, nrec <= nst, key_at, get, get_staged, post_heads, key0, npix, C, dst, w, bg, reduce);
}
template <typename OutT, int CMAX, bool HOT, bool STAGE = true, typename Digest, typename Reduce>
__device__ inline void emit_chunk(const UnitRecs &u, Digest digest, int key0, int npix, int C, OutT *__restrict__ dst,
                                  WaveLds<OutT, HOT> &w, const OutT *bg, Reduce reduce) {
    emit_chunk<OutT, CMAX, HOT, STAGE>(u, digest, digest, key0, npix, C, dst, w, bg, reduce);
}
// no digest; STAGE false: the builder reads ONE record per segment (EventStack) -- a unit beyond the stage is walked from
// the stream instead of being copied to LDS first
template <typename OutT, int CMAX, bool HOT, bool STAGE = true, typename Reduce>
__device__ inline void emit_chunk(const UnitRecs &u, int key0, int npix, int C, OutT *__restrict__ dst, WaveLds<OutT, HOT> &w,
                                  const OutT *bg, Reduce reduce) {
    emit_chunk<OutT, CMAX, HOT, STAGE>(u, [](const Rec &r) -> Rec { return r; }, key0, npix, C, dst, w, bg, reduce);
}

// --------------------------------------------------------------------------------------------
// r06: the STREAM front end shared by the order-free builders after the key-sorted pass.
// Every record of the sensor keys [klo, khi) of window b (consecutive 128-pixel chunks of one row), 64 at a time, in array
// order: `pre(q)` (optional per-record gather from global memory, issued for every batch of a group before the first `f`) and
// `f(have, q, aux)`.  Units of up to 64 * RB records: every record load in flight at once, a record's run found by a max-scan over
// the runs' first positions; larger ones four batches at a time, run by run when the runs fill batches.  No order by pixel, no
// segment list: what a builder keeps per pixel it keeps by LDS atomics (or elections) on its own tile.
// head: [64 * RB] words of LDS, srcs: [128].  Returns the unit's record count.
struct StreamNoPre { __device__ inline uint2 operator()(const Rec8 &) const { return make_uint2(0u, 0u); } };
// The run tables of a unit's keys, up to 128 block runs: lane l holds run l and, in windows of more than 64 runs, run 64 + l.
struct StreamRuns {
    uint32_t a0, len0, pre0, a1, len1, pre1, nrec;
    int nb;
    // of run k (wave-uniform k): its length / the address of its first record of the unit, given the lane-wise values v0 (runs 0..63), v1 (64..127)
    __device__ inline uint32_t pick(uint32_t v0, uint32_t v1, int k) const {
        return k < kWave ? (uint32_t)__builtin_amdgcn_readlane((int)v0, k) : (uint32_t)__builtin_amdgcn_readlane((int)v1, k - kWave);
    }
};
__device__ inline StreamRuns stream_runs(const BinView &bv, int b, int64_t n_win, int NK, int klo, int khi) {
    const int lane = threadIdx.x & (kWave - 1);   // (lane-relative: k_voxel_hot's workgroups hold several waves, each reads the tables itself)
    StreamRuns r;
    r.a0 = 0; r.a1 = 0; r.len0 = 0; r.len1 = 0; r.pre0 = 0; r.pre1 = 0; r.nrec = 0;
    uint32_t k0v = 0, k1v = 0;
    if (lane < bv.nblk) {
        const uint32_t *tb = bv.table + ((size_t)b * bv.nblk + lane) * ((size_t)NK + 1);
        r.a0 = tb[klo];
        k0v = tb[khi];
    }
    if (kWave + lane < bv.nblk) {   // (only windows of more than 64 runs)
        const uint32_t *tb = bv.table + ((size_t)b * bv.nblk + kWave + lane) * ((size_t)NK + 1);
        r.a1 = tb[klo];
        k1v = tb[khi];
    }
    r.nb = min((int)(((uint32_t)n_win + (1u << bv.chunk_shift) - 1u) >> bv.chunk_shift), bv.nblk);
    if (r.nb <= 0) return r;
    r.len0 = k0v - r.a0; r.len1 = k1v - r.a1;
    if (lane >= r.nb) { r.a0 = 0; r.len0 = 0; }
    if (kWave + lane >= r.nb) { r.a1 = 0; r.len1 = 0; }
    const uint32_t i0 = wave_incl_scan(r.len0);
    r.pre0 = i0 - r.len0;
    r.nrec = (uint32_t)__builtin_amdgcn_readlane((int)i0, 63);
    if (r.nb > kWave) {   // wave-uniform
        const uint32_t i1 = wave_incl_scan(r.len1);
        r.pre1 = r.nrec + i1 - r.len1;
        r.nrec += (uint32_t)__builtin_amdgcn_readlane((int)i1, 63);
    }
    return r;
}
template <int RB, typename Pre, typename F>
__device__ inline uint32_t stream_unit_records(const BinView &bv, int b, int64_t beg, int64_t n_win, int NK, int klo, int khi,
                                               uint32_t *head, uint32_t *srcs, Pre pre_f, F f, uint32_t byrun_min = EVREP_STREAM_BYRUN,
                                               bool reverse = false, int *is_big = nullptr) {
    // reverse (wave-uniform; order-free builders only): the batches from the unit's LAST records to its first -- a pixel's records are
    // time-ordered inside a run and the runs are time slices, so every pixel's records then arrive latest first.  is_big: set
    // before the first batch of a unit of more than 64 * RB records (the caller's batch function may treat those differently)
    static_assert(RB >= 4, "head[] doubles as the 2 x 128-word run table of the LDS search");
    const int lane = threadIdx.x;
    const StreamRuns R = stream_runs(bv, b, n_win, NK, klo, khi);
    const int nb = R.nb;
    const uint32_t nrec = R.nrec;
    if (nb <= 0 || nrec == 0u) return 0u;
    const bool wide = nb > kWave;   // wave-uniform
    const Rec8 *__restrict__ s8 = reinterpret_cast<const Rec8 *>(bv.sorted);
    // record j of the unit, if in run `lane` / run 64 + lane: src + j
    const uint32_t src0 = (uint32_t)beg + ((uint32_t)lane << bv.chunk_shift) + R.a0 - R.pre0;
    const uint32_t src1 = (uint32_t)beg + ((uint32_t)(kWave + lane) << bv.chunk_shift) + R.a1 - R.pre1;
    if (nrec <= (uint32_t)(64 * RB)) {
#pragma unroll
        for (int i = 0; i < RB; ++i) head[lane + 64 * i] = 0u;
        srcs[lane] = src0;
        if (wide) srcs[kWave + lane] = src1;
        wave_phase();
        if (R.len0 > 0u && R.pre0 < (uint32_t)(64 * RB)) head[R.pre0] = (uint32_t)lane;
        if (wide && R.len1 > 0u && R.pre1 < (uint32_t)(64 * RB)) head[R.pre1] = (uint32_t)(kWave + lane);
        wave_phase();
        Rec8 q[RB];
        uint32_t carry = 0u;
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            q[i] = make_uint2(0u, 0u);
            if ((uint32_t)(64 * i) < nrec) {   // uniform
                const uint32_t k = max(carry, wave_incl_max_scan(head[lane + 64 * i]));
                carry = (uint32_t)__builtin_amdgcn_readlane((int)k, 63);
                const uint32_t j = (uint32_t)(64 * i + lane);
                if (j < nrec) q[i] = s8[srcs[k] + j];
            }
        }
        uint2 aux[RB];
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            aux[i] = make_uint2(0u, 0u);
            if ((uint32_t)(64 * i + lane) < nrec) aux[i] = pre_f(q[i]);
        }
        if (reverse) {
#pragma unroll
            for (int i = RB - 1; i >= 0; --i)
                if ((uint32_t)(64 * i) < nrec) f((uint32_t)(64 * i + lane) < nrec, q[i], aux[i]);
            return nrec;
        }
#pragma unroll
        for (int i = 0; i < RB; ++i)
            if ((uint32_t)(64 * i) < nrec) f((uint32_t)(64 * i + lane) < nrec, q[i], aux[i]);
        return nrec;
    }

    constexpr int G = EVREP_STREAM_G;
    const bool by_run = nrec >= byrun_min * (uint32_t)nb;   // wave-uniform
    const uint32_t run00 = (uint32_t)beg + ((uint32_t)lane << bv.chunk_shift) + R.a0;
    const uint32_t run01 = (uint32_t)beg + ((uint32_t)(kWave + lane) << bv.chunk_shift) + R.a1;
    uint32_t *rt = head;   // [2][128]: the runs' first records of the unit (pre) and their addresses (src)
    if (!by_run && nb > kBsChainBlocks) {
        rt[lane] = R.pre0; rt[128 + lane] = src0;
        if (wide) { rt[kWave + lane] = R.pre1; rt[128 + kWave + lane] = src1; }
    }
    if (is_big) *is_big = (!by_run && nb > kBsChainBlocks) ? 2 : 1;   // 1: `head` is free during the sweep (the caller may use it), 2: it holds the run table
    wave_phase();
    auto src_of = [&](uint32_t j) -> uint32_t {
        if (nb <= kBsChainBlocks) {
            uint32_t sx = (uint32_t)__builtin_amdgcn_readlane((int)src0, 0);
            uint32_t prev = sx;
            for (int k = 1; k < nb; ++k) {
                const uint32_t pk = (uint32_t)__builtin_amdgcn_readlane((int)R.pre0, k);
                const uint32_t sk = (uint32_t)__builtin_amdgcn_readlane((int)src0, k);
                sx += (j >= pk) ? sk - prev : 0u;
                prev = sk;
            }
            return sx + j;
        }
        uint32_t lo = 0, hi = (uint32_t)nb;
#pragma unroll
        for (int step = 0; step < 7; ++step) {
            const uint32_t mid = (lo + hi) >> 1;
            const bool go = hi - lo > 1 && rt[mid] <= j;
            if (hi - lo > 1) { if (go) lo = mid; else hi = mid; }
        }
        return rt[128 + lo] + j;
    };
    int rk_ = 0;
    uint32_t ro_ = 0;
    bool more = true;
    uint32_t run_len = 0u, run_base = 0u;   // of run rk_ (in visiting order)
    auto load_run = [&]() {
        run_len = 0u;
        if (rk_ < nb) {
            const int r = reverse ? nb - 1 - rk_ : rk_;
            run_len = R.pick(R.len0, R.len1, r);
            run_base = R.pick(run00, run01, r);
        }
    };
    if (by_run) load_run();
#ifdef EVREP_TIMING   // (experiment builds: where a big unit's sweep spends its time -- issue / wait for the records / process)
    long long ta_ = 0, acc_i = 0, acc_w = 0, acc_p = 0, nbat = 0;
#endif
    while (more) {
#ifdef EVREP_TIMING
        ta_ = (long long)wall_clock64();
#endif
        Rec8 q[G];
        uint32_t bcnt[G];
#pragma unroll
        for (int sl = 0; sl < G; ++sl) {
            bcnt[sl] = 0u;
            uint32_t addr = 0u;
            // (reverse: the cursor counts the runs from the last one and the records from a run's / the unit's END)
            if (by_run) {
                while (rk_ < nb && ro_ >= run_len) { ++rk_; ro_ = 0; load_run(); }   // (the current run's length and base are kept in scalars)
                if (rk_ < nb) {
                    bcnt[sl] = min(run_len - ro_, (uint32_t)kWave);
                    addr = run_base + (reverse ? run_len - ro_ - bcnt[sl] : ro_) + (uint32_t)lane;
                    ro_ += kWave;
                }
            } else {
                if (ro_ < nrec) {
                    bcnt[sl] = min(nrec - ro_, (uint32_t)kWave);
                    const uint32_t j0 = reverse ? nrec - ro_ - bcnt[sl] : ro_;
                    if ((uint32_t)lane < bcnt[sl]) addr = src_of(j0 + (uint32_t)lane);
                    ro_ += kWave;
                }
            }
            q[sl] = make_uint2(0u, 0u);
            if ((uint32_t)lane < bcnt[sl]) q[sl] = s8[addr];
        }
        if (by_run) {
            while (rk_ < nb && ro_ >= run_len) { ++rk_; ro_ = 0; load_run(); }
            more = rk_ < nb;
        } else {
            more = ro_ < nrec;
        }
#ifdef EVREP_TIMING
        const long long tb_ = (long long)wall_clock64();
        __builtin_amdgcn_s_waitcnt(0);
        const long long tc_ = (long long)wall_clock64();
#endif
        uint2 aux[G];
#pragma unroll
        for (int sl = 0; sl < G; ++sl) {
            aux[sl] = make_uint2(0u, 0u);
            if ((uint32_t)lane < bcnt[sl]) aux[sl] = pre_f(q[sl]);
        }
#pragma unroll
        for (int sl = 0; sl < G; ++sl) {
            if (bcnt[sl] == 0u) break;   // uniform
            f((uint32_t)lane < bcnt[sl], q[sl], aux[sl]);
#ifdef EVREP_TIMING
            ++nbat;
#endif
        }
#ifdef EVREP_TIMING
        { const long long td_ = (long long)wall_clock64(); acc_i += tb_ - ta_; acc_w += tc_ - tb_; acc_p += td_ - tc_; }
#endif
    }
#ifdef EVREP_TIMING
    if (bv.dbg_wave && lane == 0) { bv.dbg_wave[0] = (unsigned long long)acc_i; bv.dbg_wave[1] = (unsigned long long)acc_w; bv.dbg_wave[2] = (unsigned long long)acc_p; bv.dbg_wave[3] = (unsigned long long)nbat; }
#endif
    return nrec;
}

// --------------------------------------------------------------------------------------------
// A3/A4/A5: MixedDensityEventStack.stack + Operations
// (representation_search/mixed_density_event_stack.py:25-151, operations.py:15-89)
// --------------------------------------------------------------------------------------------
struct MdesParams {
    int32_t C;
    int32_t win[EVREP_MAX_CHANNELS], func[EVREP_MAX_CHANNELS], agg[EVREP_MAX_CHANNELS];
    // "SBT" stacking (evrep_mdes_ex): the caller's windows as rank ranges, [B][8][2] {lo, hi}, and their polarity /
    // out-of-frame flags, [B][2] {neg_flags, oob_flags with a stride of 8 bits per polarity class}; nullptr = "SBN"
    const int32_t *bounds;
    const uint32_t *wflags;
};

// Descriptor sources: RuntimeDesc reads the caller's triples from the kernel arguments;
// StaticDesc<T> reads a constexpr table, so after unrolling every per-channel branch folds away
// and unused accumulators disappear (ERGO-12: 2 variance, 3 max, 1 mean-of-timestamps, ...).
template <int N>
struct RuntimeDesc {  // N = compile-time capacity (4, 8, 12, 16): register arrays are sized by it
    static constexpr int kMaxC = N;
    static constexpr bool kCustomWindows = true;   // may run over the caller's windows ("SBT", evrep_mdes_ex)
    __device__ static inline int C(const MdesParams &P) { return P.C; }
    __device__ static inline int win(const MdesParams &P, int c) { return P.win[c]; }
    __device__ static inline int func(const MdesParams &P, int c) { return P.func[c]; }
    __device__ static inline int agg(const MdesParams &P, int c) { return P.agg[c]; }
};

// the ERGO-12 triples, optimized_representation.py:87-115
struct Ergo12Table {
    static constexpr int kC = 12;
    static constexpr int kWin[12] = {0, 3, 2, 6, 5, 6, 2, 5, 1, 0, 4, 1};
    static constexpr int kFunc[12] = {EVREP_F_POLARITY, EVREP_F_TIMESTAMP_NEG, EVREP_F_COUNT_NEG, EVREP_F_POLARITY,
                                      EVREP_F_COUNT_POS, EVREP_F_COUNT, EVREP_F_TIMESTAMP_POS, EVREP_F_COUNT_NEG,
                                      EVREP_F_TIMESTAMP_NEG, EVREP_F_TIMESTAMP_POS, EVREP_F_TIMESTAMP, EVREP_F_COUNT};
    static constexpr int kAgg[12] = {EVREP_A_VARIANCE, EVREP_A_VARIANCE, EVREP_A_MEAN, EVREP_A_SUM, EVREP_A_MEAN,
                                     EVREP_A_SUM, EVREP_A_MEAN, EVREP_A_MEAN, EVREP_A_MAX, EVREP_A_MAX, EVREP_A_MAX,
                                     EVREP_A_MEAN};
};

template <typename T>
struct StaticDesc {
    static constexpr int kMaxC = T::kC;
    static constexpr bool kCustomWindows = false;
    __device__ static inline int C(const MdesParams &) { return T::kC; }
    __device__ static inline int win(const MdesParams &, int c) { return T::kWin[c]; }
    __device__ static inline int func(const MdesParams &, int c) { return T::kFunc[c]; }
    __device__ static inline int agg(const MdesParams &, int c) { return T::kAgg[c]; }
};

constexpr int kWantAny = 2;

__device__ inline bool is_count_func(int f) { return f == EVREP_F_COUNT || f == EVREP_F_COUNT_POS || f == EVREP_F_COUNT_NEG; }

// grid (nchunk, H, B), 64 threads; dynamic LDS = chunk_lds_bytes(C, sizeof(OutT)).
// One unit of MixedDensityEventStack: the window's statistics -> per-channel set-up -> digest / reduce -> emit.
//
// The ERGO-12 split (r05, unit_records' Split): ten of the twelve channels need no time order --
//   ch0  (w0, polarity, variance)       n, sum p, sum p^2: integers, exact in float64 in any order (p in {-1, 0, 1})
//   ch3  (w6, polarity, sum), ch5 (w6, count, sum)                                     integer sums
//   ch2, ch4, ch7, ch11 (count_* , mean)                                               1 iff the pixel holds such an event
//   ch8, ch9, ch10 (timestamp_*, max)   max of the quotients = the quotient of the max (a correctly rounded division is monotone)
// -- and are kept as seven 32-bit words per pixel, bumped by LDS atomics in ONE unordered sweep; only ch1 (w3, timestamp_neg,
// variance) and ch6 (w2, timestamp_pos, mean) are float64 sums that have to run in time order, each over a sixth of the records
// (one third of the ranks, one polarity class): those records are kept, ordered by pixel out of LDS and walked.  w2's ranks lie
// in front of w3's, so a pixel's kept records are its ch6 records, then its ch1 records: the pixel's ch6 count is all the walk
// needs to tell them apart.  Words of pixel px (7 px + k):
//   0: ch0 #(p > 0) | #(p < 0) << 16      1: ch0 #(p == 0) | ch5 n << 16      2: ch3 #(p > 0) | #(p < 0) << 16
//   3: bit c: channel c present (ch2, ch4, ch7, ch11: occupancy; ch8, ch9, ch10: a maximum exists) | kept ch6 records << 16
//   4, 5, 6: max (t - tmin) of ch8, ch9, ch10
// A record of escaped polarity (p outside {-1, 0, 1}: sum p^2 may leave the integers float64 holds exactly) sends its unit to the
// ordered paths.
constexpr int kErgoSplitWords = 7;
// (measured, r06, library variants alternated -- tools/experiments/lib_ab.sh; float32 ERGO-12 build in us at 4096 / 2048 / 1536 / 1024:
//  1 Mpx circle 193 / 168 / 162 / 162, 640x480 circle 137 / 136 / 148 / 147, 1 Mpx edges 108 / 109 / 111 / 118; every other row within 1 %)
#ifndef EVREP_ERGO_COOP_MIN
#define EVREP_ERGO_COOP_MIN 2048
#endif
constexpr uint32_t kErgoCoopMin = EVREP_ERGO_COOP_MIN;   // records from which the ordered float32 builder's hot units go to k_mdes_coop
// the ERGO-12 channels of rank window `wnd`, as bits
constexpr uint32_t ergo_chans_of(int wnd) {
    uint32_t m = 0;
    for (int c = 0; c < 12; ++c) if (Ergo12Table::kWin[c] == wnd) m |= 1u << c;
    return m;
}
static_assert(ergo_chans_of(0) == ((1u << 0) | (1u << 9)) && ergo_chans_of(6) == ((1u << 3) | (1u << 5)), "ergo_chans_of");
template <typename D> struct MdesIsErgo12 { static constexpr bool value = false; };
template <> struct MdesIsErgo12<StaticDesc<Ergo12Table>> { static constexpr bool value = true; };
static_assert(Ergo12Table::kWin[0] == 0 && Ergo12Table::kFunc[0] == EVREP_F_POLARITY && Ergo12Table::kAgg[0] == EVREP_A_VARIANCE &&
              Ergo12Table::kWin[1] == 3 && Ergo12Table::kFunc[1] == EVREP_F_TIMESTAMP_NEG && Ergo12Table::kAgg[1] == EVREP_A_VARIANCE &&
              Ergo12Table::kFunc[2] == EVREP_F_COUNT_NEG && Ergo12Table::kAgg[2] == EVREP_A_MEAN &&
              Ergo12Table::kFunc[3] == EVREP_F_POLARITY && Ergo12Table::kAgg[3] == EVREP_A_SUM &&
              Ergo12Table::kFunc[4] == EVREP_F_COUNT_POS && Ergo12Table::kAgg[4] == EVREP_A_MEAN &&
              Ergo12Table::kFunc[5] == EVREP_F_COUNT && Ergo12Table::kAgg[5] == EVREP_A_SUM &&
              Ergo12Table::kWin[6] == 2 && Ergo12Table::kFunc[6] == EVREP_F_TIMESTAMP_POS && Ergo12Table::kAgg[6] == EVREP_A_MEAN &&
              Ergo12Table::kFunc[7] == EVREP_F_COUNT_NEG && Ergo12Table::kAgg[7] == EVREP_A_MEAN &&
              Ergo12Table::kFunc[8] == EVREP_F_TIMESTAMP_NEG && Ergo12Table::kAgg[8] == EVREP_A_MAX &&
              Ergo12Table::kFunc[9] == EVREP_F_TIMESTAMP_POS && Ergo12Table::kAgg[9] == EVREP_A_MAX &&
              Ergo12Table::kFunc[10] == EVREP_F_TIMESTAMP && Ergo12Table::kAgg[10] == EVREP_A_MAX &&
              Ergo12Table::kFunc[11] == EVREP_F_COUNT && Ergo12Table::kAgg[11] == EVREP_A_MEAN,
              "mdes_unit's split path is written against these triples");

template <typename OutT, typename D, bool HOT>
__device__ inline void mdes_unit(const BinView &bv, const int64_t *__restrict__ off, const MdesParams &P, int C, int H, int W,
                                 int nchunk, const UnitCfg &uc, double scale, OutT *__restrict__ out, WaveLds<OutT, HOT> &w,
                                 int uid, int part, int b0, int64_t n_win, const MetaRaw &mraw) {
    // The window's statistics and the per-channel set-up are formed AFTER the unit's front end (their loads were issued in front
    // of it: meta_prefetch), so that they are not a dependent step of every wave's latency chain.
    int32_t tmin = 0;
    double interval = 0.0;
    int lo[D::kMaxC], hi[D::kMaxC], want[D::kMaxC];
    bool active[D::kMaxC];
    auto setup = [&]() {
        const WindowMeta m = meta_finish(bv, off, b0, mraw);
        tmin = m.tmin;
        const int32_t tmax = m.tmax;
        // t = t - t.min(); t_s = t / (t.max() - t.min())  (mixed_density_event_stack.py:33,112-114)
        interval = (double)((int64_t)tmax - (int64_t)tmin);
        // "SBT": eight windows handed over as rank ranges (k_mdes_sbt_windows).  Never with the compile-time ERGO-12 descriptors;
        // the loaded values are made wave-uniform explicitly (readfirstlane), or every per-channel bound below moves from the
        // scalar to the vector registers (+25 VGPRs, an occupancy step).
        const bool custom = D::kCustomWindows && P.bounds != nullptr;
        MdesWindows mw = mdes_windows(n_win);
        uint32_t neg_flags = m.neg_flags, oob_flags = m.oob_flags;
        const int fstride = custom ? 8 : 7, wmax = custom ? 7 : 6;
        if (custom) {
            const int32_t *bw = P.bounds + (size_t)b0 * 16;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                mw.lo[i] = __builtin_amdgcn_readfirstlane(bw[2 * i]);
                mw.hi[i] = __builtin_amdgcn_readfirstlane(bw[2 * i + 1]);
            }
            neg_flags = (uint32_t)__builtin_amdgcn_readfirstlane((int)P.wflags[2 * b0]);
            oob_flags = (uint32_t)__builtin_amdgcn_readfirstlane((int)P.wflags[2 * b0 + 1]);
        }
        // per-channel uniform setup
#pragma unroll
        for (int c = 0; c < D::kMaxC; ++c) {
            lo[c] = 0; hi[c] = 0; want[c] = kWantAny; active[c] = false;
            if (c < C) {
                const int wi = D::win(P, c), f = D::func(P, c), a = D::agg(P, c);
                bool ok = wi >= 0 && wi <= wmax && f >= 0 && f <= 6 && a >= 0 && a <= 3 && n_win > 0;
                int l = 0, h = 0;
#pragma unroll
                for (int i = 0; i < 8; ++i) if (wi == i) { l = mw.lo[i]; h = mw.hi[i]; }
                int wn = kWantAny, field = 0;
                if (f == EVREP_F_TIMESTAMP_POS || f == EVREP_F_COUNT_POS) { wn = 1; field = 1; }
                if (f == EVREP_F_TIMESTAMP_NEG || f == EVREP_F_COUNT_NEG) {
                    // rows with p == -1; if the window has none, rows with p == 0 (operations.py:59-61,78-80)
                    const bool has_neg = ok && ((neg_flags >> (wi & 7)) & 1u);
                    wn = has_neg ? -1 : 0;
                    field = has_neg ? 2 : 3;
                }
                // an out-of-range index inside the selected rows raises in torch_scatter -> zero channel
                if (ok && ((oob_flags >> (fstride * field + (wi & 7))) & 1u)) ok = false;
                lo[c] = l; hi[c] = h; want[c] = wn; active[c] = ok;
            }
        }
    };

    // (cnt, s, s2) of every channel -> the pixel's values
    // (always inlined: out of line its array arguments live in scratch -- 672 bytes per lane and a 3.5x slower launch, seen in r05c when an
    //  unrelated change tipped the inliner)
    auto finish = [&](const int(&cnt)[D::kMaxC], const double(&s)[D::kMaxC], const double(&s2)[D::kMaxC], OutT(&vals)[D::kMaxC]) __attribute__((always_inline)) {
        double rr[D::kMaxC];
#pragma unroll
        for (int c = 0; c < D::kMaxC; ++c) {
            double r = 0.0;
            if (c < C && active[c]) {
                const int f = D::func(P, c), a = D::agg(P, c);
                const double n = (double)cnt[c];
                const double d = (double)(cnt[c] < 1 ? 1 : cnt[c]);
                if (is_count_func(f) && a != EVREP_A_MAX) {
                    // sum = n; mean = n / max(n,1) = 1 or 0; variance = mean(1) - mean(1)^2 = 0 exactly
                    r = (a == EVREP_A_SUM) ? n : ((a == EVREP_A_MEAN) ? (cnt[c] > 0 ? 1.0 : 0.0) : 0.0);
                } else if (a == EVREP_A_SUM) r = s[c];
                else if (a == EVREP_A_MEAN) {
                    // x / 1.0 == x; a float64 division is ~13 double-rate instructions for the WHOLE wave, so it is only
                    // entered when some pixel of the wave really holds more than one event of the channel
                    r = s[c];
                    if (__any(cnt[c] > 1)) r = cnt[c] > 1 ? s[c] / d : s[c];
                } else if (a == EVREP_A_MAX) r = cnt[c] > 0 ? s[c] : 0.0;
                else {
                    double mean = s[c], mean2 = s2[c];
                    if (__any(cnt[c] > 1)) {
                        mean = cnt[c] > 1 ? s[c] / d : s[c];
                        mean2 = cnt[c] > 1 ? s2[c] / d : s2[c];
                    }
                    const double mm = mean * mean;
                    r = mean2 - mm;
                }
            }
            rr[c] = r;
        }
        if (scale != 1.0) {   // x * 1.0 == x: the unscaled call (wave-uniform) skips its float64 multiplies
#pragma unroll
            for (int c = 0; c < D::kMaxC; ++c) rr[c] = rr[c] * scale;
        }
#pragma unroll
        for (int c = 0; c < D::kMaxC; ++c) vals[c] = (OutT)rr[c];
    };

    // the float64 instance sweeps in its MAIN launch, the float32 instance in its HOT launch (UnitSplit, IN_HOT): kSplit = this
    // instance runs the sweep and emits from it, kSplitDefer = this instance hands such units over
    constexpr bool kSplitHot = MdesIsErgo12<D>::value && sizeof(OutT) < 8;
    constexpr bool kSplit = MdesIsErgo12<D>::value && (kSplitHot ? HOT : !HOT);
    constexpr bool kSplitDefer = kSplitHot && !HOT;
    ChunkGeom g;
    UnitRecs u;
    bool esc = false;   // the split sweep met a record it cannot take
    if constexpr (kSplit) {
        uint32_t *words = reinterpret_cast<uint32_t *>(w.tile);
        // What the sweep needs of the window, in a handful of scalars (the per-channel set-up proper stays behind the front end,
        // where it always was: its values would ride through the sweep in registers): the seven rank windows, tmin, and per
        // channel the polarity classes it takes -- bit 3 c + k of `cmask`, k = 0 / 1 / 2 for p < 0 / p == 0 / p > 0; none: the
        // channel is inactive.
        MdesWindows smw;
        int32_t stmin = 0;
        uint64_t pmask = 0ull;   // bits [16 k, 16 k + 12), k = 0 / 1 / 2 for p < 0 / p == 0 / p > 0: bit c = channel c takes a record of that class
                                 // (ONE captured scalar: three of them, picked by a select, kept the closure -- and with it the kernel's argument
                                 //  structs -- in scratch: 672 bytes per lane, the launch 3.5x slower)
        // the channels of a rank window as bits (compile-time): a record's window membership (7 bits) -> its channels (12 bits) in
        // seven selects, its polarity class -> one of the three masks above, and every channel's hit test is one bit of their AND
        // (r05c: twelve window tests and twelve 64-bit shifts per record until then -- the sweep is bound by instruction issue)
        auto sbegin = [&]() -> bool {
            const WindowMeta m = meta_finish(bv, off, b0, mraw);
            stmin = __builtin_amdgcn_readfirstlane(m.tmin);
            smw = mdes_windows(n_win);
            const uint32_t neg_flags = (uint32_t)__builtin_amdgcn_readfirstlane((int)m.neg_flags);
            const uint32_t oob_flags = (uint32_t)__builtin_amdgcn_readfirstlane((int)m.oob_flags);
            uint32_t a0 = 0, a1 = 0, a2 = 0;
#pragma unroll
            for (int c = 0; c < D::kMaxC; ++c) {
                const int wi = D::win(P, c), f = D::func(P, c);
                uint32_t cls = 7u;   // bit k: the channel takes class k (0: p < 0, 1: p == 0, 2: p > 0)
                int field = 0;
                if (f == EVREP_F_TIMESTAMP_POS || f == EVREP_F_COUNT_POS) { cls = 4u; field = 1; }
                if (f == EVREP_F_TIMESTAMP_NEG || f == EVREP_F_COUNT_NEG) {
                    const bool has_neg = (neg_flags >> wi) & 1u;   // operations.py:59-61,78-80
                    cls = has_neg ? 1u : 2u;
                    field = has_neg ? 2 : 3;
                }
                if (n_win <= 0 || ((oob_flags >> (7 * field + wi)) & 1u)) cls = 0u;
                a0 |= (cls & 1u) << c; a1 |= ((cls >> 1) & 1u) << c; a2 |= ((cls >> 2) & 1u) << c;
            }
            pmask = (uint64_t)a0 | ((uint64_t)a1 << 16) | ((uint64_t)a2 << 32);
            return true;
        };
        auto sf = [&](uint32_t px, const Rec8 &q, uint2 &e, const uint2 &) -> bool {
            const int rank = (int)(q.y >> 11);
            const uint32_t p2 = (q.y >> 9) & 3u;
            if (p2 == 3u) { esc = true; return false; }
            const int p = (int)p2 - 1;
            const uint32_t tt = (uint32_t)((int64_t)(int32_t)q.x - (int64_t)stmin);   // 0 <= t - tmin < 2^32
            uint32_t *wd = words + px * (uint32_t)kErgoSplitWords;
            // the record's windows -> its channels: w0 holds every rank, w1..w3 are consecutive thirds [lo1, hi1) [hi1, hi2) [hi2, hi3), w4..w6
            // run from their start to the window's end (mdes_windows) -- six compares instead of seven range tests and seven selects
            uint32_t chans = ergo_chans_of(0);
            chans |= rank < smw.hi[1] ? ergo_chans_of(1) : (rank < smw.hi[2] ? ergo_chans_of(2) : (rank < smw.hi[3] ? ergo_chans_of(3) : 0u));
            chans |= rank >= smw.lo[4] ? ergo_chans_of(4) : 0u;
            chans |= rank >= smw.lo[5] ? ergo_chans_of(5) : 0u;
            chans |= rank >= smw.lo[6] ? ergo_chans_of(6) : 0u;
            const uint32_t hits = chans & (uint32_t)(pmask >> (16u * p2));
            auto hit = [&](int c) -> bool { return (hits >> c) & 1u; };
            const uint32_t pinc = p > 0 ? 1u : 0x10000u;
            if (hit(0)) { if (p != 0) atomicAdd(wd + 0, pinc); else atomicAdd(wd + 1, 1u); }
            if (hit(3) && p != 0) atomicAdd(wd + 2, pinc);
            if (hit(5)) atomicAdd(wd + 1, 0x10000u);
            const bool h8 = hit(8), h9 = hit(9), h10 = hit(10), h6 = hit(6), h1 = hit(1);
            // word 3, low half: presence flags by CHANNEL index (ch2, ch4, ch7, ch11: occupancy; ch8, ch9, ch10: a maximum exists)
            const uint32_t bits = hits & ((1u << 2) | (1u << 4) | (1u << 7) | (1u << 11) | (1u << 8) | (1u << 9) | (1u << 10));
            if (bits) atomicOr(wd + 3, bits);
            if (h6) atomicAdd(wd + 3, 0x10000u);
            if (h8) atomicMax(wd + 4, tt);
            if (h9) atomicMax(wd + 5, tt);
            if (h10) atomicMax(wd + 6, tt);
            e = make_uint2(tt, px);
            return h6 || h1;
        };
        auto sdone = [&]() -> bool { return !__any(esc); };
        // a time slice's words of one pixel into the unit's (sub-waves of a sliced hot unit, unit_records): the 16-bit counts add, the
        // flags or, the maxima max; word 3 holds both flags (low half) and a count (high half)
        auto smerge = [](const uint32_t *m, uint32_t *gw) {
            if (m[0]) atomicAdd(gw + 0, m[0]);
            if (m[1]) atomicAdd(gw + 1, m[1]);
            if (m[2]) atomicAdd(gw + 2, m[2]);
            if (m[3] & 0xffffu) atomicOr(gw + 3, m[3] & 0xffffu);
            if (m[3] >> 16) atomicAdd(gw + 3, m[3] & 0xffff0000u);
            if (m[4]) atomicMax(gw + 4, m[4]);
            if (m[5]) atomicMax(gw + 5, m[5]);
            if (m[6]) atomicMax(gw + 6, m[6]);
        };
        u = unit_front<OutT, HOT, false, NoVisit>(bv, off, H, W, nchunk, uc, w, g, uid, part, NoVisit(), unit_split_merge<kSplitHot>(sbegin, sf, sdone, kErgoSplitWords, smerge));
    } else if constexpr (kSplitDefer) {
        auto never = []() -> bool { return false; };
        auto nof = [](uint32_t, const Rec8 &, uint2 &, const uint2 &) -> bool { return false; };
        // lane k: the status word of the window's block k (meta_prefetch: q2.x), merged by unit_records only when a unit is hot
        auto sp = unit_split<true>(never, nof, never, kErgoSplitWords, (uint32_t)mraw.q2.x | ((uc.xflags & 14) ? 0u : kStEscaped),
                                   (uc.xflags & 2) ? 0u : kHotSubMin);
        sp.coop_min = (uc.xflags & 8) ? kErgoCoopMin : 0u;   // r06: a cooperative launch of sixteen waves per unit takes the big ones (k_mdes_coop)
        u = unit_front<OutT, HOT, false, NoVisit>(bv, off, H, W, nchunk, uc, w, g, uid, part, NoVisit(), sp);
        // (two-chunk units -- sparse windows, 640x480 / 1280x720 at 50 000 - 200 000 events -- keep the ordered ways: measured, r05b,
        //  their hot units are few and huge -- 4 000 to 20 000 records, one wave's instruction stream each, 30 to 100 us of sweep --
        //  and the hot launch's tail costs 5-8 % more than it saves; at the reference's Gen1 shape the hand-over takes the
        //  circle / edge streams from 133 / 125 us to 90 / 106)
    } else {
        u = unit_front(bv, off, H, W, nchunk, uc, w, g, uid, part);
    }
    if (u.deferred) return;
    w.mark(0);
    OutT *dst = out + (((size_t)g.b * H + g.row) * (size_t)W + g.c0) * C;
    setup();
    w.mark(1);

    if constexpr (kSplit) {
        if (u.part == -5) {   // wave-uniform: the unit was swept by the split; its kept records wait, in time order, in the list
            const int lane = threadIdx.x;
            const uint32_t nk = u.pst;
            const uint32_t *words = reinterpret_cast<const uint32_t *>(w.tile);
            volatile uint32_t *cnt = reinterpret_cast<volatile uint32_t *>(w.segs);
            scan_pixel_counters(reinterpret_cast<uint32_t *>(w.segs), u.npixu);
            const uint32_t room = (uint32_t)(reinterpret_cast<const unsigned char *>(w.segs) - reinterpret_cast<const unsigned char *>(w.tile));
            const uint32_t lcap = split_list_cap(room, u.pen);
            const uint32_t nrec = u.ce - u.cs;
            const bool sliced = u.sub > 0;   // wave-uniform: the unit was swept in time slices; their kept lists lie in its slot, one region each
            const bool big = nk > lcap || sliced;   // wave-uniform: a hot unit -- its kept records are ordered in global memory
            const uint2 *list = reinterpret_cast<const uint2 *>(reinterpret_cast<const unsigned char *>(w.tile) + u.pen);
            double *placed = reinterpret_cast<double *>(reinterpret_cast<unsigned char *>(w.tile) + u.pen);
            const uint2 *glist = reinterpret_cast<const uint2 *>(bv.spill + u.cs);
            double *gplaced = sliced ? bv.placed_pool + u.cs : reinterpret_cast<double *>(bv.spill + u.cs) + nrec;
            // sliced: kept record j of the unit = record j - spre of the slice whose exclusive kept prefix spre covers j; lane s
            // holds slice s's prefix and its region's first slot
            uint32_t spre = 0xffffffffu, soff = 0u;
            const unsigned long long *sublists = reinterpret_cast<const unsigned long long *>(
                reinterpret_cast<const unsigned char *>(bv.spill + u.cs) + ((kHotSubHdrBytes + 4u * split_gwpp(kErgoSplitWords) * (uint32_t)u.npixu + 15u) & ~15u));
            if (sliced) {
                const uint32_t *hdr = reinterpret_cast<const uint32_t *>(bv.spill + u.cs);
                uint32_t tot = 0;
                if (lane < u.sub) tot = __hip_atomic_load(hdr + 1 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const uint32_t incl = wave_incl_scan(tot);
                if (lane < u.sub) { spre = incl - tot; soff = (uint32_t)lane * hot_sub_quota(nrec, (uint32_t)u.sub); }
            }
            auto sub_at = [&](uint32_t j) -> uint2 {
                uint32_t at = (uint32_t)__builtin_amdgcn_readlane((int)soff, 0) + j;
                for (int k = 1; k < u.sub; ++k) {
                    const uint32_t pk = (uint32_t)__builtin_amdgcn_readlane((int)spre, k);
                    const uint32_t ok = (uint32_t)__builtin_amdgcn_readlane((int)soff, k);
                    if (j >= pk) at = ok + (j - pk);
                }
                const unsigned long long v = __hip_atomic_load(sublists + at, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return make_uint2((uint32_t)v, (uint32_t)(v >> 32));
            };
            const int nbits = 32 - __builtin_clz((unsigned)u.npixu - 1u);
            auto place = [&](const uint2 &e, bool valid, double *dstp) {
                const uint32_t px = valid ? e.y : 0u;
                uint32_t rk; bool last;
                wave_match(px, nbits, valid, lane, rk, last);
                uint32_t pos = 0;
                // the digest: the normalised timestamp's division, one record per lane (mixed_density_event_stack.py:112-114)
                if (valid) { pos = cnt[px] + rk; dstp[pos] = (double)e.x / interval; }
                __builtin_amdgcn_wave_barrier();
                if (valid && last) cnt[px] = pos + 1;
                __builtin_amdgcn_wave_barrier();
            };
            if (!big) {
                // the whole list rides in registers while its records move to their pixels' places: the ordered copy takes the list's room
                uint2 ek[kSplitBatches];
#pragma unroll
                for (int i = 0; i < kSplitBatches; ++i) {
                    const uint32_t j = (uint32_t)(i * kWave + lane);
                    ek[i] = make_uint2(0u, 0u);
                    if (j < nk) ek[i] = list[j];
                }
                wave_phase();
#pragma unroll
                for (int i = 0; i < kSplitBatches; ++i)
                    if ((uint32_t)(i * kWave) < nk) place(ek[i], (uint32_t)(i * kWave + lane) < nk, placed);   // uniform
            } else {
                // the wave reads back what its own lanes stored in the slot: same CU, same vector L1 -- workgroup-scope release / acquire
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                for (uint32_t j0 = 0; j0 < nk; j0 += 4 * kWave) {
                    uint2 ek[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const uint32_t j = j0 + (uint32_t)(i * kWave + lane);
                        ek[i] = make_uint2(0u, 0u);
                        if (j < nk) {
                            if (sliced) ek[i] = sub_at(j);
                            else if (j < lcap) ek[i] = list[j];
                            else { const double d = gload_f64(reinterpret_cast<const double *>(glist + j)); ek[i] = make_uint2((uint32_t)__double2loint(d), (uint32_t)__double2hiint(d)); }
                        }
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (j0 + (uint32_t)(i * kWave) < nk) place(ek[i], j0 + (uint32_t)(i * kWave + lane) < nk, gplaced);
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            }
            wave_phase();
            w.mark(4);
            constexpr int V = 16 / (int)sizeof(OutT);
            const bool vec = (C % V) == 0 && (reinterpret_cast<uintptr_t>(dst) & 15u) == 0;   // wave-uniform
            w.pace();
            for (int pt = 0; pt * kWave < g.npix; ++pt) {
                const int np = min(kWave, g.npix - pt * kWave);
                const bool own = lane < np;
                const uint32_t px = (uint32_t)(pt * kWave + lane);
                uint32_t st = 0, en = 0, a3 = 0;
                const uint32_t *wd = words + px * (uint32_t)kErgoSplitWords;
                if (own) { en = cnt[px]; st = px ? cnt[px - 1] : 0u; a3 = wd[3]; }
                // the ordered part: the pixel's ch6 records, then its ch1 records, each in time order
                const uint32_t mid = st + (a3 >> 16);
                double acc6 = 0.0, acc1 = 0.0, sq1 = 0.0;
                if (!big) {
                    for (uint32_t j = st; j < mid; ++j) acc6 = acc6 + placed[j];
                    for (uint32_t j = mid; j < en; ++j) { const double v = placed[j]; acc1 = acc1 + v; const double vv = v * v; sq1 = sq1 + vv; }
                } else {
                    // the part's records, [ra, rb) of the slot, come through the list's room in LDS `lcap` at a time (coalesced loads,
                    // all in flight together); a lane walks what the piece holds of its pixel and carries its sums to the next piece
                    const uint32_t ra = pt ? cnt[pt * kWave - 1] : 0u, rb = cnt[pt * kWave + np - 1];   // wave-uniform
                    for (uint32_t c0 = ra; c0 < rb; c0 += lcap) {
                        const uint32_t n = min(lcap, rb - c0);
                        double tmp[kSplitBatches];
#pragma unroll
                        for (int i = 0; i < kSplitBatches; ++i) {
                            const uint32_t j = (uint32_t)(i * kWave + lane);
                            tmp[i] = 0.0;
                            if (j < n) tmp[i] = gload_f64(gplaced + c0 + j);
                        }
                        wave_phase();   // the previous piece's walks are done
#pragma unroll
                        for (int i = 0; i < kSplitBatches; ++i) {
                            const uint32_t j = (uint32_t)(i * kWave + lane);
                            if (j < n) placed[j] = tmp[i];
                        }
                        wave_phase();
                        const uint32_t jl = max(st, c0), jh = min(en, c0 + n);
                        for (uint32_t j = jl; j < jh; ++j) {
                            const double v = placed[j - c0];
                            if (j < mid) acc6 = acc6 + v;
                            else { acc1 = acc1 + v; const double vv = v * v; sq1 = sq1 + vv; }
                        }
                    }
                }
                if (own) {
                    const uint32_t a0 = wd[0], a1 = wd[1], a2 = wd[2], m8 = wd[4], m9 = wd[5], m10 = wd[6];
                    int cn[D::kMaxC];
                    double s[D::kMaxC], s2[D::kMaxC];
#pragma unroll
                    for (int c = 0; c < D::kMaxC; ++c) { cn[c] = 0; s[c] = 0.0; s2[c] = 0.0; }
                    const int np0 = (int)(a0 & 0xffffu), nn0 = (int)(a0 >> 16), nz0 = (int)(a1 & 0xffffu);
                    cn[0] = np0 + nn0 + nz0; s[0] = (double)(np0 - nn0); s2[0] = (double)(np0 + nn0);
                    s[3] = (double)((int)(a2 & 0xffffu) - (int)(a2 >> 16));
                    cn[5] = (int)(a1 >> 16);
                    cn[2] = (int)((a3 >> 2) & 1u); cn[4] = (int)((a3 >> 4) & 1u); cn[7] = (int)((a3 >> 7) & 1u); cn[11] = (int)((a3 >> 11) & 1u);
                    cn[8] = (int)((a3 >> 8) & 1u); cn[10] = (int)((a3 >> 10) & 1u); cn[9] = (int)((a3 >> 9) & 1u);
                    s[8] = (double)m8 / interval; s[9] = (double)m9 / interval; s[10] = (double)m10 / interval;
                    cn[6] = (int)(mid - st); s[6] = acc6;
                    cn[1] = (int)(en - mid); s[1] = acc1; s2[1] = sq1;
                    OutT vals[D::kMaxC];
                    finish(cn, s, s2, vals);
                    store_pixel<OutT, D::kMaxC>(dst + ((size_t)pt * kWave + lane) * C, vals, C, vec);
                }
            }
            w.mark(5);
            return;
        }
    }

    // Float64 divisions are the expensive instructions of this kernel, and the segment walks are divergent (one step per
    // event of the longest segment among the 64 lanes).  So the normalised timestamp t_s = (t - tmin) / interval of the
    // staged records is formed BEFORE the walks (emit_chunk's digest), one record per lane, all lanes at once, and
    // staged in place of the fields the walks do not need: {t_s lo, t_s hi, rank, p}.  Bit for bit the same quotient the reference forms
    // per event (mixed_density_event_stack.py:112-114).  Further:
    //  * `max` of the normalised timestamp is the maximum of the quotients themselves (the reference's own order);
    //  * a count of 1 (the usual case: most pixels see one event of a window): x / 1.0 == x exactly, so mean and
    //    variance skip their divisions.
    auto digest = [&](const Rec &r) -> Rec {
        const double ts = (double)((int64_t)r.z - (int64_t)tmin) / interval;  // exact numerator: |t - tmin| < 2^32
        return make_int4(__double2loint(ts), __double2hiint(ts), r.y, r.w);
    };
    auto reduce = [&](uint32_t jb, uint32_t je, auto get, OutT(&vals)[D::kMaxC]) {
        double s[D::kMaxC], s2[D::kMaxC];
        int cnt[D::kMaxC];
#pragma unroll
        for (int c = 0; c < D::kMaxC; ++c) { s[c] = 0.0; s2[c] = 0.0; cnt[c] = 0; }
        for (uint32_t j = jb; j < je; ++j) {
            const Rec e = get(j);  // digest: {t_s lo, t_s hi, rank, p}
            const int rank = e.z, p = e.w;
            const double tn = __hiloint2double(e.y, e.x);
            const double pv = (double)p;
#pragma unroll
            for (int c = 0; c < D::kMaxC; ++c) {
                if (c < C && active[c]) {
                    const bool hit = rank >= lo[c] && rank < hi[c] && (want[c] == kWantAny || p == want[c]);
                    const int f = D::func(P, c), a = D::agg(P, c);
                    const bool is_t = !(f == EVREP_F_POLARITY) && !is_count_func(f);
                    if (hit) {
                        if (a == EVREP_A_MAX) {
                            const double v = is_t ? tn : ((f == EVREP_F_POLARITY) ? pv : 1.0);
                            if (cnt[c] == 0 || v > s[c]) s[c] = v;
                        } else if (is_count_func(f)) {
                            // src = ones: sum, sum of squares and count coincide (exact small integers)
                        } else {
                            const double v = (f == EVREP_F_POLARITY) ? pv : tn;
                            s[c] = s[c] + v;
                            if (a == EVREP_A_VARIANCE) { const double vv = v * v; s2[c] = s2[c] + vv; }
                        }
                        ++cnt[c];
                    }
                }
            }
        }
        finish(cnt, s, s2, vals);
    };
    emit_chunk<OutT, D::kMaxC, HOT>(u, digest, g.row * W + g.c0, g.npix, C, dst, w, (const OutT *)nullptr, reduce);
}

template <typename OutT, typename D, bool HOT = false>
__global__ __launch_bounds__(kWave, HOT ? ((MdesIsErgo12<D>::value && sizeof(OutT) < 8) ? 3 : 4) : 1) void k_mdes(BinView bv, const int64_t *__restrict__ off,
                                               MdesParams P, int H, int W, int nchunk, UnitCfg uc, double scale,
                                               OutT *__restrict__ out) {
    extern __shared__ __align__(16) unsigned char smem[];
    run_units<HOT>(bv, [&](int uid, int part) {
        if (HOT && part == kHotCoop) return;   // (k_mdes_coop's item)
        const int C = D::C(P);
        WaveLds<OutT, HOT> w(smem, C, (uc.span + uc.merge) * kChunkPx, uc.stage, uc.partpx);
        w.arm(uc.hold);
#ifdef EVREP_TIMING
        w.dbg = bv.dbg + 8 * (size_t)uid;
#endif
        // every independent global load first: the window's extent and block statistics, the unit's run tables -- then the
        // unit's records
        int chunk0;
        const int b0 = unit_geom(H, W, nchunk, uc, chunk0, uid).b;
        const int64_t n_win = off[b0 + 1] - off[b0];
        const MetaRaw mraw = meta_prefetch(bv, b0);
        w.mark(6);
        mdes_unit<OutT, D, HOT>(bv, off, P, C, H, W, nchunk, uc, scale, out, w, uid, part, b0, n_win, mraw);
    });
}


// --------------------------------------------------------------------------------------------
// A5 (r06), after the key-sorted pass: ERGO-12 as a STREAM (optimized_representation.py:87-115 over mixed_density_event_stack.py /
// operations.py, as mdes_unit's split path reads them).  Ten of the twelve channels are order-free and live as integer words per pixel
// of the unit, bumped by LDS atomics in ONE sweep of the unit's records in array order (no grouping, no stage, no spill slot, no hot
// launch); the two float64 sums that have to run in array order -- ch6 (w2, timestamp_pos, mean) and ch1 (w3, timestamp_neg,
// variance), a sixth of the records each -- are added by the record's own lane after a leader election per pixel and batch (the
// lowest pending lane of a pixel goes first: k_voxel_stream's rounds), the normalised timestamp's division formed one record per
// lane.  A window that holds polarity values outside {-1, 0, 1} (escaped in its 8-byte records) cannot keep ch0 / ch3 as integer
// counts (the sums of p and p^2 may leave the integers float64 holds exactly): there EVERY record goes through the election and
// those sums run in array order too.  Then a lane per pixel forms the twelve values (mdes_unit's finish) over the pixel's own
// state and the unit leaves as one coalesced burst.
// State of pixel px, 20 words:  0 ch0 #(p > 0), 1 ch0 #(p < 0), 2 ch0 #(p == 0), 3 ch5 n, 4 ch3 #(p > 0), 5 ch3 #(p < 0), 6 presence
// bits by channel, 7-9 max (t - tmin) of ch8 / ch9 / ch10, 10 n6, 11 n1, 12-13 sum6, 14-15 sum1, 16-17 sumsq1, 18-19 (escaped) sumsq0;
// escaped windows: 0-1 sum0 (float64), 2 n0, 4-5 sum3 (float64).
// LDS: state [npixa * 20] u32, overlaid by the tile [npixa * 12] OutT (float64: written from the last pixel batch down) | tag [npixa] |
// head [64 * RB] | srcs [128]
constexpr int kErgoStreamWords = 20;
__host__ __device__ inline size_t mdes_stream_lds_bytes(int npixa, size_t elem, int rb) {
    const size_t st = (size_t)npixa * kErgoStreamWords * 4, tl = (size_t)npixa * 12 * elem;
    return align16(st > tl ? st : tl) + (size_t)npixa * 4 + (size_t)(64 * rb) * 4 + 128 * 4;
}
#ifndef EVREP_MS_WAVES
#define EVREP_MS_WAVES 4
#endif
template <typename OutT, int RB>
__global__ __launch_bounds__(kWave, EVREP_MS_WAVES) void k_mdes_stream(BinView bv, const int64_t *__restrict__ off, int H, int W, int nchunk,
                                                                    UnitCfg uc, double scale, OutT *__restrict__ out) {
    extern __shared__ __align__(16) unsigned char smem[];
    using T = Ergo12Table;
    constexpr int C = 12, NWD = kErgoStreamWords;
    const int lane = threadIdx.x;
    const int uid = chunk_unit((int)(gridDim.x * gridDim.y * gridDim.z));
    int chunk, nch;
    const ChunkGeom g = unit_geom(H, W, nchunk, uc, chunk, nch, uid);
    const int b = g.b;
    const int64_t beg = off[b];
    const int64_t n_win = off[b + 1] - beg;
    const MetaRaw mraw = meta_prefetch(bv, b);
    const int npixa = (uc.span + uc.merge) * kChunkPx;
    const size_t st_bytes = (size_t)npixa * NWD * 4, tl_bytes = (size_t)npixa * C * sizeof(OutT);
    uint32_t *words = reinterpret_cast<uint32_t *>(smem);
    uint32_t *tag = reinterpret_cast<uint32_t *>(smem + align16(st_bytes > tl_bytes ? st_bytes : tl_bytes));
    uint32_t *head = tag + npixa;
    uint32_t *srcs = head + 64 * RB;
    {
        uint4 *z = reinterpret_cast<uint4 *>(words);
        const int nvec = (g.npix * NWD + 3) / 4;
        for (int v = lane; v < nvec; v += kWave) z[v] = make_uint4(0u, 0u, 0u, 0u);
        uint4 *t4 = reinterpret_cast<uint4 *>(tag);
        for (int v = lane; v * 4 < npixa; v += kWave) t4[v] = make_uint4(~0u, ~0u, ~0u, ~0u);
    }
    const WindowMeta m = meta_finish(bv, off, b, mraw);
    const int32_t tmin = m.tmin;
    const double interval = (double)((int64_t)m.tmax - (int64_t)tmin);   // t_s = (t - t.min()) / (t.max() - t.min())  (:33,112-114)
    const MdesWindows mw = mdes_windows(n_win);
    const bool escaped = (m.status & kStEscaped) != 0u;   // wave-uniform
    // per polarity class k = 0 / 1 / 2 (p == -1 / 0 / +1) the channels that take a record of the class; `many`: the channels that take
    // every record (the only ones an escaped polarity value can hit).  mdes_unit's set-up: an out-of-range index inside a channel's rows
    // raises in torch_scatter -> zero channel (no hits); timestamp_neg / count_neg fall back to p == 0 when the window has no p == -1.
    uint32_t a0 = 0, a1 = 0, a2 = 0, many = 0;
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const int wi = T::kWin[c], f = T::kFunc[c];
        uint32_t cls = 7u;
        int field = 0;
        if (f == EVREP_F_TIMESTAMP_POS || f == EVREP_F_COUNT_POS) { cls = 4u; field = 1; }
        if (f == EVREP_F_TIMESTAMP_NEG || f == EVREP_F_COUNT_NEG) {
            const bool has_neg = (m.neg_flags >> wi) & 1u;   // operations.py:59-61,78-80
            cls = has_neg ? 1u : 2u;
            field = has_neg ? 2 : 3;
        }
        if (n_win <= 0 || ((m.oob_flags >> (7 * field + wi)) & 1u)) cls = 0u;
        a0 |= (cls & 1u) << c; a1 |= ((cls >> 1) & 1u) << c; a2 |= ((cls >> 2) & 1u) << c;
        if (cls == 7u) many |= 1u << c;
    }
    const uint64_t pmask = (uint64_t)a0 | ((uint64_t)a1 << 16) | ((uint64_t)a2 << 32);
    const int4 *evw = bv.ev + beg;
    const int c0 = g.c0;
    wave_phase();
    stream_unit_records<RB>(bv, b, beg, n_win, H * nchunk, g.row * nchunk + chunk, g.row * nchunk + chunk + nch, head, srcs, StreamNoPre(),
        [&](bool have, const Rec8 &q, const uint2 &) {
            const uint32_t px = have ? ((q.y & 511u) - (uint32_t)c0) & 511u : 0u;
            const int rank = (int)(q.y >> 11);
            const uint32_t p2 = (q.y >> 9) & 3u;
            int p = (int)p2 - 1;
            if (have && p2 == 3u) p = evw[rank].w;
            const uint32_t tt = (uint32_t)((int64_t)(int32_t)q.x - (int64_t)tmin);   // 0 <= t - tmin < 2^32
            uint32_t *wd = words + px * (uint32_t)NWD;
            // the record's rank windows -> its channels (mdes_unit's sweep: w0 every rank, w1..w3 consecutive thirds, w4..w6 run to the end)
            uint32_t chans = ergo_chans_of(0);
            chans |= rank < mw.hi[1] ? ergo_chans_of(1) : (rank < mw.hi[2] ? ergo_chans_of(2) : (rank < mw.hi[3] ? ergo_chans_of(3) : 0u));
            chans |= rank >= mw.lo[4] ? ergo_chans_of(4) : 0u;
            chans |= rank >= mw.lo[5] ? ergo_chans_of(5) : 0u;
            chans |= rank >= mw.lo[6] ? ergo_chans_of(6) : 0u;
            const uint32_t cmask = p2 == 3u ? many : (uint32_t)(pmask >> (16u * p2));
            const uint32_t hits = have ? (chans & cmask) : 0u;
            auto hit = [&](int c) -> bool { return (hits >> c) & 1u; };
            if (!escaped) {
                if (hit(0)) atomicAdd(wd + (p > 0 ? 0 : (p < 0 ? 1 : 2)), 1u);
                if (hit(3) && p != 0) atomicAdd(wd + (p > 0 ? 4 : 5), 1u);
            }
            if (hit(5)) atomicAdd(wd + 3, 1u);
            // presence bits by channel; bit 31: the pixel holds a record at all (an EMPTY pixel is +0 unscaled, as the ordered paths leave it)
            const uint32_t bits = hits & ((1u << 2) | (1u << 4) | (1u << 7) | (1u << 11) | (1u << 8) | (1u << 9) | (1u << 10));
            if (have) atomicOr(wd + 6, bits | 0x80000000u);
            if (hit(8)) atomicMax(wd + 7, tt);
            if (hit(9)) atomicMax(wd + 8, tt);
            if (hit(10)) atomicMax(wd + 9, tt);
            // the ordered part: in array order per pixel
            const bool h6 = hit(6), h1 = hit(1);
            const bool h0 = escaped && hit(0), h3 = escaped && hit(3);
            bool pend = h6 || h1 || h0 || h3;
            if (__any(pend)) {
                const double ts = (double)tt / interval;   // the digest: one division per record and lane (mixed_density_event_stack.py:112-114)
                const double pv = (double)p;
                while (__any(pend)) {
                    if (pend) atomicMin(&tag[px], (uint32_t)lane);
                    wave_phase();
                    const bool win = pend && tag[px] == (uint32_t)lane;
                    wave_phase();
                    if (win) {
                        double *dd = reinterpret_cast<double *>(wd);
                        if (h6) { dd[6] = dd[6] + ts; wd[10] = wd[10] + 1u; }
                        if (h1) { dd[7] = dd[7] + ts; const double vv = ts * ts; dd[8] = dd[8] + vv; wd[11] = wd[11] + 1u; }
                        if (h0) { dd[0] = dd[0] + pv; const double vv = pv * pv; dd[9] = dd[9] + vv; wd[2] = wd[2] + 1u; }
                        if (h3) { dd[2] = dd[2] + pv; }
                        tag[px] = ~0u;
                        pend = false;
                    }
                    wave_phase();
                }
            }
        }, (uint32_t)EVREP_MDES_STREAM_BYRUN);
    wave_phase();
    // A lane per NON-EMPTY pixel: the twelve values from the pixel's state (mdes_unit's finish, channel by channel).  The non-empty
    // pixels of the unit are listed first (sparse windows: a sixth of the pixels), every round's values wait in registers until all
    // states are read, then the tile -- which overlays the states -- is zero-filled and patched.  Float64 divisions (~35 instructions
    // each for the whole wave) are only entered when some pixel of the wave needs them: x / 1.0 == x.
    OutT *tile = reinterpret_cast<OutT *>(smem);
    unsigned char *nelist = reinterpret_cast<unsigned char *>(head);   // [npixa] pixel indices (npixa <= 256)
    int nne = 0;
    for (int pt = 0; pt * kWave < g.npix; ++pt) {
        const int px = pt * kWave + lane;
        const bool ne = px < g.npix && (words[(uint32_t)px * (uint32_t)NWD + 6] >> 31);
        const uint64_t mne = __ballot(ne);
        if (ne) nelist[nne + __popcll(mne & ((1ull << lane) - 1ull))] = (unsigned char)px;
        nne += __popcll(mne);
    }
    wave_phase();
    constexpr int kRounds = 2;   // npixa = 128: at most two rounds of 64 non-empty pixels
    OutT rv[kRounds][C];
    int rpx[kRounds];
#pragma unroll
    for (int rd = 0; rd < kRounds; ++rd) {
        rpx[rd] = -1;
#pragma unroll
        for (int c = 0; c < C; ++c) rv[rd][c] = (OutT)0;
        if (rd * kWave < nne) {   // uniform
            const bool own = rd * kWave + lane < nne;
            const int px = own ? (int)nelist[rd * kWave + lane] : 0;
            const uint32_t *wd = words + (uint32_t)px * (uint32_t)NWD;
            const double *dd = reinterpret_cast<const double *>(wd);
            double r[C];
#pragma unroll
            for (int c = 0; c < C; ++c) r[c] = 0.0;
            const uint32_t fl = own ? wd[6] : 0u;
            // ch0 (w0, polarity, variance): n, sum p, sum p^2
            int n0 = 0; double s0 = 0.0, q0 = 0.0;
            if (own) {
                if (!escaped) { const int np0 = (int)wd[0], nn0 = (int)wd[1], nz0 = (int)wd[2]; n0 = np0 + nn0 + nz0; s0 = (double)(np0 - nn0); q0 = (double)(np0 + nn0); }
                else { n0 = (int)wd[2]; s0 = dd[0]; q0 = dd[9]; }
            }
            {
                double mean = s0, mean2 = q0;
                if (__any(n0 > 1)) { const double d = (double)(n0 < 1 ? 1 : n0); mean = n0 > 1 ? s0 / d : s0; mean2 = n0 > 1 ? q0 / d : q0; }
                const double mm = mean * mean;
                r[0] = mean2 - mm;
            }
            // ch1 (w3, timestamp_neg, variance)
            {
                const int n1 = own ? (int)wd[11] : 0;
                const double s1 = own ? dd[7] : 0.0, q1 = own ? dd[8] : 0.0;
                double mean = s1, mean2 = q1;
                if (__any(n1 > 1)) { const double d = (double)(n1 < 1 ? 1 : n1); mean = n1 > 1 ? s1 / d : s1; mean2 = n1 > 1 ? q1 / d : q1; }
                const double mm = mean * mean;
                r[1] = mean2 - mm;
            }
            r[2] = (fl >> 2) & 1u ? 1.0 : 0.0;                                        // (w2, count_neg, mean)
            if (own) r[3] = escaped ? dd[2] : (double)((int)wd[4] - (int)wd[5]);       // (w6, polarity, sum)
            r[4] = (fl >> 4) & 1u ? 1.0 : 0.0;                                        // (w5, count_pos, mean)
            if (own) r[5] = (double)(int)wd[3];                                        // (w6, count, sum)
            {                                                                          // (w2, timestamp_pos, mean)
                const int n6 = own ? (int)wd[10] : 0;
                const double s6 = own ? dd[6] : 0.0;
                r[6] = s6;
                if (__any(n6 > 1)) r[6] = n6 > 1 ? s6 / (double)n6 : s6;
            }
            r[7] = (fl >> 7) & 1u ? 1.0 : 0.0;                                        // (w5, count_neg, mean)
            // (w1, timestamp_neg, max), (w0, timestamp_pos, max), (w4, timestamp, max): the quotient of the maximum
            if (__any((fl >> 8) & 1u)) r[8] = (fl >> 8) & 1u ? (double)wd[7] / interval : 0.0;
            if (__any((fl >> 9) & 1u)) r[9] = (fl >> 9) & 1u ? (double)wd[8] / interval : 0.0;
            if (__any((fl >> 10) & 1u)) r[10] = (fl >> 10) & 1u ? (double)wd[9] / interval : 0.0;
            r[11] = (fl >> 11) & 1u ? 1.0 : 0.0;                                      // (w1, count, mean)
            if (scale != 1.0) {
#pragma unroll
                for (int c = 0; c < C; ++c) r[c] = r[c] * scale;
            }
            if (own) rpx[rd] = px;
#pragma unroll
            for (int c = 0; c < C; ++c) rv[rd][c] = (OutT)r[c];
        }
    }
    wave_phase();   // every state is read: the tile may take the states' place
    tile_fill(tile, g.npix, C, (const OutT *)nullptr);
    wave_phase();
#pragma unroll
    for (int rd = 0; rd < kRounds; ++rd) {
        if (rpx[rd] >= 0) {
            OutT *mine = tile + (size_t)rpx[rd] * C;
#pragma unroll
            for (int c = 0; c < C; ++c) mine[c] = rv[rd][c];
        }
    }
    wave_phase();
    OutT *dst = out + (((size_t)b * H + g.row) * (size_t)W + g.c0) * C;
    tile_store(tile, g.npix * C, dst);
}

// --------------------------------------------------------------------------------------------
// A5 (r06): the big hot units of the ORDERED float32 ERGO-12 builder (sparse windows) -- SIXTEEN waves per unit.
// Until r05 such a unit was swept in time slices by up to sixteen independent hot waves that merged their words through the unit's
// spill slot with global atomics, and the wave that took the last ticket ordered the kept records of ch6 / ch1 -- a third of the unit,
// 6 600 records of a 1 Mpx circle monster -- through global memory and walked them: 95-100 us of ONE wave, the launch's tail.  Here
// the main wave hands a unit of >= kErgoCoopMin records over as ONE item (kHotCoop) and a workgroup of kMcWaves waves takes it, in
// chunks of kMcChunk records (array order): every wave sweeps its contiguous share -- the order-free channels by LDS atomics on the
// unit's SHARED state (k_mdes_stream's twenty words per pixel), the kept records counted per (wave, cell = pixel x {ch6, ch1}); a scan
// makes cell starts and per-(wave, cell) cursors; the same sweep again places the kept records' normalised timestamps (the float64
// division, one record per lane) stably in the chunk's stage; a thread per cell adds them onto the pixel's sums in array order; at the
// end a thread per pixel forms the twelve values (mdes_unit's finish) and stores them.  Windows with escaped polarity values never
// come here (their units stay with the main launch's ordered paths).
// LDS: state [npixa * 20] u32 | cnt [kMcWaves][2 npixa] | seg [2 npixa + 1] | rt [2][128] | window scalars | tmp | stage [kMcChunk] f64
constexpr int kMcWaves = 16, kMcThreads = kMcWaves * kWave, kMcChunk = 4096, kMcGrid = 256;
static_assert(kMcGrid % kHotLists == 0, "every sublist is worked off by kMcGrid / kHotLists workgroups");
__host__ __device__ inline size_t mdes_coop_lds_bytes(int npixa) {
    return align16((size_t)npixa * kErgoStreamWords * 4) + (size_t)kMcWaves * 2 * npixa * 4 + align16((size_t)(2 * npixa + 1) * 4) + 256 * 4 + 128 + 64 +
           (size_t)kMcChunk * 8;
}
#ifdef EVREP_TU_MDES   // (compiled by the one translation unit that launches it)
static __global__ __launch_bounds__(kMcThreads) void k_mdes_coop(BinView bv, const int64_t *__restrict__ off, int H, int W, int nchunk, UnitCfg uc,
                                                             double scale, float *__restrict__ out) {
    extern __shared__ __align__(16) unsigned char smem[];
    using T = Ergo12Table;
    constexpr int C = 12, NWD = kErgoStreamWords;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const uint32_t l = blockIdx.x % kHotLists, capl = hot_sublist_cap(bv.hot_cap);
    const uint32_t nraw = (uint32_t)__builtin_amdgcn_readfirstlane((int)bv.hot[l * 16]);
    if (nraw == 0u) return;
    const uint32_t nitems = min(nraw, capl);
    const uint32_t *items = bv.hot + kHotHdrWords + (size_t)l * capl;
    const int npixa = (uc.span + uc.merge) * kChunkPx, ncella = 2 * npixa;
    uint32_t *words = reinterpret_cast<uint32_t *>(smem);
    uint32_t *cnt = reinterpret_cast<uint32_t *>(smem + align16((size_t)npixa * NWD * 4));                        // [kMcWaves][ncella]
    uint32_t *seg = cnt + kMcWaves * ncella;                                                                       // [ncella + 1]
    uint32_t *rt = reinterpret_cast<uint32_t *>(reinterpret_cast<unsigned char *>(seg) + align16((size_t)(ncella + 1) * 4));   // [2][128]
    struct WinScalars { double interval; unsigned long long pmask; int32_t tmin, hi1, hi2, hi3, lo4, lo5, lo6; };
    WinScalars *ws = reinterpret_cast<WinScalars *>(rt + 256);                                                    // (128 bytes reserved)
    uint32_t *tmp = reinterpret_cast<uint32_t *>(reinterpret_cast<unsigned char *>(ws) + 128);                     // [16]
    double *stage = reinterpret_cast<double *>(tmp + 16);
    // (the one-wave hot launch behind this one clears the list: it takes the exit tickets)
    for (uint32_t it = blockIdx.x / kHotLists; it < nitems; it += gridDim.x / kHotLists) {
        const int item = __builtin_amdgcn_readfirstlane((int)items[it]);
        if (item >= 0 && item % kHotCodes == kHotCoop) {
            const int uid = item / kHotCodes;
            int chunk, nch;
            const ChunkGeom g = unit_geom(H, W, nchunk, uc, chunk, nch, uid);
            const int b = g.b;
            const int64_t beg = off[b];
            const int64_t n_win = off[b + 1] - beg;
            const int NK = H * nchunk, klo = g.row * nchunk + chunk;
            const StreamRuns R = stream_runs(bv, b, n_win, NK, klo, klo + nch);   // (every wave reads the unit's run tables itself)
            const int nb = R.nb;
            const uint32_t nrec = nb > 0 ? R.nrec : 0u;
            {
                uint4 *z = reinterpret_cast<uint4 *>(words);
                const int nvec = (g.npix * NWD + 3) / 4;
                for (int v = tid; v < nvec; v += kMcThreads) z[v] = make_uint4(0u, 0u, 0u, 0u);
            }
            const uint32_t src0 = (uint32_t)beg + ((uint32_t)lane << bv.chunk_shift) + R.a0 - R.pre0;
            const uint32_t src1 = (uint32_t)beg + ((uint32_t)(kWave + lane) << bv.chunk_shift) + R.a1 - R.pre1;
            if (wv == 0) {   // threadIdx.x == lane here: the window's statistics and the per-channel classes, once per item
                if (nb > kBsChainBlocks) {
                    rt[lane] = R.pre0; rt[128 + lane] = src0;
                    if (nb > kWave) { rt[kWave + lane] = R.pre1; rt[128 + kWave + lane] = src1; }
                }
                const WindowMeta m = window_meta(bv, off, b);
                const MdesWindows mw = mdes_windows(n_win);
                uint32_t a0 = 0, a1 = 0, a2 = 0;
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    const int wi = T::kWin[c], f = T::kFunc[c];
                    uint32_t cls = 7u;
                    int field = 0;
                    if (f == EVREP_F_TIMESTAMP_POS || f == EVREP_F_COUNT_POS) { cls = 4u; field = 1; }
                    if (f == EVREP_F_TIMESTAMP_NEG || f == EVREP_F_COUNT_NEG) {
                        const bool has_neg = (m.neg_flags >> wi) & 1u;   // operations.py:59-61,78-80
                        cls = has_neg ? 1u : 2u;
                        field = has_neg ? 2 : 3;
                    }
                    if (n_win <= 0 || ((m.oob_flags >> (7 * field + wi)) & 1u)) cls = 0u;
                    a0 |= (cls & 1u) << c; a1 |= ((cls >> 1) & 1u) << c; a2 |= ((cls >> 2) & 1u) << c;
                }
                if (lane == 0) {
                    ws->interval = (double)((int64_t)m.tmax - (int64_t)m.tmin);
                    ws->pmask = (unsigned long long)a0 | ((unsigned long long)a1 << 16) | ((unsigned long long)a2 << 32);
                    ws->tmin = m.tmin; ws->hi1 = mw.hi[1]; ws->hi2 = mw.hi[2]; ws->hi3 = mw.hi[3]; ws->lo4 = mw.lo[4]; ws->lo5 = mw.lo[5]; ws->lo6 = mw.lo[6];
                }
            }
            __syncthreads();
            const double interval = ws->interval;
            const unsigned long long pmask = ws->pmask;
            const int32_t tmin = ws->tmin, hi1 = ws->hi1, hi2 = ws->hi2, hi3 = ws->hi3, lo4 = ws->lo4, lo5 = ws->lo5, lo6 = ws->lo6;
            const int c0 = g.c0;
            const Rec8 *__restrict__ s8 = reinterpret_cast<const Rec8 *>(bv.sorted);
            auto src_of = [&](uint32_t j) -> uint32_t {    // the address of record j of the unit in the block runs
                if (nb <= kBsChainBlocks) {
                    uint32_t sx = (uint32_t)__builtin_amdgcn_readlane((int)src0, 0);
                    uint32_t prev = sx;
                    for (int k = 1; k < nb; ++k) {
                        const uint32_t pk = (uint32_t)__builtin_amdgcn_readlane((int)R.pre0, k);
                        const uint32_t sk = (uint32_t)__builtin_amdgcn_readlane((int)src0, k);
                        sx += (j >= pk) ? sk - prev : 0u;
                        prev = sk;
                    }
                    return sx + j;
                }
                uint32_t lo = 0, hi = (uint32_t)nb;
#pragma unroll
                for (int step = 0; step < 7; ++step) {
                    const uint32_t mid = (lo + hi) >> 1;
                    const bool go = hi - lo > 1 && rt[mid] <= j;
                    if (hi - lo > 1) { if (go) lo = mid; else hi = mid; }
                }
                return rt[128 + lo] + j;
            };
            // a record's channels (mdes_unit's sweep); p in {-1, 0, 1}: escaped windows never come here
            auto hits_of = [&](const Rec8 &q) -> uint32_t {
                const int rank = (int)(q.y >> 11);
                const uint32_t p2 = (q.y >> 9) & 3u;
                uint32_t chans = ergo_chans_of(0);
                chans |= rank < hi1 ? ergo_chans_of(1) : (rank < hi2 ? ergo_chans_of(2) : (rank < hi3 ? ergo_chans_of(3) : 0u));
                chans |= rank >= lo4 ? ergo_chans_of(4) : 0u;
                chans |= rank >= lo5 ? ergo_chans_of(5) : 0u;
                chans |= rank >= lo6 ? ergo_chans_of(6) : 0u;
                return p2 == 3u ? 0u : (chans & (uint32_t)(pmask >> (16u * p2)));
            };
            const int nbits = 32 - __builtin_clz((unsigned)ncella - 1u);
            const uint32_t nchunks = (nrec + (uint32_t)kMcChunk - 1u) / (uint32_t)kMcChunk;
            uint32_t *mycnt = cnt + wv * ncella;
            volatile uint32_t *vcnt = mycnt;
            for (uint32_t ch = 0; ch < nchunks; ++ch) {
                const uint32_t lo = ch * (uint32_t)kMcChunk, hi = min(nrec, lo + (uint32_t)kMcChunk);
                const uint32_t piece = (((hi - lo + (uint32_t)kMcWaves - 1u) / (uint32_t)kMcWaves) + 63u) & ~63u;
                const uint32_t mlo = min(hi, lo + (uint32_t)wv * piece), mhi = min(hi, mlo + piece);
                for (int v = tid; v < kMcWaves * ncella; v += kMcThreads) cnt[v] = 0u;
                __syncthreads();
                constexpr int G = 4;
                for (uint32_t j0 = mlo; j0 < mhi; j0 += (uint32_t)(G * kWave)) {
                    Rec8 q[G];
#pragma unroll
                    for (int sl = 0; sl < G; ++sl) {
                        const uint32_t j = j0 + (uint32_t)(sl * kWave + lane);
                        q[sl] = make_uint2(0u, 0u);
                        if (j < mhi) q[sl] = s8[src_of(j)];
                    }
#pragma unroll
                    for (int sl = 0; sl < G; ++sl) {
                        if (j0 + (uint32_t)(sl * kWave + lane) < mhi) {
                            const Rec8 r = q[sl];
                            const uint32_t px = ((r.y & 511u) - (uint32_t)c0) & 511u;
                            const int p = (int)((r.y >> 9) & 3u) - 1;
                            const uint32_t tt = (uint32_t)((int64_t)(int32_t)r.x - (int64_t)tmin);
                            const uint32_t hits = hits_of(r);
                            uint32_t *wd = words + px * (uint32_t)NWD;
                            auto hit = [&](int c) -> bool { return (hits >> c) & 1u; };
                            if (hit(0)) atomicAdd(wd + (p > 0 ? 0 : (p < 0 ? 1 : 2)), 1u);
                            if (hit(3) && p != 0) atomicAdd(wd + (p > 0 ? 4 : 5), 1u);
                            if (hit(5)) atomicAdd(wd + 3, 1u);
                            const uint32_t bits = hits & ((1u << 2) | (1u << 4) | (1u << 7) | (1u << 11) | (1u << 8) | (1u << 9) | (1u << 10));
                            atomicOr(wd + 6, bits | 0x80000000u);
                            if (hit(8)) atomicMax(wd + 7, tt);
                            if (hit(9)) atomicMax(wd + 8, tt);
                            if (hit(10)) atomicMax(wd + 9, tt);
                            if (hit(6)) atomicAdd(&mycnt[2u * px], 1u);
                            else if (hit(1)) atomicAdd(&mycnt[2u * px + 1u], 1u);
                        }
                    }
                }
                __syncthreads();
                {   // cell starts of the chunk; cursors of every (wave, cell)
                    const int cpt = (ncella + kMcThreads - 1) / kMcThreads;
                    const int k0 = tid * cpt;
                    uint32_t local = 0;
                    for (int k = 0; k < cpt; ++k)
                        if (k0 + k < ncella) for (int w2 = 0; w2 < kMcWaves; ++w2) local += cnt[w2 * ncella + k0 + k];
                    uint32_t total;
                    uint32_t run = block_exclusive_scan<kMcWaves>(local, tmp, &total);
                    for (int k = 0; k < cpt; ++k) {
                        if (k0 + k < ncella) {
                            seg[k0 + k] = run;
                            for (int w2 = 0; w2 < kMcWaves; ++w2) { const uint32_t c = cnt[w2 * ncella + k0 + k]; cnt[w2 * ncella + k0 + k] = run; run += c; }
                        }
                    }
                    if (tid == 0) seg[ncella] = total;
                }
                __syncthreads();
                for (uint32_t j0 = mlo; j0 < mhi; j0 += (uint32_t)(G * kWave)) {
                    Rec8 q[G];
#pragma unroll
                    for (int sl = 0; sl < G; ++sl) {
                        const uint32_t j = j0 + (uint32_t)(sl * kWave + lane);
                        q[sl] = make_uint2(0u, 0u);
                        if (j < mhi) q[sl] = s8[src_of(j)];
                    }
#pragma unroll
                    for (int sl = 0; sl < G; ++sl) {
                        if (j0 + (uint32_t)(sl * kWave) >= mhi) break;   // uniform
                        const bool have = j0 + (uint32_t)(sl * kWave + lane) < mhi;
                        const Rec8 r = q[sl];
                        const uint32_t hits = have ? hits_of(r) : 0u;
                        const bool h6 = (hits >> 6) & 1u, h1 = (hits >> 1) & 1u;
                        const bool valid = h6 || h1;
                        if (!__any(valid)) continue;
                        const uint32_t px = ((r.y & 511u) - (uint32_t)c0) & 511u;
                        const uint32_t cell = valid ? 2u * px + (h6 ? 0u : 1u) : 0u;
                        const uint32_t tt = (uint32_t)((int64_t)(int32_t)r.x - (int64_t)tmin);
                        const double ts = (double)tt / interval;   // the digest: one division per kept record and lane (mixed_density_event_stack.py:112-114)
                        uint32_t rk; bool last;
                        wave_match(cell, nbits, valid, lane, rk, last);
                        uint32_t pos = 0;
                        if (valid) { pos = vcnt[cell] + rk; stage[pos] = ts; }
                        __builtin_amdgcn_wave_barrier();
                        if (valid && last) vcnt[cell] = pos + 1;
                        __builtin_amdgcn_wave_barrier();
                    }
                }
                __syncthreads();
                // fold: a thread per cell, the cell's terms in array order onto the pixel's sums
                for (int cell = tid; cell < 2 * g.npix; cell += kMcThreads) {
                    const uint32_t st = seg[cell], en = seg[cell + 1];
                    if (en > st) {
                        uint32_t *wd = words + (uint32_t)(cell >> 1) * (uint32_t)NWD;
                        double *dd = reinterpret_cast<double *>(wd);
                        if (!(cell & 1)) {
                            double run = dd[6];
                            for (uint32_t j = st; j < en; ++j) run = run + stage[j];
                            dd[6] = run;
                            wd[10] += en - st;
                        } else {
                            double run = dd[7], sq = dd[8];
                            for (uint32_t j = st; j < en; ++j) { const double v = stage[j]; run = run + v; const double vv = v * v; sq = sq + vv; }
                            dd[7] = run; dd[8] = sq;
                            wd[11] += en - st;
                        }
                    }
                }
                __syncthreads();
            }
            // a thread per pixel: the twelve values (k_mdes_stream's emit = mdes_unit's finish), straight to the tensor
            float *dst = out + (((size_t)b * H + g.row) * (size_t)W + g.c0) * C;
            const bool vec = (reinterpret_cast<uintptr_t>(dst) & 15u) == 0;
            for (int px = tid; px < g.npix; px += kMcThreads) {
                const uint32_t *wd = words + (uint32_t)px * (uint32_t)NWD;
                const double *dd = reinterpret_cast<const double *>(wd);
                const uint32_t fl = wd[6];
                double r[C];
#pragma unroll
                for (int c = 0; c < C; ++c) r[c] = 0.0;
                if (fl >> 31) {
                    const int np0 = (int)wd[0], nn0 = (int)wd[1], nz0 = (int)wd[2];
                    const int n0 = np0 + nn0 + nz0;
                    const double s0 = (double)(np0 - nn0), q0 = (double)(np0 + nn0);
                    {
                        const double d = (double)(n0 < 1 ? 1 : n0);
                        const double mean = n0 > 1 ? s0 / d : s0, mean2 = n0 > 1 ? q0 / d : q0;
                        const double mm = mean * mean;
                        r[0] = mean2 - mm;
                    }
                    {
                        const int n1 = (int)wd[11];
                        const double d = (double)(n1 < 1 ? 1 : n1);
                        const double mean = n1 > 1 ? dd[7] / d : dd[7], mean2 = n1 > 1 ? dd[8] / d : dd[8];
                        const double mm = mean * mean;
                        r[1] = mean2 - mm;
                    }
                    r[2] = (fl >> 2) & 1u ? 1.0 : 0.0;
                    r[3] = (double)((int)wd[4] - (int)wd[5]);
                    r[4] = (fl >> 4) & 1u ? 1.0 : 0.0;
                    r[5] = (double)(int)wd[3];
                    { const int n6 = (int)wd[10]; r[6] = n6 > 1 ? dd[6] / (double)n6 : dd[6]; }
                    r[7] = (fl >> 7) & 1u ? 1.0 : 0.0;
                    r[8] = (fl >> 8) & 1u ? (double)wd[7] / interval : 0.0;
                    r[9] = (fl >> 9) & 1u ? (double)wd[8] / interval : 0.0;
                    r[10] = (fl >> 10) & 1u ? (double)wd[9] / interval : 0.0;
                    r[11] = (fl >> 11) & 1u ? 1.0 : 0.0;
                    if (scale != 1.0) {
#pragma unroll
                        for (int c = 0; c < C; ++c) r[c] = r[c] * scale;
                    }
                }
                float vals[C];
#pragma unroll
                for (int c = 0; c < C; ++c) vals[c] = (float)r[c];
                store_pixel<float, C>(dst + (size_t)px * C, vals, C, vec);
            }
        }
        __syncthreads();
    }
}
#endif

// "SBT" stacking (mixed_density_event_stack.py:76-107): eight windows cut by the normalised time t_s = (t - tmin) / (tmax -
// tmin) -- w0 all, w1..w3 i/3 <= t_s <= (i+1)/3 (both ends inclusive), w4..w7 t_s <= 1/2, 1/4, 1/8, 1/16.  On ascending
// timestamps (which every builder here requires) a window is a rank range: lo = #(t_s < a), hi = #(t_s <= b), counted with the
// reference's own float64 comparisons (a NaN t_s -- a window of one timestamp -- fails them all: empty windows, as there).
// grid (B), 1024 threads: bounds [B][8][2], wflags [B][2] = {bit w: window w holds a p == -1 event; bit 8 k + w: window w
// holds an out-of-frame event of polarity class k (0 any, 1 p == 1, 2 p == -1, 3 p == 0)}.
#ifdef EVREP_TU_MDES   // (compiled by the one translation unit that launches it)
static __global__ __launch_bounds__(1024) void k_mdes_sbt_windows(const int4 *__restrict__ ev, const int64_t *__restrict__ off, int H, int W,
                                                          int32_t *__restrict__ bounds, uint32_t *__restrict__ wflags) {
    __shared__ int cnt[10];
    __shared__ uint32_t fl[2];
    __shared__ int lo_s[8], hi_s[8];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int64_t beg = off[b];
    const int n = (int)(off[b + 1] - beg);
    const int4 *e = ev + beg;
    if (tid < 10) cnt[tid] = 0;
    if (tid < 2) fl[tid] = 0u;
    __syncthreads();
    if (n > 0) {
        const int64_t tmin = e[0].z, tmax = e[n - 1].z;
        const double interval = (double)(tmax - tmin);
        const double ef = 1.0 / 3.0;
        int c[10];
#pragma unroll
        for (int k = 0; k < 10; ++k) c[k] = 0;
        for (int i = tid; i < n; i += 1024) {
            const double ts = (double)((int64_t)e[i].z - tmin) / interval;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                c[k] += ts < (double)k * ef ? 1 : 0;            // the events in front of window 1 + k
                c[3 + k] += ts <= (double)(k + 1) * ef ? 1 : 0;   // ... up to its end
            }
            double factor = 1.0;
#pragma unroll
            for (int k = 0; k < 4; ++k) { factor = factor / 2.0; c[6 + k] += ts <= factor ? 1 : 0; }
        }
#pragma unroll
        for (int k = 0; k < 10; ++k) {
            int v = c[k];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
            if ((tid & 63) == 0 && v) atomicAdd(&cnt[k], v);
        }
    }
    __syncthreads();
    if (tid < 8) {
        int l = 0, h = 0;
        if (tid == 0) h = n;
        else if (tid < 4) { l = cnt[tid - 1]; h = cnt[3 + tid - 1]; if (h < l) h = l; }
        else h = cnt[6 + tid - 4];
        lo_s[tid] = l; hi_s[tid] = h;
        bounds[(size_t)b * 16 + 2 * tid] = l;
        bounds[(size_t)b * 16 + 2 * tid + 1] = h;
    }
    __syncthreads();
    uint32_t neg = 0u, oob = 0u;
    const int64_t HW = (int64_t)H * W;
    for (int i = tid; i < n; i += 1024) {
        const int4 r = e[i];
        uint32_t memb = 0u;
#pragma unroll
        for (int k = 0; k < 8; ++k) memb |= (i >= lo_s[k] && i < hi_s[k]) ? (1u << k) : 0u;
        if (r.w == -1) neg |= memb;
        // in-frame test on the flat index x + y * W, as the reference's scatter sees it (k_block_keysort)
        bool valid = (uint32_t)r.x < (uint32_t)W && (uint32_t)r.y < (uint32_t)H;
        if (!valid) { const int64_t key = (int64_t)r.x + (int64_t)r.y * W; valid = key >= 0 && key < HW; }
        if (!valid) {
            const int cls = r.w == 1 ? 1 : (r.w == -1 ? 2 : (r.w == 0 ? 3 : 0));
            oob |= memb | (cls ? (memb << (8 * cls)) : 0u);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { neg |= __shfl_xor(neg, o, 64); oob |= __shfl_xor(oob, o, 64); }
    if ((tid & 63) == 0) { if (neg) atomicOr(&fl[0], neg); if (oob) atomicOr(&fl[1], oob); }
    __syncthreads();
    if (tid < 2) wflags[2 * b + tid] = fl[tid];
}
#endif

// --------------------------------------------------------------------------------------------
// A6: EventStack.pre_stack / post_stack (event_stack.py:15-131), last_timestamp = t[-1]
// --------------------------------------------------------------------------------------------
template <int CM, bool HOT = false>  // compile-time channel capacity (8, 12 or 16)
__global__ __launch_bounds__(kWave, HOT ? 4 : EVREP_ES_WAVES) void k_event_stack(BinView bv,
                                                      const int64_t *__restrict__ off, int H, int W, int nchunk, UnitCfg uc,
                                                      int S, int premap, float scale, float *__restrict__ out) {
    extern __shared__ __align__(16) unsigned char smem[];
    run_units<HOT>(bv, [&](int uid, int part) {
        WaveLds<float, HOT> w(smem, S, (uc.span + uc.merge) * kChunkPx, uc.stage, uc.partpx);
        w.arm(uc.hold);
        ChunkGeom g;
        const UnitRecs u = unit_front<float, HOT, true>(bv, off, H, W, nchunk, uc, w, g, uid, part);
        if (u.deferred) return;
        float *dst = out + (((size_t)g.b * H + g.row) * (size_t)W + g.c0) * S;
        const int64_t n_win = off[g.b + 1] - off[g.b];
        // level k keeps events[off_k:], off_k = sum_{j=1..k} N // 2^j  (event_stack.py:70-82)
        int offk[CM];
        {
            int cur = (int)n_win, o = 0;
#pragma unroll
            for (int k = 0; k < CM; ++k) { offk[k] = o; cur /= 2; o += cur; }
        }
        auto reduce = [&](uint32_t jb, uint32_t je, auto get, float(&vals)[CM]) {
            const Rec e = get(je - 1);  // ndarray.put is last-write-wins (event_stack.py:125)
            int p = e.w;
            if (premap == 1) p = (p + 1) >> 1;                   // (p + 1) // 2   (gen1_transforms.py:34)
            // 2*p - 1 as int8 (event_stack.py:18); premap 2: the column already holds that int8 value (the host
            // formed it, negated for the reversed "future" half, event_stack.py:35)
            const float v = (float)(int8_t)(premap == 2 ? p : 2 * p - 1) * scale;
#pragma unroll
            for (int l = 0; l < CM; ++l) vals[l] = (e.y >= offk[l]) ? v : 0.0f;
        };
        emit_chunk<float, CM, HOT, false>(u, g.row * W + g.c0, g.npix, S, dst, w, (const float *)nullptr, reduce);
    });
}

// --------------------------------------------------------------------------------------------
// A7: ToTimesurface (time_surface.py:25-74) driven as gen1_transforms.py:69-87
// --------------------------------------------------------------------------------------------
constexpr int kMaxSlices = 8;

// exp(x) for the time surface's argument range (x = (t_last - t_cut)/tau - ... in [-64, 0]):
// n = rint(x*log2 e), r = x - n*ln2 (two-part ln2), degree-12 Taylor polynomial in r (|r| <= 0.347,
// truncation 1e-16), exponent patched in.  ~20 float64 instructions instead of the ~110 of the
// general libm path, relative error < 4e-16 on this range (the parity budget is 1e-5).
__device__ inline double exp_neg_range(double x) {
    if (x < -700.0) return 0.0;
    const double n = rint(x * 1.4426950408889634);
    double r = fma(n, -6.93147180369123816490e-01, x);
    r = fma(n, -1.90821492927058770002e-10, r);
    double p = 2.08767569878680989792e-09;             // 1/12!
    p = fma(p, r, 2.50521083854417187751e-08);         // 1/11!
    p = fma(p, r, 2.75573192239858906526e-07);         // 1/10!
    p = fma(p, r, 2.75573192239858906526e-06);         // 1/9!
    p = fma(p, r, 2.48015873015873015873e-05);         // 1/8!
    p = fma(p, r, 1.98412698412698412698e-04);         // 1/7!
    p = fma(p, r, 1.38888888888888888889e-03);         // 1/6!
    p = fma(p, r, 8.33333333333333333333e-03);         // 1/5!
    p = fma(p, r, 4.16666666666666666667e-02);         // 1/4!
    p = fma(p, r, 1.66666666666666666667e-01);         // 1/3!
    p = fma(p, r, 5.00000000000000000000e-01);         // 1/2!
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    return ldexp(p, (int)n);
}

struct TsCuts {
    int32_t idx[kMaxSlices];   // searchsorted(t_norm, s+1, 'left')  (or the caller's indices)
    int32_t tcut[kMaxSlices];  // t[idx[s]]
    int32_t live[kMaxSlices];  // 1 iff the sequential scan reaches this slice (strictly increasing idx)
    int32_t tref;              // reference time of the factorised exponentials: the last live cut's timestamp
    int32_t direct;            // 1: the window spans more than 600 tau -- exponentials are taken per slice, unfactorised
    int32_t pad[6];
    double bg[2 * kMaxSlices];  // value of an untouched (pixel, polarity) entry of slice s, already scaled
    double fac[kMaxSlices];    // exp((tref - tcut[s]) / tau) * scale
    double tcutf[kMaxSlices];  // the cut's timestamp as float64: t[idx[s]], or tf[idx[s]] (float timestamps, evrep_time_surface_ftime)
};
static_assert(sizeof(TsCuts) == kTsCutsBytes, "TsCuts");

// grid (B), 64 threads.  indices == nullptr: the dispatcher's searchsorted cuts; otherwise DEVICE
// int32 [B, S] event indices as ToTimesurface.__call__(events, indices) receives them.
#ifdef EVREP_TU_BUILDERS   // (compiled by the one translation unit that launches it)
static __global__ void k_ts_cuts(const int4 *__restrict__ ev, const int64_t *__restrict__ off, int S,
                          const int32_t *__restrict__ indices, double tau, double scale, TsCuts *__restrict__ cuts,
                          const double *__restrict__ tf) {
    const int b = blockIdx.x, s = threadIdx.x;
    __shared__ int sidx[kMaxSlices];
    const int64_t beg = off[b];
    const int64_t n = off[b + 1] - beg;
    if (s < S) {
        int idx = 0, tc = 0;
        if (n > 0) {
            if (indices) {
                idx = indices[b * S + s];
            } else {
                const int32_t t0 = ev[beg].z, tl = ev[beg + n - 1].z;
                const double den = (double)(int32_t)(tl - t0);
                const double target = (double)(s + 1);
                int64_t lo = 0, hi = n;
                while (lo < hi) {
                    const int64_t mid = lo + (hi - lo) / 2;
                    // t_norm = (t - t[0]) / (t[-1] - t[0]) * 6   (gen1_transforms.py:80)
                    const double q = (double)(int32_t)(ev[beg + mid].z - t0) / den;
                    const double tn = q * (double)S;
                    if (tn < target) lo = mid + 1; else hi = mid;
                }
                idx = (int)lo;
            }
            tc = (idx >= 0 && idx < n) ? ev[beg + idx].z : 0;
        }
        sidx[s] = idx;
        cuts[b].idx[s] = idx;
        cuts[b].tcut[s] = tc;
        cuts[b].tcutf[s] = (tf && n > 0 && idx >= 0 && idx < n) ? tf[beg + idx] : (double)tc;
    }
    __syncthreads();
    if (s == 0) {
        bool alive = n > 0;
        for (int k = 0; k < kMaxSlices; ++k) {
            // `if index == indices[pos]` fires once per event: a repeated / decreasing idx is never reached
            if (k < S) {
                if (k > 0 && sidx[k] <= sidx[k - 1]) alive = false;
                if (sidx[k] < 0 || sidx[k] >= n) alive = false;
                cuts[b].live[k] = alive ? 1 : 0;
            } else {
                cuts[b].live[k] = 0;
                cuts[b].idx[k] = 0;
                cuts[b].tcut[k] = 0;
            }
            // untouched pixels are not zero: exp((-(3 tau + 1) - t_i) / tau)  (time_surface.py:26-29,68-72);
            // slices the scan never reaches stay exactly 0
            double v = 0.0;
            if (k < S && alive) v = exp_neg_range((-(tau * 3.0 + 1.0) - cuts[b].tcutf[k]) * (1.0 / tau)) * scale;
            cuts[b].bg[2 * k] = v;
            cuts[b].bg[2 * k + 1] = v;
        }
        // exp((m - t_s) / tau) = exp((m - tref) / tau) * exp((tref - t_s) / tau): the first factor depends on the event
        // only (one exponential per EVENT, formed by the builder's digest), the second on the slice only (here)
        int tref = 0;
        for (int k = 0; k < kMaxSlices; ++k) if (k < S && cuts[b].live[k]) tref = cuts[b].tcut[k];  // cut times ascend
        int direct = 0;
        for (int k = 0; k < kMaxSlices; ++k) {
            double f = 0.0;
            if (k < S && cuts[b].live[k]) {
                const double x = ((double)tref - (double)cuts[b].tcut[k]) * (1.0 / tau);
                if (x > 600.0) direct = 1;
                f = exp(x) * scale;
            }
            cuts[b].fac[k] = f;
        }
        cuts[b].tref = tref;
        cuts[b].direct = tf ? 1 : direct;   // float timestamps: exponentials per slice
    }
}
#endif

// CM = compile-time channel capacity, 2 * slices <= CM (12 or 16); FACT = the factorised exponentials are compiled in
// (launches on sparse windows; dense windows run the leaner per-slice form)
// (waves per SIMD asked of the compiler: 4 = 128 VGPRs, no scratch -- r05b: at 5 the float64 instances carried 44-108 bytes of scratch,
//  and a launch of tens of thousands of one-wave workgroups pays for scratch per wave: Gen1 88.7 -> 78.9 us, config 3 135.8 -> 129.3)
#ifndef EVREP_TS_WAVES
#define EVREP_TS_WAVES 4
#endif
template <typename OutT, int CM, bool FACT, bool HOT = false>
__global__ __launch_bounds__(kWave, HOT ? 4 : EVREP_TS_WAVES) void k_time_surface(BinView bv, const int64_t *__restrict__ off,
                                                       const TsCuts *__restrict__ cuts, int H, int W, int nchunk, UnitCfg uc,
                                                       int S, double tau, int premap, double scale, const double *__restrict__ tf,
                                                       OutT *__restrict__ out) {
    extern __shared__ __align__(16) unsigned char smem[];
    run_units<HOT>(bv, [&](int uid, int part) {
        const int C = 2 * S;
        WaveLds<OutT, HOT> w(smem, C, (uc.span + uc.merge) * kChunkPx, uc.stage);
        w.arm(uc.hold);
        // the window's cuts, held in registers with compile-time indices only (no scratch); read BEFORE the unit's front end, so
        // that their latency runs beside the run-table / record loads instead of behind them (r03)
        int chunk0;
        const TsCuts *cp = cuts + unit_geom(H, W, nchunk, uc, chunk0, uid).b;
        struct { int idx[(CM / 2)], tcut[(CM / 2)], live[(CM / 2)]; } cu;
#pragma unroll
        for (int q = 0; q < (CM / 2); ++q) { cu.idx[q] = cp->idx[q]; cu.tcut[q] = cp->tcut[q]; cu.live[q] = cp->live[q]; }
        ChunkGeom g;
        // a unit beyond the record stage (r04): per (pixel, polarity class, slice) the rank + 1 of the last event at or before the
        // slice's cut -- all a slice reads of its pixel -- kept by LDS atomicMax in one sweep over the unit's records
        const int4 *evw = bv.ev + off[unit_geom(H, W, nchunk, uc, chunk0, uid).b];
        // word = (rank + 1) << 9 | id: the record's timestamp waits in the record stage under its sweep id (ids beyond the
        // stage -- a hot unit of a clustered window -- are marked 511: the timestamp is then fetched from the caller's events)
        uint32_t *vw = reinterpret_cast<uint32_t *>(w.tile);
        int32_t *vt = reinterpret_cast<int32_t *>(w.evbuf);
        const uint32_t vcap = min((uint32_t)uc.stage * 4u, 511u);
        auto visit = unit_visit([&](uint32_t px, const Rec8 &r, uint32_t id) {
            const uint32_t rank = r.y >> 11, p2 = (r.y >> 9) & 3u;
            int p = p2 == 3u ? evw[rank].w : (int)p2 - 1;
            if (premap & 1) p = (int)(int8_t)(int)((double)(p + 1) / 2.0);
            uint32_t *at = vw + (px * 2u + (uint32_t)(p & 1)) * (uint32_t)S;
            const uint32_t word = ((rank + 1u) << 9) | min(id, 511u);
            if (id < vcap) vt[id] = (int32_t)r.x;
#pragma unroll
            for (int q = 0; q < (CM / 2); ++q)
                if (q < S && (int)rank <= cu.idx[q]) atomicMax(at + q, id < vcap ? word : (word | 511u));
        }, 2 * S);
        const UnitRecs u = unit_front<OutT, HOT, false>(bv, off, H, W, nchunk, uc, w, g, uid, part, visit);
        if (u.deferred) return;
        OutT *dst = out + (((size_t)g.b * H + g.row) * (size_t)W + g.c0) * C;
        // (m - t_i) / tau is evaluated as (m - t_i) * (1/tau): one rounding of 1/tau instead of a float64
        // division per exponential; the surface moves by < 1e-15 relative (budget 1e-5)
        const double inv_tau = 1.0 / tau;
        // exp((m - t_s)/tau) = exp((m - tref)/tau) * fac[s]: ONE exponential per event, formed by the digest -- one record per
        // lane, all lanes at once -- instead of two per slice and touched pixel inside the per-pixel code (12 per pixel for the
        // reference's 6 slices).  The digest keeps {E lo, rank, E hi, p}.  This factorised form is used by the waves whose
        // whole unit is staged (every unit of a sparse window) in windows of up to 600 tau (beyond, the factors would
        // overflow); the other waves keep the timestamps and take their exponentials per slice, as round 1 did.
        const int tref = cp->tref;
        const double *tw = tf ? tf + off[unit_geom(H, W, nchunk, uc, chunk0, uid).b] : nullptr;   // float timestamps, by rank
        const bool fact = FACT && !tf && !(premap & 2) && cp->direct == 0 && (u.ce - u.cs) <= max((uint32_t)kWave, (uint32_t)u.nstaged);  // wave-uniform
        double fac[(CM / 2)];
#pragma unroll
        for (int q = 0; q < (CM / 2); ++q) fac[q] = cp->fac[q];
        // the background of every slice was computed once per window by k_ts_cuts
        if ((int)threadIdx.x < CM) w.bg[threadIdx.x] = (OutT)cp->bg[threadIdx.x];
        wave_phase();
        const OutT *bg = w.bg;
        auto digest = [&](const Rec &r) -> Rec {
            if (!fact) return r;
            const double E = exp_neg_range(((double)r.z - (double)tref) * inv_tau);
            return make_int4(__double2loint(E), r.y, __double2hiint(E), r.w);
        };
        // pass 2 of a pixel: the slices' values from the timestamp memory each cut saw (snap: INT32_MIN = never written)
        auto pass2 = [&](const int(&snap0)[(CM / 2)], const int(&snap1)[(CM / 2)], auto get, OutT(&vals)[CM]) {
#pragma unroll
            for (int q = 0; q < (CM / 2); ++q) {
                OutT v0 = bg[2 * q], v1 = bg[2 * q + 1];
                if (q < S && cu.live[q]) {
                    if (fact) {  // the snapshot records' exponentials times the slice factor
                        if (snap0[q] != INT32_MIN) { const Rec e = get((uint32_t)snap0[q]); v0 = (OutT)(__hiloint2double(e.z, e.x) * fac[q]); }
                        if (snap1[q] != INT32_MIN) { const Rec e = get((uint32_t)snap1[q]); v1 = (OutT)(__hiloint2double(e.z, e.x) * fac[q]); }
                    } else {     // one straight-line batch of exponentials, the same for every lane of the wave
                        const double tc = tw ? gload_f64(&cp->tcutf[q]) : (double)cu.tcut[q];
                        if (__any(snap0[q] != INT32_MIN)) {
                            double m0 = (double)snap0[q];
                            if (tw) m0 = snap0[q] != INT32_MIN ? gload_f64(tw + snap0[q]) : 0.0;   // the caller's float64 time of that event
                            const double e0 = exp_neg_range((m0 - tc) * inv_tau) * scale;
                            if (snap0[q] != INT32_MIN) v0 = (OutT)e0;
                        }
                        if (__any(snap1[q] != INT32_MIN)) {
                            double m1 = (double)snap1[q];
                            if (tw) m1 = snap1[q] != INT32_MIN ? gload_f64(tw + snap1[q]) : 0.0;
                            const double e1 = exp_neg_range((m1 - tc) * inv_tau) * scale;
                            if (snap1[q] != INT32_MIN) v1 = (OutT)e1;
                        }
                    }
                }
                vals[2 * q] = v0;
                vals[2 * q + 1] = v1;
            }
        };
        auto reduce = [&](uint32_t jb, uint32_t je, auto get, OutT(&vals)[CM]) {
            // pass 1 (integers only): the timestamp memory of this pixel as each cut sees it, per polarity -- the record
            // index of the last event (factorised form) or its timestamp.  INT32_MIN = never written.  Slices cut strictly
            // before an event see the memory as it stands before it.
            int snap0[(CM / 2)], snap1[(CM / 2)];
#pragma unroll
            for (int q = 0; q < (CM / 2); ++q) { snap0[q] = INT32_MIN; snap1[q] = INT32_MIN; }
            int cur0 = INT32_MIN, cur1 = INT32_MIN;
            uint32_t done = 0;  // bit q: slice q has taken its snapshot (all indexing stays compile-time: no scratch)
            for (uint32_t j = jb; j <= je; ++j) {
                int rank = INT32_MAX, t = 0, p = 0;
                if (j < je) { const Rec e = get(j); rank = e.y; t = fact ? (int)j : (tw ? e.y : e.z); p = e.w; }   // float timestamps: the memory holds the event's RANK
#pragma unroll
                for (int q = 0; q < (CM / 2); ++q) {
                    if (q < S && !((done >> q) & 1u) && cu.idx[q] < rank) { snap0[q] = cur0; snap1[q] = cur1; done |= 1u << q; }
                }
                if (j < je) {
                    if (premap & 1) p = (int)(int8_t)(int)((double)(p + 1) / 2.0);  // ((p+1)/2).astype(int8)
                    if (p & 1) cur1 = t; else cur0 = t;
                }
            }
            pass2(snap0, snap1, get, vals);
        };
        if (u.part == -4) {   // wave-uniform: the unit was visited, not ordered: one lane per pixel, straight from the visit words
            constexpr int V = 16 / (int)sizeof(OutT);
            const bool vec = (C % V) == 0 && (reinterpret_cast<uintptr_t>(dst) & 15u) == 0;
            w.pace();
            for (int pt = 0; pt * kWave < g.npix; ++pt) {
                const int np = min(kWave, g.npix - pt * kWave);
                const uint32_t px = (uint32_t)(pt * kWave + (int)threadIdx.x);
                int snap0[(CM / 2)], snap1[(CM / 2)];
#pragma unroll
                for (int q = 0; q < (CM / 2); ++q) {
                    snap0[q] = INT32_MIN; snap1[q] = INT32_MIN;
                    if (q < S && (int)threadIdx.x < np) {
                        const uint32_t w0 = vw[(px * 2u) * (uint32_t)S + (uint32_t)q], w1 = vw[(px * 2u + 1u) * (uint32_t)S + (uint32_t)q];
                        // the memory holds the event's timestamp (its RANK with float timestamps, as in reduce)
                        if (w0) snap0[q] = tw ? (int)(w0 >> 9) - 1 : ((w0 & 511u) != 511u ? vt[w0 & 511u] : evw[(w0 >> 9) - 1u].z);
                        if (w1) snap1[q] = tw ? (int)(w1 >> 9) - 1 : ((w1 & 511u) != 511u ? vt[w1 & 511u] : evw[(w1 >> 9) - 1u].z);
                    }
                }
                OutT vals[CM];
                pass2(snap0, snap1, [](uint32_t) -> Rec { return make_int4(0, 0, 0, 0); }, vals);
                if ((int)threadIdx.x < np) store_pixel<OutT, CM>(dst + ((size_t)pt * kWave + threadIdx.x) * C, vals, C, vec);
            }
            w.mark(5);
            return;
        }
        emit_chunk<OutT, CM, HOT>(u, digest, [](const Rec &r) -> Rec { return r; }, g.row * W + g.c0, g.npix, C, dst, w, bg, reduce);
    });
}

// --------------------------------------------------------------------------------------------
// A7 (r06), after the key-sorted pass: the time surface as a STREAM.  Slice s of a (pixel, polarity) entry reads ONE event: the last
// one in array order at or before the slice's cut.  Every record goes, by ONE 64-bit LDS atomicMax, into the word of the FIRST live
// slice whose cut lies at or behind it (the live cuts ascend strictly); a running maximum over the slices then hands every slice the
// last event it saw.  What the word holds is a KEY that grows with the event's place in the window:
//   * ascending integer timestamps, a window of up to 600 tau (every window the dispatcher hands over): the bits of
//     E = exp((t - tref) / tau) -- ONE exponential per EVENT, one record per lane, all lanes at once -- and a slice's value is
//     E * fac[s], k_ts_cuts' exp((tref - t_cut) / tau) * scale: k_time_surface's factorised form, now for units of every size
//     (the ordered builder takes it for fully staged units only: twelve exponentials per touched pixel are what bound it on dense
//     windows -- 8 x 500 000 events 124 us -- and the two forms differ by an ulp or two, far inside the 1e-5 budget);
//   * else (a longer window, timestamps that are not ascending): the event's rank; its time is gathered by the rank from the
//     caller's events and the exponential of (t_event - t_cut) / tau taken per slice and polarity for 64 pixels at a time.
// No grouping, no stage, no walk, no hot launch, any unit size.  (Float64 timestamps and the caller's array-order flag stay with
// k_time_surface: the host decides.)
// LDS: words [npixa * 2 * S] u64, overlaid by the tile [npixa * 2 S] OutT | head | srcs
__host__ __device__ inline size_t time_surface_stream_lds_bytes(int S, int npixa, size_t elem, int rb) {
    (void)elem;   // (the words are as wide as a float64 value: the tile always fits their room)
    return align16((size_t)npixa * 2 * S * 8) + (size_t)(64 * rb) * 4 + 128 * 4;
}
template <typename OutT, int CM, int RB>
__global__ __launch_bounds__(kWave, 4) void k_time_surface_stream(BinView bv, const int64_t *__restrict__ off, const TsCuts *__restrict__ cuts,
                                                              int H, int W, int nchunk, UnitCfg uc, int S, double tau, int premap,
                                                              double scale, OutT *__restrict__ out) {
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr int SM = CM / 2;
    const int lane = threadIdx.x;
    const int C = 2 * S;
    const int uid = chunk_unit((int)(gridDim.x * gridDim.y * gridDim.z));
    int chunk, nch;
    const ChunkGeom g = unit_geom(H, W, nchunk, uc, chunk, nch, uid);
    const int b = g.b;
    const int64_t beg = off[b];
    const int64_t n_win = off[b + 1] - beg;
    const TsCuts *cp = cuts + b;
    int idx[SM], live[SM];
#pragma unroll
    for (int q = 0; q < SM; ++q) { idx[q] = cp->idx[q]; live[q] = cp->live[q]; }
    const int tref = cp->tref;
    const MetaRaw mraw = meta_prefetch(bv, b);
    const int npixa = (uc.span + uc.merge) * kChunkPx;
    unsigned long long *words = reinterpret_cast<unsigned long long *>(smem);
    uint32_t *head = reinterpret_cast<uint32_t *>(smem + align16((size_t)npixa * C * 8));
    uint32_t *srcs = head + 64 * RB;
    {
        uint4 *z = reinterpret_cast<uint4 *>(words);
        const int nvec = (g.npix * C + 1) / 2;
        for (int v = lane; v < nvec; v += kWave) z[v] = make_uint4(0u, 0u, 0u, 0u);
    }
    int nlive = 0;   // the live slices are a prefix (k_ts_cuts)
#pragma unroll
    for (int q = 0; q < SM; ++q) nlive += (q < S && live[q]) ? 1 : 0;
    const WindowMeta m = meta_finish(bv, off, b, mraw);
    const bool byrank = cp->direct != 0 || (m.status & EVREP_ST_UNSORTED) != 0u;   // wave-uniform
    const double inv_tau = 1.0 / tau;
    const int4 *evw = bv.ev + beg;
    const int c0 = g.c0;
    wave_phase();
    stream_unit_records<RB>(bv, b, beg, n_win, H * nchunk, g.row * nchunk + chunk, g.row * nchunk + chunk + nch, head, srcs, StreamNoPre(),
        [&](bool have, const Rec8 &q8, const uint2 &) {
            const uint32_t px = have ? ((q8.y & 511u) - (uint32_t)c0) & 511u : 0u;
            const uint32_t rank = q8.y >> 11, p2 = (q8.y >> 9) & 3u;
            int p = (have && p2 == 3u) ? evw[rank].w : (int)p2 - 1;
            if (premap & 1) p = (int)(int8_t)(int)((double)(p + 1) / 2.0);   // ((p + 1) / 2).astype(int8)  (gen1_transforms.py:70-72)
            int q0 = 0;   // the live slices whose cut lies strictly before the event
#pragma unroll
            for (int q = 0; q < SM; ++q) q0 += (q < nlive && idx[q] < (int)rank) ? 1 : 0;
            unsigned long long key = (unsigned long long)rank + 1ull;
            if (!byrank) {   // one exponential per event: E > 0, and E grows with t, so its bits order as the events do
                const double E = exp_neg_range(((double)(int32_t)q8.x - (double)tref) * inv_tau);
                key = (unsigned long long)__double_as_longlong(E) + 1ull;   // (+ 1: an E that underflowed to +0 still marks the entry as touched)
            }
            if (have && q0 < nlive) atomicMax(words + (px * 2u + (uint32_t)(p & 1)) * (uint32_t)S + (uint32_t)q0, key);
        });
    wave_phase();
    OutT *tile = reinterpret_cast<OutT *>(smem);
    const int nbatch = (g.npix + kWave - 1) / kWave;
    for (int pt = 0; pt < nbatch; ++pt) {   // (a value is never wider than its word: a batch's values end in front of the next batch's words)
        const int px = pt * kWave + lane;
        const bool own = px < g.npix;
        unsigned long long last[2][SM];   // the last event each slice saw: a running maximum over the slices
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
            unsigned long long run = 0ull;
#pragma unroll
            for (int q = 0; q < SM; ++q) {
                if (q < S && own) { const unsigned long long wv = words[((uint32_t)px * 2u + (uint32_t)pl) * (uint32_t)S + (uint32_t)q]; run = wv > run ? wv : run; }
                last[pl][q] = (q < nlive) ? run : 0ull;
            }
        }
        OutT vals[CM];
        if (!byrank) {
#pragma unroll
            for (int q = 0; q < SM; ++q) {
                OutT v0 = (OutT)0, v1 = (OutT)0;
                if (q < S) {
                    const double fq = gload_f64(&cp->fac[q]);
                    v0 = last[0][q] ? (OutT)(__longlong_as_double((long long)(last[0][q] - 1ull)) * fq) : (OutT)gload_f64(&cp->bg[2 * q]);
                    v1 = last[1][q] ? (OutT)(__longlong_as_double((long long)(last[1][q] - 1ull)) * fq) : (OutT)gload_f64(&cp->bg[2 * q + 1]);
                }
                vals[2 * q] = v0;
                vals[2 * q + 1] = v1;
            }
        } else {
            double tm[2][SM];   // the events' times, by rank (every gather in flight together)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) {
#pragma unroll
                for (int q = 0; q < SM; ++q) {
                    tm[pl][q] = 0.0;
                    if (last[pl][q]) tm[pl][q] = (double)evw[(uint32_t)last[pl][q] - 1u].z;
                }
            }
#pragma unroll
            for (int q = 0; q < SM; ++q) {
                OutT v0 = (OutT)0, v1 = (OutT)0;
                if (q < S) {
                    v0 = (OutT)gload_f64(&cp->bg[2 * q]); v1 = (OutT)gload_f64(&cp->bg[2 * q + 1]);
                    if (q < nlive) {
                        const double tc = gload_f64(&cp->tcutf[q]);
                        if (__any(last[0][q] != 0ull)) {
                            const double e0 = exp_neg_range((tm[0][q] - tc) * inv_tau) * scale;
                            if (last[0][q]) v0 = (OutT)e0;
                        }
                        if (__any(last[1][q] != 0ull)) {
                            const double e1 = exp_neg_range((tm[1][q] - tc) * inv_tau) * scale;
                            if (last[1][q]) v1 = (OutT)e1;
                        }
                    }
                }
                vals[2 * q] = v0;
                vals[2 * q + 1] = v1;
            }
        }
        wave_phase();   // the batch's words are in registers: its values may take their place
        if (own) {
            OutT *mine = tile + (size_t)px * C;
#pragma unroll
            for (int c = 0; c < CM; ++c) if (c < C) mine[c] = vals[c];
        }
        wave_phase();
    }
    OutT *dst = out + (((size_t)b * H + g.row) * (size_t)W + g.c0) * C;
    tile_store(tile, g.npix * C, dst);
}

// --------------------------------------------------------------------------------------------
// A8: events2ToreFeature (tore.py:6-83), one sample time per window
// --------------------------------------------------------------------------------------------
constexpr int kMaxToreK = 8;

// grid (ceil(nchunk/span), H, B) over OUTPUT units / rows, 64 threads.
// sample_times == nullptr: T = ts[-1] (gen1_transforms.py:63); else DEVICE int32 [B].
template <int CM, bool HOT = false, bool SM = false>  // compile-time channel capacity, 2 * K <= CM (12 or 16); SM: the main launch sweeps itself (see k_polstats)
__global__ __launch_bounds__(kWave, HOT ? 4 : (CM <= 12 ? EVREP_TORE_WAVES : 1)) void k_tore(const int4 *__restrict__ ev, BinView bv, const int64_t *__restrict__ off,
                                               const int32_t *__restrict__ sample_times,
                                               const double *__restrict__ tf, const double *__restrict__ sample_times_f,
                                               int H, int W, int nchunk, UnitCfg uc, int K, int frame_mode, float scale,
                                               float *__restrict__ out) {
    extern __shared__ __align__(16) unsigned char smem[];
    run_units<HOT>(bv, [&](int uid, int part) {
        const int C = 2 * K;
        const int span = uc.span;
        WaveLds<float, HOT> w(smem, C, (span + 1) * kChunkPx, uc.stage);
        w.arm(uc.hold);
        const int nunit = uc.nunit;   // = (nchunk + span - 1) / span: TORE's units never merge (unit_cfg, extra_chunks)
        const int u = uid;
        const uint32_t urow = fastdiv((uint32_t)u, uc.nunit_m, uc.nunit_sh);
        const int b = (int)fastdiv(urow, uc.h_m, uc.h_sh), orow = (int)urow - b * H, oc0 = (u - (int)urow * nunit) * span * kChunkPx;
        const int64_t beg = off[b];
        const int64_t n_win = off[b + 1] - beg;
        // an empty window has no bounding box: mode 0 has nothing to write (the dispatcher raises on it), the
        // full-frame modes still owe the caller the empty-FIFO background in every element of the window's slice
        const bool empty = n_win <= 0;
        if (empty && frame_mode == 0) return;
        const WindowMeta m = window_meta(bv, off, b);
        int x0 = 0, y0 = 0, Hf = H, Wf = W;
        if (!empty && (frame_mode == 0 || frame_mode == 1)) { x0 = m.xmin; y0 = m.ymin; }  // x - min(x) + 1, then [.., j - 1]
        if (frame_mode == 0) { Hf = m.ymax - m.ymin + 1; Wf = m.xmax - m.xmin + 1; }
        if (orow >= Hf || oc0 >= Wf) return;
        const int npix = min(span * kChunkPx, Wf - oc0);
        const int row = orow + y0;  // sensor row feeding this output row
        const int T = empty ? 0 : (sample_times ? sample_times[b] : ev[beg + n_win - 1].z);
        // float64 timestamps (n_imagenet hands seconds as floats, imagenet.py:1002-1006,1093-1103): the time of a record
        // is gathered by its rank from the caller's array, the sample time is a float64 too, and the FIFOs hold RANKS
        const double *tw = tf ? tf + beg : nullptr;
        const double Td = (tf && !empty) ? (sample_times_f ? sample_times_f[b] : tw[n_win - 1]) : 0.0;
        // sensor columns [oc0 + x0, oc0 + x0 + npix) can straddle span + 1 sensor chunks
        const int sc_lo = oc0 + x0, sc_hi = sc_lo + npix;
        UnitRecs ur;
        ur.sorted = bv.sorted; ur.cs = 0; ur.ce = 0; ur.nstaged = kEvStage;
        ur.r0 = make_int4(INT32_MIN, 0, 0, 0);
        ur.nseg = -1; ur.pos = (int)threadIdx.x; ur.deferred = false; ur.part = -1; ur.hot_lds = false; ur.sub = 0;
        if (!empty && row >= 0 && row < H && sc_hi > 0 && sc_lo < W) {
            const int ch_lo = max(sc_lo, 0) / kChunkPx, ch_hi = (min(sc_hi, W) - 1) / kChunkPx;
            if (bv.fused) {
                // TORE keeps, per pixel and polarity, the K most recent events: order-free too (r05b) -- a unit beyond the record stage
                // whose frame is not shifted against the sensor chunks goes to the hot launch whole, where one sweep pushes every
                // counted event's time (t - tmin + 1, one word) down a K-deep cascade of LDS atomicMax per (pixel, polarity): slot k
                // ends up holding the (k + 1)-th largest, whatever the order of arrival.  Ascending integer timestamps only (array
                // order is not time order otherwise; float times live in the caller's array).
                uint32_t *words = reinterpret_cast<uint32_t *>(w.tile);
                const int32_t tmin_w = m.tmin;
                if constexpr (HOT || SM) {
                    // (the sweeping main launch, SM: only where the order-free form holds -- else the ordered ways, in this launch or deferred)
                    const bool sweepable = HOT || (tf == nullptr && !(m.status & EVREP_ST_UNSORTED));   // wave-uniform
                    auto yes = [&]() -> bool { return sweepable; };
                    auto tsf = [&](uint32_t px, const Rec8 &q, uint2 &, const uint2 &) -> bool {
                        const int32_t t = (int32_t)q.x;
                        if (!(t < T)) return false;   // events at the sample time are dropped (tore.py:17)
                        const uint32_t p2 = (q.y >> 9) & 3u;
                        int p = (int)p2 - 1;
                        if (p2 == 3u) p = ev[beg + (q.y >> 11)].w;
                        uint32_t *wd = words + px * (uint32_t)(2 * K) + (p > 0 ? 0u : (uint32_t)K);
                        uint32_t v = (uint32_t)((int64_t)t - (int64_t)tmin_w) + 1u;
                        for (int k = 0; k < K && v != 0u; ++k) { const uint32_t old = atomicMax(wd + k, v); v = min(old, v); }
                        return false;
                    };
                    ur = unit_records<float, HOT, false, NoVisit>(bv, off, b, H * nchunk, row * nchunk + ch_lo, row * nchunk + ch_hi + 1,
                                      row * W + ch_lo * kChunkPx, (ch_hi - ch_lo + 1) * kChunkPx, w, row * W + sc_lo, ch_lo * kChunkPx, uid, npix, part,
                                      NoVisit(), unit_split_whole<HOT>(yes, tsf, yes, 2 * K));
                } else {
                    auto never = []() -> bool { return false; };
                    auto nof = [](uint32_t, const Rec8 &, uint2 &, const uint2 &) -> bool { return false; };
                    const bool hand = (uc.xflags & 6) && tf == nullptr && !(m.status & EVREP_ST_UNSORTED);   // wave-uniform
                    ur = unit_records<float, HOT, false, NoVisit>(bv, off, b, H * nchunk, row * nchunk + ch_lo, row * nchunk + ch_hi + 1,
                                      row * W + ch_lo * kChunkPx, (ch_hi - ch_lo + 1) * kChunkPx, w, row * W + sc_lo, ch_lo * kChunkPx, uid, npix, part,
                                      NoVisit(), unit_split_whole<true>(never, nof, never, 2 * K, hand ? 0u : kStEscaped, (uc.xflags & 2) ? 0u : kHotSubMin));
                }
            } else {
                const uint32_t *co = bv.chunk_off + ((size_t)b * H + row) * (nchunk + 1);
                ur.cs = co[ch_lo];
                ur.ce = co[ch_hi + 1];
                const int nr = (int)(ur.ce - ur.cs);
                if ((int)threadIdx.x < nr) ur.r0 = ur.sorted[ur.cs + threadIdx.x];
                stage_classic(ur, w);
            }
        }
        if (ur.deferred) return;
        // empty FIFO slot: inf -> clamp 5e8 -> log(5e8 + 1) - log(151)   (tore.py:69-79)
        const double log_min = log(151.0);
        const float bgv = fmaxf((float)((double)logf(500e6f + 1.0f) - log_min), 0.0f) * scale;
        // (a unit swept by the split keeps its words where the background vector lies: it is emitted from registers, below)
        if ((int)threadIdx.x < CM && ur.part != -5) w.bg[threadIdx.x] = bgv;
        wave_phase();
        float *dst = out + (size_t)b * H * W * C + ((size_t)orow * Wf + oc0) * C;
        if constexpr (HOT || SM) {
            if (ur.part == -5) {   // wave-uniform: the unit was swept by the cascade; slot k of a pixel holds its (k + 1)-th latest time
                const uint32_t *words = reinterpret_cast<const uint32_t *>(w.tile);
                const int lane = threadIdx.x;
                const bool vec = (C % 4) == 0 && (reinterpret_cast<uintptr_t>(dst) & 15u) == 0;
                for (int pt = 0; pt * kWave < npix; ++pt) {
                    if (pt * kWave + lane < npix) {
                        const uint32_t *wd = words + (uint32_t)(pt * kWave + lane) * (uint32_t)(2 * K);
                        float vals[CM];
#pragma unroll
                        for (int c = 0; c < CM; ++c) {
                            float v = bgv;
                            if (c < C) {
                                const uint32_t wv = wd[c];
                                if (wv) {
                                    const int64_t t = (int64_t)m.tmin + (int64_t)(wv - 1u);
                                    v = (float)(double)((int64_t)T - t);
                                    v = fminf(v, 500e6f);
                                    v = fmaxf((float)((double)logf(v + 1.0f) - log_min), 0.0f) * scale;
                                }
                            }
                            vals[c] = v;
                        }
                        store_pixel<float, CM>(dst + ((size_t)pt * kWave + lane) * C, vals, C, vec);
                    }
                }
                return;
            }
        }
        // The value of a FIFO slot depends on its event alone (one sample time per window): log(min(T - t, 5e8) + 1) - log(151),
        // floored at 0 (tore.py:63-79).  So the digest forms it per EVENT -- one record per lane, all lanes at once, one logf --
        // and the FIFOs hold finished values: 2 K logarithms per touched pixel become one per event.  Digest:
        // {pixel id, value bits, 1 iff the event counts (ts < currentSampleTime, tore.py:17: events at T are dropped), p}.
        auto digest = [&](const Rec &r) -> Rec {
            bool counts;
            float v;
            if (tw) { const double te = tw[r.y]; counts = te < Td; v = (float)(Td - te); }
            else { counts = r.z < T; v = (float)(double)((int64_t)T - (int64_t)r.z); }
            v = fminf(v, 500e6f);
            v = fmaxf((float)((double)logf(v + 1.0f) - log_min), 0.0f) * scale;
            return make_int4(r.x, __float_as_int(v), counts ? 1 : 0, r.w);
        };
        const bool unsorted = (m.status & EVREP_ST_UNSORTED) != 0u;   // wave-uniform
        auto reduce = [&](uint32_t jb, uint32_t je, auto get, float(&vals)[CM]) {
            float fp[(CM / 2)], fn[(CM / 2)];  // most recent first; slots never filled keep the empty-FIFO value
#pragma unroll
            for (int q = 0; q < (CM / 2); ++q) { fp[q] = bgv; fn[q] = bgv; }
            if (!unsorted) {
                for (uint32_t j = jb; j < je; ++j) {
                    const Rec e = get(j);
                    if (!e.z) continue;
                    const float v = __int_as_float(e.y);
                    if (e.w > 0) {
#pragma unroll
                        for (int q = (CM / 2) - 1; q > 0; --q) fp[q] = fp[q - 1];
                        fp[0] = v;
                    } else {
#pragma unroll
                        for (int q = (CM / 2) - 1; q > 0; --q) fn[q] = fn[q - 1];
                        fn[0] = v;
                    }
                }
            } else {
                // Timestamps in any order (r05; wave-uniform): the reference keeps np.partition([t] + v[:k-1], k-1)[:k] per
                // event in ARRAY order (tore.py:22-25).  numpy's float64 partition of such a short vector SORTS it (numpy >= 2.0
                // on AVX2 / AVX-512 hosts: x86-simd-sort's bitonic network for <= 256 elements -- what the golden of this path
                // was generated with; the scalar introselect only swaps the maximum to the end, a different ORDER of the same
                // values), so v stays ascending and a step inserts t into the k - 1 smallest kept so far; the largest of the k
                // is only dropped by the NEXT event of the pixel.  A slot's value is a non-decreasing function of its interval
                // (scale >= 0; non-increasing below), so the finished values are inserted instead of the intervals.  For
                // ascending timestamps this IS the FIFO above.
                const float sg = scale < 0.0f ? -1.0f : 1.0f;
                auto insert = [&](float(&a)[(CM / 2)], float v) {
                    float r[(CM / 2)];
#pragma unroll
                    for (int q = 0; q < (CM / 2); ++q) {
                        const float below = q > 0 ? a[q - 1] : v;            // max(a[q-1], v), a[-1] = -inf
                        const float hi = q > 0 ? (sg * below > sg * v ? below : v) : v;
                        const bool last = q >= K - 1;                        // a'[q] = +inf: the old v[k-1] is not kept
                        r[q] = (last || sg * hi < sg * a[q]) ? hi : a[q];
                    }
#pragma unroll
                    for (int q = 0; q < (CM / 2); ++q) if (q < K) a[q] = r[q];
                };
                for (uint32_t j = jb; j < je; ++j) {
                    const Rec e = get(j);
                    if (!e.z) continue;
                    const float v = __int_as_float(e.y);
                    if (e.w > 0) insert(fp, v); else insert(fn, v);
                }
            }
#pragma unroll
            for (int c = 0; c < CM; ++c) vals[c] = bgv;
            // channel layout: positives [0, K), negatives [K, 2K)
#pragma unroll
            for (int q = 0; q < (CM / 2); ++q) {
#pragma unroll
                for (int c = 0; c < CM; ++c) {
                    if (q < K && c == q) vals[c] = fp[q];
                    if (q < K && c == K + q) vals[c] = fn[q];
                }
            }
        };
        emit_chunk<float, CM, HOT>(ur, digest, row * W + sc_lo, npix, C, dst, w, (const float *)w.bg, reduce);
    });
}

// --------------------------------------------------------------------------------------------
// A6 (r06), after the key-sorted pass: EventStack as a STREAM.  A pixel's levels only depend on its LAST record in array order
// (ndarray.put is last-write-wins, event_stack.py:125): one word per pixel of the unit, ((rank + 1) << 2 | polarity code) under LDS
// atomicMax, in ONE sweep of the unit's records whatever it holds; then a lane per pixel forms the pixel's S levels into the unit's
// tile and the tile leaves as one coalesced burst.  No grouping, no stage, no hot launch.
// LDS: tile [npixa * S] f32 | last [npixa] u32 | head [64 * RB] | srcs [128]
__host__ __device__ inline size_t event_stack_stream_lds_bytes(int S, int npixa, int rb) {
    return align16((size_t)npixa * S * 4) + (size_t)npixa * 4 + (size_t)(64 * rb) * 4 + 128 * 4;
}
template <int CM, int RB>
__global__ __launch_bounds__(kWave, 6) void k_event_stack_stream(BinView bv, const int64_t *__restrict__ off, int H, int W, int nchunk,
                                                             UnitCfg uc, int S, int premap, float scale, float *__restrict__ out) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int lane = threadIdx.x;
    const int uid = chunk_unit((int)(gridDim.x * gridDim.y * gridDim.z));
    int chunk, nch;
    const ChunkGeom g = unit_geom(H, W, nchunk, uc, chunk, nch, uid);
    const int b = g.b;
    const int64_t beg = off[b];
    const int64_t n_win = off[b + 1] - beg;
    const int npixa = (uc.span + uc.merge) * kChunkPx;
    float *tile = reinterpret_cast<float *>(smem);
    uint32_t *last = reinterpret_cast<uint32_t *>(smem + align16((size_t)npixa * S * 4));
    uint32_t *head = last + npixa;
    uint32_t *srcs = head + 64 * RB;
    for (int v = lane; v * 4 < npixa; v += kWave) reinterpret_cast<uint4 *>(last)[v] = make_uint4(0u, 0u, 0u, 0u);
    const int4 *evw = bv.ev + beg;
    const int c0 = g.c0;
    wave_phase();
    stream_unit_records<RB>(bv, b, beg, n_win, H * nchunk, g.row * nchunk + chunk, g.row * nchunk + chunk + nch, head, srcs, StreamNoPre(),
        [&](bool have, const Rec8 &q, const uint2 &) {
            if (have) atomicMax(&last[((q.y & 511u) - (uint32_t)c0) & 511u], (((q.y >> 11) + 1u) << 2) | ((q.y >> 9) & 3u));
        });
    wave_phase();
    // level k keeps events[off_k:], off_k = sum_{j=1..k} N // 2^j  (event_stack.py:70-82)
    int offk[CM];
    {
        int cur = (int)n_win, o = 0;
#pragma unroll
        for (int k = 0; k < CM; ++k) { offk[k] = o; cur /= 2; o += cur; }
    }
    for (int pt = 0; pt * kWave < g.npix; ++pt) {
        const int px = pt * kWave + lane;
        if (px < g.npix) {
            const uint32_t wv = last[px];
            const int rank = (int)(wv >> 2) - 1;
            float v = 0.0f;
            if (wv) {
                const uint32_t p2 = wv & 3u;
                int p = p2 == 3u ? evw[rank].w : (int)p2 - 1;
                if (premap == 1) p = (p + 1) >> 1;                   // (p + 1) // 2   (gen1_transforms.py:34)
                v = (float)(int8_t)(premap == 2 ? p : 2 * p - 1) * scale;   // 2*p - 1 as int8 (event_stack.py:18)
            }
            float *mine = tile + (size_t)px * S;
#pragma unroll
            for (int l = 0; l < CM; ++l) if (l < S) mine[l] = (wv && rank >= offk[l]) ? v : 0.0f;
        }
    }
    wave_phase();
    float *dst = out + (((size_t)b * H + g.row) * (size_t)W + g.c0) * S;
    tile_store(tile, g.npix * S, dst);
}

// --------------------------------------------------------------------------------------------
// A8 (r06), after the key-sorted pass: TORE as a STREAM -- the unit's tile in LDS IS the K-deep FIFOs.
// A FIFO slot's value depends on its event alone (one sample time per window) and is a non-increasing function of the event's
// time, so "the K most recent events of a pixel and polarity, most recent first" = "the K SMALLEST finished values, ascending":
// the tile (pixels x 2 K float32, output layout) starts as the empty-FIFO value everywhere and every record pushes its value --
// the digest: one logf per record and lane, all lanes at once -- down a K-deep cascade of LDS atomicMin on the value's bits (values
// are >= +0: their bit patterns order as unsigned integers); slot k ends up with the (k + 1)-th smallest whatever the order of
// arrival, hot pixels are serialised by the LDS atomic unit and not by rounds of the wave, and a cascade stops as soon as what it
// carries is the empty value (a sparse pixel's first record: one atomic).  No grouping, no order, no walk, no per-slot logarithm,
// no hot launch: the tile leaves as one burst.  Any frame (bounding box, shifted, full): the unit's records are those of the
// sensor keys its output columns cover, a record's output pixel is its column minus the frame's offset.
// Windows whose timestamps are not ascending keep the reference's array-order semantics (np.partition on the k-vector, see
// k_tore's reduce): there one record per (pixel, polarity) and round inserts into its FIFO -- an election as in k_voxel_stream.
// LDS: tile [npixa * 2 K] u32 | tag [2 * npixa] | head [64 * RB] (a big unit's queue: 128 x 8 bytes, RB >= 4) | srcs [64] / run table [128]
__host__ __device__ inline size_t tore_stream_lds_bytes(int K, int npixa, int rb) {
    return align16((size_t)npixa * 2 * K * 4) + (size_t)2 * npixa * 4 + (size_t)(64 * rb) * 4 + 128 * 4;
}
#ifndef EVREP_TST_WAVES
#define EVREP_TST_WAVES 6
#endif
template <int RB>
__global__ __launch_bounds__(kWave, EVREP_TST_WAVES) void k_tore_stream(const int4 *__restrict__ ev, BinView bv, const int64_t *__restrict__ off,
                                               const int32_t *__restrict__ sample_times, int H, int W, int nchunk, UnitCfg uc,
                                               int K, int frame_mode, float scale, float *__restrict__ out) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int lane = threadIdx.x;
    const int C = 2 * K;
    const int uid = chunk_unit((int)(gridDim.x * gridDim.y * gridDim.z));
#ifdef EVREP_TIMING   // (tools/experiments/wave_timeline.py BUILDER=tore: slot 7 = the wave's start, 5 = its lifetime, 6 = its records,
    const long long ts_t0 = (long long)wall_clock64();   //  0-3 = a big unit's sweep: issue / wait / process ticks, batches)
    uint32_t ts_nrec = 0u;
    bv.dbg_wave = bv.dbg ? bv.dbg + (size_t)uid * 8 : nullptr;
#endif
    const int nunit = uc.nunit;
    const uint32_t urow = fastdiv((uint32_t)uid, uc.nunit_m, uc.nunit_sh);
    const int b = (int)fastdiv(urow, uc.h_m, uc.h_sh), orow = (int)urow - b * H, oc0 = (uid - (int)urow * nunit) * uc.span * kChunkPx;
    const int64_t beg = off[b];
    const int64_t n_win = off[b + 1] - beg;
    const bool empty = n_win <= 0;
    if (empty && frame_mode == 0) return;   // no bounding box: nothing to write (k_tore)
    const MetaRaw mraw = meta_prefetch(bv, b);
    int Tl = 0;
    if (!empty) Tl = sample_times ? sample_times[b] : ev[beg + n_win - 1].z;
    const WindowMeta m = meta_finish(bv, off, b, mraw);
    int x0 = 0, y0 = 0, Hf = H, Wf = W;
    if (!empty && (frame_mode == 0 || frame_mode == 1)) { x0 = m.xmin; y0 = m.ymin; }
    if (frame_mode == 0) { Hf = m.ymax - m.ymin + 1; Wf = m.xmax - m.xmin + 1; }
    if (orow >= Hf || oc0 >= Wf) return;
    const int npix = min(uc.span * kChunkPx, Wf - oc0);
    const int row = orow + y0;
    const int T = Tl;
    const int sc_lo = oc0 + x0, sc_hi = sc_lo + npix;   // the sensor columns behind the unit's output columns
    const int npixa = uc.span * kChunkPx;
    uint32_t *tile = reinterpret_cast<uint32_t *>(smem);
    uint32_t *tag = reinterpret_cast<uint32_t *>(smem + align16((size_t)npixa * C * 4));
    uint32_t *head = tag + 2 * npixa;
    uint32_t *srcs = head + 64 * RB;
    static_assert(RB >= 4, "a big unit's queue (128 x 8 bytes) takes the place of head");
    uint2 *queue = reinterpret_cast<uint2 *>(head);   // [128]: the records of a big unit that still have a FIFO to enter (below); `head` is idle then
    const double log_min = log(151.0);
    const float bgv = fmaxf((float)((double)logf(500e6f + 1.0f) - log_min), 0.0f) * scale;
    const uint32_t bgb = __float_as_uint(bgv);
    const bool unsorted = (m.status & EVREP_ST_UNSORTED) != 0u;   // wave-uniform
    {
        uint4 *t4 = reinterpret_cast<uint4 *>(tile);
        const int nvec = (npix * C + 3) / 4;
        for (int v = lane; v < nvec; v += kWave) t4[v] = make_uint4(bgb, bgb, bgb, bgb);
        // tag: the election words of an unsorted window (all ones) / the FIFOs' record COUNTERS of a big unit of a sorted one (zero)
        { const uint32_t tv = unsorted ? ~0u : 0u; uint4 *g4 = reinterpret_cast<uint4 *>(tag); for (int v = lane; v * 4 < 2 * npixa; v += kWave) g4[v] = make_uint4(tv, tv, tv, tv); }
    }
    float *dst = out + (size_t)b * H * W * C + ((size_t)orow * Wf + oc0) * C;
    int c0 = 0, klo = 0, khi = 0;
    const bool has = !empty && row >= 0 && row < H && sc_hi > 0 && sc_lo < W;   // wave-uniform: sensor keys lie behind the unit
    if (has) {
        const int ch_lo = max(sc_lo, 0) / kChunkPx, ch_hi = (min(sc_hi, W) - 1) / kChunkPx;
        klo = row * nchunk + ch_lo; khi = row * nchunk + ch_hi + 1;
        c0 = ch_lo * kChunkPx;
    }
    const int4 *evw = ev + beg;
    const int pxoff = sc_lo - c0;   // output pixel = (column inside the gathered chunks) - pxoff
    // one batch: a record's finished value down its FIFO
    // (digest: the record's time, output pixel and polarity; insert: the value and the FIFO)
    auto digest = [&](bool have, const Rec8 &q, int32_t &t, int &px, int &p) -> bool {
        t = (int32_t)q.x;
        px = (int)(((q.y & 511u) - (uint32_t)c0) & 511u) - pxoff;
        const bool ok = have && t < T && px >= 0 && px < npix;   // events at the sample time are dropped (tore.py:17)
        const uint32_t p2 = (q.y >> 9) & 3u;
        p = (int)p2 - 1;
        if (ok && p2 == 3u) p = evw[q.y >> 11].w;
        return ok;
    };
    auto insert = [&](bool ok, int32_t t, int px, int p) {
        float v = (float)(double)((int64_t)T - (int64_t)t);
        v = fminf(v, 500e6f);
        v = fmaxf((float)((double)logf(v + 1.0f) - log_min), 0.0f) * scale;
        uint32_t *wd = tile + (uint32_t)(ok ? px : 0) * (uint32_t)C + (p > 0 ? 0u : (uint32_t)K);
        if (!unsorted) {
            uint32_t vb = __float_as_uint(v);
            if (ok) {
                for (int k = 0; k < K && vb < bgb; ++k) { const uint32_t old = atomicMin(wd + k, vb); vb = max(old, vb); }
            }
        } else {
            // array order (k_tore's reduce, unsorted): insert v into the ascending k - 1 smallest kept so far; the largest of
            // the k is only dropped by the NEXT event of the FIFO.  One record per FIFO and round.
            uint32_t *tg = tag + 2u * (uint32_t)(ok ? px : 0) + (p > 0 ? 0u : 1u);
            bool pend = ok;
            while (__any(pend)) {
                if (pend) atomicMin(tg, (uint32_t)lane);
                wave_phase();
                const bool win = pend && *tg == (uint32_t)lane;
                wave_phase();
                if (win) {
                    float *fa = reinterpret_cast<float *>(wd);
                    float prev = v;   // a'[q - 1] of the OLD list (q > 0), v for q == 0
                    for (int q2 = 0; q2 < K; ++q2) {
                        const float aq = fa[q2];
                        const float hi = q2 > 0 ? (prev > v ? prev : v) : v;
                        const bool last = q2 >= K - 1;
                        fa[q2] = (last || hi < aq) ? hi : aq;
                        prev = aq;
                    }
                    *tg = ~0u;
                    pend = false;
                }
                wave_phase();
            }
        }
    };
    // A unit of more than 64 * RB records of a window with ascending timestamps (r06b): its batches arrive LAST records first, so the
    // first K records a FIFO meets are its K latest -- a counter per FIFO (the tag words), and only the records that meet a FIFO
    // still below K are QUEUED; the values (int64 / float64 conversions, logf) and the K-deep cascade of dependent returning LDS
    // atomics (~200 cycles each for a lone wave: 0.5 us per batch) run on full batches of queued records.  A 4 500-record unit of a
    // 1 Mpx circle window holds at most 256 x K = 1 536 records that matter: 80 cascades -> ~25.  (The counter over-admits inside a
    // batch -- every lane of a FIFO below K goes -- and the cascade keeps the K smallest values whatever it is fed: exact.)
    int big = 0;
    uint32_t qn = 0u;   // queued records (wave-uniform)
    auto drain = [&](uint32_t n) {   // the first n (<= 64) queued records
        const uint2 e = queue[lane];
        const bool ok = (uint32_t)lane < n;
        insert(ok, (int32_t)e.x, (int)(e.y & 0xffffu), (int)(e.y >> 16) - 1);
    };
    auto batch = [&](bool have, const Rec8 &q) {
        int32_t t; int px, p;
        const bool ok = digest(have, q, t, px, p);
        if (big != 1 || unsorted) { insert(ok, t, px, p); return; }   // (big == 2: the sweep keeps its run table in `head`, no queue)
        uint32_t *cn = tag + 2u * (uint32_t)(ok ? px : 0) + (p > 0 ? 0u : 1u);
        const bool live = ok && *cn < (uint32_t)K;
        const unsigned long long lm = __ballot(live);
        if (lm == 0ull) return;                                    // (wave-uniform) nothing of this batch matters any more
        if (live) {
            atomicAdd(cn, 1u);
            queue[qn + (uint32_t)__popcll(lm & ((1ull << lane) - 1ull))] = make_uint2((uint32_t)t, (uint32_t)px | ((uint32_t)(p > 0 ? 2 : 0) << 16));
        }
        qn += (uint32_t)__popcll(lm);
        wave_phase();
        if (qn >= (uint32_t)kWave) {
            drain((uint32_t)kWave);
            const uint2 mv = queue[kWave + lane];                  // the rest moves to the front
            wave_phase();
            if ((uint32_t)(kWave + lane) < qn) queue[lane] = mv;
            qn -= (uint32_t)kWave;
            wave_phase();
        }
    };
    wave_phase();
    if (has) {
#ifdef EVREP_TIMING
        ts_nrec =
#endif
        stream_unit_records<RB>(bv, b, beg, n_win, H * nchunk, klo, khi, head, srcs, StreamNoPre(),
                                [&](bool have, const Rec8 &q, const uint2 &) { batch(have, q); }, (uint32_t)EVREP_STREAM_BYRUN, !unsorted, &big);
    }
    wave_phase();
    if (qn) drain(qn);
    wave_phase();
    tile_store(reinterpret_cast<const float *>(tile), npix * C, dst);
#ifdef EVREP_TIMING
    if (lane == 0 && bv.dbg) { bv.dbg[(size_t)uid * 8 + 7] = (unsigned long long)ts_t0; bv.dbg[(size_t)uid * 8 + 5] = (unsigned long long)((long long)wall_clock64() - ts_t0); bv.dbg[(size_t)uid * 8 + 6] = ts_nrec; }
#endif
}

// --------------------------------------------------------------------------------------------
// A2: compute_repr (representation_search/gromov_wasserstein.py:72-82), t normalised as :96.
// mode 1: tonic.transforms.ToVoxelGrid as consumed at gen1_transforms.py:22-25 (parity unpinned).
// mode 2: ev-licious events_to_voxel_grid, integer-pixel path (ev-licious/src/evlicious/tools/utils.py:
//         52-108): the bilinear weight is taken from the INTEGER bin (:74), so the lower bin receives p
//         and the upper bin an exact zero -- a signed event count per (time bin, y, x).
// --------------------------------------------------------------------------------------------
// CM = compile-time channel capacity (8 or 16): the register arrays of a <= 8-bin grid are half the size
template <int CM, bool HOT = false>
__global__ __launch_bounds__(kWave, HOT ? 4 : (CM <= 8 ? EVREP_VOXEL_WAVES : 1)) void k_voxel(const int4 *__restrict__ ev, BinView bv, const int64_t *__restrict__ off,
                                                int H, int W, int nchunk, UnitCfg uc, int bins, int mode, double scale,
                                                const int64_t *__restrict__ t_range, const double *__restrict__ tnorm,
                                                double *__restrict__ out) {
    extern __shared__ __align__(16) unsigned char smem[];
    run_units<HOT>(bv, [&](int uid, int part) {
        WaveLds<double, HOT> w(smem, bins, (uc.span + uc.merge) * kChunkPx, uc.stage);
        w.arm(uc.hold);
#ifdef EVREP_TIMING
        w.dbg = bv.dbg + 8 * (size_t)uid;
#endif
        // the window's first / last timestamp: two dependent load levels (extent, then events) issued BEFORE the unit's own two
        // levels (run tables, then records), not behind them (r03: they were a third and fourth step of the wave's latency chain)
        int chunk0;
        const int b0 = unit_geom(H, W, nchunk, uc, chunk0, uid).b;
        const int64_t beg = off[b0];
        const int64_t n_win = off[b0 + 1] - beg;
        int tz0 = 0, tz1 = 0;
        if (n_win > 0) { tz0 = ev[beg].z; tz1 = ev[beg + n_win - 1].z; }
        ChunkGeom g;
        w.mark(6);
        const UnitRecs u = unit_front(bv, off, H, W, nchunk, uc, w, g, uid, part);
        w.mark(0);
        if (u.deferred) return;
        double *dst = out + (((size_t)g.b * H + g.row) * (size_t)W + g.c0) * bins;
        double t0 = 0.0, den = 1.0;
        if (n_win > 0) { t0 = (double)tz0; den = (double)tz1 - t0; }
        // explicit [t0_us, t1_us] of ev-licious' events_to_voxel_grid (utils.py:60-63), mode 2 only
        if (t_range) { t0 = (double)t_range[2 * g.b]; den = (double)(t_range[2 * g.b + 1] - t_range[2 * g.b]); }
        // the fractional bin position of an event: one float64 division
        auto bin_pos = [&](int t) -> double {
            if (mode == 2) {
                // t_norm = (num_bins - 1) * (t - t0) / deltaT: int64 product, one float64 division
                const int64_t num = (int64_t)(bins - 1) * ((int64_t)t - (int64_t)t0);
                return (double)num / (den == 0.0 ? 1.0 : den);
            }
            if (mode == 0) {
                const double tn = ((double)t - t0) / den;
                return (double)(bins - 1) * tn;
            }
            const double num = (double)bins * ((double)t - t0);
            return num / den;
        };
        // emit_chunk digests the staged records -- one record per lane, all lanes at once: the bin position rides in the
        // (rank, t) fields, which the segment walks below do not need.  Both np.add.at passes of every segment then run
        // without a division (they were the bulk of this kernel's VALU work: the walks are divergent, one division per step
        // and lane).
        // tnorm (evrep_voxel_tnorm): the caller's own normalised time, gathered by the record's rank -- one load per record
        const double *tw = tnorm ? tnorm + beg : nullptr;
        auto digest = [&](const Rec &r) -> Rec {
            const double bp = tw ? (double)(bins - 1) * tw[r.y] : bin_pos(r.z);
            return make_int4(r.x, __double2loint(bp), __double2hiint(bp), r.w);
        };
        // The pixel's `bins` sums live in the lane's own row of an LDS scratch behind the wave's carve (r06), not in a register
        // array: a register array indexed by a per-lane bin is a select chain -- one float64 add, a compare and four v_cndmask per
        // BIN and step, ~56 of the ~75 VALU instructions of a step, in walks that are divergent (the wave runs as many steps as its
        // longest pixel holds records, twice) and in launches that the counters show bound by VALU issue (SIMDs 0.6-0.8 busy on
        // clustered and dense windows, profiles/r06).  The running sum of the CURRENT bin stays in registers -- a pixel's records
        // are time-ordered, so its bin changes at most `bins` times per pass -- and moves to / from the row only when the bin
        // changes (exact: no rounding happens on the way), so every (pixel, bin) sum still adds its terms in the reference's order:
        // pass 0's in array order, then pass 1's.
        double *scr = reinterpret_cast<double *>(smem + chunk_lds_bytes(bins, 8, (uc.span + uc.merge) * kChunkPx, uc.stage)) + (size_t)threadIdx.x * bins;
        auto reduce = [&](uint32_t jb, uint32_t je, auto get, double(&vals)[CM]) {
#pragma unroll
            for (int c = 0; c < CM; ++c) if (c < bins) scr[c] = 0.0;
            // two np.add.at passes: lower bin for every event, then upper bin for every event
            for (int pass = 0; pass < (mode == 2 ? 1 : 2); ++pass) {
                int cur = -1;
                double acc = 0.0;
                for (uint32_t j = jb; j < je; ++j) {
                    const Rec ec = get(j);
                    double p = (double)ec.w;
                    if (mode == 1 && ec.w == 0) p = -1.0;
                    const double bpos = __hiloint2double(ec.z, ec.y);
                    // flat time span (0/0): the reference yields NaN garbage.  Mode 2 truncates toward zero
                    // (astype("int32"), utils.py:67), so an event up to one bin before t0_us still lands in bin 0
                    if (!(bpos > (mode == 2 ? -1.0 : -0.0) && bpos < 1.0e9) && !(bpos == 0.0)) continue;
                    const int bi = (int)bpos;
                    const int blim = bi + pass;
                    if (blim < bins) {
                        double wgt;
                        if (mode == 2) wgt = 1.0;
                        else if (mode == 0) wgt = 1.0 - fabs((double)blim - bpos);
                        else { const double dts = bpos - (double)bi; wgt = pass ? dts : 1.0 - dts; }
                        const double wp = wgt * p;
                        if (blim != cur) {
                            if (cur >= 0) scr[cur] = acc;
                            acc = scr[blim];
                            cur = blim;
                        }
                        acc = acc + wp;
                    }
                }
                if (cur >= 0) scr[cur] = acc;
            }
#pragma unroll
            for (int c = 0; c < CM; ++c) vals[c] = c < bins ? scr[c] : 0.0;
            if (scale != 1.0) {
#pragma unroll
                for (int c = 0; c < CM; ++c) vals[c] = vals[c] * scale;
            }
        };
        emit_chunk<double, CM, HOT>(u, digest, g.row * W + g.c0, g.npix, bins, dst, w, (const double *)nullptr, reduce);
    });
}

// --------------------------------------------------------------------------------------------
// A2 (r06), after the key-sorted pass: the voxel grid as a STREAM -- no order by pixel at all.
// What the reference fixes is the order of the additions into every (pixel, bin) CELL: pass 0's terms (the lower bin of every
// event) in array order, then pass 1's (the upper bin).  The records of a unit arrive in array order already (the block runs are
// visited in block order and a run is in arrival order), so the unit's cells -- a (pixels x bins) float64 tile in LDS, in output
// layout -- are simply bumped in sweep order, 64 records at a time, one sweep per pass.  The only thing to respect inside a
// batch of 64 is two lanes meeting in one pixel: a leader election per pixel (ds_min of the lane id on a tag word; the lowest
// pending lane of a pixel goes first, as the array order demands) lets one lane per pixel add per round -- one round for almost
// every batch of a sparse or uniformly dense window, as many as a pixel holds records of the batch on a hot one.  Against the
// ordered paths (count sweep, scan, placement with an 8-bit ballot match per batch, digest, divergent walks with one lane per
// pixel, part tiles): no counting sort, no segment list, no walk, no spill slot, and the tile leaves the wave as one coalesced
// burst.  Units of up to 64 * RB records keep their digested records in registers between the two sweeps; larger ones are read
// and digested twice (8 bytes per record from L2).
// LDS: cells [npixa * bins] float64 | tag [npixa] | head [64 * RB] | srcs [128] | touched [npixa] bytes
__host__ __device__ inline size_t voxel_stream_lds_bytes(int bins, int npixa, int rb) {
    return align16((size_t)npixa * bins * 8) + (size_t)npixa * 4 + (size_t)(64 * rb + 128) * 4 + align16((size_t)npixa);
}
#ifndef EVREP_VS_WAVES
#define EVREP_VS_WAVES 6
#endif
template <int RB>
__global__ __launch_bounds__(kWave, EVREP_VS_WAVES) void k_voxel_stream(const int4 *__restrict__ ev, BinView bv, const int64_t *__restrict__ off,
                                                int H, int W, int nchunk, UnitCfg uc, int bins, int mode, double scale,
                                                const int64_t *__restrict__ t_range, const double *__restrict__ tnorm,
                                                double *__restrict__ out) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int lane = threadIdx.x;
    const int uid = chunk_unit((int)(gridDim.x * gridDim.y * gridDim.z));
#ifdef EVREP_TIMING   // experiment builds (tools/experiments/wave_timeline.py): slot 7 = the wave's start (10 ns ticks, absolute), 5 = its lifetime, 6 = its records
    const long long vs_t0 = (long long)wall_clock64();
#endif
    int chunk, nch;
    const ChunkGeom g = unit_geom(H, W, nchunk, uc, chunk, nch, uid);
    const int b = g.b;
    const int64_t beg = off[b];
    const int64_t n_win = off[b + 1] - beg;
    int tz0 = 0, tz1 = 0;
    if (n_win > 0) { tz0 = ev[beg].z; tz1 = ev[beg + n_win - 1].z; }
    const int NK = H * nchunk, klo = g.row * nchunk + chunk;
    const StreamRuns R = stream_runs(bv, b, n_win, NK, klo, klo + nch);   // (up to 128 block runs: two per lane)
    const int npixa = (uc.span + uc.merge) * kChunkPx;
    double *acc = reinterpret_cast<double *>(smem);
    uint32_t *tag = reinterpret_cast<uint32_t *>(smem + align16((size_t)npixa * bins * 8));
    uint32_t *head = tag + npixa;
    uint32_t *srcs = head + 64 * RB;
    unsigned char *touched = reinterpret_cast<unsigned char *>(srcs + 128);
    const int ncell = g.npix * bins;
    {   // zero cells, free tags, nothing touched (overlaps the table loads)
        uint4 *z = reinterpret_cast<uint4 *>(acc);
        for (int v = lane; v * 2 < ncell; v += kWave) z[v] = make_uint4(0u, 0u, 0u, 0u);
        uint4 *t4 = reinterpret_cast<uint4 *>(tag);
        for (int v = lane; v * 4 < npixa; v += kWave) t4[v] = make_uint4(~0u, ~0u, ~0u, ~0u);
        if (scale != 1.0) { uint4 *c4 = reinterpret_cast<uint4 *>(touched); for (int v = lane; v * 16 < npixa; v += kWave) c4[v] = make_uint4(0u, 0u, 0u, 0u); }
    }
    const int nb = R.nb;
    const uint32_t nrec = nb > 0 ? R.nrec : 0u;
    const bool wide = nb > kWave;   // wave-uniform
    {
        // a BURST unit -- more records than uc.stage and than three times its window's average -- goes to the hot launch, untouched:
        // several waves per unit there (k_voxel_hot).  A uniformly dense window (every unit near the average: one or two election
        // rounds per batch) streams whatever its units hold.
        const uint32_t avg = (uint32_t)((uint64_t)n_win / (uint64_t)(H * nchunk));   // records per unit, on average
        if (nrec > (uint32_t)uc.stage && nrec > 3u * avg) {   // wave-uniform
            if (!defer_items(bv, uid, (uint32_t)kHotWhole, 1u) && lane == 0)
                atomicOr(&bv.stats_rw[(size_t)b * bv.nblk].status, EVREP_ST_HOT_OVERFLOW);
            return;
        }
    }
    double *dst = out + (((size_t)b * H + g.row) * (size_t)W + g.c0) * bins;
    double t0 = 0.0, den = 1.0;
    if (n_win > 0) { t0 = (double)tz0; den = (double)tz1 - t0; }
    if (t_range) { t0 = (double)t_range[2 * b]; den = (double)(t_range[2 * b + 1] - t_range[2 * b]); }
    const double *tw = tnorm ? tnorm + beg : nullptr;
    const int4 *evw = ev + beg;
    const int c0 = g.c0;
    const Rec8 *__restrict__ s8 = reinterpret_cast<const Rec8 *>(bv.sorted);
    // the fractional bin position of a record (k_voxel's digest): one float64 division, or the caller's own normalised time
    auto bin_of = [&](const Rec8 &q) -> double {
        if (tw) return (double)(bins - 1) * tw[q.y >> 11];
        const int t = (int)q.x;
        if (mode == 2) {
            const int64_t num = (int64_t)(bins - 1) * ((int64_t)t - (int64_t)t0);
            return (double)num / (den == 0.0 ? 1.0 : den);
        }
        if (mode == 0) {
            const double tn = ((double)t - t0) / den;
            return (double)(bins - 1) * tn;
        }
        const double num = (double)bins * ((double)t - t0);
        return num / den;
    };
    auto pol_of = [&](const Rec8 &q) -> int {
        const uint32_t p2 = (q.y >> 9) & 3u;
        int p = (int)p2 - 1;
        if (p2 == 3u) p = evw[q.y >> 11].w;
        return p;
    };
    auto px_of = [&](const Rec8 &q) -> uint32_t { return ((q.y & 511u) - (uint32_t)c0) & 511u; };
    const double lowlim = mode == 2 ? -1.0 : -0.0;
    // one batch of one pass: lane holds (have, pixel, bin position, polarity) of one record; lanes are in array order
    auto apply = [&](int pass, bool have, uint32_t px, double bpos, int p) {
        double pd = (double)p;
        if (mode == 1 && p == 0) pd = -1.0;
        // flat time span (0/0): the reference yields NaN garbage.  Mode 2 truncates toward zero (astype("int32"), utils.py:67),
        // so an event up to one bin before t0_us still lands in bin 0
        bool ok = have && ((bpos > lowlim && bpos < 1.0e9) || bpos == 0.0);
        const int bi = ok ? (int)bpos : 0;
        const int blim = bi + pass;
        ok = ok && blim < bins;
        double wgt;
        if (mode == 2) wgt = 1.0;
        else if (mode == 0) wgt = 1.0 - fabs((double)blim - bpos);
        else { const double dts = bpos - (double)bi; wgt = pass ? dts : 1.0 - dts; }
        const double wp = wgt * pd;
        double *cell = acc + (px * (uint32_t)bins + (uint32_t)blim);
        bool pend = ok;
        while (__any(pend)) {
            if (pend) atomicMin(&tag[px], (uint32_t)lane);
            wave_phase();
            const bool win = pend && tag[px] == (uint32_t)lane;
            wave_phase();
            if (win) {
                *cell = *cell + wp;
                tag[px] = ~0u;
                pend = false;
            }
            wave_phase();
        }
    };
    if (nrec != 0u) {
        // record j of the unit, if in run `lane` / run 64 + lane: src + j
        const uint32_t src0 = (uint32_t)beg + ((uint32_t)lane << bv.chunk_shift) + R.a0 - R.pre0;
        const uint32_t src1 = (uint32_t)beg + ((uint32_t)(kWave + lane) << bv.chunk_shift) + R.a1 - R.pre1;
        const int npass = mode == 2 ? 1 : 2;
        if (nrec <= (uint32_t)(64 * RB)) {
            // the whole unit in registers: the run of record j by a max-scan over the runs' first positions (unit_records)
#pragma unroll
            for (int i = 0; i < RB; ++i) head[lane + 64 * i] = 0u;
            srcs[lane] = src0;
            if (wide) srcs[kWave + lane] = src1;
            wave_phase();
            if (R.len0 > 0u && R.pre0 < (uint32_t)(64 * RB)) head[R.pre0] = (uint32_t)lane;
            if (wide && R.len1 > 0u && R.pre1 < (uint32_t)(64 * RB)) head[R.pre1] = (uint32_t)(kWave + lane);
            wave_phase();
            Rec8 q[RB];
            uint32_t carry = 0u;
#pragma unroll
            for (int i = 0; i < RB; ++i) {
                q[i] = make_uint2(0u, 0u);
                if ((uint32_t)(64 * i) < nrec) {   // uniform
                    const uint32_t k = max(carry, wave_incl_max_scan(head[lane + 64 * i]));
                    carry = (uint32_t)__builtin_amdgcn_readlane((int)k, 63);
                    const uint32_t j = (uint32_t)(64 * i + lane);
                    if (j < nrec) q[i] = s8[srcs[k] + j];
                }
            }
            double bp[RB];
            uint32_t pp[RB];   // pixel | (polarity + bias) is not needed: the polarity is kept whole in its own bits
            int pol[RB];
#pragma unroll
            for (int i = 0; i < RB; ++i) {
                bp[i] = 0.0; pp[i] = 0u; pol[i] = 0;
                if ((uint32_t)(64 * i + lane) < nrec) { pp[i] = px_of(q[i]); pol[i] = pol_of(q[i]); bp[i] = bin_of(q[i]); }
            }
            if (scale != 1.0) {
#pragma unroll
                for (int i = 0; i < RB; ++i) if ((uint32_t)(64 * i + lane) < nrec) touched[pp[i]] = 1;
            }
            for (int pass = 0; pass < npass; ++pass) {
#pragma unroll
                for (int i = 0; i < RB; ++i)
                    if ((uint32_t)(64 * i) < nrec) apply(pass, (uint32_t)(64 * i + lane) < nrec, pp[i], bp[i], pol[i]);
            }
        } else {
            // a larger unit: swept once per pass, four batches in flight; run by run when the runs are long enough to fill
            // batches, else 64 consecutive records of the unit with the run found per record (unit_records)
            constexpr int G = EVREP_STREAM_G;
            const bool by_run = nrec >= (uint32_t)EVREP_VOXEL_BYRUN * (uint32_t)nb;   // wave-uniform
            const uint32_t run00 = (uint32_t)beg + ((uint32_t)lane << bv.chunk_shift) + R.a0;
            const uint32_t run01 = (uint32_t)beg + ((uint32_t)(kWave + lane) << bv.chunk_shift) + R.a1;
            uint32_t *runs2 = head;   // [2][128]
            if (!by_run && nb > kBsChainBlocks) {
                runs2[lane] = R.pre0; runs2[128 + lane] = src0;
                if (wide) { runs2[kWave + lane] = R.pre1; runs2[128 + kWave + lane] = src1; }
            }
            wave_phase();
            auto src_of = [&](uint32_t j) -> uint32_t {
                if (nb <= kBsChainBlocks) {
                    uint32_t sx = (uint32_t)__builtin_amdgcn_readlane((int)src0, 0);
                    uint32_t prev = sx;
                    for (int k = 1; k < nb; ++k) {
                        const uint32_t pk = (uint32_t)__builtin_amdgcn_readlane((int)R.pre0, k);
                        const uint32_t sk = (uint32_t)__builtin_amdgcn_readlane((int)src0, k);
                        sx += (j >= pk) ? sk - prev : 0u;
                        prev = sk;
                    }
                    return sx + j;
                }
                uint32_t lo = 0, hi = (uint32_t)nb;
#pragma unroll
                for (int step = 0; step < 7; ++step) {
                    const uint32_t mid = (lo + hi) >> 1;
                    const bool go = hi - lo > 1 && runs2[mid] <= j;
                    if (hi - lo > 1) { if (go) lo = mid; else hi = mid; }
                }
                return runs2[128 + lo] + j;
            };
            for (int pass = 0; pass < npass; ++pass) {
                int rk_ = 0;
                uint32_t ro_ = 0;
                bool more = true;
                uint32_t run_len = 0u, run_base = 0u;   // of run rk_
                auto load_run = [&]() {
                    run_len = 0u;
                    if (rk_ < nb) { run_len = R.pick(R.len0, R.len1, rk_); run_base = R.pick(run00, run01, rk_); }
                };
                if (by_run) load_run();
                while (more) {
                    Rec8 q[G];
                    uint32_t bcnt[G];
#pragma unroll
                    for (int sl = 0; sl < G; ++sl) {
                        bcnt[sl] = 0u;
                        uint32_t addr = 0u;
                        if (by_run) {
                            while (rk_ < nb && ro_ >= run_len) { ++rk_; ro_ = 0; load_run(); }   // (the current run's length and base are kept in scalars)
                            if (rk_ < nb) {
                                addr = run_base + ro_ + (uint32_t)lane;
                                bcnt[sl] = min(run_len - ro_, (uint32_t)kWave);
                                ro_ += kWave;
                            }
                        } else {
                            const uint32_t j0 = ro_;
                            if (j0 < nrec) {
                                bcnt[sl] = min(nrec - j0, (uint32_t)kWave);
                                if ((uint32_t)lane < bcnt[sl]) addr = src_of(j0 + (uint32_t)lane);
                                ro_ += kWave;
                            }
                        }
                        q[sl] = make_uint2(0u, 0u);
                        if ((uint32_t)lane < bcnt[sl]) q[sl] = s8[addr];
                    }
                    if (by_run) {
                        while (rk_ < nb && ro_ >= run_len) { ++rk_; ro_ = 0; load_run(); }
                        more = rk_ < nb;
                    } else {
                        more = ro_ < nrec;
                    }
#pragma unroll
                    for (int sl = 0; sl < G; ++sl) {
                        if (bcnt[sl] == 0u) break;   // uniform
                        const bool have = (uint32_t)lane < bcnt[sl];
                        uint32_t px = 0u; int pol = 0; double bp = 0.0;
                        if (have) { px = px_of(q[sl]); pol = pol_of(q[sl]); bp = bin_of(q[sl]); }
                        if (pass == 0 && scale != 1.0 && have) touched[px] = 1;
                        apply(pass, have, px, bp, pol);
                    }
                }
            }
        }
    }
    wave_phase();
    if (scale != 1.0) {   // the pixels that hold records are scaled, bin by bin (k_voxel's reduce); empty ones stay +0
        const uint32_t inv = (65536u + (uint32_t)bins - 1u) / (uint32_t)bins;   // v / bins == (v * inv) >> 16 for v < 4096
        for (int v = lane; v < ncell; v += kWave) {
            const uint32_t px = ((uint32_t)v * inv) >> 16;
            if (touched[px]) acc[v] = acc[v] * scale;
        }
        wave_phase();
    }
    tile_store(acc, ncell, dst);
#ifdef EVREP_TIMING
    if (lane == 0 && bv.dbg) { bv.dbg[(size_t)uid * 8 + 7] = (unsigned long long)vs_t0; bv.dbg[(size_t)uid * 8 + 5] = (unsigned long long)((long long)wall_clock64() - vs_t0); bv.dbg[(size_t)uid * 8 + 6] = nrec; }
#endif
}

// --------------------------------------------------------------------------------------------
// A2 (r06): BURST units of the voxel grid -- several waves per unit.
// A unit of a clustered window whose records come in bursts on a few pixels (a moving blob) costs the stream as many election rounds
// as its hottest pixel holds records of each batch -- 540 rounds of ~0.2 us of a lone wave for ONE Gen1 circle unit of 2 100 records,
// the launch's tail -- while the chain that has to be sequential is short: the terms of ONE cell (pixel, bin), 60 of them.  The main
// wave hands a unit of more than uc.stage records and more than three times its window's average over untouched (item kHotWhole); a
// workgroup of kVhWaves waves takes it, pass by pass, in chunks of kVhChunk records (array order):
//   count   every wave sweeps its contiguous share of the chunk: the record's cell of this pass (its digest: the float64 division),
//           LDS atomicAdd on its OWN row of cell counters;
//   scan    cell starts and per-(wave, cell) cursors: an earlier wave's terms of a cell lie in front of a later wave's;
//   place   the same sweep again: ranks inside a batch by a ballot match over the cell index, the record's 8-byte TERM (w * p) to its
//           place of the chunk's stage -- stable: every cell's terms stay in array order;
//   fold    one thread per cell adds the cell's terms onto the cell, four LDS reads in flight: the chain is one float64 add per term.
// Pass 1's terms go on top of pass 0's of the WHOLE unit, so the passes are the outer loop.  The tile leaves as one burst.
// LDS: cells [npixa * bins] f64 | cnt [kVhWaves][ncella] | seg [ncella + 1] | rt [2][128] | touched [npixa] | tmp | stage [kVhChunk] f64
#ifndef EVREP_VH_WAVES
#define EVREP_VH_WAVES 16
#endif
constexpr int kVhWaves = EVREP_VH_WAVES, kVhThreads = kVhWaves * kWave, kVhChunk = 4096, kVhGrid = 256, kVhMaxBins = 8;
static_assert(kVhGrid % kHotLists == 0, "every sublist is worked off by kVhGrid / kHotLists workgroups");
__host__ __device__ inline size_t voxel_hot_lds_bytes(int bins, int npixa) {
    const size_t ncella = (size_t)npixa * bins;
    return align16(ncella * 8) + (size_t)kVhWaves * ncella * 4 + align16((ncella + 1) * 4) + 256 * 4 + align16((size_t)npixa) + 64 +
           (size_t)kVhChunk * 8;
}
#ifdef EVREP_TU_BUILDERS   // (compiled by the one translation unit that launches it)
static __global__ __launch_bounds__(kVhThreads) void k_voxel_hot(const int4 *__restrict__ ev, BinView bv, const int64_t *__restrict__ off,
                                                              int H, int W, int nchunk, UnitCfg uc, int bins, int mode, double scale,
                                                              const int64_t *__restrict__ t_range, const double *__restrict__ tnorm,
                                                              double *__restrict__ out) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const uint32_t l = blockIdx.x % kHotLists, capl = hot_sublist_cap(bv.hot_cap);
    const uint32_t nraw = (uint32_t)__builtin_amdgcn_readfirstlane((int)bv.hot[l * 16]);
    if (nraw == 0u) return;
    const uint32_t nitems = min(nraw, capl);
    const uint32_t *items = bv.hot + kHotHdrWords + (size_t)l * capl;
    const int npixa = (uc.span + uc.merge) * kChunkPx;
    const int ncella = npixa * bins;
    double *acc = reinterpret_cast<double *>(smem);
    uint32_t *cnt = reinterpret_cast<uint32_t *>(smem + align16((size_t)ncella * 8));                        // [kVhWaves][ncella]
    uint32_t *seg = cnt + kVhWaves * ncella;                                                                  // [ncella + 1]
    uint32_t *rt = reinterpret_cast<uint32_t *>(reinterpret_cast<unsigned char *>(seg) + align16((size_t)(ncella + 1) * 4));   // [2][128]
    unsigned char *touched = reinterpret_cast<unsigned char *>(rt + 256);
    uint32_t *tmp = reinterpret_cast<uint32_t *>(touched + align16((size_t)npixa));                          // [16]
    double *stage = reinterpret_cast<double *>(tmp + 16);
    for (uint32_t it = blockIdx.x / kHotLists; it < nitems; it += gridDim.x / kHotLists) {
        const int item = __builtin_amdgcn_readfirstlane((int)items[it]);
        if (item >= 0 && item % kHotCodes == kHotWhole) {
            const int uid = item / kHotCodes;
#ifdef EVREP_TIMING   // experiment builds: 10 ns ticks since the item started, cumulative per phase: 0 tables, 1 count, 2 scan, 3 place, 4 fold, 5 total, 6 records
            const long long tm0 = (long long)wall_clock64();
            long long tph[5] = {0, 0, 0, 0, 0}, tlast = tm0;
            unsigned long long *dbg = bv.dbg ? bv.dbg + 8 * (size_t)uid : nullptr;
#define VH_PH(i_) do { const long long t_ = (long long)wall_clock64(); tph[i_] += t_ - tlast; tlast = t_; } while (0)
#else
#define VH_PH(i_) do {} while (0)
#endif
            int chunk, nch;
            const ChunkGeom g = unit_geom(H, W, nchunk, uc, chunk, nch, uid);
            const int b = g.b;
            const int64_t beg = off[b];
            const int64_t n_win = off[b + 1] - beg;
            int tz0 = 0, tz1 = 0;
            if (n_win > 0) { tz0 = ev[beg].z; tz1 = ev[beg + n_win - 1].z; }
            const int NK = H * nchunk, klo = g.row * nchunk + chunk;
            const StreamRuns R = stream_runs(bv, b, n_win, NK, klo, klo + nch);   // (every wave reads the unit's run tables itself)
            const int nb = R.nb;
            const uint32_t nrec = nb > 0 ? R.nrec : 0u;
            const int ncell = g.npix * bins;
            {
                uint4 *z = reinterpret_cast<uint4 *>(acc);
                for (int v = tid; v * 2 < ncell; v += kVhThreads) z[v] = make_uint4(0u, 0u, 0u, 0u);
                for (int v = tid; v < npixa; v += kVhThreads) touched[v] = 0;
            }
            const uint32_t src0 = (uint32_t)beg + ((uint32_t)lane << bv.chunk_shift) + R.a0 - R.pre0;
            const uint32_t src1 = (uint32_t)beg + ((uint32_t)(kWave + lane) << bv.chunk_shift) + R.a1 - R.pre1;
            if (wv == 0 && nb > kBsChainBlocks) {
                rt[lane] = R.pre0; rt[128 + lane] = src0;
                if (nb > kWave) { rt[kWave + lane] = R.pre1; rt[128 + kWave + lane] = src1; }
            }
            double *dst = out + (((size_t)b * H + g.row) * (size_t)W + g.c0) * bins;
            double t0 = 0.0, den = 1.0;
            if (n_win > 0) { t0 = (double)tz0; den = (double)tz1 - t0; }
            if (t_range) { t0 = (double)t_range[2 * b]; den = (double)(t_range[2 * b + 1] - t_range[2 * b]); }
            const double *tw = tnorm ? tnorm + beg : nullptr;
            const int4 *evw = ev + beg;
            const int c0 = g.c0;
            const Rec8 *__restrict__ s8 = reinterpret_cast<const Rec8 *>(bv.sorted);
            const double lowlim = mode == 2 ? -1.0 : -0.0;
            auto src_of = [&](uint32_t j) -> uint32_t {    // the address of record j of the unit in the block runs
                if (nb <= kBsChainBlocks) {
                    uint32_t sx = (uint32_t)__builtin_amdgcn_readlane((int)src0, 0);
                    uint32_t prev = sx;
                    for (int k = 1; k < nb; ++k) {
                        const uint32_t pk = (uint32_t)__builtin_amdgcn_readlane((int)R.pre0, k);
                        const uint32_t sk = (uint32_t)__builtin_amdgcn_readlane((int)src0, k);
                        sx += (j >= pk) ? sk - prev : 0u;
                        prev = sk;
                    }
                    return sx + j;
                }
                uint32_t lo = 0, hi = (uint32_t)nb;
#pragma unroll
                for (int step = 0; step < 7; ++step) {
                    const uint32_t mid = (lo + hi) >> 1;
                    const bool go = hi - lo > 1 && rt[mid] <= j;
                    if (hi - lo > 1) { if (go) lo = mid; else hi = mid; }
                }
                return rt[128 + lo] + j;
            };
            // a record's cell and term of `pass` (k_voxel's digest + the body of its reduce); false: no contribution
            auto term = [&](int pass, const Rec8 &q, uint32_t &px, uint32_t &cell, double &wp) -> bool {
                px = ((q.y & 511u) - (uint32_t)c0) & 511u;
                const uint32_t p2 = (q.y >> 9) & 3u;
                int p = (int)p2 - 1;
                if (p2 == 3u) p = evw[q.y >> 11].w;
                double bpos;
                if (tw) bpos = (double)(bins - 1) * tw[q.y >> 11];
                else {
                    const int t = (int)q.x;
                    if (mode == 2) { const int64_t num = (int64_t)(bins - 1) * ((int64_t)t - (int64_t)t0); bpos = (double)num / (den == 0.0 ? 1.0 : den); }
                    else if (mode == 0) { const double tn = ((double)t - t0) / den; bpos = (double)(bins - 1) * tn; }
                    else { const double num = (double)bins * ((double)t - t0); bpos = num / den; }
                }
                double pd = (double)p;
                if (mode == 1 && p == 0) pd = -1.0;
                if (!((bpos > lowlim && bpos < 1.0e9) || bpos == 0.0)) return false;
                const int bi = (int)bpos;
                const int blim = bi + pass;
                if (blim >= bins) return false;
                double wgt;
                if (mode == 2) wgt = 1.0;
                else if (mode == 0) wgt = 1.0 - fabs((double)blim - bpos);
                else { const double dts = bpos - (double)bi; wgt = pass ? dts : 1.0 - dts; }
                wp = wgt * pd;
                cell = px * (uint32_t)bins + (uint32_t)blim;
                return true;
            };
            const int nbits = 32 - __builtin_clz((unsigned)ncella - 1u);
            const int npass = mode == 2 ? 1 : 2;
            const uint32_t nchunks = (nrec + (uint32_t)kVhChunk - 1u) / (uint32_t)kVhChunk;
            uint32_t *mycnt = cnt + wv * ncella;
            volatile uint32_t *vcnt = mycnt;
            __syncthreads();
            VH_PH(0);
            for (int pass = 0; pass < npass; ++pass) {
                for (uint32_t ch = 0; ch < nchunks; ++ch) {
                    const uint32_t lo = ch * (uint32_t)kVhChunk, hi = min(nrec, lo + (uint32_t)kVhChunk);
                    const uint32_t piece = (((hi - lo + (uint32_t)kVhWaves - 1u) / (uint32_t)kVhWaves) + 63u) & ~63u;
                    const uint32_t mlo = min(hi, lo + (uint32_t)wv * piece), mhi = min(hi, mlo + piece);
                    for (int v = tid; v < kVhWaves * ncella; v += kVhThreads) cnt[v] = 0u;
                    __syncthreads();
                    constexpr int G = 4;
                    for (uint32_t j0 = mlo; j0 < mhi; j0 += (uint32_t)(G * kWave)) {
                        Rec8 q[G];
#pragma unroll
                        for (int sl = 0; sl < G; ++sl) {
                            const uint32_t j = j0 + (uint32_t)(sl * kWave + lane);
                            q[sl] = make_uint2(0u, 0u);
                            if (j < mhi) q[sl] = s8[src_of(j)];
                        }
#pragma unroll
                        for (int sl = 0; sl < G; ++sl) {
                            if (j0 + (uint32_t)(sl * kWave + lane) < mhi) {
                                uint32_t px, cell; double wp;
                                const bool ok = term(pass, q[sl], px, cell, wp);
                                if (pass == 0) touched[px] = 1;
                                if (ok) atomicAdd(&mycnt[cell], 1u);
                            }
                        }
                    }
                    __syncthreads();
                    VH_PH(1);
                    {   // cell starts of the chunk; cursors of every (wave, cell): `cpt` consecutive cells per thread
                        const int cpt = (ncella + kVhThreads - 1) / kVhThreads;
                        const int k0 = tid * cpt;
                        uint32_t local = 0;
                        for (int k = 0; k < cpt; ++k)
                            if (k0 + k < ncella) for (int w2 = 0; w2 < kVhWaves; ++w2) local += cnt[w2 * ncella + k0 + k];
                        uint32_t total;
                        uint32_t run = block_exclusive_scan<kVhWaves>(local, tmp, &total);
                        for (int k = 0; k < cpt; ++k) {
                            if (k0 + k < ncella) {
                                seg[k0 + k] = run;
                                for (int w2 = 0; w2 < kVhWaves; ++w2) { const uint32_t c = cnt[w2 * ncella + k0 + k]; cnt[w2 * ncella + k0 + k] = run; run += c; }
                            }
                        }
                        if (tid == 0) seg[ncella] = total;
                    }
                    __syncthreads();
                    VH_PH(2);
                    for (uint32_t j0 = mlo; j0 < mhi; j0 += (uint32_t)(G * kWave)) {
                        Rec8 q[G];
#pragma unroll
                        for (int sl = 0; sl < G; ++sl) {
                            const uint32_t j = j0 + (uint32_t)(sl * kWave + lane);
                            q[sl] = make_uint2(0u, 0u);
                            if (j < mhi) q[sl] = s8[src_of(j)];
                        }
#pragma unroll
                        for (int sl = 0; sl < G; ++sl) {
                            if (j0 + (uint32_t)(sl * kWave) >= mhi) break;   // uniform
                            const bool have = j0 + (uint32_t)(sl * kWave + lane) < mhi;
                            uint32_t px = 0u, cell = 0u; double wp = 0.0;
                            const bool valid = have && term(pass, q[sl], px, cell, wp);
                            if (!__any(valid)) continue;
                            if (!valid) cell = 0u;
                            uint32_t rk; bool last;
                            wave_match(cell, nbits, valid, lane, rk, last);
                            uint32_t pos = 0;
                            if (valid) { pos = vcnt[cell] + rk; stage[pos] = wp; }
                            __builtin_amdgcn_wave_barrier();
                            if (valid && last) vcnt[cell] = pos + 1;
                            __builtin_amdgcn_wave_barrier();
                        }
                    }
                    __syncthreads();
                    VH_PH(3);
                    // fold: a thread per cell, the cell's terms in array order, four reads in flight
                    for (int cell = tid; cell < ncell; cell += kVhThreads) {
                        const uint32_t st = seg[cell], en = seg[cell + 1];
                        if (en > st) {
                            double run = acc[cell];
                            uint32_t j = st;
                            for (; j + 4u <= en; j += 4u) {
                                const double a0 = stage[j], a1 = stage[j + 1], a2 = stage[j + 2], a3 = stage[j + 3];
                                run = run + a0; run = run + a1; run = run + a2; run = run + a3;
                            }
                            for (; j < en; ++j) run = run + stage[j];
                            acc[cell] = run;
                        }
                    }
                    __syncthreads();
                    VH_PH(4);
                }
            }
            if (scale != 1.0) {   // the pixels that hold records are scaled, bin by bin (k_voxel's reduce); empty ones stay +0
                const uint32_t inv = (65536u + (uint32_t)bins - 1u) / (uint32_t)bins;
                for (int v = tid; v < ncell; v += kVhThreads) {
                    const uint32_t px = ((uint32_t)v * inv) >> 16;
                    if (touched[px]) acc[v] = acc[v] * scale;
                }
                __syncthreads();
            }
            {
                const int nvec = ((reinterpret_cast<uintptr_t>(dst) & 15u) == 0) ? ncell / 2 : 0;
                const float4 *s4 = reinterpret_cast<const float4 *>(acc);
                float4 *d4 = reinterpret_cast<float4 *>(dst);
                for (int v = tid; v < nvec; v += kVhThreads) d4[v] = s4[v];
                for (int e = nvec * 2 + tid; e < ncell; e += kVhThreads) dst[e] = acc[e];
            }
#ifdef EVREP_TIMING
            if (tid == 0 && dbg) {
                for (int q_ = 0; q_ < 5; ++q_) dbg[q_] = (unsigned long long)tph[q_];
                dbg[5] = (unsigned long long)((long long)wall_clock64() - tm0); dbg[6] = (unsigned long long)nrec * 100ull;
            }
#endif
        }
        __syncthreads();
    }
    if (tid == 0) {
        const uint32_t ticket = atomicAdd(&bv.hot[(kHotLists + l) * 16], 1u);
        if (ticket + 1u == gridDim.x / kHotLists) {   // the sublist's last workgroup: every other one has read its items
            bv.hot[(kHotLists + l) * 16] = 0u;
            bv.hot[l * 16] = 0u;
        }
    }
}
#endif

// --------------------------------------------------------------------------------------------
// F4: n_imagenet's per-polarity accumulators (n_imagenet/real_cnn_model/data/imagenet.py:169-511,841-871)
// --------------------------------------------------------------------------------------------
struct PolStatParams {
    int32_t C;
    int32_t pol[EVREP_MAX_CHANNELS], stat[EVREP_MAX_CHANNELS];
    double tau;
};

// grid (ceil(nchunk/span), H, B), 64 threads.  tnorm[off[b] + rank] = the record's normalised float64 time.
// (6 waves per SIMD asked for: 110 -> 80 VGPRs with 52 bytes of scratch, 81 -> 64 us at 32 x 50 000 events, 640x480x6;
// the same hint does nothing for EventStack / TORE, which sit at the store ceiling, and hurts k_voxel, r02)
// SM (r05b): the MAIN launch runs the order-free sweep itself (dense windows: every unit is beyond the record stage and the hot launch
// is the slower place for bulk work; the host gives this instance a stage that holds the words) -- nothing is deferred, no hot launch
#ifndef EVREP_PS_WAVES
#define EVREP_PS_WAVES 6
#endif
template <int CM, bool HOT = false, bool SM = false>
__global__ __launch_bounds__(kWave, (HOT || SM) ? 4 : EVREP_PS_WAVES) void k_polstats(BinView bv,
                                                   const int64_t *__restrict__ off, const double *__restrict__ tnorm,
                                                   PolStatParams P, int H, int W, int nchunk, UnitCfg uc,
                                                   float *__restrict__ out) {
    extern __shared__ __align__(16) unsigned char smem[];
    run_units<HOT>(bv, [&](int uid, int part) {
        const int C = P.C;
        WaveLds<float, HOT> w(smem, C, (uc.span + uc.merge) * kChunkPx, uc.stage, uc.partpx);
        w.arm(uc.hold);
        ChunkGeom g;
        // Every statistic here is order-free (counts, maxima and minima of the caller's per-event time, per polarity class), so
        // a unit beyond the record stage needs no order at all (r05b, unit_records' Split, IN_HOT): the main wave hands it to the
        // hot launch whole, where ONE sweep keeps fourteen words per pixel by LDS atomics --
        //   0: #(p > 0) | #(p < 0) << 16     1: #(p == 0)     then per class (p > 0, p < 0, p == 0) two 64-bit words:
        //   max key(t_n), max ~key(t_n)  (key = the order-preserving integer image of a float64; ~key turns the minimum into a maximum,
        //   so that zero-filled words are the identity of both)
        // -- and the pixels are emitted from them (u.part == -5).  No record is kept.
        constexpr int kPsWords = 14;
        auto dkey = [](double v) -> unsigned long long {
            const unsigned long long b = (unsigned long long)__double_as_longlong(v);
            return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
        };
        auto dval = [](unsigned long long k) -> double {
            const unsigned long long b = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
            return __longlong_as_double((long long)b);
        };
        int chunk0_;
        const int b0 = unit_geom(H, W, nchunk, uc, chunk0_, uid).b;
        const double *tw0 = tnorm + off[b0];
        const int4 *evw0 = bv.ev + off[b0];
        UnitRecs u;
        if constexpr (HOT || SM) {
            uint32_t *words = reinterpret_cast<uint32_t *>(w.tile);
            auto yes = []() -> bool { return true; };
            auto ppre = [&](const Rec8 &q) -> uint2 {   // the record's time, gathered for every batch of a round before the first atomic
                const double tn = tw0[q.y >> 11];
                return make_uint2((uint32_t)__double2loint(tn), (uint32_t)__double2hiint(tn));
            };
            auto psf = [&](uint32_t px, const Rec8 &q, uint2 &, const uint2 &aux) -> bool {
                const uint32_t p2 = (q.y >> 9) & 3u;
                int p = (int)p2 - 1;
                if (p2 == 3u) p = evw0[q.y >> 11].w;   // an escaped polarity value: only its sign counts
                const int cls = p > 0 ? 0 : (p < 0 ? 1 : 2);
                uint32_t *wd = words + px * (uint32_t)kPsWords;
                if (cls == 2) atomicAdd(wd + 1, 1u); else atomicAdd(wd, cls == 0 ? 1u : 0x10000u);
                const unsigned long long k = dkey(__hiloint2double((int)aux.y, (int)aux.x));
                unsigned long long *w64 = reinterpret_cast<unsigned long long *>(wd + 2 + 4 * cls);
                atomicMax(w64, k);
                atomicMax(w64 + 1, ~k);
                return false;
            };
            auto pmerge = [](const uint32_t *m, uint32_t *gw) {
                if (m[0]) atomicAdd(gw, m[0]);
                if (m[1]) atomicAdd(gw + 1, m[1]);
                const unsigned long long *m64 = reinterpret_cast<const unsigned long long *>(m + 2);
                unsigned long long *g64 = reinterpret_cast<unsigned long long *>(gw + 2);
#pragma unroll
                for (int k = 0; k < 6; ++k) if (m64[k]) atomicMax(g64 + k, m64[k]);
            };
            u = unit_front<float, HOT, false, NoVisit>(bv, off, H, W, nchunk, uc, w, g, uid, part, NoVisit(),
                                                       unit_split_full<HOT>(yes, psf, yes, kPsWords, pmerge, ppre));
        } else {
            auto never = []() -> bool { return false; };
            auto nof = [](uint32_t, const Rec8 &, uint2 &, const uint2 &) -> bool { return false; };
            u = unit_front<float, HOT, false, NoVisit>(bv, off, H, W, nchunk, uc, w, g, uid, part, NoVisit(),
                                                       unit_split<true>(never, nof, never, kPsWords, (uc.xflags & 2) ? 0u : kStEscaped));
        }
        if (u.deferred) return;
        float *dst = out + (((size_t)g.b * H + g.row) * (size_t)W + g.c0) * C;
        const int lane = threadIdx.x;
        const double *tw = tnorm + off[g.b];
        // empty pixels: 0, except EXP channels = exp(-(1 - 0)/tau)  (imagenet.py:463,466)
        bool any_bg = false;
#pragma unroll
        for (int c = 0; c < CM; ++c) any_bg |= (c < C && P.stat[c] == EVREP_PS_EXP);
        if (any_bg && u.part != -5) {   // (a unit swept by the split keeps its words where the background vector lies)
            if (lane < CM) {
                float v = 0.0f;
                if (lane < C && P.stat[lane] == EVREP_PS_EXP) v = (float)exp_neg_range(-(1.0 - 0.0) / P.tau);
                w.bg[lane] = v;
            }
            wave_phase();
        }
        const float *bg = any_bg ? w.bg : nullptr;
        // a pixel's statistics -> its C values
        auto channels = [&](int n_any, int n_pos, int n_neg, double mx_any, double mx_pos, double mx_neg, double mn_any, double mn_pos,
                            double mn_neg, float(&vals)[CM]) {
#pragma unroll
            for (int c = 0; c < CM; ++c) {
                float v = 0.0f;
                if (c < C) {
                    const int k = P.pol[c], st = P.stat[c];
                    const int n = k == EVREP_PS_POS ? n_pos : (k == EVREP_PS_NEG ? n_neg : n_any);
                    const double mx = k == EVREP_PS_POS ? mx_pos : (k == EVREP_PS_NEG ? mx_neg : mx_any);
                    const double mn = k == EVREP_PS_POS ? mn_pos : (k == EVREP_PS_NEG ? mn_neg : mn_any);
                    if (st == EVREP_PS_COUNT) v = (float)n;
                    else if (st == EVREP_PS_TMAX) v = n ? (float)mx : 0.0f;
                    else if (st == EVREP_PS_TMIN) v = n ? (float)mn : 0.0f;
                    else if (st == EVREP_PS_FLAG) v = n ? 1.0f : 0.0f;
                    else if (st == EVREP_PS_EXP) v = (float)exp_neg_range(-(1.0 - (n ? mx : 0.0)) / P.tau);
                    else if (st == EVREP_PS_SIGNED) v = (float)n_pos - (float)n_neg;
                }
                vals[c] = v;
            }
        };
        if constexpr (HOT || SM) {
            if (u.part == -5) {   // wave-uniform: the unit was swept by the split; every pixel's statistics wait in its words
                const uint32_t *words = reinterpret_cast<const uint32_t *>(w.tile);
                const bool vec = (C % 4) == 0 && (reinterpret_cast<uintptr_t>(dst) & 15u) == 0;
                for (int pt = 0; pt * kWave < g.npix; ++pt) {
                    const int np = min(kWave, g.npix - pt * kWave);
                    const uint32_t *wd = words + (uint32_t)(pt * kWave + lane) * (uint32_t)kPsWords;
                    float vals[CM];
                    if (lane < np) {
                        const int n_pos = (int)(wd[0] & 0xffffu), n_neg = (int)(wd[0] >> 16), n_zero = (int)wd[1];
                        const unsigned long long *w64 = reinterpret_cast<const unsigned long long *>(wd + 2);
                        const double mxp = n_pos ? dval(w64[0]) : 0.0, mnp = n_pos ? dval(~w64[1]) : 0.0;
                        const double mxn = n_neg ? dval(w64[2]) : 0.0, mnn = n_neg ? dval(~w64[3]) : 0.0;
                        const double mxz = n_zero ? dval(w64[4]) : 0.0, mnz = n_zero ? dval(~w64[5]) : 0.0;
                        double mxa = 0.0, mna = 0.0;
                        bool have = false;
                        if (n_pos) { mxa = mxp; mna = mnp; have = true; }
                        if (n_neg) { mxa = have ? fmax(mxa, mxn) : mxn; mna = have ? fmin(mna, mnn) : mnn; have = true; }
                        if (n_zero) { mxa = have ? fmax(mxa, mxz) : mxz; mna = have ? fmin(mna, mnz) : mnz; }
                        channels(n_pos + n_neg + n_zero, n_pos, n_neg, mxa, mxp, mxn, mna, mnp, mnn, vals);
                        store_pixel<float, CM>(dst + ((size_t)pt * kWave + lane) * C, vals, C, vec);
                    }
                }
                return;
            }
        }
        auto reduce = [&](uint32_t jb, uint32_t je, auto get, float(&vals)[CM]) {
            int n_any = 0, n_pos = 0, n_neg = 0;
            double mx_any = 0.0, mx_pos = 0.0, mx_neg = 0.0, mn_any = 0.0, mn_pos = 0.0, mn_neg = 0.0;
            for (uint32_t j = jb; j < je; ++j) {
                const Rec e = get(j);   // digest: {t_n lo, t_n hi, rank, p}
                const double tn = __hiloint2double(e.y, e.x);
                if (n_any == 0 || tn > mx_any) mx_any = tn;
                if (n_any == 0 || tn < mn_any) mn_any = tn;
                ++n_any;
                if (e.w > 0) {
                    if (n_pos == 0 || tn > mx_pos) mx_pos = tn;
                    if (n_pos == 0 || tn < mn_pos) mn_pos = tn;
                    ++n_pos;
                } else if (e.w < 0) {
                    if (n_neg == 0 || tn > mx_neg) mx_neg = tn;
                    if (n_neg == 0 || tn < mn_neg) mn_neg = tn;
                    ++n_neg;
                }
            }
            channels(n_any, n_pos, n_neg, mx_any, mx_pos, mx_neg, mn_any, mn_pos, mn_neg, vals);
        };
        // the record's normalised time is gathered from the caller's array by the DIGEST -- one load per record, all lanes at once,
        // staged in place of the fields the walks do not need -- instead of one dependent global load per step of the divergent
        // segment walks (r03)
        auto digest = [&](const Rec &r) -> Rec {
            const double tn = tw[r.y];
            return make_int4(__double2loint(tn), __double2hiint(tn), r.y, r.w);
        };
        emit_chunk<float, CM, HOT>(u, digest, g.row * W + g.c0, g.npix, C, dst, w, bg, reduce);
    });
}

// --------------------------------------------------------------------------------------------
// F4 (r06), after the key-sorted pass: the n_imagenet accumulators as a STREAM.  Every statistic is order-free (counts, maxima and
// minima of the caller's per-event time per polarity class): sixteen words per pixel of the unit in LDS (three full-width counts,
// six 64-bit extremes as order-preserving keys: k_polstats' words), bumped by LDS atomics in ONE sweep of the unit's records --
// whatever it holds -- then every pixel's C values are formed from its words, 64 pixels at a time, and written over the words
// already consumed (C * 4 <= 64 bytes per pixel: a batch's values end in front of the next batch's words), so that the unit leaves
// as one coalesced burst.  No grouping, no stage, no hot launch.
// LDS: words [npixa * 16 (K32: 9)] u32 (then the tile) | head [64 * RB] | srcs [128]
// K32 (no EXP channel, C <= 8): the extremes as 32-bit keys of the time ROUNDED to float32 -- the outputs are (float) max / (float) min
// of the float64 times, and the rounding is monotone, so the maximum of the rounded times is the rounded maximum: nine words per
// pixel instead of sixteen, 32-bit LDS atomics, a three-instruction decode per extreme (sparse 640x480 windows: 102 -> us, the
// kernel is bound by its instruction count there).
__host__ __device__ inline int polstats_stream_words(bool k32) { return k32 ? 9 : 16; }   // #(p > 0), #(p < 0), #(p == 0), [pad,] per class max key(t_n), max ~key(t_n)
__host__ __device__ inline size_t polstats_stream_lds_bytes(int npixa, int rb, bool k32) {
    return align16((size_t)npixa * polstats_stream_words(k32) * 4) + (size_t)(64 * rb) * 4 + 128 * 4;
}
template <int CM, int RB, bool K32>
__global__ __launch_bounds__(kWave, 6) void k_polstats_stream(BinView bv, const int64_t *__restrict__ off, const double *__restrict__ tnorm,
                                                          PolStatParams P, int H, int W, int nchunk, UnitCfg uc, float *__restrict__ out) {
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr int NW = K32 ? 9 : 16;
    const int lane = threadIdx.x;
    const int C = P.C;
    const int uid = chunk_unit((int)(gridDim.x * gridDim.y * gridDim.z));
    int chunk, nch;
    const ChunkGeom g = unit_geom(H, W, nchunk, uc, chunk, nch, uid);
    const int b = g.b;
    const int64_t beg = off[b];
    const int64_t n_win = off[b + 1] - beg;
    const int npixa = (uc.span + uc.merge) * kChunkPx;
    uint32_t *words = reinterpret_cast<uint32_t *>(smem);
    uint32_t *head = reinterpret_cast<uint32_t *>(smem + align16((size_t)npixa * NW * 4));
    uint32_t *srcs = head + 64 * RB;
    {
        uint4 *z = reinterpret_cast<uint4 *>(words);
        const int nvec = (g.npix * NW + 3) / 4;
        for (int v = lane; v < nvec; v += kWave) z[v] = make_uint4(0u, 0u, 0u, 0u);
    }
    const double *tw0 = tnorm + beg;
    const int4 *evw0 = bv.ev + beg;
    const int c0 = g.c0;
    auto dkey = [](double v) -> unsigned long long {
        const unsigned long long bts = (unsigned long long)__double_as_longlong(v);
        return (bts >> 63) ? ~bts : (bts | 0x8000000000000000ull);
    };
    auto dval = [](unsigned long long k) -> double {
        const unsigned long long bts = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
        return __longlong_as_double((long long)bts);
    };
    auto fkey = [](float v) -> uint32_t { const uint32_t bts = __float_as_uint(v); return (bts >> 31) ? ~bts : (bts | 0x80000000u); };
    auto fval = [](uint32_t k) -> float { return __uint_as_float((k >> 31) ? (k & 0x7fffffffu) : ~k); };
    wave_phase();
    stream_unit_records<RB>(bv, b, beg, n_win, H * nchunk, g.row * nchunk + chunk, g.row * nchunk + chunk + nch, head, srcs,
        [&](const Rec8 &q) -> uint2 {   // the record's time from the caller's array: every gather of a group in flight together
            const double tn = tw0[q.y >> 11];
            if constexpr (K32) return make_uint2(__float_as_uint((float)tn), 0u);
            return make_uint2((uint32_t)__double2loint(tn), (uint32_t)__double2hiint(tn));
        },
        [&](bool have, const Rec8 &q, const uint2 &aux) {
            if (!have) return;
            const uint32_t px = ((q.y & 511u) - (uint32_t)c0) & 511u;
            const uint32_t p2 = (q.y >> 9) & 3u;
            int p = (int)p2 - 1;
            if (p2 == 3u) p = evw0[q.y >> 11].w;   // an escaped polarity value: only its sign counts
            const int cls = p > 0 ? 0 : (p < 0 ? 1 : 2);
            uint32_t *wd = words + px * (uint32_t)NW;
            atomicAdd(wd + cls, 1u);
            if constexpr (K32) {
                const uint32_t k = fkey(__uint_as_float(aux.x));
                atomicMax(wd + 3 + 2 * cls, k);
                atomicMax(wd + 4 + 2 * cls, ~k);
            } else {
                const unsigned long long k = dkey(__hiloint2double((int)aux.y, (int)aux.x));
                unsigned long long *w64 = reinterpret_cast<unsigned long long *>(wd + 4 + 4 * cls);
                atomicMax(w64, k);
                atomicMax(w64 + 1, ~k);
            }
        });
    wave_phase();
    float *tile = reinterpret_cast<float *>(smem);
    for (int pt = 0; pt * kWave < g.npix; ++pt) {
        const int px = pt * kWave + lane;
        float vals[CM];
#pragma unroll
        for (int c = 0; c < CM; ++c) vals[c] = 0.0f;
        if (px < g.npix) {
            const uint32_t *wd = words + (uint32_t)px * (uint32_t)NW;
            const int n_pos = (int)wd[0], n_neg = (int)wd[1], n_zero = (int)wd[2];
            const int n_any = n_pos + n_neg + n_zero;
            if constexpr (K32) {
                // (float32 extremes: no EXP channel reads them)
                const float mxp = n_pos ? fval(wd[3]) : 0.0f, mnp = n_pos ? fval(~wd[4]) : 0.0f;
                const float mxn = n_neg ? fval(wd[5]) : 0.0f, mnn = n_neg ? fval(~wd[6]) : 0.0f;
                const float mxz = n_zero ? fval(wd[7]) : 0.0f, mnz = n_zero ? fval(~wd[8]) : 0.0f;
                float mxa = 0.0f, mna = 0.0f;
                bool hv = false;
                if (n_pos) { mxa = mxp; mna = mnp; hv = true; }
                if (n_neg) { mxa = hv ? fmaxf(mxa, mxn) : mxn; mna = hv ? fminf(mna, mnn) : mnn; hv = true; }
                if (n_zero) { mxa = hv ? fmaxf(mxa, mxz) : mxz; mna = hv ? fminf(mna, mnz) : mnz; }
#pragma unroll
                for (int c = 0; c < CM; ++c) {
                    float v = 0.0f;
                    if (c < C) {
                        const int k = P.pol[c], st = P.stat[c];
                        const int n = k == EVREP_PS_POS ? n_pos : (k == EVREP_PS_NEG ? n_neg : n_any);
                        const float mx = k == EVREP_PS_POS ? mxp : (k == EVREP_PS_NEG ? mxn : mxa);
                        const float mn = k == EVREP_PS_POS ? mnp : (k == EVREP_PS_NEG ? mnn : mna);
                        if (st == EVREP_PS_COUNT) v = (float)n;
                        else if (st == EVREP_PS_TMAX) v = n ? mx : 0.0f;
                        else if (st == EVREP_PS_TMIN) v = n ? mn : 0.0f;
                        else if (st == EVREP_PS_FLAG) v = n ? 1.0f : 0.0f;
                        else if (st == EVREP_PS_SIGNED) v = (float)n_pos - (float)n_neg;
                    }
                    vals[c] = v;
                }
            } else {
                const unsigned long long *w64 = reinterpret_cast<const unsigned long long *>(wd + 4);
                const double mxp = n_pos ? dval(w64[0]) : 0.0, mnp = n_pos ? dval(~w64[1]) : 0.0;
                const double mxn = n_neg ? dval(w64[2]) : 0.0, mnn = n_neg ? dval(~w64[3]) : 0.0;
                const double mxz = n_zero ? dval(w64[4]) : 0.0, mnz = n_zero ? dval(~w64[5]) : 0.0;
                double mxa = 0.0, mna = 0.0;
                bool hv = false;
                if (n_pos) { mxa = mxp; mna = mnp; hv = true; }
                if (n_neg) { mxa = hv ? fmax(mxa, mxn) : mxn; mna = hv ? fmin(mna, mnn) : mnn; hv = true; }
                if (n_zero) { mxa = hv ? fmax(mxa, mxz) : mxz; mna = hv ? fmin(mna, mnz) : mnz; }
#pragma unroll
                for (int c = 0; c < CM; ++c) {
                    float v = 0.0f;
                    if (c < C) {
                        const int k = P.pol[c], st = P.stat[c];
                        const int n = k == EVREP_PS_POS ? n_pos : (k == EVREP_PS_NEG ? n_neg : n_any);
                        const double mx = k == EVREP_PS_POS ? mxp : (k == EVREP_PS_NEG ? mxn : mxa);
                        const double mn = k == EVREP_PS_POS ? mnp : (k == EVREP_PS_NEG ? mnn : mna);
                        if (st == EVREP_PS_COUNT) v = (float)n;
                        else if (st == EVREP_PS_TMAX) v = n ? (float)mx : 0.0f;
                        else if (st == EVREP_PS_TMIN) v = n ? (float)mn : 0.0f;
                        else if (st == EVREP_PS_FLAG) v = n ? 1.0f : 0.0f;
                        else if (st == EVREP_PS_EXP) v = (float)exp_neg_range(-(1.0 - (n ? mx : 0.0)) / P.tau);
                        else if (st == EVREP_PS_SIGNED) v = (float)n_pos - (float)n_neg;
                    }
                    vals[c] = v;
                }
            }
        }
        wave_phase();   // the batch's words are in registers: its values may take their place
        if (px < g.npix) {
            float *mine = tile + (size_t)px * C;
#pragma unroll
            for (int c = 0; c < CM; ++c) if (c < C) mine[c] = vals[c];
        }
        wave_phase();
    }
    float *dst = out + (((size_t)b * H + g.row) * (size_t)W + g.c0) * C;
    tile_store(tile, g.npix * C, dst);
}

// --------------------------------------------------------------------------------------------
// F4: EST quantisation layer, forward (ev-YOLOv6/yolov6/models/learned_repr.py:143-179)
// --------------------------------------------------------------------------------------------
constexpr int kEstMaxBins = EVREP_MAX_CHANNELS / 2;

struct EstParams {
    int32_t C, nseg, nbucket, pad;
    double lo, inv_width;          // bucket = (u - lo) * inv_width
    float shift[kEstMaxBins];      // float32(i / (C - 1)), as `t - i_bin / (C - 1)` rounds it (:167)
};

// f(u) of the value MLP through its exact piecewise-linear form: segment k covers u < seg[3k] (ascending),
// f = seg[3k+1] * u + seg[3k+2].
__device__ inline float est_value(float u, const double *__restrict__ seg, const uint32_t *__restrict__ bucket,
                                  const EstParams &P) {
    const double ud = (double)u;
    int g = (int)((ud - P.lo) * P.inv_width);
    g = g < 0 ? 0 : (g >= P.nbucket ? P.nbucket - 1 : g);
    int k = (int)bucket[g];
    while (k + 1 < P.nseg && ud >= seg[3 * k]) ++k;
    return (float)(seg[3 * k + 1] * ud + seg[3 * k + 2]);
}

// grid (ceil(nchunk/span), H, B), 64 threads.  tnorm[off[b] + rank] = the record's float32 t / t.max().
template <bool HOT = false>
__global__ __launch_bounds__(kWave, HOT ? 4 : 1) void k_est(BinView bv,
                                              const int64_t *__restrict__ off, const float *__restrict__ tnorm,
                                              const double *__restrict__ seg, const uint32_t *__restrict__ bucket,
                                              EstParams P, int H, int W, int nchunk, UnitCfg uc, float *__restrict__ out) {
    extern __shared__ __align__(16) unsigned char smem[];
    run_units<HOT>(bv, [&](int uid, int part) {
        const int C2 = 2 * P.C;
        WaveLds<float, HOT> w(smem, C2, (uc.span + uc.merge) * kChunkPx, uc.stage);
        w.arm(uc.hold);
        ChunkGeom g;
        const UnitRecs u = unit_front(bv, off, H, W, nchunk, uc, w, g, uid, part);
        if (u.deferred) return;
        float *dst = out + (((size_t)g.b * H + g.row) * (size_t)W + g.c0) * C2;
        const float *tw = tnorm + off[g.b];
        auto reduce = [&](uint32_t jb, uint32_t je, auto get, float(&vals)[EVREP_MAX_CHANNELS]) {
            float lo_half[kEstMaxBins], hi_half[kEstMaxBins];
#pragma unroll
            for (int i = 0; i < kEstMaxBins; ++i) { lo_half[i] = 0.0f; hi_half[i] = 0.0f; }
            for (uint32_t j = jb; j < je; ++j) {
                const Rec e = get(j);   // digest: {t_n bits, rank, t, p}
                const float tn = __int_as_float(e.x);
#pragma unroll
                for (int i = 0; i < kEstMaxBins; ++i) {
                    if (i < P.C) {
                        const float u = tn - P.shift[i];
                        const float v = tn * est_value(u, seg, bucket, P);   // values = t * value_layer(t - i/(C-1))  (:167)
                        if (e.w > 0) hi_half[i] = hi_half[i] + v; else lo_half[i] = lo_half[i] + v;
                    }
                }
            }
#pragma unroll
            for (int c = 0; c < EVREP_MAX_CHANNELS; ++c) {
                float v = 0.0f;
#pragma unroll
                for (int i = 0; i < kEstMaxBins; ++i) {
                    if (c == i && i < P.C) v = lo_half[i];
                    if (c == P.C + i && i < P.C) v = hi_half[i];
                }
                vals[c] = v;
            }
        };
        // the normalised time is gathered by the digest, one load per record with all lanes busy, not inside the walks (see k_polstats)
        auto digest = [&](const Rec &r) -> Rec { return make_int4(__float_as_int(tw[r.y]), r.y, r.z, r.w); };
        emit_chunk<float, EVREP_MAX_CHANNELS, HOT>(u, digest, g.row * W + g.c0, g.npix, C2, dst, w, (const float *)nullptr, reduce);
    });
}

// --------------------------------------------------------------------------------------------
// F2: per-channel image resize of a channel-last representation, as resize_image / resize_image_process do with
// cv2.resize per channel (ev-YOLOv6/yolov6/data/gen4/precompute_reps.py:179-260, gen1_2yolo.py:230-265).
// cv2's INTER_AREA (shrinking) and INTER_LINEAR are separable: output (oy, ox) = sum over a few source rows of
// beta * (sum over a few source columns of alpha * src) -- x first, then y, in float64, which is the order of
// OpenCV's resizeArea_ / the vertical pass of its linear resize.  The taps come as small per-axis tables built on
// the host from OpenCV's published table construction (gwd_pipeline.area_weights / linear_weights; PARITY UNPINNED
// against cv2 itself, which is absent).  One thread per output element; every source element is read once from
// HBM (neighbouring outputs share cache lines), the result is written once -- in place of two dense
// (dst x src) float64 GEMMs over the whole frame.
// --------------------------------------------------------------------------------------------
struct ResizeTaps {
    const int32_t *ystart, *ycount, *xstart, *xcount;  // [Ho], [Ho], [Wo], [Wo]
    const double *ywt, *xwt;                           // [Ho][T], [Wo][T]
    int32_t T;
};

template <typename InT, typename OutT>
__global__ __launch_bounds__(kThreads) void k_resize_taps(const InT *__restrict__ in, int H, int W, int C, ResizeTaps tp,
                                                         int Ho, int Wo, double scale, OutT *__restrict__ out) {
    const size_t per = (size_t)Ho * Wo * C;
    const size_t e = (size_t)blockIdx.x * kThreads + threadIdx.x;
    if (e >= per) return;
    const int b = blockIdx.y;
    const int c = (int)(e % C);
    const int ox = (int)((e / C) % Wo), oy = (int)(e / ((size_t)C * Wo));
    const InT *src = in + (size_t)b * H * W * C + c;
    const int ys = tp.ystart[oy], yn = tp.ycount[oy], xs = tp.xstart[ox], xn = tp.xcount[ox];
    double acc = 0.0;
    for (int ty = 0; ty < yn; ++ty) {
        const InT *row = src + (size_t)(ys + ty) * W * C;
        double rx = 0.0;
        for (int tx = 0; tx < xn; ++tx) rx += (double)row[(size_t)(xs + tx) * C] * tp.xwt[ox * tp.T + tx];
        acc += rx * tp.ywt[oy * tp.T + ty];
    }
    out[(size_t)b * per + e] = (OutT)(acc * scale);
}

// --------------------------------------------------------------------------------------------
// F3: ev-licious events_to_voxel_grid with SUB-PIXEL coordinates (Events.divider > 1, e.g. after
// resize_to_resolution; ev-licious/src/evlicious/tools/utils.py:86-102): every event is drawn into the four
// pixels around (x, y) with weight (1 - |xlim - x|)(1 - |ylim - y|) p, accumulated in a float32 grid by np.add.at.
// The stream is binned by the TRUNCATED coordinates; one thread per output cell (Y, X) gathers, in the reference's
// accumulation order, the events of pixels (X, Y), (X, Y-1), (X-1, Y), (X-1, Y-1) -- the (xlim, ylim) loop nest of
// _draw_xy_to_voxel_grid, events in time order inside each tap -- so the float32 sums round exactly as the
// reference's.  xy[rank] = the event's float64 (x, y), gathered by the record's rank.  A gather kernel, not a
// tile builder: this path serves resized event streams, not the headline windows.
// grid (ceil(H*W / 256), B), 256 threads; out DEVICE float32 (B, H, W, bins).
// --------------------------------------------------------------------------------------------
#ifdef EVREP_TU_BUILDERS   // (compiled by the one translation unit that launches it)
static __global__ __launch_bounds__(kThreads) void k_voxel_subpixel(const int4 *__restrict__ ev, const Rec *__restrict__ sorted,
                                                            const uint32_t *__restrict__ chunk_off, const int64_t *__restrict__ off,
                                                            const double *__restrict__ xy, int H, int W, int nchunk, int bins,
                                                            const int64_t *__restrict__ t_range, float *__restrict__ out) {
    const int b = blockIdx.y;
    const int cell = blockIdx.x * kThreads + threadIdx.x;
    if (cell >= H * W) return;
    const int Y = cell / W, X = cell - Y * W;
    const int64_t beg = off[b];
    const int64_t n_win = off[b + 1] - beg;
    float acc[EVREP_MAX_CHANNELS];
#pragma unroll
    for (int c = 0; c < EVREP_MAX_CHANNELS; ++c) acc[c] = 0.0f;
    if (n_win >= 2) {  // `if len(events) < 2: return voxel_grid` (:58-59)
        int64_t t0 = ev[beg].z, t1 = ev[beg + n_win - 1].z;
        if (t_range) { t0 = t_range[2 * b]; t1 = t_range[2 * b + 1]; }
        const double den = (t1 - t0) == 0 ? 1.0 : (double)(t1 - t0);
        for (int tap = 0; tap < 4; ++tap) {
            const int px = X - (tap >> 1), py = Y - (tap & 1);  // the source pixel whose (xlim, ylim) tap is this cell
            if (px < 0 || py < 0) continue;
            const uint32_t *co = chunk_off + ((size_t)b * H + py) * (nchunk + 1);
            uint32_t lo = co[px / kChunkPx], hi = co[px / kChunkPx + 1];
            const int key = py * W + px;
            const uint32_t end = hi;
            while (lo < hi) {  // first record of the chunk with pixel id >= key
                const uint32_t mid = (lo + hi) >> 1;
                if (sorted[mid].x < key) lo = mid + 1; else hi = mid;
            }
            for (uint32_t j = lo; j < end; ++j) {
                const Rec e = sorted[j];
                if (e.x != key) break;
                const int64_t num = (int64_t)(bins - 1) * ((int64_t)e.z - t0);
                const double bpos = (double)num / den;
                if (!(bpos > -1.0 && bpos < 1.0e9)) continue;
                const int bi = (int)bpos;  // astype("int32"): toward zero (:67)
                if (bi < 0 || bi >= bins) continue;
                const double x = xy[2 * (beg + e.y)], y = xy[2 * (beg + e.y) + 1];
                const double w = (1.0 - fabs((double)X - x)) * (1.0 - fabs((double)Y - y));
                // np.add.at(float32 grid, ..., float64 values) runs numpy's float64 add loop and rounds the SUM back to
                // float32: one rounding per addition, of the exact float64 sum
                const double val = w * (double)e.w;
#pragma unroll
                for (int c = 0; c < EVREP_MAX_CHANNELS; ++c) if (c == bi) acc[c] = (float)((double)acc[c] + val);
            }
        }
    }
    float *dst = out + ((size_t)b * H * W + cell) * bins;
#pragma unroll
    for (int c = 0; c < EVREP_MAX_CHANNELS; ++c) if (c < bins) dst[c] = acc[c];
}
#endif

// --------------------------------------------------------------------------------------------
// Placement probe: the write footprint of the float64 12-channel builder (one wave per 12 KiB tile, XCD-contiguous
// eighths, non-temporal 16-byte stores, 19 waves per CU) with nothing else.  On MI355X the time of this kernel into a
// 0.9 GB tensor takes one of three levels depending on where the tensor lies physically (NOTES.md 8,
// tools/microbench/placement_patterns.hip); engine.probe_output_placement times it into candidate allocations.
// grid (tiles), 64 threads, dynamic LDS 8320 B.  Writes zeros.
// --------------------------------------------------------------------------------------------
#ifdef EVREP_TU_BUILDERS   // (compiled by the one translation unit that launches it)
static __global__ __launch_bounds__(kWave) void k_store_probe(float *__restrict__ out, int ntiles) {
    extern __shared__ __align__(16) unsigned char smem[];
    typedef float nt4 __attribute__((ext_vector_type(4)));
    const int t = chunk_unit(ntiles);
    nt4 z = {0.f, 0.f, 0.f, 0.f};
    if (out == nullptr) z.x = (float)smem[threadIdx.x];  // keeps the LDS allocation (the occupancy) alive
    nt4 *b = reinterpret_cast<nt4 *>(out) + (size_t)t * 12 * kWave + threadIdx.x;
#pragma unroll
    for (int q = 0; q < 12; ++q) __builtin_nontemporal_store(z, b + q * kWave);
}
#endif

}  // namespace evrep
