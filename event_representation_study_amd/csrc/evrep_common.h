// evrep_common.h -- shared device-side definitions of libevrep (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "evrep.h"

namespace evrep {

constexpr int kThreads = 256;  // 4 wave64 per workgroup
constexpr int kWaves = 4;
constexpr int kWave = 64;
constexpr int kChunkPx = 128;  // pixels of one row that one builder wavefront owns

// The hot list of a workspace (evrep_builders.hip, run_units / defer_unit; sized by evrep_plan_init, cleared by the binning
// pass): kHotLists SUBLISTS of unit-piece ids, each with a counter and an exit ticket of its own, 64 bytes apart.
#ifndef EVREP_HOT_STAGE
#define EVREP_HOT_STAGE 256
#endif
#ifndef EVREP_HOT_GRID
#define EVREP_HOT_GRID 4096
#endif
constexpr size_t kTsCutsBytes = 384;             // sizeof(TsCuts) (evrep_builders.hip: a window's TimeSurface cuts in the workspace)
constexpr int kHotStage = EVREP_HOT_STAGE;    // records of a hot wave's LDS stage (4 KB; with the tile: 640 records for float64 x 12)
constexpr int kHotGrid = EVREP_HOT_GRID;      // workgroups of a hot launch
constexpr int kHotParts = 8;                  // parts per unit at most (TORE's two-chunk units straddle three chunks: six)
constexpr int kHotCodes = 64;                 // item = unit id * kHotCodes + piece code
constexpr int kHotLists = 64;
constexpr int kHotWhole = kHotCodes - 1;      // piece code: the WHOLE unit, taken by the hot wave's split sweep (unit_records, Split::in_hot)
constexpr int kHotCoop = kHotCodes - 2;       // piece code (r06): the WHOLE unit, for the cooperative launch of several waves per unit (k_mdes_coop); the one-wave hot launch skips it
constexpr int kHotSub0 = 40, kHotSubMax = 16; // piece codes kHotSub0 + s: the s-th TIME slice of a hot unit of >= kHotSubMin records (unit_records, sub-waves)
#ifdef EVREP_TIMING
constexpr uint32_t kHotSubMin = 0x7fffffffu, kHotSubRecs = 1024u;   // (the experiment build keeps its phase marks where sliced units order their kept records)
#else
constexpr uint32_t kHotSubMin = 4096u, kHotSubRecs = 1024u;   // (measured, r05b: slicing units of 2 000 - 4 000 records costs the Gen1 circle stream 7 %: zeroing, merging and the ticket outweigh the shorter sweep)
#endif
// slices of a hot unit of nrec >= kHotSubMin records, and the records of a slice (a multiple of 64; every slice is non-empty)
__host__ __device__ inline uint32_t hot_sub_count(uint32_t nrec) { const uint32_t s = nrec / kHotSubRecs; return s > (uint32_t)kHotSubMax ? (uint32_t)kHotSubMax : s; }
__host__ __device__ inline uint32_t hot_sub_quota(uint32_t nrec, uint32_t S) { return (((nrec + S - 1u) / S) + 63u) & ~63u; }
constexpr uint32_t kHotSubHdrBytes = 128u;    // a sliced unit's slot: ticket, kept counts of the slices; then 8 words per pixel, then the slices' kept lists
constexpr int kHotSplitStage = 576;           // records of the LDS stage of a hot launch whose waves take whole units by the split path
constexpr uint32_t kStEscaped = 1u << 16;     // BlockStats::status, internal: the block holds a polarity outside {-1, 0, 1} (escaped in its 8-byte record)
constexpr int kHotHdrWords = 2 * kHotLists * 16;   // [l * 16]: sublist l's item count, [(kHotLists + l) * 16]: its exit ticket
static_assert(kHotGrid % kHotLists == 0, "every sublist is worked off by kHotGrid / kHotLists workgroups");
// items one sublist holds: eight times its fair share (a full one sends the unit to the next)
__host__ __device__ inline uint32_t hot_sublist_cap(uint32_t cap_total) { return 2u * (cap_total / (kHotLists / 4) + 64u); }
__host__ __device__ inline uint32_t hot_items_total(int64_t total_events) { return (uint32_t)(kHotParts * ((size_t)total_events / 65 + 1)); }

// Per-window statistics produced by the binning pass (workspace, one per window).
struct WindowMeta {
    int32_t tmin, tmax;              // over all events of the window
    int32_t xmin, xmax, ymin, ymax;  // over all events of the window (raw coordinates)
    uint32_t neg_flags;              // bit w: an event with p == -1 exists in MDES window w
    uint32_t oob_flags;              // bit 7*c + w: out-of-frame key in MDES window w, class c (0 any, 1 p==1, 2 p==-1, 3 p==0)
    uint32_t status;                 // EVREP_ST_*
    int32_t n_valid;                 // in-frame events
    int32_t pad[6];
};
static_assert(sizeof(WindowMeta) == 64, "WindowMeta is one 64-byte line");

// One binned event: 16 bytes, what both partition levels move and every builder reads.
//   x = pixel id (x + y*W) inside the window, y = rank (index inside the window, i.e. time order),
//   z = t (raw int32 timestamp), w = p (raw polarity)
using Rec = int4;

// The key-sorted passes (plan->reserved 2 and 3) move 8-byte records (r03): the run position of a record already names its
// key = (sensor row, 128-pixel chunk), so a record only carries
//   x = t (raw int32 timestamp),   y = column mod 512 | (p + 1) << 9 | rank << 11
// -- 9 column bits resolve any unit of up to four chunks (a builder unit spans at most three: TORE's shifted frame), 21
// rank bits cover the 1 048 576-event windows these passes serve, and the two polarity bits hold {-1, 0, 1}; any other
// polarity value is escaped (3) and re-read from the caller's event row `rank`.  Half the bytes k_block_keysort writes and
// every builder wave / the per-key column sort reads.
using Rec8 = uint2;
__device__ inline Rec8 rec8_pack(int col, int rank, int t, int p) {
    const uint32_t p2 = (uint32_t)(p + 1) <= 2u ? (uint32_t)(p + 1) : 3u;
    return make_uint2((uint32_t)t, ((uint32_t)col & 511u) | (p2 << 9) | ((uint32_t)rank << 11));
}
// row_base = row * W, base_col = first column of the unit that owns the record, ev_win = the window's events
__device__ inline Rec rec8_unpack(const Rec8 q, int row_base, int base_col, const int4 *__restrict__ ev_win) {
    const uint32_t w = q.y;
    const int col = base_col + (int)(((w & 511u) - (uint32_t)base_col) & 511u);
    const int rank = (int)(w >> 11);
    const uint32_t p2 = (w >> 9) & 3u;
    int p = (int)p2 - 1;
    if (p2 == 3u) p = ev_win[rank].w;
    return make_int4(row_base + col, rank, (int)q.x, p);
}

// The 7 "SBN" windows of MixedDensityEventStack.create_windows
// (representation_search/mixed_density_event_stack.py:48-74) as [lo, hi) rank ranges.
struct MdesWindows {
    int32_t lo[8], hi[8];   // entry 7 only exists under the "SBT" stacking (eight windows cut by time, evrep_mdes_ex)
};

__host__ __device__ inline MdesWindows mdes_windows(int64_t n64) {
    MdesWindows w;
    int32_t n = (int32_t)n64;
    int32_t third = n / 3;
    w.lo[0] = 0; w.hi[0] = n;
    for (int i = 0; i < 3; ++i) { w.lo[1 + i] = i * third; w.hi[1 + i] = (i + 1) * third; }
    int32_t cur = n, start = 0;
    for (int i = 0; i < 3; ++i) { cur /= 2; start += cur; w.lo[4 + i] = start; w.hi[4 + i] = n; }
    w.lo[7] = 0; w.hi[7] = 0;
    return w;
}

// bit w set iff rank r lies in MDES window w
__device__ inline uint32_t mdes_membership(const MdesWindows &w, int32_t r) {
    uint32_t m = 0;
#pragma unroll
    for (int i = 0; i < 7; ++i) m |= (r >= w.lo[i] && r < w.hi[i]) ? (1u << i) : 0u;
    return m;
}

// 16-byte loads from an address known to be in global memory.  Pointers that a kernel reads out of a table in memory are
// generic to the compiler: it emits flat_load, which also counts against lgkmcnt -- every LDS wait then waits for those
// loads too.
using u32x4_t = __attribute__((ext_vector_type(4))) uint32_t;
__device__ inline u32x4_t gload16_raw(const void *p) {
    return *reinterpret_cast<const u32x4_t __attribute__((address_space(1))) *>(reinterpret_cast<uintptr_t>(p));
}
// base (wave-uniform: SGPR pair) + 32-bit byte offset: one address register per lane instead of a 64-bit multiply-add chain
__device__ inline uint4 gload16_at(const void *base, uint32_t byte_off) {
    const u32x4_t v = *reinterpret_cast<const u32x4_t __attribute__((address_space(1))) *>(
        reinterpret_cast<const char __attribute__((address_space(1))) *>(reinterpret_cast<uintptr_t>(base)) + byte_off);
    return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ inline double gload_f64(const double *p) {
    return *reinterpret_cast<const double __attribute__((address_space(1))) *>(reinterpret_cast<uintptr_t>(p));
}
__device__ inline void gstore16(void *p, uint4 v) {
    u32x4_t w; w.x = v.x; w.y = v.y; w.z = v.z; w.w = v.w;
    *reinterpret_cast<u32x4_t __attribute__((address_space(1))) *>(reinterpret_cast<uintptr_t>(p)) = w;
}
__device__ inline void gstore_f32(float *p, float v) {
    *reinterpret_cast<float __attribute__((address_space(1))) *>(reinterpret_cast<uintptr_t>(p)) = v;
}
__device__ inline void gstore_f64(double *p, double v) {
    *reinterpret_cast<double __attribute__((address_space(1))) *>(reinterpret_cast<uintptr_t>(p)) = v;
}
__device__ inline uint4 gload16(const void *p) { const u32x4_t v = gload16_raw(p); return make_uint4(v.x, v.y, v.z, v.w); }
__device__ inline float4 gload16f(const void *p) {
    const u32x4_t v = gload16_raw(p);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}

__device__ inline int lane_id() { return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

// Stable in-wave multisplit: for every valid lane, `rank` = number of lower valid lanes holding
// the same key, `last` = no higher valid lane holds it.  nbits = significant key bits (uniform).
__device__ inline void wave_match(uint32_t key, int nbits, bool valid, int lane, uint32_t &rank, bool &last) {
    uint64_t mask = __ballot(valid);
    for (int b = 0; b < nbits; ++b) {
        bool bit = (key >> b) & 1u;
        uint64_t bal = __ballot(bit);
        mask &= bit ? bal : ~bal;
    }
    uint64_t below = (1ull << lane) - 1ull;
    rank = (uint32_t)__popcll(mask & below);
    last = (mask & ~below & ~(1ull << lane)) == 0ull;
}

__device__ inline int bits_for(int n) {  // bits needed to represent values in [0, n)
    int b = 0;
    while ((1 << b) < n) ++b;
    return b;
}

// Exclusive scan of one uint32 per thread across a workgroup of NW waves. tmp: >= NW uint32 of LDS.
// Returns the exclusive prefix; *total receives the workgroup sum.  Contains __syncthreads().
template <int NW = kWaves>
__device__ inline uint32_t block_exclusive_scan(uint32_t v, uint32_t *tmp, uint32_t *total) {
    int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t o = __shfl_up(incl, d, 64);
        if (lane >= d) incl += o;
    }
    if (lane == 63) tmp[wave] = incl;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        uint32_t t = tmp[w];
        if (w < wave) base += t;
        tot += t;
    }
    __syncthreads();
    *total = tot;
    return base + incl - v;
}

}  // namespace evrep
