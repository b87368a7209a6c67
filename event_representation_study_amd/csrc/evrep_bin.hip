// evrep_bin.hip -- the (y,x) binning pass: every window's events partitioned by pixel, stably (time order kept inside
// a pixel), written for gfx950 (wave64, LDS counters, ballot multisplit).  Four passes, chosen by evrep_plan_init:
//   * key-sorted (round 2, sparse windows): k_block_keysort orders each workgroup's events by (sensor row, 128-pixel
//     chunk) in LDS; the builder waves gather their unit from the block runs and finish the order (evrep_builders.hip);
//   * key pass for dense windows: k_block_keysort + k_col_sort_runs per (row, chunk) key -> the pixel-sorted stream;
//   * two-kernel pass: k_block_rowsort (by sensor row) + k_col_sort_runs per row;
//   * three-kernel pass (round 1): k_row_hist, k_row_scan / k_row_scatter(_fused), k_col_sort -- the first section
//     of this file; very long windows and very tall sensors.
//
// This replaces the `index = y*W + x` scatter every reference builder starts with
// (event_stack.py:123-125, operations.py:40, time_surface.py:67, tore.py:23-47): after it, each
// builder is a per-pixel segmented reduction with coalesced, write-once output.
#include "evrep_common.h"

namespace evrep {

#ifndef EVREP_BIN_THREADS
#define EVREP_BIN_THREADS 256
#endif
constexpr int kBinThreads = EVREP_BIN_THREADS;  // workgroup size of the row-partition kernels
constexpr int kBinWaves = kBinThreads / kWave;

// Wave-wide reductions on the VALU (DPP row shifts + row broadcasts, gfx9 family), result in every
// lane via readlane(63).  No LDS traffic, no lgkmcnt waits -- ten of these run per block in k_row_hist.
template <typename Op>
__device__ inline int wave_reduce_dpp(int v, int identity, Op op) {
    v = op(v, __builtin_amdgcn_update_dpp(identity, v, 0x111, 0xf, 0xf, false));  // row_shr:1
    v = op(v, __builtin_amdgcn_update_dpp(identity, v, 0x112, 0xf, 0xf, false));  // row_shr:2
    v = op(v, __builtin_amdgcn_update_dpp(identity, v, 0x114, 0xf, 0xf, false));  // row_shr:4
    v = op(v, __builtin_amdgcn_update_dpp(identity, v, 0x118, 0xf, 0xf, false));  // row_shr:8
    v = op(v, __builtin_amdgcn_update_dpp(identity, v, 0x142, 0xa, 0xf, false));  // row_bcast:15 -> rows 1,3
    v = op(v, __builtin_amdgcn_update_dpp(identity, v, 0x143, 0xc, 0xf, false));  // row_bcast:31 -> rows 2,3
    return __builtin_amdgcn_readlane(v, 63);
}
__device__ inline int wave_min(int v) { return wave_reduce_dpp(v, INT32_MAX, [](int a, int b) { return min(a, b); }); }
__device__ inline int wave_max(int v) { return wave_reduce_dpp(v, INT32_MIN, [](int a, int b) { return max(a, b); }); }
__device__ inline uint32_t wave_or(uint32_t v) { return (uint32_t)wave_reduce_dpp((int)v, 0, [](int a, int b) { return a | b; }); }
__device__ inline int wave_sum(int v) { return wave_reduce_dpp(v, 0, [](int a, int b) { return a + b; }); }

// Ordering point between LDS phases of ONE wave (LDS operations of a wave execute in program order; the
// compiler only has to keep them in order).  Nothing is drained, no s_barrier.
__device__ inline void wave_phase_lds() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Per-(window, block) partial statistics; reduced into WindowMeta by k_row_scan (no global atomics).
struct BlockStats {
    int32_t tmin, tmax, xmin, xmax, ymin, ymax;
    uint32_t neg_flags, oob_flags, status;
    int32_t n_valid;
    int32_t pad[2];
};
static_assert(sizeof(BlockStats) == 48, "BlockStats");

__device__ inline void stats_identity(BlockStats &s) {
    s.tmin = INT32_MAX; s.tmax = INT32_MIN; s.xmin = INT32_MAX; s.xmax = INT32_MIN; s.ymin = INT32_MAX; s.ymax = INT32_MIN;
    s.neg_flags = 0; s.oob_flags = 0; s.status = 0; s.n_valid = 0; s.pad[0] = 0; s.pad[1] = 0;
}
__device__ inline void stats_merge(BlockStats &a, const BlockStats &b) {
    a.tmin = min(a.tmin, b.tmin); a.tmax = max(a.tmax, b.tmax);
    a.xmin = min(a.xmin, b.xmin); a.xmax = max(a.xmax, b.xmax);
    a.ymin = min(a.ymin, b.ymin); a.ymax = max(a.ymax, b.ymax);
    a.neg_flags |= b.neg_flags; a.oob_flags |= b.oob_flags; a.status |= b.status; a.n_valid += b.n_valid;
}
__device__ inline void stats_wave_reduce(BlockStats &s) {
    s.tmin = wave_min(s.tmin); s.tmax = wave_max(s.tmax);
    s.xmin = wave_min(s.xmin); s.xmax = wave_max(s.xmax);
    s.ymin = wave_min(s.ymin); s.ymax = wave_max(s.ymax);
    s.neg_flags = wave_or(s.neg_flags); s.oob_flags = wave_or(s.oob_flags); s.status = wave_or(s.status);
    s.n_valid = wave_sum(s.n_valid);
}

constexpr int kRegBatch = 8;  // records a lane keeps in registers at a time

// XCD-aware decode of a 1-D grid into (window, block): workgroup id i runs on XCD i % 8 (observed
// dispatch order), so all blocks of window b are given ids congruent to b mod 8.  A window's
// scattered 16-byte record writes then meet in ONE XCD's L2 and leave it as full lines, and the
// events the histogram pass pulled through that L2 are re-read there by the scatter pass.
// Grid size = 8 * ceil(B/8) * nblk; placement only affects speed, never results.
// A last, partial group of windows (B % 8 of them: 4 windows of 10^6 events, one window of a per-sample call) is laid out
// window-major instead -- consecutive ids = consecutive blocks of one window, round-robin over ALL eight XCDs (r04: with
// the mapping above half the chip idled on 4 windows: k_block_keysort 71 us where 8 windows of the same total took 39).
__device__ inline bool decode_window_block(int B, int nblk, int &b, int &blk) {
    const int id = blockIdx.x;
    const int per_group = 8 * nblk;
    const int grp = id / per_group, r = id - grp * per_group;
    const int nw = min(8, B - 8 * grp);
    if (nw == 8) {
        b = grp * 8 + (r & 7);
        blk = r >> 3;
        return true;
    }
    if (r >= nw * nblk) return false;
    b = grp * 8 + r / nblk;
    blk = r - (r / nblk) * nblk;
    return true;
}

// grid (8 * ceil(B/8) * nblk), 256 threads, dynamic LDS = H * 4 bytes.
#ifdef EVREP_TU_CORE   // (compiled by the one translation unit that launches it)
static __global__ __launch_bounds__(kBinThreads) void k_row_hist(const int4 *__restrict__ ev, const int64_t *__restrict__ off,
                                                      int B, int H, int W, int chunk, int nblk,
                                                      uint32_t *__restrict__ table, BlockStats *__restrict__ stats) {
    extern __shared__ uint32_t hist[];
    __shared__ BlockStats wstats[kBinWaves];
    int b, blk;
    if (!decode_window_block(B, nblk, b, blk)) return;
    const int64_t beg = off[b];
    const int64_t n = off[b + 1] - beg;
    const int64_t lo = (int64_t)blk * chunk;
    if (lo >= n) return;  // k_row_scan only reads the blocks a window really has
    for (int i = threadIdx.x; i < H; i += kBinThreads) hist[i] = 0;
    __syncthreads();
    const int64_t hi = (lo + chunk < n) ? lo + chunk : n;
    const MdesWindows mw = mdes_windows(n);
    const int64_t HW = (int64_t)H * W;
    BlockStats st;
    stats_identity(st);
    for (int64_t r0 = lo; r0 < hi; r0 += (int64_t)kRegBatch * kBinThreads) {
        int4 e[kRegBatch];
        int tprev[kRegBatch];
#pragma unroll
        for (int i = 0; i < kRegBatch; ++i) {  // all loads of the batch in flight together
            const int64_t r = r0 + (int64_t)i * kBinThreads + threadIdx.x;
            e[i] = make_int4(0, 0, INT32_MAX, 0);
            tprev[i] = INT32_MIN;
            if (r < hi) {
                e[i] = ev[beg + r];
                // the predecessor's timestamp comes from the neighbouring lane; only lane 0 of a wave loads it
                if ((threadIdx.x & 63) == 0 && r > 0) tprev[i] = ev[beg + r - 1].z;
            }
        }
#pragma unroll
        for (int i = 0; i < kRegBatch; ++i) {
            const int up = __shfl_up(e[i].z, 1, 64);
            if ((threadIdx.x & 63) != 0) tprev[i] = up;
        }
#pragma unroll
        for (int i = 0; i < kRegBatch; ++i) {
            const int64_t r = r0 + (int64_t)i * kBinThreads + threadIdx.x;
            if (r < hi) {
                const int64_t key = (int64_t)e[i].x + (int64_t)e[i].y * W;
                const uint32_t memb = mdes_membership(mw, (int32_t)r);
                if (e[i].w == -1) st.neg_flags |= memb;
                if (key >= 0 && key < HW) {
                    atomicAdd(&hist[(uint32_t)key / (uint32_t)W], 1u);
                    ++st.n_valid;
                } else {
                    st.status |= EVREP_ST_OOB;
                    const int cls = e[i].w == 1 ? 1 : (e[i].w == -1 ? 2 : (e[i].w == 0 ? 3 : 0));
                    st.oob_flags |= memb | (cls ? (memb << (7 * cls)) : 0u);
                }
                if (tprev[i] > e[i].z) st.status |= EVREP_ST_UNSORTED;
                st.tmin = min(st.tmin, e[i].z); st.tmax = max(st.tmax, e[i].z);
                st.xmin = min(st.xmin, e[i].x); st.xmax = max(st.xmax, e[i].x);
                st.ymin = min(st.ymin, e[i].y); st.ymax = max(st.ymax, e[i].y);
            }
        }
    }
    stats_wave_reduce(st);
    if ((threadIdx.x & 63) == 0) wstats[threadIdx.x >> 6] = st;
    __syncthreads();
    if (threadIdx.x == 0) {
        BlockStats t = wstats[0];
        for (int w = 1; w < kBinWaves; ++w) stats_merge(t, wstats[w]);
        stats[(size_t)b * nblk + blk] = t;
    }
    uint32_t *dst = table + ((size_t)b * nblk + blk) * H;
    for (int i = threadIdx.x; i < H; i += kBinThreads) dst[i] = hist[i];
}
#endif

// grid (B), 256 threads, dynamic LDS = (H + 8) * 4 bytes.
// table[b][blk][row] -> exclusive prefix over blk; row_off[b][row] = global start of the row;
// meta[b] = reduction of the window's block statistics.
#ifdef EVREP_TU_CORE   // (compiled by the one translation unit that launches it)
static __global__ __launch_bounds__(kBinThreads) void k_row_scan(const int64_t *__restrict__ off, int H, int chunk, int nblk,
                                                      uint32_t *__restrict__ table, uint32_t *__restrict__ row_off,
                                                      const BlockStats *__restrict__ stats, WindowMeta *__restrict__ meta) {
    extern __shared__ uint32_t rowtot[];
    __shared__ BlockStats wstats[kBinWaves];
    __shared__ uint32_t tmp[kBinWaves];
    const int b = blockIdx.x;
    const int64_t beg = off[b];
    const int64_t n = off[b + 1] - beg;
    const int nb = (int)((n + chunk - 1) / chunk);
    // window statistics
    BlockStats st;
    stats_identity(st);
    for (int blk = threadIdx.x; blk < nb; blk += kBinThreads) stats_merge(st, stats[(size_t)b * nblk + blk]);
    stats_wave_reduce(st);
    if ((threadIdx.x & 63) == 0) wstats[threadIdx.x >> 6] = st;
    // exclusive prefix over the window's blocks, row by row (batched so the loads overlap)
    for (int r = threadIdx.x; r < H; r += kBinThreads) {
        uint32_t run = 0;
        for (int b0 = 0; b0 < nb; b0 += kRegBatch) {
            uint32_t v[kRegBatch];
#pragma unroll
            for (int i = 0; i < kRegBatch; ++i)
                v[i] = (b0 + i < nb) ? table[((size_t)b * nblk + b0 + i) * H + r] : 0u;
#pragma unroll
            for (int i = 0; i < kRegBatch; ++i)
                if (b0 + i < nb) { table[((size_t)b * nblk + b0 + i) * H + r] = run; run += v[i]; }
        }
        rowtot[r] = run;
    }
    __syncthreads();
    const int per = (H + kBinThreads - 1) / kBinThreads;
    const int r0 = threadIdx.x * per;
    uint32_t local = 0;
    for (int k = 0; k < per; ++k) if (r0 + k < H) local += rowtot[r0 + k];
    uint32_t total;
    uint32_t run = block_exclusive_scan<kBinWaves>(local, tmp, &total);
    for (int k = 0; k < per; ++k)
        if (r0 + k < H) { const uint32_t t = rowtot[r0 + k]; rowtot[r0 + k] = run; run += t; }
    __syncthreads();
    uint32_t *ro = row_off + (size_t)b * (H + 1);
    for (int r = threadIdx.x; r < H; r += kBinThreads) ro[r] = (uint32_t)beg + rowtot[r];
    if (threadIdx.x == 0) {
        ro[H] = (uint32_t)beg + total;
        BlockStats t = wstats[0];
        for (int w = 1; w < kBinWaves; ++w) stats_merge(t, wstats[w]);
        WindowMeta m;
        m.tmin = t.tmin; m.tmax = t.tmax; m.xmin = t.xmin; m.xmax = t.xmax; m.ymin = t.ymin; m.ymax = t.ymax;
        m.neg_flags = t.neg_flags; m.oob_flags = t.oob_flags; m.status = t.status; m.n_valid = t.n_valid;
        if (n <= 0) m.status |= EVREP_ST_EMPTY;
        else if (t.tmin == t.tmax) m.status |= EVREP_ST_FLAT_TIME;
        for (int i = 0; i < 6; ++i) m.pad[i] = 0;
        meta[b] = m;
    }
}
#endif

// grid (8 * ceil(B/8) * nblk), 256 threads, dynamic LDS = 4 * H * 4 bytes.  Stable placement by sensor row.
// Each wave owns a contiguous quarter of the block's events and keeps them in registers between
// the counting and the placement phase (one HBM read of the events for both).
#ifdef EVREP_TU_CORE   // (compiled by the one translation unit that launches it)
static __global__ __launch_bounds__(kBinThreads) void k_row_scatter(const int4 *__restrict__ ev, const int64_t *__restrict__ off,
                                                         int B, int H, int W, int chunk, int nblk,
                                                         const uint32_t *__restrict__ table,
                                                         const uint32_t *__restrict__ row_off, Rec *__restrict__ sorted1) {
    extern __shared__ uint32_t cnt[];  // [kBinWaves][H]
    int b, blk;
    if (!decode_window_block(B, nblk, b, blk)) return;
    const int64_t beg = off[b];
    const int64_t n = off[b + 1] - beg;
    const int64_t lo = (int64_t)blk * chunk;
    if (lo >= n) return;
    const int64_t hi = (lo + chunk < n) ? lo + chunk : n;
    const int64_t nloc = hi - lo;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t wlo = lo + nloc * wave / kBinWaves, whi = lo + nloc * (wave + 1) / kBinWaves;
    const int64_t HW = (int64_t)H * W;
    constexpr int kSuper = kRegBatch * kWave;  // 512 events per register-resident super batch
    const int nsuper = (int)((whi - wlo + kSuper - 1) / kSuper);
    for (int i = threadIdx.x; i < kBinWaves * H; i += kBinThreads) cnt[i] = 0;
    __syncthreads();
    uint32_t *mycnt = cnt + wave * H;
    int4 e[kRegBatch];
    for (int sb = 0; sb < nsuper; ++sb) {
#pragma unroll
        for (int i = 0; i < kRegBatch; ++i) {
            const int64_t r = wlo + (int64_t)sb * kSuper + i * kWave + lane;
            e[i] = make_int4(-1, -1, 0, 0);
            if (r < whi) e[i] = ev[beg + r];
        }
#pragma unroll
        for (int i = 0; i < kRegBatch; ++i) {
            const int64_t r = wlo + (int64_t)sb * kSuper + i * kWave + lane;
            const int64_t key = (int64_t)e[i].x + (int64_t)e[i].y * W;
            if (r < whi && key >= 0 && key < HW) atomicAdd(&mycnt[(uint32_t)key / (uint32_t)W], 1u);
        }
    }
    __syncthreads();
    for (int row = threadIdx.x; row < H; row += kBinThreads) {
        uint32_t base = row_off[(size_t)b * (H + 1) + row] + table[((size_t)b * nblk + blk) * H + row];
#pragma unroll
        for (int w = 0; w < kBinWaves; ++w) { const uint32_t t = cnt[w * H + row]; cnt[w * H + row] = base; base += t; }
    }
    __syncthreads();
    const int nbits = bits_for(H);
    volatile uint32_t *vcnt = mycnt;
    for (int sb = 0; sb < nsuper; ++sb) {
        if (nsuper > 1) {  // otherwise the registers still hold the only super batch
#pragma unroll
            for (int i = 0; i < kRegBatch; ++i) {
                const int64_t r = wlo + (int64_t)sb * kSuper + i * kWave + lane;
                e[i] = make_int4(-1, -1, 0, 0);
                if (r < whi) e[i] = ev[beg + r];
            }
        }
#pragma unroll
        for (int i = 0; i < kRegBatch; ++i) {
            const int64_t r0 = wlo + (int64_t)sb * kSuper + i * kWave;
            if (r0 >= whi) break;  // uniform
            const int64_t r = r0 + lane;
            const int64_t key = (int64_t)e[i].x + (int64_t)e[i].y * W;
            const bool valid = r < whi && key >= 0 && key < HW;
            const uint32_t row = valid ? (uint32_t)key / (uint32_t)W : 0u;
            uint32_t rk; bool last;
            wave_match(row, nbits, valid, lane, rk, last);
            uint32_t pos = 0;
            if (valid) {
                pos = vcnt[row] + rk;
                sorted1[pos] = make_int4((int)key, (int)r, e[i].z, e[i].w);
            }
            __builtin_amdgcn_wave_barrier();
            if (valid && last) vcnt[row] = pos + 1;
            __builtin_amdgcn_wave_barrier();
        }
    }
}
#endif

// Fused form of k_row_scan + k_row_scatter for sensors whose per-row tables fit one workgroup's LDS
// (fused_scatter_lds_bytes() <= 64 KB; 640x480 and 1280x720 do): every block derives its own
// destinations from the raw (block,row) histogram table -- a few dozen coalesced, independent loads
// per thread instead of a separate 1-block-per-window scan kernel and its launch boundary -- places
// its records stably into an LDS stage in OUTPUT order, and writes the stage out so that consecutive
// lanes write consecutive records of a row (64-byte runs instead of scattered 16-byte stores).
// Block 0 of every window also publishes the row offsets and the reduced window statistics.
// grid (8 * ceil(B/8) * nblk), 256 threads, requires chunk <= kStageRecs.
constexpr int kStageRecs = 2048;
__host__ __device__ inline size_t fused_scatter_lds_bytes(int H) {
    return (size_t)(kBinWaves + 3) * H * sizeof(uint32_t) + (size_t)kStageRecs * sizeof(Rec);
}

#ifdef EVREP_TU_CORE   // (compiled by the one translation unit that launches it)
static __global__ __launch_bounds__(kBinThreads) void k_row_scatter_fused(const int4 *__restrict__ ev, const int64_t *__restrict__ off,
                                                               int B, int H, int W, int chunk, int nblk,
                                                               const uint32_t *__restrict__ table,
                                                               const BlockStats *__restrict__ stats,
                                                               uint32_t *__restrict__ row_off, WindowMeta *__restrict__ meta,
                                                               Rec *__restrict__ sorted1) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    Rec *stage = reinterpret_cast<Rec *>(smem_raw);                          // [kStageRecs]
    uint32_t *cnt = reinterpret_cast<uint32_t *>(stage + kStageRecs);        // [kBinWaves][H] per-wave counts -> local bases
    uint32_t *rowtot = cnt + kBinWaves * H;                                     // [H] window row totals -> row starts
    uint32_t *lbase = rowtot + H;                                            // [H] this block's row counts -> local starts
    uint32_t *delta = lbase + H;                                             // [H] global minus local position
    __shared__ BlockStats wstats[kBinWaves];
    __shared__ uint32_t tmp[kBinWaves];
    int b, blk;
    if (!decode_window_block(B, nblk, b, blk)) return;
    const int64_t beg = off[b];
    const int64_t n = off[b + 1] - beg;
    const int64_t lo = (int64_t)blk * chunk;
    if (lo >= n && !(blk == 0)) return;
    const int nb = (int)((n + chunk - 1) / chunk);
    if (n <= 0) {  // empty window: block 0 still publishes its (empty) rows and status
        uint32_t *ro = row_off + (size_t)b * (H + 1);
        for (int r = threadIdx.x; r <= H; r += kBinThreads) ro[r] = (uint32_t)beg;
        if (threadIdx.x == 0) {
            WindowMeta m;
            m.tmin = INT32_MAX; m.tmax = INT32_MIN; m.xmin = INT32_MAX; m.xmax = INT32_MIN; m.ymin = INT32_MAX; m.ymax = INT32_MIN;
            m.neg_flags = 0; m.oob_flags = 0; m.status = EVREP_ST_EMPTY; m.n_valid = 0;
            for (int i = 0; i < 6; ++i) m.pad[i] = 0;
            meta[b] = m;
        }
        return;
    }
    const int64_t hi = (lo + chunk < n) ? lo + chunk : n;
    const int64_t nloc = hi - lo;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t wlo = lo + nloc * wave / kBinWaves, whi = lo + nloc * (wave + 1) / kBinWaves;
    const int64_t HW = (int64_t)H * W;
    for (int i = threadIdx.x; i < kBinWaves * H; i += kBinThreads) cnt[i] = 0;
    __syncthreads();
    // the wave's events stay in registers from counting to placement (chunk <= 2048: <= 8 per lane)
    uint32_t *mycnt = cnt + wave * H;
    int4 e[kRegBatch];
#pragma unroll
    for (int i = 0; i < kRegBatch; ++i) {
        const int64_t r = wlo + i * kWave + lane;
        e[i] = make_int4(-1, -1, 0, 0);
        if (r < whi) e[i] = ev[beg + r];
    }
#pragma unroll
    for (int i = 0; i < kRegBatch; ++i) {
        const int64_t r = wlo + i * kWave + lane;
        const int64_t key = (int64_t)e[i].x + (int64_t)e[i].y * W;
        if (r < whi && key >= 0 && key < HW) atomicAdd(&mycnt[(uint32_t)key / (uint32_t)W], 1u);
    }
    // window statistics (published by block 0)
    if (blk == 0) {
        BlockStats st;
        stats_identity(st);
        for (int k = threadIdx.x; k < nb; k += kBinThreads) stats_merge(st, stats[(size_t)b * nblk + k]);
        stats_wave_reduce(st);
        if (lane == 0) wstats[wave] = st;
    }
    __syncthreads();
    // per row: this window's total, the part that lies in earlier blocks, this block's own count
    for (int r = threadIdx.x; r < H; r += kBinThreads) {
        uint32_t tot = 0, before = 0;
        for (int b0 = 0; b0 < nb; b0 += kRegBatch) {
            uint32_t v[kRegBatch];
#pragma unroll
            for (int i = 0; i < kRegBatch; ++i) v[i] = (b0 + i < nb) ? table[((size_t)b * nblk + b0 + i) * H + r] : 0u;
#pragma unroll
            for (int i = 0; i < kRegBatch; ++i) { tot += v[i]; if (b0 + i < blk) before += v[i]; }
        }
        rowtot[r] = tot;
        delta[r] = before;  // (parked here until the scans are done)
        uint32_t own = 0;
#pragma unroll
        for (int w = 0; w < kBinWaves; ++w) own += cnt[w * H + r];
        lbase[r] = own;
    }
    __syncthreads();
    // two exclusive scans over the rows: window totals -> row starts, own counts -> local starts
    const int per = (H + kBinThreads - 1) / kBinThreads;
    const int r0 = threadIdx.x * per;
    uint32_t loc_t = 0, loc_o = 0;
    for (int k = 0; k < per; ++k) if (r0 + k < H) { loc_t += rowtot[r0 + k]; loc_o += lbase[r0 + k]; }
    uint32_t total_t, total_o;
    uint32_t run_t = block_exclusive_scan<kBinWaves>(loc_t, tmp, &total_t);
    uint32_t run_o = block_exclusive_scan<kBinWaves>(loc_o, tmp, &total_o);
    for (int k = 0; k < per; ++k)
        if (r0 + k < H) {
            const uint32_t t = rowtot[r0 + k]; rowtot[r0 + k] = run_t; run_t += t;
            const uint32_t o = lbase[r0 + k];  lbase[r0 + k] = run_o;  run_o += o;
        }
    __syncthreads();
    for (int r = threadIdx.x; r < H; r += kBinThreads) {
        const uint32_t gstart = (uint32_t)beg + rowtot[r] + delta[r];
        uint32_t base = lbase[r];
        delta[r] = gstart - base;
#pragma unroll
        for (int w = 0; w < kBinWaves; ++w) { const uint32_t t = cnt[w * H + r]; cnt[w * H + r] = base; base += t; }
        if (blk == 0) row_off[(size_t)b * (H + 1) + r] = (uint32_t)beg + rowtot[r];
    }
    if (blk == 0 && threadIdx.x == 0) {
        row_off[(size_t)b * (H + 1) + H] = (uint32_t)beg + total_t;
        BlockStats t = wstats[0];
        for (int w = 1; w < kBinWaves; ++w) stats_merge(t, wstats[w]);
        WindowMeta m;
        m.tmin = t.tmin; m.tmax = t.tmax; m.xmin = t.xmin; m.xmax = t.xmax; m.ymin = t.ymin; m.ymax = t.ymax;
        m.neg_flags = t.neg_flags; m.oob_flags = t.oob_flags; m.status = t.status; m.n_valid = t.n_valid;
        if (t.tmin == t.tmax) m.status |= EVREP_ST_FLAT_TIME;
        for (int i = 0; i < 6; ++i) m.pad[i] = 0;
        meta[b] = m;
    }
    __syncthreads();
    // stable placement into the stage, in output order
    const int nbits = bits_for(H);
    volatile uint32_t *vcnt = mycnt;
#pragma unroll
    for (int i = 0; i < kRegBatch; ++i) {
        const int64_t rr0 = wlo + i * kWave;
        if (rr0 >= whi) break;  // uniform
        const int64_t r = rr0 + lane;
        const int64_t key = (int64_t)e[i].x + (int64_t)e[i].y * W;
        const bool valid = r < whi && key >= 0 && key < HW;
        const uint32_t row = valid ? (uint32_t)key / (uint32_t)W : 0u;
        uint32_t rk; bool last;
        wave_match(row, nbits, valid, lane, rk, last);
        uint32_t pos = 0;
        if (valid) {
            pos = vcnt[row] + rk;
            stage[pos] = make_int4((int)key, (int)r, e[i].z, e[i].w);
        }
        __builtin_amdgcn_wave_barrier();
        if (valid && last) vcnt[row] = pos + 1;
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    // write-out: consecutive lanes hold consecutive records of a row
    for (uint32_t t = threadIdx.x; t < total_o; t += kBinThreads) {
        const Rec rec = stage[t];
        const uint32_t row = (uint32_t)rec.x / (uint32_t)W;
        sorted1[delta[row] + t] = rec;
    }
}
#endif

// grid (H, B), 64 threads, dynamic LDS = W * 4 bytes.  Stable placement by column inside one row,
// by ONE wave: afterwards sorted2 is ordered by (window, pixel id, rank).  Rows of up to 256 records
// (the common case) keep their records in registers between counting and placement; longer rows
// are walked twice.  Also emits, per row, the record offsets of every kChunkPx-pixel column chunk
// (what one builder wavefront consumes).
#ifdef EVREP_TU_CORE   // (compiled by the one translation unit that launches it)
static __global__ __launch_bounds__(kWave) void k_col_sort(const Rec *__restrict__ sorted1, const uint32_t *__restrict__ row_off,
                                                   int H, int W, int nchunk, Rec *__restrict__ sorted2,
                                                   uint32_t *__restrict__ chunk_off) {
    extern __shared__ uint32_t cnt[];  // [W]
    const int b = blockIdx.y, row = blockIdx.x;
    const int lane = threadIdx.x;
    const uint32_t rs = row_off[(size_t)b * (H + 1) + row], re = row_off[(size_t)b * (H + 1) + row + 1];
    const uint32_t n = re - rs;
    uint32_t *co = chunk_off + ((size_t)b * H + row) * (nchunk + 1);
    if (n == 0) {
        for (int c = lane; c <= nchunk; c += kWave) co[c] = rs;
        return;
    }
    constexpr uint32_t kSuper = 4 * kWave;  // records per register-resident super batch
    const int rowbase = row * W;
    const uint32_t nsuper = (n + kSuper - 1) / kSuper;
    Rec e[4];
    for (int i = lane; i < W; i += kWave) cnt[i] = 0;
    __syncthreads();
    for (uint32_t sb = 0; sb < nsuper; ++sb) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t j = sb * kSuper + i * kWave + lane;
            e[i] = make_int4(0, 0, 0, 0);
            if (j < n) e[i] = sorted1[rs + j];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (sb * kSuper + i * kWave + lane < n) atomicAdd(&cnt[e[i].x - rowbase], 1u);
    }
    __syncthreads();
    // exclusive scan over the W column counters: `per` consecutive columns per lane
    const int per = (W + kWave - 1) / kWave;
    const int c0 = lane * per;
    uint32_t local = 0;
    for (int k = 0; k < per; ++k) if (c0 + k < W) local += cnt[c0 + k];
    uint32_t incl = local;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(incl, d, 64); if (lane >= d) incl += o; }
    uint32_t run = incl - local;
    for (int k = 0; k < per; ++k)
        if (c0 + k < W) { const uint32_t t = cnt[c0 + k]; cnt[c0 + k] = run; run += t; }
    __syncthreads();
    for (int c = lane; c <= nchunk; c += kWave) co[c] = (c * kChunkPx < W) ? rs + cnt[c * kChunkPx] : re;
    __syncthreads();
    const int nbits = bits_for(W);
    volatile uint32_t *vcnt = cnt;
    for (uint32_t sb = 0; sb < nsuper; ++sb) {
        if (nsuper > 1) {  // otherwise the registers still hold the only super batch
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint32_t j = sb * kSuper + i * kWave + lane;
                e[i] = make_int4(0, 0, 0, 0);
                if (j < n) e[i] = sorted1[rs + j];
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t j0 = sb * kSuper + i * kWave;
            if (j0 >= n) break;  // uniform
            const bool valid = j0 + lane < n;
            const uint32_t col = valid ? (uint32_t)(e[i].x - rowbase) : 0u;
            uint32_t rk; bool last;
            wave_match(col, nbits, valid, lane, rk, last);
            uint32_t pos = 0;
            if (valid) {
                pos = vcnt[col] + rk;
                sorted2[rs + pos] = e[i];
            }
            __builtin_amdgcn_wave_barrier();
            if (valid && last) vcnt[col] = pos + 1;
            __builtin_amdgcn_wave_barrier();
        }
    }
}
#endif

// =====================================================================================================
// Two-kernel binning pass (windows of up to kBsMaxBlocks * 8192 events on sensors of up to ~720 rows; the
// three-kernel pass above stays for everything larger).
//
// The three-kernel pass cuts a window into 2048-event blocks that must agree on where every (block,row) run
// goes in ONE row-major stream -- hence a histogram kernel, a table every scatter block re-reads in full, and
// three launch boundaries.  Here NO block depends on any other:
//   k_block_rowsort : a 1024-thread workgroup (16 waves per CU) orders ITS 8192 events by sensor row, stably,
//                     inside LDS and writes them to ITS OWN 8192-record slot of sorted1, plus the exclusive
//                     row offsets of its run (table[b][blk][0..H]).  One read of the events, one coalesced
//                     write, no inter-block communication of any kind.
//   k_col_sort_runs : the wave that owns (window, row) gathers the row's records from the <= 16 block runs
//                     (block order = time order), orders them by column exactly as k_col_sort does and
//                     writes sorted2 / the chunk offsets.  Its output position needs no scan either:
//                     records of earlier rows = sum over blocks of table[b][blk][row].
// =====================================================================================================
#ifndef BS_DEBUG
#define BS_DEBUG 0  // timing experiments only (tools/experiments): 1 = no statistics, 2 = no placement, 4 = no write-out, 8 = no counting
#endif
constexpr int kBsThreads = 1024;
constexpr int kBsWaves = kBsThreads / kWave;     // 16
constexpr int kBsPerLane = 8;
constexpr int kBsChunk = kBsThreads * kBsPerLane;  // 8192 events per workgroup
constexpr int kBsMaxBlocks = 64;                 // block runs a builder wave of the key-sorted pass gathers (one lane each)
constexpr int kBsChainBlocks = 16;               // up to here the run of a record is found by a readlane chain, above by an LDS search

__host__ __device__ inline int bs_hp(int H) { return (H + 1) & ~1; }
__host__ __device__ inline size_t block_rowsort_lds_bytes(int H) {
    return (size_t)kBsChunk * sizeof(Rec) + (size_t)kBsWaves * bs_hp(H) * sizeof(uint16_t) + (size_t)(H + 2) * sizeof(uint32_t);
}

// grid (8 * ceil(B/8) * nblk), 1024 threads, dynamic LDS = block_rowsort_lds_bytes(H).
// table: [B][nblk][H + 1] exclusive offsets of the block's rows inside its run (entry H = in-frame events).
#ifdef EVREP_TU_CORE   // (compiled by the one translation unit that launches it)
static __global__ __launch_bounds__(kBsThreads) void k_block_rowsort(const int4 *__restrict__ ev, const int64_t *__restrict__ off,
                                                             int B, int H, int W, int nblk, uint32_t *__restrict__ table,
                                                             BlockStats *__restrict__ stats, Rec *__restrict__ sorted1) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    Rec *stage = reinterpret_cast<Rec *>(smem_raw);                               // [kBsChunk], output order
    const int Hp = bs_hp(H), Hw = Hp >> 1;
    uint16_t *cnt16 = reinterpret_cast<uint16_t *>(stage + kBsChunk);             // [kBsWaves][Hp] counts -> next positions
    uint32_t *cnt32 = reinterpret_cast<uint32_t *>(cnt16);                        // the same, two rows per word
    uint32_t *lbase = reinterpret_cast<uint32_t *>(cnt16 + (size_t)kBsWaves * Hp);  // [H + 1]
    __shared__ BlockStats wstats[kBsWaves];
    __shared__ uint32_t tmp[kBsWaves];
    int b, blk;
    if (!decode_window_block(B, nblk, b, blk)) return;
    const int64_t beg = off[b];
    const int64_t n = off[b + 1] - beg;
    const int64_t lo = (int64_t)blk * kBsChunk;
    if (lo >= n) return;  // k_col_sort_runs only reads the blocks a window really has
    const int64_t hi = (lo + kBsChunk < n) ? lo + kBsChunk : n;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t wlo = lo + (int64_t)wave * (kBsPerLane * kWave);
    const int64_t whi = (wlo + kBsPerLane * kWave < hi) ? wlo + kBsPerLane * kWave : hi;
    const int64_t HW = (int64_t)H * W;
    // the wave's 512 events stay in registers from counting to placement
    int4 e[kBsPerLane];
    int tprev[kBsPerLane];
#pragma unroll
    for (int i = 0; i < kBsPerLane; ++i) {
        const int64_t r = wlo + i * kWave + lane;
        e[i] = make_int4(-1, -1, INT32_MAX, 0);
        tprev[i] = INT32_MIN;
        if (r < whi) {
            e[i] = ev[beg + r];
            if (lane == 0 && r > 0) tprev[i] = ev[beg + r - 1].z;  // other lanes take it from their neighbour
        }
    }
    for (int i = threadIdx.x; i < kBsWaves * Hw; i += kBsThreads) cnt32[i] = 0;
    __syncthreads();
    // MDES window membership of a rank: uniform for the whole block unless a window boundary cuts it
    const MdesWindows mw = mdes_windows(n);
    uint32_t full = 0, part = 0;
#pragma unroll
    for (int i = 0; i < 7; ++i) {
        if (mw.lo[i] <= lo && hi <= mw.hi[i]) full |= 1u << i;
        else if (!(hi <= mw.lo[i] || lo >= mw.hi[i])) part |= 1u << i;
    }
    uint32_t *mycnt32 = cnt32 + wave * Hw;
    BlockStats st;
    stats_identity(st);
    uint32_t rows[kBsPerLane];  // row of the event, 0xffffffff = not placed (padding lane or out of frame)
    uint32_t sneg = 0;          // wave-uniform part of neg_flags
    // MDES window membership of this wave's 512 consecutive ranks: wave-uniform unless a window boundary cuts them
    bool cut = false;
    if (part) {
        const int32_t a0 = (int32_t)wlo, a1 = (int32_t)whi;
#pragma unroll
        for (int q = 0; q < 7; ++q) cut |= (mw.lo[q] > a0 && mw.lo[q] < a1) || (mw.hi[q] > a0 && mw.hi[q] < a1);
    }
    const uint32_t memb_u = cut ? 0u : (full | (part ? (mdes_membership(mw, (int32_t)wlo) & part) : 0u));
#pragma unroll
    for (int i = 0; i < kBsPerLane; ++i) {
        const int64_t r0 = wlo + i * kWave;
        const int64_t r = r0 + lane;
        const bool in = r < whi;
        // the predecessor's timestamp: one DPP wave shift (no LDS crossbar); lane 0 loaded it itself
        const int up = __builtin_amdgcn_update_dpp(0, e[i].z, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
        if (lane != 0) tprev[i] = up;

        // in-frame test on the flat index x + y*W, as the reference's scatter sees it (an x outside [0, W) can
        // still land in the frame); the common case x in [0, W) needs no 64-bit arithmetic and no division
        uint32_t row = (uint32_t)e[i].y;
        bool valid = in && (uint32_t)e[i].x < (uint32_t)W && (uint32_t)e[i].y < (uint32_t)H;
        if (__any(in && (uint32_t)e[i].x >= (uint32_t)W)) {
            const int64_t key = (int64_t)e[i].x + (int64_t)e[i].y * W;
            valid = in && key >= 0 && key < HW;
            if (valid) row = (uint32_t)key / (uint32_t)W;
        }
        rows[i] = valid ? row : 0xffffffffu;
        if (valid && !(BS_DEBUG & 8)) atomicAdd(&mycnt32[row >> 1], 1u << ((row & 1u) * 16));
        if (!(BS_DEBUG & 1)) {
            if (!cut) {
                if (__any(in && e[i].w == -1)) sneg |= memb_u;
            } else if (in && e[i].w == -1) {
                st.neg_flags |= full | (mdes_membership(mw, (int32_t)r) & part);
            }
            if (__any(in && !valid)) {  // rare: out-of-frame events
                if (in && !valid) {
                    const uint32_t memb = cut ? (full | (mdes_membership(mw, (int32_t)r) & part)) : memb_u;
                    st.status |= EVREP_ST_OOB;
                    const int cls = e[i].w == 1 ? 1 : (e[i].w == -1 ? 2 : (e[i].w == 0 ? 3 : 0));
                    st.oob_flags |= memb | (cls ? (memb << (7 * cls)) : 0u);
                }
            }
        }
        if (in && !(BS_DEBUG & 1)) {
            if (valid) ++st.n_valid;
            if (tprev[i] > e[i].z) st.status |= EVREP_ST_UNSORTED;
            if ((uint32_t)(e[i].w + 1) > 2u) st.status |= kStEscaped;   // (rec8_pack escapes it: builders that hand hot units to a split sweep need to know)
            st.tmin = min(st.tmin, e[i].z); st.tmax = max(st.tmax, e[i].z);
            st.xmin = min(st.xmin, e[i].x); st.xmax = max(st.xmax, e[i].x);
            st.ymin = min(st.ymin, e[i].y); st.ymax = max(st.ymax, e[i].y);
        }
    }
    st.neg_flags |= sneg;
    stats_wave_reduce(st);
    if (lane == 0) wstats[wave] = st;
    __syncthreads();
    // two rows per thread (one packed word): exclusive running count over the waves, then over the rows
    uint32_t run = 0;
    if ((int)threadIdx.x < Hw) {
#pragma unroll
        for (int w = 0; w < kBsWaves; ++w) { const uint32_t c = cnt32[w * Hw + threadIdx.x]; cnt32[w * Hw + threadIdx.x] = run; run += c; }
    }
    const uint32_t own0 = run & 0xffffu, own1 = run >> 16;
    uint32_t total;
    const uint32_t ex = block_exclusive_scan<kBsWaves>(own0 + own1, tmp, &total);
    if ((int)threadIdx.x < Hw) {
        const uint32_t l0 = ex, l1 = ex + own0;  // <= 8192: fits the 16-bit halves, no carry between them
        const uint32_t add = l0 | (l1 << 16);
#pragma unroll
        for (int w = 0; w < kBsWaves; ++w) cnt32[w * Hw + threadIdx.x] += add;
        uint32_t *tb = table + ((size_t)b * nblk + blk) * (H + 1);
        const int r0 = 2 * (int)threadIdx.x;
        tb[r0] = l0; lbase[r0] = l0;
        if (r0 + 1 < H) { tb[r0 + 1] = l1; lbase[r0 + 1] = l1; }
    }
    if (threadIdx.x == 0) {
        table[((size_t)b * nblk + blk) * (H + 1) + H] = total;
        BlockStats t = wstats[0];
        for (int w = 1; w < kBsWaves; ++w) stats_merge(t, wstats[w]);
        stats[(size_t)b * nblk + blk] = t;
    }
    __syncthreads();
    // stable placement into the stage, in output order
    const int nbits = bits_for(H);
    volatile uint16_t *vpos = cnt16 + (size_t)wave * Hp;
#pragma unroll
    for (int i = 0; i < kBsPerLane; ++i) {
        if (wlo + i * kWave >= whi) break;  // uniform
        const bool valid = rows[i] != 0xffffffffu;
        const uint32_t row = valid ? rows[i] : 0u;
        uint32_t rk = 0; bool last = false;
        if (!(BS_DEBUG & 2)) wave_match(row, nbits, valid, lane, rk, last);
        uint32_t pos = 0;
        if (valid) {
            pos = (BS_DEBUG & 2) ? (uint32_t)(wave * 512 + i * 64 + lane) : (uint32_t)vpos[row] + rk;
            const int64_t r = wlo + i * kWave + lane;
            const Rec rec = make_int4((int)((int64_t)e[i].x + (int64_t)e[i].y * W), (int)r, e[i].z, e[i].w);
            if (BS_DEBUG & 16) sorted1[beg + lo + pos] = rec; else stage[pos] = rec;
        }
        __builtin_amdgcn_wave_barrier();
        if (valid && last) vpos[row] = (uint16_t)(pos + 1);
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    Rec *dst = sorted1 + beg + lo;
    if (!(BS_DEBUG & (4 | 16)))
        for (uint32_t t = threadIdx.x; t < total; t += kBsThreads) dst[t] = stage[t];
}
#endif

// grid (ceil(H / kCsWaves), B), kCsWaves * 64 threads = kCsWaves independent waves (no block barrier), one sensor row each;
// dynamic LDS = kCsWaves * col_sort_wave_words(W) * 4.  A lane reads the row's offsets in TWO block runs (lane and
// lane + 64): up to kCsMaxRuns = 128 runs per window, 1 048 576 events with 8192-event blocks.
// (Several rows per wave, with all their fetches in flight together, were slower: R = 2, 4 measured in round 2.)
#ifndef EVREP_CS_WAVES
#define EVREP_CS_WAVES 8   // 8 keys per workgroup share the table and record lines in one CU's L1: 4 x 10^6-event windows at 1280x720 bin in 88 us (4: 97)
#endif
constexpr int kCsWaves = EVREP_CS_WAVES;   // by key (128 column counters per wave)
constexpr int kCsRowWaves = 4;             // by row (W column counters per wave: the workgroup's LDS grows with the sensor width)
constexpr int kCsMaxRuns = 128;
__host__ __device__ inline int col_sort_per4(int W) { return ((W + kWave - 1) / kWave + 3) / 4; }
__host__ __device__ inline int col_sort_words(int W) { return kWave * 4 * col_sort_per4(W); }  // per-wave counter array
__host__ __device__ inline int col_sort_wave_words(int W) { return col_sort_words(W) + 2 * kCsMaxRuns; }   // + the run table (pre, src)

#ifndef CS_WAVES
#define CS_WAVES 8   // 63 VGPRs, no scratch: dense binning 123 -> 119 us (1 Mpx), 80 -> 78 us (640x480)
#endif
#ifdef EVREP_TU_CORE   // (compiled by the one translation unit that launches it)
static __global__ __launch_bounds__(kCsWaves * kWave) __attribute__((amdgpu_waves_per_eu(CS_WAVES))) void k_col_sort_runs(const int4 *__restrict__ ev, const Rec *__restrict__ sorted1, const int64_t *__restrict__ off,
                                                                     const uint32_t *__restrict__ table,
                                                                     const BlockStats *__restrict__ stats, int H, int W, int nblk,
                                                                     int nchunk, int kpr, int chunk_shift, int by_key,
                                                                     Rec *__restrict__ sorted2,
                                                                     uint32_t *__restrict__ chunk_off, WindowMeta *__restrict__ meta) {
    // by_key (runs of k_block_keysort, kpr = nchunk): the wave's unit is one KEY = (row, 128-pixel chunk) instead of a
    // whole sensor row -- nchunk times the waves, each with a register-resident unit and 128 column counters; what dense
    // windows want (a 1280-pixel row of a 10^6-event window holds ~1400 records: six register batches walked twice).
    extern __shared__ __align__(16) uint32_t cnt_all[];  // [kCsWaves][col_sort_wave_words(unit width)]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.y, unit = blockIdx.x * (int)(blockDim.x >> 6) + wave;   // (the row form is launched with kCsRowWaves waves)
    if (unit >= (by_key ? H * kpr : H)) return;
    const int row = by_key ? unit / kpr : unit;
    const int ck = by_key ? unit - row * kpr : 0;
    const int Wu = by_key ? min(kChunkPx, W - ck * kChunkPx) : W;  // columns of the unit
    uint32_t *cnt = cnt_all + (size_t)wave * col_sort_wave_words(by_key ? kChunkPx : W);
    const int64_t beg = off[b];
    const int64_t n_win = off[b + 1] - beg;
    const int nb = min((int)(((uint32_t)n_win + (1u << chunk_shift) - 1u) >> chunk_shift), nblk);  // <= kCsMaxRuns block runs of 1 << chunk_shift events
    // run k of the row = sorted1[beg + (k << chunk_shift) + t_k[row], ... + t_k[row + 1]); records of earlier rows = sum_k t_k[row]
    // kpr = keys per row of the run table: 1 after k_block_rowsort; nchunk after k_block_keysort, whose runs hold a
    // row's records chunk by chunk (time-ordered inside a chunk, which is all the stable column sort below needs)
    uint32_t ta[2] = {0u, 0u}, tb[2] = {0u, 0u};  // [0]: run `lane`, [1]: run `lane + 64`; a = offset of the row, b = of the next row
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int k = lane + h * kWave;
        if (k < nb) {
            const uint32_t *tk = table + ((size_t)b * nblk + k) * ((size_t)H * kpr + 1);
            ta[h] = tk[by_key ? (size_t)unit : (size_t)row * kpr];
            tb[h] = tk[by_key ? (size_t)unit + 1 : (size_t)(row + 1) * kpr];
        }
    }
    if (unit == 0) {  // this wave also publishes the window's statistics
        BlockStats st;
        stats_identity(st);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int k = lane + h * kWave;
            if (k < nb) {  // three 16-byte loads, fields merged one by one (a struct copy would go through scratch)
                const int4 *sp = reinterpret_cast<const int4 *>(stats + (size_t)b * nblk + k);
                const int4 q0 = sp[0], q1 = sp[1], q2 = sp[2];
                st.tmin = min(st.tmin, q0.x); st.tmax = max(st.tmax, q0.y); st.xmin = min(st.xmin, q0.z); st.xmax = max(st.xmax, q0.w);
                st.ymin = min(st.ymin, q1.x); st.ymax = max(st.ymax, q1.y);
                st.neg_flags |= (uint32_t)q1.z; st.oob_flags |= (uint32_t)q1.w;
                st.status |= (uint32_t)q2.x; st.n_valid += q2.y;
            }
        }
        stats_wave_reduce(st);
        if (lane == 0) {
            WindowMeta m;
            m.tmin = st.tmin; m.tmax = st.tmax; m.xmin = st.xmin; m.xmax = st.xmax; m.ymin = st.ymin; m.ymax = st.ymax;
            m.neg_flags = st.neg_flags; m.oob_flags = st.oob_flags; m.status = st.status & 0xffffu; m.n_valid = st.n_valid;   // (upper half: the library's own bits)
            if (n_win <= 0) m.status |= EVREP_ST_EMPTY;
            else if (st.tmin == st.tmax) m.status |= EVREP_ST_FLAT_TIME;
            for (int i = 0; i < 6; ++i) m.pad[i] = 0;
            meta[b] = m;
        }
    }
    constexpr uint32_t kSuper = 4 * kWave;  // records of a row held in registers (longer rows are walked twice)
    const uint32_t rbeg = (uint32_t)beg + (uint32_t)wave_sum((int)(ta[0] + ta[1]));
    // exclusive prefix of the run lengths over the <= 128 runs: two wave scans, the second carried by the first's total
    const uint32_t len0 = tb[0] - ta[0], len1 = tb[1] - ta[1];
    uint32_t incl0 = len0, incl1 = len1;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o0 = __shfl_up(incl0, d, 64), o1 = __shfl_up(incl1, d, 64);
        if (lane >= d) { incl0 += o0; incl1 += o1; }
    }
    const uint32_t tot0 = (uint32_t)__builtin_amdgcn_readlane((int)incl0, 63);
    incl1 += tot0;
    const uint32_t n = (uint32_t)__builtin_amdgcn_readlane((int)incl1, 63);  // records of the row
    const uint32_t pre0 = incl0 - len0, pre1 = incl1 - len1;                   // the row's records in earlier runs
    const uint32_t src0 = (uint32_t)beg + ((uint32_t)lane << chunk_shift) + ta[0] - pre0;             // record j of the row, if in this run: src + j
    const uint32_t src1 = (uint32_t)beg + ((uint32_t)(lane + kWave) << chunk_shift) + ta[1] - pre1;
    const uint32_t rend = rbeg + n;
    // record j lies in the run k with pre_k <= j < pre_{k+1}: <= kBsChainBlocks runs: a sum of conditional steps over the
    // runs (lanes >= nb hold pre = n, so they never match a j < n); more: pre / src go through 2 x 128 words of the wave's
    // LDS and every lane finds its run by a 7-step binary search -- pre is non-decreasing, the LAST k with pre_k <= j
    // holds record j
    uint32_t *runs = cnt + col_sort_words(by_key ? kChunkPx : W);  // [2][128], only used when nb > kBsChainBlocks
    if (nb > kBsChainBlocks) {
        runs[lane] = pre0;
        runs[kWave + lane] = pre1;
        runs[kCsMaxRuns + lane] = src0;
        runs[kCsMaxRuns + kWave + lane] = src1;
        wave_phase_lds();
    }
    // the runs of k_block_keysort hold 8-byte records (Rec8): the wave's key supplies row and chunk
    const Rec8 *sorted8 = reinterpret_cast<const Rec8 *>(sorted1);
    const int4 *evw = ev + beg;
    auto load = [&](uint32_t at) -> Rec {
        return by_key ? rec8_unpack(sorted8[at], row * W, ck * kChunkPx, evw) : sorted1[at];
    };
    auto fetch = [&](uint32_t j) -> Rec {
        if (nb <= kBsChainBlocks) {
            uint32_t s = (uint32_t)__builtin_amdgcn_readlane((int)src0, 0);
            uint32_t prev = s;
            for (int k = 1; k < nb; ++k) {
                const uint32_t pk = (uint32_t)__builtin_amdgcn_readlane((int)pre0, k);
                const uint32_t sk = (uint32_t)__builtin_amdgcn_readlane((int)src0, k);
                s += (j >= pk) ? sk - prev : 0u;
                prev = sk;
            }
            return load(s + j);
        }
        uint32_t lo = 0, hi = (uint32_t)nb;
#pragma unroll
        for (int step = 0; step < 7; ++step) {
            const uint32_t mid = (lo + hi) >> 1;
            const bool go = hi - lo > 1 && runs[mid] <= j;
            if (hi - lo > 1) { if (go) lo = mid; else hi = mid; }
        }
        return load(runs[kCsMaxRuns + lo] + j);
    };
    Rec e[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t j = i * kWave + lane;
        e[i] = make_int4(0, 0, 0, 0);
        if (j < n) e[i] = fetch(j);
    }
    const int nbits = bits_for(Wu);
    const int per4 = col_sort_per4(Wu);  // 16-byte vectors of column counters per lane (the array is padded to 64 * per4 * 4)
    uint4 *cnt4 = reinterpret_cast<uint4 *>(cnt) + (size_t)lane * per4;
    volatile uint32_t *vcnt = cnt;
    uint32_t *co = chunk_off + ((size_t)b * H + row) * (nchunk + 1);
    if (by_key && lane == 0) {  // the unit IS one chunk: its offset, and the row's end behind the last chunk
        co[ck] = rbeg;
        if (ck == kpr - 1) co[kpr] = rend;
    }
    if (n == 0) {
        if (!by_key) for (int c = lane; c <= nchunk; c += kWave) co[c] = rbeg;
        return;
    }
    const int rowbase = row * W + ck * kChunkPx;
    const uint32_t nsuper = (n + kSuper - 1) / kSuper;
    for (int k = 0; k < per4; ++k) cnt4[k] = make_uint4(0u, 0u, 0u, 0u);
    wave_phase_lds();
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if ((uint32_t)(i * kWave + lane) < n) atomicAdd(&cnt[e[i].x - rowbase], 1u);
    for (uint32_t sb = 1; sb < nsuper; ++sb)  // a row longer than the register batch
        for (int i = 0; i < 4; ++i) {
            const uint32_t j = sb * kSuper + i * kWave + lane;
            if (j < n) atomicAdd(&cnt[fetch(j).x - rowbase], 1u);
        }
    wave_phase_lds();
    // exclusive scan over the W column counters: `per` consecutive columns per lane
    uint32_t local = 0;
    for (int k = 0; k < per4; ++k) { const uint4 v = cnt4[k]; local += v.x + v.y + v.z + v.w; }
    uint32_t inc2 = local;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(inc2, d, 64); if (lane >= d) inc2 += o; }
    uint32_t run = inc2 - local;
    for (int k = 0; k < per4; ++k) {
        const uint4 v = cnt4[k];
        uint4 o;
        o.x = run; o.y = o.x + v.x; o.z = o.y + v.y; o.w = o.z + v.z;
        run = o.w + v.w;
        cnt4[k] = o;
    }
    wave_phase_lds();
    if (!by_key) for (int c = lane; c <= nchunk; c += kWave) co[c] = (c * kChunkPx < W) ? rbeg + cnt[c * kChunkPx] : rend;
    wave_phase_lds();
    for (uint32_t sb = 0; sb < nsuper; ++sb) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t j0 = sb * kSuper + i * kWave;
            if (j0 >= n) break;  // uniform
            const bool valid = j0 + lane < n;
            Rec rec = e[i];
            if (sb > 0) { rec = make_int4(0, 0, 0, 0); if (valid) rec = fetch(j0 + lane); }
            const uint32_t col = valid ? (uint32_t)(rec.x - rowbase) : 0u;
            uint32_t rk; bool last;
            wave_match(col, nbits, valid, lane, rk, last);
            uint32_t pos = 0;
            if (valid) {
                pos = vcnt[col] + rk;
                sorted2[rbeg + pos] = rec;
            }
            __builtin_amdgcn_wave_barrier();
            if (valid && last) vcnt[col] = pos + 1;
            __builtin_amdgcn_wave_barrier();
        }
    }
}
#endif

// =====================================================================================================
// Key-sorted binning pass (plan->reserved == 2): ONE kernel.  A workgroup orders its 8192 events by the KEY
// (sensor row, 128-pixel chunk) -- the unit a builder wave owns -- and the builder wave itself gathers its
// unit's records from the window's block runs and finishes the order by pixel inside its LDS
// (evrep_builders.hip, unit_records).  The column sort kernel, its launch boundary, and one write + one read of
// every record disappear from the step; sorted2 / chunk_off are only produced on demand
// (k_col_sort_runs with kpr = nchunk) for the few consumers that walk the pixel-sorted stream directly.
//
// The order inside a key must be the time order.  Instead of per-wave counter tables (16 waves x keys would not
// fit next to the record stage) the block counts with ONE table of LDS atomics, whose returned values order the
// arrivals arbitrarily, and then repairs every key group: a record's place inside its group = the number of
// group members with a smaller rank, read from a 16-bit rank array (groups hold 3-4 records on the headline
// windows; a group of g records costs g reads per member, so a hot pixel makes its own block slower, never
// wrong).  Nothing depends on another workgroup.
// =====================================================================================================
#ifndef KS_DIRECT_WRITE
#define KS_DIRECT_WRITE 0
#endif
#ifndef KS_DEBUG
#define KS_DEBUG 0  // timing experiments only (tools/experiments): 1 = no group repair, 2 = no stage / write-out, 4 = no statistics, 8 = no table copy
#endif
constexpr int kKsHotGroup = 16;   // key groups of more records (per block) are ordered by per-wave counters, not by the repair walk
__host__ __device__ inline size_t block_keysort_lds_bytes(int NK, int cap, int chunk) {
    return (size_t)cap * sizeof(Rec8) + (size_t)chunk * sizeof(uint16_t) + (size_t)(NK + 4) * sizeof(uint32_t);
}

// grid (8 * ceil(B/8) * nblk), TPB threads = TPB * 8 events per workgroup (1024 -> 8192, 512 -> 4096: windows of up to
// 16 x 4096 events take the smaller block -- twice the workgroups, two or three resident per CU, half-size key
// groups to repair), dynamic LDS = block_keysort_lds_bytes(H * kpr, cap, TPB * 8); cap = records the stage holds
// (the block is written out in rounds of cap records).
// table: [B][nblk][H * kpr + 1] exclusive offsets of the block's keys inside its run (last entry = in-frame events).
// Occupancy asked of the compiler (r03): the kernel is a chain of barrier-separated latencies, so a second resident
// workgroup per CU fills them.  1024-thread instance (dense windows): 8 waves per SIMD = TWO workgroups per CU (64 VGPRs,
// 36 bytes of scratch; left alone the compiler takes 114 VGPRs = one workgroup per CU): binning of 4 x 10^6-event 1280x720
// windows 145 -> 123 us, of 8 x 500 000-event 640x480 windows 91 -> 80 us.  512-thread instance: 6 (80 VGPRs, no scratch).
#ifndef KS_WAVES_512
#define KS_WAVES_512 6
#endif
#ifndef KS_WAVES_1024
#define KS_WAVES_1024 8
#endif
template <int TPB, int PL = 8>   // PL = events per lane: a workgroup orders TPB * PL events
__global__ __launch_bounds__(TPB) __attribute__((amdgpu_waves_per_eu(TPB == 1024 ? KS_WAVES_1024 : KS_WAVES_512))) void k_block_keysort(const int4 *__restrict__ ev, const int64_t *__restrict__ off,
                                                             int B, int H, int W, int kpr, int nblk, int cap,
                                                             uint32_t *__restrict__ table, BlockStats *__restrict__ stats,
                                                             Rec *__restrict__ sorted1, int64_t *__restrict__ nwin,
                                                             uint32_t *__restrict__ hot) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    constexpr int kChunk = TPB * PL, kNW = TPB / kWave, kPerWave = PL * kWave;
    if (blockIdx.x == 0 && threadIdx.x < 128) hot[threadIdx.x * 16] = 0u;   // the builders' hot lists (2 x 64 sublist counters) start empty
    const int NK = H * kpr;
    Rec8 *stage = reinterpret_cast<Rec8 *>(smem_raw);                       // [cap], output order, 8-byte records
    uint16_t *rankbuf = reinterpret_cast<uint16_t *>(stage + cap);          // [kChunk] rank inside the block, arrival order
    uint32_t *base = reinterpret_cast<uint32_t *>(rankbuf + kChunk);        // [NK + 1] counts -> exclusive offsets
    __shared__ int bstats[12];  // the block's statistics, merged by one LDS atomic per wave and field (BlockStats order)
    __shared__ uint32_t tmp[kNW];
    __shared__ uint32_t nhot_s;  // key groups of more than kKsHotGroup records in this block (see "hot groups" below)
    int b, blk;
    if (!decode_window_block(B, nblk, b, blk)) return;
    const int64_t beg = off[b];
    const int64_t n = off[b + 1] - beg;
    const int64_t lo = (int64_t)blk * kChunk;
    if (blk == 0 && threadIdx.x == 0) nwin[b] = n;  // for k_window_meta, which is not handed the offsets
    if (lo >= n) return;  // the builders only read the blocks a window really has
    const int nblock = (int)((lo + kChunk < n) ? kChunk : n - lo);  // events of this block; everything below is 32-bit
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int w0 = wave * kPerWave;                       // first block-local index of the wave
    const int lo32 = (int)lo;
    const int4 *evb = ev + beg + lo;
    int4 e[PL];
    int tprev[PL];
#pragma unroll
    for (int i = 0; i < PL; ++i) {
        const int li = w0 + i * kWave + lane;
        e[i] = make_int4(-1, -1, INT32_MAX, 0);
        tprev[i] = INT32_MIN;
        if (li < nblock) {
            e[i] = evb[li];
            if (lane == 0 && (li > 0 || lo > 0)) tprev[i] = evb[li - 1].z;  // other lanes take it from their neighbour
        }
    }
    for (int i = threadIdx.x; i <= NK; i += TPB) base[i] = 0;
    if (threadIdx.x == 0) nhot_s = 0u;
    if (threadIdx.x < 12) {
        const int f = threadIdx.x;  // tmin, tmax, xmin, xmax, ymin, ymax, neg, oob, status, n_valid, pad, pad
        bstats[f] = (f == 0 || f == 2 || f == 4) ? INT32_MAX : ((f == 1 || f == 3 || f == 5) ? INT32_MIN : 0);
    }
    __syncthreads();
    const MdesWindows mw = mdes_windows(n);
    uint32_t full = 0, part = 0;
#pragma unroll
    for (int i = 0; i < 7; ++i) {
        if (mw.lo[i] <= lo32 && lo32 + nblock <= mw.hi[i]) full |= 1u << i;
        else if (!(lo32 + nblock <= mw.lo[i] || lo32 >= mw.hi[i])) part |= 1u << i;
    }
    BlockStats st;
    stats_identity(st);
    uint32_t ko[PL];  // key | arrival << 16, then the record's place in the block; 0xffffffff = not placed
    uint32_t sneg = 0;
    bool cut = false;
    const int a0 = lo32 + w0, a1 = min(lo32 + w0 + kPerWave, lo32 + nblock);  // the wave's rank range
    if (part) {
#pragma unroll
        for (int q = 0; q < 7; ++q) cut |= (mw.lo[q] > a0 && mw.lo[q] < a1) || (mw.hi[q] > a0 && mw.hi[q] < a1);
    }
    const uint32_t memb_u = cut ? 0u : (full | (part ? (mdes_membership(mw, a0) & part) : 0u));
    const int HW = H * W;  // < 2^30
#pragma unroll
    for (int i = 0; i < PL; ++i) {
        const int li = w0 + i * kWave + lane;
        const bool in = li < nblock;
        const int up = __builtin_amdgcn_update_dpp(0, e[i].z, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
        if (lane != 0) tprev[i] = up;
        // in-frame test on the flat index x + y*W, as the reference's scatter sees it (see k_block_rowsort)
        uint32_t row = (uint32_t)e[i].y, col = (uint32_t)e[i].x;
        bool valid = in && (uint32_t)e[i].x < (uint32_t)W && (uint32_t)e[i].y < (uint32_t)H;
        if (__any(in && (uint32_t)e[i].x >= (uint32_t)W)) {
            const int64_t key = (int64_t)e[i].x + (int64_t)e[i].y * W;
            valid = in && key >= 0 && key < (int64_t)HW;
            if (valid) { row = (uint32_t)key / (uint32_t)W; col = (uint32_t)key - row * (uint32_t)W; }
        }
        ko[i] = 0xffffffffu;
        if (valid) {
            const uint32_t key = row * (uint32_t)kpr + (col >> 7);
            ko[i] = key | (atomicAdd(&base[key], 1u) << 16);
        }
        if (KS_DEBUG & 4) continue;
        if (!cut) {
            if (__any(in && e[i].w == -1)) sneg |= memb_u;
        } else if (in && e[i].w == -1) {
            st.neg_flags |= full | (mdes_membership(mw, lo32 + li) & part);
        }
        if (__any(in && !valid)) {  // rare: out-of-frame events
            if (in && !valid) {
                const uint32_t memb = cut ? (full | (mdes_membership(mw, lo32 + li) & part)) : memb_u;
                st.status |= EVREP_ST_OOB;
                const int cls = e[i].w == 1 ? 1 : (e[i].w == -1 ? 2 : (e[i].w == 0 ? 3 : 0));
                st.oob_flags |= memb | (cls ? (memb << (7 * cls)) : 0u);
            }
        }
        if (in) {
            if (valid) ++st.n_valid;
            if (tprev[i] > e[i].z) st.status |= EVREP_ST_UNSORTED;
            if ((uint32_t)(e[i].w + 1) > 2u) st.status |= kStEscaped;   // (rec8_pack escapes it: builders that hand hot units to a split sweep need to know)
            st.tmin = min(st.tmin, e[i].z); st.tmax = max(st.tmax, e[i].z);
            st.xmin = min(st.xmin, e[i].x); st.xmax = max(st.xmax, e[i].x);
            st.ymin = min(st.ymin, e[i].y); st.ymax = max(st.ymax, e[i].y);
        }
    }
    st.neg_flags |= sneg;
    stats_wave_reduce(st);
    if (lane == 0) {
        atomicMin(&bstats[0], st.tmin); atomicMax(&bstats[1], st.tmax);
        atomicMin(&bstats[2], st.xmin); atomicMax(&bstats[3], st.xmax);
        atomicMin(&bstats[4], st.ymin); atomicMax(&bstats[5], st.ymax);
        atomicOr(reinterpret_cast<uint32_t *>(&bstats[6]), st.neg_flags);
        atomicOr(reinterpret_cast<uint32_t *>(&bstats[7]), st.oob_flags);
        atomicOr(reinterpret_cast<uint32_t *>(&bstats[8]), st.status);
        atomicAdd(&bstats[9], st.n_valid);
    }
    __syncthreads();
    // exclusive scan over the key counters: `per` consecutive keys per thread
    const int per = (NK + TPB - 1) / TPB;
    const int k0 = (int)threadIdx.x * per;
    uint32_t local = 0;
    for (int k = 0; k < per; ++k) if (k0 + k < NK) local += base[k0 + k];
    uint32_t total;
    uint32_t run = block_exclusive_scan<kNW>(local, tmp, &total);
    // (offsets are < 2^14: the upper half of a key's word carries its hot-group slot + 1, 0 = an ordinary group)
    for (int k = 0; k < per; ++k)
        if (k0 + k < NK) {
            const uint32_t c = base[k0 + k];
            uint32_t tag = 0u;
            if (c > (uint32_t)kKsHotGroup) tag = (atomicAdd(&nhot_s, 1u) + 1u) << 16;
            base[k0 + k] = run | tag;
            run += c;
        }
    if (threadIdx.x == 0) base[NK] = total;
    if (threadIdx.x < 12) reinterpret_cast<int *>(stats + (size_t)b * nblk + blk)[threadIdx.x] = bstats[threadIdx.x];
    __syncthreads();
    uint32_t *tb = table + ((size_t)b * nblk + blk) * ((size_t)NK + 1);
    if (!(KS_DEBUG & 8))
        for (int k = threadIdx.x; k <= NK; k += TPB) tb[k] = base[k] & 0xffffu;
    const uint32_t nhot = nhot_s;   // block-uniform (written before the barrier above)
    // arrival order -> time order inside every key group
    const uint32_t mine0 = (uint32_t)(w0 + lane);
#pragma unroll
    for (int i = 0; i < PL; ++i) {
        if (ko[i] != 0xffffffffu) {
            const uint32_t bw = base[ko[i] & 0xffffu];   // upper half: hot-group slot + 1, 0 = an ordinary group
            if (!(bw >> 16)) rankbuf[(bw & 0xffffu) + (ko[i] >> 16)] = (uint16_t)(mine0 + i * kWave);
        }
    }
    // Hot groups (r04).  The repair below reads g ranks per member of a g-record group -- fine for the 2-4 records a key holds
    // on uniform windows, quadratic for a key that collects hundreds (a moving edge parallel to the sensor rows puts 4 % of a
    // window into one 128-pixel chunk: binning of 32 such windows 88 us instead of 20).  A group of more than kKsHotGroup
    // records is ordered the way k_block_rowsort orders its rows instead: one counter per (wave, hot group) -- there are at
    // most kChunk / (kKsHotGroup + 1) hot groups, so the counters fit the record stage, which is idle until the write-out --
    // an exclusive prefix over the waves, and a ballot multisplit inside the wave, batch by batch (rank = wave-major, then
    // batch, then lane: exactly the block rank order).  Linear in the group's size.
    if (nhot) {
        constexpr int kHs = kChunk / 16;                                // slots per wave row; nhot < kHs
        uint32_t *hcnt = reinterpret_cast<uint32_t *>(stage);           // [kNW][kHs], cap * 8 >= kNW * kHs * 4
        for (uint32_t t = lane; t < nhot; t += kWave) hcnt[wave * kHs + t] = 0u;
        wave_phase_lds();
#pragma unroll
        for (int i = 0; i < PL; ++i)
            if (ko[i] != 0xffffffffu) {
                const uint32_t hsl = base[ko[i] & 0xffffu] >> 16;
                if (hsl) atomicAdd(&hcnt[wave * kHs + hsl - 1u], 1u);
            }
        __syncthreads();
        if (threadIdx.x < nhot) {
            uint32_t acc = 0;
            for (int w = 0; w < kNW; ++w) { const uint32_t c = hcnt[w * kHs + threadIdx.x]; hcnt[w * kHs + threadIdx.x] = acc; acc += c; }
        }
        __syncthreads();
        const int hbits = bits_for((int)nhot);
        volatile uint32_t *vh = hcnt + wave * kHs;
#pragma unroll
        for (int i = 0; i < PL; ++i) {
            const uint32_t bw = ko[i] != 0xffffffffu ? base[ko[i] & 0xffffu] : 0u;
            const bool hot = (bw >> 16) != 0u;
            if (!__any(hot)) continue;
            const uint32_t slot = hot ? (bw >> 16) - 1u : 0u;
            uint32_t rk; bool last;
            wave_match(slot, hbits, hot, lane, rk, last);
            uint32_t pos = 0;
            if (hot) {
                pos = vh[slot] + rk;
                ko[i] = ((bw & 0xffffu) + pos) | 0x80000000u;   // final place; bit 31 marks it for the repair below (cleared there)
            }
            __builtin_amdgcn_wave_barrier();
            if (hot && last) vh[slot] = pos + 1;
            __builtin_amdgcn_wave_barrier();
        }
    }
    __syncthreads();
    {
        // the eight group walks of a lane advance together: eight independent LDS reads in flight per step instead of
        // one dependent read per step (the walk is latency-bound: 8 -> 2 us per block on the headline windows)
        uint32_t gb[PL], g[PL], sm[PL];
        uint32_t gmax = 0;
#pragma unroll
        for (int i = 0; i < PL; ++i) {
            gb[i] = 0; g[i] = 0; sm[i] = 0;
            if (ko[i] != 0xffffffffu && !(ko[i] >> 31)) {
                const uint32_t key = ko[i] & 0xffffu;
                const uint32_t bw = base[key];
                gb[i] = bw & 0xffffu;
                g[i] = (bw >> 16) ? 0u : (base[key + 1] & 0xffffu) - gb[i];   // (a hot group without hot placement cannot occur: nhot counts them all)
                if (KS_DEBUG & 1) { sm[i] = ko[i] >> 16; g[i] = 0; }
                if (g[i] == 1) g[i] = 0;  // alone in its group
                gmax = max(gmax, g[i]);
            }
        }
        gmax = (uint32_t)wave_max((int)gmax);
        for (uint32_t j = 0; j < gmax; ++j) {
#pragma unroll
            for (int i = 0; i < PL; ++i)
                if (j < g[i]) sm[i] += (uint32_t)rankbuf[gb[i] + j] < mine0 + (uint32_t)(i * kWave) ? 1u : 0u;
        }
#pragma unroll
        for (int i = 0; i < PL; ++i)
            if (ko[i] != 0xffffffffu) ko[i] = (ko[i] >> 31) ? (ko[i] & 0x7fffffffu) : gb[i] + sm[i];
    }
    if (nhot) __syncthreads();   // the hot-group counters live in the stage: every wave has read its own before the write-out
    Rec8 *dst = reinterpret_cast<Rec8 *>(sorted1) + beg + lo;   // the block's own slot, 8 bytes per record
#if KS_DIRECT_WRITE
    // experiment: the records go from the lanes' registers straight to their places in the block's slot (scattered 8-byte stores
    // that meet in the XCD's L2 and leave it as full lines) -- no stage, no barrier, no second pass over the block
#pragma unroll
    for (int i = 0; i < PL; ++i) {
        if (ko[i] != 0xffffffffu && !(KS_DEBUG & 2)) {
            int col = e[i].x;
            if ((uint32_t)col >= (uint32_t)W) { const uint32_t key = (uint32_t)(e[i].x + e[i].y * W); col = (int)(key % (uint32_t)W); }
            dst[ko[i]] = rec8_pack(col, lo32 + w0 + i * kWave + lane, e[i].z, e[i].w);
        }
    }
    return;
#endif
    for (uint32_t pb = 0; pb < total && !(KS_DEBUG & 2); pb += (uint32_t)cap) {
        if (pb) __syncthreads();  // the previous round has left the stage
#pragma unroll
        for (int i = 0; i < PL; ++i) {
            const uint32_t pos = ko[i] - pb;  // 0xffffffff - pb >= cap for every pb < 8192
            if (pos < (uint32_t)cap) {  // placed records are in frame: their column is x, or (x + y*W) mod W for an x >= W
                int col = e[i].x;
                if ((uint32_t)col >= (uint32_t)W) { const uint32_t key = (uint32_t)(e[i].x + e[i].y * W); col = (int)(key % (uint32_t)W); }
                stage[pos] = rec8_pack(col, lo32 + w0 + i * kWave + lane, e[i].z, e[i].w);
            }
        }
        __syncthreads();
        const uint32_t cnt = min((uint32_t)cap, total - pb);
        for (uint32_t t = threadIdx.x; t < cnt; t += TPB) dst[pb + t] = stage[t];
    }
}

// grid (B), 64 threads: the window statistics of the key-sorted pass, for the synchronous read-backs only
// (the builders merge the block statistics they need themselves).
#ifdef EVREP_TU_CORE   // (compiled by the one translation unit that launches it)
static __global__ __launch_bounds__(kWave) void k_window_meta(const int64_t *__restrict__ nwin, const BlockStats *__restrict__ stats,
                                                      int nblk, int chunk_shift, WindowMeta *__restrict__ meta) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const int64_t n_win = nwin[b];
    const int nb = min((int)(((uint32_t)n_win + (1u << chunk_shift) - 1u) >> chunk_shift), nblk);
    BlockStats st;
    stats_identity(st);
    for (int k = lane; k < nb; k += kWave) stats_merge(st, stats[(size_t)b * nblk + k]);
    stats_wave_reduce(st);
    if (lane == 0) {
        WindowMeta m;
        m.tmin = st.tmin; m.tmax = st.tmax; m.xmin = st.xmin; m.xmax = st.xmax; m.ymin = st.ymin; m.ymax = st.ymax;
        m.neg_flags = st.neg_flags; m.oob_flags = st.oob_flags; m.status = st.status & 0xffffu; m.n_valid = st.n_valid;   // (upper half: the library's own bits)
        if (n_win <= 0) m.status |= EVREP_ST_EMPTY;
        else if (st.tmin == st.tmax) m.status |= EVREP_ST_FLAT_TIME;
        for (int i = 0; i < 6; ++i) m.pad[i] = 0;
        meta[b] = m;
    }
}
#endif

}  // namespace evrep
