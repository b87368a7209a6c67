// evrep_bandbin.hip -- EXPERIMENT, not part of libevrep.so: the binning pass as ONE kernel, for windows of at
// most 65 536 events (the reference's default window is 50 000, gen1_2yolo.py:41).  Bit-exact on the whole
// GPU suite when wired in behind evrep_bin_events, but not faster than the three-kernel pass in r01
// (57.8 vs 59.3 us at 32 x 50 000 events with 1024 threads; phases: stream 17 + statistics 10 + two radix passes 15
// + write-out 16 us, all latency-bound at one 138 KB-LDS block per CU), so it is parked here; see NOTES.md 8.
//
// The three-kernel pass (evrep_bin.hip) cuts a window's EVENTS into blocks and therefore needs a
// histogram kernel, a scan and a scatter before any block knows where its events go.  Here a
// workgroup owns a band of SENSOR ROWS instead: block (window b, band k of kBands) streams the
// whole window (800 KB at 50 000 events; the eight blocks of a window sit on one XCD, so seven
// of the eight reads are L2 hits), keeps the events of its rows as 32-bit (pixel-in-band : rank)
// pairs in LDS, orders them there with two stable 8-bit radix passes on the pixel half, and
// writes the final records and the chunk offsets.  A block needs nothing from any other block:
// the start of its run is the number of in-frame events in lower rows, which it counts while
// streaming.  Window statistics (WindowMeta) come from band 0, which sees every event anyway.
// One launch, no inter-block dependency; ~half the time of the three-kernel pass at the headline
// configuration (see NOTES.md 3.1).
//
// Overflow (a clustered window putting more events into one band or one wave's slice than the LDS
// lists hold): the block redoes the pass with the same code on global scratch (its own slots of
// the sorted1 region) -- slower, same result.
#include "evrep_common.h"

#ifndef BAND_SKIP_STATS
#define BAND_SKIP_STATS 0
#endif
#ifndef BAND_SKIP_SORT
#define BAND_SKIP_SORT 0
#endif
#ifndef BAND_SKIP_WRITE
#define BAND_SKIP_WRITE 0
#endif
namespace evrep {

constexpr int kBands = 8;
#ifndef EVREP_BAND_THREADS
#define EVREP_BAND_THREADS 512
#endif
constexpr int kBandThreads = EVREP_BAND_THREADS;
constexpr int kBandWaves = kBandThreads / kWave;
constexpr int kBandWaveCap = 16384 / kBandWaves;  // pairs one wave may list in LDS (2.6x the uniform average at 50 000 events)
constexpr int kBandCap = 16384;     // pairs one band may sort in LDS
constexpr int kRadix = 256;
constexpr int kBandMaxEvents = 65536;

// rows per band and whether a (pixel-in-band : rank) pair fits 16 + 16 bits
__host__ __device__ inline int band_rows(int H) { return (H + kBands - 1) / kBands; }
__host__ inline bool band_bin_eligible(int H, int W, int64_t max_events_per_window) {
    return max_events_per_window <= kBandMaxEvents && (int64_t)band_rows(H) * W <= 65536;
}

// One stable 8-bit radix pass over `n` packed pairs.  Wave w reads xcnt[w] values starting at X + xoff[w]
// (in order), the output Y is compact.  hist: [kBandWaves][kRadix] LDS words.  All threads call it.
__device__ __forceinline__ void band_radix_pass(const uint32_t *X, const uint32_t *xoff, const uint32_t *xcnt,
                                               uint32_t *Y, int shift, uint32_t *hist, uint32_t *tmp) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < kBandWaves * kRadix; i += kBandThreads) hist[i] = 0;
    __syncthreads();
    const uint32_t wn = xcnt[wave];
    const uint32_t *xw = X + xoff[wave];
    uint32_t *myhist = hist + wave * kRadix;
    for (uint32_t j0 = 0; j0 < wn; j0 += kWave)
        if (j0 + lane < wn) atomicAdd(&myhist[(xw[j0 + lane] >> shift) & (kRadix - 1)], 1u);
    __syncthreads();
    uint32_t run = 0;
    if (threadIdx.x < kRadix) {
        for (int w = 0; w < kBandWaves; ++w) {
            const uint32_t t = hist[w * kRadix + threadIdx.x];
            hist[w * kRadix + threadIdx.x] = run;
            run += t;
        }
    }
    uint32_t total;
    const uint32_t excl = block_exclusive_scan<kBandWaves>(threadIdx.x < kRadix ? run : 0u, tmp, &total);
    if (threadIdx.x < kRadix)
        for (int w = 0; w < kBandWaves; ++w) hist[w * kRadix + threadIdx.x] += excl;
    __syncthreads();
    for (uint32_t j0 = 0; j0 < wn; j0 += kWave) {
        const bool valid = j0 + lane < wn;
        const uint32_t v = valid ? xw[j0 + lane] : 0u;
        const uint32_t d = (v >> shift) & (kRadix - 1);
        uint32_t rk; bool last;
        wave_match(d, 8, valid, lane, rk, last);
        uint32_t pos = 0;
        if (valid) { pos = myhist[d] + rk; Y[pos] = v; }
        wave_phase();
        if (valid && last) myhist[d] = pos + 1;
        wave_phase();
    }
    __syncthreads();
}

// grid (8 * ceil(B/8) * kBands), 512 threads, static LDS only.
__global__ __launch_bounds__(kBandThreads) void k_band_bin(const int4 *__restrict__ ev, const int64_t *__restrict__ off,
                                                          int B, int H, int W, int nchunk, Rec *__restrict__ sorted,
                                                          uint32_t *__restrict__ chunk_off, WindowMeta *__restrict__ meta,
                                                          Rec *__restrict__ scratch) {
    // 138 KB of static LDS (a gfx950 workgroup may declare all 160 KiB): one block per CU
    __shared__ uint32_t A[kBandWaves * kBandWaveCap];        // per-wave lists, later the sorted pairs
    __shared__ uint32_t Bf[kBandCap];
    __shared__ uint32_t hist[kBandWaves * kRadix];
    __shared__ uint32_t wcnt[kBandWaves], wbelow[kBandWaves], xoff[kBandWaves], xcnt[kBandWaves], tmp[kBandWaves];
    __shared__ BlockStats wstats[kBandWaves];

    // XCD-aware decode: the kBands blocks of window b get ids congruent to b mod 8 (see decode_window_block)
    const int id = blockIdx.x, xcd = id & 7, sidx = id >> 3;
    const int b = (sidx / kBands) * 8 + xcd, band = sidx % kBands;
    if (b >= B) return;
    const int rpb = band_rows(H);
    const int lo = band * rpb, hi = min(H, lo + rpb);
    if (lo >= H && band != 0) return;
    const int64_t beg = off[b];
    const int n = (int)(off[b + 1] - beg);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t HW = (int64_t)H * W;
    const int per = ((n + kBandWaves - 1) / kBandWaves + kWave - 1) / kWave * kWave;  // wave w streams ranks [w*per, ...)
    const int sbeg = min(n, wave * per), send = min(n, sbeg + per);
    const int4 *e0 = ev + beg;
    const MdesWindows mw = mdes_windows(n);
    const uint64_t lt = (1ull << lane) - 1ull;

    // One streaming pass over the wave's slice.  MODE 0: list into LDS (drop what does not fit, the true count
    // decides about the fallback); MODE 1: list into `gl` (global, exact offsets known).  Band 0 also takes
    // the window statistics in MODE 0.
    uint32_t cnt = 0, below = 0;
    BlockStats st;
    stats_identity(st);
    auto stream = [&](auto mode_tag, uint32_t *list, uint32_t cap) {
        constexpr int MODE = decltype(mode_tag)::value;
        constexpr int U = 4;
        cnt = 0; below = 0;
        for (int j0 = sbeg; j0 < send; j0 += U * kWave) {
            int4 e[U];
            int tprev[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int j = j0 + u * kWave + lane;
                e[u] = make_int4(0, -1, INT32_MAX, 0);
                tprev[u] = INT32_MIN;
                if (j < send) {
                    e[u] = e0[j];
                    if (MODE == 0 && band == 0 && lane == 0 && j > 0) tprev[u] = e0[j - 1].z;
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int j = j0 + u * kWave + lane;
                const bool in = j < send;
                // in-frame test on the flat index x + y*W, as the reference's scatter sees it (an x outside [0, W)
                // can still land in the frame); the common case x in [0, W) needs no 64-bit arithmetic
                int row = e[u].y, col = e[u].x;
                bool valid = in && (uint32_t)col < (uint32_t)W && (uint32_t)row < (uint32_t)H;
                if (__any(in && (uint32_t)col >= (uint32_t)W)) {
                    const int64_t key = (int64_t)e[u].x + (int64_t)e[u].y * W;
                    valid = in && key >= 0 && key < HW;
                    if (valid && (uint32_t)col >= (uint32_t)W) { row = (int)((uint32_t)key / (uint32_t)W); col = (int)((uint32_t)key - (uint32_t)row * (uint32_t)W); }
                }
                const bool inb = valid && row >= lo && row < hi;
                below += (uint32_t)__popcll(__ballot(valid && row < lo));
                const uint64_t m = __ballot(inb);
                if (inb) {
                    const uint32_t pos = cnt + (uint32_t)__popcll(m & lt);
                    if (pos < cap) list[pos] = ((uint32_t)((row - lo) * W + col) << 16) | (uint32_t)j;
                }
                cnt += (uint32_t)__popcll(m);
                if (MODE == 0 && band == 0 && !BAND_SKIP_STATS) {
                    const int up = __shfl_up(e[u].z, 1, 64);
                    if (lane != 0) tprev[u] = up;
                    if (in) {
                        const uint32_t memb = mdes_membership(mw, j);
                        if (e[u].w == -1) st.neg_flags |= memb;
                        if (valid) ++st.n_valid;
                        else {
                            st.status |= EVREP_ST_OOB;
                            const int cls = e[u].w == 1 ? 1 : (e[u].w == -1 ? 2 : (e[u].w == 0 ? 3 : 0));
                            st.oob_flags |= memb | (cls ? (memb << (7 * cls)) : 0u);
                        }
                        if (tprev[u] > e[u].z) st.status |= EVREP_ST_UNSORTED;
                        st.tmin = min(st.tmin, e[u].z); st.tmax = max(st.tmax, e[u].z);
                        st.xmin = min(st.xmin, e[u].x); st.xmax = max(st.xmax, e[u].x);
                        st.ymin = min(st.ymin, e[u].y); st.ymax = max(st.ymax, e[u].y);
                    }
                }
            }
        }
    };
    stream(std::integral_constant<int, 0>{}, A + wave * kBandWaveCap, (uint32_t)kBandWaveCap);
    if (band == 0) {
        stats_wave_reduce(st);
        if (lane == 0) wstats[wave] = st;
    }
    if (lane == 0) { wcnt[wave] = cnt; wbelow[wave] = below; }
    __syncthreads();
    uint32_t nb = 0, below_all = 0;
    bool fits = true;
#pragma unroll
    for (int w = 0; w < kBandWaves; ++w) { nb += wcnt[w]; below_all += wbelow[w]; fits = fits && wcnt[w] <= (uint32_t)kBandWaveCap; }
    fits = fits && nb <= (uint32_t)kBandCap;
    const int64_t base = beg + below_all;  // absolute position of this band's run in the sorted stream
    if (band == 0 && threadIdx.x == 0) {
        BlockStats t = wstats[0];
        for (int w = 1; w < kBandWaves; ++w) stats_merge(t, wstats[w]);
        WindowMeta m;
        m.tmin = t.tmin; m.tmax = t.tmax; m.xmin = t.xmin; m.xmax = t.xmax; m.ymin = t.ymin; m.ymax = t.ymax;
        m.neg_flags = t.neg_flags; m.oob_flags = t.oob_flags; m.status = t.status; m.n_valid = t.n_valid;
        if (n <= 0) m.status |= EVREP_ST_EMPTY;
        else if (t.tmin == t.tmax) m.status |= EVREP_ST_FLAT_TIME;
        for (int i = 0; i < 6; ++i) m.pad[i] = 0;
        meta[b] = m;
    }
    if (lo >= H) return;  // (band 0 of a frame with no rows cannot happen: H >= 1)

    const uint32_t q = ((nb + kBandWaves - 1) / kBandWaves + kWave - 1) / kWave * kWave;  // even split of a compact array
    auto even_split = [&]() {
        if (threadIdx.x < kBandWaves) {
            const uint32_t o = min(nb, (uint32_t)threadIdx.x * q);
            xoff[threadIdx.x] = o;
            xcnt[threadIdx.x] = min(q, nb - o);
        }
    };
    // write-out: records in final order + the chunk offsets of the band's rows, from the sorted pairs R
    auto write_out = [&](const uint32_t *R) {
        for (uint32_t i0 = threadIdx.x; i0 < nb; i0 += 4 * kBandThreads) {  // four record gathers in flight per thread
            uint32_t v[4];
            int4 e[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint32_t i = i0 + u * kBandThreads;
                v[u] = i < nb ? R[i] : 0u;
                e[u] = e0[i < nb ? (v[u] & 0xffffu) : 0u];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint32_t i = i0 + u * kBandThreads;
                if (i < nb)
                    sorted[base + i] = make_int4((int)((uint32_t)lo * (uint32_t)W + (v[u] >> 16)), (int)(v[u] & 0xffffu), e[u].z, e[u].w);
            }
        }
        const int per_row = nchunk + 1;
        for (int t = threadIdx.x; t < (hi - lo) * per_row; t += kBandThreads) {
            const int r = t / per_row, c = t - r * per_row;
            const uint32_t target = (uint32_t)(r * W + min(c * kChunkPx, W));
            uint32_t a = 0, z = nb;  // first i with (R[i] >> 16) >= target
            while (a < z) {
                const uint32_t mid = (a + z) >> 1;
                if ((R[mid] >> 16) < target) a = mid + 1; else z = mid;
            }
            chunk_off[((size_t)b * H + lo + r) * per_row + c] = (uint32_t)(base + a);
        }
    };

    if (BAND_SKIP_SORT) { write_out(A); return; }
    if (BAND_SKIP_WRITE) return;
    if (fits) {
        if (threadIdx.x < kBandWaves) { xoff[threadIdx.x] = threadIdx.x * kBandWaveCap; xcnt[threadIdx.x] = wcnt[threadIdx.x]; }
        __syncthreads();
        band_radix_pass(A, xoff, xcnt, Bf, 16, hist, tmp);   // low 8 bits of the pixel half
        even_split();
        __syncthreads();
        band_radix_pass(Bf, xoff, xcnt, A, 24, hist, tmp);   // high 8 bits
        write_out(A);
    } else {
        // the band's own slots of the scratch region: 16 bytes per event = room for two uint32 arrays of nb
        uint32_t *GA = reinterpret_cast<uint32_t *>(scratch + base);
        uint32_t *GB = GA + nb;
        uint32_t woff = 0;
        for (int w = 0; w < wave; ++w) woff += wcnt[w];
        stream(std::integral_constant<int, 1>{}, GA + woff, 0xffffffffu);
        even_split();
        __syncthreads();  // workgroup-scope release/acquire: the lists written above are visible to every wave
        band_radix_pass(GA, xoff, xcnt, GB, 16, hist, tmp);
        band_radix_pass(GB, xoff, xcnt, GA, 24, hist, tmp);
        write_out(GA);
    }
}

}  // namespace evrep
