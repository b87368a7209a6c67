// evrep_builders.hip -- the event->dense-tensor builders as per-pixel segmented reductions over
// the binned stream (evrep_bin.hip), gfx950.
//
// One workgroup (4 wave64) owns one sensor row of one window.  It finds every pixel's segment
// [start, end) in the row's column-sorted events, reduces it with one thread per pixel IN TIME
// ORDER (so float64 sums round exactly as the reference's sequential scatter does), stages the
// (pixels, C) tile in LDS in output layout, and streams it out with 16-byte-per-lane coalesced
// stores.  Every output element is written exactly once; the zero / background fill is fused.
// HBM traffic per window = 16 B per event (binned record) + sizeof(out) per output element.
#include "evrep_common.h"

namespace evrep {

// --------------------------------------------------------------------------------------------
// shared pieces
// --------------------------------------------------------------------------------------------
struct RowCtx {
    uint32_t rs;  // global index of the row's first record
    uint32_t n;   // records in the row
};

// seg_start / seg_end (W entries each, LDS): per column, the [start, end) range inside the row.
__device__ inline void build_segments(const Rec *__restrict__ sorted, RowCtx rc, int rowbase, int W,
                                      uint32_t *seg_start, uint32_t *seg_end) {
    for (int i = threadIdx.x; i < W; i += kThreads) { seg_start[i] = 0; seg_end[i] = 0; }
    __syncthreads();
    for (uint32_t j = threadIdx.x; j < rc.n; j += kThreads) {
        const int key = sorted[rc.rs + j].x;
        const int prev = j > 0 ? sorted[rc.rs + j - 1].x : -1;
        const int next = j + 1 < rc.n ? sorted[rc.rs + j + 1].x : -1;
        if (prev != key) seg_start[key - rowbase] = j;
        if (next != key) seg_end[key - rowbase] = j + 1;
    }
    __syncthreads();
}

// Stream `count` staged elements (LDS, output layout) to global memory, 16 B per lane when the
// destination is 16-byte aligned.  Contains no barrier.
template <typename OutT>
__device__ inline void copy_out(const OutT *stage, int count, OutT *__restrict__ dst) {
    constexpr int V = 16 / (int)sizeof(OutT);
    if ((reinterpret_cast<uintptr_t>(dst) & 15u) == 0) {
        const int nvec = count / V;
        const float4 *s4 = reinterpret_cast<const float4 *>(stage);
        float4 *d4 = reinterpret_cast<float4 *>(dst);
        for (int v = threadIdx.x; v < nvec; v += kThreads) d4[v] = s4[v];
        for (int e = nvec * V + threadIdx.x; e < count; e += kThreads) dst[e] = stage[e];
    } else {
        for (int e = threadIdx.x; e < count; e += kThreads) dst[e] = stage[e];
    }
}

// Fill `count` staged elements with a per-channel pattern pat[ch], element e -> ch = e % C.
template <typename OutT>
__device__ inline void fill_pattern(OutT *stage, int count, int C, const OutT *pat) {
    for (int e = threadIdx.x; e < count; e += kThreads) stage[e] = pat[e % C];
}

template <typename OutT>
__device__ inline void fill_zero(OutT *stage, int count) {
    constexpr int V = 16 / (int)sizeof(OutT);
    float4 *s4 = reinterpret_cast<float4 *>(stage);
    const int nvec = (count + V - 1) / V;  // stage is padded to a multiple of 16 bytes
    for (int v = threadIdx.x; v < nvec; v += kThreads) s4[v] = make_float4(0.f, 0.f, 0.f, 0.f);
}

__host__ __device__ inline size_t align16(size_t v) { return (v + 15) & ~(size_t)15; }

// --------------------------------------------------------------------------------------------
// A3/A4/A5: MixedDensityEventStack.stack + Operations
// (representation_search/mixed_density_event_stack.py:25-151, operations.py:15-89)
// --------------------------------------------------------------------------------------------
struct MdesParams {
    int32_t C;
    int32_t win[EVREP_MAX_CHANNELS], func[EVREP_MAX_CHANNELS], agg[EVREP_MAX_CHANNELS];
};

constexpr int kWantAny = 2;

// grid (H, B); dynamic LDS = 2*W*4 + align16(256*C*sizeof(OutT)).
template <typename OutT>
__global__ __launch_bounds__(kThreads) void k_mdes(const Rec *__restrict__ sorted, const uint32_t *__restrict__ row_off,
                                                  const int64_t *__restrict__ off, const WindowMeta *__restrict__ meta,
                                                  MdesParams P, int H, int W, double scale, OutT *__restrict__ out) {
    extern __shared__ __align__(16) unsigned char smem[];
    uint32_t *seg_start = reinterpret_cast<uint32_t *>(smem);
    uint32_t *seg_end = seg_start + W;
    OutT *stage = reinterpret_cast<OutT *>(smem + align16((size_t)2 * W * 4));

    const int b = blockIdx.y, row = blockIdx.x;
    const int C = P.C;
    const int64_t n_win = off[b + 1] - off[b];
    RowCtx rc;
    rc.rs = row_off[(size_t)b * (H + 1) + row];
    rc.n = row_off[(size_t)b * (H + 1) + row + 1] - rc.rs;
    const WindowMeta m = meta[b];
    const int32_t tmin = m.tmin;
    // t = t - t.min(); t_s = t / (t.max() - t.min())  (mixed_density_event_stack.py:33,112-114)
    const double interval = (double)((int64_t)m.tmax - (int64_t)m.tmin);
    const MdesWindows mw = mdes_windows(n_win);

    // per-channel uniform setup
    int lo[EVREP_MAX_CHANNELS], hi[EVREP_MAX_CHANNELS], want[EVREP_MAX_CHANNELS];
    bool active[EVREP_MAX_CHANNELS];
#pragma unroll
    for (int c = 0; c < EVREP_MAX_CHANNELS; ++c) {
        lo[c] = 0; hi[c] = 0; want[c] = kWantAny; active[c] = false;
        if (c < C) {
            const int w = P.win[c], f = P.func[c], a = P.agg[c];
            bool ok = w >= 0 && w <= 6 && f >= 0 && f <= 6 && a >= 0 && a <= 3 && n_win > 0;
            int l = 0, h = 0;
#pragma unroll
            for (int i = 0; i < 7; ++i) if (w == i) { l = mw.lo[i]; h = mw.hi[i]; }
            int wn = kWantAny, field = 0;
            if (f == EVREP_F_TIMESTAMP_POS || f == EVREP_F_COUNT_POS) { wn = 1; field = 1; }
            if (f == EVREP_F_TIMESTAMP_NEG || f == EVREP_F_COUNT_NEG) {
                // rows with p == -1; if the window has none, rows with p == 0 (operations.py:59-61,78-80)
                const bool has_neg = ok && ((m.neg_flags >> w) & 1u);
                wn = has_neg ? -1 : 0;
                field = has_neg ? 2 : 3;
            }
            // an out-of-range index inside the selected rows raises in torch_scatter -> zero channel
            if (ok && ((m.oob_flags >> (7 * field + w)) & 1u)) ok = false;
            lo[c] = l; hi[c] = h; want[c] = wn; active[c] = ok;
        }
    }

    build_segments(sorted, rc, row * W, W, seg_start, seg_end);

    OutT *out_row = out + ((size_t)b * H + row) * (size_t)W * C;
    for (int cb = 0; cb < W; cb += kThreads) {
        const int npix = min(kThreads, W - cb);
        fill_zero(stage, npix * C);
        __syncthreads();
        const int col = cb + threadIdx.x;
        if (col < W) {
            const uint32_t js = seg_start[col], je = seg_end[col];
            if (je > js) {
                double s[EVREP_MAX_CHANNELS], s2[EVREP_MAX_CHANNELS];
                int cnt[EVREP_MAX_CHANNELS];
#pragma unroll
                for (int c = 0; c < EVREP_MAX_CHANNELS; ++c) { s[c] = 0.0; s2[c] = 0.0; cnt[c] = 0; }
                for (uint32_t j = js; j < je; ++j) {
                    const Rec e = sorted[rc.rs + j];
                    const int rank = e.y, p = e.w;
                    const double tn = (double)((int64_t)e.z - (int64_t)tmin) / interval;
                    const double pv = (double)p;
#pragma unroll
                    for (int c = 0; c < EVREP_MAX_CHANNELS; ++c) {
                        if (c < C && active[c]) {
                            const bool hit = rank >= lo[c] && rank < hi[c] && (want[c] == kWantAny || p == want[c]);
                            const int f = P.func[c];
                            const double v = (f == EVREP_F_POLARITY) ? pv
                                           : ((f == EVREP_F_COUNT || f == EVREP_F_COUNT_POS || f == EVREP_F_COUNT_NEG) ? 1.0 : tn);
                            if (hit) {
                                if (P.agg[c] == EVREP_A_MAX) {
                                    if (cnt[c] == 0 || v > s[c]) s[c] = v;
                                } else {
                                    s[c] = s[c] + v;
                                    const double vv = v * v;
                                    s2[c] = s2[c] + vv;
                                }
                                ++cnt[c];
                            }
                        }
                    }
                }
                OutT *mine = stage + (size_t)threadIdx.x * C;
#pragma unroll
                for (int c = 0; c < EVREP_MAX_CHANNELS; ++c) {
                    if (c < C) {
                        double r = 0.0;
                        if (active[c]) {
                            const double d = (double)(cnt[c] < 1 ? 1 : cnt[c]);
                            const int a = P.agg[c];
                            if (a == EVREP_A_SUM) r = s[c];
                            else if (a == EVREP_A_MEAN) r = s[c] / d;
                            else if (a == EVREP_A_MAX) r = cnt[c] > 0 ? s[c] : 0.0;
                            else {
                                const double mean = s[c] / d, mean2 = s2[c] / d;
                                const double mm = mean * mean;
                                r = mean2 - mm;
                            }
                        }
                        mine[c] = (OutT)(r * scale);
                    }
                }
            }
        }
        __syncthreads();
        copy_out(stage, npix * C, out_row + (size_t)cb * C);
        __syncthreads();
    }
}

// --------------------------------------------------------------------------------------------
// A6: EventStack.pre_stack / post_stack (event_stack.py:15-131), last_timestamp = t[-1]
// --------------------------------------------------------------------------------------------
// grid (H, B); dynamic LDS = 2*W*4 + align16(256*S*4).
__global__ __launch_bounds__(kThreads) void k_event_stack(const Rec *__restrict__ sorted, const uint32_t *__restrict__ row_off,
                                                         const int64_t *__restrict__ off, int H, int W, int S, int premap,
                                                         float scale, float *__restrict__ out) {
    extern __shared__ __align__(16) unsigned char smem[];
    uint32_t *seg_start = reinterpret_cast<uint32_t *>(smem);
    uint32_t *seg_end = seg_start + W;
    float *stage = reinterpret_cast<float *>(smem + align16((size_t)2 * W * 4));
    const int b = blockIdx.y, row = blockIdx.x;
    const int64_t n_win = off[b + 1] - off[b];
    RowCtx rc;
    rc.rs = row_off[(size_t)b * (H + 1) + row];
    rc.n = row_off[(size_t)b * (H + 1) + row + 1] - rc.rs;
    // level k keeps events[off_k:], off_k = sum_{j=1..k} N // 2^j  (event_stack.py:70-82)
    int offk[EVREP_MAX_CHANNELS];
    {
        int cur = (int)n_win, o = 0;
#pragma unroll
        for (int k = 0; k < EVREP_MAX_CHANNELS; ++k) { offk[k] = o; cur /= 2; o += cur; }
    }
    build_segments(sorted, rc, row * W, W, seg_start, seg_end);
    float *out_row = out + ((size_t)b * H + row) * (size_t)W * S;
    for (int cb = 0; cb < W; cb += kThreads) {
        const int npix = min(kThreads, W - cb);
        fill_zero(stage, npix * S);
        __syncthreads();
        const int col = cb + threadIdx.x;
        if (col < W) {
            const uint32_t js = seg_start[col], je = seg_end[col];
            if (je > js) {
                const Rec e = sorted[rc.rs + je - 1];  // ndarray.put is last-write-wins (event_stack.py:125)
                int p = e.w;
                if (premap) p = (p + 1) >> 1;               // (p + 1) // 2   (gen1_transforms.py:34)
                const float v = (float)(int8_t)(2 * p - 1) * scale;  // 2*p - 1 as int8 (event_stack.py:18)
                float *mine = stage + (size_t)threadIdx.x * S;
#pragma unroll
                for (int k = 0; k < EVREP_MAX_CHANNELS; ++k)
                    if (k < S) mine[k] = (e.y >= offk[k]) ? v : 0.0f;
            }
        }
        __syncthreads();
        copy_out(stage, npix * S, out_row + (size_t)cb * S);
        __syncthreads();
    }
}

// --------------------------------------------------------------------------------------------
// A7: ToTimesurface (time_surface.py:25-74) driven as gen1_transforms.py:69-87
// --------------------------------------------------------------------------------------------
constexpr int kMaxSlices = 8;
struct TsCuts {
    int32_t idx[kMaxSlices];   // searchsorted(t_norm, s+1, 'left')
    int32_t tcut[kMaxSlices];  // t[idx[s]]
    int32_t live[kMaxSlices];  // 1 iff the sequential scan reaches this slice (strictly increasing idx)
    int32_t pad[8];
};
static_assert(sizeof(TsCuts) == 128, "TsCuts");

// grid (B), 64 threads.
__global__ void k_ts_cuts(const int4 *__restrict__ ev, const int64_t *__restrict__ off, int S, TsCuts *__restrict__ cuts) {
    const int b = blockIdx.x, s = threadIdx.x;
    __shared__ int sidx[kMaxSlices];
    const int64_t beg = off[b];
    const int64_t n = off[b + 1] - beg;
    if (s < S) {
        int idx = 0, tc = 0;
        if (n > 0) {
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
            tc = idx < n ? ev[beg + idx].z : 0;
        }
        sidx[s] = idx;
        cuts[b].idx[s] = idx;
        cuts[b].tcut[s] = tc;
    }
    __syncthreads();
    if (s == 0) {
        bool alive = n > 0;
        for (int k = 0; k < kMaxSlices; ++k) {
            // `if index == indices[pos]` fires once per event: a repeated idx is never reached
            if (k < S) {
                if (k > 0 && sidx[k] <= sidx[k - 1]) alive = false;
                if (sidx[k] >= n) alive = false;
                cuts[b].live[k] = alive ? 1 : 0;
            } else {
                cuts[b].live[k] = 0;
                cuts[b].idx[k] = 0;
                cuts[b].tcut[k] = 0;
            }
        }
    }
}

// grid (H, B); dynamic LDS = 2*W*4 + align16(256*2S*sizeof(OutT)) + 16*sizeof(OutT).
template <typename OutT>
__global__ __launch_bounds__(kThreads) void k_time_surface(const Rec *__restrict__ sorted, const uint32_t *__restrict__ row_off,
                                                          const TsCuts *__restrict__ cuts, int H, int W, int S, double tau,
                                                          int premap, double scale, OutT *__restrict__ out) {
    extern __shared__ __align__(16) unsigned char smem[];
    uint32_t *seg_start = reinterpret_cast<uint32_t *>(smem);
    uint32_t *seg_end = seg_start + W;
    const int C = 2 * S;
    OutT *stage = reinterpret_cast<OutT *>(smem + align16((size_t)2 * W * 4));
    OutT *pat = reinterpret_cast<OutT *>(smem + align16((size_t)2 * W * 4) + align16((size_t)kThreads * C * sizeof(OutT)));
    const int b = blockIdx.y, row = blockIdx.x;
    RowCtx rc;
    rc.rs = row_off[(size_t)b * (H + 1) + row];
    rc.n = row_off[(size_t)b * (H + 1) + row + 1] - rc.rs;
    const TsCuts cu = cuts[b];
    const double init = -(tau * 3.0 + 1.0);  // timestamp_memory -= tau*3 + 1 (time_surface.py:29)
    if (threadIdx.x < C) {
        const int s = threadIdx.x >> 1;
        double v = 0.0;
        // untouched pixels are not zero: exp((-(3 tau + 1) - t_i) / tau)
        if (cu.live[s]) { const double d = init - (double)cu.tcut[s]; v = exp(d / tau) * scale; }
        pat[threadIdx.x] = (OutT)v;
    }
    build_segments(sorted, rc, row * W, W, seg_start, seg_end);  // (barriers inside also publish pat)
    OutT *out_row = out + ((size_t)b * H + row) * (size_t)W * C;
    for (int cb = 0; cb < W; cb += kThreads) {
        const int npix = min(kThreads, W - cb);
        fill_pattern(stage, npix * C, C, pat);
        __syncthreads();
        const int col = cb + threadIdx.x;
        if (col < W) {
            const uint32_t js = seg_start[col], je = seg_end[col];
            if (je > js) {
                OutT *mine = stage + (size_t)threadIdx.x * C;
                double mem0 = init, mem1 = init;
                bool touched0 = false, touched1 = false;
                int s = 0;
                for (uint32_t j = js; j <= je; ++j) {
                    int rank = INT32_MAX, t = 0, p = 0;
                    if (j < je) { const Rec e = sorted[rc.rs + j]; rank = e.y; t = e.z; p = e.w; }
                    // slices cut strictly before this event see the memory as it stands
                    while (s < S && cu.idx[s] < rank) {
                        if (cu.live[s]) {
                            const double tc = (double)cu.tcut[s];
                            if (touched0) mine[2 * s] = (OutT)(exp((mem0 - tc) / tau) * scale);
                            if (touched1) mine[2 * s + 1] = (OutT)(exp((mem1 - tc) / tau) * scale);
                        }
                        ++s;
                    }
                    if (j < je) {
                        if (premap) p = (int)(int8_t)(int)((double)(p + 1) / 2.0);  // ((p+1)/2).astype(int8)
                        if (p & 1) { mem1 = (double)t; touched1 = true; } else { mem0 = (double)t; touched0 = true; }
                    }
                }
            }
        }
        __syncthreads();
        copy_out(stage, npix * C, out_row + (size_t)cb * C);
        __syncthreads();
    }
}

// --------------------------------------------------------------------------------------------
// A8: events2ToreFeature (tore.py:6-83), one sample time T = t[-1]
// --------------------------------------------------------------------------------------------
constexpr int kMaxToreK = 8;

// grid (H, B); dynamic LDS = 2*W*4 + align16(256*2k*4).
__global__ __launch_bounds__(kThreads) void k_tore(const int4 *__restrict__ ev, const Rec *__restrict__ sorted,
                                                  const uint32_t *__restrict__ row_off, const int64_t *__restrict__ off,
                                                  const WindowMeta *__restrict__ meta, int H, int W, int K, int frame_mode,
                                                  float scale, float *__restrict__ out) {
    extern __shared__ __align__(16) unsigned char smem[];
    uint32_t *seg_start = reinterpret_cast<uint32_t *>(smem);
    uint32_t *seg_end = seg_start + W;
    float *stage = reinterpret_cast<float *>(smem + align16((size_t)2 * W * 4));
    const int b = blockIdx.y, orow = blockIdx.x;  // orow = output row
    const int C = 2 * K;
    const int64_t beg = off[b];
    const int64_t n_win = off[b + 1] - beg;
    if (n_win <= 0) return;
    const WindowMeta m = meta[b];
    int x0 = 0, y0 = 0, Hf = H, Wf = W;
    if (frame_mode == 0 || frame_mode == 1) { x0 = m.xmin; y0 = m.ymin; }  // x - min(x) + 1, then [.., j - 1]
    if (frame_mode == 0) { Hf = m.ymax - m.ymin + 1; Wf = m.xmax - m.xmin + 1; }
    if (orow >= Hf) return;
    const int row = orow + y0;  // sensor row feeding this output row
    const int T = ev[beg + n_win - 1].z;  // sampleTimes = ts[-1]
    RowCtx rc; rc.rs = 0; rc.n = 0;
    if (row >= 0 && row < H) {
        rc.rs = row_off[(size_t)b * (H + 1) + row];
        rc.n = row_off[(size_t)b * (H + 1) + row + 1] - rc.rs;
    }
    build_segments(sorted, rc, row * W, W, seg_start, seg_end);
    // empty FIFO slot: inf -> clamp 5e8 -> log(5e8 + 1) - log(151)   (tore.py:69-79)
    const double log_min = log(151.0);
    const float bg = fmaxf((float)((double)logf(500e6f + 1.0f) - log_min), 0.0f) * scale;
    float *out_row = out + (size_t)b * H * W * C + (size_t)orow * Wf * C;
    for (int cb = 0; cb < Wf; cb += kThreads) {
        const int npix = min(kThreads, Wf - cb);
        for (int e = threadIdx.x; e < npix * C; e += kThreads) stage[e] = bg;
        __syncthreads();
        const int ocol = cb + threadIdx.x;
        const int col = ocol + x0;
        if (ocol < Wf && col >= 0 && col < W) {
            const uint32_t js = seg_start[col], je = seg_end[col];
            if (je > js) {
                int fp[kMaxToreK], fn[kMaxToreK];
                int np_ = 0, nn_ = 0;
#pragma unroll
                for (int k = 0; k < kMaxToreK; ++k) { fp[k] = 0; fn[k] = 0; }
                for (uint32_t j = js; j < je; ++j) {
                    const Rec e = sorted[rc.rs + j];
                    if (!(e.z < T)) continue;  // ts < currentSampleTime (tore.py:17): events at T are dropped
                    if (e.w > 0) {
#pragma unroll
                        for (int k = kMaxToreK - 1; k > 0; --k) fp[k] = fp[k - 1];
                        fp[0] = e.z; ++np_;
                    } else {
#pragma unroll
                        for (int k = kMaxToreK - 1; k > 0; --k) fn[k] = fn[k - 1];
                        fn[0] = e.z; ++nn_;
                    }
                }
                float *mine = stage + (size_t)threadIdx.x * C;
#pragma unroll
                for (int k = 0; k < kMaxToreK; ++k) {
                    if (k < K) {
                        if (k < np_) {
                            float v = (float)(double)((int64_t)T - (int64_t)fp[k]);
                            v = fminf(v, 500e6f);
                            const float r = (float)((double)logf(v + 1.0f) - log_min);
                            mine[k] = fmaxf(r, 0.0f) * scale;
                        }
                        if (k < nn_) {
                            float v = (float)(double)((int64_t)T - (int64_t)fn[k]);
                            v = fminf(v, 500e6f);
                            const float r = (float)((double)logf(v + 1.0f) - log_min);
                            mine[K + k] = fmaxf(r, 0.0f) * scale;
                        }
                    }
                }
            }
        }
        __syncthreads();
        copy_out(stage, npix * C, out_row + (size_t)cb * C);
        __syncthreads();
    }
}

// --------------------------------------------------------------------------------------------
// A2: compute_repr (representation_search/gromov_wasserstein.py:72-82), t normalised as :96.
// mode 1: tonic.transforms.ToVoxelGrid as consumed at gen1_transforms.py:22-25 (parity unpinned).
// --------------------------------------------------------------------------------------------
// grid (H, B); dynamic LDS = 2*W*4 + align16(256*bins*8).
__global__ __launch_bounds__(kThreads) void k_voxel(const int4 *__restrict__ ev, const Rec *__restrict__ sorted,
                                                   const uint32_t *__restrict__ row_off, const int64_t *__restrict__ off,
                                                   int H, int W, int bins, int mode, double scale, double *__restrict__ out) {
    extern __shared__ __align__(16) unsigned char smem[];
    uint32_t *seg_start = reinterpret_cast<uint32_t *>(smem);
    uint32_t *seg_end = seg_start + W;
    double *stage = reinterpret_cast<double *>(smem + align16((size_t)2 * W * 4));
    const int b = blockIdx.y, row = blockIdx.x;
    const int64_t beg = off[b];
    const int64_t n_win = off[b + 1] - beg;
    RowCtx rc;
    rc.rs = row_off[(size_t)b * (H + 1) + row];
    rc.n = row_off[(size_t)b * (H + 1) + row + 1] - rc.rs;
    double t0 = 0.0, den = 0.0;
    if (n_win > 0) { t0 = (double)ev[beg].z; den = (double)ev[beg + n_win - 1].z - t0; }
    build_segments(sorted, rc, row * W, W, seg_start, seg_end);
    double *out_row = out + ((size_t)b * H + row) * (size_t)W * bins;
    for (int cb = 0; cb < W; cb += kThreads) {
        const int npix = min(kThreads, W - cb);
        fill_zero(stage, npix * bins);
        __syncthreads();
        const int col = cb + threadIdx.x;
        if (col < W) {
            const uint32_t js = seg_start[col], je = seg_end[col];
            double *mine = stage + (size_t)threadIdx.x * bins;
            // two np.add.at passes: lower bin for every event, then upper bin for every event
            for (int pass = 0; pass < 2 && je > js; ++pass) {
                for (uint32_t j = js; j < je; ++j) {
                    const Rec e = sorted[rc.rs + j];
                    double p = (double)e.w;
                    double bpos;
                    if (mode == 0) {
                        const double tn = ((double)e.z - t0) / den;
                        bpos = (double)(bins - 1) * tn;
                    } else {
                        const double num = (double)bins * ((double)e.z - t0);
                        bpos = num / den;
                        if (e.w == 0) p = -1.0;
                    }
                    const int bi = (int)bpos;
                    const int blim = bi + pass;
                    if (blim < bins && blim >= 0) {
                        double w;
                        if (mode == 0) w = 1.0 - fabs((double)blim - bpos);
                        else { const double dts = bpos - (double)bi; w = pass ? dts : 1.0 - dts; }
                        const double wp = mode == 0 ? w * p : p * w;
                        mine[blim] = mine[blim] + wp;
                    }
                }
            }
            if (scale != 1.0 && je > js)
                for (int k = 0; k < bins; ++k) mine[k] = mine[k] * scale;
        }
        __syncthreads();
        copy_out(stage, npix * bins, out_row + (size_t)cb * bins);
        __syncthreads();
    }
}

// explicit instantiations used by the C API
template __global__ void k_mdes<double>(const Rec *, const uint32_t *, const int64_t *, const WindowMeta *, MdesParams, int, int, double, double *);
template __global__ void k_mdes<float>(const Rec *, const uint32_t *, const int64_t *, const WindowMeta *, MdesParams, int, int, double, float *);
template __global__ void k_time_surface<double>(const Rec *, const uint32_t *, const TsCuts *, int, int, int, double, int, double, double *);
template __global__ void k_time_surface<float>(const Rec *, const uint32_t *, const TsCuts *, int, int, int, double, int, double, float *);

}  // namespace evrep
