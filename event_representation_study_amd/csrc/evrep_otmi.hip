// evrep_otmi.hip -- the point clouds of the GWD harness `otmi(events, rep, height, width, rep_size)`
// (representations/representation_search/compute_otmi.py:96-211) built ON THE DEVICE (r03): the quadrant split of the
// events, the re-origin of quadrants 2-4, the float32 normalisation, the mask, the cut of the letterboxed
// representation, its two positional channels and the `sum |feat| > 0` compaction -- every list in the reference's own
// order (the score compares Ks[i, j] with Kt[i, j] index by index, so the compactions are STABLE).  Nothing is read back:
// the cloud sizes stay on the device and feed evrep_gwd_padded_l1_batch.
#include "evrep_common.h"

namespace evrep {

constexpr int kOtmiThreads = 1024;
constexpr int kOtmiWaves = kOtmiThreads / 64;

// quadrant of an event in the SENSOR frame (:97-132): 0 left-top, 1 right-top, 2 left-bottom, 3 right-bottom, -1 none.
// x <= width / 2 - 1 (true division) <=> 2 x + 2 <= width.
__device__ inline int otmi_quadrant(int x, int y, int width, int height) {
    const bool left = x >= 0 && 2 * (int64_t)x + 2 <= width, right = 2 * (int64_t)x + 2 > width && x <= width - 1;
    const bool top = y >= 0 && 2 * (int64_t)y + 2 <= height, bottom = 2 * (int64_t)y + 2 > height && y <= height - 1;
    if (left && top) return 0;
    if (right && top) return 1;
    if (left && bottom) return 2;
    if (right && bottom) return 3;
    return -1;
}

struct OtmiQuadStats {
    int count, minx, miny, pmin, pmax, first, last;
};

// A window's events / a quadrant's pixels are cut into slices, one workgroup each (r03: one workgroup per window and one per
// (item, quadrant) left 244 of 256 CUs idle -- 0.43 / 0.39 ms per launch, more than the GWD solves they feed).  A slice
// is a contiguous range and a thread owns a contiguous part of it, so every compaction stays STABLE; the slices meet
// through a small scratch area (evrep_otmi_scratch_bytes):
//   events:  [B] OtmiWindowPlan | [B][kOtmiEvSlices][4] OtmiQuadStats | [B][kOtmiEvSlices][3] uint32 counts
//   rep:     [items][3][kOtmiRepSlices] uint32 counts
constexpr int kOtmiEvSlices = 16;
constexpr int kOtmiRepSlices = 32;
struct OtmiWindowPlan {
    OtmiQuadStats qs[4];
    int quad_of[3];
    int pad;
};
__host__ __device__ inline size_t otmi_ev_scratch_bytes(int B) {
    return (size_t)B * (sizeof(OtmiWindowPlan) + (size_t)kOtmiEvSlices * 4 * sizeof(OtmiQuadStats) + (size_t)kOtmiEvSlices * 3 * sizeof(uint32_t));
}
__host__ __device__ inline size_t otmi_rep_scratch_bytes(int items) { return (size_t)items * 3 * kOtmiRepSlices * sizeof(uint32_t); }

struct OtmiEvScratch {
    OtmiWindowPlan *plan;
    OtmiQuadStats *slice_stats;   // [B][slices][4]
    uint32_t *slice_cnt;          // [B][slices][3]
};
__host__ __device__ inline OtmiEvScratch otmi_ev_scratch(void *scratch, int B) {
    OtmiEvScratch w;
    char *p = static_cast<char *>(scratch);
    w.plan = reinterpret_cast<OtmiWindowPlan *>(p); p += (size_t)B * sizeof(OtmiWindowPlan);
    w.slice_stats = reinterpret_cast<OtmiQuadStats *>(p); p += (size_t)B * kOtmiEvSlices * 4 * sizeof(OtmiQuadStats);
    w.slice_cnt = reinterpret_cast<uint32_t *>(p);
    return w;
}

// the slice [lo, hi) of thread `tid` of workgroup-slice `s` among `ns` slices of n items
__device__ inline void otmi_range(int n, int s, int ns, int tid, int &lo, int &hi) {
    const int per_slice = (n + ns - 1) / ns;
    const int s0 = min(s * per_slice, n), s1 = min(s0 + per_slice, n);
    const int per = (s1 - s0 + kOtmiThreads - 1) / kOtmiThreads;
    lo = min(s0 + tid * per, s1);
    hi = min(lo + per, s1);
}

// grid (kOtmiEvSlices, B), 1024 threads.  Per quadrant of the slice: count, minima of x and y (the re-origin, :140-147),
// p range, first / last member.
__global__ __launch_bounds__(kOtmiThreads) void k_otmi_ev_stats(const int4 *__restrict__ ev, const int64_t *__restrict__ off,
                                                               int height, int width, OtmiEvScratch w) {
    __shared__ OtmiQuadStats qs[4];
    const int s = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const int64_t beg = off[b];
    const int n = (int)(off[b + 1] - beg);
    const int4 *e = ev + beg;
    if (tid < 4) {
        qs[tid].count = 0; qs[tid].minx = INT32_MAX; qs[tid].miny = INT32_MAX; qs[tid].pmin = INT32_MAX; qs[tid].pmax = INT32_MIN;
        qs[tid].first = INT32_MAX; qs[tid].last = -1;
    }
    __syncthreads();
    int lo, hi;
    otmi_range(n, s, kOtmiEvSlices, tid, lo, hi);
    for (int q = 0; q < 4; ++q) {
        int c = 0, mx = INT32_MAX, my = INT32_MAX, pl = INT32_MAX, ph = INT32_MIN, f = INT32_MAX, l = -1;
        for (int i = lo; i < hi; ++i) {
            const int4 r = e[i];
            if (otmi_quadrant(r.x, r.y, width, height) != q) continue;
            ++c; mx = min(mx, r.x); my = min(my, r.y); pl = min(pl, r.w); ph = max(ph, r.w);
            if (f == INT32_MAX) f = i;
            l = i;
        }
        if (c) {
            atomicAdd(&qs[q].count, c); atomicMin(&qs[q].minx, mx); atomicMin(&qs[q].miny, my);
            atomicMin(&qs[q].pmin, pl); atomicMax(&qs[q].pmax, ph); atomicMin(&qs[q].first, f); atomicMax(&qs[q].last, l);
        }
    }
    __syncthreads();
    if (tid < 4) w.slice_stats[((size_t)b * kOtmiEvSlices + s) * 4 + tid] = qs[tid];
}

// grid (B), 64 threads.  The window's quadrant statistics from its slices; its three scored quadrants (the most populated
// one is skipped, first maximum, :134-135), slot k = the k-th scored quadrant in ascending order.
__global__ __launch_bounds__(64) void k_otmi_ev_plan(OtmiEvScratch w, int32_t *__restrict__ quad_out) {
    const int b = blockIdx.x, q = threadIdx.x;
    __shared__ OtmiQuadStats qs[4];
    if (q < 4) {
        OtmiQuadStats a;
        a.count = 0; a.minx = INT32_MAX; a.miny = INT32_MAX; a.pmin = INT32_MAX; a.pmax = INT32_MIN; a.first = INT32_MAX; a.last = -1;
        for (int s = 0; s < kOtmiEvSlices; ++s) {
            const OtmiQuadStats t = w.slice_stats[((size_t)b * kOtmiEvSlices + s) * 4 + q];
            a.count += t.count; a.minx = min(a.minx, t.minx); a.miny = min(a.miny, t.miny); a.pmin = min(a.pmin, t.pmin);
            a.pmax = max(a.pmax, t.pmax); a.first = min(a.first, t.first); a.last = max(a.last, t.last);
        }
        qs[q] = a;
        w.plan[b].qs[q] = a;
    }
    __syncthreads();
    if (q == 0) {
        int skip = 0;
        for (int i = 1; i < 4; ++i) if (qs[i].count > qs[skip].count) skip = i;   // list.index(max(...)): first maximum
        int k = 0;
        for (int i = 0; i < 4; ++i)
            if (i != skip) { w.plan[b].quad_of[k] = i; quad_out[b * 3 + k] = i; ++k; }
    }
}

// grid (kOtmiEvSlices, B), 1024 threads.  WRITE = false: the rows of the slice that survive the mask (:170-172), per scored
// slot -> slice_cnt.  WRITE = true: the rows themselves, behind those of the earlier slices:
//   Xs [B][3][cap][4] float64 (the float32 values of :164-169), n_out [B][3].
template <bool WRITE>
__global__ __launch_bounds__(kOtmiThreads) void k_otmi_ev_rows(const int4 *__restrict__ ev, const int64_t *__restrict__ off,
                                                              int height, int width, int64_t cap, OtmiEvScratch w,
                                                              double *__restrict__ Xs, int64_t *__restrict__ n_out) {
    __shared__ uint32_t scan_tmp[kOtmiWaves];
    const int s = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const int64_t beg = off[b];
    const int n = (int)(off[b + 1] - beg);
    const int4 *e = ev + beg;
    int lo, hi;
    otmi_range(n, s, kOtmiEvSlices, tid, lo, hi);
    const int hx = (width - 1) / 2, hy = (height - 1) / 2;   // (width - 1) // 2, (height - 1) // 2
    const OtmiWindowPlan &plan = w.plan[b];
    for (int k = 0; k < 3; ++k) {
        const int q = plan.quad_of[k];
        const OtmiQuadStats st = plan.qs[q];
        const int ox = q == 0 ? 0 : st.minx, oy = q == 0 ? 0 : st.miny;   // the first quadrant stays as it is
        uint32_t c = 0;
        for (int i = lo; i < hi; ++i) {
            const int4 r = e[i];
            if (otmi_quadrant(r.x, r.y, width, height) == q && r.x - ox < hx && r.y - oy < hy) ++c;
        }
        uint32_t total;
        uint32_t pos = block_exclusive_scan<kOtmiWaves>(c, scan_tmp, &total);
        if (!WRITE) {
            if (tid == 0) w.slice_cnt[((size_t)b * kOtmiEvSlices + s) * 3 + k] = total;
            continue;
        }
        uint32_t before = 0, all = 0;
        for (int t = 0; t < kOtmiEvSlices; ++t) {
            const uint32_t v = w.slice_cnt[((size_t)b * kOtmiEvSlices + t) * 3 + k];
            if (t < s) before += v;
            all += v;
        }
        if (s == 0 && tid == 0) n_out[b * 3 + k] = st.count > 0 ? (int64_t)all : 0;
        if (st.count == 0) continue;   // uniform: an empty quadrant (the reference raises on min() of nothing)
        pos += before;
        const int64_t t0 = e[st.first].z, t1 = e[st.last].z;
        const float fx = (float)hx, fy = (float)hy, ft = (float)(t1 - t0), fp = (float)((int64_t)st.pmax - (int64_t)st.pmin);
        double *dst = Xs + ((size_t)(b * 3 + k) * (size_t)cap) * 4;
        for (int i = lo; i < hi; ++i) {
            const int4 r = e[i];
            if (otmi_quadrant(r.x, r.y, width, height) != q) continue;
            const int x = r.x - ox, y = r.y - oy;
            if (!(x < hx && y < hy)) continue;
            if ((int64_t)pos < cap) {
                double *row = dst + (size_t)pos * 4;
                // integer tensor / python int -> float32 in torch (:164-169): the operands are rounded to float32, the
                // quotient is one float32 division
                row[0] = (double)((float)x / fx);
                row[1] = (double)((float)y / fy);
                row[2] = (double)((float)((int64_t)r.z - t0) / ft);
                row[3] = (double)((float)((int64_t)r.w - (int64_t)st.pmin) / fp);
            }
            ++pos;
        }
    }
}

// grid (kOtmiRepSlices, 3, NB), 1024 threads.  Item i = (representation r, window b = i % B), slot k -> the cut of the
// letterboxed representation for quadrant quad[b][k] (:150-155,177-179, rows / columns int(lo) .. int(hi) inclusive), the
// two positional channels row / (rows - 1), column / (columns - 1) (:181-198), rows with sum |feat| > 0 kept (:200-202):
//   Xt [NB][3][m_cap][C + 2] float64, m_out [NB][3].
// WRITE = false: the kept pixels of slice blockIdx.x -> cnt [NB][3][kOtmiRepSlices]; WRITE = true: the rows.
template <typename RepT, bool WRITE>
__global__ __launch_bounds__(kOtmiThreads) void k_otmi_rep(const RepT *__restrict__ rep, int B, int S, int C,
                                                          const int32_t *__restrict__ quad, int64_t m_cap, uint32_t *__restrict__ cnt,
                                                          double *__restrict__ Xt, int64_t *__restrict__ m_out) {
    __shared__ uint32_t scan_tmp[kOtmiWaves];
    const int sl = blockIdx.x, k = blockIdx.y, item = blockIdx.z, tid = threadIdx.x;
    const int q = quad[(item % B) * 3 + k];
    // half = rep_size / 2 - 1 (true division), cut with int(): for the right / bottom boxes the first index is
    // int(half), for the left / top ones the last index is int(half) (rep_size // 2 - 1 is the same number:
    // S even -> S / 2 - 1; S odd -> trunc(S / 2 - 0.5) = S / 2 - 1 in integers)
    const int ih = S / 2 - 1;
    const int x0 = (q & 1) ? ih : 0, x1 = (q & 1) ? S - 1 : ih;
    const int y0 = (q & 2) ? ih : 0, y1 = (q & 2) ? S - 1 : ih;
    const int nrows = y1 - y0 + 1, ncols = x1 - x0 + 1, npx = nrows * ncols;
    const RepT *img = rep + (size_t)item * S * S * C;
    int lo, hi;
    otmi_range(npx, sl, kOtmiRepSlices, tid, lo, hi);
    auto keep = [&](int px) -> bool {
        const int r = px / ncols, c = px - r * ncols;
        const RepT *f = img + ((size_t)(y0 + r) * S + (x0 + c)) * C;
        // np.abs(feat).sum(-1) > 0: a sum of non-negative float64 terms is positive iff one term is, and NaN iff one is
        bool any = false, nan = false;
        for (int ch = 0; ch < C; ++ch) { const double v = (double)f[ch]; any |= v != 0.0; nan |= v != v; }
        return any && !nan;
    };
    uint32_t c = 0;
    for (int px = lo; px < hi; ++px) c += keep(px) ? 1u : 0u;
    uint32_t total;
    uint32_t pos = block_exclusive_scan<kOtmiWaves>(c, scan_tmp, &total);
    uint32_t *mine = cnt + ((size_t)item * 3 + k) * kOtmiRepSlices;
    if (!WRITE) {
        if (tid == 0) mine[sl] = total;
        return;
    }
    uint32_t before = 0, all = 0;
    for (int t = 0; t < kOtmiRepSlices; ++t) {
        const uint32_t v = mine[t];
        if (t < sl) before += v;
        all += v;
    }
    if (sl == 0 && tid == 0) m_out[item * 3 + k] = (int64_t)all;
    pos += before;
    const int D = C + 2;
    double *dst = Xt + ((size_t)(item * 3 + k) * (size_t)m_cap) * D;
    for (int px = lo; px < hi; ++px) {
        if (!keep(px)) continue;
        if ((int64_t)pos < m_cap) {
            const int r = px / ncols, cc = px - r * ncols;
            const RepT *f = img + ((size_t)(y0 + r) * S + (x0 + cc)) * C;
            double *row = dst + (size_t)pos * D;
            for (int ch = 0; ch < C; ++ch) row[ch] = (double)f[ch];
            row[C] = (double)r / (double)(nrows - 1);
            row[C + 1] = (double)cc / (double)(ncols - 1);
        }
        ++pos;
    }
}

}  // namespace evrep
