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

// grid (B), 1024 threads.  Window b -> its three scored quadrants (the most populated one is skipped, first maximum,
// :134-135), slot k = the k-th scored quadrant in ascending order:
//   Xs [B][3][cap][4] float64 (the float32 values of :164-169), n_out [B][3], quad_out [B][3].
// A thread owns a contiguous range of the window's events; two counting passes and a block scan keep the order.
__global__ __launch_bounds__(kOtmiThreads) void k_otmi_events(const int4 *__restrict__ ev, const int64_t *__restrict__ off,
                                                             int height, int width, int64_t cap, double *__restrict__ Xs,
                                                             int64_t *__restrict__ n_out, int32_t *__restrict__ quad_out) {
    __shared__ OtmiQuadStats qs[4];
    __shared__ uint32_t scan_tmp[kOtmiWaves];
    __shared__ int quad_of[3];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int64_t beg = off[b];
    const int n = (int)(off[b + 1] - beg);
    const int4 *e = ev + beg;
    if (tid < 4) {
        qs[tid].count = 0; qs[tid].minx = INT32_MAX; qs[tid].miny = INT32_MAX; qs[tid].pmin = INT32_MAX; qs[tid].pmax = INT32_MIN;
        qs[tid].first = INT32_MAX; qs[tid].last = -1;
    }
    __syncthreads();
    const int per = (n + kOtmiThreads - 1) / kOtmiThreads;
    const int lo = min(tid * per, n), hi = min(lo + per, n);
    // pass 1: per quadrant count, minima of x and y (the re-origin, :140-147), p range, first / last member
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
    if (tid == 0) {
        int skip = 0;
        for (int q = 1; q < 4; ++q) if (qs[q].count > qs[skip].count) skip = q;   // list.index(max(...)): first maximum
        int k = 0;
        for (int q = 0; q < 4; ++q)
            if (q != skip) { quad_of[k] = q; quad_out[b * 3 + k] = q; ++k; }
    }
    __syncthreads();
    const int hx = (width - 1) / 2, hy = (height - 1) / 2;   // (width - 1) // 2, (height - 1) // 2
    for (int k = 0; k < 3; ++k) {
        const int q = quad_of[k];
        const OtmiQuadStats st = qs[q];
        const int ox = q == 0 ? 0 : st.minx, oy = q == 0 ? 0 : st.miny;   // the first quadrant stays as it is
        // pass 2: the rows that survive the mask (:170-172), counted per thread, scanned over the block
        uint32_t c = 0;
        for (int i = lo; i < hi; ++i) {
            const int4 r = e[i];
            if (otmi_quadrant(r.x, r.y, width, height) == q && r.x - ox < hx && r.y - oy < hy) ++c;
        }
        uint32_t total;
        uint32_t pos = block_exclusive_scan<kOtmiWaves>(c, scan_tmp, &total);
        if (tid == 0) n_out[b * 3 + k] = st.count > 0 ? (int64_t)total : 0;
        if (st.count == 0) continue;   // uniform: an empty quadrant (the reference raises on min() of nothing)
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

// grid (3, NB), 1024 threads.  Item i = (representation r, window b = i % B), slot k -> the cut of the letterboxed
// representation for quadrant quad[b][k] (:150-155,177-179, rows / columns int(lo) .. int(hi) inclusive), the two
// positional channels row / (rows - 1), column / (columns - 1) (:181-198), rows with sum |feat| > 0 kept (:200-202):
//   Xt [NB][3][m_cap][C + 2] float64, m_out [NB][3].
template <typename RepT>
__global__ __launch_bounds__(kOtmiThreads) void k_otmi_rep(const RepT *__restrict__ rep, int B, int S, int C,
                                                          const int32_t *__restrict__ quad, int64_t m_cap,
                                                          double *__restrict__ Xt, int64_t *__restrict__ m_out) {
    __shared__ uint32_t scan_tmp[kOtmiWaves];
    const int k = blockIdx.x, item = blockIdx.y, tid = threadIdx.x;
    const int q = quad[(item % B) * 3 + k];
    // half = rep_size / 2 - 1 (true division), cut with int(): for the right / bottom boxes the first index is
    // int(half), for the left / top ones the last index is int(half) (rep_size // 2 - 1 is the same number)
    const int ih = S / 2 - 1 + ((S & 1) ? 0 : 0);   // int(S / 2 - 1): S even -> S/2 - 1; S odd -> trunc(k - 0.5) = k - 1 = S/2 - 1
    const int x0 = (q & 1) ? ih : 0, x1 = (q & 1) ? S - 1 : ih;
    const int y0 = (q & 2) ? ih : 0, y1 = (q & 2) ? S - 1 : ih;
    const int nrows = y1 - y0 + 1, ncols = x1 - x0 + 1, npx = nrows * ncols;
    const RepT *img = rep + (size_t)item * S * S * C;
    const int per = (npx + kOtmiThreads - 1) / kOtmiThreads;
    const int lo = min(tid * per, npx), hi = min(lo + per, npx);
    auto keep = [&](int px) -> bool {
        const int r = px / ncols, c = px - r * ncols;
        const RepT *f = img + ((size_t)(y0 + r) * S + (x0 + c)) * C;
        // np.abs(feat).sum(-1) > 0: a sum of non-negative float64 terms is positive iff one term is, and NaN iff one is
        bool any = false, nan = false;
        for (int ch = 0; ch < C; ++ch) { const double v = (double)f[ch]; any |= v != 0.0; nan |= v != v; }
        return any && !nan;
    };
    uint32_t cnt = 0;
    for (int px = lo; px < hi; ++px) cnt += keep(px) ? 1u : 0u;
    uint32_t total;
    uint32_t pos = block_exclusive_scan<kOtmiWaves>(cnt, scan_tmp, &total);
    if (tid == 0) m_out[item * 3 + k] = (int64_t)total;
    const int D = C + 2;
    double *dst = Xt + ((size_t)(item * 3 + k) * (size_t)m_cap) * D;
    for (int px = lo; px < hi; ++px) {
        if (!keep(px)) continue;
        if ((int64_t)pos < m_cap) {
            const int r = px / ncols, c = px - r * ncols;
            const RepT *f = img + ((size_t)(y0 + r) * S + (x0 + c)) * C;
            double *row = dst + (size_t)pos * D;
            for (int ch = 0; ch < C; ++ch) row[ch] = (double)f[ch];
            row[C] = (double)r / (double)(nrows - 1);
            row[C + 1] = (double)c / (double)(ncols - 1);
        }
        ++pos;
    }
}

}  // namespace evrep
