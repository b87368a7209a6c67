// evrep_bin.hip -- the (y,x) binning pass: a stable two-level partition of every window's events
// by pixel id, written for gfx950 (wave64, LDS counters, ballot multisplit).
//
// Level 1 (across workgroups): events of a window are cut into chunks; each workgroup histograms
// its chunk by sensor row (k_row_hist), a per-window scan turns the (chunk,row) table into
// destinations (k_row_scan), and each workgroup re-walks its chunk placing events stably
// (k_row_scatter).  Level 2 (inside one workgroup per row): k_col_sort orders the row's events by
// column, stably, so every pixel's events end up contiguous and in time order.
//
// This replaces the `index = y*W + x` scatter every reference builder starts with
// (event_stack.py:123-125, operations.py:40, time_surface.py:67, tore.py:23-47): after it, each
// builder is a per-pixel segmented reduction with coalesced, write-once output.
#include "evrep_common.h"

namespace evrep {

__global__ void k_init_meta(WindowMeta *meta, int B) {
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    WindowMeta m;
    m.tmin = INT32_MAX; m.tmax = INT32_MIN;
    m.xmin = INT32_MAX; m.xmax = INT32_MIN;
    m.ymin = INT32_MAX; m.ymax = INT32_MIN;
    m.neg_flags = 0; m.oob_flags = 0; m.status = 0; m.n_valid = 0;
    for (int i = 0; i < 6; ++i) m.pad[i] = 0;
    meta[b] = m;
}

__device__ inline int wave_min(int v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v = min(v, __shfl_xor(v, d, 64));
    return v;
}
__device__ inline int wave_max(int v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v = max(v, __shfl_xor(v, d, 64));
    return v;
}
__device__ inline uint32_t wave_or(uint32_t v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v |= __shfl_xor(v, d, 64);
    return v;
}
__device__ inline int wave_sum(int v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}

// grid (nblk, B), 256 threads, dynamic LDS = H * 4 bytes.
__global__ __launch_bounds__(kThreads) void k_row_hist(const int4 *__restrict__ ev, const int64_t *__restrict__ off,
                                                      int H, int W, int chunk, int nblk,
                                                      uint32_t *__restrict__ table, WindowMeta *__restrict__ meta) {
    extern __shared__ uint32_t hist[];
    const int b = blockIdx.y, blk = blockIdx.x;
    const int64_t beg = off[b];
    const int64_t n = off[b + 1] - beg;
    const int64_t lo = (int64_t)blk * chunk;
    if (lo >= n) {
        if (blk == 0 && threadIdx.x == 0) atomicOr(&meta[b].status, EVREP_ST_EMPTY);
        return;
    }
    for (int i = threadIdx.x; i < H; i += kThreads) hist[i] = 0;
    __syncthreads();
    const int64_t hi = (lo + chunk < n) ? lo + chunk : n;
    const MdesWindows mw = mdes_windows(n);
    const int64_t HW = (int64_t)H * W;
    int tmin = INT32_MAX, tmax = INT32_MIN, xmin = INT32_MAX, xmax = INT32_MIN, ymin = INT32_MAX, ymax = INT32_MIN;
    uint32_t negf = 0, oobf = 0, st = 0;
    int nvalid = 0;
    for (int64_t r = lo + threadIdx.x; r < hi; r += kThreads) {
        const int4 e = ev[beg + r];
        const int64_t key = (int64_t)e.x + (int64_t)e.y * W;
        const uint32_t memb = mdes_membership(mw, (int32_t)r);
        if (e.w == -1) negf |= memb;
        if (key >= 0 && key < HW) {
            atomicAdd(&hist[(uint32_t)key / (uint32_t)W], 1u);
            ++nvalid;
        } else {
            st |= EVREP_ST_OOB;
            const int cls = e.w == 1 ? 1 : (e.w == -1 ? 2 : (e.w == 0 ? 3 : 0));
            oobf |= memb | (cls ? (memb << (7 * cls)) : 0u);
        }
        if (r > 0 && ev[beg + r - 1].z > e.z) st |= EVREP_ST_UNSORTED;
        tmin = min(tmin, e.z); tmax = max(tmax, e.z);
        xmin = min(xmin, e.x); xmax = max(xmax, e.x);
        ymin = min(ymin, e.y); ymax = max(ymax, e.y);
    }
    tmin = wave_min(tmin); tmax = wave_max(tmax);
    xmin = wave_min(xmin); xmax = wave_max(xmax);
    ymin = wave_min(ymin); ymax = wave_max(ymax);
    negf = wave_or(negf); oobf = wave_or(oobf); st = wave_or(st);
    nvalid = wave_sum(nvalid);
    if ((threadIdx.x & 63) == 0) {
        WindowMeta *m = meta + b;
        atomicMin(&m->tmin, tmin); atomicMax(&m->tmax, tmax);
        atomicMin(&m->xmin, xmin); atomicMax(&m->xmax, xmax);
        atomicMin(&m->ymin, ymin); atomicMax(&m->ymax, ymax);
        if (negf) atomicOr(&m->neg_flags, negf);
        if (oobf) atomicOr(&m->oob_flags, oobf);
        if (st) atomicOr(&m->status, st);
        if (nvalid) atomicAdd(&m->n_valid, nvalid);
    }
    __syncthreads();
    uint32_t *dst = table + ((size_t)b * nblk + blk) * H;
    for (int i = threadIdx.x; i < H; i += kThreads) dst[i] = hist[i];
}

// grid (B), 256 threads, dynamic LDS = (H + 8) * 4 bytes.
// table[b][blk][row] -> exclusive prefix over blk; row_off[b][row] = global start of the row.
__global__ __launch_bounds__(kThreads) void k_row_scan(const int64_t *__restrict__ off, int H, int chunk, int nblk,
                                                      uint32_t *__restrict__ table, uint32_t *__restrict__ row_off,
                                                      WindowMeta *__restrict__ meta) {
    extern __shared__ uint32_t rowtot[];
    uint32_t *tmp = rowtot + H;
    const int b = blockIdx.x;
    const int64_t beg = off[b];
    const int64_t n = off[b + 1] - beg;
    const int nb = (int)((n + chunk - 1) / chunk);
    for (int r = threadIdx.x; r < H; r += kThreads) {
        uint32_t run = 0;
        for (int blk = 0; blk < nb; ++blk) {
            const size_t idx = ((size_t)b * nblk + blk) * H + r;
            const uint32_t v = table[idx];
            table[idx] = run;
            run += v;
        }
        rowtot[r] = run;
    }
    __syncthreads();
    const int per = (H + kThreads - 1) / kThreads;
    const int r0 = threadIdx.x * per;
    uint32_t local = 0;
    for (int k = 0; k < per; ++k) if (r0 + k < H) local += rowtot[r0 + k];
    uint32_t total;
    uint32_t run = block_exclusive_scan(local, tmp, &total);
    for (int k = 0; k < per; ++k)
        if (r0 + k < H) { const uint32_t t = rowtot[r0 + k]; rowtot[r0 + k] = run; run += t; }
    __syncthreads();
    uint32_t *ro = row_off + (size_t)b * (H + 1);
    for (int r = threadIdx.x; r < H; r += kThreads) ro[r] = (uint32_t)beg + rowtot[r];
    if (threadIdx.x == 0) {
        ro[H] = (uint32_t)beg + total;
        if (n > 0 && meta[b].tmin == meta[b].tmax) atomicOr(&meta[b].status, EVREP_ST_FLAT_TIME);
    }
}

// grid (nblk, B), 256 threads, dynamic LDS = 4 * H * 4 bytes.  Stable placement by sensor row.
__global__ __launch_bounds__(kThreads) void k_row_scatter(const int4 *__restrict__ ev, const int64_t *__restrict__ off,
                                                         int H, int W, int chunk, int nblk,
                                                         const uint32_t *__restrict__ table,
                                                         const uint32_t *__restrict__ row_off, Rec *__restrict__ sorted1) {
    extern __shared__ uint32_t cnt[];  // [kWaves][H]
    const int b = blockIdx.y, blk = blockIdx.x;
    const int64_t beg = off[b];
    const int64_t n = off[b + 1] - beg;
    const int64_t lo = (int64_t)blk * chunk;
    if (lo >= n) return;
    const int64_t hi = (lo + chunk < n) ? lo + chunk : n;
    const int64_t nloc = hi - lo;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t wlo = lo + nloc * wave / kWaves, whi = lo + nloc * (wave + 1) / kWaves;
    const int64_t HW = (int64_t)H * W;
    for (int i = threadIdx.x; i < kWaves * H; i += kThreads) cnt[i] = 0;
    __syncthreads();
    uint32_t *mycnt = cnt + wave * H;
    for (int64_t r = wlo + lane; r < whi; r += kWave) {
        const int4 e = ev[beg + r];
        const int64_t key = (int64_t)e.x + (int64_t)e.y * W;
        if (key >= 0 && key < HW) atomicAdd(&mycnt[(uint32_t)key / (uint32_t)W], 1u);
    }
    __syncthreads();
    for (int row = threadIdx.x; row < H; row += kThreads) {
        uint32_t base = row_off[(size_t)b * (H + 1) + row] + table[((size_t)b * nblk + blk) * H + row];
#pragma unroll
        for (int w = 0; w < kWaves; ++w) { const uint32_t t = cnt[w * H + row]; cnt[w * H + row] = base; base += t; }
    }
    __syncthreads();
    const int nbits = bits_for(H);
    volatile uint32_t *vcnt = mycnt;
    for (int64_t c0 = wlo; c0 < whi; c0 += kWave) {
        const int64_t r = c0 + lane;
        bool valid = r < whi;
        int4 e = make_int4(0, 0, 0, 0);
        if (valid) e = ev[beg + r];
        const int64_t key = (int64_t)e.x + (int64_t)e.y * W;
        valid = valid && key >= 0 && key < HW;
        const uint32_t row = valid ? (uint32_t)key / (uint32_t)W : 0u;
        uint32_t rk; bool last;
        wave_match(row, nbits, valid, lane, rk, last);
        uint32_t pos = 0;
        if (valid) {
            pos = vcnt[row] + rk;
            sorted1[pos] = make_int4((int)key, (int)r, e.z, e.w);
        }
        __builtin_amdgcn_wave_barrier();
        if (valid && last) vcnt[row] = pos + 1;
        __builtin_amdgcn_wave_barrier();
    }
}

// grid (H, B), 256 threads, dynamic LDS = (4 * W + 8) * 4 bytes.  Stable placement by column
// inside one row: afterwards sorted2 is ordered by (window, pixel id, rank).  Also emits, per row,
// the record offsets of every kChunkPx-pixel column chunk (what one builder wavefront consumes).
__global__ __launch_bounds__(kThreads) void k_col_sort(const Rec *__restrict__ sorted1, const uint32_t *__restrict__ row_off,
                                                      int H, int W, int nchunk, Rec *__restrict__ sorted2,
                                                      uint32_t *__restrict__ chunk_off) {
    extern __shared__ uint32_t cnt[];  // [kWaves][W] + tmp[8]
    uint32_t *tmp = cnt + kWaves * W;
    const int b = blockIdx.y, row = blockIdx.x;
    const uint32_t rs = row_off[(size_t)b * (H + 1) + row], re = row_off[(size_t)b * (H + 1) + row + 1];
    const uint32_t n = re - rs;
    // chunk_off[b][row][c] = global index of the first record whose column is >= c * kChunkPx
    uint32_t *co = chunk_off + ((size_t)b * H + row) * (nchunk + 1);
    if (n == 0) {
        for (int c = threadIdx.x; c <= nchunk; c += kThreads) co[c] = rs;
        return;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t wlo = rs + (uint32_t)((uint64_t)n * wave / kWaves), whi = rs + (uint32_t)((uint64_t)n * (wave + 1) / kWaves);
    const int rowbase = row * W;
    for (int i = threadIdx.x; i < kWaves * W; i += kThreads) cnt[i] = 0;
    __syncthreads();
    uint32_t *mycnt = cnt + wave * W;
    for (uint32_t j = wlo + lane; j < whi; j += kWave) atomicAdd(&mycnt[sorted1[j].x - rowbase], 1u);
    __syncthreads();
    const int per = (W + kThreads - 1) / kThreads;
    const int c0 = threadIdx.x * per;
    uint32_t local = 0;
    for (int k = 0; k < per; ++k)
        if (c0 + k < W) {
#pragma unroll
            for (int w = 0; w < kWaves; ++w) local += cnt[w * W + c0 + k];
        }
    uint32_t total;
    uint32_t run = block_exclusive_scan(local, tmp, &total);
    for (int k = 0; k < per; ++k)
        if (c0 + k < W) {
#pragma unroll
            for (int w = 0; w < kWaves; ++w) { const uint32_t t = cnt[w * W + c0 + k]; cnt[w * W + c0 + k] = run; run += t; }
        }
    __syncthreads();
    for (int c = threadIdx.x; c <= nchunk; c += kThreads) co[c] = (c * kChunkPx < W) ? rs + cnt[c * kChunkPx] : re;
    __syncthreads();
    const int nbits = bits_for(W);
    volatile uint32_t *vcnt = mycnt;
    for (uint32_t j0 = wlo; j0 < whi; j0 += kWave) {
        const uint32_t j = j0 + lane;
        const bool valid = j < whi;
        Rec e = make_int4(0, 0, 0, 0);
        if (valid) e = sorted1[j];
        const uint32_t col = valid ? (uint32_t)(e.x - rowbase) : 0u;
        uint32_t rk; bool last;
        wave_match(col, nbits, valid, lane, rk, last);
        uint32_t pos = 0;
        if (valid) {
            pos = vcnt[col] + rk;
            sorted2[rs + pos] = e;
        }
        __builtin_amdgcn_wave_barrier();
        if (valid && last) vcnt[col] = pos + 1;
        __builtin_amdgcn_wave_barrier();
    }
}

}  // namespace evrep
