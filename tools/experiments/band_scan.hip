// Microbenchmark 6: cost of the "every band block scans its whole window" phase of a single-kernel
// binning pass.  grid (nbands, B): block (band, window) streams the window's N events (16 B each, L2
// resident after the first touch), keeps those whose row falls in its band (ballot + popcount only --
// the floor of the phase), and writes one count.  Compared with the three-kernel pass (66 us at B=32,
// N=50000, 640x480) this decides whether such a kernel can pay off.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int THREADS, int UNROLL>
__global__ __launch_bounds__(THREADS) void k_scan(const int4 *__restrict__ ev, int N, int H, int nbands, unsigned *__restrict__ out) {
    // XCD-aware: all bands of a window on one XCD (id % 8 == window % 8) so the window's 800 KB sits in ONE L2
    const int id = blockIdx.x, xcd = id & 7, s = id >> 3;
    const int b = (s / nbands) * 8 + xcd, band = s % nbands;
    const int rows = (H + nbands - 1) / nbands;
    const int lo = band * rows, hi = lo + rows;
    const int4 *e = ev + (size_t)b * N;
    constexpr int NW = THREADS / 64;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // wave w owns a contiguous slice of the window (time order inside the slice)
    const int per = ((N + NW - 1) / NW + 63) / 64 * 64;
    const int beg = wave * per, end = min(N, beg + per);
    unsigned cnt = 0;
    for (int j0 = beg; j0 < end; j0 += 64 * UNROLL) {
        int4 r[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const int j = j0 + u * 64 + lane;
            r[u] = make_int4(0, -1, 0, 0);
            if (j < end) r[u] = e[j];
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) cnt += __popcll(__ballot(r[u].y >= lo && r[u].y < hi));
    }
    if (lane == 0) atomicAdd(&out[b * nbands + band], cnt);
}

template <int THREADS, int UNROLL>
float run(const int4 *ev, int B, int N, int H, int nbands, unsigned *out) {
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    const int grid = 8 * ((B + 7) / 8) * nbands;
    k_scan<THREADS, UNROLL><<<grid, THREADS>>>(ev, N, H, nbands, out);
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
        CHECK(hipEventRecord(a));
        for (int i = 0; i < 10; ++i) k_scan<THREADS, UNROLL><<<grid, THREADS>>>(ev, N, H, nbands, out);
        CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
        float ms; CHECK(hipEventElapsedTime(&ms, a, b));
        if (ms / 10 < best) best = ms / 10;
    }
    return best * 1e3f;
}

int main() {
    const int B = 32, N = 50000, H = 480, W = 640;
    int4 *h = (int4 *)malloc((size_t)B * N * 16);
    unsigned s = 12345;
    for (size_t i = 0; i < (size_t)B * N; ++i) {
        s = s * 1664525u + 1013904223u; const int x = (s >> 8) % W;
        s = s * 1664525u + 1013904223u; const int y = (s >> 8) % H;
        h[i] = make_int4(x, y, (int)(i % N), (s >> 30) & 1 ? 1 : -1);
    }
    int4 *ev; unsigned *out;
    CHECK(hipMalloc(&ev, (size_t)B * N * 16)); CHECK(hipMalloc(&out, 4096 * 4));
    CHECK(hipMemcpy(ev, h, (size_t)B * N * 16, hipMemcpyHostToDevice));
    CHECK(hipMemset(out, 0, 4096 * 4));
    for (int nb : {4, 8, 16}) {
        printf("bands %2d: 256thr/u4 %6.1f us | 512thr/u4 %6.1f us | 512thr/u8 %6.1f us | 1024thr/u4 %6.1f us | 1024thr/u8 %6.1f us   (L2 reads %.0f MB)\n", nb,
               run<256, 4>(ev, B, N, H, nb, out), run<512, 4>(ev, B, N, H, nb, out), run<512, 8>(ev, B, N, H, nb, out),
               run<1024, 4>(ev, B, N, H, nb, out), run<1024, 8>(ev, B, N, H, nb, out), (double)nb * B * N * 16 / 1e6);
    }
    return 0;
}
