// stream_probe.hip -- development aid: what HBM bandwidth the scan's ACCESS TOPOLOGY can reach, without any of its arithmetic.
// Every wave is a worker over a contiguous range of 4-KiB tiles (as k_scan_seg); variants change how a tile is read and whether
// records are written.  Prints GB/s over n bytes read (+ bytes written).
//   hipcc --offload-arch=gfx950 -O3 -o tools/stream_probe.bin tools/stream_probe.hip && tools/stream_probe.bin [n_bytes]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
typedef unsigned int u32; typedef unsigned long long u64;

// MODE 0: lane owns 64 contiguous bytes (4 x 16 B at stride 64 across the wave), prefetch one tile ahead      [the scan's pattern]
// MODE 1: MODE 0 + the two edge words
// MODE 2: coalesced: load q reads lane*16 + 1024*q
// MODE 3: MODE 0, tiles interleaved over the workers (tile = w + k * n_workers) instead of contiguous ranges
// MODE 4: MODE 1 + record-like writes: ~0.032 records per byte, 8 B + 2 B each, 64 per store, into the worker's own segment
// MODE 5: MODE 2 + the same writes
template <int MODE>
__global__ __launch_bounds__(256) void k_probe(const uint8_t* __restrict__ src, u64 n_tiles, u32 n_workers, u64* starts, uint16_t* lens, u64 seg_cap, u32* sink) {
    const u32 lane = threadIdx.x & 63u;
    const u32 w = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (w >= n_workers) return;
    const u64 per = n_tiles / n_workers, rem = n_tiles % n_workers;
    const u64 t0 = (u64)w * per + (w < rem ? w : rem), t1 = t0 + per + (w < rem ? 1 : 0);
    u32 acc = 0; u64 run = 0;
    u64* const ss = starts + (u64)w * seg_cap; uint16_t* const sl = lens + (u64)w * seg_cap;
    auto tile_of = [&](u64 k) -> u64 { return MODE == 3 ? (u64)w + k * n_workers : t0 + k; };
    const u64 nk = MODE == 3 ? (n_tiles - w + n_workers - 1) / n_workers : t1 - t0;
    uint4 A[4], B[4]; u32 ea = 0, eb = 0, fa = 0, fb = 0;
    auto load = [&](u64 tile, uint4* V, u32& e0, u32& e1) {
        const uint8_t* p = src + tile * 4096;
#pragma unroll
        for (int q = 0; q < 4; ++q) V[q] = (MODE == 2 || MODE == 5) ? *(const uint4*)(p + lane * 16u + 1024u * q) : *(const uint4*)(p + lane * 64u + 16u * q);
        if (MODE == 1 || MODE == 4) { e0 = tile ? *(const u32*)(p + lane * 64u - 4) : 0u; e1 = tile + 1 < n_tiles ? *(const u32*)(p + lane * 64u + 64) : 0u; }
    };
    auto use = [&](const uint4* V, u32 e0, u32 e1) {
#pragma unroll
        for (int q = 0; q < 4; ++q) acc += V[q].x ^ V[q].y ^ V[q].z ^ V[q].w;
        acc += e0 + e1;
        if (MODE >= 4) {
            const u32 cnt = 131;                                    // records of a typical tile
            for (u32 r = lane; r < cnt; r += 64) { ss[run + r] = (u64)acc + r; sl[run + r] = (uint16_t)(acc + r); }
            run += cnt;
        }
    };
    if (nk == 0) return;
    load(tile_of(0), A, ea, fa);
    u64 k = 0;
    for (; k + 2 < nk; k += 2) {
        load(tile_of(k + 1), B, eb, fb); use(A, ea, fa);
        load(tile_of(k + 2), A, ea, fa); use(B, eb, fb);
    }
    for (; k < nk; ++k) { if (k + 1 < nk) load(tile_of(k + 1), B, eb, fb); use(A, ea, fa); for (int q = 0; q < 4; ++q) A[q] = B[q]; ea = eb; fa = fb; }
    if (acc == 0x12345u) sink[0] = acc;
}

template <int MODE> static void run(const char* name, const uint8_t* src, u64 n, u32 workers, u64* starts, uint16_t* lens, u64 seg_cap, u32* sink) {
    const u64 n_tiles = n / 4096;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e9f;
    for (int it = 0; it < 4; ++it) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k_probe<MODE>, dim3((workers + 3) / 4), dim3(256), 0, 0, src, n_tiles, workers, starts, lens, seg_cap, sink);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (it && ms < best) best = ms;
    }
    const double wr = MODE >= 4 ? (double)n_tiles * 131 * 10 : 0.0;
    printf("%-58s workers %5u: %.4f ms  read %.0f GB/s  read+write %.0f GB/s\n", name, workers, best, n / (best * 1e-3) / 1e9, (n + wr) / (best * 1e-3) / 1e9);
    fflush(stdout);
}

int main(int argc, char** argv) {
    const u64 n = argc > 1 ? strtoull(argv[1], nullptr, 10) : 3020000000ull;
    uint8_t* src; u64* starts; uint16_t* lens; u32* sink;
    (void)hipMalloc(&src, n + 8192); (void)hipMemset(src, 1, n + 8192);
    const u32 maxw = 8192; const u64 seg_cap = (n / 4096 / 1024 + 2) * 131 + 256;     // room for the fewest workers tried
    (void)hipMalloc(&starts, (u64)maxw * seg_cap * 8); (void)hipMalloc(&lens, (u64)maxw * seg_cap * 2); (void)hipMalloc(&sink, 64);
    for (u32 workers : {7168u, 4096u, 2048u, 1024u}) {
        const u64 sc = seg_cap * 1024 / workers;                    // (the same total footprint for every worker count)
        run<0>("0 lane-contiguous 64 B, worker ranges", src, n, workers, starts, lens, sc, sink);
        run<1>("1 ... + edge words", src, n, workers, starts, lens, sc, sink);
        run<2>("2 coalesced 16 B/lane, worker ranges", src, n, workers, starts, lens, sc, sink);
        run<3>("3 lane-contiguous, tiles interleaved over the workers", src, n, workers, starts, lens, sc, sink);
        run<4>("4 (1) + record writes (8 B + 2 B, 0.032 / byte)", src, n, workers, starts, lens, sc, sink);
        run<5>("5 (2) + record writes", src, n, workers, starts, lens, sc, sink);
    }
    return 0;
}
