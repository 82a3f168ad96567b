// write_probe.hip -- development aid: the scan reads at the HBM ceiling until it writes its records (tools/stream_probe.hip: 3.0 GB
// read in 0.55 ms alone, 0.86 ms with 0.94 GB of 8-B + 2-B record stores).  This probe keeps the scan's read side (wave workers over
// contiguous tile ranges, lane = 64 B, edge words) and varies HOW the ~131 records of a tile are written.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
typedef unsigned int u32; typedef unsigned long long u64;
typedef unsigned int v4u __attribute__((ext_vector_type(4)));

// WM 0 none | 1 u64 + u16 arrays, per tile | 2 one 16-B record, per tile | 3 packed 10-B stream as 16-B stores, per tile
//    4 = 1 every 4 tiles | 5 = 3 every 4 tiles | 6 = 1 non-temporal | 7 = 3 non-temporal | 8 = 3 every 8 tiles | 9 = 1 every 8 tiles
template <int WM, int EDGE>
__global__ __launch_bounds__(256) void k_probe(const uint8_t* __restrict__ src, u64 n_tiles, u32 n_workers, uint8_t* out, u64 seg_bytes, u32* sink) {
    const u32 lane = threadIdx.x & 63u;
    const u32 w = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (w >= n_workers) return;
    const u64 per = n_tiles / n_workers, rem = n_tiles % n_workers;
    const u64 t0 = (u64)w * per + (w < rem ? w : rem), t1 = t0 + per + (w < rem ? 1 : 0);
    u32 acc = 0; u64 run = 0; u32 pend = 0;
    uint8_t* const seg = out + (u64)w * seg_bytes;                  // u64 part first, u16 part behind it (WM 1, 4, 6, 9)
    u64* const ss = (u64*)seg; uint16_t* const sl = (uint16_t*)(seg + (seg_bytes / 10 * 8 & ~63ull));
    uint4 A[4], B[4]; u32 ea = 0, eb = 0, fa = 0, fb = 0;
    auto load = [&](u64 tile, uint4* V, u32& e0, u32& e1) {
        const uint8_t* p = src + tile * 4096;
#pragma unroll
        for (int q = 0; q < 4; ++q) V[q] = *(const uint4*)(p + lane * 64u + 16u * q);
        if (EDGE == 1) { e0 = tile ? *(const u32*)(p + lane * 64u - 4) : 0u; e1 = tile + 1 < n_tiles ? *(const u32*)(p + lane * 64u + 64) : 0u; }
        if (EDGE == 2) {                                            // only the two outer lanes load; the others would take their neighbour's word by DPP
            e0 = 0; e1 = 0;
            if (lane == 0 && tile) e0 = *(const u32*)(p - 4);
            if (lane == 63 && tile + 1 < n_tiles) e1 = *(const u32*)(p + 4096);
        }
    };
    auto flush = [&](u32 cnt) {
        if (WM == 1 || WM == 4 || WM == 9) for (u32 r = lane; r < cnt; r += 64) { ss[run + r] = (u64)acc + r; sl[run + r] = (uint16_t)(acc + r); }
        if (WM == 10) { uint32_t* s4 = (uint32_t*)seg; for (u32 r = lane; r < cnt; r += 64) { s4[run + r] = acc + r; sl[run + r] = (uint16_t)(acc + r); } }
        if (WM == 11) { uint32_t* s4 = (uint32_t*)seg; for (u32 r = lane; r < cnt; r += 64) s4[run + r] = acc + r; }
        if (WM == 6) for (u32 r = lane; r < cnt; r += 64) { __builtin_nontemporal_store((u64)acc + r, &ss[run + r]); __builtin_nontemporal_store((uint16_t)(acc + r), &sl[run + r]); }
        if (WM == 2) for (u32 r = lane; r < cnt; r += 64) ((uint4*)seg)[run + r] = make_uint4(acc, r, acc, r);
        if (WM == 3 || WM == 5 || WM == 8) { const u32 q16 = (cnt * 10 + 15) / 16; for (u32 r = lane; r < q16; r += 64) ((uint4*)seg)[run * 10 / 16 + r] = make_uint4(acc, r, acc, r); }
        if (WM == 7) { const u32 q16 = (cnt * 10 + 15) / 16; for (u32 r = lane; r < q16; r += 64) { v4u v = {acc, r, acc, r}; __builtin_nontemporal_store(v, (v4u*)seg + run * 10 / 16 + r); } }
        run += cnt;
    };
    auto use = [&](const uint4* V, u32 e0, u32 e1) {
#pragma unroll
        for (int q = 0; q < 4; ++q) acc += V[q].x ^ V[q].y ^ V[q].z ^ V[q].w;
        acc += e0 + e1;
        if (WM == 0) return;
        pend += 131;
        const u32 every = (WM == 4 || WM == 5) ? 4 * 131 : (WM == 8 || WM == 9) ? 8 * 131 : 131;
        if (pend >= every) { flush(pend); pend = 0; }
    };
    const u64 nk = t1 - t0;
    if (nk == 0) return;
    load(t0, A, ea, fa);
    u64 k = 0;
    for (; k + 2 < nk; k += 2) {
        load(t0 + k + 1, B, eb, fb); use(A, ea, fa);
        load(t0 + k + 2, A, ea, fa); use(B, eb, fb);
    }
    for (; k < nk; ++k) { if (k + 1 < nk) load(t0 + k + 1, B, eb, fb); use(A, ea, fa); for (int q = 0; q < 4; ++q) A[q] = B[q]; ea = eb; fa = fb; }
    if (pend) flush(pend);
    if (acc == 0x12345u) sink[0] = acc;
}

template <int WM, int EDGE = 1> static void run(const char* name, const uint8_t* src, u64 n, u32 workers, uint8_t* out, u64 seg_bytes, u32* sink) {
    const u64 n_tiles = n / 4096;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e9f;
    for (int it = 0; it < 4; ++it) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k_probe<WM, EDGE>), dim3((workers + 3) / 4), dim3(256), 0, 0, src, n_tiles, workers, out, seg_bytes, sink);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (it && ms < best) best = ms;
    }
    const double wr = WM == 0 ? 0.0 : (double)n_tiles * 131 * (WM == 2 ? 16 : WM == 10 ? 6 : WM == 11 ? 4 : 10);
    printf("%-62s %.4f ms  read %.0f GB/s  read+write %.0f GB/s  (as 10-B records: %.0f GB/s)\n", name, best, n / (best * 1e-3) / 1e9, (n + wr) / (best * 1e-3) / 1e9,
           (n + (WM ? (double)n_tiles * 1310 : 0.0)) / (best * 1e-3) / 1e9);
    fflush(stdout);
}

int main(int argc, char** argv) {
    const u64 n = argc > 1 ? strtoull(argv[1], nullptr, 10) : 3020000000ull;
    const u32 workers = argc > 2 ? (u32)atoi(argv[2]) : 7168u;
    uint8_t* src; uint8_t* out; u32* sink;
    (void)hipMalloc(&src, n + 8192); (void)hipMemset(src, 1, n + 8192);
    const u64 seg_bytes = ((n / 4096 / workers + 2) * 131 * 16 + 4096) & ~255ull;
    (void)hipMalloc(&out, (u64)workers * seg_bytes + 4096); (void)hipMalloc(&sink, 64);
    run<0>("0 no writes", src, n, workers, out, seg_bytes, sink);
    run<0, 0>("0 no writes, no edge words", src, n, workers, out, seg_bytes, sink);
    run<0, 2>("0 no writes, edge words by the two outer lanes only", src, n, workers, out, seg_bytes, sink);
    run<1, 0>("1 u64 + u16 arrays per tile, no edge words", src, n, workers, out, seg_bytes, sink);
    run<1, 2>("1 u64 + u16 arrays per tile, edge words by the outer lanes", src, n, workers, out, seg_bytes, sink);
    run<1>("1 u64 + u16 arrays, 131 records per tile (the scan today)", src, n, workers, out, seg_bytes, sink);
    run<10>("10 u32 + u16 arrays (6 B per record), per tile", src, n, workers, out, seg_bytes, sink);
    run<11>("11 one u32 per record (4 B), per tile", src, n, workers, out, seg_bytes, sink);
    run<2>("2 one 16-B record, per tile", src, n, workers, out, seg_bytes, sink);
    run<3>("3 packed 10-B stream as 16-B stores, per tile", src, n, workers, out, seg_bytes, sink);
    run<4>("4 u64 + u16 arrays, every 4 tiles", src, n, workers, out, seg_bytes, sink);
    run<5>("5 packed stream, every 4 tiles", src, n, workers, out, seg_bytes, sink);
    run<9>("9 u64 + u16 arrays, every 8 tiles", src, n, workers, out, seg_bytes, sink);
    run<8>("8 packed stream, every 8 tiles", src, n, workers, out, seg_bytes, sink);
    run<6>("6 u64 + u16 arrays, non-temporal stores", src, n, workers, out, seg_bytes, sink);
    run<7>("7 packed stream, non-temporal stores", src, n, workers, out, seg_bytes, sink);
    return 0;
}
