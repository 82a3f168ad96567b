// gather_probe: what random short-row gathers cost on MI355X -- the floor under k_consensus' left-context gather.
// N rows of 31 bytes at random (unaligned) offsets of a large buffer, fetched (a) one byte per lane, 32 lanes per row (as the
// kernel does), (b) one unaligned dword per lane, 8 lanes per row, (c) 16 bytes per lane, 2 lanes per row (aligned down).
// Every wave keeps B loads in flight.  Build: hipcc --offload-arch=gfx950 -O3 -o /tmp/gather_probe tools/gather_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef unsigned long long u64; typedef unsigned int u32;
__device__ __forceinline__ u64 mix(u64 x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x; }
template <int MODE, int B>
__global__ __launch_bounds__(256) void k_gather(const uint8_t* buf, u64 bytes, u64 n_rows, u32* sink) {
    const u32 lane = threadIdx.x & 63u;
    const u64 wave = (u64)blockIdx.x * 4 + (threadIdx.x >> 6), n_waves = (u64)gridDim.x * 4;
    constexpr u32 LPR = MODE == 0 ? 32u : MODE == 1 ? 8u : 2u;      // lanes per row
    constexpr u32 RPI = 64u / LPR;                                  // rows per instruction
    u32 acc = 0;
    for (u64 r0 = wave * RPI * B; r0 < n_rows; r0 += n_waves * RPI * B) {
        u32 v[B];
#pragma unroll
        for (int b = 0; b < B; ++b) {
            const u64 row = r0 + (u64)b * RPI + lane / LPR;
            const u64 off = mix(row) % (bytes - 64);
            if (MODE == 0) v[b] = buf[off + (lane % LPR)];
            else if (MODE == 1) v[b] = *(const u32*)(buf + off + 4 * (lane % LPR));
            else { const uint4 q = *(const uint4*)(buf + (off & ~15ull) + 16 * (lane % LPR)); v[b] = q.x ^ q.y ^ q.z ^ q.w; }
        }
#pragma unroll
        for (int b = 0; b < B; ++b) acc += v[b];
    }
    if (acc == 0x12345678u) sink[0] = acc;
}
template <int MODE, int B> static void run(const uint8_t* buf, u64 bytes, u64 n_rows, u32* sink, int blocks, const char* name) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int it = 0; it < 6; ++it) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((k_gather<MODE, B>), dim3(blocks), dim3(256), 0, 0, buf, bytes, n_rows, sink);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (it && ms < best) best = ms;
    }
    printf("%-34s B=%2d blocks=%5d: %.3f ms for %.1e rows = %.1f rows/us\n", name, B, blocks, best, (double)n_rows, n_rows / (best * 1e3)); fflush(stdout);
}
int main(int argc, char** argv) {
    const u64 bytes = argc > 1 ? strtoull(argv[1], 0, 10) : 3000000000ull, n_rows = argc > 2 ? strtoull(argv[2], 0, 10) : 10000000ull;
    uint8_t* buf; u32* sink; CK(hipMalloc(&buf, bytes)); CK(hipMalloc(&sink, 4)); CK(hipMemset(buf, 1, bytes));
    for (int blocks : {2048, 8192}) {
        run<0, 4>(buf, bytes, n_rows, sink, blocks, "byte loads, 32 lanes per row");
        run<0, 16>(buf, bytes, n_rows, sink, blocks, "byte loads, 32 lanes per row");
        run<1, 4>(buf, bytes, n_rows, sink, blocks, "dword loads, 8 lanes per row");
        run<1, 16>(buf, bytes, n_rows, sink, blocks, "dword loads, 8 lanes per row");
        run<2, 4>(buf, bytes, n_rows, sink, blocks, "16-B loads, 2 lanes per row");
        run<2, 16>(buf, bytes, n_rows, sink, blocks, "16-B loads, 2 lanes per row");
    }
    return 0;
}
