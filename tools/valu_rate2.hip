// valu_rate2.hip -- development aid, second table: v_cndmask forms, compare -> select pairs, lane-crossing ops, LDS and scalar
// ops of the scan's bookkeeping, each as SIMD cycles per listed group with 1..8 waves per SIMD (see valu_rate.hip).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned int u32; typedef unsigned long long u64;
constexpr int ITER = 2048, UN = 16;
#define KERNEL32(NAME, ASM)                                                                            \
__global__ __launch_bounds__(256) void NAME(u32* out, u32 s0, u32 s1) {                               \
    __shared__ u32 lds[4096];                                                                          \
    u32 a[UN]; u32 b = threadIdx.x ^ s1, c = threadIdx.x + s0; u32 la = (threadIdx.x * 8u) & 0x3ff8u;  \
    lds[threadIdx.x] = s0;                                                                             \
    _Pragma("unroll") for (int i = 0; i < UN; ++i) a[i] = threadIdx.x * 2654435761u + i * s0;          \
    for (int it = 0; it < ITER; ++it) {                                                                \
        _Pragma("unroll") for (int i = 0; i < UN; ++i) asm volatile(ASM : "+v"(a[i]) : "v"(b), "v"(c), "s"(s1), "v"(la) : "vcc", "scc", "s20", "s21", "s22", "s23", "s24", "memory"); \
    }                                                                                                  \
    u32 r = 0; _Pragma("unroll") for (int i = 0; i < UN; ++i) r ^= a[i];                               \
    if (r == 0x12345u) out[threadIdx.x] = r + lds[1];                                                  \
}
KERNEL32(k_cnd_vcc, "v_cndmask_b32 %0, %0, %1, vcc")
KERNEL32(k_cnd_vcc2, "v_cndmask_b32 %0, %1, %2, vcc")
KERNEL32(k_cnd_sg, "v_cndmask_b32_e64 %0, %0, %1, s[20:21]")
KERNEL32(k_cmp_cnd, "v_cmp_lt_u32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %2, vcc")
KERNEL32(k_cmp_cnd_sg, "v_cmp_lt_u32_e64 s[20:21], %0, %1\n v_cndmask_b32_e64 %0, %0, %2, s[20:21]")
KERNEL32(k_cmp_nop_cnd, "v_cmp_lt_u32 vcc, %0, %1\n v_add_u32 %0, %0, %2\n v_cndmask_b32 %0, %0, %2, vcc")
KERNEL32(k_min, "v_min_u32 %0, %0, %1")
KERNEL32(k_bitop3v, "v_bitop3_b32 %0, %0, %1, %2 bitop3:0xe8")
KERNEL32(k_bitop3_2v, "v_bitop3_b32 %0, %0, %1, %1 bitop3:0x6c")
KERNEL32(k_or, "v_or_b32 %0, %0, %1")
KERNEL32(k_and_s, "v_and_b32 %0, %3, %0")
KERNEL32(k_and_lit, "v_and_b32 %0, 0x80808080, %0")
KERNEL32(k_and_inl, "v_and_b32 %0, 15, %0")
KERNEL32(k_add_lit, "v_add_u32 %0, 0x12345, %0")
KERNEL32(k_lshl_add, "v_lshl_add_u32 %0, %0, 3, %1")
KERNEL32(k_ashr, "v_ashrrev_i32 %0, 31, %0")
KERNEL32(k_mul24, "v_mul_u32_u24 %0, %0, %1")
KERNEL32(k_mad24, "v_mad_u32_u24 %0, %0, %1, %2")
KERNEL32(k_subrev, "v_subrev_u32 %0, %0, %1")
KERNEL32(k_readfl, "v_readfirstlane_b32 s20, %0\n v_add_u32 %0, s20, %0")
KERNEL32(k_dpp_mov, "v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf")
KERNEL32(k_dpp_max, "v_max_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf")
KERNEL32(k_dsw, "ds_write_b32 %4, %0 offset:0")
KERNEL32(k_dsr, "ds_read_b32 %0, %4 offset:0\n s_waitcnt lgkmcnt(0)")
KERNEL32(k_bperm, "ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)")
KERNEL32(k_sbcnt, "v_add_u32 %0, %0, %1\n s_bcnt1_i32_b64 s20, s[22:23]\n s_ff1_i32_b64 s21, s[22:23]")
KERNEL32(k_branchy, "v_cmp_lt_u32 vcc, %0, %1\n s_and_saveexec_b64 s[22:23], vcc\n v_add_u32 %0, %0, %2\n s_or_b64 exec, exec, s[22:23]")
KERNEL32(k_fast_mix, "v_and_b32 %0, %0, %1\n v_xor_b32 %0, %0, %2\n v_add_u32 %0, %0, %1\n v_bitop3_b32 %0, %0, %1, %2 bitop3:0xe8")
KERNEL32(k_slowfast, "v_lerp_u8 %0, %0, %1, %2\n v_and_b32 %0, %0, %1\n v_dot4_u32_u8 %0, %0, %1, %2\n v_xor_b32 %0, %0, %2")
struct K { const char* name; void (*fn)(u32*, u32, u32); int nops; };
int main() {
    u32* out; (void)hipMalloc(&out, 4096);
    K ks[] = {{"v_cndmask vcc (dst=src0)", k_cnd_vcc, 1}, {"v_cndmask vcc (3 regs)", k_cnd_vcc2, 1}, {"v_cndmask_e64 sgpr pair", k_cnd_sg, 1},
              {"v_cmp vcc + v_cndmask", k_cmp_cnd, 2}, {"v_cmp sgpr + v_cndmask_e64", k_cmp_cnd_sg, 2}, {"v_cmp + v_add + v_cndmask", k_cmp_nop_cnd, 3},
              {"v_min_u32", k_min, 1}, {"v_bitop3 3 vgpr (0xe8)", k_bitop3v, 1}, {"v_bitop3 2 vgpr", k_bitop3_2v, 1}, {"v_or_b32", k_or, 1},
              {"v_and_b32 sgpr", k_and_s, 1}, {"v_and_b32 literal", k_and_lit, 1}, {"v_and_b32 inline const", k_and_inl, 1}, {"v_add_u32 literal", k_add_lit, 1},
              {"v_lshl_add_u32", k_lshl_add, 1}, {"v_ashrrev_i32", k_ashr, 1}, {"v_mul_u32_u24", k_mul24, 1}, {"v_mad_u32_u24", k_mad24, 1},
              {"v_subrev_u32", k_subrev, 1}, {"readfirstlane + v_add s", k_readfl, 2}, {"v_mov_b32_dpp wave_shr", k_dpp_mov, 1},
              {"v_max_u32_dpp row_shr", k_dpp_max, 1}, {"ds_write_b32", k_dsw, 1}, {"ds_read_b32 + wait", k_dsr, 1}, {"ds_bpermute + wait", k_bperm, 1},
              {"v_add + s_bcnt1 + s_ff1", k_sbcnt, 3}, {"cmp+saveexec+add+restore", k_branchy, 4},
              {"and,xor,add,bitop3 (4 fast)", k_fast_mix, 4}, {"lerp,and,dot4,xor (2+2)", k_slowfast, 4}};
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    int clk_khz = 0; (void)hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0);
    printf("clock attr %d kHz; SIMD cycles per GROUP (nops listed) = time * clk / (ITER*UN*waves_per_simd)\n", clk_khz);
    for (auto& k : ks) {
        printf("%-30s (%d)", k.name, k.nops);
        for (int wps : {1, 2, 4, 8}) {
            const int blocks = 256 * wps;
            hipLaunchKernelGGL(k.fn, dim3(blocks), dim3(256), 0, 0, out, 3u, 5u);
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(k.fn, dim3(blocks), dim3(256), 0, 0, out, 3u, 5u);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            const double cyc = ms * 1e-3 * clk_khz * 1e3 / ((double)ITER * UN * wps);
            printf("  wps%d: %6.2f", wps, cyc);
        }
        printf("\n"); fflush(stdout);
    }
    return 0;
}
