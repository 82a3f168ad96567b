// valu_rate.hip -- development aid: issue cost (cycles per wave-instruction on one SIMD) of the integer VALU ops
// the scan kernels are made of, with 1, 2 and 4 waves per SIMD.  Every op is an `asm volatile`, 16 independent
// chains per wave, so the count is exact.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_rate tools/valu_rate.hip && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned int u32; typedef unsigned long long u64;
constexpr int ITER = 4096, UN = 16;

#define KERNEL32(NAME, ASM)                                                                            \
__global__ __launch_bounds__(256) void NAME(u32* out, u32 s0, u32 s1) {                               \
    u32 a[UN]; u32 b = threadIdx.x ^ s1, c = threadIdx.x + s0;                                         \
    _Pragma("unroll") for (int i = 0; i < UN; ++i) a[i] = threadIdx.x * 2654435761u + i * s0;          \
    for (int it = 0; it < ITER; ++it) {                                                                \
        _Pragma("unroll") for (int i = 0; i < UN; ++i) asm volatile(ASM : "+v"(a[i]) : "v"(b), "v"(c), "s"(s1)); \
    }                                                                                                  \
    u32 r = 0; _Pragma("unroll") for (int i = 0; i < UN; ++i) r ^= a[i];                               \
    if (r == 0x12345u) out[threadIdx.x] = r;                                                           \
}
#define KERNEL64(NAME, ASM)                                                                            \
__global__ __launch_bounds__(256) void NAME(u32* out, u32 s0, u32 s1) {                               \
    u64 a[UN]; u64 b = threadIdx.x ^ s1; u32 c = threadIdx.x + s0;                                     \
    _Pragma("unroll") for (int i = 0; i < UN; ++i) a[i] = threadIdx.x * 2654435761ull + i * s0;        \
    for (int it = 0; it < ITER; ++it) {                                                                \
        _Pragma("unroll") for (int i = 0; i < UN; ++i) asm volatile(ASM : "+v"(a[i]) : "v"(b), "v"(c), "s"(s1)); \
    }                                                                                                  \
    u64 r = 0; _Pragma("unroll") for (int i = 0; i < UN; ++i) r ^= a[i];                               \
    if (r == 0x12345u) out[threadIdx.x] = (u32)r;                                                      \
}
KERNEL32(k_add, "v_add_u32 %0, %0, %1")
KERNEL32(k_xor, "v_xor_b32 %0, %0, %1")
KERNEL32(k_bfi, "v_bfi_b32 %0, %0, %1, %2")
KERNEL32(k_bitop3, "v_bitop3_b32 %0, %0, %1, %2 bitop3:0x6c")
KERNEL32(k_alignbyte, "v_alignbyte_b32 %0, %0, %1, 3")
KERNEL32(k_dot4, "v_dot4_u32_u8 %0, %1, %2, %0")
KERNEL32(k_mul, "v_mul_lo_u32 %0, %0, %1")
KERNEL32(k_ffbl, "v_ffbl_b32 %0, %0")
KERNEL32(k_ffbh, "v_ffbh_u32 %0, %0")
KERNEL32(k_bcnt, "v_bcnt_u32_b32 %0, %1, %0")
KERNEL32(k_lshlor, "v_lshl_or_b32 %0, %0, 1, %1")
KERNEL32(k_and_or, "v_and_or_b32 %0, %0, %1, %2")
KERNEL32(k_add3, "v_add3_u32 %0, %0, %1, %2")
KERNEL32(k_perm, "v_perm_b32 %0, %0, %1, %2")
KERNEL32(k_lshlrev, "v_lshlrev_b32 %0, %2, %0")
KERNEL32(k_dpp, "v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf")
KERNEL32(k_cndmask, "v_cndmask_b32 %0, %0, %1, vcc")
KERNEL32(k_cmp, "v_cmp_lt_u32 vcc, %0, %1")
KERNEL32(k_sad, "v_sad_u8 %0, %0, %1, %2")
KERNEL32(k_mov, "v_mov_b32 %0, %1")
KERNEL64(k_shl64, "v_lshlrev_b64 %0, %2, %0")
KERNEL64(k_add64, "v_lshl_add_u64 %0, %0, 0, %1")
KERNEL32(k_pkadd, "v_pk_add_u16 %0, %0, %1")
KERNEL32(k_lerp, "v_lerp_u8 %0, %0, %1, %2")
KERNEL32(k_and, "v_and_b32 %0, %0, %1")
KERNEL32(k_not, "v_not_b32 %0, %0")
KERNEL32(k_or3, "v_or3_b32 %0, %0, %1, %2")
KERNEL32(k_xnor, "v_xnor_b32 %0, %0, %1")
KERNEL32(k_lshlrev_i, "v_lshlrev_b32 %0, 5, %0")
KERNEL32(k_and_s, "v_and_b32 %0, %3, %0")
KERNEL32(k_and_lit, "v_and_b32 %0, 0x80808080, %0")
KERNEL32(k_bitop3_s, "v_bitop3_b32 %0, %0, %1, %3 bitop3:0x80")
KERNEL32(k_sub, "v_sub_u32 %0, %0, %1")
KERNEL32(k_max, "v_max_u32 %0, %0, %1")
KERNEL32(k_mbcnt, "v_mbcnt_lo_u32_b32 %0, %1, %0")
KERNEL32(k_readlane, "v_readlane_b32 s20, %0, 3\n v_mov_b32 %0, s20")
KERNEL32(k_cmp_s, "v_cmp_lt_u32 s[20:21], %0, %1")
KERNEL32(k_salu_mix, "v_add_u32 %0, %0, %1\n s_add_u32 s20, s20, 1\n s_and_b32 s21, s21, s20")
KERNEL32(k_salu_mix2, "v_add_u32 %0, %0, %1\n s_add_u32 s20, s20, 1\n s_and_b32 s21, s21, s20\n s_or_b64 s[22:23], s[22:23], s[20:21]\n s_bcnt1_i32_b64 s24, s[22:23]")
KERNEL32(k_mix23, "v_and_b32 %0, %0, %1\n v_lerp_u8 %0, %0, %1, %2")
KERNEL64(k_shr64, "v_lshrrev_b64 %0, %2, %0")
KERNEL64(k_mov64, "v_mov_b64 %0, %1")

struct K { const char* name; void (*fn)(u32*, u32, u32); };
int main() {
    u32* out; (void)hipMalloc(&out, 4096);
    K ks[] = {{"v_add_u32", k_add}, {"v_xor_b32", k_xor}, {"v_bfi_b32", k_bfi}, {"v_bitop3_b32", k_bitop3}, {"v_alignbyte_b32", k_alignbyte},
              {"v_dot4_u32_u8", k_dot4}, {"v_mul_lo_u32", k_mul}, {"v_ffbl_b32", k_ffbl}, {"v_ffbh_u32", k_ffbh}, {"v_bcnt_u32_b32", k_bcnt},
              {"v_lshl_or_b32", k_lshlor}, {"v_and_or_b32", k_and_or}, {"v_add3_u32", k_add3}, {"v_perm_b32", k_perm}, {"v_lshlrev_b32", k_lshlrev},
              {"v_add_u32_dpp row_shr", k_dpp}, {"v_cndmask_b32", k_cndmask}, {"v_cmp_lt_u32", k_cmp}, {"v_sad_u8", k_sad}, {"v_mov_b32", k_mov},
              {"v_lshlrev_b64", k_shl64}, {"v_lshl_add_u64", k_add64}, {"v_pk_add_u16", k_pkadd},
              {"v_lerp_u8", k_lerp}, {"v_and_b32", k_and}, {"v_not_b32", k_not}, {"v_or3_b32", k_or3}, {"v_xnor_b32", k_xnor}, {"v_lshlrev_b32 imm", k_lshlrev_i},
              {"v_bitop3 sgpr", k_bitop3_s}, {"v_sub_u32", k_sub}, {"v_max_u32", k_max}, {"v_mbcnt_lo", k_mbcnt}, {"readlane+mov (2 ops)", k_readlane},
              {"v_cmp -> sgpr pair", k_cmp_s}, {"v_add + 2 salu (per group)", k_salu_mix}, {"v_add + 4 salu (per group)", k_salu_mix2},
              {"v_and + v_lerp (2 ops)", k_mix23}, {"v_lshrrev_b64", k_shr64}, {"v_mov_b64", k_mov64}};
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    int clk_khz = 0; (void)hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0);
    printf("clock attr %d kHz; SIMD cycles per wave-instruction = time * clk / (ITER*UN*waves_per_simd)\n", clk_khz);
    for (auto& k : ks) {
        printf("%-24s", k.name);
        for (int wps : {1, 2, 4, 8}) {
            const int blocks = 256 * wps;                       // 256 threads = 1 wave per SIMD of a CU
            hipLaunchKernelGGL(k.fn, dim3(blocks), dim3(256), 0, 0, out, 3u, 5u);
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(k.fn, dim3(blocks), dim3(256), 0, 0, out, 3u, 5u);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            const double cyc = ms * 1e-3 * clk_khz * 1e3 / ((double)ITER * UN * wps);
            printf("  wps%d: %5.2f", wps, cyc);
        }
        printf("\n");
    }
    return 0;
}
