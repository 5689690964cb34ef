// valu_probe.hip -- issue cost of the VALU instructions the per-frame kernels are made of, on this MI355X (round 4).
//   hipcc --offload-arch=gfx950 -O3 tools/valu_probe.hip -o ab/valu_probe && ab/valu_probe
// Every kernel of the frame is VALU-busy 0.6-0.8 of its time (profiles/r03_report.md), so which instructions are full rate (a wave64
// instruction in 2 cycles on a SIMD-32) and which are not decides what a rewrite can buy.  Per instruction: a loop of 16 independent
// copies (no dependent chain shorter than 16 instructions), W waves per SIMD (1, 2, 4, 8), time from s_memtime inside the kernel
// (shader cycles) of the slowest wave of SIMD-filling workgroups.  Printed: cycles per wave-instruction per SIMD = elapsed / (instructions per wave x W).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <cstring>

#define CK(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int ITERS = 512, UNROLL = 16;

#define R16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

// BODY(i) is one asm statement on r[i] (read-write) with extra operands a, b (read-only)
#define PROBE(NAME, ASM)                                                                                                   \
    __global__ void __launch_bounds__(256) NAME(unsigned long long *out, unsigned seed)                                    \
    {                                                                                                                      \
        unsigned r[UNROLL];                                                                                                \
        for (int i = 0; i < UNROLL; ++i) r[i] = seed * (threadIdx.x + 1) + i * 0x01010101u;                                 \
        unsigned a = seed + threadIdx.x, b = seed ^ 0x3f800000u;                                                           \
        unsigned long long pa = ((unsigned long long)b << 32) | a;                                                         \
        (void)pa;                                                                                                          \
        const unsigned long long t0 = __builtin_readcyclecounter();                                                        \
        for (int it = 0; it < ITERS; ++it) {                                                                               \
            _Pragma("unroll") for (int i = 0; i < UNROLL; ++i) asm volatile(ASM : "+v"(r[i]) : "v"(a), "v"(b));            \
        }                                                                                                                  \
        const unsigned long long t1 = __builtin_readcyclecounter();                                                        \
        unsigned acc = 0;                                                                                                  \
        for (int i = 0; i < UNROLL; ++i) acc ^= r[i];                                                                      \
        if (acc == 0x12345u) out[1] = acc;                                                                                 \
        if ((threadIdx.x & 63) == 0) atomicMax(out, t1 - t0);                                                              \
    }

// 64-bit register pair variants (packed fp32, mad_u64)
#define PROBE64(NAME, ASM)                                                                                                 \
    __global__ void __launch_bounds__(256) NAME(unsigned long long *out, unsigned seed)                                    \
    {                                                                                                                      \
        unsigned long long r[UNROLL];                                                                                      \
        for (int i = 0; i < UNROLL; ++i) r[i] = (unsigned long long)(seed * (threadIdx.x + 1) + i) * 0x100000001ull;       \
        unsigned a = seed + threadIdx.x, b = seed ^ 0x3f800000u;                                                           \
        unsigned long long pa = ((unsigned long long)b << 32) | a;                                                         \
        const unsigned long long t0 = __builtin_readcyclecounter();                                                        \
        for (int it = 0; it < ITERS; ++it) {                                                                               \
            _Pragma("unroll") for (int i = 0; i < UNROLL; ++i) asm volatile(ASM : "+v"(r[i]) : "v"(a), "v"(b), "v"(pa));   \
        }                                                                                                                  \
        const unsigned long long t1 = __builtin_readcyclecounter();                                                        \
        unsigned long long acc = 0;                                                                                        \
        for (int i = 0; i < UNROLL; ++i) acc ^= r[i];                                                                      \
        if (acc == 0x12345u) out[1] = acc;                                                                                 \
        if ((threadIdx.x & 63) == 0) atomicMax(out, t1 - t0);                                                              \
    }

PROBE(p_fma_f32, "v_fma_f32 %0, %0, %1, %2")
PROBE(p_fmac_f32, "v_fmac_f32 %0, %1, %2")
PROBE(p_mul_f32, "v_mul_f32 %0, %0, %1")
PROBE(p_add_f32, "v_add_f32 %0, %0, %1")
PROBE(p_sub_f32, "v_sub_f32 %0, %0, %1")
PROBE(p_floor_f32, "v_floor_f32 %0, %0")
PROBE(p_fract_f32, "v_fract_f32 %0, %0")
PROBE(p_cvt_f32_ubyte0, "v_cvt_f32_ubyte0 %0, %0")
PROBE(p_cvt_f32_ubyte1, "v_cvt_f32_ubyte1 %0, %0")
PROBE(p_cvt_f32_ubyte3, "v_cvt_f32_ubyte3 %0, %0")
PROBE(p_cvt_pk_u8_f32, "v_cvt_pk_u8_f32 %0, %1, 1, %0")
PROBE(p_cvt_i32_f32, "v_cvt_i32_f32 %0, %0")
PROBE(p_cvt_f32_i32, "v_cvt_f32_i32 %0, %0")
PROBE(p_cvt_f32_u32, "v_cvt_f32_u32 %0, %0")
PROBE(p_cvt_flr_i32_f32, "v_cvt_flr_i32_f32 %0, %0")
PROBE(p_rcp_f32, "v_rcp_f32 %0, %0")
PROBE(p_alignbyte, "v_alignbyte_b32 %0, %0, %1, %2")
PROBE(p_alignbit, "v_alignbit_b32 %0, %0, %1, %2")
PROBE(p_perm, "v_perm_b32 %0, %0, %1, %2")
PROBE(p_and, "v_and_b32 %0, %0, %1")
PROBE(p_or3, "v_or3_b32 %0, %0, %1, %2")
PROBE(p_and_or, "v_and_or_b32 %0, %0, %1, %2")
PROBE(p_lshrrev, "v_lshrrev_b32 %0, 8, %0")
PROBE(p_lshlrev, "v_lshlrev_b32 %0, 3, %0")
PROBE(p_bfe_u32, "v_bfe_u32 %0, %0, 8, 8")
PROBE(p_add_u32, "v_add_u32 %0, %0, %1")
PROBE(p_sub_u32, "v_sub_u32 %0, %0, %1")
PROBE(p_add3_u32, "v_add3_u32 %0, %0, %1, %2")
PROBE(p_lshl_add_u32, "v_lshl_add_u32 %0, %0, 2, %1")
PROBE(p_add_lshl_u32, "v_add_lshl_u32 %0, %0, %1, 2")
PROBE(p_lshl_or_b32, "v_lshl_or_b32 %0, %0, 2, %1")
PROBE(p_mad_u32_u24, "v_mad_u32_u24 %0, %0, %1, %2")
PROBE(p_mul_u32_u24, "v_mul_u32_u24 %0, %0, %1")
PROBE(p_mad_i32_i24, "v_mad_i32_i24 %0, %0, %1, %2")
PROBE(p_mul_lo_u32, "v_mul_lo_u32 %0, %0, %1")
PROBE(p_cndmask, "v_cndmask_b32 %0, %0, %1, vcc")
PROBE(p_max_i32, "v_max_i32 %0, %0, %1")
PROBE(p_min_u32, "v_min_u32 %0, %0, %1")
PROBE(p_med3_i32, "v_med3_i32 %0, %0, %1, %2")
PROBE(p_max3_i32, "v_max3_i32 %0, %0, %1, %2")
PROBE(p_mov, "v_mov_b32 %0, %1")
PROBE(p_pk_add_u16, "v_pk_add_u16 %0, %0, %1")
PROBE(p_pk_sub_i16, "v_pk_sub_i16 %0, %0, %1")
PROBE(p_pk_lshrrev_b16, "v_pk_lshrrev_b16 %0, 8, %0 op_sel_hi:[0,1]")
PROBE(p_pk_mad_u16, "v_pk_mad_u16 %0, %0, %1, %2")
PROBE(p_pk_mul_lo_u16, "v_pk_mul_lo_u16 %0, %0, %1")
PROBE(p_pk_max_i16, "v_pk_max_i16 %0, %0, %1")
PROBE(p_mad_u16, "v_mad_u16 %0, %0, %1, %2")
PROBE(p_dot4_u32_u8, "v_dot4_u32_u8 %0, %0, %1, %2")
PROBE(p_dot2_u32_u16, "v_dot2_u32_u16 %0, %0, %1, %2")
PROBE(p_sad_u8, "v_sad_u8 %0, %0, %1, %2")
PROBE(p_lerp_u8, "v_lerp_u8 %0, %0, %1, %2")
PROBE(p_msad_u8, "v_msad_u8 %0, %0, %1, %2")
PROBE(p_add_u32_sdwa, "v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1")
PROBE(p_cvt_f32_i32_sdwa, "v_cvt_f32_i32_sdwa %0, sext(%0) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1")
PROBE(p_cmp_cnd, "v_cmp_lt_i32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %2, vcc")
PROBE(p_cmp_only, "v_cmp_lt_i32 vcc, %0, %1")
PROBE(p_readfirstlane_like_mov_dpp, "v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
PROBE64(p_pk_fma_f32, "v_pk_fma_f32 %0, %0, %3, %3")
PROBE64(p_pk_mul_f32, "v_pk_mul_f32 %0, %0, %3")
PROBE64(p_pk_add_f32, "v_pk_add_f32 %0, %0, %3")
PROBE64(p_mad_u64_u32, "v_mad_u64_u32 %0, vcc, %1, %2, %0")
PROBE64(p_lshl_add_u64, "v_lshl_add_u64 %0, %0, 2, %3")
PROBE64(p_lshrrev_b64, "v_lshrrev_b64 %0, 8, %0")

struct Probe { const char *name; void (*fn)(unsigned long long *, unsigned); int per_iter; };
#define P1(NAME) {#NAME, NAME, 1}
#define P2(NAME) {#NAME, NAME, 2}

int main(int argc, char **argv)
{
    std::vector<Probe> probes = {
        P1(p_fma_f32), P1(p_fmac_f32), P1(p_mul_f32), P1(p_add_f32), P1(p_sub_f32), P1(p_floor_f32), P1(p_fract_f32),
        P1(p_cvt_f32_ubyte0), P1(p_cvt_f32_ubyte1), P1(p_cvt_f32_ubyte3), P1(p_cvt_pk_u8_f32), P1(p_cvt_i32_f32), P1(p_cvt_f32_i32), P1(p_cvt_f32_u32),
        P1(p_cvt_flr_i32_f32), P1(p_rcp_f32),
        P1(p_alignbyte), P1(p_alignbit), P1(p_perm), P1(p_and), P1(p_or3), P1(p_and_or), P1(p_lshrrev), P1(p_lshlrev), P1(p_bfe_u32),
        P1(p_add_u32), P1(p_sub_u32), P1(p_add3_u32), P1(p_lshl_add_u32), P1(p_add_lshl_u32), P1(p_lshl_or_b32),
        P1(p_mad_u32_u24), P1(p_mul_u32_u24), P1(p_mad_i32_i24), P1(p_mul_lo_u32),
        P1(p_cndmask), P1(p_max_i32), P1(p_min_u32), P1(p_med3_i32), P1(p_max3_i32), P1(p_mov),
        P1(p_pk_add_u16), P1(p_pk_sub_i16), P1(p_pk_lshrrev_b16), P1(p_pk_mad_u16), P1(p_pk_mul_lo_u16), P1(p_pk_max_i16), P1(p_mad_u16),
        P1(p_dot4_u32_u8), P1(p_dot2_u32_u16), P1(p_sad_u8), P1(p_lerp_u8), P1(p_msad_u8),
        P1(p_add_u32_sdwa), P1(p_cvt_f32_i32_sdwa), P2(p_cmp_cnd), P1(p_cmp_only), P1(p_readfirstlane_like_mov_dpp),
        P1(p_pk_fma_f32), P1(p_pk_mul_f32), P1(p_pk_add_f32), P1(p_mad_u64_u32), P1(p_lshl_add_u64), P1(p_lshrrev_b64),
    };
    const char *only = argc > 1 ? argv[1] : nullptr;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    unsigned long long *d;
    CK(hipMalloc(&d, 16));
    printf("# %s, %d CUs; cycles per wave64 instruction per SIMD at W waves per SIMD (s_memtime ticks of the slowest wave / (instructions x W))\n", prop.name, cus);
    printf("%-28s %8s %8s %8s %8s\n", "instruction", "W=1", "W=2", "W=4", "W=8");
    for (auto &p : probes) {
        if (only && !strstr(p.name, only)) continue;
        printf("%-28s", p.name + 2);
        for (int W : {1, 2, 4, 8}) {
            // W workgroups of 256 lanes (4 waves = one per SIMD) per CU
            double best = 1e30;
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipMemset(d, 0, 16));
                hipLaunchKernelGGL(p.fn, dim3(cus * W), dim3(256), 0, 0, d, 12345u + rep);
                CK(hipDeviceSynchronize());
                unsigned long long t = 0;
                CK(hipMemcpy(&t, d, 8, hipMemcpyDeviceToHost));
                best = std::min(best, (double)t / ((double)ITERS * UNROLL * p.per_iter * W));
            }
            printf(" %8.2f", best);
        }
        printf("\n");
    }
    return 0;
}
