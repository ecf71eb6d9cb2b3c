// ubench_valu.hip — measures the issue cost (cycles per wave64 instruction per SIMD) of the VALU
// instructions the pricing kernels are made of, on the GPU it runs on.  Used to weight the
// instruction counts of the shipped inner loops into "full-rate-equivalent issue slots"
// (profiles/valu_slots.json) for the VALU roofline of the in-register path.
//   hipcc --offload-arch=gfx950 -O2 tools/ubench_valu.hip -o tools/ubench_valu && tools/ubench_valu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <string>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

#ifndef MCAMD_UBENCH_UNROLL
#define MCAMD_UBENCH_UNROLL 32
#endif
constexpr int UNROLL = MCAMD_UBENCH_UNROLL;  // instructions per loop trip (multiple of 8)

// Each kernel: ITERS x UNROLL copies of one instruction on 8 rotating registers.
#define KERNEL32(NAME, ASM)                                                                         \
    __global__ __launch_bounds__(256) void NAME(uint32_t *out, uint64_t *clk, int ITERS)                     \
    {                                                                                               \
        uint32_t r0 = threadIdx.x, r1 = r0 * 3 + 1, r2 = r0 * 5 + 2, r3 = r0 * 7 + 3, r4 = r0 + 11, \
                 r5 = r0 + 13, r6 = r0 + 17, r7 = r0 + 19;                                          \
        uint32_t a = 0x3f8ccccd, b = 0x3f99999a;                                                    \
        uint64_t t0 = __builtin_amdgcn_s_memtime(), w0 = __builtin_amdgcn_s_memrealtime();          \
        for (int i = 0; i < ITERS; ++i) {                                                           \
            _Pragma("unroll") for (int u = 0; u < UNROLL / 8; ++u) {                                \
                asm volatile(ASM(0) "\n" ASM(1) "\n" ASM(2) "\n" ASM(3) "\n" ASM(4) "\n" ASM(5) "\n" ASM(6) "\n" ASM(7) \
                             : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) \
                             : "v"(a), "v"(b));                                                     \
            }                                                                                       \
        }                                                                                           \
        uint64_t t1 = __builtin_amdgcn_s_memtime(), w1 = __builtin_amdgcn_s_memrealtime();          \
        out[blockIdx.x * 256 + threadIdx.x] = r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7;               \
        if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = w1 - w0; } \
    }

#define KERNEL64(NAME, ASM)                                                                         \
    __global__ __launch_bounds__(256) void NAME(uint32_t *out, uint64_t *clk, int ITERS)                     \
    {                                                                                               \
        double r0 = threadIdx.x * 1e-3 + 1.0, r1 = r0 + 0.1, r2 = r0 + 0.2, r3 = r0 + 0.3, r4 = r0 + 0.4, \
               r5 = r0 + 0.5, r6 = r0 + 0.6, r7 = r0 + 0.7;                                         \
        double a = 1.0000001, b = 1e-9;                                                             \
        uint64_t t0 = __builtin_amdgcn_s_memtime(), w0 = __builtin_amdgcn_s_memrealtime();          \
        for (int i = 0; i < ITERS; ++i) {                                                           \
            _Pragma("unroll") for (int u = 0; u < UNROLL / 8; ++u) {                                \
                asm volatile(ASM(0) "\n" ASM(1) "\n" ASM(2) "\n" ASM(3) "\n" ASM(4) "\n" ASM(5) "\n" ASM(6) "\n" ASM(7) \
                             : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) \
                             : "v"(a), "v"(b));                                                     \
            }                                                                                       \
        }                                                                                           \
        uint64_t t1 = __builtin_amdgcn_s_memtime(), w1 = __builtin_amdgcn_s_memrealtime();          \
        double s = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7;                                           \
        out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)__double_as_longlong(s);                    \
        if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = w1 - w0; } \
    }

#define A_FMA32(i) "v_fma_f32 %" #i ", %" #i ", %8, %9"
#define A_MUL32(i) "v_mul_f32 %" #i ", %" #i ", %8"
#define A_PKFMA32(i) "v_pk_fma_f32 %" #i ", %" #i ", %8, %9"
#define A_XOR(i) "v_xor_b32 %" #i ", %" #i ", %8"
#define A_ADDU(i) "v_add_u32 %" #i ", %" #i ", %8"
#define A_MULLO(i) "v_mul_lo_u32 %" #i ", %" #i ", %8"
#define A_MULHI(i) "v_mul_hi_u32 %" #i ", %" #i ", %8"
#define A_MUL24(i) "v_mul_u32_u24 %" #i ", %" #i ", %8"
#define A_MAD24(i) "v_mad_u32_u24 %" #i ", %" #i ", %8, %9"
#define A_EXP(i) "v_exp_f32 %" #i ", %" #i
#define A_LOG(i) "v_log_f32 %" #i ", %" #i
#define A_SIN(i) "v_sin_f32 %" #i ", %" #i
#define A_SQRT(i) "v_sqrt_f32 %" #i ", %" #i
#define A_RCP(i) "v_rcp_f32 %" #i ", %" #i
#define A_CVTU(i) "v_cvt_f32_u32 %" #i ", %" #i
#define A_CNDMASK(i) "v_cndmask_b32 %" #i ", %" #i ", %8, vcc"
#define A_FMA64(i) "v_fma_f64 %" #i ", %" #i ", %8, %9"
#define A_MUL64(i) "v_mul_f64 %" #i ", %" #i ", %8"
#define A_ADD64(i) "v_add_f64 %" #i ", %" #i ", %9"
#define A_RCP64(i) "v_rcp_f64 %" #i ", %" #i
#define A_RSQ64(i) "v_rsq_f64 %" #i ", %" #i
#define A_SQRT64(i) "v_sqrt_f64 %" #i ", %" #i
#define A_LDEXP64(i) "v_ldexp_f64 %" #i ", %" #i ", 1"
#define A_FRACT64(i) "v_fract_f64 %" #i ", %" #i
#define A_RNDNE64(i) "v_rndne_f64 %" #i ", %" #i
#define A_MADU64(i) "v_mad_u64_u32 %" #i ", vcc, %8, %9, %" #i
#define A_LSHL64(i) "v_lshlrev_b64 %" #i ", 3, %" #i
#define A_CVTF64U(i) "v_cvt_f64_u32 %" #i ", %8"
#define A_FREXPM64(i) "v_frexp_mant_f64 %" #i ", %" #i
#define A_MOV64(i) "v_mov_b64 %" #i ", %8"
#define A_MAX64(i) "v_max_f64 %" #i ", %" #i ", %8"
#define A_CMP64(i) "v_cmp_gt_f64 vcc, %" #i ", %8"
#define A_CVTI64(i) "v_cvt_i32_f64 %" #i ", %8"
#define A_LSHLADD64(i) "v_lshl_add_u64 %" #i ", %" #i ", 1, %8"
#define A_XOR3(i) "v_xor3_b32 %" #i ", %" #i ", %8, %9"
#define A_BITOP3(i) "v_bitop3_b32 %" #i ", %" #i ", %8, %9 bitop3:0x96"
#define A_CNDS(i) "v_cndmask_b32 %" #i ", %" #i ", %8, s[10:11]"
#define A_CMP32(i) "v_cmp_gt_f32 vcc, %" #i ", %8"
#define A_COS(i) "v_cos_f32 %" #i ", %" #i
#define A_LSHL(i) "v_lshlrev_b32 %" #i ", 3, %" #i
#define A_ALIGNBIT(i) "v_alignbit_b32 %" #i ", %" #i ", %8, 11"
#define A_ADD3(i) "v_add3_u32 %" #i ", %" #i ", %8, %9"
#define A_CVTF64F32(i) "v_cvt_f64_f32 %" #i ", %8"

KERNEL32(k_fma32, A_FMA32)
KERNEL32(k_mul32, A_MUL32)
KERNEL32(k_xor, A_XOR)
KERNEL32(k_addu, A_ADDU)
KERNEL32(k_mullo, A_MULLO)
KERNEL32(k_mulhi, A_MULHI)
KERNEL32(k_mul24, A_MUL24)
KERNEL32(k_mad24, A_MAD24)
KERNEL32(k_exp, A_EXP)
KERNEL32(k_log, A_LOG)
KERNEL32(k_sin, A_SIN)
KERNEL32(k_sqrt, A_SQRT)
KERNEL32(k_rcp, A_RCP)
KERNEL32(k_cvtu, A_CVTU)
KERNEL32(k_cndmask, A_CNDMASK)
KERNEL64(k_pkfma32, A_PKFMA32)
KERNEL64(k_fma64, A_FMA64)
KERNEL64(k_mul64, A_MUL64)
KERNEL64(k_add64, A_ADD64)
KERNEL64(k_rcp64, A_RCP64)
KERNEL64(k_rsq64, A_RSQ64)
KERNEL64(k_sqrt64, A_SQRT64)
KERNEL64(k_ldexp64, A_LDEXP64)
KERNEL64(k_fract64, A_FRACT64)
KERNEL64(k_rndne64, A_RNDNE64)
KERNEL64(k_lshl64, A_LSHL64)
KERNEL64(k_frexpm64, A_FREXPM64)
KERNEL64(k_mov64, A_MOV64)
KERNEL64(k_max64, A_MAX64)
KERNEL64(k_cmp64, A_CMP64)
KERNEL64(k_lshladd64, A_LSHLADD64)
KERNEL32(k_bitop3, A_BITOP3)
KERNEL32(k_cnds, A_CNDS)
KERNEL32(k_cmp32, A_CMP32)
KERNEL32(k_cos, A_COS)
KERNEL32(k_lshl, A_LSHL)
KERNEL32(k_alignbit, A_ALIGNBIT)
KERNEL32(k_add3, A_ADD3)


// v_mad_u64_u32: 64-bit accumulators, 32-bit multiplicands
__global__ __launch_bounds__(256) void k_madu64(uint32_t *out, uint64_t *clk, int ITERS)
{
    uint64_t r0 = threadIdx.x, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3, r4 = r0 + 4, r5 = r0 + 5, r6 = r0 + 6, r7 = r0 + 7;
    uint32_t a = 0xD2511F53u, b = threadIdx.x * 2654435761u;
    uint64_t t0 = __builtin_amdgcn_s_memtime(), w0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < ITERS; ++i) {
        _Pragma("unroll") for (int u = 0; u < UNROLL / 8; ++u) {
            asm volatile(A_MADU64(0) "\n" A_MADU64(1) "\n" A_MADU64(2) "\n" A_MADU64(3) "\n" A_MADU64(4) "\n" A_MADU64(5) "\n" A_MADU64(6) "\n" A_MADU64(7)
                         : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7)
                         : "v"(a), "v"(b) : "vcc");
        }
    }
    uint64_t t1 = __builtin_amdgcn_s_memtime(), w1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)(r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7);
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = w1 - w0; }
}

struct Entry { const char *name; void (*fn)(uint32_t *, uint64_t *, int); };

int main()
{
    hipDeviceProp_t p;
    CK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    const int waves_per_simd = 8;
    const int grid = cus * waves_per_simd;  // 256-thread blocks: 4 waves = 1 per SIMD -> 8 blocks per CU
    uint32_t *out; uint64_t *clk;
    CK(hipMalloc(&out, (size_t)grid * 256 * 4));
    CK(hipMalloc(&clk, (size_t)grid * 16));
    std::vector<uint64_t> h(2 * grid);
    Entry es[] = {{"v_fma_f32", k_fma32}, {"v_mul_f32", k_mul32}, {"v_pk_fma_f32", k_pkfma32}, {"v_xor_b32", k_xor},
                  {"v_add_u32", k_addu}, {"v_mul_lo_u32", k_mullo}, {"v_mul_hi_u32", k_mulhi}, {"v_mul_u32_u24", k_mul24},
                  {"v_mad_u32_u24", k_mad24}, {"v_mad_u64_u32", k_madu64}, {"v_exp_f32", k_exp}, {"v_log_f32", k_log},
                  {"v_sin_f32", k_sin}, {"v_sqrt_f32", k_sqrt}, {"v_rcp_f32", k_rcp}, {"v_cvt_f32_u32", k_cvtu},
                  {"v_cndmask_b32", k_cndmask}, {"v_fma_f64", k_fma64}, {"v_mul_f64", k_mul64}, {"v_add_f64", k_add64},
                  {"v_rcp_f64", k_rcp64}, {"v_rsq_f64", k_rsq64}, {"v_sqrt_f64", k_sqrt64}, {"v_ldexp_f64", k_ldexp64},
                  {"v_fract_f64", k_fract64}, {"v_rndne_f64", k_rndne64}, {"v_lshlrev_b64", k_lshl64},
                  {"v_frexp_mant_f64", k_frexpm64}, {"v_mov_b64", k_mov64}, {"v_max_f64", k_max64},
                  {"v_cmp_gt_f64", k_cmp64}, {"v_lshl_add_u64", k_lshladd64}, {"v_bitop3_b32", k_bitop3}, {"v_cndmask_b32(sgpr mask)", k_cnds}, {"v_cmp_gt_f32", k_cmp32},
                  {"v_cos_f32", k_cos}, {"v_lshlrev_b32", k_lshl}, {"v_alignbit_b32", k_alignbit},
                  {"v_add3_u32", k_add3}};
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("{\"device\": \"%s\", \"cus\": %d, \"waves_per_simd\": %d, \"rows\": [\n", p.name, cus, waves_per_simd);
    bool first = true;
    for (auto &e : es) {
        double ms_n[2], clock_ghz = 0;
        const int iters[2] = {1000, 3000};
        for (int k = 0; k < 2; ++k) {
            hipLaunchKernelGGL(e.fn, dim3(grid), dim3(256), 0, 0, out, clk, iters[k]);  // warm
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(e.fn, dim3(grid), dim3(256), 0, 0, out, clk, iters[k]);
            CK(hipEventRecord(e1));
            CK(hipDeviceSynchronize());
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            ms_n[k] = ms;
            CK(hipMemcpy(h.data(), clk, h.size() * 8, hipMemcpyDeviceToHost));
            double cyc = 0, real = 0;
            for (int i = 0; i < grid; ++i) { cyc += h[2 * i]; real += h[2 * i + 1]; }
            clock_ghz = cyc / (real * 10.0);  // memrealtime ticks at 100 MHz; clock of the long run
        }
        // wall-time difference of the two lengths cancels launch + ramp-up/down
        const double d_inst = (double)(iters[1] - iters[0]) * UNROLL * waves_per_simd;  // wave-instructions per SIMD
        const double cyc_per_inst = (ms_n[1] - ms_n[0]) * 1e-3 * clock_ghz * 1e9 / d_inst;
        printf("%s  {\"inst\": \"%s\", \"cycles_per_wave_inst_per_simd\": %.2f, \"clock_ghz\": %.3f, \"ms_1000\": %.3f, \"ms_3000\": %.3f}",
               first ? "" : ",\n", e.name, cyc_per_inst, clock_ghz, ms_n[0], ms_n[1]);
        first = false;
    }
    printf("\n]}\n");
    return 0;
}
