// VGPR bank-conflict probe: does operand register placement change the issue cost of fp64 / integer VALU ops on gfx950?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// 64 copies per loop trip of one instruction with fixed registers; sources chosen per variant
#define REP8(X) X X X X X X X X
#define REP64(X) REP8(REP8(X))
#define KERNEL(NAME, INIT, BODY, CLOB)                                                            \
    __global__ __launch_bounds__(256) void NAME(uint32_t *out, int iters)                         \
    {                                                                                             \
        asm volatile(INIT ::: CLOB);                                                              \
        for (int i = 0; i < iters; ++i) asm volatile(REP64(BODY "\n") ::: CLOB);                \
        uint32_t r;                                                                               \
        asm volatile("v_mov_b32 %0, v40" : "=v"(r)::CLOB);                                        \
        out[blockIdx.x * 256 + threadIdx.x] = r;                                                  \
    }
#define CL "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "vcc"
#define INIT64 "v_mov_b32 v44, 0\n v_mov_b32 v45, 0x3ff00000\n v_mov_b32 v48, 0\n v_mov_b32 v49, 0x3ff00000\n v_mov_b32 v52, 0\n v_mov_b32 v53, 0x3ff00000\n v_mov_b32 v46, 0\n v_mov_b32 v47, 0x3ff00000\n v_mov_b32 v50, 0\n v_mov_b32 v51, 0x3ff00000\n v_mov_b32 v54,0\n v_mov_b32 v55,0x3ff00000\n v_mov_b32 v56,0\n v_mov_b32 v57,0x3ff00000\n v_mov_b32 v58, 0\n v_mov_b32 v59, 0x3ff00000"
// all three sources start at registers = 0 mod 4 (44, 48, 52): same bank if banks = reg mod 4
KERNEL(fma64_same, INIT64, "v_fma_f64 v[40:41], v[44:45], v[48:49], v[52:53]", CL)
// sources at 44, 46 (2 mod 4), 50 (2 mod 4)...: mixed
KERNEL(fma64_mix, INIT64, "v_fma_f64 v[40:41], v[44:45], v[46:47], v[50:51]", CL)
// sources at 44 (0), 46 (2), 53?? keep pairs even: 44, 46, 48 -> banks 0,2,0
KERNEL(fma64_020, INIT64, "v_fma_f64 v[40:41], v[44:45], v[46:47], v[48:49]", CL)
// two sources identical register (one read)
KERNEL(fma64_dup, INIT64, "v_fma_f64 v[40:41], v[44:45], v[44:45], v[46:47]", CL)
KERNEL(mul64_same, INIT64, "v_mul_f64 v[40:41], v[44:45], v[48:49]", CL)
KERNEL(mul64_diff, INIT64, "v_mul_f64 v[40:41], v[44:45], v[46:47]", CL)
KERNEL(fma32_same, INIT64, "v_fma_f32 v40, v44, v48, v52", CL)
KERNEL(fma32_diff, INIT64, "v_fma_f32 v40, v44, v45, v46", CL)
KERNEL(bitop_same, INIT64, "v_bitop3_b32 v40, v44, v48, v52 bitop3:0x96", CL)
KERNEL(bitop_diff, INIT64, "v_bitop3_b32 v40, v44, v45, v46 bitop3:0x96", CL)
KERNEL(bitop_sgpr, INIT64, "v_bitop3_b32 v40, v44, v45, s8 bitop3:0x96", CL)
KERNEL(mad_same, INIT64, "v_mad_u64_u32 v[40:41], vcc, v44, v48, v[52:53]", CL)
KERNEL(mad_diff, INIT64, "v_mad_u64_u32 v[40:41], vcc, v44, v45, v[46:47]", CL)
KERNEL(mad_sgpr0, INIT64, "v_mad_u64_u32 v[40:41], s[10:11], v44, s8, 0", CL)
KERNEL(fma64_sgpr, INIT64, "v_fma_f64 v[40:41], v[44:45], s[8:9], v[46:47]", CL)
KERNEL(fmac64_lit, INIT64, "v_fmac_f64 v[40:41], 0x41f00000, v[44:45]", CL)
KERNEL(add64_sgpr, INIT64, "v_add_f64 v[40:41], v[44:45], s[8:9]", CL)
KERNEL(xor_vv_diff, INIT64, "v_xor_b32 v40, v44, v45", CL)
KERNEL(xor_vv_same, INIT64, "v_xor_b32 v40, v44, v48", CL)
KERNEL(xor_sv, INIT64, "v_xor_b32 v40, s8, v44", CL)
KERNEL(bitop_2same, INIT64, "v_bitop3_b32 v40, v44, v48, v45 bitop3:0x96", CL)
KERNEL(bitop_dst_src, INIT64, "v_bitop3_b32 v44, v44, v45, v46 bitop3:0x96", CL)
KERNEL(fma32_sgpr, INIT64, "v_fma_f32 v40, v44, s8, v45", CL)
KERNEL(fma32_2same, INIT64, "v_fma_f32 v40, v44, v48, v45", CL)
KERNEL(fmac32_diff, INIT64, "v_fmac_f32 v40, v44, v45", CL)
KERNEL(fmamk32, INIT64, "v_fmamk_f32 v40, v44, 0x2f800000, v45", CL)
KERNEL(mul32_sv, INIT64, "v_mul_f32 v40, s8, v44", CL)
KERNEL(add3_same, INIT64, "v_add3_u32 v40, v44, v48, v52", CL)
KERNEL(and_or_diff, INIT64, "v_and_or_b32 v40, v44, v45, v46", CL)
KERNEL(mad_vvs, INIT64, "v_mad_u64_u32 v[40:41], vcc, v44, s8, v[46:47]", CL)
KERNEL(mad_2same, INIT64, "v_mad_u64_u32 v[40:41], vcc, v44, v48, v[46:47]", CL)

// mixes: two (or three) different instructions per repetition, independent registers — do their issue costs add?
#define INITB INIT64 "\n v_mov_b32 v60, 1\n v_mov_b32 v61, 2\n v_mov_b32 v62, 3\n v_mov_b32 v63, 5"
KERNEL(mix_fma64_bitop, INITB, "v_fma_f64 v[40:41], v[44:45], v[46:47], v[50:51]\n v_bitop3_b32 v60, v61, v62, v63 bitop3:0x96", CL)
KERNEL(mix_fma64_mad, INITB, "v_fma_f64 v[40:41], v[44:45], v[46:47], v[50:51]\n v_mad_u64_u32 v[42:43], vcc, v61, v62, v[56:57]", CL)
KERNEL(mix_mad_bitop, INITB, "v_mad_u64_u32 v[42:43], vcc, v61, v62, v[56:57]\n v_bitop3_b32 v60, v61, v62, v63 bitop3:0x96", CL)
KERNEL(mix_fma64_2bitop, INITB, "v_fma_f64 v[40:41], v[44:45], v[46:47], v[50:51]\n v_bitop3_b32 v60, v61, v62, v63 bitop3:0x96\n v_bitop3_b32 v58, v61, v62, v63 bitop3:0x96", CL)
KERNEL(mix_fma64_bitop_sgpr, INITB, "v_fma_f64 v[40:41], v[44:45], v[46:47], v[50:51]\n v_bitop3_b32 v60, v61, v62, s8 bitop3:0x96", CL)
// dependent chains: every instruction reads the previous result
KERNEL(dep_fma64, INITB, "v_fma_f64 v[40:41], v[40:41], v[46:47], v[50:51]", CL)
KERNEL(dep_mad_bitop, INITB, "v_mad_u64_u32 v[42:43], vcc, v60, v62, v[56:57]\n v_bitop3_b32 v60, v43, v62, v63 bitop3:0x96", CL)
KERNEL(mix_fma64_exp, INITB, "v_fma_f64 v[40:41], v[44:45], v[46:47], v[50:51]\n v_exp_f32 v60, v61", CL)
KERNEL(mix_fma32_exp, INITB, "v_fma_f32 v40, v44, v45, v46\n v_exp_f32 v60, v61", CL)
KERNEL(mix_fma64_ds, INITB, "v_fma_f64 v[40:41], v[44:45], v[46:47], v[50:51]\n v_fma_f64 v[42:43], v[44:45], v[46:47], v[50:51]\n v_fma_f64 v[58:59], v[44:45], v[46:47], v[50:51]", CL)

// operand forms of the integer / shift / conversion instructions of the fp64 loop: inline constant vs literal vs register
KERNEL(lshl_inl, INITB, "v_lshlrev_b32 v40, 3, v61", CL)
KERNEL(lshl_vv, INITB, "v_lshlrev_b32 v40, v60, v61", CL)
KERNEL(lshr_inl, INITB, "v_lshrrev_b32 v40, 8, v61", CL)
KERNEL(ashr_inl, INITB, "v_ashrrev_i32 v40, 16, v61", CL)
KERNEL(and_inl, INITB, "v_and_b32 v40, 15, v61", CL)
KERNEL(and_lit, INITB, "v_and_b32 v40, 0xff0, v61", CL)
KERNEL(and_vv, INITB, "v_and_b32 v40, v60, v61", CL)
KERNEL(add_inl, INITB, "v_add_u32 v40, 1, v61", CL)
KERNEL(add_vv, INITB, "v_add_u32 v40, v60, v61", CL)
KERNEL(sub_vv, INITB, "v_sub_u32 v40, v60, v61", CL)
KERNEL(or_vv, INITB, "v_or_b32 v40, v60, v61", CL)
KERNEL(bfe_inl, INITB, "v_bfe_u32 v40, v61, 8, 8", CL)
KERNEL(bfe_vv, INITB, "v_bfe_u32 v40, v61, v62, v63", CL)
KERNEL(lshl_or_inl, INITB, "v_lshl_or_b32 v40, v61, 4, v62", CL)
KERNEL(lshl_add_inl, INITB, "v_lshl_add_u32 v40, v61, 4, v62", CL)
KERNEL(perm_vvv, INITB, "v_perm_b32 v40, v61, v62, v63", CL)
KERNEL(alignbit_vvv, INITB, "v_alignbit_b32 v40, v61, v62, v63", CL)
KERNEL(mov_sdwa, INITB, "v_mov_b32_sdwa v40, v61 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1", CL)
KERNEL(lshl_sdwa, INITB, "v_lshlrev_b32_sdwa v40, v60, v61 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1", CL)
KERNEL(cvt_f64_u32, INITB, "v_cvt_f64_u32 v[40:41], v61", CL)
KERNEL(cvt_f64_i32, INITB, "v_cvt_f64_i32 v[40:41], v61", CL)
KERNEL(cvt_f32_u32, INITB, "v_cvt_f32_u32 v40, v61", CL)
KERNEL(cvt_f32_ubyte, INITB, "v_cvt_f32_ubyte1 v40, v61", CL)
KERNEL(fma64_inl, INITB, "v_fma_f64 v[40:41], v[44:45], 1.0, v[46:47]", CL)
KERNEL(mul64_inl, INITB, "v_mul_f64 v[40:41], 0.5, v[44:45]", CL)
KERNEL(mul32_inl, INITB, "v_mul_f32 v40, 2.0, v61", CL)
KERNEL(fma32_inl, INITB, "v_fma_f32 v40, v61, 2.0, v62", CL)
KERNEL(mov_b32, INITB, "v_mov_b32 v40, v61", CL)
KERNEL(ldexp64_vv, INITB, "v_ldexp_f64 v[40:41], v[44:45], v60", CL)
KERNEL(mix_mad_2bitop, INITB, "v_mad_u64_u32 v[42:43], vcc, v61, v62, v[56:57]\n v_bitop3_b32 v60, v61, v62, v63 bitop3:0x96\n v_bitop3_b32 v58, v61, v62, v63 bitop3:0x96", CL)
KERNEL(mix_2mad_2bitop, INITB, "v_mad_u64_u32 v[42:43], vcc, v61, v62, v[56:57]\n v_mad_u64_u32 v[40:41], vcc, v61, v62, v[56:57]\n v_bitop3_b32 v60, v61, v62, v63 bitop3:0x96\n v_bitop3_b32 v58, v61, v62, v63 bitop3:0x96", CL)
KERNEL(mix_4mad_4bitop, INITB, "v_mad_u64_u32 v[42:43], vcc, v61, v62, v[56:57]\n v_mad_u64_u32 v[40:41], vcc, v61, v62, v[56:57]\n v_mad_u64_u32 v[54:55], vcc, v61, v62, v[56:57]\n v_mad_u64_u32 v[52:53], vcc, v61, v62, v[56:57]\n v_bitop3_b32 v60, v61, v62, v63 bitop3:0x96\n v_bitop3_b32 v58, v61, v62, v63 bitop3:0x96\n v_bitop3_b32 v59, v61, v62, v63 bitop3:0x96\n v_bitop3_b32 v51, v61, v62, v63 bitop3:0x96", CL)
KERNEL(mix_2fma64_2bitop, INITB, "v_fma_f64 v[40:41], v[44:45], v[46:47], v[50:51]\n v_fma_f64 v[42:43], v[44:45], v[46:47], v[50:51]\n v_bitop3_b32 v60, v61, v62, v63 bitop3:0x96\n v_bitop3_b32 v58, v61, v62, v63 bitop3:0x96", CL)

struct E { const char *n; void (*f)(uint32_t *, int); };
int main()
{
    const int grid = 256 * 8;
    uint32_t *out; CK(hipMalloc(&out, (size_t)grid * 256 * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    E es[] = {{"fma64_same", fma64_same}, {"fma64_mix", fma64_mix}, {"fma64_020", fma64_020}, {"fma64_dup", fma64_dup}, {"mul64_same", mul64_same}, {"mul64_diff", mul64_diff},
              {"fma32_same", fma32_same}, {"fma32_diff", fma32_diff}, {"bitop_same", bitop_same}, {"bitop_diff", bitop_diff}, {"bitop_sgpr", bitop_sgpr},
              {"mad_same", mad_same}, {"mad_diff", mad_diff}, {"mad_sgpr0", mad_sgpr0}, {"fma64_sgpr", fma64_sgpr}, {"fmac64_lit", fmac64_lit}, {"add64_sgpr", add64_sgpr},
              {"xor_vv_diff", xor_vv_diff}, {"xor_vv_same", xor_vv_same}, {"xor_sv", xor_sv}, {"bitop_2same", bitop_2same}, {"bitop_dst_src", bitop_dst_src},
              {"fma32_sgpr", fma32_sgpr}, {"fma32_2same", fma32_2same}, {"fmac32_diff", fmac32_diff}, {"fmamk32", fmamk32},
              {"mul32_sv", mul32_sv}, {"add3_same", add3_same}, {"and_or_diff", and_or_diff}, 
              {"mad_vvs", mad_vvs}, {"mad_2same", mad_2same},
              {"mix_fma64_bitop(2)", mix_fma64_bitop}, {"mix_fma64_mad(2)", mix_fma64_mad}, {"mix_mad_bitop(2)", mix_mad_bitop},
              {"mix_fma64_2bitop(3)", mix_fma64_2bitop}, {"mix_fma64_bitop_sgpr(2)", mix_fma64_bitop_sgpr}, {"dep_fma64", dep_fma64},
              {"dep_mad_bitop(2)", dep_mad_bitop}, {"mix_fma64_exp(2)", mix_fma64_exp}, {"mix_fma32_exp(2)", mix_fma32_exp}, {"3x_fma64(3)", mix_fma64_ds},
              {"lshl_inl", lshl_inl}, {"lshl_vv", lshl_vv}, {"lshr_inl", lshr_inl}, {"ashr_inl", ashr_inl}, {"and_inl", and_inl}, {"and_lit", and_lit},
              {"and_vv", and_vv}, {"add_inl", add_inl}, {"add_vv", add_vv}, {"sub_vv", sub_vv}, {"or_vv", or_vv}, {"bfe_inl", bfe_inl}, {"bfe_vv", bfe_vv},
              {"lshl_or_inl", lshl_or_inl}, {"lshl_add_inl", lshl_add_inl}, {"perm_vvv", perm_vvv}, {"alignbit_vvv", alignbit_vvv}, {"mov_sdwa", mov_sdwa},
              {"lshl_sdwa", lshl_sdwa}, {"cvt_f64_u32", cvt_f64_u32}, {"cvt_f64_i32", cvt_f64_i32}, {"cvt_f32_u32", cvt_f32_u32}, {"cvt_f32_ubyte1", cvt_f32_ubyte},
              {"fma64_inl", fma64_inl}, {"mul64_inl", mul64_inl}, {"mul32_inl", mul32_inl}, {"fma32_inl", fma32_inl}, {"mov_b32", mov_b32}, {"ldexp64_vv", ldexp64_vv},
              {"mix_mad_2bitop(3)", mix_mad_2bitop}, {"mix_2mad_2bitop(4)", mix_2mad_2bitop}, {"mix_4mad_4bitop(8)", mix_4mad_4bitop}, {"mix_2fma64_2bitop(4)", mix_2fma64_2bitop}};
    for (auto &e : es) {
        float ms[2];
        const int it[2] = {500, 1500};
        for (int k = 0; k < 2; ++k) {
            hipLaunchKernelGGL(e.f, dim3(grid), dim3(256), 0, 0, out, it[k]); CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0)); hipLaunchKernelGGL(e.f, dim3(grid), dim3(256), 0, 0, out, it[k]); CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
            CK(hipEventElapsedTime(&ms[k], e0, e1));
        }
        // per-SIMD wave-instructions: 8 waves x 64 x iters; time difference -> ns per wave-instruction per SIMD
        const double ninst = 8.0 * 64.0 * (it[1] - it[0]);
        // a "(k)" suffix in the name = k instructions per repetition: the figure is per REPETITION
        printf("%-24s %.3f ns per repetition per SIMD  (x2.4 GHz = %.2f cycles)\n", e.n, (ms[1] - ms[0]) * 1e6 / ninst, (ms[1] - ms[0]) * 1e6 / ninst * 2.4);
    }
    return 0;
}
