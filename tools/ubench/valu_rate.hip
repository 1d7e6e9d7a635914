// Issue rate of every VALU form the DP sweeps use or could use (gfx950), in wave-instructions per shader
// cycle per SIMD, for 1 .. 8 waves per SIMD.  The instructions are written in inline asm so that the
// listing is exactly the mix named; the shader clock comes from s_memtime inside the kernel, so the
// ceiling and a kernel's SQ / GRBM counters share one clock (no 2.4 GHz assumption).
// Also: semantic probes for DPP row_newbcast + bank_mask (a one-instruction "lane 15 -> lanes of a bank").
// build: hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip ; run: ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>

#define R8(OPS) OPS(a0, a1) OPS(a1, a2) OPS(a2, a3) OPS(a3, a4) OPS(a4, a5) OPS(a5, a6) OPS(a6, a7) OPS(a7, a0)
// 8 independent chains: every instruction writes its own register, reading (itself, one constant)
#define SELF8(INS) \
    asm volatile(INS(%0) INS(%1) INS(%2) INS(%3) INS(%4) INS(%5) INS(%6) INS(%7) \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c1), "v"(c2) : "vcc", "s20", "s21");
#define PAIR8(INS) \
    asm volatile(INS(%0, %1) INS(%1, %2) INS(%2, %3) INS(%3, %4) INS(%4, %5) INS(%5, %6) INS(%6, %7) INS(%7, %0) \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c1), "v"(c2) : "vcc", "s20", "s21");
#define SELF4D(INS) \
    asm volatile(INS(%0) INS(%1) INS(%2) INS(%3) INS(%0) INS(%1) INS(%2) INS(%3) \
                 : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(dc) : "vcc");

#define I_ADD_U32(x)    "v_add_u32 " #x ", " #x ", %8\n"
#define I_MAX_I32(x)    "v_max_i32 " #x ", " #x ", %8\n"
#define I_ADD_F32(x)    "v_add_f32 " #x ", " #x ", %8\n"
#define I_MAX_F32(x)    "v_max_f32 " #x ", " #x ", %8\n"
#define I_FMA_F32(x)    "v_fma_f32 " #x ", " #x ", %8, %9\n"
#define I_CNDMASK(x)    "v_cndmask_b32 " #x ", " #x ", %8, vcc\n"
#define I_CMP_I32(x)    "v_cmp_gt_i32 vcc, " #x ", %8\n"
#define I_CMP_F32(x)    "v_cmp_gt_f32 vcc, " #x ", %8\n"
#define I_CMPSEL_I32(x) "v_cmp_gt_i32 vcc, " #x ", %8\n v_cndmask_b32 " #x ", " #x ", %9, vcc\n"
#define I_CMPSEL_F32(x) "v_cmp_gt_f32 vcc, " #x ", %8\n v_cndmask_b32 " #x ", " #x ", %9, vcc\n"
#define I_MAX3_I32(x)   "v_max3_i32 " #x ", " #x ", %8, %9\n"
#define I_MAX3_F32(x)   "v_max3_f32 " #x ", " #x ", %8, %9\n"
#define I_ADD3_U32(x)   "v_add3_u32 " #x ", " #x ", %8, %9\n"
#define I_LSHLADD(x)    "v_lshl_add_u32 " #x ", " #x ", 1, %9\n"
#define I_ANDOR(x)      "v_and_or_b32 " #x ", " #x ", %8, %9\n"
#define I_BFI(x)        "v_bfi_b32 " #x ", %8, " #x ", %9\n"
#define I_MED3_I32(x)   "v_med3_i32 " #x ", " #x ", %8, %9\n"
#define I_MIN_U32(x)    "v_min_u32 " #x ", " #x ", %8\n"
#define I_ASHR(x)       "v_ashrrev_i32 " #x ", 1, " #x "\n"
#define I_BFE(x)        "v_bfe_i32 " #x ", " #x ", 0, 16\n"
#define I_PK_ADD_I16(x) "v_pk_add_i16 " #x ", " #x ", %8 clamp\n"
#define I_PK_MAX_I16(x) "v_pk_max_i16 " #x ", " #x ", %8\n"
#define I_PK_SUB_I16(x) "v_pk_sub_i16 " #x ", " #x ", %8\n"
#define I_ADD_I16(x)    "v_add_i16 " #x ", " #x ", %8 clamp\n"
#define I_MAX_I16(x)    "v_max_i16 " #x ", " #x ", %8\n"
#define I_CMPSEL_I16(x) "v_cmp_gt_i16 vcc, " #x ", %8\n v_cndmask_b32 " #x ", " #x ", %9, vcc\n"
#define I_ADD_I32C(x)   "v_add_i32 " #x ", " #x ", %8 clamp\n"
#define I_MAD_I24(x)    "v_mad_i32_i24 " #x ", " #x ", 1, %8\n"
#define I_SDWA_ADD(x)   "v_add_u32_sdwa " #x ", " #x ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n"
#define I_PERM(x)       "v_perm_b32 " #x ", " #x ", %8, %9\n"
// cross-lane forms: dst <- src of the NEXT chain register
#define I_DPP_MOV(x, y)     "v_mov_b32_dpp " #x ", " #y " row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define I_DPP_NEWB(x, y)    "v_mov_b32_dpp " #x ", " #y " row_newbcast:15 row_mask:0xf bank_mask:0x2\n"
#define I_DPP_ADD(x, y)     "v_add_u32_dpp " #x ", " #y ", " #x " row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define I_DPP_MAX(x, y)     "v_max_i32_dpp " #x ", " #y ", " #x " row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define I_DPP_ADDF(x, y)    "v_add_f32_dpp " #x ", " #y ", " #x " row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define I_DPP_MAXF(x, y)    "v_max_f32_dpp " #x ", " #y ", " #x " row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define I_DPP_CND(x, y)     "v_cndmask_b32_dpp " #x ", " #x ", " #y ", vcc row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define I_DPP_CMP(x, y)     "v_cmp_gt_i32_e64 s[20:21], " #y ", " #x "\n"
#define I_BPERM(x, y)       "ds_bpermute_b32 " #x ", %8, " #y "\n"
#define I_M_SUB_U32(x) "v_sub_u32 " #x ", " #x ", %8\n"
#define I_M_SUBREV(x) "v_subrev_u32 " #x ", " #x ", %8\n"
#define I_M_AND(x) "v_and_b32 " #x ", " #x ", %8\n"
#define I_M_OR(x) "v_or_b32 " #x ", " #x ", %8\n"
#define I_M_XOR(x) "v_xor_b32 " #x ", " #x ", %8\n"
#define I_M_LSHL(x) "v_lshlrev_b32 " #x ", 1, " #x "\n"
#define I_M_LSHR(x) "v_lshrrev_b32 " #x ", 1, " #x "\n"
#define I_M_MOV(x) "v_mov_b32 " #x ", %8\n"
#define I_M_MUL_F32(x) "v_mul_f32 " #x ", " #x ", %8\n"
#define I_M_SUB_F32(x) "v_sub_f32 " #x ", " #x ", %8\n"
#define I_M_MIN_I32(x) "v_min_i32 " #x ", " #x ", %8\n"
#define I_M_MAX_U32(x) "v_max_u32 " #x ", " #x ", %8\n"
#define I_M_MIN_F32(x) "v_min_f32 " #x ", " #x ", %8\n"
#define I_M_MAX_U16(x) "v_max_u16 " #x ", " #x ", %8\n"
#define I_M_MIN_I16(x) "v_min_i16 " #x ", " #x ", %8\n"
#define I_M_ADD_U16(x) "v_add_u16 " #x ", " #x ", %8\n"
#define I_M_SUB_U16(x) "v_sub_u16 " #x ", " #x ", %8\n"
#define I_M_MUL_LO_U16(x) "v_mul_lo_u16 " #x ", " #x ", %8\n"
#define I_M_ASHR16(x) "v_ashrrev_i16 " #x ", 1, " #x "\n"
#define I_M_LSHL16(x) "v_lshlrev_b16 " #x ", 1, " #x "\n"
#define I_M_ADD_CO(x) "v_add_co_u32 " #x ", vcc, " #x ", %8\n"
#define I_M_FMAC(x) "v_fmac_f32 " #x ", %8, %9\n"
#define I_M_MUL_I24(x) "v_mul_i32_i24 " #x ", " #x ", %8\n"
#define I_M_CVT(x) "v_cvt_f32_i32 " #x ", " #x "\n"
#define I_M_NOT(x) "v_not_b32 " #x ", " #x "\n"
#define I_M_MAX_F16(x) "v_max_f16 " #x ", " #x ", %8\n"
#define I_M_ADD_F16(x) "v_add_f16 " #x ", " #x ", %8\n"
#define I_M_PK_ADD_U16(x) "v_pk_add_u16 " #x ", " #x ", %8\n"
#define I_M_PK_MIN_I16(x) "v_pk_min_i16 " #x ", " #x ", %8\n"
#define I_M_PK_MAX_F16(x) "v_pk_max_f16 " #x ", " #x ", %8\n"
#define I_M_ADD_I16_NC(x) "v_add_i16 " #x ", " #x ", %8\n"
#define I_M_SUB_I16_C(x) "v_sub_i16 " #x ", " #x ", %8 clamp\n"
#define I_M_ADD_U16_C(x) "v_add_u16_e64 " #x ", " #x ", %8 clamp\n"
#define I_M_MAX_I32_E64(x) "v_max_i32_e64 " #x ", " #x ", %8\n"
#define I_M_ADD_U32_E64(x) "v_add_u32_e64 " #x ", " #x ", %8\n"
#define I_M_CND_E64(x) "v_cndmask_b32_e64 " #x ", " #x ", %8, s[20:21]\n"
#define I_M_MIX_ADD_MAX(x) "v_add_u32 " #x ", " #x ", %8\n v_max_i32 " #x ", " #x ", %9\n"
#define I_M_MIX_ADD_ADD_MAX(x) "v_add_u32 " #x ", " #x ", %8\n v_add_u32 " #x ", " #x ", %9\n v_max_i32 " #x ", " #x ", %9\n"
#define I_M_MIX_MAX16_ADD(x) "v_max_i16 " #x ", " #x ", %8\n v_add_u32 " #x ", " #x ", %9\n"
#define I_M_MIX_CMP_MAX_CND(x) "v_cmp_gt_i32 vcc, " #x ", %8\n v_max_i32 " #x ", " #x ", %8\n v_cndmask_b32 " #x ", " #x ", %9, vcc\n"
#define I_M_MIX_CLAMPADD(x) "v_add_i32 " #x ", " #x ", %8 clamp\n v_add_u32 " #x ", " #x ", %9\n"
// 64-bit register forms
#define I_PK_ADD_F32(x) "v_pk_add_f32 " #x ", " #x ", %4\n"
#define I_PK_FMA_F32(x) "v_pk_fma_f32 " #x ", " #x ", %4, %4\n"
#define I_MAX_F64(x)    "v_max_f64 " #x ", " #x ", %4\n"

enum {
    M_ADD_U32, M_MAX_I32, M_ADD_F32, M_MAX_F32, M_FMA_F32, M_CNDMASK, M_CMP_I32, M_CMP_F32, M_CMPSEL_I32, M_CMPSEL_F32,
    M_MAX3_I32, M_MAX3_F32, M_ADD3, M_LSHLADD, M_ANDOR, M_BFI, M_MED3, M_MIN_U32, M_ASHR, M_BFE,
    M_PK_ADD_I16, M_PK_MAX_I16, M_PK_SUB_I16, M_ADD_I16, M_MAX_I16, M_CMPSEL_I16, M_ADD_I32C, M_MAD_I24, M_SDWA, M_PERM,
    M_DPP_MOV, M_DPP_NEWB, M_DPP_ADD, M_DPP_MAX, M_DPP_ADDF, M_DPP_MAXF, M_DPP_CND, M_DPP_CMP, M_BPERM,
    M_PK_ADD_F32, M_PK_FMA_F32, M_MAX_F64, M_SUB_U32, M_SUBREV, M_AND, M_OR, M_XOR, M_LSHL, M_LSHR, M_MOV, M_MUL_F32, M_SUB_F32, M_MIN_I32, M_MAX_U32, M_MIN_F32, M_MAX_U16, M_MIN_I16, M_ADD_U16, M_SUB_U16, M_MUL_LO_U16, M_ASHR16, M_LSHL16, M_ADD_CO, M_FMAC, M_MUL_I24, M_CVT, M_NOT, M_MAX_F16, M_ADD_F16, M_PK_ADD_U16, M_PK_MIN_I16, M_PK_MAX_F16, M_ADD_I16_NC, M_SUB_I16_C, M_ADD_U16_C, M_MAX_I32_E64, M_ADD_U32_E64, M_CND_E64, M_MIX_ADD_MAX, M_MIX_ADD_ADD_MAX, M_MIX_MAX16_ADD, M_MIX_CMP_MAX_CND, M_MIX_CLAMPADD, M_COUNT
};

struct ModeInfo { const char* name; int per8; };
static const ModeInfo g_modes[M_COUNT] = {
    {"v_add_u32", 8}, {"v_max_i32", 8}, {"v_add_f32", 8}, {"v_max_f32", 8}, {"v_fma_f32", 8}, {"v_cndmask_b32", 8},
    {"v_cmp_gt_i32", 8}, {"v_cmp_gt_f32", 8}, {"cmp_i32+cndmask", 16}, {"cmp_f32+cndmask", 16},
    {"v_max3_i32", 8}, {"v_max3_f32", 8}, {"v_add3_u32", 8}, {"v_lshl_add_u32", 8}, {"v_and_or_b32", 8}, {"v_bfi_b32", 8},
    {"v_med3_i32", 8}, {"v_min_u32", 8}, {"v_ashrrev_i32", 8}, {"v_bfe_i32", 8},
    {"v_pk_add_i16 clamp", 8}, {"v_pk_max_i16", 8}, {"v_pk_sub_i16", 8}, {"v_add_i16 clamp", 8}, {"v_max_i16", 8},
    {"cmp_i16+cndmask", 16}, {"v_add_i32 clamp", 8}, {"v_mad_i32_i24", 8}, {"v_add_u32_sdwa", 8}, {"v_perm_b32", 8},
    {"v_mov_b32_dpp shr1", 8}, {"v_mov_dpp newbcast", 8}, {"v_add_u32_dpp", 8}, {"v_max_i32_dpp", 8}, {"v_add_f32_dpp", 8},
    {"v_max_f32_dpp", 8}, {"v_cndmask_dpp", 8}, {"v_cmp_gt_i32_e64 sgpr", 8}, {"ds_bpermute_b32", 8},
    {"v_pk_add_f32", 8}, {"v_pk_fma_f32", 8}, {"v_max_f64", 8},
    {"v_sub_u32", 8}, {"v_subrev_u32", 8}, {"v_and_b32", 8}, {"v_or_b32", 8}, {"v_xor_b32", 8}, {"v_lshlrev_b32", 8}, {"v_lshrrev_b32", 8}, {"v_mov_b32", 8}, {"v_mul_f32", 8}, {"v_sub_f32", 8}, {"v_min_i32", 8}, {"v_max_u32", 8}, {"v_min_f32", 8}, {"v_max_u16", 8}, {"v_min_i16", 8}, {"v_add_u16", 8}, {"v_sub_u16", 8}, {"v_mul_lo_u16", 8}, {"v_ashrrev_i16", 8}, {"v_lshlrev_b16", 8}, {"v_add_co_u32", 8}, {"v_fmac_f32", 8}, {"v_mul_i32_i24", 8}, {"v_cvt_f32_i32", 8}, {"v_not_b32", 8}, {"v_max_f16", 8}, {"v_add_f16", 8}, {"v_pk_add_u16", 8}, {"v_pk_min_i16", 8}, {"v_pk_max_f16", 8}, {"v_add_i16 (no clamp)", 8}, {"v_sub_i16 clamp", 8}, {"v_add_u16_e64 clamp", 8}, {"v_max_i32_e64", 8}, {"v_add_u32_e64", 8}, {"v_cndmask_e64 sgpr", 8}, {"mix add,max (i32)", 16}, {"mix add,add,max", 24}, {"mix max_i16,add_u32", 16}, {"mix cmp,max,cnd", 24}, {"v_add_i32 clamp+v_add_u32", 16},
};

template <int MODE>
__global__ void __launch_bounds__(256) spin(int* out, long long* cyc, int iters, int seed)
{
    int a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 ^ 11, a5 = a0 ^ 13, a6 = a0 + 17, a7 = a0 - 19;
    const int c1 = seed | 1, c2 = seed | 3;
    double d0 = a0, d1 = a1, d2 = a2, d3 = a3; const double dc = seed;
    asm volatile("s_mov_b64 vcc, 0x5555\n s_mov_b64 s[20:21], 0x3333" ::: "vcc", "s20", "s21");
    const long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if constexpr (MODE == M_ADD_U32) SELF8(I_ADD_U32)
            else if constexpr (MODE == M_MAX_I32) SELF8(I_MAX_I32)
            else if constexpr (MODE == M_ADD_F32) SELF8(I_ADD_F32)
            else if constexpr (MODE == M_MAX_F32) SELF8(I_MAX_F32)
            else if constexpr (MODE == M_FMA_F32) SELF8(I_FMA_F32)
            else if constexpr (MODE == M_CNDMASK) SELF8(I_CNDMASK)
            else if constexpr (MODE == M_CMP_I32) SELF8(I_CMP_I32)
            else if constexpr (MODE == M_CMP_F32) SELF8(I_CMP_F32)
            else if constexpr (MODE == M_CMPSEL_I32) SELF8(I_CMPSEL_I32)
            else if constexpr (MODE == M_CMPSEL_F32) SELF8(I_CMPSEL_F32)
            else if constexpr (MODE == M_MAX3_I32) SELF8(I_MAX3_I32)
            else if constexpr (MODE == M_MAX3_F32) SELF8(I_MAX3_F32)
            else if constexpr (MODE == M_ADD3) SELF8(I_ADD3_U32)
            else if constexpr (MODE == M_LSHLADD) SELF8(I_LSHLADD)
            else if constexpr (MODE == M_ANDOR) SELF8(I_ANDOR)
            else if constexpr (MODE == M_BFI) SELF8(I_BFI)
            else if constexpr (MODE == M_MED3) SELF8(I_MED3_I32)
            else if constexpr (MODE == M_MIN_U32) SELF8(I_MIN_U32)
            else if constexpr (MODE == M_ASHR) SELF8(I_ASHR)
            else if constexpr (MODE == M_BFE) SELF8(I_BFE)
            else if constexpr (MODE == M_PK_ADD_I16) SELF8(I_PK_ADD_I16)
            else if constexpr (MODE == M_PK_MAX_I16) SELF8(I_PK_MAX_I16)
            else if constexpr (MODE == M_PK_SUB_I16) SELF8(I_PK_SUB_I16)
            else if constexpr (MODE == M_ADD_I16) SELF8(I_ADD_I16)
            else if constexpr (MODE == M_MAX_I16) SELF8(I_MAX_I16)
            else if constexpr (MODE == M_CMPSEL_I16) SELF8(I_CMPSEL_I16)
            else if constexpr (MODE == M_ADD_I32C) SELF8(I_ADD_I32C)
            else if constexpr (MODE == M_MAD_I24) SELF8(I_MAD_I24)
            else if constexpr (MODE == M_SDWA) SELF8(I_SDWA_ADD)
            else if constexpr (MODE == M_PERM) SELF8(I_PERM)
            else if constexpr (MODE == M_DPP_MOV) PAIR8(I_DPP_MOV)
            else if constexpr (MODE == M_DPP_NEWB) PAIR8(I_DPP_NEWB)
            else if constexpr (MODE == M_DPP_ADD) PAIR8(I_DPP_ADD)
            else if constexpr (MODE == M_DPP_MAX) PAIR8(I_DPP_MAX)
            else if constexpr (MODE == M_DPP_ADDF) PAIR8(I_DPP_ADDF)
            else if constexpr (MODE == M_DPP_MAXF) PAIR8(I_DPP_MAXF)
            else if constexpr (MODE == M_DPP_CND) PAIR8(I_DPP_CND)
            else if constexpr (MODE == M_DPP_CMP) PAIR8(I_DPP_CMP)
            else if constexpr (MODE == M_BPERM) { PAIR8(I_BPERM) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
            else if constexpr (MODE == M_PK_ADD_F32) SELF4D(I_PK_ADD_F32)
            else if constexpr (MODE == M_PK_FMA_F32) SELF4D(I_PK_FMA_F32)
            else if constexpr (MODE == M_MAX_F64) SELF4D(I_MAX_F64)
            else if constexpr (MODE == M_SUB_U32) SELF8(I_M_SUB_U32)
            else if constexpr (MODE == M_SUBREV) SELF8(I_M_SUBREV)
            else if constexpr (MODE == M_AND) SELF8(I_M_AND)
            else if constexpr (MODE == M_OR) SELF8(I_M_OR)
            else if constexpr (MODE == M_XOR) SELF8(I_M_XOR)
            else if constexpr (MODE == M_LSHL) SELF8(I_M_LSHL)
            else if constexpr (MODE == M_LSHR) SELF8(I_M_LSHR)
            else if constexpr (MODE == M_MOV) SELF8(I_M_MOV)
            else if constexpr (MODE == M_MUL_F32) SELF8(I_M_MUL_F32)
            else if constexpr (MODE == M_SUB_F32) SELF8(I_M_SUB_F32)
            else if constexpr (MODE == M_MIN_I32) SELF8(I_M_MIN_I32)
            else if constexpr (MODE == M_MAX_U32) SELF8(I_M_MAX_U32)
            else if constexpr (MODE == M_MIN_F32) SELF8(I_M_MIN_F32)
            else if constexpr (MODE == M_MAX_U16) SELF8(I_M_MAX_U16)
            else if constexpr (MODE == M_MIN_I16) SELF8(I_M_MIN_I16)
            else if constexpr (MODE == M_ADD_U16) SELF8(I_M_ADD_U16)
            else if constexpr (MODE == M_SUB_U16) SELF8(I_M_SUB_U16)
            else if constexpr (MODE == M_MUL_LO_U16) SELF8(I_M_MUL_LO_U16)
            else if constexpr (MODE == M_ASHR16) SELF8(I_M_ASHR16)
            else if constexpr (MODE == M_LSHL16) SELF8(I_M_LSHL16)
            else if constexpr (MODE == M_ADD_CO) SELF8(I_M_ADD_CO)
            else if constexpr (MODE == M_FMAC) SELF8(I_M_FMAC)
            else if constexpr (MODE == M_MUL_I24) SELF8(I_M_MUL_I24)
            else if constexpr (MODE == M_CVT) SELF8(I_M_CVT)
            else if constexpr (MODE == M_NOT) SELF8(I_M_NOT)
            else if constexpr (MODE == M_MAX_F16) SELF8(I_M_MAX_F16)
            else if constexpr (MODE == M_ADD_F16) SELF8(I_M_ADD_F16)
            else if constexpr (MODE == M_PK_ADD_U16) SELF8(I_M_PK_ADD_U16)
            else if constexpr (MODE == M_PK_MIN_I16) SELF8(I_M_PK_MIN_I16)
            else if constexpr (MODE == M_PK_MAX_F16) SELF8(I_M_PK_MAX_F16)
            else if constexpr (MODE == M_ADD_I16_NC) SELF8(I_M_ADD_I16_NC)
            else if constexpr (MODE == M_SUB_I16_C) SELF8(I_M_SUB_I16_C)
            else if constexpr (MODE == M_ADD_U16_C) SELF8(I_M_ADD_U16_C)
            else if constexpr (MODE == M_MAX_I32_E64) SELF8(I_M_MAX_I32_E64)
            else if constexpr (MODE == M_ADD_U32_E64) SELF8(I_M_ADD_U32_E64)
            else if constexpr (MODE == M_CND_E64) SELF8(I_M_CND_E64)
            else if constexpr (MODE == M_MIX_ADD_MAX) SELF8(I_M_MIX_ADD_MAX)
            else if constexpr (MODE == M_MIX_ADD_ADD_MAX) SELF8(I_M_MIX_ADD_ADD_MAX)
            else if constexpr (MODE == M_MIX_MAX16_ADD) SELF8(I_M_MIX_MAX16_ADD)
            else if constexpr (MODE == M_MIX_CMP_MAX_CND) SELF8(I_M_MIX_CMP_MAX_CND)
            else if constexpr (MODE == M_MIX_CLAMPADD) SELF8(I_M_MIX_CLAMPADD)
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (int) (d0 + d1 + d2 + d3);
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

// s_memtime runs at a constant 100 MHz-class reference on some parts; calibrate it against the wall clock and
// report both "per s_memtime tick" and "per wall-derived cycle at the GRBM clock" -- the caller reads the latter
template <int MODE>
static void run(int cus, int* out, long long* cyc, std::vector<long long>& hc)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4000;
    const int wpsv[4] = {1, 2, 4, 8};
    printf("%-26s", g_modes[MODE].name);
    for (int wi = 0; wi < 4; ++wi) {
        const int wps = wpsv[wi];
        const int grid = cus * wps;
        spin<MODE><<<grid, 256>>>(out, cyc, 50, 1);
        hipDeviceSynchronize();
        hipEventRecord(e0); spin<MODE><<<grid, 256>>>(out, cyc, iters, 1); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(hc.data(), cyc, sizeof(long long) * grid * 4, hipMemcpyDeviceToHost);
        double mean = 0; for (int i = 0; i < grid * 4; ++i) mean += hc[i]; mean /= grid * 4;
        const double winst_per_wave = (double) iters * 16 * g_modes[MODE].per8;
        // instructions issued per SIMD per tick of the wave's own clock: wps waves share the SIMD
        const double per_tick = winst_per_wave * wps / mean;
        const double per_s = winst_per_wave * grid * 4 / (cus * 4) / (ms * 1e-3);
        printf("  w%d: %.3f/tick %.2fG/s", wps, per_tick, per_s / 1e9);
    }
    printf("\n");
}

template <int M> static void run_all(int cus, int* out, long long* cyc, std::vector<long long>& hc)
{
    if constexpr (M < M_COUNT) { run<M>(cus, out, cyc, hc); run_all<M + 1>(cus, out, cyc, hc); }
}

// ---- semantics of DPP row_newbcast with a bank mask: which lanes are written, with what
__global__ void newbcast_probe(int* o)
{
    const int lane = threadIdx.x;
    int src = 1000 + lane, dst = -1;
    asm volatile("v_mov_b32_dpp %0, %1 row_newbcast:15 row_mask:0xf bank_mask:0x2" : "+v"(dst) : "v"(src));
    o[lane] = dst;
    int dst2 = -1;
    asm volatile("v_mov_b32_dpp %0, %1 row_newbcast:15 row_mask:0x5 bank_mask:0xf" : "+v"(dst2) : "v"(src));
    o[64 + lane] = dst2;
    int dst3 = -1;
    asm volatile("v_mov_b32_dpp %0, %1 row_ror:1 row_mask:0xf bank_mask:0x1" : "+v"(dst3) : "v"(src));
    o[128 + lane] = dst3;
}

int main()
{
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    printf("# %s, %d CUs, clockRate %d kHz\n", prop.gcnArchName, cus, prop.clockRate);
    int* out; hipMalloc(&out, sizeof(int) * 256 * cus * 8);
    long long* cyc; hipMalloc(&cyc, sizeof(long long) * cus * 8 * 4);
    std::vector<long long> hc(cus * 8 * 4);
    // calibrate the s_memtime tick: a known-length kernel, ticks / wall
    {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        spin<M_ADD_U32><<<cus, 256>>>(out, cyc, 50, 1); hipDeviceSynchronize();
        hipEventRecord(e0); spin<M_ADD_U32><<<cus, 256>>>(out, cyc, 20000, 1); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(hc.data(), cyc, sizeof(long long) * cus * 4, hipMemcpyDeviceToHost);
        double mean = 0; for (int i = 0; i < cus * 4; ++i) mean += hc[i]; mean /= cus * 4;
        printf("# s_memtime: %.1f ticks per us of kernel wall time (= %.3f GHz tick)\n", mean / (ms * 1e3), mean / (ms * 1e6));
    }
    printf("# columns: waves per SIMD; wave-instructions per SIMD per s_memtime tick; G wave-instructions / s / SIMD (wall)\n");
    run_all<0>(cus, out, cyc, hc);
    int* o; hipMalloc(&o, sizeof(int) * 192);
    newbcast_probe<<<1, 64>>>(o);
    int ho[192]; hipMemcpy(ho, o, sizeof(ho), hipMemcpyDeviceToHost);
    const char* nm[3] = {"row_newbcast:15 bank_mask:0x2", "row_newbcast:15 row_mask:0x5", "row_ror:1 bank_mask:0x1"};
    for (int t = 0; t < 3; ++t) {
        printf("# probe %s (src = 1000 + lane, dst preset -1):\n#  ", nm[t]);
        for (int l = 0; l < 64; ++l) printf("%d%s", ho[t * 64 + l], (l & 15) == 15 ? "\n#  " : " ");
        printf("\n");
    }
    return 0;
}
