// valu_issue_microbench.hip -- issue cost of the instruction mix of lk_track_g16_kernel on gfx950 ("Next" 4c).
//
// bench.py's `valu_issue_frac` prices every VALU wave-instruction of the LK kernel at 2 cycles (the guide's figure for
// v_fma_f32 at >= 2 waves per SIMD).  The kernel's mix is v_dot2_i32_i16 / v_dot2c_i32_i16_dpp / v_mad_i32_i16 op_sel /
// v_perm_b32 / v_alignbyte / DPP adds plus ~20 fp64 instructions per Newton iteration: this program measures each of them.
//
// Method: W waves per SIMD (one 256-thread workgroup = one wave per SIMD; W workgroups per CU, 256 CUs), every wave runs
// N x 32 instances of ONE instruction (32 independent destinations, or one dependent chain), s_memtime around the loop;
// cycles per wave-instruction and SIMD = (t1 - t0) / (N * 32 * W) with all W waves of the SIMD running the same loop.
//
// The number of waves per SIMD is the workgroup size: ONE workgroup of 256 W threads per CU (100 KB of LDS each, grid = 256 CUs, one
// round), its 4 W waves dealt over the four SIMDs.  (Earlier versions launched W 256-thread workgroups per CU and let the dispatcher
// place them: the launch then took twice a wave's loop time -- not every workgroup was resident at once -- and the per-SIMD figures came
// out too low; `launch_over_loop` checks that here.)
// s_memtime ticks are shader cycles: the ticks_per_us column (s_memtime against the constant 100 MHz s_memrealtime) reads 2,300-2,400.
//
//   hipcc --offload-arch=gfx950 -O3 -o valu_issue_microbench valu_issue_microbench.hip && ./valu_issue_microbench > valu_issue_microbench.json
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <algorithm>
#include <string>
#include <vector>

#define R8(X, a, b) X(a, b, 0) X(a, b, 1) X(a, b, 2) X(a, b, 3) X(a, b, 4) X(a, b, 5) X(a, b, 6) X(a, b, 7)

// 32 instructions per block: destinations v[8..39] (independent) or all on v8 (chain)
#define BLK32_IND(OP_STR)                                                                       \
    asm volatile(OP_STR(8) OP_STR(9) OP_STR(10) OP_STR(11) OP_STR(12) OP_STR(13) OP_STR(14) OP_STR(15) \
                 OP_STR(16) OP_STR(17) OP_STR(18) OP_STR(19) OP_STR(20) OP_STR(21) OP_STR(22) OP_STR(23) \
                 OP_STR(24) OP_STR(25) OP_STR(26) OP_STR(27) OP_STR(28) OP_STR(29) OP_STR(30) OP_STR(31) \
                 OP_STR(32) OP_STR(33) OP_STR(34) OP_STR(35) OP_STR(36) OP_STR(37) OP_STR(38) OP_STR(39) \
                 ::: "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", \
                     "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", \
                     "v40", "v41", "vcc")

// 64-bit destinations: 16 register pairs v[8:9] .. v[38:39], two rounds
#define BLK32_IND64(OP_STR)                                                                     \
    asm volatile(OP_STR(8, 9) OP_STR(10, 11) OP_STR(12, 13) OP_STR(14, 15) OP_STR(16, 17) OP_STR(18, 19) OP_STR(20, 21) OP_STR(22, 23) \
                 OP_STR(24, 25) OP_STR(26, 27) OP_STR(28, 29) OP_STR(30, 31) OP_STR(32, 33) OP_STR(34, 35) OP_STR(36, 37) OP_STR(38, 39) \
                 OP_STR(8, 9) OP_STR(10, 11) OP_STR(12, 13) OP_STR(14, 15) OP_STR(16, 17) OP_STR(18, 19) OP_STR(20, 21) OP_STR(22, 23) \
                 OP_STR(24, 25) OP_STR(26, 27) OP_STR(28, 29) OP_STR(30, 31) OP_STR(32, 33) OP_STR(34, 35) OP_STR(36, 37) OP_STR(38, 39) \
                 ::: "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", \
                     "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "vcc")

#define S(x) #x
// sources: v2..v7 hold live lane data (initialised from the thread id, never written)
#define OP_FMA_F32(d)      "v_fma_f32 v" S(d) ", v2, v3, v4\n\t"
#define OP_ADD_U32(d)      "v_add_u32 v" S(d) ", v2, v3\n\t"
#define OP_ASHR(d)         "v_ashrrev_i32 v" S(d) ", 9, v2\n\t"
#define OP_AND(d)          "v_and_b32 v" S(d) ", 0xfffffe00, v2\n\t"
#define OP_DOT2(d)         "v_dot2_i32_i16 v" S(d) ", v2, v3, v4\n\t"
#define OP_DOT2C_DPP(d)    "v_dot2c_i32_i16_dpp v" S(d) ", v2, v3 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
#define OP_DOT2C(d)        "v_dot2c_i32_i16 v" S(d) ", v2, v3\n\t"
#define OP_MAD_I16(d)      "v_mad_i32_i16 v" S(d) ", v2, v3, v4\n\t"
#define OP_MAD_I16_SEL(d)  "v_mad_i32_i16 v" S(d) ", v2, v3, v4 op_sel:[0,1,0,0]\n\t"
#define OP_MAD_I24(d)      "v_mad_i32_i24 v" S(d) ", v2, v3, v4\n\t"
#define OP_PERM(d)         "v_perm_b32 v" S(d) ", v2, v3, s20\n\t"
#define OP_ALIGNBYTE(d)    "v_alignbyte_b32 v" S(d) ", v2, v3, v5\n\t"
#define OP_PK_ADD_U16(d)   "v_pk_add_u16 v" S(d) ", v2, v3\n\t"
#define OP_PK_MUL_U16(d)   "v_pk_mul_lo_u16 v" S(d) ", v2, v3\n\t"
#define OP_ADD_DPP_QUAD(d) "v_add_u32_dpp v" S(d) ", v2, v3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
#define OP_ADD_DPP_MIRR(d) "v_add_u32_dpp v" S(d) ", v2, v3 row_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
#define OP_CVT_F32_I32(d)  "v_cvt_f32_i32 v" S(d) ", v2\n\t"
#define OP_CVT_I32_F32(d)  "v_cvt_i32_f32 v" S(d) ", v6\n\t"
#define OP_RNDNE(d)        "v_rndne_f32 v" S(d) ", v6\n\t"
#define OP_FLOOR(d)        "v_floor_f32 v" S(d) ", v6\n\t"
#define OP_MUL_F32(d)      "v_mul_f32 v" S(d) ", v6, v7\n\t"
#define OP_PK_MUL_F32(d)   "v_mul_f32 v" S(d) ", v6, v7\n\t"
#define OP_SQRT_F32(d)     "v_sqrt_f32 v" S(d) ", v6\n\t"
#define OP_RCP_F32(d)      "v_rcp_f32 v" S(d) ", v6\n\t"
#define OP_CNDMASK(d)      "v_cndmask_b32 v" S(d) ", v2, v3, vcc\n\t"
#define OP_MOV(d)          "v_mov_b32 v" S(d) ", v2\n\t"
#define OP_SNOP(d)         "s_nop 0\n\t"
#define OP_CMP_F32(d)      "v_cmp_lt_f32 vcc, v6, v7\n\t"
#define OP_MIN_I32(d)      "v_min_i32 v" S(d) ", v2, v3\n\t"
#define OP_MAX_U32(d)      "v_max_u32 v" S(d) ", v2, v3\n\t"
#define OP_BFE_U32(d)      "v_bfe_u32 v" S(d) ", v2, 8, 8\n\t"
#define OP_LSHL_OR(d)      "v_lshl_or_b32 v" S(d) ", v2, 16, v3\n\t"
#define OP_AND_OR(d)       "v_and_or_b32 v" S(d) ", v2, v3, v4\n\t"
#define OP_ADD3(d)         "v_add3_u32 v" S(d) ", v2, v3, v4\n\t"
#define OP_BFI(d)          "v_bfi_b32 v" S(d) ", v2, v3, v4\n\t"
#define OP_ALIGNBIT(d)     "v_alignbit_b32 v" S(d) ", v2, v3, 9\n\t"
#define OP_SUB_SDWA(d)     "v_sub_u32_sdwa v" S(d) ", v2, v3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_2\n\t"
#define OP_PK_MIN_I16(d)   "v_pk_min_i16 v" S(d) ", v2, v3\n\t"
#define OP_PK_SUB_I16(d)   "v_pk_sub_i16 v" S(d) ", v2, v3\n\t"
#define OP_PK_MAD_U16(d)   "v_pk_mad_u16 v" S(d) ", v2, v3, v4\n\t"
#define OP_CMP_GT_I32(d)   "v_cmp_gt_i32 vcc, v2, v3\n\t"
#define OP_CNDMASK_E64(d)  "v_cndmask_b32_e64 v" S(d) ", v2, v3, s[22:23]\n\t"
#define OP_MUL_U24(d)      "v_mul_u32_u24 v" S(d) ", v2, v3\n\t"
#define OP_MOV_DPP(d)      "v_mov_b32_dpp v" S(d) ", v2 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
#define OP_DOT4_U8(d)      "v_dot4_u32_u8 v" S(d) ", v2, v3, v4\n\t"
#define OP_MAD_U32_U24(d)  "v_mad_u32_u24 v" S(d) ", v2, v3, v4\n\t"
#define OP_DS_READ(d)      "ds_read_b32 v" S(d) ", v5\n\ts_waitcnt lgkmcnt(8)\n\t"
// chains: every instruction reads the previous result
#define CH_FMA_F32(d)      "v_fma_f32 v8, v8, v3, v4\n\t"
#define CH_ADD_U32(d)      "v_add_u32 v8, v8, v3\n\t"
#define CH_DOT2(d)         "v_dot2_i32_i16 v8, v2, v3, v8\n\t"
#define CH_DOT2C_DPP(d)    "v_dot2c_i32_i16_dpp v8, v2, v3 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
#define CH_MAD_I16(d)      "v_mad_i32_i16 v8, v2, v3, v8\n\t"
#define CH_MAD_I16_SEL(d)  "v_mad_i32_i16 v8, v2, v3, v8 op_sel:[0,1,0,0]\n\t"
#define CH_MAD_I24(d)      "v_mad_i32_i24 v8, v2, v3, v8\n\t"
#define CH_PERM(d)         "v_perm_b32 v8, v8, v3, s20\n\t"
#define CH_ADD_DPP(d)      "s_nop 1\n\tv_add_u32_dpp v8, v8, v8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
#define CH_ASHR(d)         "v_ashrrev_i32 v8, 1, v8\n\t"
// fp64 (register pairs)
#define OP_CVT_F64_I32(a, b) "v_cvt_f64_i32 v[" S(a) ":" S(b) "], v2\n\t"
#define OP_CVT_F64_F32(a, b) "v_cvt_f64_f32 v[" S(a) ":" S(b) "], v6\n\t"
#define OP_CVT_F32_F64(a, b) "v_cvt_f32_f64 v" S(a) ", v[42:43]\n\t"
#define OP_LDEXP_F64(a, b)   "v_ldexp_f64 v[" S(a) ":" S(b) "], v[42:43], 16\n\t"
#define OP_ADD_F64(a, b)     "v_add_f64 v[" S(a) ":" S(b) "], v[42:43], v[44:45]\n\t"
#define OP_MUL_F64(a, b)     "v_mul_f64 v[" S(a) ":" S(b) "], v[42:43], v[44:45]\n\t"
#define OP_FMA_F64(a, b)     "v_fma_f64 v[" S(a) ":" S(b) "], v[42:43], v[44:45], v[46:47]\n\t"
#define OP_CMP_F64(a, b)     "v_cmp_lt_f64 vcc, v[42:43], v[44:45]\n\t"
#define CH_FMA_F64(a, b)     "v_fma_f64 v[8:9], v[8:9], v[44:45], v[46:47]\n\t"
#define CH_ADD_F64(a, b)     "v_add_f64 v[8:9], v[8:9], v[44:45]\n\t"

#define PROLOGUE                                                                                                        \
    asm volatile("v_mov_b32 v2, %0\n\tv_mov_b32 v3, %1\n\tv_mov_b32 v4, %2\n\tv_and_b32 v5, 0xfc, %0\n\t"               \
                 "v_cvt_f32_u32 v6, %0\n\tv_cvt_f32_u32 v7, %1\n\ts_mov_b32 s20, 0x0c010c00\n\ts_mov_b64 s[22:23], 0x5555\n\t"                        \
                 "v_cvt_f64_u32 v[42:43], %0\n\tv_cvt_f64_u32 v[44:45], %1\n\tv_cvt_f64_u32 v[46:47], %2\n\tv_mov_b32 v8, %0\n\tv_mov_b32 v9, 0\n\t" \
                 ::"v"(x), "v"(y), "v"(z) : "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v42", "v43", "v44", "v45", "v46", "v47", "s20", "s22", "s23")

#define KERNEL(NAME, BLOCK)                                                                             \
    __global__ __launch_bounds__(1024) void k_##NAME(uint64_t* out, int n, int seed)                     \
    {                                                                                                   \
        extern __shared__ uint32_t lds_dyn[];                                                           \
        __shared__ uint32_t lds[1024];                                                                  \
        if (seed == -2) lds_dyn[threadIdx.x] = 1;                                                       \
        lds[threadIdx.x] = threadIdx.x * seed;                                                          \
        __syncthreads();                                                                                \
        const uint32_t x = threadIdx.x * 2654435761u + seed, y = x ^ 0x5bd1e995u, z = x >> 3;           \
        PROLOGUE;                                                                                       \
        const uint64_t r0 = __builtin_amdgcn_s_memrealtime();                                           \
        const uint64_t t0 = __builtin_readcyclecounter();                                               \
        for (int i = 0; i < n; ++i) { BLOCK; }                                                          \
        const uint64_t t1 = __builtin_readcyclecounter();                                               \
        const uint64_t r1 = __builtin_amdgcn_s_memrealtime();                                           \
        if (threadIdx.x == 0 && blockIdx.x == 0) out[(size_t)gridDim.x * 16] = r1 - r0;                 \
        uint32_t sink;                                                                                  \
        asm volatile("v_add_u32 %0, v8, v9" : "=v"(sink));                                              \
        if (sink == 0x12345u && seed == -1) lds[0] = sink;                                              \
        if ((threadIdx.x & 63) == 0) out[(size_t)blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;       \
    }

KERNEL(fma_f32, BLK32_IND(OP_FMA_F32))
KERNEL(add_u32, BLK32_IND(OP_ADD_U32))
KERNEL(ashrrev_i32, BLK32_IND(OP_ASHR))
KERNEL(and_b32_literal, BLK32_IND(OP_AND))
KERNEL(dot2_i32_i16, BLK32_IND(OP_DOT2))
KERNEL(dot2c_i32_i16_dpp, BLK32_IND(OP_DOT2C_DPP))
KERNEL(dot2c_i32_i16, BLK32_IND(OP_DOT2C))
KERNEL(mad_i32_i16, BLK32_IND(OP_MAD_I16))
KERNEL(mad_i32_i16_opsel, BLK32_IND(OP_MAD_I16_SEL))
KERNEL(mad_i32_i24, BLK32_IND(OP_MAD_I24))
KERNEL(perm_b32, BLK32_IND(OP_PERM))
KERNEL(alignbyte_b32, BLK32_IND(OP_ALIGNBYTE))
KERNEL(pk_add_u16, BLK32_IND(OP_PK_ADD_U16))
KERNEL(pk_mul_lo_u16, BLK32_IND(OP_PK_MUL_U16))
KERNEL(add_u32_dpp_quad, BLK32_IND(OP_ADD_DPP_QUAD))
KERNEL(add_u32_dpp_row_mirror, BLK32_IND(OP_ADD_DPP_MIRR))
KERNEL(cvt_f32_i32, BLK32_IND(OP_CVT_F32_I32))
KERNEL(cvt_i32_f32, BLK32_IND(OP_CVT_I32_F32))
KERNEL(rndne_f32, BLK32_IND(OP_RNDNE))
KERNEL(floor_f32, BLK32_IND(OP_FLOOR))
KERNEL(mul_f32, BLK32_IND(OP_MUL_F32))
KERNEL(sqrt_f32, BLK32_IND(OP_SQRT_F32))
KERNEL(rcp_f32, BLK32_IND(OP_RCP_F32))
KERNEL(cndmask_b32, BLK32_IND(OP_CNDMASK))
KERNEL(mov_b32, BLK32_IND(OP_MOV))
KERNEL(s_nop, BLK32_IND(OP_SNOP))
KERNEL(cmp_lt_f32, BLK32_IND(OP_CMP_F32))
KERNEL(min_i32, BLK32_IND(OP_MIN_I32))
KERNEL(max_u32, BLK32_IND(OP_MAX_U32))
KERNEL(bfe_u32, BLK32_IND(OP_BFE_U32))
KERNEL(lshl_or_b32, BLK32_IND(OP_LSHL_OR))
KERNEL(and_or_b32, BLK32_IND(OP_AND_OR))
KERNEL(add3_u32, BLK32_IND(OP_ADD3))
KERNEL(bfi_b32, BLK32_IND(OP_BFI))
KERNEL(alignbit_b32, BLK32_IND(OP_ALIGNBIT))
KERNEL(sub_u32_sdwa_bytes, BLK32_IND(OP_SUB_SDWA))
KERNEL(pk_min_i16, BLK32_IND(OP_PK_MIN_I16))
KERNEL(pk_sub_i16, BLK32_IND(OP_PK_SUB_I16))
KERNEL(pk_mad_u16, BLK32_IND(OP_PK_MAD_U16))
KERNEL(cmp_gt_i32, BLK32_IND(OP_CMP_GT_I32))
KERNEL(cndmask_b32_sgpr_mask, BLK32_IND(OP_CNDMASK_E64))
KERNEL(mul_u32_u24, BLK32_IND(OP_MUL_U24))
KERNEL(mov_b32_dpp, BLK32_IND(OP_MOV_DPP))
KERNEL(dot4_u32_u8, BLK32_IND(OP_DOT4_U8))
KERNEL(mad_u32_u24, BLK32_IND(OP_MAD_U32_U24))
KERNEL(ds_read_b32, BLK32_IND(OP_DS_READ))
KERNEL(chain_fma_f32, BLK32_IND(CH_FMA_F32))
KERNEL(chain_add_u32, BLK32_IND(CH_ADD_U32))
KERNEL(chain_dot2_i32_i16, BLK32_IND(CH_DOT2))
KERNEL(chain_dot2c_dpp, BLK32_IND(CH_DOT2C_DPP))
KERNEL(chain_mad_i32_i16, BLK32_IND(CH_MAD_I16))
KERNEL(chain_mad_i32_i16_opsel, BLK32_IND(CH_MAD_I16_SEL))
KERNEL(chain_mad_i32_i24, BLK32_IND(CH_MAD_I24))
KERNEL(chain_perm_b32, BLK32_IND(CH_PERM))
KERNEL(chain_add_u32_dpp_snop1, BLK32_IND(CH_ADD_DPP))
KERNEL(chain_ashrrev, BLK32_IND(CH_ASHR))
KERNEL(cvt_f64_i32, BLK32_IND64(OP_CVT_F64_I32))
KERNEL(cvt_f64_f32, BLK32_IND64(OP_CVT_F64_F32))
KERNEL(cvt_f32_f64, BLK32_IND64(OP_CVT_F32_F64))
KERNEL(ldexp_f64, BLK32_IND64(OP_LDEXP_F64))
KERNEL(add_f64, BLK32_IND64(OP_ADD_F64))
KERNEL(mul_f64, BLK32_IND64(OP_MUL_F64))
KERNEL(fma_f64, BLK32_IND64(OP_FMA_F64))
KERNEL(cmp_lt_f64, BLK32_IND64(OP_CMP_F64))
KERNEL(chain_fma_f64, BLK32_IND64(CH_FMA_F64))
KERNEL(chain_add_f64, BLK32_IND64(CH_ADD_F64))

struct Entry { const char* name; void (*fn)(uint64_t*, int, int); };
#define E(NAME) {#NAME, k_##NAME}
static const Entry entries[] = {
    E(fma_f32), E(add_u32), E(ashrrev_i32), E(and_b32_literal), E(dot2_i32_i16), E(dot2c_i32_i16_dpp), E(dot2c_i32_i16), E(mad_i32_i16),
    E(mad_i32_i16_opsel), E(mad_i32_i24), E(perm_b32), E(alignbyte_b32), E(pk_add_u16), E(pk_mul_lo_u16), E(add_u32_dpp_quad),
    E(add_u32_dpp_row_mirror), E(cvt_f32_i32), E(cvt_i32_f32), E(rndne_f32), E(floor_f32), E(mul_f32), E(sqrt_f32), E(rcp_f32),
    E(cndmask_b32), E(mov_b32), E(s_nop), E(cmp_lt_f32), E(min_i32), E(max_u32), E(bfe_u32), E(lshl_or_b32), E(and_or_b32), E(add3_u32), E(bfi_b32),
    E(alignbit_b32), E(sub_u32_sdwa_bytes), E(pk_min_i16), E(pk_sub_i16), E(pk_mad_u16), E(cmp_gt_i32), E(cndmask_b32_sgpr_mask), E(mul_u32_u24),
    E(mov_b32_dpp), E(dot4_u32_u8), E(mad_u32_u24), E(ds_read_b32),
    E(chain_fma_f32), E(chain_add_u32), E(chain_dot2_i32_i16), E(chain_dot2c_dpp), E(chain_mad_i32_i16), E(chain_mad_i32_i16_opsel),
    E(chain_mad_i32_i24), E(chain_perm_b32), E(chain_add_u32_dpp_snop1), E(chain_ashrrev),
    E(cvt_f64_i32), E(cvt_f64_f32), E(cvt_f32_f64), E(ldexp_f64), E(add_f64), E(mul_f64), E(fma_f64), E(cmp_lt_f64), E(chain_fma_f64), E(chain_add_f64)};

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main()
{
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount, n = 2000;
    const int waves[] = {1, 2, 3, 4};       // waves per SIMD = workgroup size / 256: ONE workgroup per CU (100 KB of LDS each), grid = CUs: one round
    uint64_t* d_out;
    hipEvent_t ev0, ev1;
    CK(hipEventCreate(&ev0)); CK(hipEventCreate(&ev1));
    std::vector<uint64_t> h((size_t)cus * 16 + 1);
    CK(hipMalloc(&d_out, h.size() * sizeof(uint64_t)));
    printf("{\"device\": \"%s\", \"cus\": %d, \"instructions_per_wave\": %d, \"unit\": \"s_memtime ticks per wave-instruction and SIMD = loop ticks / (instructions x waves per SIMD); median over waves; w<k>_ns: the same from the launch's duration (HIP events)\", \"rows\": [\n", prop.gcnArchName, cus, n * 32);
    bool first = true;
    for (const Entry& e : entries) {
        printf("%s {\"op\": \"%s\"", first ? "" : ",\n", e.name);
        first = false;
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(e.fn), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
        for (int W : waves) {
            const int blocks = cus;
            const size_t dyn = 100 * 1024;                    // more than half a CU's LDS: one workgroup per CU, 4 W waves = W per SIMD
            hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(256 * W), dyn, 0, d_out, 50, 1);      // warm the clock / instruction cache
            CK(hipEventRecord(ev0, 0));
            hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(256 * W), dyn, 0, d_out, n, 1);
            CK(hipEventRecord(ev1, 0));
            CK(hipDeviceSynchronize());
            float ms = 0;
            CK(hipEventElapsedTime(&ms, ev0, ev1));
            CK(hipMemcpy(h.data(), d_out, ((size_t)blocks * 16 + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost));
            const double loop_us = (double)h[(size_t)blocks * 16] / 100.;          // s_memrealtime: constant 100 MHz
            const double wave0_ticks = (double)h[0];
            std::vector<uint64_t> v;
            for (int b = 0; b < blocks; ++b) for (int w = 0; w < 4 * W; ++w) v.push_back(h[(size_t)b * 16 + w]);
            std::sort(v.begin(), v.end());
            const double med = (double)v[v.size() / 2] / ((double)n * 32 * W);
            printf(", \"w%d\": %.2f", W, med);
            printf(", \"w%d_ns\": %.3f", W, (double)ms * 1e6 / ((double)n * 32 * W));
            if (W == 4) printf(", \"ticks_per_us\": %.1f, \"launch_over_loop\": %.2f", wave0_ticks / loop_us, (double)ms * 1e3 / loop_us);
        }
        printf("}");
        fflush(stdout);
    }
    printf("\n]}\n");
    (void)hipFree(d_out);
    return 0;
}
