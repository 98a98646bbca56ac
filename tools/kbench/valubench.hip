// valubench.hip - developer micro-benchmark: issue cost of the vector-ALU instructions the framing kernels are
// made of, on gfx950.  Every kernel runs N iterations of 32 independent instructions of one kind per wavefront, at
// W wavefronts per SIMD on every CU; cycles per instruction per SIMD = shader cycles of the launch * SIMDs /
// (wavefronts * instructions).  The shader clock is read with s_memtime beside the 100 MHz wall clock.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o valubench valubench.hip && ./valubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

#define R8(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)
// 32 instructions: 4 rounds over 8 destination registers
#define BODY(ASM3)                                                                                      \
  asm volatile(ASM3("%0") ASM3("%1") ASM3("%2") ASM3("%3") ASM3("%4") ASM3("%5") ASM3("%6") ASM3("%7") \
               ASM3("%0") ASM3("%1") ASM3("%2") ASM3("%3") ASM3("%4") ASM3("%5") ASM3("%6") ASM3("%7") \
               ASM3("%0") ASM3("%1") ASM3("%2") ASM3("%3") ASM3("%4") ASM3("%5") ASM3("%6") ASM3("%7") \
               ASM3("%0") ASM3("%1") ASM3("%2") ASM3("%3") ASM3("%4") ASM3("%5") ASM3("%6") ASM3("%7") \
               : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]), "+s"(sc) \
               : "v"(a), "v"(b) : "vcc", "scc")

#define KERNEL(NAME, ASM3)                                                                        \
  __global__ __launch_bounds__(256) void NAME(uint32_t* out, int iters, uint32_t seed, unsigned long long* clk) { \
    uint32_t r[8];                                                                                \
    for (int i = 0; i < 8; ++i) r[i] = threadIdx.x * 2654435761u + i + seed;                      \
    uint32_t a = seed ^ threadIdx.x, b = seed * 3u + 1u;                                          \
    uint32_t sc = seed;                                                                           \
    const unsigned long long t0 = __builtin_readcyclecounter(), w0 = wall_clock64();              \
    for (int it = 0; it < iters; ++it) { BODY(ASM3); }                                            \
    const unsigned long long t1 = __builtin_readcyclecounter(), w1 = wall_clock64();              \
    uint32_t x = 0;                                                                               \
    for (int i = 0; i < 8; ++i) x ^= r[i];                                                        \
    if (x == 0x12345u) out[0] = x;                                                                \
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = w1 - w0; }             \
  }

#define A_XOR(D) "v_xor_b32 " D ", " D ", %9\n"
#define A_ADD(D) "v_add_u32 " D ", " D ", %9\n"
#define A_SUB(D) "v_sub_u32 " D ", %9, " D "\n"
#define A_AND_LIT(D) "v_and_b32 " D ", 0x80808080, " D "\n"
#define A_XOR_LIT(D) "v_xor_b32 " D ", 0x0a0a0a0a, " D "\n"
#define A_LSHL_OR(D) "v_lshl_or_b32 " D ", " D ", 1, %10\n"
#define A_OR3(D) "v_or3_b32 " D ", " D ", %9, %10\n"
#define A_AND_OR(D) "v_and_or_b32 " D ", " D ", %9, %10\n"
#define A_BITOP3(D) "v_bitop3_b32 " D ", " D ", %9, %10 bitop3:0x80\n"
#define A_PERM(D) "v_perm_b32 " D ", %9, %10, " D "\n"
#define A_DOT4(D) "v_dot4_u32_u8 " D ", " D ", %9, %10\n"
#define A_DOT4_LIT(D) "v_dot4_u32_u8 " D ", " D ", 0x08040201, %10\n"
#define A_ALIGNBIT(D) "v_alignbit_b32 " D ", %9, " D ", 1\n"
#define A_XAD(D) "v_xad_u32 " D ", " D ", %9, %10\n"
#define A_BFI(D) "v_bfi_b32 " D ", %9, " D ", %10\n"
#define A_MUL24(D) "v_mul_u32_u24 " D ", " D ", %9\n"
#define A_MULLO(D) "v_mul_lo_u32 " D ", " D ", %9\n"
#define A_MAD24(D) "v_mad_u32_u24 " D ", " D ", %9, %10\n"
#define A_BCNT(D) "v_bcnt_u32_b32 " D ", " D ", %10\n"
#define A_LSHR(D) "v_lshrrev_b32 " D ", 1, " D "\n"
#define A_ADD3(D) "v_add3_u32 " D ", " D ", %9, %10\n"
#define A_LSHL_ADD(D) "v_lshl_add_u32 " D ", " D ", 1, %10\n"
#define A_DPP(D) "v_mov_b32_dpp " D ", " D " row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define A_XOR_DPP(D) "v_xor_b32_dpp " D ", " D ", %9 row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define A_CNDMASK(D) "v_cndmask_b32 " D ", " D ", %9, vcc\n"
#define A_PKADD(D) "v_pk_add_u16 " D ", " D ", %9\n"
#define A_PKMIN(D) "v_pk_min_u16 " D ", " D ", %9\n"
#define A_SAD(D) "v_sad_u8 " D ", " D ", %9, %10\n"
#define A_MSAD(D) "v_msad_u8 " D ", " D ", %9, %10\n"
#define A_XOR_SGPR(D) "v_xor_b32 " D ", %8, " D "\n"
#define A_FMA(D) "v_fma_f32 " D ", " D ", %9, %10\n"
#define A_PKFMA(D) "v_pk_add_f32 " D ", " D ", " D "\n"
#define A_SDWA(D) "v_or_b32_sdwa " D ", " D ", %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n"
#define A_BFE(D) "v_bfe_u32 " D ", " D ", 3, 8\n"
#define A_FFBL(D) "v_ffbl_b32 " D ", " D "\n"
#define A_MIX_SALU(D) "v_xor_b32 " D ", " D ", %9\n s_add_u32 %8, %8, 1\n"
#define A_MIX_SALU2(D) "v_xor_b32 " D ", " D ", %9\n s_add_u32 %8, %8, 1\n s_xor_b32 %8, %8, 5\n"
#define A_CMP(D) "v_cmp_eq_u32 vcc, " D ", %9\n"
#define A_CMP_SDWA(D) "v_cmp_eq_u32_sdwa vcc, " D ", %9 src0_sel:BYTE_1 src1_sel:DWORD\n"
#define A_MAX3(D) "v_max3_u32 " D ", " D ", %9, %10\n"
#define A_MED3(D) "v_med3_u32 " D ", " D ", %9, %10\n"
#define A_PERM_LIT(D) "v_perm_b32 " D ", 0x474e7f54, 0x437f417f, " D "\n"
#define A_MBCNT(D) "v_mbcnt_lo_u32_b32 " D ", %9, " D "\n"
#define A_READLANE(D) "v_readlane_b32 %8, " D ", 3\n"

#define A_OR(D) "v_or_b32 " D ", " D ", %9\n"
#define A_AND(D) "v_and_b32 " D ", " D ", %9\n"
#define A_LSHL(D) "v_lshlrev_b32 " D ", 1, " D "\n"
#define A_LSHL_V(D) "v_lshlrev_b32 " D ", %9, " D "\n"
#define A_MOV(D) "v_mov_b32 " D ", %9\n"
#define A_NOT(D) "v_not_b32 " D ", " D "\n"
#define A_AND_SGPR(D) "v_and_b32 " D ", %8, " D "\n"
#define A_ADD_SGPR(D) "v_add_u32 " D ", %8, " D "\n"
#define A_XOR_INL(D) "v_xor_b32 " D ", 10, " D "\n"
#define A_SUB_LIT(D) "v_sub_u32 " D ", 0x80808080, " D "\n"
#define A_SUBREV(D) "v_subrev_u32 " D ", %9, " D "\n"
#define A_ADD_LIT(D) "v_add_u32 " D ", 0x60606060, " D "\n"
#define A_BITOP3_SGPR(D) "v_bitop3_b32 " D ", " D ", %8, %10 bitop3:0x80\n"
#define A_BITOP3_OR3(D) "v_bitop3_b32 " D ", " D ", %9, %10 bitop3:0xfe\n"
#define A_MIN(D) "v_min_u32 " D ", " D ", %9\n"
#define A_MAX(D) "v_max_u32 " D ", " D ", %9\n"
#define A_ADDCO(D) "v_add_co_u32 " D ", vcc, " D ", %9\n"
#define A_XNOR(D) "v_xnor_b32 " D ", " D ", %9\n"
#define A_ASHR(D) "v_ashrrev_i32 " D ", 1, " D "\n"
#define A_MIX_2V_1S(D) "v_xor_b32 " D ", " D ", %9\n v_add_u32 " D ", " D ", %10\n s_add_u32 %8, %8, 1\n"
#define A_PAIR_2_4(D) "v_xor_b32 " D ", " D ", %9\n v_perm_b32 " D ", %9, %10, " D "\n"
#define A_DEP_XOR(D) "v_xor_b32 %0, %0, %9\n"
#define A_DEP_PERM(D) "v_perm_b32 %0, %9, %10, %0\n"
#define A_DEP_BITOP3(D) "v_bitop3_b32 %0, %0, %9, %10 bitop3:0x96\n"
KERNEL(k_xor, A_XOR) KERNEL(k_add, A_ADD) KERNEL(k_sub, A_SUB) KERNEL(k_and_lit, A_AND_LIT) KERNEL(k_xor_lit, A_XOR_LIT)
KERNEL(k_lshl_or, A_LSHL_OR) KERNEL(k_or3, A_OR3) KERNEL(k_and_or, A_AND_OR) KERNEL(k_bitop3, A_BITOP3) KERNEL(k_perm, A_PERM)
KERNEL(k_dot4, A_DOT4) KERNEL(k_alignbit, A_ALIGNBIT) KERNEL(k_xad, A_XAD) KERNEL(k_bfi, A_BFI)
KERNEL(k_mul24, A_MUL24) KERNEL(k_mullo, A_MULLO) KERNEL(k_mad24, A_MAD24) KERNEL(k_bcnt, A_BCNT) KERNEL(k_lshr, A_LSHR)
KERNEL(k_add3, A_ADD3) KERNEL(k_lshl_add, A_LSHL_ADD) KERNEL(k_dpp, A_DPP) KERNEL(k_xor_dpp, A_XOR_DPP) KERNEL(k_pkadd, A_PKADD)
KERNEL(k_pkmin, A_PKMIN) KERNEL(k_sad, A_SAD) KERNEL(k_msad, A_MSAD) KERNEL(k_xor_sgpr, A_XOR_SGPR) KERNEL(k_fma, A_FMA)
KERNEL(k_sdwa, A_SDWA) KERNEL(k_bfe, A_BFE) KERNEL(k_ffbl, A_FFBL) KERNEL(k_mix_salu, A_MIX_SALU) KERNEL(k_mix_salu2, A_MIX_SALU2)
KERNEL(k_max3, A_MAX3) KERNEL(k_med3, A_MED3) KERNEL(k_mbcnt, A_MBCNT) KERNEL(k_cndmask, A_CNDMASK)
KERNEL(k_cmp, A_CMP) KERNEL(k_cmp_sdwa, A_CMP_SDWA) KERNEL(k_readlane, A_READLANE)
KERNEL(k_or, A_OR) KERNEL(k_and, A_AND) KERNEL(k_lshl, A_LSHL) KERNEL(k_lshl_v, A_LSHL_V) KERNEL(k_mov, A_MOV) KERNEL(k_not, A_NOT)
KERNEL(k_and_sgpr, A_AND_SGPR) KERNEL(k_add_sgpr, A_ADD_SGPR) KERNEL(k_xor_inl, A_XOR_INL) KERNEL(k_sub_lit, A_SUB_LIT) KERNEL(k_subrev, A_SUBREV)
KERNEL(k_add_lit, A_ADD_LIT) KERNEL(k_bitop3_sgpr, A_BITOP3_SGPR) KERNEL(k_bitop3_or3, A_BITOP3_OR3) KERNEL(k_min, A_MIN) KERNEL(k_max, A_MAX)
KERNEL(k_addco, A_ADDCO) KERNEL(k_xnor, A_XNOR) KERNEL(k_ashr, A_ASHR) KERNEL(k_mix_2v_1s, A_MIX_2V_1S) KERNEL(k_pair_2_4, A_PAIR_2_4)
KERNEL(k_dep_xor, A_DEP_XOR) KERNEL(k_dep_perm, A_DEP_PERM) KERNEL(k_dep_bitop3, A_DEP_BITOP3)

typedef void (*kern_t)(uint32_t*, int, uint32_t, unsigned long long*);

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 2000;
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  uint32_t* out; unsigned long long* clk; CK(hipMalloc(&out, 64)); CK(hipMalloc(&clk, 64));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  printf("device %s, %d CUs, clockRate %d kHz\n", prop.gcnArchName, cus, prop.clockRate);
  struct { const char* name; kern_t k; } ks[] = {
    {"v_xor_b32", k_xor}, {"v_add_u32", k_add}, {"v_sub_u32", k_sub}, {"v_and_b32 literal", k_and_lit}, {"v_xor_b32 literal", k_xor_lit},
    {"v_xor_b32 sgpr", k_xor_sgpr}, {"v_lshrrev_b32", k_lshr}, {"v_lshl_or_b32", k_lshl_or}, {"v_lshl_add_u32", k_lshl_add}, {"v_or3_b32", k_or3},
    {"v_and_or_b32", k_and_or}, {"v_add3_u32", k_add3}, {"v_xad_u32", k_xad}, {"v_bitop3_b32", k_bitop3}, {"v_bfi_b32", k_bfi},
    {"v_perm_b32", k_perm}, {"v_dot4_u32_u8", k_dot4}, {"v_alignbit_b32", k_alignbit},
    {"v_bfe_u32", k_bfe}, {"v_bcnt_u32_b32", k_bcnt}, {"v_ffbl_b32", k_ffbl}, {"v_mbcnt_lo", k_mbcnt}, {"v_mul_u32_u24", k_mul24}, {"v_mad_u32_u24", k_mad24},
    {"v_mul_lo_u32", k_mullo}, {"v_sad_u8", k_sad}, {"v_msad_u8", k_msad}, {"v_max3_u32", k_max3}, {"v_med3_u32", k_med3},
    {"v_pk_add_u16", k_pkadd}, {"v_pk_min_u16", k_pkmin}, {"v_mov_b32 dpp row_shr", k_dpp}, {"v_xor_b32 dpp", k_xor_dpp}, {"v_or_b32 sdwa", k_sdwa},
    {"v_cmp_eq_u32", k_cmp}, {"v_cmp_eq_u32 sdwa", k_cmp_sdwa}, {"v_readlane_b32", k_readlane},
    {"v_fma_f32", k_fma},
    {"v_xor + v_perm [x2 instr]", k_pair_2_4},
    {"v_or_b32", k_or}, {"v_and_b32", k_and}, {"v_lshlrev_b32 imm", k_lshl}, {"v_lshlrev_b32 vgpr", k_lshl_v}, {"v_mov_b32", k_mov}, {"v_not_b32", k_not},
    {"v_and_b32 sgpr", k_and_sgpr}, {"v_add_u32 sgpr", k_add_sgpr}, {"v_xor_b32 inline const", k_xor_inl}, {"v_sub_u32 literal", k_sub_lit}, {"v_subrev_u32", k_subrev},
    {"v_add_u32 literal", k_add_lit}, {"v_bitop3 with sgpr", k_bitop3_sgpr}, {"v_bitop3 (or3)", k_bitop3_or3}, {"v_min_u32", k_min}, {"v_max_u32", k_max},
    {"v_add_co_u32", k_addco}, {"v_xnor_b32", k_xnor}, {"v_ashrrev_i32", k_ashr},
    {"DEPENDENT v_xor chain", k_dep_xor}, {"DEPENDENT v_perm chain", k_dep_perm}, {"DEPENDENT v_bitop3 chain", k_dep_bitop3},
    {"v_xor + 1 salu", k_mix_salu}, {"v_xor + 2 salu", k_mix_salu2}, {"2 valu(2cyc) + 1 salu [x3 instr]", k_mix_2v_1s},
  };
  for (int wps : {8, 4, 2, 1}) {  // wavefronts per SIMD
    printf("---- %d wavefront(s) per SIMD ----\n", wps); fflush(stdout);
    const int blocks = cus * wps;  // 256 threads = 4 wavefronts = one per SIMD
    for (auto& k : ks) {
      hipLaunchKernelGGL(k.k, dim3(blocks), dim3(256), 0, 0, out, iters / 10, 1u, clk);
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(k.k, dim3(blocks), dim3(256), 0, 0, out, iters, 1u, clk);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      CK(hipGetLastError());
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      unsigned long long c[2]; CK(hipMemcpy(c, clk, 16, hipMemcpyDeviceToHost));
      const double ghz = c[1] ? (double)c[0] / ((double)c[1] * 10.0) : 0.0;  // wall clock: 100 MHz
      const double inst = (double)iters * 32.0;
      // one wavefront's view: shader cycles it spent / its instructions, times 1 / wps = per-SIMD issue cost
      printf("%-26s %8.3f ms  shader clock %.2f GHz  %6.2f (one wavefront: cycles / asm line / waves per SIMD)  %6.2f cycles / asm line / SIMD (wall)\n", k.name, ms, ghz,
             (double)c[0] / inst / wps, ms * 1e-3 * ghz * 1e9 / inst / wps); fflush(stdout);
    }
  }
  return 0;
}
