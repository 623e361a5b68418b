// tools/ubench/valu_rates.hip — measures gfx950 VALU issue cost (cycles per wave64
// instruction per SIMD) for the integer ops Blake2b/Keccak/SHA-256 are made of.
// Method: each wave runs REPS × (UNROLL independent instructions over 8 register sets);
// s_memtime brackets the loop; cycles/instr = Δticks / instrs at W waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_rates valu_rates.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>


#define DEF_KERNEL(NAME, BODY)                                                                       \
    __global__ void NAME(unsigned long long* out, unsigned seed, int REPS) {                                   \
        unsigned a0 = seed + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11,       \
                 a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19;                                           \
        unsigned b0 = a0 ^ 0x55, b1 = a1 ^ 0x66, b2 = a2 ^ 0x77, b3 = a3 ^ 0x88, b4 = a4 ^ 0x99,     \
                 b5 = a5 ^ 0xaa, b6 = a6 ^ 0xbb, b7 = a7 ^ 0xcc;                                     \
        unsigned long long t0 = __builtin_readcyclecounter();                                        \
        unsigned long long m0 = __builtin_amdgcn_s_memtime();                                        \
        for (int r = 0; r < REPS; ++r) {                                                             \
            asm volatile(BODY                                                                        \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6),     \
                           "+v"(a7), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(b4), "+v"(b5),     \
                           "+v"(b6), "+v"(b7)::"vcc");                                               \
        }                                                                                            \
        unsigned long long m1 = __builtin_amdgcn_s_memtime();                                        \
        unsigned long long t1 = __builtin_readcyclecounter();                                        \
        unsigned x = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ b0 ^ b1 ^ b2 ^ b3 ^ b4 ^ b5 ^ b6 ^ b7;  \
        if (threadIdx.x == 0) {                                                                      \
            out[blockIdx.x * 4 + 0] = m1 - m0;                                                       \
            out[blockIdx.x * 4 + 1] = t1 - t0;                                                       \
        }                                                                                            \
        if (x == 0x12345678) out[blockIdx.x * 4 + 2] = x;                                            \
    }

// 16 independent instructions per asm block (8 register pairs twice)
#define R8(OP)                                                                    \
    OP("%0", "%8") OP("%1", "%9") OP("%2", "%10") OP("%3", "%11") OP("%4", "%12") \
        OP("%5", "%13") OP("%6", "%14") OP("%7", "%15")
#define XOR(a, b) "v_xor_b32 " a ", " a ", " b "\n"
#define ADD(a, b) "v_add_u32 " a ", " a ", " b "\n"
#define ALIGN(a, b) "v_alignbit_b32 " a ", " a ", " b ", 7\n"
#define PERM(a, b) "v_perm_b32 " a ", " a ", " b ", " b "\n"
#define ADD3(a, b) "v_add3_u32 " a ", " a ", " b ", " b "\n"
#define XAD(a, b) "v_xad_u32 " a ", " a ", " b ", " b "\n"
#define BFI(a, b) "v_bfi_b32 " a ", " a ", " b ", " b "\n"
#define ANDOR(a, b) "v_and_or_b32 " a ", " a ", " b ", " b "\n"
#define ADDCO(a, b) "v_add_co_u32 " a ", vcc, " a ", " b "\n"
#define FMA(a, b) "v_fma_f32 " a ", " a ", " b ", " b "\n"
#define MOV(a, b) "v_mov_b32 " a ", " b "\n"
#define PKADD(a, b) "v_pk_add_u16 " a ", " a ", " b "\n"
#define LSHL(a, b) "v_lshlrev_b32 " a ", 3, " b "\n"
#define MULLO(a, b) "v_mul_lo_u32 " a ", " a ", " b "\n"

#define XORSDWA(a, b) "v_xor_b32_sdwa " a ", " a ", " b " dst_sel:WORD_0 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:WORD_1\n"
#define XORSDWAB(a, b) "v_xor_b32_sdwa " a ", " a ", " b " dst_sel:BYTE_0 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_3 src1_sel:BYTE_3\n"
#define MOVSDWA(a, b) "v_mov_b32_sdwa " a ", " b " dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0\n"
#define XORDPP(a, b) "v_xor_b32_dpp " a ", " a ", " b " quad_perm:[1,2,3,0] row_mask:0xf bank_mask:0xf\n"
#define ADDDPP(a, b) "v_add_u32_dpp " a ", " a ", " b " quad_perm:[1,2,3,0] row_mask:0xf bank_mask:0xf\n"
#define AND(a, b) "v_and_b32 " a ", " a ", " b "\n"
#define OR(a, b) "v_or_b32 " a ", " a ", " b "\n"
#define NOT(a, b) "v_not_b32 " a ", " b "\n"
#define SUB(a, b) "v_sub_u32 " a ", " a ", " b "\n"
#define LSHLV(a, b) "v_lshlrev_b32 " a ", " b ", " a "\n"
#define LSHR(a, b) "v_lshrrev_b32 " a ", 3, " b "\n"
#define LSHLOR(a, b) "v_lshl_or_b32 " a ", " a ", 3, " b "\n"
#define LSHLADD(a, b) "v_lshl_add_u32 " a ", " a ", 3, " b "\n"
#define OR3(a, b) "v_or3_b32 " a ", " a ", " b ", " b "\n"
#define BFE(a, b) "v_bfe_u32 " a ", " b ", 3, 8\n"
#define CNDMASK(a, b) "v_cndmask_b32 " a ", " a ", " b ", vcc\n"
#define ALIGNBYTE(a, b) "v_alignbyte_b32 " a ", " a ", " b ", 3\n"
#define XORE64(a, b) "v_xor_b32_e64 " a ", " a ", " b "\n"
#define MAD24(a, b) "v_mad_u32_u24 " a ", " a ", " b ", " b "\n"
#define MUL24(a, b) "v_mul_u32_u24 " a ", " a ", " b "\n"
#define ADDC(a, b) "v_addc_co_u32 " a ", vcc, " a ", " b ", vcc\n"
DEF_KERNEL(k_xorsdwa, R8(XORSDWA) R8(XORSDWA))
DEF_KERNEL(k_xorsdwab, R8(XORSDWAB) R8(XORSDWAB))
DEF_KERNEL(k_movsdwa, R8(MOVSDWA) R8(MOVSDWA))
DEF_KERNEL(k_xordpp, R8(XORDPP) R8(XORDPP))
DEF_KERNEL(k_adddpp, R8(ADDDPP) R8(ADDDPP))
DEF_KERNEL(k_and, R8(AND) R8(AND))
DEF_KERNEL(k_or, R8(OR) R8(OR))
DEF_KERNEL(k_not, R8(NOT) R8(NOT))
DEF_KERNEL(k_sub, R8(SUB) R8(SUB))
DEF_KERNEL(k_lshlv, R8(LSHLV) R8(LSHLV))
DEF_KERNEL(k_lshr, R8(LSHR) R8(LSHR))
DEF_KERNEL(k_lshlor, R8(LSHLOR) R8(LSHLOR))
DEF_KERNEL(k_lshladd, R8(LSHLADD) R8(LSHLADD))
DEF_KERNEL(k_or3, R8(OR3) R8(OR3))
DEF_KERNEL(k_bfe, R8(BFE) R8(BFE))
DEF_KERNEL(k_cndmask, R8(CNDMASK) R8(CNDMASK))
DEF_KERNEL(k_alignbyte, R8(ALIGNBYTE) R8(ALIGNBYTE))
DEF_KERNEL(k_xore64, R8(XORE64) R8(XORE64))
DEF_KERNEL(k_mad24, R8(MAD24) R8(MAD24))
DEF_KERNEL(k_mul24, R8(MUL24) R8(MUL24))
DEF_KERNEL(k_addc, R8(ADDC) R8(ADDC))
DEF_KERNEL(k_xor, R8(XOR) R8(XOR))
DEF_KERNEL(k_add, R8(ADD) R8(ADD))
DEF_KERNEL(k_align, R8(ALIGN) R8(ALIGN))
DEF_KERNEL(k_perm, R8(PERM) R8(PERM))
DEF_KERNEL(k_add3, R8(ADD3) R8(ADD3))
DEF_KERNEL(k_xad, R8(XAD) R8(XAD))
DEF_KERNEL(k_bfi, R8(BFI) R8(BFI))
DEF_KERNEL(k_andor, R8(ANDOR) R8(ANDOR))
DEF_KERNEL(k_addco, R8(ADDCO) R8(ADDCO))
DEF_KERNEL(k_fma, R8(FMA) R8(FMA))
DEF_KERNEL(k_mov, R8(MOV) R8(MOV))
DEF_KERNEL(k_pkadd, R8(PKADD) R8(PKADD))
DEF_KERNEL(k_lshl, R8(LSHL) R8(LSHL))
DEF_KERNEL(k_mullo, R8(MULLO) R8(MULLO))

// 64-bit forms on register PAIRS: 8 independent instructions per block
#define DEF_KERNEL64(NAME, BODY)                                                                     \
    __global__ void NAME(unsigned long long* out, unsigned seed, int REPS) {                                   \
        unsigned long long a0 = seed + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7,           \
                           a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19;                   \
        unsigned long long m0 = __builtin_amdgcn_s_memtime();                                        \
        for (int r = 0; r < REPS; ++r) {                                                             \
            asm volatile(BODY BODY                                                                   \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6),     \
                           "+v"(a7));                                                                \
        }                                                                                            \
        unsigned long long m1 = __builtin_amdgcn_s_memtime();                                        \
        unsigned long long x = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;                                \
        if (threadIdx.x == 0) out[blockIdx.x * 4 + 0] = m1 - m0;                                     \
        if (x == 0x12345678) out[blockIdx.x * 4 + 2] = x;                                            \
    }
#define P8(OP) OP("%0", "%1") OP("%1", "%2") OP("%2", "%3") OP("%3", "%4") OP("%4", "%5") OP("%5", "%6") OP("%6", "%7") OP("%7", "%0")
#define LADD64(a, b) "v_lshl_add_u64 " a ", " a ", 0, " b "\n"
#define LADD64S(a, b) "v_lshl_add_u64 " a ", " a ", 1, " b "\n"
#define SHR64(a, b) "v_lshrrev_b64 " a ", 7, " b "\n"
#define SHL64(a, b) "v_lshlrev_b64 " a ", 7, " b "\n"
#define MOV64(a, b) "v_mov_b64 " a ", " b "\n"
#define PKFMA(a, b) "v_pk_fma_f32 " a ", " a ", " b ", " b "\n"
#define PKADDF(a, b) "v_pk_add_f32 " a ", " a ", " b "\n"
#define PKMOV(a, b) "v_pk_mov_b32 " a ", " a ", " b "\n"
DEF_KERNEL64(k_ladd64, P8(LADD64))
DEF_KERNEL64(k_ladd64s, P8(LADD64S))
DEF_KERNEL64(k_shr64, P8(SHR64))
DEF_KERNEL64(k_shl64, P8(SHL64))
DEF_KERNEL64(k_mov64, P8(MOV64))
DEF_KERNEL64(k_pkfma, P8(PKFMA))
DEF_KERNEL64(k_pkaddf, P8(PKADDF))
DEF_KERNEL64(k_pkmov, P8(PKMOV))

template <typename K>
double time_us(K kern, unsigned long long* out, int blocks, int threads, int reps) {
    double best = 1e30;
    for (int it = 0; it < 5; ++it) {
        (void)hipDeviceSynchronize();
        auto t0 = std::chrono::high_resolution_clock::now();
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, out, 2u, reps);
        (void)hipDeviceSynchronize();
        auto t1 = std::chrono::high_resolution_clock::now();
        double us = std::chrono::duration<double, std::micro>(t1 - t0).count();
        if (us < best) best = us;
    }
    return best;
}

template <typename K>
void run(const char* name, K kern, int instr_per_rep, int waves_per_simd) {
    int blocks = 256, threads = 64 * 4 * waves_per_simd;
    unsigned long long* out;
    (void)hipMalloc(&out, blocks * 4 * sizeof(unsigned long long));
    const int r1 = 4000, r2 = 20000;
    double t1 = time_us(kern, out, blocks, threads, r1), t2 = time_us(kern, out, blocks, threads, r2);
    double ns = (t2 - t1) * 1e3 / (double(r2 - r1) * instr_per_rep * waves_per_simd);
    printf("%-14s waves/SIMD=%d  %.4f ns per instr-slot  = %.2f cyc @2.4GHz  (%.2f @2.0GHz)\n", name, waves_per_simd, ns,
           ns * 2.4, ns * 2.0);
    (void)hipFree(out);
}

template <typename K>
void run_old(const char* name, K kern, int instr_per_rep, int waves_per_simd) {
    const int REPS = 2000;
    int dev_cus = 256;
    int blocks = dev_cus;  // one workgroup per CU
    int threads = 64 * 4 * waves_per_simd;
    unsigned long long* out;
    hipMalloc(&out, blocks * 4 * sizeof(unsigned long long));
    hipMemset(out, 0, blocks * 4 * sizeof(unsigned long long));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, out, 1u, REPS);
    hipDeviceSynchronize();
    auto t0 = std::chrono::high_resolution_clock::now();
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, out, 2u, REPS);
    hipDeviceSynchronize();
    auto t1 = std::chrono::high_resolution_clock::now();
    std::vector<unsigned long long> h(blocks * 4);
    hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost);
    double ticks = 0;
    for (int b = 0; b < blocks; ++b) ticks += double(h[b * 4]);
    ticks /= blocks;
    double wall_us = std::chrono::duration<double, std::micro>(t1 - t0).count();
    double instrs = double(REPS) * instr_per_rep;  // per wave
    // memtime ticks at 100 MHz on gfx9 (constant clock); report wall-based cycles too
    double wall_cycles_at_2p4 = wall_us * 2400.0;
    printf("%-10s waves/SIMD=%d  memtime ticks/instr/wave=%.4f  wall=%.1f us  => wall ns per (instr x waves/SIMD)=%.4f  "
           "[= %.2f cyc @2.4GHz]\n",
           name, waves_per_simd, ticks / instrs, wall_us, wall_us * 1e3 / (instrs * waves_per_simd),
           wall_cycles_at_2p4 / (instrs * waves_per_simd));
    hipFree(out);
}

int main() {
    for (int w : {1, 4}) {
        run("xor_b32", k_xor, 16, w);
        run("add_u32", k_add, 16, w);
        run("alignbit", k_align, 16, w);
        run("perm_b32", k_perm, 16, w);
        run("add3_u32", k_add3, 16, w);
        run("xad_u32", k_xad, 16, w);
        run("bfi_b32", k_bfi, 16, w);
        run("and_or", k_andor, 16, w);
        run("add_co", k_addco, 16, w);
        run("fma_f32", k_fma, 16, w);
        run("mov_b32", k_mov, 16, w);
        run("pk_add_u16", k_pkadd, 16, w);
        run("lshl_b32", k_lshl, 16, w);
        run("mul_lo_u32", k_mullo, 16, w);
        run("xor_sdwa_w", k_xorsdwa, 16, w);
        run("xor_sdwa_b", k_xorsdwab, 16, w);
        run("mov_sdwa", k_movsdwa, 16, w);
        run("xor_dpp", k_xordpp, 16, w);
        run("add_dpp", k_adddpp, 16, w);
        run("and_b32", k_and, 16, w);
        run("or_b32", k_or, 16, w);
        run("not_b32", k_not, 16, w);
        run("sub_u32", k_sub, 16, w);
        run("lshlrev_vv", k_lshlv, 16, w);
        run("lshr_b32", k_lshr, 16, w);
        run("lshl_or", k_lshlor, 16, w);
        run("lshl_add_u32", k_lshladd, 16, w);
        run("or3", k_or3, 16, w);
        run("bfe_u32", k_bfe, 16, w);
        run("cndmask", k_cndmask, 16, w);
        run("alignbyte", k_alignbyte, 16, w);
        run("xor_e64", k_xore64, 16, w);
        run("mad_u32_u24", k_mad24, 16, w);
        run("mul_u32_u24", k_mul24, 16, w);
        run("addc_co", k_addc, 16, w);
        run("lshl_add64", k_ladd64, 16, w);
        run("lshl1add64", k_ladd64s, 16, w);
        run("lshr_b64", k_shr64, 16, w);
        run("lshl_b64", k_shl64, 16, w);
        run("mov_b64", k_mov64, 16, w);
        run("pk_fma_f32", k_pkfma, 16, w);
        run("pk_add_f32", k_pkaddf, 16, w);
        run("pk_mov_b32", k_pkmov, 16, w);
        printf("\n");
    }
    return 0;
}
