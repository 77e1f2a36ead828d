// Issue cost per instruction class on gfx950 (cycles per wave-instruction per SIMD at the clock the chip reports),
// 8 independent instructions per loop body, 1 / 4 / 6 waves per SIMD.  Feeds the cost model of DESIGN.md 4.6:
// the path-trace kernel is half integer (xoshiro256++ in 64-bit halves) and half IEEE div / sqrt expansions.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int ITER = 2048;

#define R8(fmt)                                                                                         \
    asm volatile(fmt(0) fmt(1) fmt(2) fmt(3) fmt(4) fmt(5) fmt(6) fmt(7)                                   \
                 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7)          \
                 : "v"(a), "v"(b)                                                                          \
                 : "vcc", "s20", "s21")

#define I_XOR(n) "v_xor_b32 %" #n ", %" #n ", %8\n"
#define I_ADDU(n) "v_add_u32 %" #n ", %" #n ", %8\n"
#define I_ADDCO(n) "v_add_co_u32 %" #n ", vcc, %" #n ", %8\n"
#define I_ADDC(n) "v_addc_co_u32 %" #n ", vcc, %" #n ", %8, vcc\n"
#define I_ALIGN(n) "v_alignbit_b32 %" #n ", %" #n ", %8, 9\n"
#define I_LSHL(n) "v_lshlrev_b32 %" #n ", 7, %" #n "\n"
#define I_MULLO(n) "v_mul_lo_u32 %" #n ", %" #n ", %8\n"
#define I_MULHI(n) "v_mul_hi_u32 %" #n ", %" #n ", %8\n"
#define I_CNDM(n) "v_cndmask_b32 %" #n ", %" #n ", %8, vcc\n"
#define I_MINF(n) "v_min_f32 %" #n ", %" #n ", %8\n"
#define I_MAX3(n) "v_max3_f32 %" #n ", %" #n ", %8, %9\n"
#define I_CVTU(n) "v_cvt_f32_u32 %" #n ", %" #n "\n"
#define I_CVTB(n) "v_cvt_f32_ubyte1 %" #n ", %" #n "\n"
#define I_RCP(n) "v_rcp_f32 %" #n ", %" #n "\n"
#define I_SQRT(n) "v_sqrt_f32 %" #n ", %" #n "\n"
#define I_DSCALE(n) "v_div_scale_f32 %" #n ", vcc, %" #n ", %8, %9\n"
#define I_DFMAS(n) "v_div_fmas_f32 %" #n ", %" #n ", %8, %9\n"
#define I_DFIX(n) "v_div_fixup_f32 %" #n ", %" #n ", %8, %9\n"
#define I_CMPV(n) "v_cmp_lt_f32 vcc, %" #n ", %8\n"
#define I_CMPS(n) "v_cmp_lt_f32 s[20:21], %" #n ", %8\n"
#define I_FMA(n) "v_fma_f32 %" #n ", %" #n ", %8, %9\n"
#define I_MUL(n) "v_mul_f32 %" #n ", %" #n ", %8\n"
#define I_SUB(n) "v_sub_f32 %" #n ", %" #n ", %8\n"
#define I_MOV(n) "v_mov_b32 %" #n ", %8\n"
#define I_BPERM(n) "ds_bpermute_b32 %" #n ", %8, %" #n "\n"
#define I_AND(n) "v_and_b32 %" #n ", %" #n ", %8\n"
#define I_LSHL64(n) "v_lshlrev_b64 v[40:41], 17, v[40:41]\n"
#define I_CNDS(n) "v_cndmask_b32 %" #n ", %" #n ", %8, s[20:21]\n"
#define I_CND0(n) "v_cndmask_b32_e64 %" #n ", %8, %9, s[20:21]\n"
#define I_MED3(n) "v_med3_f32 %" #n ", %" #n ", %8, %9\n"
#define I_MIN3(n) "v_min3_f32 %" #n ", %" #n ", %8, %9\n"
#define I_MAXF(n) "v_max_f32 %" #n ", %" #n ", %8\n"
#define I_FMAMIX(n) "v_fma_mix_f32 %" #n ", %" #n ", %8, %9 op_sel_hi:[1,0,0]\n"
#define I_SDWA(n) "v_cvt_f32_u32_sdwa %" #n ", %" #n " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n"
#define I_PAIRV(n) "v_cmp_lt_f32 vcc, %" #n ", %8\n s_nop 1\n v_cndmask_b32 %" #n ", %" #n ", %9, vcc\n"
#define I_PAIRS(n) "v_cmp_lt_f32 s[20:21], %" #n ", %8\n s_nop 1\n v_cndmask_b32 %" #n ", %" #n ", %9, s[20:21]\n"
#define I_CNDV64(n) "v_cndmask_b32_e64 %" #n ", %" #n ", %8, vcc\n"
#define I_LSHLADD(n) "v_lshl_add_u32 %" #n ", %" #n ", 2, %8\n"
#define I_ADD3(n) "v_add3_u32 %" #n ", %" #n ", %8, %9\n"
#define I_MAD24(n) "v_mad_u32_u24 %" #n ", %" #n ", 4, %8\n"
#define I_MINU(n) "v_min_u32 %" #n ", %" #n ", %8\n"
#define I_LSHR(n) "v_lshrrev_b32 %" #n ", 16, %" #n "\n"
#define I_BFE(n) "v_bfe_u32 %" #n ", %" #n ", 3, 9\n"
#define I_ANDOR(n) "v_and_or_b32 %" #n ", %" #n ", %8, %9\n"
#define I_PKADD(n) "v_pk_add_f32 v[40:41], v[40:41], v[42:43]\n"
#define I_PKMUL(n) "v_pk_mul_f32 v[40:41], v[40:41], v[42:43]\n"
#define I_CMPLEU(n) "v_cmp_lt_u32 vcc, %" #n ", %8\n"
#define I_SUBREV(n) "v_subrev_f32 %" #n ", %8, %" #n "\n"
#define I_LSHLOR(n) "v_lshl_or_b32 %" #n ", %" #n ", 2, %8\n"
#define I_PERM(n) "v_perm_b32 %" #n ", %" #n ", %8, %9\n"
#define I_MAXI16(n) "v_pk_max_i16 %" #n ", %" #n ", %8\n"
#define I_SUBU(n) "v_sub_u32 %" #n ", %" #n ", %8\n"
#define I_MADF(n) "v_mad_f32 %" #n ", %" #n ", %8, %9\n"
#define I_MAD64(n) "v_mad_u64_u32 v[40:41], s[20:21], %" #n ", %8, v[40:41]\n"

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, float a, float b) {
    float x0 = threadIdx.x + 1.5f, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    asm volatile("s_mov_b32 s20, 0x55555555\n s_mov_b32 s21, 0x55555555\n s_mov_b32 vcc_lo, 0x33333333\n s_mov_b32 vcc_hi, 0x33333333" ::: "s20", "s21", "vcc");
    for (int i = 0; i < ITER; i++) {
        if (MODE == 0) R8(I_FMA);
        if (MODE == 1) R8(I_XOR);
        if (MODE == 2) R8(I_ADDU);
        if (MODE == 3) R8(I_ADDCO);
        if (MODE == 4) R8(I_ADDC);
        if (MODE == 5) R8(I_ALIGN);
        if (MODE == 6) R8(I_LSHL);
        if (MODE == 7) R8(I_MULLO);
        if (MODE == 8) R8(I_MULHI);
        if (MODE == 9) R8(I_CNDM);
        if (MODE == 10) R8(I_MINF);
        if (MODE == 11) R8(I_MAX3);
        if (MODE == 12) R8(I_CVTU);
        if (MODE == 13) R8(I_CVTB);
        if (MODE == 14) R8(I_RCP);
        if (MODE == 15) R8(I_SQRT);
        if (MODE == 16) R8(I_DSCALE);
        if (MODE == 17) R8(I_DFMAS);
        if (MODE == 18) R8(I_DFIX);
        if (MODE == 19) R8(I_CMPV);
        if (MODE == 20) R8(I_CMPS);
        if (MODE == 21) R8(I_MUL);
        if (MODE == 22) R8(I_SUB);
        if (MODE == 23) R8(I_MOV);
        if (MODE == 24) { R8(I_BPERM); asm volatile("s_waitcnt lgkmcnt(0)"); }
        if (MODE == 25) R8(I_AND);
        if (MODE == 26) asm volatile(I_LSHL64(0) I_LSHL64(0) I_LSHL64(0) I_LSHL64(0) I_LSHL64(0) I_LSHL64(0) I_LSHL64(0) I_LSHL64(0) ::: "v40", "v41");
        if (MODE == 27) R8(I_MAD64);
        if (MODE == 28) R8(I_CNDS);
        if (MODE == 29) R8(I_CND0);
        if (MODE == 30) R8(I_MED3);
        if (MODE == 31) R8(I_MIN3);
        if (MODE == 32) R8(I_MAXF);
        if (MODE == 33) R8(I_FMAMIX);
        if (MODE == 34) R8(I_SDWA);
        if (MODE == 35) R8(I_PAIRV);
        if (MODE == 36) R8(I_PAIRS);
        if (MODE == 37) R8(I_CNDV64);
        if (MODE == 38) R8(I_LSHLADD);
        if (MODE == 39) R8(I_ADD3);
        if (MODE == 40) R8(I_MAD24);
        if (MODE == 41) R8(I_MINU);
        if (MODE == 42) R8(I_LSHR);
        if (MODE == 43) R8(I_BFE);
        if (MODE == 44) R8(I_ANDOR);
        if (MODE == 45) asm volatile(I_PKADD(0) I_PKADD(0) I_PKADD(0) I_PKADD(0) I_PKADD(0) I_PKADD(0) I_PKADD(0) I_PKADD(0) ::: "v40", "v41", "v42", "v43");
        if (MODE == 46) asm volatile(I_PKMUL(0) I_PKMUL(0) I_PKMUL(0) I_PKMUL(0) I_PKMUL(0) I_PKMUL(0) I_PKMUL(0) I_PKMUL(0) ::: "v40", "v41", "v42", "v43");
        if (MODE == 47) R8(I_CMPLEU);
        if (MODE == 48) R8(I_SUBREV);
        if (MODE == 49) R8(I_LSHLOR);
        if (MODE == 50) R8(I_PERM);
        if (MODE == 51) R8(I_MAXI16);
        if (MODE == 52) R8(I_SUBU);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}

static double g_clock_hz = 2.4e9;

template <int MODE>
int run(const char* name, float* d_out) {
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    printf("%-22s", name);
    for (int occ : {1, 4, 6}) {
        int grid = 256 * occ;
        hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, d_out, 0.999f, 0.001f);
        CHK(hipDeviceSynchronize());
        CHK(hipEventRecord(e0));
        hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, d_out, 0.999f, 0.001f);
        CHK(hipEventRecord(e1));
        CHK(hipEventSynchronize(e1));
        float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
        double wave_instr_per_simd = (double)ITER * 8 * occ;
        printf("  occ%d %6.2f", occ, ms * 1e-3 * g_clock_hz / wave_instr_per_simd);
    }
    printf("   cycles/wave-instr/SIMD\n");
    return 0;
}

int main() {
    float* d_out; CHK(hipMalloc(&d_out, 256 * 8 * 256 * sizeof(float)));
    run<0>("v_fma_f32", d_out);
    run<21>("v_mul_f32", d_out);
    run<22>("v_sub_f32", d_out);
    run<23>("v_mov_b32", d_out);
    run<1>("v_xor_b32", d_out);
    run<25>("v_and_b32", d_out);
    run<2>("v_add_u32", d_out);
    run<3>("v_add_co_u32", d_out);
    run<4>("v_addc_co_u32", d_out);
    run<5>("v_alignbit_b32", d_out);
    run<6>("v_lshlrev_b32", d_out);
    run<26>("v_lshlrev_b64", d_out);
    run<7>("v_mul_lo_u32", d_out);
    run<8>("v_mul_hi_u32", d_out);
    run<27>("v_mad_u64_u32", d_out);
    run<9>("v_cndmask_b32", d_out);
    run<28>("v_cndmask sgpr mask", d_out);
    run<37>("v_cndmask_e64 vcc", d_out);
    run<35>("cmp+nop+cndmask vcc (x3 instr)", d_out);
    run<36>("cmp+nop+cndmask sgpr", d_out);
    run<29>("v_cndmask e64 2src", d_out);
    run<30>("v_med3_f32", d_out);
    run<31>("v_min3_f32", d_out);
    run<32>("v_max_f32", d_out);
    run<33>("v_fma_mix_f32 (f16 src0)", d_out);
    run<34>("v_cvt_f32_u32 sdwa", d_out);
    run<10>("v_min_f32", d_out);
    run<11>("v_max3_f32", d_out);
    run<12>("v_cvt_f32_u32", d_out);
    run<13>("v_cvt_f32_ubyte1", d_out);
    run<14>("v_rcp_f32", d_out);
    run<15>("v_sqrt_f32", d_out);
    run<16>("v_div_scale_f32", d_out);
    run<17>("v_div_fmas_f32", d_out);
    run<18>("v_div_fixup_f32", d_out);
    run<19>("v_cmp_lt_f32 vcc", d_out);
    run<20>("v_cmp_lt_f32 sgpr", d_out);
    run<24>("ds_bpermute_b32", d_out);
    // round 3: the rest of the LDS-tree step's instruction kinds, and what could stand in for them
    run<38>("v_lshl_add_u32", d_out);
    run<49>("v_lshl_or_b32", d_out);
    run<39>("v_add3_u32", d_out);
    run<40>("v_mad_u32_u24", d_out);
    run<41>("v_min_u32", d_out);
    run<52>("v_sub_u32", d_out);
    run<42>("v_lshrrev_b32", d_out);
    run<43>("v_bfe_u32", d_out);
    run<44>("v_and_or_b32", d_out);
    run<50>("v_perm_b32", d_out);
    run<51>("v_pk_max_i16", d_out);
    run<45>("v_pk_add_f32 (dep chain)", d_out);
    run<46>("v_pk_mul_f32 (dep chain)", d_out);
    run<47>("v_cmp_lt_u32 vcc", d_out);
    run<48>("v_subrev_f32", d_out);
    return 0;
}
