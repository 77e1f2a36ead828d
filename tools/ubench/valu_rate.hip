// VALU issue-rate microbenchmark for gfx950: cycles per wave-instruction per SIMD for
// v_fma_f32, v_pk_fma_f32, v_sub_f32, v_cmp + and an LDS-broadcast / scalar-load fed loop.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int ITER = 4096;

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, float a, float b) {
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7}, p4 = {x1, x0}, p5 = {x3, x2}, p6 = {x5, x4}, p7 = {x7, x6};
    f2 pa = {a, a}, pb = {b, b};
    for (int i = 0; i < ITER; i++) {
        if (MODE == 0) {   // 8 independent v_fma_f32
            asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                         "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
        } else if (MODE == 1) {   // 8 independent v_pk_fma_f32
            asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
                         "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pa), "v"(pb));
        } else if (MODE == 2) {   // 8 v_sub_f32 (VOP2)
            asm volatile("v_sub_f32 %0, %0, %8\n v_sub_f32 %1, %1, %8\n v_sub_f32 %2, %2, %8\n v_sub_f32 %3, %3, %8\n"
                         "v_sub_f32 %4, %4, %8\n v_sub_f32 %5, %5, %8\n v_sub_f32 %6, %6, %8\n v_sub_f32 %7, %7, %8\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a));
        } else if (MODE == 3) {   // 8 v_pk_add_f32
            asm volatile("v_pk_add_f32 %0, %0, %8\n v_pk_add_f32 %1, %1, %8\n v_pk_add_f32 %2, %2, %8\n v_pk_add_f32 %3, %3, %8\n"
                         "v_pk_add_f32 %4, %4, %8\n v_pk_add_f32 %5, %5, %8\n v_pk_add_f32 %6, %6, %8\n v_pk_add_f32 %7, %7, %8\n"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pa));
        } else if (MODE == 4) {   // 8 v_cmp_lt_f32 to sgpr pairs (VOP3)
            asm volatile("v_cmp_lt_f32 s[20:21], %0, %8\n v_cmp_lt_f32 s[22:23], %1, %8\n v_cmp_lt_f32 s[24:25], %2, %8\n v_cmp_lt_f32 s[26:27], %3, %8\n"
                         "v_cmp_lt_f32 s[28:29], %4, %8\n v_cmp_lt_f32 s[30:31], %5, %8\n v_cmp_lt_f32 s[32:33], %6, %8\n v_cmp_lt_f32 s[34:35], %7, %8\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a)
                         : "s20","s21","s22","s23","s24","s25","s26","s27","s28","s29","s30","s31","s32","s33","s34","s35");
        } else if (MODE == 5) {   // 8 v_fma_f32 with an SGPR operand
            asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                         "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "s"(a), "v"(b));
        } else if (MODE == 6) {   // 8 v_mul_f32 (VOP2)
            asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                         "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a));
        } else if (MODE == 7) {   // 8 v_pk_mul_f32
            asm volatile("v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n"
                         "v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8\n"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pa));
        } else if (MODE == 8) {   // dependent chain of v_fma_f32 (latency)
            asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                         "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                         : "+v"(x0) : "v"(a), "v"(b));
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + p0.x + p1.x + p2.x + p3.x + p4.y + p5.y + p6.y + p7.y;
}

template <int MODE>
int run(const char* name, float* d_out, int blocks_per_cu) {
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    int grid = 256 * blocks_per_cu;
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, d_out, 0.999f, 0.001f);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, d_out, 0.999f, 0.001f);
    CHK(hipEventRecord(e1));
    CHK(hipEventSynchronize(e1));
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    // waves per SIMD = blocks_per_cu (256 thr = 4 waves = 1 per SIMD)
    double wave_instr_per_simd = (double)ITER * 8 * blocks_per_cu;
    double cycles = ms * 1e-3 * 2.4e9;
    printf("%-28s waves/SIMD=%d  %.3f ms  -> %.2f cycles/wave-instr/SIMD (at 2.4GHz)\n", name, blocks_per_cu, ms, cycles / wave_instr_per_simd);
    return 0;
}

int main() {
    float* d_out; CHK(hipMalloc(&d_out, 256 * 8 * 256 * sizeof(float)));
    for (int occ : {1, 2, 4, 8}) {
        run<0>("v_fma_f32", d_out, occ);
        run<1>("v_pk_fma_f32", d_out, occ);
        run<2>("v_sub_f32", d_out, occ);
        run<3>("v_pk_add_f32", d_out, occ);
        run<4>("v_cmp_lt_f32->sgpr", d_out, occ);
        run<5>("v_fma_f32 sgpr operand", d_out, occ);
        run<6>("v_mul_f32", d_out, occ);
        run<7>("v_pk_mul_f32", d_out, occ);
        run<8>("v_fma_f32 dependent", d_out, occ);
    }
    return 0;
}
