// v_pk_fma_f32 / v_fma_f32 issue cost vs VGPR bank placement of the operands (gfx950).
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int ITER = 20000;
#define REP8(s) s s s s s s s s
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out) {
    // explicit registers: v[40..79] scratch
    asm volatile("v_mov_b32 v40, 1.0\n v_mov_b32 v41, 1.0\n v_mov_b32 v42, 1.0\n v_mov_b32 v43, 1.0\n"
                 "v_mov_b32 v44, 0.5\n v_mov_b32 v45, 0.5\n v_mov_b32 v46, 0.5\n v_mov_b32 v47, 0.5\n"
                 "v_mov_b32 v48, 0\n v_mov_b32 v49, 0\n v_mov_b32 v50, 0\n v_mov_b32 v51, 0\n"
                 "v_mov_b32 v52, 0\n v_mov_b32 v53, 0\n v_mov_b32 v54, 0\n v_mov_b32 v55, 0\n"
                 "v_mov_b32 v56, 0\n v_mov_b32 v57, 0\n v_mov_b32 v58, 0\n v_mov_b32 v59, 0\n"
                 "v_mov_b32 v60, 0\n v_mov_b32 v61, 0\n v_mov_b32 v62, 0\n v_mov_b32 v63, 0\n"
                 ::: "v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63");
    for (int i = 0; i < ITER; i++) {
        if (MODE == 0)   // pk_fma, all three sources in banks {0,1}: a=v[40:41] b=v[44:45] c=dst (48,52,56,60 = bank 0)
            asm volatile("v_pk_fma_f32 v[48:49], v[40:41], v[44:45], v[48:49]\n v_pk_fma_f32 v[52:53], v[40:41], v[44:45], v[52:53]\n"
                         "v_pk_fma_f32 v[56:57], v[40:41], v[44:45], v[56:57]\n v_pk_fma_f32 v[60:61], v[40:41], v[44:45], v[60:61]\n"
                         "v_pk_fma_f32 v[48:49], v[40:41], v[44:45], v[48:49]\n v_pk_fma_f32 v[52:53], v[40:41], v[44:45], v[52:53]\n"
                         "v_pk_fma_f32 v[56:57], v[40:41], v[44:45], v[56:57]\n v_pk_fma_f32 v[60:61], v[40:41], v[44:45], v[60:61]\n" ::: "v48","v49","v52","v53","v56","v57","v60","v61");
        else if (MODE == 1)   // a in {0,1}, b in {2,3} (v[46:47]), c/dst in {0,1}
            asm volatile("v_pk_fma_f32 v[48:49], v[40:41], v[46:47], v[48:49]\n v_pk_fma_f32 v[52:53], v[40:41], v[46:47], v[52:53]\n"
                         "v_pk_fma_f32 v[56:57], v[40:41], v[46:47], v[56:57]\n v_pk_fma_f32 v[60:61], v[40:41], v[46:47], v[60:61]\n"
                         "v_pk_fma_f32 v[48:49], v[40:41], v[46:47], v[48:49]\n v_pk_fma_f32 v[52:53], v[40:41], v[46:47], v[52:53]\n"
                         "v_pk_fma_f32 v[56:57], v[40:41], v[46:47], v[56:57]\n v_pk_fma_f32 v[60:61], v[40:41], v[46:47], v[60:61]\n" ::: "v48","v49","v52","v53","v56","v57","v60","v61");
        else if (MODE == 2)   // a {0,1}, b {2,3}, c/dst in {2,3} (v[50:51] ...)
            asm volatile("v_pk_fma_f32 v[50:51], v[40:41], v[46:47], v[50:51]\n v_pk_fma_f32 v[54:55], v[40:41], v[46:47], v[54:55]\n"
                         "v_pk_fma_f32 v[58:59], v[40:41], v[46:47], v[58:59]\n v_pk_fma_f32 v[62:63], v[40:41], v[46:47], v[62:63]\n"
                         "v_pk_fma_f32 v[50:51], v[40:41], v[46:47], v[50:51]\n v_pk_fma_f32 v[54:55], v[40:41], v[46:47], v[54:55]\n"
                         "v_pk_fma_f32 v[58:59], v[40:41], v[46:47], v[58:59]\n v_pk_fma_f32 v[62:63], v[40:41], v[46:47], v[62:63]\n" ::: "v50","v51","v54","v55","v58","v59","v62","v63");
        else if (MODE == 3)   // op_sel broadcast of a single register for src0 (lo,lo): reads v40 twice
            asm volatile("v_pk_fma_f32 v[48:49], v[40:41], v[46:47], v[48:49] op_sel_hi:[0,1,1]\n v_pk_fma_f32 v[52:53], v[40:41], v[46:47], v[52:53] op_sel_hi:[0,1,1]\n"
                         "v_pk_fma_f32 v[56:57], v[40:41], v[46:47], v[56:57] op_sel_hi:[0,1,1]\n v_pk_fma_f32 v[60:61], v[40:41], v[46:47], v[60:61] op_sel_hi:[0,1,1]\n"
                         "v_pk_fma_f32 v[48:49], v[40:41], v[46:47], v[48:49] op_sel_hi:[0,1,1]\n v_pk_fma_f32 v[52:53], v[40:41], v[46:47], v[52:53] op_sel_hi:[0,1,1]\n"
                         "v_pk_fma_f32 v[56:57], v[40:41], v[46:47], v[56:57] op_sel_hi:[0,1,1]\n v_pk_fma_f32 v[60:61], v[40:41], v[46:47], v[60:61] op_sel_hi:[0,1,1]\n" ::: "v48","v49","v52","v53","v56","v57","v60","v61");
        else if (MODE == 4)   // dst different from sources (no accumulate): d=v[48..], a=v[40:41], b=v[46:47], c=v[42:43]
            asm volatile("v_pk_fma_f32 v[48:49], v[40:41], v[46:47], v[42:43]\n v_pk_fma_f32 v[52:53], v[40:41], v[46:47], v[42:43]\n"
                         "v_pk_fma_f32 v[56:57], v[40:41], v[46:47], v[42:43]\n v_pk_fma_f32 v[60:61], v[40:41], v[46:47], v[42:43]\n"
                         "v_pk_fma_f32 v[50:51], v[40:41], v[46:47], v[42:43]\n v_pk_fma_f32 v[54:55], v[40:41], v[46:47], v[42:43]\n"
                         "v_pk_fma_f32 v[58:59], v[40:41], v[46:47], v[42:43]\n v_pk_fma_f32 v[62:63], v[40:41], v[46:47], v[42:43]\n" ::: "v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63");
        else if (MODE == 5)   // scalar v_fma_f32, three sources in three different banks: a=v40(b0) b=v45(b1) c/dst=v50,v54,v58,v62 (b2)
            asm volatile("v_fma_f32 v50, v40, v45, v50\n v_fma_f32 v54, v40, v45, v54\n v_fma_f32 v58, v40, v45, v58\n v_fma_f32 v62, v40, v45, v62\n"
                         "v_fma_f32 v50, v40, v45, v50\n v_fma_f32 v54, v40, v45, v54\n v_fma_f32 v58, v40, v45, v58\n v_fma_f32 v62, v40, v45, v62\n" ::: "v50","v54","v58","v62");
        else if (MODE == 6)   // scalar v_fma_f32, all three sources in bank 0
            asm volatile("v_fma_f32 v48, v40, v44, v48\n v_fma_f32 v52, v40, v44, v52\n v_fma_f32 v56, v40, v44, v56\n v_fma_f32 v60, v40, v44, v60\n"
                         "v_fma_f32 v48, v40, v44, v48\n v_fma_f32 v52, v40, v44, v52\n v_fma_f32 v56, v40, v44, v56\n v_fma_f32 v60, v40, v44, v60\n" ::: "v48","v52","v56","v60");
        else if (MODE == 7)   // v_fmac_f32 (VOP2: dst += a*b), banks distinct
            asm volatile("v_fmac_f32 v50, v40, v45\n v_fmac_f32 v54, v40, v45\n v_fmac_f32 v58, v40, v45\n v_fmac_f32 v62, v40, v45\n"
                         "v_fmac_f32 v50, v40, v45\n v_fmac_f32 v54, v40, v45\n v_fmac_f32 v58, v40, v45\n v_fmac_f32 v62, v40, v45\n" ::: "v50","v54","v58","v62");
        else if (MODE == 8)   // pk_fma with SGPR pair as src1
            asm volatile("v_pk_fma_f32 v[48:49], v[40:41], s[20:21], v[48:49]\n v_pk_fma_f32 v[52:53], v[40:41], s[20:21], v[52:53]\n"
                         "v_pk_fma_f32 v[56:57], v[40:41], s[20:21], v[56:57]\n v_pk_fma_f32 v[60:61], v[40:41], s[20:21], v[60:61]\n"
                         "v_pk_fma_f32 v[50:51], v[40:41], s[20:21], v[50:51]\n v_pk_fma_f32 v[54:55], v[40:41], s[20:21], v[54:55]\n"
                         "v_pk_fma_f32 v[58:59], v[40:41], s[20:21], v[58:59]\n v_pk_fma_f32 v[62:63], v[40:41], s[20:21], v[62:63]\n" ::: "v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63","s20","s21");
    }
    float r;
    asm volatile("v_add_f32 %0, v48, v50\n v_add_f32 %0, %0, v52\n v_add_f32 %0, %0, v54\n v_add_f32 %0, %0, v56\n v_add_f32 %0, %0, v58\n v_add_f32 %0, %0, v60\n v_add_f32 %0, %0, v62" : "=v"(r));
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <int MODE>
int run(const char* name, float* d_out, int occ) {
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    int grid = 256 * occ;
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, d_out);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, d_out);
    CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    double n = (double)ITER * 8 * occ;
    printf("%-44s waves/SIMD=%d %.3f ms -> %.2f cyc/instr/SIMD @2.4GHz\n", name, occ, ms, ms * 1e-3 * 2.4e9 / n);
    return 0;
}
int main() {
    float* d; CHK(hipMalloc(&d, 256 * 8 * 256 * sizeof(float)));
    for (int occ : {1, 4, 8}) {
        run<0>("pk_fma a,b,c all banks{0,1}", d, occ);
        run<1>("pk_fma a{0,1} b{2,3} c{0,1}", d, occ);
        run<2>("pk_fma a{0,1} b{2,3} c{2,3}", d, occ);
        run<3>("pk_fma src0 op_sel broadcast", d, occ);
        run<4>("pk_fma dst!=src, a{0,1} b{2,3} c{2,3}", d, occ);
        run<5>("v_fma 3 distinct banks", d, occ);
        run<6>("v_fma all bank 0", d, occ);
        run<7>("v_fmac (VOP2) distinct banks", d, occ);
        run<8>("pk_fma sgpr-pair src1", d, occ);
    }
    return 0;
}
