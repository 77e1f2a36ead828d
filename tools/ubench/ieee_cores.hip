// Correct-rounding cores of sqrtf and the f32 quotient without the compiler's scaling / fix-up wrappers, checked on the device against
// the full IEEE sequences (__builtin_sqrtf, operator/ with -ffp-contract=off): every f32 input for the square roots, 2^34 operand pairs
// with both magnitudes in [2^-47, 2^47] (and every sign combination) for the quotient.  Prints where a core's domain ends.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o tools/ubench/ieee_cores tools/ubench/ieee_cores.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ float sqrt_core_a(float x) {          // v_sqrt_f32 and the two one-ulp residual tests (the compiler's core)
    float s = __builtin_amdgcn_sqrtf(x);
    const float sd = __uint_as_float(__float_as_uint(s) - 1u), su = __uint_as_float(__float_as_uint(s) + 1u);
    const float r1 = __builtin_fmaf(-sd, s, x), r2 = __builtin_fmaf(-su, s, x);
    s = (r1 <= 0.0f) ? sd : s;
    s = (r2 > 0.0f) ? su : s;
    return s;
}
__device__ __forceinline__ float sqrt_core_b(float x) {          // v_rsq_f32 and one coupled Newton step (seven fast-class instructions)
    const float r = __builtin_amdgcn_rsqf(x);
    float g = x * r, h = 0.5f * r;
    const float e = __builtin_fmaf(-h, g, 0.5f);
    g = __builtin_fmaf(g, e, g);
    h = __builtin_fmaf(h, e, h);
    const float d = __builtin_fmaf(-g, g, x);
    return __builtin_fmaf(d, h, g);
}
__device__ __forceinline__ float div_core(float a, float b) {    // the compiler's sequence between v_div_scale and v_div_fmas / v_div_fixup
    float r = __builtin_amdgcn_rcpf(b);
    const float e0 = __builtin_fmaf(-b, r, 1.0f);
    r = __builtin_fmaf(e0, r, r);
    float q = a * r;
    const float e1 = __builtin_fmaf(-b, q, a);
    q = __builtin_fmaf(e1, r, q);
    const float e2 = __builtin_fmaf(-b, q, a);
    return __builtin_fmaf(e2, r, q);
}

struct Res { unsigned long long bad_a, bad_b; uint32_t lo_a, hi_a, lo_b, hi_b; };

__global__ void k_sqrt(Res* res, uint32_t from, uint32_t to) {      // bit patterns [from, to)
    unsigned long long n = (unsigned long long)(to - from);
    for (unsigned long long i = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * blockDim.x) {
        const uint32_t u = from + (uint32_t)i;
        const float x = __uint_as_float(u);
        float xv = x;
        asm volatile("" : "+v"(xv));
        const float ref = __builtin_sqrtf(xv);
        const float a = sqrt_core_a(xv), b = sqrt_core_b(xv);
        if (__float_as_uint(a) != __float_as_uint(ref)) { atomicAdd(&res->bad_a, 1ull); atomicMin(&res->lo_a, u); atomicMax(&res->hi_a, u); }
        if (__float_as_uint(b) != __float_as_uint(ref)) { atomicAdd(&res->bad_b, 1ull); atomicMin(&res->lo_b, u); atomicMax(&res->hi_b, u); }
    }
}

__device__ __forceinline__ uint64_t mix(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull; z = (z ^ (z >> 27)) * 0x94d049bb133111ebull; return z ^ (z >> 31);
}
// operand magnitudes 2^elo .. 2^ehi, any mantissa, any sign
__global__ void k_div(Res* res, unsigned long long n, int elo, int ehi, uint64_t seed) {
    for (unsigned long long i = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * blockDim.x) {
        const uint64_t r0 = mix(seed + 2 * i), r1 = mix(seed + 2 * i + 1);
        const uint32_t span = (uint32_t)(ehi - elo + 1);
        const uint32_t ea = (uint32_t)(127 + elo) + (uint32_t)((r0 >> 40) % span), eb = (uint32_t)(127 + elo) + (uint32_t)((r1 >> 40) % span);
        float a = __uint_as_float(((uint32_t)r0 & 0x807fffffu) | (ea << 23)), b = __uint_as_float(((uint32_t)r1 & 0x807fffffu) | (eb << 23));
        if (i & 1) b = __uint_as_float((__float_as_uint(b) & 0xff800000u) | ((uint32_t)(r1 >> 32) & 0x7u));      // divisors next to a power of two as well
        asm volatile("" : "+v"(a), "+v"(b));
        const float ref = a / b;
        const float q = div_core(a, b);
        if (__float_as_uint(q) != __float_as_uint(ref)) { atomicAdd(&res->bad_a, 1ull); atomicMin(&res->lo_a, __float_as_uint(a) & 0x7fffffffu); atomicMax(&res->hi_a, __float_as_uint(a) & 0x7fffffffu); }
    }
}

int main() {
    Res* d; CHK(hipMalloc(&d, sizeof(Res)));
    auto reset = [&]() { Res h = {0, 0, 0xffffffffu, 0, 0xffffffffu, 0}; return hipMemcpy(d, &h, sizeof(h), hipMemcpyHostToDevice); };
    Res h;
    // square roots: all positive finite inputs, in exponent bands so that the end of each core's domain shows
    const uint32_t bands[][2] = {{0x00000001u, 0x00800000u}, {0x00800000u, 0x0f800000u}, {0x0f800000u, 0x10000000u}, {0x10000000u, 0x7f000000u}, {0x7f000000u, 0x7f800000u}};
    const char* names[] = {"denormals", "2^-126 .. 2^-96", "2^-96 .. 2^-95", "2^-95 .. 2^127", "2^127 .. inf"};
    for (int b = 0; b < 5; b++) {
        CHK(reset());
        hipLaunchKernelGGL(k_sqrt, dim3(4096), dim3(256), 0, 0, d, bands[b][0], bands[b][1]);
        CHK(hipDeviceSynchronize());
        CHK(hipMemcpy(&h, d, sizeof(h), hipMemcpyDeviceToHost));
        printf("sqrt %-18s inputs %10u   core A (v_sqrt + 2 residual tests) mismatches %10llu [%08x..%08x]   core B (v_rsq + Newton) mismatches %10llu [%08x..%08x]\n",
               names[b], bands[b][1] - bands[b][0], h.bad_a, h.lo_a, h.hi_a, h.bad_b, h.lo_b, h.hi_b);
    }
    const int ranges[][2] = {{-47, 47}, {-30, 30}, {-62, 62}};
    for (int r = 0; r < 3; r++) {
        CHK(reset());
        const unsigned long long n = r == 0 ? (1ull << 34) : (1ull << 32);
        hipLaunchKernelGGL(k_div, dim3(8192), dim3(256), 0, 0, d, n, ranges[r][0], ranges[r][1], 0x5eed0000ull + r);
        CHK(hipDeviceSynchronize());
        CHK(hipMemcpy(&h, d, sizeof(h), hipMemcpyDeviceToHost));
        printf("div  |a|, |b| in 2^%d .. 2^%d: %llu pairs, core mismatches %llu\n", ranges[r][0], ranges[r][1] + 1, n, h.bad_a);
    }
    return 0;
}
