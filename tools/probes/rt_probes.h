// rt_probes.h — timing probes of the tile kernels.  MEASUREMENT BUILDS ONLY: included by ray_tracer_s8_amd/csrc/rt_kernel.hip.h when a
// library VARIANT is compiled with  -DRT_PROBES -Itools/probes  plus one of the probe macros below (tools/abv.sh, RT_LIB_VARIANT):
// the product translation units never see this file.  Several probes render WRONG images by design: they price work, not pixels.
//
//   -DRT_PROBE_FAST_SQRT       every RT_SQRT as the bare v_sqrt_f32 (no IEEE rounding)           — wrong images
//   -DRT_PROBE_FAST_DIV        every RT_DIV as a * v_rcp_f32(b)                                  — wrong images
//   -DRT_PROBE_LT_LDS64=n      LDS-tree step: n (1..4) more single ds_read_b64 of the node's neighbourhood
//   -DRT_PROBE_LT_LDS=n        LDS-tree step: n more dword reads of the node, results discarded
//   -DRT_PROBE_LT_VALU=n       LDS-tree step: n more independent v_add_f32
//   -DRT_PROBE_LT_VSLOW=n      LDS-tree step: n more independent v_max_f32
//   -DRT_PROBE_EXTRA_GATHER=n  quantised walk: |n| more 16-byte gathers per node step (n > 0: the same node, n < 0: another node's line)
// Results: DESIGN.md 4.7 / 4.8, profiles/r02_c5_gather_probe.txt.
#pragma once

#ifdef RT_PROBE_FAST_SQRT
#undef RT_SQRT
#undef RT_SQRT_NEG_OK
#define RT_SQRT(x_) __builtin_amdgcn_sqrtf(x_)
#define RT_SQRT_NEG_OK(x_) __builtin_amdgcn_sqrtf(x_)
#endif
#ifdef RT_PROBE_FAST_DIV
#undef RT_DIV
#define RT_DIV(a_, b_) ((a_) * __builtin_amdgcn_rcpf(b_))
#endif

#if defined(RT_PROBE_LT_LDS64) || defined(RT_PROBE_LT_LDS)
#undef RT_HOOK_LT_STEP_LOADS
#ifndef RT_PROBE_LT_LDS64
#define RT_PROBE_LT_LDS64 0
#endif
#ifndef RT_PROBE_LT_LDS
#define RT_PROBE_LT_LDS 0
#endif
// (inline asm: the compiler would merge neighbouring reads into ds_read2_b64, which runs at half the rate); consumed at the end of
// the step, after the step's own waits
#define RT_HOOK_LT_STEP_LOADS(nd_, ni_, lnodes_)                                                                                  \
    typedef float f2_ __attribute__((ext_vector_type(2)));                                                                        \
    f2_ e0_ = {0.f, 0.f}, e1_ = {0.f, 0.f}, e2_ = {0.f, 0.f}, e3_ = {0.f, 0.f};                                                    \
    if (RT_PROBE_LT_LDS64 > 0) {                                                                                                  \
        const uint32_t a_ = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const float*)((lnodes_) + ((ni_) & ~1u));     \
        asm volatile("ds_read_b64 %0, %1" : "=v"(e0_) : "v"(a_));                                                                 \
        if (RT_PROBE_LT_LDS64 > 1) asm volatile("ds_read_b64 %0, %1 offset:8" : "=v"(e1_) : "v"(a_));                             \
        if (RT_PROBE_LT_LDS64 > 2) asm volatile("ds_read_b64 %0, %1 offset:16" : "=v"(e2_) : "v"(a_));                            \
        if (RT_PROBE_LT_LDS64 > 3) asm volatile("ds_read_b64 %0, %1 offset:24" : "=v"(e3_) : "v"(a_));                            \
    }                                                                                                                             \
    {                                                                                                                             \
        float e_[RT_PROBE_LT_LDS + 1];                                                                                            \
        _Pragma("unroll") for (int i = 0; i < RT_PROBE_LT_LDS; i++) e_[i] = (nd_)[(2 + 3 * i) % 18];                              \
        _Pragma("unroll") for (int i = 0; i < RT_PROBE_LT_LDS; i++) asm volatile("" ::"v"(e_[i]));                                \
    }
#undef RT_HOOK_LT_STEP_END
#define RT_HOOK_LT_STEP_END                                                                                                       \
    if (RT_PROBE_LT_LDS64 > 0) {                                                                                                  \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                                        \
        asm volatile("" ::"v"(e0_), "v"(e1_), "v"(e2_), "v"(e3_));                                                                \
    }
#endif

#if defined(RT_PROBE_LT_VALU) || defined(RT_PROBE_LT_VSLOW)
#undef RT_HOOK_LT_STEP_ALU
#ifndef RT_PROBE_LT_VALU
#define RT_PROBE_LT_VALU 0
#endif
#ifndef RT_PROBE_LT_VSLOW
#define RT_PROBE_LT_VSLOW 0
#endif
#define RT_HOOK_LT_STEP_ALU(p0_, p1_, p2_, p3_, oy_)                                                                              \
    {                                                                                                                             \
        float a_[4] = {p0_, p1_, p2_, p3_};                                                                                       \
        _Pragma("unroll") for (int i = 0; i < RT_PROBE_LT_VALU; i++) a_[i & 3] = a_[i & 3] + (oy_);                               \
        _Pragma("unroll") for (int i = 0; i < RT_PROBE_LT_VSLOW; i++) a_[i & 3] = __builtin_fmaxf(a_[i & 3], a_[(i + 1) & 3]);    \
        asm volatile("" ::"v"(a_[0]), "v"(a_[1]), "v"(a_[2]), "v"(a_[3]));                                                        \
    }
#endif

#ifdef RT_PROBE_EXTRA_GATHER
#undef RT_HOOK_Q_GATHER
#define RT_HOOK_Q_GATHER(t_ref_, cl_, travq_, n_internal_)                                                                        \
    {                                                                                                                             \
        uint32_t zero = 0;                                                                                                        \
        asm volatile("" : "+v"(zero));                                                                                            \
        for (int e = 0; e < (RT_PROBE_EXTRA_GATHER > 0 ? RT_PROBE_EXTRA_GATHER : -RT_PROBE_EXTRA_GATHER); e++) {                  \
            const uint32_t other = RT_PROBE_EXTRA_GATHER > 0 ? (t_ref_) : (((t_ref_) + 1u + e) * 2654435761u) % (n_internal_);    \
            const uint4* nx = reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(travq_) + (other << 5));               \
            asm volatile("" : "+v"(nx));                                                                                          \
            uint4 qx = nx[0];                                                                                                     \
            (cl_) ^= qx.x & zero;                                                                                                 \
        }                                                                                                                         \
    }
#endif
