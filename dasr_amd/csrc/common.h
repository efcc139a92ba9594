// Shared device helpers for the gfx950 kernels (CDNA4, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include "../../include/dasr_hip.h"

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef _Float16 f16_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

#define DASR_LDS __attribute__((address_space(3)))

#define HIP_TRY(expr)                                   \
    do {                                                \
        hipError_t _e = (expr);                         \
        if (_e != hipSuccess) return (int)_e;           \
    } while (0)

__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return (float)v; }

// round-to-nearest-even fp32 -> bf16 (v_cvt_pk_bf16_f32 on gfx950)
__device__ __forceinline__ bf16_t f32_to_bf16(float v) { return (bf16_t)v; }

__device__ __forceinline__ void split_bf16(float v, bf16_t& hi, bf16_t& lo) {
    hi = (bf16_t)v;
    lo = (bf16_t)(v - (float)hi);
}

// f16 hi + lo pair of a (pre-scaled) fp32 value: 22 mantissa bits where the bf16 pair has 16 (values below 2^-14 fall into f16's subnormal
// range: absolute error 2^-25; callers pre-scale tiny gradients by a power of two)
__device__ __forceinline__ void split_f16(float v, bf16_t& hi, bf16_t& lo) {
    const f16_t h = (f16_t)v;
    hi = __builtin_bit_cast(bf16_t, h);
    lo = __builtin_bit_cast(bf16_t, (f16_t)(v - (float)h));
}

// one MFMA k-step (32 x 32 x 16) on 16-bit operand fragments held as bf16x8 bit patterns: F16 selects the f16 instruction
template <bool F16>
__device__ __forceinline__ f32x16 mfma16(bf16x8 a, bf16x8 b, f32x16 c) {
    if constexpr (F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

static inline hipStream_t as_stream(void* s) { return (hipStream_t)s; }

// Every kernel launch of the library goes through DASR_LAUNCH.  While a profiling session is open (dasr_prof_begin, used by
// bench.py for the `roofline` block) the launch carries its own start/stop events (hipExtLaunchKernelGGL: the dispatch's own
// begin/end timestamps, the same ones rocprofv3 --kernel-trace reports); otherwise it is a plain launch.
bool dasr_prof_slot(const char* tag, hipEvent_t* e0, hipEvent_t* e1);  // misc.hip
#define DASR_LAUNCH_TAG(tag, kfn, grid, block, lds, s, ...)                                         \
    do {                                                                                            \
        hipEvent_t _e0 = nullptr, _e1 = nullptr;                                                    \
        if (dasr_prof_slot(tag, &_e0, &_e1)) hipExtLaunchKernelGGL(kfn, grid, block, lds, s, _e0, _e1, 0, __VA_ARGS__); \
        else hipLaunchKernelGGL(kfn, grid, block, lds, s, __VA_ARGS__);                             \
    } while (0)
#define DASR_LAUNCH(kfn, ...) DASR_LAUNCH_TAG(#kfn, kfn, __VA_ARGS__)
