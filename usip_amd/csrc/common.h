// usip_amd/csrc/common.h -- shared device/host helpers for libusip_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/usip_hip.h"

#define USIP_WAVE 64

// hipGetLastError() is sticky per thread: clear whatever an unrelated earlier runtime call left
// behind, launch, then USIP_LAUNCH_CHECK() reports this launch only.
#define USIP_LAUNCH(...)                                      \
    do {                                                      \
        (void)hipGetLastError();                              \
        hipLaunchKernelGGL(__VA_ARGS__);                      \
    } while (0)

#define USIP_LAUNCH_CHECK()                                   \
    do {                                                      \
        hipError_t e__ = hipGetLastError();                   \
        if (e__ != hipSuccess) return (int)e__;               \
    } while (0)

static inline int usip_ceil_div(long long a, long long b) { return (int)((a + b - 1) / b); }

// Number of lanes below `lane` whose bit is set in a 64-bit wave mask.
__device__ __forceinline__ int usip_mbcnt(unsigned long long mask)
{
    return __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32),
                                     __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0));
}

// The distance the whole path uses, in the arithmetic order of the pinned oracle platform
// (torch CPU norm: FMA chain over channels 0,1,2, then a correctly rounded sqrt).
__device__ __forceinline__ float usip_sqdist(float ax, float ay, float az, float bx, float by, float bz)
{
    float dx = ax - bx, dy = ay - by, dz = az - bz;
    float s = dx * dx;                 // -ffp-contract=off: never fused by the compiler
    s = __builtin_fmaf(dy, dy, s);     // v_fma_f32
    s = __builtin_fmaf(dz, dz, s);
    return s;
}
// Two of them at once on the packed-fp32 VALU (v_pk_add/mul/fma_f32: two IEEE operations per lane and instruction,
// each rounded exactly like its scalar twin, in the same order) -- the brute-force distance loops are VALU-issue bound.
typedef float usip_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ usip_f32x2 usip_sqdist2(usip_f32x2 ax, usip_f32x2 ay, usip_f32x2 az,
                                                  usip_f32x2 bx, usip_f32x2 by, usip_f32x2 bz)
{
    const usip_f32x2 dx = ax - bx, dy = ay - by, dz = az - bz;
    usip_f32x2 s = dx * dx;
    s = __builtin_elementwise_fma(dy, dy, s);
    s = __builtin_elementwise_fma(dz, dz, s);
    return s;
}
__device__ __forceinline__ float usip_dist(float ax, float ay, float az, float bx, float by, float bz)
{
    // sqrtf is the correctly rounded IEEE sqrt under hipcc's default
    // -fhip-fp32-correctly-rounded-divide-sqrt (HIP's __fsqrt_rn is the 1-ulp native sqrt).
    return sqrtf(usip_sqdist(ax, ay, az, bx, by, bz));
}

// 16-byte streaming (non-temporal) load: data that is read exactly once.
typedef float usip_f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 usip_load_stream4(const float* p)
{
    usip_f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const usip_f32x4*>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}

// Sum over the 16 lanes of a DPP row, result in every lane of the row: quad_perm xor 1, xor 2, then the two
// mirror steps.  Pure VALU (v_add_f32 with a DPP modifier): no ds_bpermute, no LDS traffic.
__device__ __forceinline__ float usip_row16_sum(float v)
{
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, false));  // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, false));  // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, false)); // row_half_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, false)); // row_mirror
    return v;
}
