// usip_amd/csrc/split_common.h -- exact splits of fp32 operands into 16-bit planes for the matrix cores, shared by the
// split-product kernels (shared_mlp_x3.hip, layer_bwd_x2.hip).  See shared_mlp_x3.hip for the arithmetic.
#pragma once
#include "mlp_common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int XBK = 16;                                        // k per stage: one MFMA step

// 2^e with bound * 2^e in [2^(top-1), 2^top) for a positive normal bound; e clamped to [-60, 60] (an all-zero operand
// gets 2^60, which still maps it to zero); inf / NaN bounds leave the operand unscaled (the result is inf / NaN then,
// as it is in fp32).
__device__ __forceinline__ float pow2_scale(float bound, int top)
{
    const int eb = (int)((__float_as_uint(bound) >> 23) & 0xffu);          // bound = m 2^(eb - 127), m in [1, 2)
    int e = top - (eb - 126);
    e = e < -60 ? -60 : (e > 60 ? 60 : e);
    if (eb == 255) e = 0;
    return __uint_as_float((unsigned)(127 + e) << 23);
}
constexpr int X2H_TOP = 15;
                                    // scaled operands stay below 2^15 (fp16 max: 65504)

// Two fp32 -> three packed bf16 pairs (low half = first element), x = h + m + l exactly up to 2^-26 |x|.
__device__ __forceinline__ void split_pair(float x, float y, unsigned& p0, unsigned& p1, unsigned& p2)
{
    f32x2 v = {x, y};
    bf16x2 h = __builtin_convertvector(v, bf16x2);             // v_cvt_pk_bf16_f32, RNE
    p0 = __builtin_bit_cast(unsigned, h);
    v.x = x - __uint_as_float(p0 << 16);                       // exact
    v.y = y - __uint_as_float(p0 & 0xffff0000u);
    bf16x2 m = __builtin_convertvector(v, bf16x2);
    p1 = __builtin_bit_cast(unsigned, m);
    v.x = v.x - __uint_as_float(p1 << 16);
    v.y = v.y - __uint_as_float(p1 & 0xffff0000u);
    bf16x2 l = __builtin_convertvector(v, bf16x2);
    p2 = __builtin_bit_cast(unsigned, l);
}

// "f32x2h": the same idea with TWO fp16 planes (11 + 11 significant bits) and the THREE plane pairs of weight >= 2^-11:
// half the matrix products of f32x3.  fp16 has 5 exponent bits, so the caller scales both operands by powers of two
// (exact) such that the largest element sits near the top of the fp16 range; the low plane of an element x is then a
// normal fp16 number unless |x| < 2^-3 in scaled units, where what is lost is below 2^-26 of the operand's scale.
// Two fp32 -> two packed fp16 pairs, x = h + l up to 2^-22 |x|.
__device__ __forceinline__ void split_pair_h(float x, float y, unsigned& p0, unsigned& p1)
{
    f32x2 v = {x, y};
    f16x2 h = __builtin_convertvector(v, f16x2);               // RNE
    p0 = __builtin_bit_cast(unsigned, h);
    f32x2 hf = __builtin_convertvector(h, f32x2);
    v = v - hf;                                                // exact
    f16x2 l = __builtin_convertvector(v, f16x2);
    p1 = __builtin_bit_cast(unsigned, l);
}
