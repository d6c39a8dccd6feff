// Split-operand helpers shared by the product kernels (gnx_gemm.hip) and the fused edge kernels (gnx_fused.hip):
// an fp32 value written EXACTLY as the sum of three bf16 pieces (see the notes at k_gemm_ws3).
#pragma once
#include "gnx_common.hpp"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
#define W3_BM 64
#define W3_LDB 272                    // bytes per row of one bf16 image
#define W3_PIECE (W3_BM * W3_LDB)     // 17408 B
#define W3_BUF (3 * W3_PIECE)         // 52224 B

// Two elements at a time: ONE v_cvt_pk_bf16_f32 per pair and piece, the piece's fp32 value taken out of the packed word by a
// shift (low half) / a mask (high half): 46 VALU instructions per 8 elements.  The element-wise form compiled to 62 (a
// conversion per element plus packing moves), and with the SLP vectoriser on to 44 that contain packed-f32 subtractions,
// which issue badly beside MFMAs (build.py).  Same roundings either way (round-to-nearest-even pieces, exact remainders).
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) float split_f32x2;
typedef __attribute__((ext_vector_type(4))) unsigned split_u32x4;
__device__ __forceinline__ void split3(const float (&x)[8], bf16x8& p1, bf16x8& p2, bf16x8& p3) {
  split_u32x4 w1, w2, w3;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const split_f32x2 v = {x[2 * q], x[2 * q + 1]};
    const unsigned u1 = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
    const split_f32x2 r1 = {v.x - __builtin_bit_cast(float, u1 << 16), v.y - __builtin_bit_cast(float, u1 & 0xFFFF0000u)};
    const unsigned u2 = __builtin_bit_cast(unsigned, __builtin_convertvector(r1, bf16x2));
    const split_f32x2 r2 = {r1.x - __builtin_bit_cast(float, u2 << 16), r1.y - __builtin_bit_cast(float, u2 & 0xFFFF0000u)};
    w1[q] = u1;
    w2[q] = u2;
    w3[q] = __builtin_bit_cast(unsigned, __builtin_convertvector(r2, bf16x2));
  }
  p1 = __builtin_bit_cast(bf16x8, w1);
  p2 = __builtin_bit_cast(bf16x8, w2);
  p3 = __builtin_bit_cast(bf16x8, w3);
}

