// ani_mlp_common.h — device helpers shared by the MLP kernels (ani_kernels_mlp.hip: one product per launch or per item;
// ani_kernels_mlpf.hip: the whole network of a row tile in one workgroup): CELU, the 16-bit operand splits, MFMA products
// of split operands.  Internal to libani_hip.so.
#pragma once
#include <hip/hip_runtime.h>

namespace ani {

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float celu_f(float z, float alpha, float inv_alpha) {
  // alpha * (exp(z/alpha) - 1) with the hardware exp2: for small |z| the subtraction loses relative, not absolute,
  // accuracy -- the absolute error stays below 1e-7 * alpha, which is what the sums downstream see
  return z > 0.f ? z : alpha * (__builtin_amdgcn_exp2f(z * inv_alpha * 1.4426950408889634f) - 1.f);
}
__device__ __forceinline__ float dcelu_from_h(float h, float inv_alpha) {
  // celu'(z) = 1 (z>0) or exp(z/alpha) = h/alpha + 1
  return h > 0.f ? 1.f : fmaf(h, inv_alpha, 1.f);
}

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void split3(float x, unsigned& h, unsigned& m, unsigned& l) {
  const unsigned xb = __float_as_uint(x);
  h = xb & 0xffff0000u;
  const float r1 = x - __uint_as_float(h);
  m = __float_as_uint(r1) & 0xffff0000u;
  const float r2 = r1 - __uint_as_float(m);
  l = __float_as_uint(r2);   // at most 8 significant bits left: its high half is exact
}
__device__ __forceinline__ unsigned pack_hi16(unsigned lo_elem, unsigned hi_elem) { return (lo_elem >> 16) | (hi_elem & 0xffff0000u); }
__device__ __forceinline__ void split2(float x0, float x1, unsigned& h, unsigned& l) {   // two elements, packed
  const f16x2 hv = {(_Float16)x0, (_Float16)x1};
  const f16x2 lv = {(_Float16)(x0 - (float)hv[0]), (_Float16)(x1 - (float)hv[1])};
  h = __builtin_bit_cast(unsigned, hv);
  l = __builtin_bit_cast(unsigned, lv);
}

template <int P>
struct Frag { uint4 p[P]; };
template <int P>
__device__ __forceinline__ Frag<P> load_frag(const unsigned char* src) {
  Frag<P> f;
#pragma unroll
  for (int i = 0; i < P; i++) f.p[i] = *reinterpret_cast<const uint4*>(src + 32 * i);
  return f;
}
template <int P>
__device__ __forceinline__ void mma_planes(const Frag<P>& a, const Frag<P>& b, f32x16& acc) {
#ifdef ABLF_NOMMA   // timing experiment only: operands kept alive, no matrix instruction
  for (int i = 0; i < P; i++)
    asm volatile("" ::"v"(a.p[i].x), "v"(a.p[i].y), "v"(a.p[i].z), "v"(a.p[i].w), "v"(b.p[i].x), "v"(b.p[i].y), "v"(b.p[i].z), "v"(b.p[i].w));
  return;
#endif
  if constexpr (P == 3) {   // smallest terms first
    const bf16x8 ah = __builtin_bit_cast(bf16x8, a.p[0]), am = __builtin_bit_cast(bf16x8, a.p[1]), al = __builtin_bit_cast(bf16x8, a.p[2]);
    const bf16x8 bh = __builtin_bit_cast(bf16x8, b.p[0]), bm = __builtin_bit_cast(bf16x8, b.p[1]), bl = __builtin_bit_cast(bf16x8, b.p[2]);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
  } else {
    const f16x8 ah = __builtin_bit_cast(f16x8, a.p[0]), al = __builtin_bit_cast(f16x8, a.p[1]);
    const f16x8 bh = __builtin_bit_cast(f16x8, b.p[0]), bl = __builtin_bit_cast(f16x8, b.p[1]);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);
  }
}
}  // namespace ani
