// Device helpers shared by the backward kernels (mpnn_edge_bwd.hip, mpnn_node_bwd.hip).
#pragma once
#include "mpnn_edge_common.h"

namespace {

// d/dx [x Phi(x)] = Phi(x) + x phi(x); same erfc approximation and the same exponential as gelu()
__device__ __forceinline__ float gelu_grad(float x) {
  const float z = x * 0.70710678118654752440f;
  const float az = fabsf(z);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, az, 1.0f));
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  p *= t;
  const float zs = z * 1.20112240878644977f;
  const float e = __builtin_amdgcn_exp2f(-(zs * zs));  // exp(-x^2 / 2)
  const float q = 0.5f * (p * e);                      // Phi(-|x|)
  const float Phi = x >= 0.f ? 1.0f - q : q;
  return fmaf(x * 0.39894228040143267794f, e, Phi);
}

__device__ __forceinline__ f32x4 gelu_grad4(f32x4 v) {
#ifdef MPNN_ABL_NOGELU  // timing-only ablation build: never shipped
  return v;
#endif
  f32x4 o;
  o.x = gelu_grad(v.x); o.y = gelu_grad(v.y); o.z = gelu_grad(v.z); o.w = gelu_grad(v.w);
  return o;
}

// GELU and its derivative together (the fused backward kernels need gelu(a) as a contraction operand and gelu'(a) for
// the chain, of the same a): one rcp + one exp2 shared, gelu = relu(x) - |x| Phi(-|x|) from the derivative's own
// Phi(-|x|) (A&S 7.1.26 erfc, 7.5e-8 absolute on Phi: the value is within 4e-7 of the exact-erf form for |x| < 6)
__device__ __forceinline__ float gelu_grad_value(float x, float& val) {
  const float z = x * 0.70710678118654752440f;
  const float az = fabsf(z);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, az, 1.0f));
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  p *= t;
  const float zs = z * 1.20112240878644977f;
  const float e = __builtin_amdgcn_exp2f(-(zs * zs));  // exp(-x^2 / 2)
  const float q = 0.5f * (p * e);                      // Phi(-|x|)
  const float Phi = x >= 0.f ? 1.0f - q : q;
  val = fmaf(-fabsf(x), q, fmaxf(x, 0.f));
  return fmaf(x * 0.39894228040143267794f, e, Phi);
}
// v <- gelu'(v) in place; returns gelu(v)
__device__ __forceinline__ f32x4 gelu_value_grad4(f32x4& v) {
#ifdef MPNN_ABL_NOGELU  // timing-only ablation build: never shipped
  return v;
#endif
  float a, b, c, d;
  v.x = gelu_grad_value(v.x, a); v.y = gelu_grad_value(v.y, b);
  v.z = gelu_grad_value(v.z, c); v.w = gelu_grad_value(v.w, d);
  return f32x4{a, b, c, d};
}

// Gradients are tiny (the loss is divided by 2000, train.py:72) while the f16 hi/lo split only resolves 6e-8
// absolute below 6e-5.  Every gradient tile is therefore multiplied by a power of two (exact) that brings the
// wave's largest magnitude to [2^12, 2^13) before it is split, and the fp32 product is scaled back.
__device__ __forceinline__ float pow2_scale_from_bits(unsigned mbits) {
  const int e = (int)(mbits >> 23) & 0xff;
  return e < 40 ? 1.0f : __int_as_float((266 - e) << 23);
}
__device__ __forceinline__ float pow2_inverse(float scale) {
  return __int_as_float((254 << 23) - __float_as_int(scale));
}
// wave-wide max |v| of a register tile (returned as float bits: non-negative floats order like unsigneds)
template <int KT>
__device__ __forceinline__ unsigned tile_absmax_bits(const f32x4 (&d)[MPNN_NT][KT]) {
  float m = 0.f;
#pragma unroll
  for (int t = 0; t < MPNN_NT; ++t)
#pragma unroll
    for (int c = 0; c < KT; ++c)
      m = fmaxf(fmaxf(m, fmaxf(fabsf(d[t][c].x), fabsf(d[t][c].y))), fmaxf(fabsf(d[t][c].z), fabsf(d[t][c].w)));
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  return (unsigned)__float_as_int(m);
}
// scaled split of a gradient tile; returns the factor that undoes the scaling
template <int KT>
__device__ __forceinline__ float grad_to_bfrag(f32x4 (&d)[MPNN_NT][KT], BFrag (&db)[NKK][KT], unsigned* amax_lds) {
  const unsigned mb = tile_absmax_bits<KT>(d);
  if (amax_lds && (threadIdx.x & 63) == 0) atomicMax(amax_lds, mb);
  const float sc = pow2_scale_from_bits(mb);
#pragma unroll
  for (int t = 0; t < MPNN_NT; ++t)
#pragma unroll
    for (int c = 0; c < KT; ++c) d[t][c] *= sc;
  tile_to_bfrag<MPNN_NT, KT, false>(d, db);
  return pow2_inverse(sc);
}

template <int KT>
__device__ __forceinline__ void zero_tile(f32x4 (&y)[MPNN_NT][KT]) {
#pragma unroll
  for (int t = 0; t < MPNN_NT; ++t)
#pragma unroll
    for (int c = 0; c < KT; ++c) y[t][c] = f32x4{0.f, 0.f, 0.f, 0.f};
}

// LayerNorm backward over each column's 128 features, in registers.
//   in : x = z (pre-LayerNorm input), d = dy;  out: d = dz, x = xhat (scratch)
//   the weight / bias gradients sum_edges dy*xhat, sum_edges dy are added to lnacc[0:128], lnacc[128:256] (LDS):
//   lnacc is THIS WAVE's accumulator (one writer lane per word, plain adds in the wave's own residue order - no atomics)
template <int KT>
__device__ __forceinline__ void tile_layernorm_bwd(f32x4 (&x)[MPNN_NT][KT], f32x4 (&d)[MPNN_NT][KT],
                                                   const float* __restrict__ ln_w, float* lnacc, int lane) {
  const int g = lane >> 4, col = lane & 15;
    // x <- xhat
    float rstd[KT];
#pragma unroll
    for (int c = 0; c < KT; ++c) {
      float sm = 0.f;
#pragma unroll
      for (int t = 0; t < MPNN_NT; ++t) sm += (x[t][c].x + x[t][c].y) + (x[t][c].z + x[t][c].w);
      const float mean = quarter_sum(sm) * (1.0f / MPNN_H);
      float q = 0.f;
#pragma unroll
      for (int t = 0; t < MPNN_NT; ++t) {
        x[t][c] -= mean;
        q += (x[t][c].x * x[t][c].x + x[t][c].y * x[t][c].y) + (x[t][c].z * x[t][c].z + x[t][c].w * x[t][c].w);
      }
      rstd[c] = rsqrtf(quarter_sum(q) * (1.0f / MPNN_H) + MPNN_LN_EPS);
#pragma unroll
      for (int t = 0; t < MPNN_NT; ++t) x[t][c] *= rstd[c];
    }
    // LayerNorm parameter gradients; d <- dxhat = dy * gamma
#pragma unroll
    for (int t = 0; t < MPNN_NT; ++t) {
      const f32x4 gm = *reinterpret_cast<const f32x4*>(ln_w + 16 * t + 4 * g);
      f32x4 sg = f32x4{0.f, 0.f, 0.f, 0.f}, sb = sg;
#pragma unroll
      for (int c = 0; c < KT; ++c) {
        sg += d[t][c] * x[t][c];
        sb += d[t][c];
        d[t][c] *= gm;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float vg = column_sum(sg[r]), vb = column_sum(sb[r]);
        if (col == 0) {
          lnacc[16 * t + 4 * g + r] += vg;
          lnacc[MPNN_H + 16 * t + 4 * g + r] += vb;
        }
      }
      FENCE();
    }
    // dz = rstd (dxhat - mean(dxhat) - xhat mean(dxhat xhat))
#pragma unroll
    for (int c = 0; c < KT; ++c) {
      float m1 = 0.f, m2 = 0.f;
#pragma unroll
      for (int t = 0; t < MPNN_NT; ++t) {
        m1 += (d[t][c].x + d[t][c].y) + (d[t][c].z + d[t][c].w);
        const f32x4 pr = d[t][c] * x[t][c];
        m2 += (pr.x + pr.y) + (pr.z + pr.w);
      }
      m1 = quarter_sum(m1) * (1.0f / MPNN_H);
      m2 = quarter_sum(m2) * (1.0f / MPNN_H);
#pragma unroll
      for (int t = 0; t < MPNN_NT; ++t) {
        d[t][c] = (d[t][c] - m1 - x[t][c] * m2) * rstd[c];
      }
    }
}

}  // namespace
