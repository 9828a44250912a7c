// Reproducer of the hipcc (ROCm 7.2) finding in DESIGN.md section 0b: an inline-asm operand TIED ("+v") to
// __builtin_bit_cast(unsigned, x[2]) of a float4 is handed element 0.
//   /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -S --cuda-device-only -o - tools/ubench/tied_operand_repro.hip | grep v_fma_mixlo
// prints, for the j = 1 pair,  v_fma_mixlo_f16 v0, -v5, 1.0, v0   where v5 = v_cvt_pkrtz(v2, v3): the fp32 operand should
// be v2.  With `float t = x[2 * j]` tied instead (and the bit cast applied to the RESULT) the operand is v2.
// tools/hazard_scan.py reports the wrong form as an S finding.
#include <hip/hip_runtime.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void split4(const f32x4 x, h4& hi, h4& lo) {
  unsigned h[2], l[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    h[j] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(x[2 * j], x[2 * j + 1]));
#ifdef TIE_AS_FLOAT  // the form the library uses
    float t = x[2 * j];
    asm("v_fma_mixlo_f16 %0, -%1, 1.0, %0 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "+v"(t) : "v"(h[j]));
    asm("v_fma_mixhi_f16 %0, -%1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(t) : "v"(h[j]), "v"(x[2 * j + 1]));
    l[j] = __builtin_bit_cast(unsigned, t);
#else
    l[j] = __builtin_bit_cast(unsigned, x[2 * j]);
    asm("v_fma_mixlo_f16 %0, -%1, 1.0, %0 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "+v"(l[j]) : "v"(h[j]));
    asm("v_fma_mixhi_f16 %0, -%1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l[j]) : "v"(h[j]), "v"(x[2 * j + 1]));
#endif
  }
  typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
  hi = __builtin_bit_cast(h4, (u32x2_t){h[0], h[1]});
  lo = __builtin_bit_cast(h4, (u32x2_t){l[0], l[1]});
}

__global__ void k(const f32x4* a, h4* o) {
  f32x4 x = a[threadIdx.x];
  h4 hi, lo;
  split4(x, hi, lo);
  o[2 * threadIdx.x] = hi;
  o[2 * threadIdx.x + 1] = lo;
}
