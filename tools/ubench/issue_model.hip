// Micro-benchmark: VALU / transcendental / f16-MFMA issue rates on gfx950 and whether they overlap
// (a) across two waves of one SIMD, (b) inside one wave's instruction stream.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

enum { FMA = 0, PKFMA = 1, EXP = 2, MFMA = 3, MIX_WAVES = 4, MIX_STREAM2 = 5, MIX_STREAM4 = 6, MIX_STREAM6 = 7 };

template <int MODE>
__global__ void __launch_bounds__(1024) k(float* out, int iters, int mfma_waves_mod) {
  const int wave = threadIdx.x >> 6;
  float a[16];
  f32x2 p[16];
  f32x4 acc[8];
  h8 ha, hb;
#pragma unroll
  for (int i = 0; i < 8; ++i) { ha[i] = (_Float16)(0.001f * (threadIdx.x + i)); hb[i] = (_Float16)(0.002f * i); }
#pragma unroll
  for (int i = 0; i < 16; ++i) { a[i] = 0.001f * (threadIdx.x + i); p[i] = f32x2{a[i], a[i] + 1.f}; }
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = f32x4{0, 0, 0, 0};
  int mode = MODE;
  bool do_mfma = false, do_valu = false;
  if (MODE == MIX_WAVES) { do_mfma = (wave / 4) % 2 == 0; do_valu = !do_mfma; }
  for (int it = 0; it < iters; ++it) {
    if (MODE == FMA || (MODE == MIX_WAVES && do_valu)) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 16; ++i) a[i] = __builtin_fmaf(a[i], 1.0001f, 0.5f);
    }
    if (MODE == PKFMA) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 16; ++i) p[i] = __builtin_elementwise_fma(p[i], f32x2{1.0001f, 1.0002f}, f32x2{0.5f, 0.25f});
    }
    if (MODE == EXP) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 16; ++i) a[i] = __builtin_amdgcn_exp2f(a[i]);
    }
    if (MODE == MFMA || (MODE == MIX_WAVES && do_mfma)) {
#pragma unroll
      for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, acc[i], 0, 0, 0);
    }
    if (MODE >= MIX_STREAM2) {
      constexpr int NV = MODE == MIX_STREAM2 ? 2 : (MODE == MIX_STREAM4 ? 4 : 6);
#pragma unroll
      for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, acc[i], 0, 0, 0);
#pragma unroll
          for (int v = 0; v < NV; ++v) a[(i * NV + v) & 15] = __builtin_fmaf(a[(i * NV + v) & 15], 1.0001f, 0.5f);
          __builtin_amdgcn_sched_barrier(0);
        }
    }
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += a[i] + p[i].x + p[i].y;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
  if (s == 12345.678f) out[threadIdx.x] = s + mode;
}

template <int MODE>
double run(int waves_per_simd, int iters, float* d) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  dim3 grid(256), block(waves_per_simd * 4 * 64);
  hipLaunchKernelGGL(k<MODE>, grid, block, 0, 0, d, 10, 0);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, grid, block, 0, 0, d, iters, 0);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

int main() {
  float* d; hipMalloc(&d, 4096 * 4);
  const int iters = 20000;
  const double clk = 2.4e9;  // nominal; cycles below are at nominal clock
  printf("mode waves/SIMD  ms   cycles/op(per SIMD, nominal 2.4GHz)\n");
  for (int w : {1, 2, 4}) { double ms = run<FMA>(w, iters, d); printf("v_fma_f32     w=%d %8.3f  %.2f\n", w, ms, ms * 1e-3 * clk / (iters * 64.0 * w)); }
  for (int w : {1, 2, 4}) { double ms = run<PKFMA>(w, iters, d); printf("v_pk_fma_f32  w=%d %8.3f  %.2f\n", w, ms, ms * 1e-3 * clk / (iters * 64.0 * w)); }
  for (int w : {1, 2, 4}) { double ms = run<EXP>(w, iters, d); printf("v_exp_f32     w=%d %8.3f  %.2f\n", w, ms, ms * 1e-3 * clk / (iters * 64.0 * w)); }
  for (int w : {1, 2, 4}) { double ms = run<MFMA>(w, iters, d); printf("mfma16x16x32h w=%d %8.3f  %.2f\n", w, ms, ms * 1e-3 * clk / (iters * 64.0 * w)); }
  { double ms = run<MIX_WAVES>(2, iters, d); printf("2 waves/SIMD: one 64 MFMA/iter, other 64 FMA/iter: %8.3f ms = %.1f cycles per (MFMA+FMA) pair\n", ms, ms * 1e-3 * clk / (iters * 64.0)); }
  { double ms = run<MIX_STREAM2>(1, iters, d); printf("1 wave, stream MFMA+2 FMA: %8.3f ms = %.1f cycles per group\n", ms, ms * 1e-3 * clk / (iters * 64.0)); }
  { double ms = run<MIX_STREAM4>(1, iters, d); printf("1 wave, stream MFMA+4 FMA: %8.3f ms = %.1f cycles per group\n", ms, ms * 1e-3 * clk / (iters * 64.0)); }
  { double ms = run<MIX_STREAM6>(1, iters, d); printf("1 wave, stream MFMA+6 FMA: %8.3f ms = %.1f cycles per group\n", ms, ms * 1e-3 * clk / (iters * 64.0)); }
  { double ms = run<MIX_STREAM4>(2, iters, d); printf("2 waves, stream MFMA+4 FMA: %8.3f ms = %.1f cycles per group per wave-pair\n", ms, ms * 1e-3 * clk / (iters * 64.0 * 2)); }
  { double ms = run<MIX_STREAM6>(2, iters, d); printf("2 waves, stream MFMA+6 FMA: %8.3f ms = %.1f cycles per group per wave-pair\n", ms, ms * 1e-3 * clk / (iters * 64.0 * 2)); }
  return 0;
}
