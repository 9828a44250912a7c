// Does VALU work hide under the 32x32x16 f16 MFMA (8 passes)?  Same harness as issue_model.hip.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

template <int NV, int SHAPE, int KIND>  // SHAPE 32: 32x32x16, 16: 16x16x32; KIND 0 fma, 1 exp
__global__ void __launch_bounds__(512) k(float* out, int iters) {
  float a[16];
  f32x16 acc[4];
  f32x4 acc4[8];
  h8 ha, hb;
#pragma unroll
  for (int i = 0; i < 8; ++i) { ha[i] = (_Float16)(0.001f * (threadIdx.x + i)); hb[i] = (_Float16)(0.002f * i); }
#pragma unroll
  for (int i = 0; i < 16; ++i) a[i] = 0.001f * (threadIdx.x + i);
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) acc4[i] = f32x4{0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (SHAPE == 32) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, acc[i], 0, 0, 0);
        else { acc4[2*i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, acc4[2*i], 0, 0, 0); }
#pragma unroll
        for (int v = 0; v < NV; ++v) {
          const int q = (i * NV + v) & 15;
          a[q] = KIND == 0 ? __builtin_fmaf(a[q], 1.0001f, 0.5f) : __builtin_amdgcn_exp2f(a[q]);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += a[i];
#pragma unroll
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][5] + acc4[2*i].x;
  if (s == 12345.678f) out[threadIdx.x] = s;
}

template <int NV, int SHAPE, int KIND>
void run(int w, float* d, const char* what) {
  const int iters = 20000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  dim3 grid(256), block(w * 256);
  hipLaunchKernelGGL((k<NV, SHAPE, KIND>), grid, block, 0, 0, d, 10);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<NV, SHAPE, KIND>), grid, block, 0, 0, d, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%-34s w=%d %8.3f ms  %.1f cycles per (MFMA + %d valu) per SIMD\n", what, w, ms, ms * 1e-3 * 2.4e9 / (iters * 32.0 * w), NV);
}

int main() {
  float* d; hipMalloc(&d, 4096 * 4);
  run<0, 32, 0>(1, d, "32x32x16 alone");
  run<0, 32, 0>(2, d, "32x32x16 alone");
  run<2, 32, 0>(1, d, "32x32x16 + 2 fma");
  run<4, 32, 0>(1, d, "32x32x16 + 4 fma");
  run<8, 32, 0>(1, d, "32x32x16 + 8 fma");
  run<12, 32, 0>(1, d, "32x32x16 + 12 fma");
  run<8, 32, 0>(2, d, "32x32x16 + 8 fma");
  run<12, 32, 0>(2, d, "32x32x16 + 12 fma");
  run<2, 32, 1>(1, d, "32x32x16 + 2 exp");
  run<4, 32, 1>(2, d, "32x32x16 + 4 exp");
  run<0, 16, 0>(1, d, "16x16x32 alone");
  run<2, 16, 0>(1, d, "16x16x32 + 2 fma");
  run<4, 16, 0>(1, d, "16x16x32 + 4 fma");
  run<4, 16, 0>(2, d, "16x16x32 + 4 fma");
  return 0;
}
