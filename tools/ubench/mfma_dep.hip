// Back-to-back v_mfma_f32_16x16x32_f16 on gfx950: cycles per instruction when an accumulator is re-used every D
// instructions (D = 1: fully dependent chain; the edge kernels' split products re-use one every KT = 3), with one and
// two waves per SIMD.  In-kernel shader-clock timing (s_memtime).
// build: hipcc -O3 --offload-arch=gfx950 tools/ubench/mfma_dep.hip -o tools/ubench/bin/mfma_dep
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));

template <int D, bool X16 = false>
__global__ void k(float* out, long long* cyc, int iters) {
  f32x4 acc[D];
  h8 ha, hb;
  h4 qa, qb;
#pragma unroll
  for (int i = 0; i < 4; ++i) { qa[i] = (_Float16)(0.001f * (threadIdx.x + i)); qb[i] = (_Float16)(0.002f * (i + 1)); }
#pragma unroll
  for (int i = 0; i < 8; ++i) { ha[i] = (_Float16)(0.001f * (threadIdx.x + i)); hb[i] = (_Float16)(0.002f * (i + 1)); }
#pragma unroll
  for (int i = 0; i < D; ++i) acc[i] = f32x4{0, 0, 0, 0};
  __syncthreads();
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 24; ++m) {
      if (X16)   // the older half-depth form (k_wgrad, K = 48 edges = 3 x 16)
        asm volatile("v_mfma_f32_16x16x16_f16 %0, %1, %2, %0" : "+v"(acc[m % D]) : "v"(qa), "v"(qb));
      else
        asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[m % D]) : "v"(ha), "v"(hb));
    }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
#pragma unroll
  for (int i = 0; i < D; ++i) s += acc[i].x + acc[i].y;
  if (s == 12345.678f) out[threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) { cyc[(blockIdx.x * 16 + (threadIdx.x >> 6)) * 2] = t0; cyc[(blockIdx.x * 16 + (threadIdx.x >> 6)) * 2 + 1] = t1; }
}

template <int D, bool X16 = false>
void run(int wps) {
  const int iters = 2000;
  float* d; long long* c;
  if (hipMalloc(&d, 4096 * 4) != hipSuccess || hipMalloc(&c, 256 * 16 * 16) != hipSuccess) return;
  (void)hipMemset(c, 0, 256 * 16 * 16);
  hipLaunchKernelGGL((k<D, X16>), dim3(256), dim3(wps * 256), 0, 0, d, c, 50);
  (void)hipDeviceSynchronize();
  hipLaunchKernelGGL((k<D, X16>), dim3(256), dim3(wps * 256), 0, 0, d, c, iters);
  (void)hipDeviceSynchronize();
  static long long h[256 * 16 * 2];
  (void)hipMemcpy(h, c, sizeof(h), hipMemcpyDeviceToHost);
  double tot = 0;
  for (int b = 0; b < 256; ++b) {
    long long s0 = h[b * 32], e1 = h[b * 32 + 1];
    for (int w = 0; w < wps * 4; ++w) {
      s0 = h[(b * 16 + w) * 2] < s0 ? h[(b * 16 + w) * 2] : s0;
      e1 = h[(b * 16 + w) * 2 + 1] > e1 ? h[(b * 16 + w) * 2 + 1] : e1;
    }
    tot += e1 - s0;
  }
  printf("%s accumulator re-used every %d MFMAs, %d wave(s)/SIMD: %6.1f SIMD cycles per MFMA\n",
         X16 ? "16x16x16:" : "16x16x32:", D, wps,
         tot / 256 / ((double)iters * 24) / wps);
  (void)hipFree(d); (void)hipFree(c);
}

int main() {
  run<1>(1); run<2>(1); run<3>(1); run<4>(1); run<6>(1); run<8>(1);
  run<1>(2); run<2>(2); run<3>(2); run<4>(2); run<6>(2);
  run<1, true>(1); run<4, true>(1); run<4, true>(2);
  return 0;
}
