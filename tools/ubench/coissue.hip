// Does plain VALU work co-issue with v_mfma_f32_16x16x32_f16 on gfx950, and how must it be placed?
// In-kernel shader-clock timing (s_memtime), so the numbers are cycles whatever the DVFS clock is.
//   pattern P(m, n): m MFMAs (independent accumulators) followed by n v_fma_f32 (independent registers), repeated
//   waves per SIMD: 1 (256-thread blocks) or 2 (512-thread blocks); ROLE: all waves run P, or waves 0-3 run the MFMAs
//   only and waves 4-7 the VALU only
// build: hipcc -O3 --offload-arch=gfx950 tools/ubench/coissue.hip -o coissue
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

template <int M, int N, int ROLE, int TRANS>
__global__ void k(float* out, long long* cyc, int iters) {
  f32x4 acc[8];
  float a[16];
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  f32x2 pa[16];
  h8 ha, hb;
#pragma unroll
  for (int i = 0; i < 8; ++i) { ha[i] = (_Float16)(0.001f * (threadIdx.x + i)); hb[i] = (_Float16)(0.002f * (i + 1)); }
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = f32x4{0, 0, 0, 0};
#pragma unroll
  for (int i = 0; i < 16; ++i) a[i] = 0.001f * (threadIdx.x + i);
#pragma unroll
  for (int i = 0; i < 16; ++i) pa[i] = f32x2{0.001f * (threadIdx.x + i), 0.002f * i};
  const f32x2 pc1 = f32x2{1.0001f, 1.0002f}, pc2 = f32x2{0.5f, 0.25f};
  const int wave = threadIdx.x >> 6;
  const bool do_m = ROLE == 0, do_v = true;  // ROLE 2: VALU only
  __syncthreads();
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int rep = 0; rep < 16 / M; ++rep) {
      if (do_m) {
#pragma unroll
        for (int m = 0; m < M; ++m) {
          asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[(rep * M + m) & 7]) : "v"(ha), "v"(hb));
          if (M == 1 || (N % M == 0 && M <= 4)) {  // fine interleave: N/M VALU right after each MFMA
            if (do_v) {
#pragma unroll
              for (int v = 0; v < N / M; ++v) {
                const int q = (m * (N / M) + v + rep * 5) & 15;
                if (TRANS == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(a[q]));
                else if (TRANS == 2) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(pa[q]) : "v"(pc1), "v"(pc2));
                else if (TRANS == 3) asm volatile("v_fmaak_f32 %0, %0, %1, 0x3f7fff8c" : "+v"(a[q]) : "v"(1.0001f));
                else if (TRANS == 4) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[0]) : "v"(1.0001f), "v"(0.5f));
                else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[q]) : "v"(1.0001f), "v"(0.5f));
              }
            }
          }
        }
      }
      if (!(M == 1 || (N % M == 0 && M <= 4)) || !do_m) {  // clustered: all N VALU after the M MFMAs
        if (do_v) {
#pragma unroll
          for (int v = 0; v < N; ++v) {
            const int q = (v + rep * 5) & 15;
            if (TRANS == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(a[q]));
            else if (TRANS == 2) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(pa[q]) : "v"(pc1), "v"(pc2));
            else if (TRANS == 3) asm volatile("v_fmaak_f32 %0, %0, %1, 0x3f7fff8c" : "+v"(a[q]) : "v"(1.0001f));
            else if (TRANS == 4) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[0]) : "v"(1.0001f), "v"(0.5f));
            else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[q]) : "v"(1.0001f), "v"(0.5f));
          }
        }
      }
    }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += a[i] + pa[i].x + pa[i].y;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += acc[i].x + acc[i].y;
  if (s == 12345.678f) out[threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) { cyc[(blockIdx.x * 16 + wave) * 2] = t0; cyc[(blockIdx.x * 16 + wave) * 2 + 1] = t1; }
}

template <int M, int N, int ROLE, int TRANS>
void run(int waves_per_simd, const char* what) {
  const int iters = 4000;
  float* d; long long* c;
  hipMalloc(&d, 4096 * 4); hipMalloc(&c, 256 * 16 * 16);
  hipMemset(c, 0, 256 * 16 * 16);
  dim3 grid(256), block(waves_per_simd * 256);
  hipLaunchKernelGGL((k<M, N, ROLE, TRANS>), grid, block, 0, 0, d, c, 50);
  hipDeviceSynchronize();
  hipLaunchKernelGGL((k<M, N, ROLE, TRANS>), grid, block, 0, 0, d, c, iters);
  hipDeviceSynchronize();
  static long long h[256 * 16 * 2];
  hipMemcpy(h, c, sizeof(h), hipMemcpyDeviceToHost);
  double tot = 0, first = 0;
  const int nw = waves_per_simd * 4;
  for (int b = 0; b < 256; ++b) {
    long long s0 = h[b * 32], e1 = h[b * 32 + 1], e0 = h[b * 32 + 1];
    for (int w = 0; w < nw; ++w) {
      s0 = h[(b * 16 + w) * 2] < s0 ? h[(b * 16 + w) * 2] : s0;
      e1 = h[(b * 16 + w) * 2 + 1] > e1 ? h[(b * 16 + w) * 2 + 1] : e1;
      e0 = h[(b * 16 + w) * 2 + 1] < e0 ? h[(b * 16 + w) * 2 + 1] : e0;
    }
    tot += e1 - s0; first += e0 - s0;
  }
  const double units = (double)iters * 16;  // MFMA slots per wave
  printf("%-52s waves/SIMD=%d  SIMD cycles per slot (all waves done / waves): %6.1f   first wave done at %6.1f per slot\n", what,
         waves_per_simd, tot / 256 / units / waves_per_simd, first / 256 / units);
  hipFree(d); hipFree(c);
}

int main() {
  run<1, 0, 0, 0>(1, "MFMA only");
  run<1, 0, 0, 0>(2, "MFMA only");
  run<1, 1, 0, 0>(1, "1 MFMA + 1 fma, interleaved");
  run<1, 2, 0, 0>(1, "1 MFMA + 2 fma, interleaved");
  run<1, 3, 0, 0>(1, "1 MFMA + 3 fma, interleaved");
  run<1, 4, 0, 0>(1, "1 MFMA + 4 fma, interleaved");
  run<1, 6, 0, 0>(1, "1 MFMA + 6 fma, interleaved");
  run<1, 2, 0, 0>(2, "1 MFMA + 2 fma, interleaved");
  run<1, 4, 0, 0>(2, "1 MFMA + 4 fma, interleaved");
  run<1, 6, 0, 0>(2, "1 MFMA + 6 fma, interleaved");
  run<8, 16, 0, 0>(1, "8 MFMA then 16 fma (2 per MFMA), clustered");
  run<8, 16, 0, 0>(2, "8 MFMA then 16 fma (2 per MFMA), clustered");
  run<16, 32, 0, 0>(2, "16 MFMA then 32 fma (2 per MFMA), clustered");
  run<16, 64, 0, 0>(2, "16 MFMA then 64 fma (4 per MFMA), clustered");
  run<8, 32, 0, 0>(2, "8 MFMA then 32 fma (4 per MFMA), clustered");
  run<1, 4, 2, 0>(1, "VALU only: 4 fma per slot");
  run<1, 4, 2, 0>(2, "VALU only: 4 fma per slot");
  run<1, 4, 2, 0>(4, "VALU only: 4 fma per slot");
  run<1, 4, 2, 1>(1, "VALU only: 4 exp per slot");
  run<1, 4, 2, 1>(2, "VALU only: 4 exp per slot");
  run<1, 4, 2, 1>(4, "VALU only: 4 exp per slot");
  run<1, 2, 0, 0>(4, "1 MFMA + 2 fma, interleaved");
  run<1, 4, 0, 0>(4, "1 MFMA + 4 fma, interleaved");
  run<16, 64, 0, 0>(4, "16 MFMA then 64 fma (4 per MFMA), clustered");
  run<1, 1, 0, 1>(1, "1 MFMA + 1 exp, interleaved");
  run<1, 2, 0, 1>(1, "1 MFMA + 2 exp, interleaved");
  if (getenv("COISSUE_MORE")) {
    run<1, 2, 0, 2>(1, "1 MFMA + 2 pk_fma, interleaved");
    run<1, 4, 0, 2>(1, "1 MFMA + 4 pk_fma, interleaved");
    run<1, 6, 0, 2>(1, "1 MFMA + 6 pk_fma, interleaved");
    run<1, 4, 0, 2>(2, "1 MFMA + 4 pk_fma, interleaved");
    run<16, 64, 0, 2>(2, "16 MFMA then 64 pk_fma (4 per MFMA), clustered");
    run<1, 4, 2, 2>(1, "VALU only: 4 pk_fma per slot");
    run<1, 4, 2, 2>(2, "VALU only: 4 pk_fma per slot");
    run<1, 4, 2, 2>(4, "VALU only: 4 pk_fma per slot");
    run<1, 4, 0, 3>(1, "1 MFMA + 4 fmaak (literal), interleaved");
    run<1, 6, 0, 3>(1, "1 MFMA + 6 fmaak (literal), interleaved");
    run<1, 4, 0, 3>(2, "1 MFMA + 4 fmaak (literal), interleaved");
    run<1, 4, 2, 3>(2, "VALU only: 4 fmaak per slot");
    run<1, 4, 0, 4>(1, "1 MFMA + 4 dependent fma, interleaved");
    run<1, 4, 0, 4>(2, "1 MFMA + 4 dependent fma, interleaved");
    run<1, 4, 2, 4>(1, "VALU only: 4 dependent fma per slot");
    run<1, 4, 2, 4>(2, "VALU only: 4 dependent fma per slot");
  }
  return 0;
}
