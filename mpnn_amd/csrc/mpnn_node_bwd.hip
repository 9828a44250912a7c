// Node-level backward pieces (training step).
//
// k_node_wgrad: weight / bias gradient of an nn.Linear applied to per-residue features,
//     dW[o][i] += sum_n dY[n][o] X[n][i],   db[o] += sum_n dY[n][o]          (autograd of reference :78-93, :151-155)
// The shape is "tiny output, long reduction" (128..512 x 128..512 outputs over ~10^4 residues), which library GEMMs
// serve badly (measured 91 us per call through hipBLASLt's 32x32 macro-tile at 10240 residues).  Here the reduction
// is split over workgroups; each wave keeps 2 x 8 output tiles in registers, feeds v_mfma_f32_16x16x4_f32 (exact
// fp32 products) straight from global memory, and adds its partial tile to dW with fp32 atomics.
#include "mpnn_common.h"

__global__ void __launch_bounds__(256)
k_node_wgrad(const float* __restrict__ dY, int ldy, int out_f, const float* __restrict__ X, int ldx, int in_f, int N,
             int rows_per_wg, float* __restrict__ dW, int ldw, float* __restrict__ db) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane >> 4, col = lane & 15;
  const int o0 = blockIdx.y * 128 + 32 * wave, i0 = blockIdx.z * 128;
  const int r0 = blockIdx.x * rows_per_wg;
  const int r1 = min(N, r0 + rows_per_wg);
  f32x4 acc[2][8];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[a][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  float bs[2] = {0.f, 0.f};
  bool ov[2], iv[8];
#pragma unroll
  for (int a = 0; a < 2; ++a) ov[a] = o0 + 16 * a + col < out_f;
#pragma unroll
  for (int j = 0; j < 8; ++j) iv[j] = i0 + 16 * j + col < in_f;
#pragma unroll 4
  for (int n = r0; n < r1; n += 4) {
    const int row = n + g;
    const bool rv = row < r1;
    const float* dy = dY + (size_t)row * ldy + o0 + col;
    const float* x = X + (size_t)row * ldx + i0 + col;
    float a[2], b[8];
#pragma unroll
    for (int q = 0; q < 2; ++q) a[q] = (rv && ov[q]) ? dy[16 * q] : 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) b[j] = (rv && iv[j]) ? x[16 * j] : 0.f;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      bs[q] += a[q];
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[q][j] = mfma4(a[q], b[j], acc[q][j]);
    }
  }
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int o = o0 + 16 * q + 4 * g + r, i = i0 + 16 * j + col;
        if (o < out_f && i < in_f) unsafeAtomicAdd(dW + (size_t)o * ldw + i, acc[q][j][r]);
      }
  if (db && blockIdx.z == 0) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const float v = quarter_sum(bs[q]);
      if (g == 0 && ov[q]) unsafeAtomicAdd(db + o0 + 16 * q + col, v);
    }
  }
}

extern "C" int mpnn_node_wgrad(const float* dY, int ldy, int out_f, const float* X, int ldx, int in_f, int n_rows,
                               float* dW, int ldw, float* dbias, void* stream) {
  MPNN_CHECK_ARG(dY && X && dW, "mpnn_node_wgrad: null pointer");
  MPNN_CHECK_ARG(n_rows > 0 && out_f > 0 && in_f > 0 && ldy >= out_f && ldx >= in_f && ldw >= in_f,
                 "mpnn_node_wgrad: bad sizes");
  int rows = 128;  // rows per workgroup: enough workgroups to fill the chip, few enough to bound the atomics
  while ((long long)(n_rows + rows - 1) / rows > 512) rows *= 2;
  dim3 grid((n_rows + rows - 1) / rows, (out_f + 127) / 128, (in_f + 127) / 128), block(256);
  hipLaunchKernelGGL(k_node_wgrad, grid, block, 0, (hipStream_t)stream, dY, ldy, out_f, X, ldx, in_f, n_rows, rows, dW,
                     ldw, dbias);
  MPNN_LAUNCH_CHECK();
}
