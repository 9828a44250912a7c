// Node-level backward pieces (training step).
//
// k_node_wgrad: weight / bias gradient of an nn.Linear applied to per-residue features,
//     dW[o][i] += sum_n dY[n][o] X[n][i],   db[o] += sum_n dY[n][o]          (autograd of reference :78-93, :151-155)
// The shape is "tiny output, long reduction" (128..512 x 128..512 outputs over ~10^4 residues), which library GEMMs
// serve badly (measured 91 us per call through hipBLASLt's 32x32 macro-tile at 10240 residues).  Here the reduction
// is split over workgroups; each wave keeps 2 x 8 output tiles in registers, feeds v_mfma_f32_16x16x4_f32 (exact
// fp32 products) straight from global memory, and stores its partial tile to a workspace that a second small
// kernel adds up (per-element atomics from every workgroup cost more than the contraction itself).
#include "mpnn_common.h"

constexpr int NW_PART = 128 * 128 + 128;  // floats of one workgroup partial (tile | bias sums)

__global__ void __launch_bounds__(256)
k_node_wgrad(const float* __restrict__ dY, int ldy, int out_f, const float* __restrict__ X, int ldx, int in_f, int N,
             int rows_per_wg, float* __restrict__ part) {
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
  // batches of 8 k-steps (32 rows): all 80 loads of a batch are issued before its MFMAs (the operands are single
  // dwords at row stride, so latency, not bandwidth, is what has to be hidden)
  for (int n = r0; n < r1; n += 32) {
    float a[8][2], b[8][8];
#pragma unroll
    for (int s8 = 0; s8 < 8; ++s8) {
      const int row = n + 4 * s8 + g;
      const bool rv = row < r1;
      const float* dy = dY + (size_t)row * ldy + o0 + col;
      const float* x = X + (size_t)row * ldx + i0 + col;
#pragma unroll
      for (int q = 0; q < 2; ++q) a[s8][q] = (rv && ov[q]) ? dy[16 * q] : 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) b[s8][j] = (rv && iv[j]) ? x[16 * j] : 0.f;
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s8 = 0; s8 < 8; ++s8)
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        bs[q] += a[s8][q];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[q][j] = mfma4(a[s8][q], b[s8][j], acc[q][j]);
      }
  }
  // partial tile of this workgroup -> workspace; k_node_wgrad_reduce adds the row chunks up
  float* P = part + ((size_t)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * NW_PART;
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) P[(32 * wave + 16 * q + 4 * g + r) * 128 + 16 * j + col] = acc[q][j][r];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const float v = quarter_sum(bs[q]);
    if (g == 0) P[128 * 128 + 32 * wave + 16 * q + col] = v;
  }
}

// dW block (blockIdx.y, blockIdx.z) += sum over row chunks of the partial tiles; chunk slices meet by atomics
__global__ void __launch_bounds__(256)
k_node_wgrad_reduce(const float* __restrict__ part, int nx, int ny, int out_f, int in_f, float* __restrict__ dW, int ldw,
                    float* __restrict__ db) {
  const int e = blockIdx.x * 256 + threadIdx.x;  // element of the 128x128 (+128 bias) partial
  const int slice = blockIdx.y % 4, blk = blockIdx.y / 4;
  const int by = blk % ny, bz = blk / ny;
  if (e >= NW_PART) return;
  const float* P = part + (size_t)blk * nx * NW_PART + e;
  float s = 0.f;
  for (int w = slice; w < nx; w += 4) s += P[(size_t)w * NW_PART];
  if (e < 128 * 128) {
    const int o = by * 128 + e / 128, i = bz * 128 + e % 128;
    if (o < out_f && i < in_f) unsafeAtomicAdd(dW + (size_t)o * ldw + i, s);
  } else if (db && bz == 0) {
    const int o = by * 128 + (e - 128 * 128);
    if (o < out_f) unsafeAtomicAdd(db + o, s);
  }
}

extern "C" int mpnn_node_wgrad(const float* dY, int ldy, int out_f, const float* X, int ldx, int in_f, int n_rows,
                               float* dW, int ldw, float* dbias, float* workspace, int workspace_floats, void* stream) {
  MPNN_CHECK_ARG(dY && X && dW && workspace, "mpnn_node_wgrad: null pointer");
  MPNN_CHECK_ARG(n_rows > 0 && out_f > 0 && in_f > 0 && ldy >= out_f && ldx >= in_f && ldw >= in_f,
                 "mpnn_node_wgrad: bad sizes");
  const int ny = (out_f + 127) / 128, nz = (in_f + 127) / 128;
  const int max_x = workspace_floats / NW_PART / (ny * nz);
  MPNN_CHECK_ARG(max_x >= 1, "mpnn_node_wgrad: workspace too small (16512 floats per 128x128 block and row chunk)");
  int rows = 32;  // rows per workgroup: as many row chunks as the workspace and ~256 workgroups per block allow
  while ((n_rows + rows - 1) / rows > (max_x < 256 ? max_x : 256)) rows *= 2;
  const int nx = (n_rows + rows - 1) / rows;
  hipLaunchKernelGGL(k_node_wgrad, dim3(nx, ny, nz), dim3(256), 0, (hipStream_t)stream, dY, ldy, out_f, X, ldx, in_f,
                     n_rows, rows, workspace);
  hipLaunchKernelGGL(k_node_wgrad_reduce, dim3((NW_PART + 255) / 256, 4 * ny * nz), dim3(256), 0, (hipStream_t)stream,
                     workspace, nx, ny, out_f, in_f, dW, ldw, dbias);
  MPNN_LAUNCH_CHECK();
}
