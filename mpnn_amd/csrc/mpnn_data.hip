// Batch assembly on the GPU (SURVEY section 8f "next" N1): padded model inputs from pre-tensorised structures.
// Replaces the per-batch host work of the reference's featurize() (src/mpnn/data/data_utils.py:51-183: numpy padding,
// NaN handling, residue numbering, chain encoding, then seven host-to-device copies) by one launch over a shard that
// already lives in HBM.  Pure byte movement: ~61 B in, ~80 B out per residue slot.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mpnn_common.h"

namespace {

// one thread per (row, position) slot of the padded batch
__global__ void __launch_bounds__(256)
k_assemble_batch(const f32x4* __restrict__ Xc, const uint8_t* __restrict__ Sc, const int32_t* __restrict__ chainc,
                 const uint8_t* __restrict__ designedc, const long long* __restrict__ offsets,
                 const int32_t* __restrict__ ids, int B, int L, f32x4* __restrict__ X, long long* __restrict__ S,
                 float* __restrict__ mask, float* __restrict__ chain_M, long long* __restrict__ residue_idx,
                 long long* __restrict__ chain_enc, int32_t* __restrict__ lengths) {
  const int slot = blockIdx.x * blockDim.x + threadIdx.x;
  if (slot >= B * L) return;
  const int b = slot / L, pos = slot - b * L;
  const int p = ids[b];
  const long long o0 = offsets[p];
  const int len = (int)(offsets[p + 1] - o0);
  if (pos == 0) lengths[b] = len;
  f32x4 x[3] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
  long long s = 0, ridx = -100, ce = 0;   // padding values of data_utils.py:57,64,67
  float m = 0.f, cm = 0.f;
  if (pos < len) {
    const long long src = o0 + pos;
    bool finite = true;  // mask = isfinite(sum over the 4 atoms' coordinates) (:171)
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      x[q] = Xc[src * 3 + q];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        finite = finite && isfinite(x[q][r]);
        if (isnan(x[q][r])) x[q][r] = 0.f;  // X[isnan] = 0 (:170-172); infinities stay, as in the reference
      }
    }
    m = finite ? 1.f : 0.f;
    s = Sc[src];
    const int c = chainc[src];
    ce = c;
    ridx = 100LL * (c - 1) + pos;        // :120
    cm = designedc[src] ? 1.f : 0.f;     // :101,127
  }
#pragma unroll
  for (int q = 0; q < 3; ++q) X[(size_t)slot * 3 + q] = x[q];
  S[slot] = s;
  mask[slot] = m;
  chain_M[slot] = cm;
  residue_idx[slot] = ridx;
  chain_enc[slot] = ce;
}

// The same with the chain order of every structure given PER BATCH (featurize reshuffles the chains of an entry each
// time it is drawn, data_utils.py:88): row b is the concatenation of the segments seg_src / seg_len[seg_ptr[b] ..
// seg_ptr[b + 1]) of the shard, in that order; chain numbers and the +100 residue-index jumps follow the NEW order
// (:117-120).  A thread walks its row's segment list (a handful of chains).
__global__ void __launch_bounds__(256)
k_assemble_batch_ordered(const f32x4* __restrict__ Xc, const uint8_t* __restrict__ Sc,
                         const uint8_t* __restrict__ designedc, const int32_t* __restrict__ seg_ptr,
                         const long long* __restrict__ seg_src, const int32_t* __restrict__ seg_len, int B, int L,
                         f32x4* __restrict__ X, long long* __restrict__ S, float* __restrict__ mask,
                         float* __restrict__ chain_M, long long* __restrict__ residue_idx,
                         long long* __restrict__ chain_enc, int32_t* __restrict__ lengths) {
  const int slot = blockIdx.x * blockDim.x + threadIdx.x;
  if (slot >= B * L) return;
  const int b = slot / L, pos = slot - b * L;
  const int k0 = seg_ptr[b], k1 = seg_ptr[b + 1];
  long long src = -1;
  int c = 0, cum = 0;
  for (int k = k0; k < k1; ++k) {
    const int n = seg_len[k];
    if (src < 0 && pos < cum + n) {
      src = seg_src[k] + (pos - cum);
      c = k - k0 + 1;
    }
    cum += n;
  }
  if (pos == 0) lengths[b] = cum;
  f32x4 x[3] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
  long long s = 0, ridx = -100, ce = 0;
  float m = 0.f, cm = 0.f;
  if (src >= 0) {
    bool finite = true;
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      x[q] = Xc[src * 3 + q];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        finite = finite && isfinite(x[q][r]);
        if (isnan(x[q][r])) x[q][r] = 0.f;
      }
    }
    m = finite ? 1.f : 0.f;
    s = Sc[src];
    ce = c;
    ridx = 100LL * (c - 1) + pos;
    cm = designedc[src] ? 1.f : 0.f;
  }
#pragma unroll
  for (int q = 0; q < 3; ++q) X[(size_t)slot * 3 + q] = x[q];
  S[slot] = s;
  mask[slot] = m;
  chain_M[slot] = cm;
  residue_idx[slot] = ridx;
  chain_enc[slot] = ce;
}

// mask_self[b,i,j] = 0 where i and j are residues of the same chain, 1 elsewhere (padding included) (:61,119)
__global__ void __launch_bounds__(256)
k_mask_self(const long long* __restrict__ chain_enc, int B, int L, float* __restrict__ out) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)B * L * L) return;
  const int j = idx % L;
  const size_t bi = idx / L;
  const int b = bi / L;
  const long long ci = chain_enc[bi], cj = chain_enc[(size_t)b * L + j];
  out[idx] = (ci != 0 && ci == cj) ? 0.f : 1.f;
}

}  // namespace

namespace {
// one 16-byte piece of one gathered row per thread: consecutive threads walk a row, rows are 512 B in every call of
// the model, so each wave-instruction moves four whole rows
__global__ void k_gather_indexed(const float* __restrict__ src, const long long* __restrict__ idx, long long G, int n_src,
                                 long long n_idx, int C4, float* __restrict__ out, int out_ld, int out_col0) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= G * n_idx * C4) return;
  const int c = (int)(i % C4);
  const long long row = i / C4, g = row / n_idx;
  long long j = idx ? idx[row] : row - g * n_idx;
  j = j < 0 ? 0 : (j >= n_src ? n_src - 1 : j);
  const float4 v = reinterpret_cast<const float4*>(src + ((size_t)g * n_src + j) * (size_t)(4 * C4))[c];
  *reinterpret_cast<float4*>(out + (size_t)row * out_ld + out_col0 + 4 * c) = v;
}

__global__ void k_positional_encodings(const long long* __restrict__ offset, const long long* __restrict__ mask,
                                       long long n, int max_rel, const float* __restrict__ table, int npos,
                                       float* __restrict__ out) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * npos) return;
  const long long e = i / npos;
  const int p = (int)(i - e * npos);
  long long d = offset[e] + max_rel;
  d = d < 0 ? 0 : (d > 2 * max_rel ? 2 * max_rel : d);
  const long long m = mask[e];
  d = d * m + (1 - m) * (2 * max_rel + 1);
  d = d < 0 ? 0 : (d > 2 * max_rel + 1 ? 2 * max_rel + 1 : d);  // a mask outside {0, 1} must not index out of the table
  out[i] = table[d * npos + p];
}
}  // namespace

extern "C" int mpnn_gather_indexed(const float* src, const int64_t* idx, long long G, int n_src, long long n_idx, int C,
                                   float* out, int out_ld, int out_col0, void* stream) {
  MPNN_CHECK_ARG(src && out, "mpnn_gather_indexed: null pointer");
  MPNN_CHECK_ARG(G > 0 && n_src > 0 && n_idx > 0 && C > 0 && (C & 3) == 0 && (out_ld & 3) == 0 && (out_col0 & 3) == 0 &&
                     out_col0 >= 0 && out_ld >= out_col0 + C && (idx || n_idx == n_src),
                 "mpnn_gather_indexed: bad sizes (C, out_ld, out_col0 multiples of 4; idx == NULL needs n_idx == n_src)");
  const long long total = G * n_idx * (C / 4);
  MPNN_CHECK_ARG(total < (1LL << 31) * 256, "mpnn_gather_indexed: too many elements for one launch");
  hipLaunchKernelGGL(k_gather_indexed, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src,
                     (const long long*)idx, G, n_src, n_idx, C / 4, out, out_ld, out_col0);
  MPNN_LAUNCH_CHECK();
}

extern "C" int mpnn_positional_encodings(const int64_t* offset, const int64_t* mask, long long n, int max_rel,
                                         const float* table, int npos, float* out, void* stream) {
  MPNN_CHECK_ARG(offset && mask && table && out && n > 0 && max_rel >= 0 && npos > 0, "mpnn_positional_encodings: bad arguments");
  const long long total = n * npos;
  hipLaunchKernelGGL(k_positional_encodings, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const long long*)offset, (const long long*)mask, n, max_rel, table, npos, out);
  MPNN_LAUNCH_CHECK();
}

extern "C" int mpnn_assemble_batch(const float* X_cat, const uint8_t* S_cat, const int32_t* chain_cat,
                                   const uint8_t* designed_cat, const int64_t* offsets, const int32_t* protein_ids,
                                   int B, int L_max, float* X, int64_t* S, float* mask, float* chain_M,
                                   int64_t* residue_idx, int64_t* chain_encoding_all, int32_t* lengths, void* stream) {
  MPNN_CHECK_ARG(X_cat && S_cat && chain_cat && designed_cat && offsets && protein_ids && X && S && mask && chain_M &&
                     residue_idx && chain_encoding_all && lengths,
                 "mpnn_assemble_batch: null pointer");
  MPNN_CHECK_ARG(B > 0 && L_max > 0 && (long long)B * L_max < (1LL << 31), "mpnn_assemble_batch: bad sizes");
  const int slots = B * L_max;
  hipLaunchKernelGGL(k_assemble_batch, dim3((slots + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                     (const f32x4*)X_cat, S_cat, chain_cat, designed_cat, (const long long*)offsets, protein_ids, B, L_max,
                     (f32x4*)X, (long long*)S, mask, chain_M, (long long*)residue_idx, (long long*)chain_encoding_all,
                     lengths);
  MPNN_LAUNCH_CHECK();
}

extern "C" int mpnn_assemble_batch_ordered(const float* X_cat, const uint8_t* S_cat, const uint8_t* designed_cat,
                                           const int32_t* seg_ptr, const int64_t* seg_src, const int32_t* seg_len, int B,
                                           int L_max, float* X, int64_t* S, float* mask, float* chain_M,
                                           int64_t* residue_idx, int64_t* chain_encoding_all, int32_t* lengths,
                                           void* stream) {
  MPNN_CHECK_ARG(X_cat && S_cat && designed_cat && seg_ptr && seg_src && seg_len && X && S && mask && chain_M &&
                     residue_idx && chain_encoding_all && lengths,
                 "mpnn_assemble_batch_ordered: null pointer");
  MPNN_CHECK_ARG(B > 0 && L_max > 0 && (long long)B * L_max < (1LL << 31), "mpnn_assemble_batch_ordered: bad sizes");
  const int slots = B * L_max;
  hipLaunchKernelGGL(k_assemble_batch_ordered, dim3((slots + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                     (const f32x4*)X_cat, S_cat, designed_cat, seg_ptr, (const long long*)seg_src, seg_len, B, L_max,
                     (f32x4*)X, (long long*)S, mask, chain_M, (long long*)residue_idx, (long long*)chain_encoding_all,
                     lengths);
  MPNN_LAUNCH_CHECK();
}

extern "C" int mpnn_mask_self(const int64_t* chain_encoding_all, int B, int L, float* mask_self, void* stream) {
  MPNN_CHECK_ARG(chain_encoding_all && mask_self, "mpnn_mask_self: null pointer");
  MPNN_CHECK_ARG(B > 0 && L > 0, "mpnn_mask_self: bad sizes");
  const size_t total = (size_t)B * L * L;
  MPNN_CHECK_ARG(total / 256 < (1ull << 31), "mpnn_mask_self: too large");
  hipLaunchKernelGGL(k_mask_self, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const long long*)chain_encoding_all, B, L, mask_self);
  MPNN_LAUNCH_CHECK();
}
