// Per-node kernels for gfx950: node update (W3, LayerNorms, FFN) + the projections the next edge kernel
// gathers, occupancy MLP, output head
// (reference: src/mpnn/models/model_utils.py :81-88, :127-137, :151-155, :414-421, :527-528).
#include "mpnn_common.h"

MPNN_KERNELS_BEGIN

// ---------------------------------------------------------------------------
// node update + projections: one wavefront per 16 nodes, 4 waves (64 nodes) per workgroup, split-f16 MFMA.
//   dh  = (W3 gsum + cnt b3) / scale                               (:78-81 with W3 hoisted)
//   x   = LN1(h + dh);  y = LN2(x + W_out gelu(W_in x + b_in) + b_out);  h' = mask * y   (:82-88)
//   PQ  = Wproj h' + bproj      (nproj blocks of 128 outputs: the next kernels' P/Q terms)
// do_update = 0 skips the update and only projects h.
// do_update = 2 is the position-wise feed-forward block on its own (PositionWiseFeedForward.forward, :151-155):
//   hV_out = W_out gelu(W_in h + b_in) + b_out  - no W3 / LayerNorm / residual / mask.
// The node level has little parallelism (N / 16 wave tiles) and ~0.8 MB of weights per layer, so the kernel
// is a weight-streaming problem: the 9 + nproj packed 128x128 matrices (64 KiB each: W3, then W_in / W_out in
// four hidden-unit quarters, then the projections) flow through a two-slot LDS ring by LDS-DMA
// (global_load_lds), fetched once per workgroup and one matrix ahead of the MFMAs that consume them.
// ---------------------------------------------------------------------------
constexpr int NKK = MPNN_NT / 2;
constexpr int WSPLIT = MPNN_NT * NKK * 2 * 64;  // h8 elements of one packed 128x128 split matrix

// fragments of the staged matrix read from LDS ahead of the MFMAs that use them
#ifndef MPNN_NODE_LDS_RING
#define MPNN_NODE_LDS_RING 6  // measured: 3 -> 0.335, 6 -> 0.309, 10 / 15 -> 0.309 ms per step (7 launches, B32 x L512)
#endif
#ifndef MPNN_NODE_WAVES
#define MPNN_NODE_WAVES 4  // waves (16-node tiles) per workgroup of k_node_update
#endif

// quarter `part` (0 .. NKK-1) of a matrix' DMA: the GEMM of the stage before issues one quarter per k-step - sixteen
// LDS-DMA loads back to back fill the address unit's queue and the wave waits at their issue (see mpnn_edge.hip)
__device__ __forceinline__ void stage_matrix_part(const h8* __restrict__ g, h8* l, int part) {
#ifdef MPNN_ABL_NODMA
  return;
#endif
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int PER = WSPLIT / (64 * MPNN_NODE_WAVES) / NKK;
#pragma unroll
  for (int i = part * PER; i < (part + 1) * PER; ++i) {
    const int chunk = i * MPNN_NODE_WAVES + wave;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + chunk * 64 + lane),
                                     (__attribute__((address_space(3))) void*)(l + chunk * 64), 16, 0, 0);
  }
}

__device__ __forceinline__ void stage_matrix(const h8* __restrict__ g, h8* l) {
#ifdef MPNN_ABL_NODMA  // timing-only ablation build: never shipped
  return;
#endif
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < WSPLIT / (64 * MPNN_NODE_WAVES); ++i) {
    const int chunk = i * MPNN_NODE_WAVES + wave;  // 1 KiB per wave-instruction, LDS image is lane-linear
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + chunk * 64 + lane),
                                     (__attribute__((address_space(3))) void*)(l + chunk * 64), 16, 0, 0);
  }
}

__global__ void __launch_bounds__(64 * MPNN_NODE_WAVES)
k_node_update(const float* __restrict__ hV_in, const float* __restrict__ gsum, const float* __restrict__ cnt,
              float cnt_const, const float* __restrict__ mask, const h8* __restrict__ W3,
              const float* __restrict__ b3_g, const float* __restrict__ ln1w_g, const float* __restrict__ ln1b_g,
              const h8* __restrict__ Win, const float* __restrict__ bin_g, const h8* __restrict__ Wout,
              const float* __restrict__ bout_g, const float* __restrict__ ln2w_g, const float* __restrict__ ln2b_g,
              float inv_scale, int do_update, const h8* __restrict__ Wproj, const float* __restrict__ bproj_g,
              int nproj, const float* __restrict__ add_table, int ld_table, int off_table,
              const long long* __restrict__ add_idx, int add_block, int N, float* __restrict__ hV_out,
              float* __restrict__ PQ, const int* __restrict__ node_list) {
  __shared__ h8 lds[2 * WSPLIT];
  // every bias / LayerNorm vector of the launch, copied to LDS once: with one wave per SIMD each of their reads from
  // global memory was an exposed L2 round trip per stage (phase stamps: a third of the kernel outside its GEMMs)
  constexpr int PRM_PROJ = 7 * MPNN_H + 4 * MPNN_H;  // b3 ln1w ln1b | bin (512) | bout ln2w ln2b | bproj (<= 8 blocks)
  __shared__ __align__(16) float prm[PRM_PROJ + 8 * MPNN_H];
  {
    const int t = threadIdx.x;
    for (int i = t; i < MPNN_H; i += blockDim.x) {
      if (do_update == 1) {
        prm[i] = b3_g[i]; prm[MPNN_H + i] = ln1w_g[i]; prm[2 * MPNN_H + i] = ln1b_g[i];
        prm[8 * MPNN_H + i] = ln2w_g[i]; prm[9 * MPNN_H + i] = ln2b_g[i];
      }
      if (do_update) prm[7 * MPNN_H + i] = bout_g[i];
    }
    if (do_update)
      for (int i = t; i < 4 * MPNN_H; i += blockDim.x) prm[3 * MPNN_H + i] = bin_g[i];
    for (int i = t; i < nproj * MPNN_H; i += blockDim.x) prm[PRM_PROJ + i] = bproj_g[i];
  }
  const float* b3 = prm;
  const float* ln1w = prm + MPNN_H;
  const float* ln1b = prm + 2 * MPNN_H;
  const float* bin = prm + 3 * MPNN_H;
  const float* bout = prm + 7 * MPNN_H;
  const float* ln2w = prm + 8 * MPNN_H;
  const float* ln2b = prm + 9 * MPNN_H;
  const float* bproj = prm + PRM_PROJ;
  const int lane = threadIdx.x & 63;
  const int tile = blockIdx.x * MPNN_NODE_WAVES + (threadIdx.x >> 6);
  const int g = lane >> 4, col = lane & 15;
  const int node = tile * 16 + col;
  const bool live = node < N;       // no early exit: every wave takes part in the staging barriers
  const int nd0 = live ? node : N - 1;
  const int nd = node_list ? node_list[nd0] : nd0;  // sampling: the launch covers a listed subset of residues

  const int nupd = do_update == 1 ? 9 : (do_update == 2 ? 8 : 0);
  const int nmat = nupd + nproj;
  auto matrix = [&](int k) -> const h8* {
    if (k < nupd) {
      if (do_update == 1) {
        if (k == 0) return W3;
        --k;
      }
      return ((k & 1) ? Wout : Win) + (size_t)(k >> 1) * WSPLIT;
    }
    return Wproj + (size_t)(k - nupd) * WSPLIT;
  };
  int k = 0;
  if (nmat > 0) stage_matrix(matrix(0), lds);
  // fetch matrix k+1 into the other slot (every wave passed the barrier that ended matrix k-1's use of it)
  auto prefetch = [&]() {
    if (k + 1 < nmat) stage_matrix(matrix(k + 1), lds + ((k + 1) & 1) * WSPLIT);
  };
  auto prefetch_part = [&](int part) {
    // the next matrix' DMA in the first two k-steps, eight loads each (the last two k-steps cover its arrival: 0.277 ->
    // 0.271 ms per C2 step against four loads in every k-step; all sixteen in the first k-step: 0.282)
    if (k + 1 < nmat && part < 2) {
      stage_matrix_part(matrix(k + 1), lds + ((k + 1) & 1) * WSPLIT, 2 * part);
      stage_matrix_part(matrix(k + 1), lds + ((k + 1) & 1) * WSPLIT, 2 * part + 1);
    }
  };
  auto retire = [&]() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    ++k;
  };
  // the tile's own rows are requested before the wait for the first matrix: they arrive with it
  f32x4 h[MPNN_NT][1], gs[MPNN_NT][1];
  float cn = cnt_const, mk = 1.0f;
#pragma unroll
  for (int t = 0; t < MPNN_NT; ++t)
    h[t][0] = *reinterpret_cast<const f32x4*>(hV_in + (size_t)nd * MPNN_H + 16 * t + 4 * g);
  if (do_update == 1) {
#pragma unroll
    for (int t = 0; t < MPNN_NT; ++t)
      gs[t][0] = *reinterpret_cast<const f32x4*>(gsum + (size_t)nd * MPNN_H + 16 * t + 4 * g);
    if (cnt) cn = cnt[nd];
    mk = mask[nd];
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  if (do_update) {
    BFrag xb[NKK][1];
    if (do_update == 1) {
      f32x4 dh[MPNN_NT][1];
#pragma unroll
      for (int t = 0; t < MPNN_NT; ++t) dh[t][0] = *reinterpret_cast<const f32x4*>(b3 + 16 * t + 4 * g) * cn;
      tile_to_bfrag<MPNN_NT, 1, false>(gs, xb);
      split_gemm_stream_hook<1, NKK, MPNN_NT, MPNN_NODE_LDS_RING>(lds + (k & 1) * WSPLIT + lane, xb, dh, prefetch_part);
      retire();
#pragma unroll
      for (int t = 0; t < MPNN_NT; ++t) h[t][0] = h[t][0] + dh[t][0] * inv_scale;
      tile_layernorm<1>(h, ln1w, ln1b, lane);
    }
    tile_to_bfrag<MPNN_NT, 1, false>(h, xb);
    f32x4 o[MPNN_NT][1];
#pragma unroll
    for (int t = 0; t < MPNN_NT; ++t) {
      o[t][0] = *reinterpret_cast<const f32x4*>(bout + 16 * t + 4 * g);
      if (do_update == 1) o[t][0] += h[t][0];
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {  // FFN hidden units in quarters of 128
      f32x4 hid[MPNN_NT][1];
#pragma unroll
      for (int t = 0; t < MPNN_NT; ++t)
        hid[t][0] = *reinterpret_cast<const f32x4*>(bin + 128 * q + 16 * t + 4 * g);
      split_gemm_stream_hook<1, NKK, MPNN_NT, MPNN_NODE_LDS_RING>(lds + (k & 1) * WSPLIT + lane, xb, hid, prefetch_part);
      retire();
      BFrag hb[NKK][1];
      tile_to_bfrag<MPNN_NT, 1, true>(hid, hb);
      split_gemm_stream_hook<1, NKK, MPNN_NT, MPNN_NODE_LDS_RING>(lds + (k & 1) * WSPLIT + lane, hb, o, prefetch_part);
      retire();
    }
    if (do_update == 1) tile_layernorm<1>(o, ln2w, ln2b, lane);
#pragma unroll
    for (int t = 0; t < MPNN_NT; ++t) {
      h[t][0] = o[t][0] * mk;
      if (live) *reinterpret_cast<f32x4*>(hV_out + (size_t)nd * MPNN_H + 16 * t + 4 * g) = h[t][0];
    }
  }
  BFrag pb[NKK][1];
  tile_to_bfrag<MPNN_NT, 1, false>(h, pb);
  const int ld = nproj * MPNN_H;
  for (int p = 0; p < nproj; ++p) {
    f32x4 y[MPNN_NT][1];
#pragma unroll
    for (int t = 0; t < MPNN_NT; ++t) y[t][0] = *reinterpret_cast<const f32x4*>(bproj + p * MPNN_H + 16 * t + 4 * g);
    split_gemm_stream_hook<1, NKK, MPNN_NT, MPNN_NODE_LDS_RING>(lds + (k & 1) * WSPLIT + lane, pb, y, prefetch_part);
    retire();
    if (add_table && p == add_block) {  // decoder: fold the sequence term W1c W_s[S_n] into the A block
      const float* tr = add_table + (size_t)add_idx[nd] * ld_table + off_table + 4 * g;
#pragma unroll
      for (int t = 0; t < MPNN_NT; ++t) y[t][0] += *reinterpret_cast<const f32x4*>(tr + 16 * t);
    }
    if (live) {
#pragma unroll
      for (int t = 0; t < MPNN_NT; ++t)
        *reinterpret_cast<f32x4*>(PQ + (size_t)nd * ld + p * MPNN_H + 16 * t + 4 * g) = y[t][0];
    }
  }
}

// ---------------------------------------------------------------------------
// node update for the sampler's dependency levels (a few hundred listed residues per launch): latency, not
// throughput, matters, so the 11 matrix stages of k_node_update are spread over the 4 waves of a 16-node workgroup:
//   W3 by output-tile pairs, the FFN by hidden quarters (W_in_q and W_out_q per wave), the projections by halves;
// partial results meet through a 32 KiB LDS exchange buffer, weights stream straight from L2.  Same maths as
// k_node_update (cnt = cnt_const, no table add); gsum is [node][4][128]: the residue's column-tile sums as
// k_message_level stores them, added here in tile order (n_tiles of them).
// ---------------------------------------------------------------------------
#ifndef MPNN_LEVEL_RING
#define MPNN_LEVEL_RING 8  // weight fragments in flight per wave (L2 -> registers) in k_node_update_level
#endif
__global__ void __launch_bounds__(256)
k_node_update_level(const float* __restrict__ hV_in, const float* __restrict__ gsum, int n_tiles, float cnt_const,
                    const float* __restrict__ mask, const h8* __restrict__ W3, const float* __restrict__ b3,
                    const float* __restrict__ ln1w, const float* __restrict__ ln1b, const h8* __restrict__ Win,
                    const float* __restrict__ bin, const h8* __restrict__ Wout, const float* __restrict__ bout,
                    const float* __restrict__ ln2w, const float* __restrict__ ln2b, float inv_scale,
                    const h8* __restrict__ Wproj, const float* __restrict__ bproj, int nproj,
                    const int* __restrict__ node_list, int n_list, float* __restrict__ hV_out, float* __restrict__ PQ) {
  __shared__ f32x4 ex[4][MPNN_NT][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane >> 4, col = lane & 15;
  const int idx = blockIdx.x * 16 + col;
  const bool live = idx < n_list;
  const int nd = node_list[live ? idx : n_list - 1];

  f32x4 h[MPNN_NT][1], gs[MPNN_NT][1];
#pragma unroll
  for (int t = 0; t < MPNN_NT; ++t) {
    h[t][0] = *reinterpret_cast<const f32x4*>(hV_in + (size_t)nd * MPNN_H + 16 * t + 4 * g);
    gs[t][0] = *reinterpret_cast<const f32x4*>(gsum + (size_t)nd * 4 * MPNN_H + 16 * t + 4 * g);
    for (int c = 1; c < n_tiles; ++c)
      gs[t][0] += *reinterpret_cast<const f32x4*>(gsum + ((size_t)nd * 4 + c) * MPNN_H + 16 * t + 4 * g);
  }
  BFrag xb[NKK][1];
  tile_to_bfrag<MPNN_NT, 1, false>(gs, xb);
  {  // ---- W3: this wave's two output tiles
    f32x4 d2[2][1];
#pragma unroll
    for (int i = 0; i < 2; ++i) d2[i][0] = *reinterpret_cast<const f32x4*>(b3 + 16 * (2 * wave + i) + 4 * g) * cnt_const;
    split_gemm_stream<1, NKK, 2, (MPNN_LEVEL_RING < 7 ? MPNN_LEVEL_RING : 7)>(W3 + (size_t)(2 * wave) * NKK * 2 * 64 + lane, xb, d2);
    ex[0][2 * wave][lane] = d2[0][0];
    ex[0][2 * wave + 1][lane] = d2[1][0];
  }
  __syncthreads();
#pragma unroll
  for (int t = 0; t < MPNN_NT; ++t) h[t][0] += ex[0][t][lane] * inv_scale;
  tile_layernorm<1>(h, ln1w, ln1b, lane);
  tile_to_bfrag<MPNN_NT, 1, false>(h, xb);
  f32x4 o[MPNN_NT][1];
  {  // ---- FFN: hidden quarter `wave`
    f32x4 hid[MPNN_NT][1];
#pragma unroll
    for (int t = 0; t < MPNN_NT; ++t) {
      hid[t][0] = *reinterpret_cast<const f32x4*>(bin + 128 * wave + 16 * t + 4 * g);
      o[t][0] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    split_gemm_stream<1, NKK, MPNN_NT, MPNN_LEVEL_RING>(Win + (size_t)wave * WSPLIT + lane, xb, hid);
    BFrag hb[NKK][1];
    tile_to_bfrag<MPNN_NT, 1, true>(hid, hb);
    split_gemm_stream<1, NKK, MPNN_NT, MPNN_LEVEL_RING>(Wout + (size_t)wave * WSPLIT + lane, hb, o);
  }
  __syncthreads();  // every wave has read the W3 exchange
#pragma unroll
  for (int t = 0; t < MPNN_NT; ++t) ex[wave][t][lane] = o[t][0];
  __syncthreads();
#pragma unroll
  for (int t = 0; t < MPNN_NT; ++t)
    o[t][0] = h[t][0] + *reinterpret_cast<const f32x4*>(bout + 16 * t + 4 * g) +
              ((ex[0][t][lane] + ex[1][t][lane]) + (ex[2][t][lane] + ex[3][t][lane]));
  tile_layernorm<1>(o, ln2w, ln2b, lane);
  const float mk = mask[nd];
#pragma unroll
  for (int t = 0; t < MPNN_NT; ++t) {
    h[t][0] = o[t][0] * mk;
    if (live && wave == 0) *reinterpret_cast<f32x4*>(hV_out + (size_t)nd * MPNN_H + 16 * t + 4 * g) = h[t][0];
  }
  // ---- projections, one half (4 output tiles) per wave at a time
  BFrag pb[NKK][1];
  tile_to_bfrag<MPNN_NT, 1, false>(h, pb);
  const int ld = nproj * MPNN_H;
  for (int u = wave; u < 2 * nproj; u += 4) {
    const int p = u >> 1, t0 = 4 * (u & 1);
    f32x4 y[4][1];
#pragma unroll
    for (int i = 0; i < 4; ++i) y[i][0] = *reinterpret_cast<const f32x4*>(bproj + p * MPNN_H + 16 * (t0 + i) + 4 * g);
    split_gemm_stream<1, NKK, 4, (MPNN_LEVEL_RING < 15 ? MPNN_LEVEL_RING : 15)>(Wproj + (size_t)p * WSPLIT + (size_t)t0 * NKK * 2 * 64 + lane, pb, y);
    if (live) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        *reinterpret_cast<f32x4*>(PQ + (size_t)nd * ld + p * MPNN_H + 16 * (t0 + i) + 4 * g) = y[i][0];
    }
  }
}

// occupancy MLP: h_V = LN(W2 gelu(w0 * occ + b0) + b2)   (reference :414-421, :461-462)
__global__ void __launch_bounds__(256)
k_occupancy_mlp(const float* __restrict__ occ, const float* __restrict__ w0, const float* __restrict__ b0,
                const f32x4* __restrict__ W2, const float* __restrict__ b2, const float* __restrict__ lnw,
                const float* __restrict__ lnb, int N, float* __restrict__ hV) {
  const int lane = threadIdx.x & 63;
  const int tile = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (tile * 16 >= N) return;
  const int g = lane >> 4, col = lane & 15;
  const int node = tile * 16 + col;
  const bool live = node < N;
  const float o = occ[live ? node : N - 1];
  f32x4 x[MPNN_NT][1], y[MPNN_NT][1];
#pragma unroll
  for (int t = 0; t < MPNN_NT; ++t) {
    const f32x4 w = *reinterpret_cast<const f32x4*>(w0 + 16 * t + 4 * g);
    const f32x4 bb = *reinterpret_cast<const f32x4*>(b0 + 16 * t + 4 * g);
    x[t][0] = gelu4(w * o + bb);
    y[t][0] = *reinterpret_cast<const f32x4*>(b2 + 16 * t + 4 * g);
  }
  tile_gemm<1, MPNN_NT, 0, 8>(W2, lane, x, y);
  tile_layernorm<1>(y, lnw, lnb, lane);
  if (live) {
#pragma unroll
    for (int t = 0; t < MPNN_NT; ++t) *reinterpret_cast<f32x4*>(hV + (size_t)node * MPNN_H + 16 * t + 4 * g) = y[t][0];
  }
}

// output head: log_softmax(W_out h + b) over n_letters <= 32 (reference :527-528)
__global__ void __launch_bounds__(256)
k_logits(const float* __restrict__ hV, const f32x4* __restrict__ Wp, const float* __restrict__ bias, int n_letters,
         int N, float* __restrict__ log_probs) {
  const int lane = threadIdx.x & 63;
  const int tile = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (tile * 16 >= N) return;
  const int g = lane >> 4, col = lane & 15;
  const int node = tile * 16 + col;
  const bool live = node < N;
  const int nd = live ? node : N - 1;
  f32x4 h[MPNN_NT][1], y[2][1];
#pragma unroll
  for (int t = 0; t < MPNN_NT; ++t) h[t][0] = *reinterpret_cast<const f32x4*>(hV + (size_t)nd * MPNN_H + 16 * t + 4 * g);
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int letter = 16 * t + 4 * g + r;
      y[t][0][r] = letter < n_letters ? bias[letter] : 0.f;
    }
  tile_gemm<1, MPNN_NT, 0, 2>(Wp, lane, h, y);
  float mx = -INFINITY;
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (16 * t + 4 * g + r < n_letters) mx = fmaxf(mx, y[t][0][r]);
  mx = fmaxf(mx, __shfl_xor(mx, 16));
  mx = fmaxf(mx, __shfl_xor(mx, 32));
  float se = 0.f;
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (16 * t + 4 * g + r < n_letters) se += expf(y[t][0][r] - mx);
  se = quarter_sum(se);
  const float lse = mx + logf(se);
  if (live) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int letter = 16 * t + 4 * g + r;
        if (letter < n_letters) log_probs[(size_t)node * n_letters + letter] = y[t][0][r] - lse;
      }
  }
}

// ---------------------------------------------------------------------------
// y[r] = sum_b W_b x[r, 128 b : 128 (b+1)] + bias : a Linear(128 nb -> 128) over rows, one wavefront per 16 rows, the
// packed 128x128 blocks streamed from L2 (each block is read once per wave: for the stand-alone DecLayer surface, where
// the caller hands over the materialised [B,L,K,384] decoder input of the reference, :118-124)
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_linear_rows(const float* __restrict__ x, int ldx, int nb, const h8* __restrict__ W, const float* __restrict__ bias,
              long long n_rows, float* __restrict__ y) {
  const int lane = threadIdx.x & 63;
  const long long tile = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int g = lane >> 4, col = lane & 15;
  const long long row = tile * 16 + col;
  if (tile * 16 >= n_rows) return;
  const bool live = row < n_rows;
  const float* xr = x + (size_t)(live ? row : n_rows - 1) * ldx + 4 * g;
  f32x4 acc[MPNN_NT][1];
#pragma unroll
  for (int t = 0; t < MPNN_NT; ++t)
    acc[t][0] = bias ? *reinterpret_cast<const f32x4*>(bias + 16 * t + 4 * g) : f32x4{0.f, 0.f, 0.f, 0.f};
  for (int b = 0; b < nb; ++b) {
    f32x4 h[MPNN_NT][1];
#pragma unroll
    for (int t = 0; t < MPNN_NT; ++t) h[t][0] = *reinterpret_cast<const f32x4*>(xr + 128 * b + 16 * t);
    BFrag xb[NKK][1];
    tile_to_bfrag<MPNN_NT, 1, false>(h, xb);
    split_gemm_stream<1, NKK, MPNN_NT, 6>(W + (size_t)b * WSPLIT + lane, xb, acc);
  }
  if (live) {
#pragma unroll
    for (int t = 0; t < MPNN_NT; ++t) *reinterpret_cast<f32x4*>(y + (size_t)row * MPNN_H + 16 * t + 4 * g) = acc[t][0];
  }
}

MPNN_KERNELS_END

// ===========================================================================
// C ABI launchers
// ===========================================================================
static inline int ceil_div(long long a, long long b) { return (int)((a + b - 1) / b); }

extern "C" int MPNN_API(mpnn_node_update)(const float* hV_in, const float* gsum, const float* cnt, float cnt_const,
                                const float* mask, const float* W3_packed, const float* b3, const float* ln1_w,
                                const float* ln1_b, const float* Win_packed, const float* b_in,
                                const float* Wout_packed, const float* b_out, const float* ln2_w, const float* ln2_b,
                                float scale, int do_update, const float* Wproj_packed, const float* bproj, int nproj,
                                const float* add_table, int ld_table, int off_table, const int64_t* add_idx,
                                int add_block, int n_nodes, float* hV_out, float* PQ, void* stream) {
  MPNN_CHECK_ARG(hV_in && n_nodes > 0 && nproj >= 0 && nproj <= 8 && do_update >= 0 && do_update <= 2,
                 "mpnn_node_update: bad arguments (at most 8 projection blocks)");
  if (do_update == 2)
    MPNN_CHECK_ARG(Win_packed && b_in && Wout_packed && b_out && hV_out, "mpnn_node_update: null pointer in the FFN-only form");
  if (do_update == 1)
    MPNN_CHECK_ARG(gsum && mask && W3_packed && b3 && ln1_w && ln1_b && Win_packed && b_in && Wout_packed && b_out &&
                       ln2_w && ln2_b && hV_out && scale != 0.f,
                   "mpnn_node_update: null pointer in update");
  if (nproj > 0) MPNN_CHECK_ARG(Wproj_packed && bproj && PQ, "mpnn_node_update: null pointer in projection");
  if (add_table)
    MPNN_CHECK_ARG(add_idx && add_block >= 0 && add_block < nproj && ((ld_table | off_table) & 3) == 0,
                   "mpnn_node_update: bad table-add arguments");
  hipLaunchKernelGGL(k_node_update, dim3(ceil_div(n_nodes, 16 * MPNN_NODE_WAVES)), dim3(64 * MPNN_NODE_WAVES), 0,
                     (hipStream_t)stream, hV_in, gsum, cnt,
                     cnt_const, mask, (const h8*)W3_packed, b3, ln1_w, ln1_b, (const h8*)Win_packed, b_in,
                     (const h8*)Wout_packed, b_out, ln2_w, ln2_b, scale != 0.f ? 1.0f / scale : 0.f, do_update,
                     (const h8*)Wproj_packed, bproj, nproj, add_table, ld_table, off_table,
                     (const long long*)add_idx, add_block, n_nodes, hV_out, PQ, (const int*)nullptr);
  MPNN_LAUNCH_CHECK();
}

extern "C" int MPNN_API(mpnn_linear_rows)(const float* x, int ldx, int in_f, const float* W_packed, const float* bias,
                                long long n_rows, float* y, void* stream) {
  MPNN_CHECK_ARG(x && W_packed && y, "mpnn_linear_rows: null pointer");
  MPNN_CHECK_ARG(n_rows > 0 && in_f >= 128 && in_f % 128 == 0 && ldx >= in_f && (ldx & 3) == 0,
                 "mpnn_linear_rows: in_f must be a multiple of 128 and ldx a multiple of 4 floats");
  hipLaunchKernelGGL(k_linear_rows, dim3((unsigned)((n_rows + 63) / 64)), dim3(256), 0, (hipStream_t)stream, x, ldx,
                     in_f / 128, (const h8*)W_packed, bias, n_rows, y);
  MPNN_LAUNCH_CHECK();
}

extern "C" int MPNN_API(mpnn_node_update_subset)(const float* hV_in, const float* gsum, int n_tiles, float cnt_const,
                                       const float* mask,
                                       const float* W3_packed, const float* b3, const float* ln1_w,
                                       const float* ln1_b, const float* Win_packed, const float* b_in,
                                       const float* Wout_packed, const float* b_out, const float* ln2_w,
                                       const float* ln2_b, float scale, const float* Wproj_packed, const float* bproj,
                                       int nproj, const int32_t* node_list, int n_list, float* hV_out, float* PQ,
                                       void* stream) {
  MPNN_CHECK_ARG(hV_in && gsum && mask && W3_packed && b3 && ln1_w && ln1_b && Win_packed && b_in && Wout_packed &&
                     b_out && ln2_w && ln2_b && hV_out && node_list && n_list > 0 && nproj >= 0 && n_tiles >= 1 &&
                     n_tiles <= 4,
                 "mpnn_node_update_subset: bad arguments (1 .. 4 column tiles)");
  MPNN_CHECK_ARG(nproj == 0 || (Wproj_packed && bproj && PQ), "mpnn_node_update_subset: projections need Wproj, bproj, PQ");
  hipLaunchKernelGGL(k_node_update_level, dim3(ceil_div(n_list, 16)), dim3(256), 0, (hipStream_t)stream, hV_in, gsum,
                     n_tiles, cnt_const, mask, (const h8*)W3_packed, b3, ln1_w, ln1_b, (const h8*)Win_packed, b_in,
                     (const h8*)Wout_packed, b_out, ln2_w, ln2_b, 1.0f / scale, (const h8*)Wproj_packed, bproj, nproj,
                     node_list, n_list, hV_out, PQ);
  MPNN_LAUNCH_CHECK();
}

#ifndef MPNN_FAST
MPNN_KERNELS_BEGIN
// ---------------------------------------------------------------------------
// Sampling head for a listed subset of residues (one workgroup = 16 residues: wave 0 draws, all four waves fold
// the sequence terms into the gathered rows); the arithmetic is in mpnn_sample.h.
// ---------------------------------------------------------------------------
#include "mpnn_sample.h"

__global__ void __launch_bounds__(256)
k_sample(const float* __restrict__ hV, const f32x4* __restrict__ Wp, const float* __restrict__ bias,
         const float* __restrict__ logit_bias, int n_letters, const int* __restrict__ node_list, int N,
         float inv_temperature, unsigned long long seed, const float* __restrict__ designable,
         const long long* __restrict__ S_in, long long* __restrict__ S_out, float* __restrict__ log_probs,
         const float* __restrict__ TS, int ld_ts, SampleRows rows) {
  __shared__ float pr[16][33];
  __shared__ int pick[16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int tile = blockIdx.x;  // host: grid = ceil(N / 16), so tile * 16 < N
  if (wave == 0) {
    const int idx = tile * 16 + (lane & 15);
    const bool live = idx < N;
    const int nd = node_list[live ? idx : N - 1];
    sample_columns(hV, Wp, bias, logit_bias, n_letters, nd, live, inv_temperature, seed, designable, S_in, S_out,
                   log_probs, pr, pick, lane);
  }
  __syncthreads();
#ifdef MPNN_ABL_SAMPLE_NOTAIL  // timing-only ablation build: never shipped
  return;
#endif
  fold_sequence_rows<4, 16>([&](int l, int n) { return rows.A[l] + (size_t)n * rows.ld[l] + rows.off[l]; },
                            rows.n_layers, TS, ld_ts, node_list + tile * 16, min(16, N - tile * 16), pick, wave, lane);
}

MPNN_KERNELS_END
#endif
#ifndef MPNN_FAST
extern "C" int mpnn_sample_subset(const float* hV, const float* Wout_packed, const float* bias, const float* logit_bias,
                                  int n_letters, const int32_t* node_list, int n_list, float temperature,
                                  uint64_t seed, const float* designable, const int64_t* S_in, int64_t* S_out,
                                  float* log_probs, const float* TS, int ld_ts, int n_layers, float* const* A_rows,
                                  const int* A_ld, const int* A_off, void* stream) {
  MPNN_CHECK_ARG(hV && Wout_packed && bias && node_list && designable && S_in && S_out && log_probs && TS && A_rows &&
                     A_ld && A_off,
                 "mpnn_sample_subset: null pointer");
  MPNN_CHECK_ARG(n_list > 0 && n_letters >= 1 && n_letters <= 32 && temperature > 0.f && n_layers >= 0 && n_layers <= 4,
                 "mpnn_sample_subset: bad arguments (n_letters <= 32, temperature > 0, at most 4 decoder layers)");
  SampleRows rows;
  rows.n_layers = n_layers;
  for (int l = 0; l < 4; ++l) {
    rows.A[l] = l < n_layers ? A_rows[l] : nullptr;
    rows.ld[l] = l < n_layers ? A_ld[l] : 0;
    rows.off[l] = l < n_layers ? A_off[l] : 0;
  }
  hipLaunchKernelGGL(k_sample, dim3(ceil_div(n_list, 16)), dim3(256), 0, (hipStream_t)stream, hV,
                     (const f32x4*)Wout_packed, bias, logit_bias, n_letters, node_list, n_list, 1.0f / temperature,
                     (unsigned long long)seed, designable, (const long long*)S_in, (long long*)S_out, log_probs, TS,
                     ld_ts, rows);
  MPNN_LAUNCH_CHECK();
}
#endif

#ifndef MPNN_FAST
extern "C" int mpnn_occupancy_mlp(const float* occ, const float* w0, const float* b0, const float* W2_packed,
                                  const float* b2, const float* ln_w, const float* ln_b, int n_nodes, float* hV,
                                  void* stream) {
  MPNN_CHECK_ARG(occ && w0 && b0 && W2_packed && b2 && ln_w && ln_b && hV && n_nodes > 0,
                 "mpnn_occupancy_mlp: bad arguments");
  hipLaunchKernelGGL(k_occupancy_mlp, dim3(ceil_div(n_nodes, 64)), dim3(256), 0, (hipStream_t)stream, occ, w0, b0,
                     (const f32x4*)W2_packed, b2, ln_w, ln_b, n_nodes, hV);
  MPNN_LAUNCH_CHECK();
}
#endif

#ifndef MPNN_FAST
extern "C" int mpnn_logits_log_softmax(const float* hV, const float* Wout_packed, const float* bias, int n_letters,
                                       int n_nodes, float* log_probs, void* stream) {
  MPNN_CHECK_ARG(hV && Wout_packed && bias && log_probs && n_nodes > 0 && n_letters >= 1 && n_letters <= 32,
                 "mpnn_logits_log_softmax: bad arguments (n_letters <= 32)");
  hipLaunchKernelGGL(k_logits, dim3(ceil_div(n_nodes, 64)), dim3(256), 0, (hipStream_t)stream, hV,
                     (const f32x4*)Wout_packed, bias, n_letters, n_nodes, log_probs);
  MPNN_LAUNCH_CHECK();
}
#endif
