// Encoder / decoder message-passing kernels for gfx950
// (reference: src/mpnn/models/model_utils.py EncLayer :48-95, DecLayer :98-138,
//  PositionWiseFeedForward :141-155, decoder assembly :478-525, head :527-528).
//
// Algebra used (exact in real arithmetic, fp32 re-association only):
//   W1 [h_i | e_ik | h_j]  =  (W1a h_i + b1)  +  W1b e_ik  +  W1c h_j
//        -> the h_i / h_j terms are per-NODE projections P, Q computed by the node kernel
//           and gathered here as the accumulator's initial value; only W1b is per edge.
//   sum_k m_k (W3 g_k + b3) = W3 (sum_k m_k g_k) + (sum_k m_k) b3
//        -> W3 moves to the node kernel; the edge kernel emits gsum = sum_k m_k g_k and cnt.
// One wavefront owns one residue (KT tiles of 16 neighbours); see mpnn_common.h for the
// fragment layout.  The two per-edge 128x128 weights live in LDS (128 KiB), workgroups are
// persistent (grid = #CUs) and stride over residues.
#include "mpnn_common.h"

// waves per persistent workgroup (LDS holds 128 KiB of weights, so one workgroup per CU):
// 8 waves = two per SIMD (<= 256 registers each), 4 waves = one per SIMD (512 registers)
#ifndef MPNN_MSG_WAVES
#define MPNN_MSG_WAVES 8
#endif
#ifndef MPNN_UPD_WAVES
#define MPNN_UPD_WAVES 8
#endif

__device__ __forceinline__ void lds_fill(f32x4* dst, const f32x4* __restrict__ src, int n4) {
  for (int i = threadIdx.x; i < n4; i += blockDim.x) dst[i] = src[i];
}

// ---------------------------------------------------------------------------
// encoder message:  gsum[i] = sum_k m_ik gelu(W2 gelu(P_i + Q_j + W1b e_ik) + b2),  cnt[i] = sum_k m_ik
//   m_ik = mask_i mask_j (edge_mask_ik)      (reference :467-470, :75-81)
// ---------------------------------------------------------------------------
template <int KT>
__global__ void __launch_bounds__(MPNN_MSG_WAVES * 64)
k_enc_message(const f32x4* __restrict__ hE, const float* __restrict__ PQ, int ldpq, int offP, int offQ,
              const int* __restrict__ E_idx, const float* __restrict__ mask, const float* __restrict__ edge_mask,
              const f32x4* __restrict__ W1b, const f32x4* __restrict__ W2, const float* __restrict__ b2, int N,
              int L, int K, float* __restrict__ gsum, float* __restrict__ cnt) {
  __shared__ f32x4 lds[2 * 64 * 64];
  lds_fill(lds, W1b, 4096);
  lds_fill(lds + 4096, W2, 4096);
  __syncthreads();
  const f32x4* W1s = lds;
  const f32x4* W2s = lds + 4096;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane >> 4, col = lane & 15;
  constexpr int Kp = KT * 16;

  for (int node = blockIdx.x * MPNN_MSG_WAVES + wave; node < N; node += gridDim.x * MPNN_MSG_WAVES) {
    const int b = node / L;
    const float mi = mask[node];
    int jn[KT];
    float m[KT];
#pragma unroll
    for (int c = 0; c < KT; ++c) {
      const int e = 16 * c + col;
      jn[c] = b * L + E_idx[(size_t)node * Kp + e];
      float mm = mi * mask[jn[c]];
      if (edge_mask && e < K) mm *= edge_mask[(size_t)node * K + e];
      m[c] = e < K ? mm : 0.f;
    }
    // ---- layer 1: acc = P_i + Q_j + W1b e
    f32x4 h1[MPNN_NT][KT];
    {
      const float* Pi = PQ + (size_t)node * ldpq + offP + 4 * g;
#pragma unroll
      for (int t = 0; t < MPNN_NT; ++t) {
        const f32x4 p = *reinterpret_cast<const f32x4*>(Pi + 16 * t);
#pragma unroll
        for (int c = 0; c < KT; ++c)
          h1[t][c] = p + *reinterpret_cast<const f32x4*>(PQ + (size_t)jn[c] * ldpq + offQ + 4 * g + 16 * t);
      }
      const f32x4* x = hE + (size_t)node * (MPNN_NT * KT * 64) + lane;
#pragma unroll
      for (int ti = 0; ti < MPNN_NT; ++ti) {
        f32x4 xin[KT];
#pragma unroll
        for (int c = 0; c < KT; ++c) xin[c] = x[(ti * KT + c) * 64];
#pragma unroll
        for (int to = 0; to < MPNN_NT; ++to) {
          const f32x4 a = W1s[(to * MPNN_NT + ti) * 64 + lane];
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < KT; ++c) h1[to][c] = mfma4(a[r], xin[c][r], h1[to][c]);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      tile_gelu<MPNN_NT, KT>(h1);
    }
    // ---- layer 2 in two output halves, masked neighbour sum
    float csum = 0.f;
#pragma unroll
    for (int c = 0; c < KT; ++c) csum += m[c];
    csum = column_sum(csum);
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      f32x4 y[4][KT];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const f32x4 bb = *reinterpret_cast<const f32x4*>(b2 + 16 * (4 * half + t) + 4 * g);
#pragma unroll
        for (int c = 0; c < KT; ++c) y[t][c] = bb;
      }
      if (half == 0)
        tile_gemm<KT, MPNN_NT, 0, 4>(W2s, lane, h1, y);
      else
        tile_gemm<KT, MPNN_NT, 4, 8>(W2s, lane, h1, y);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < KT; ++c) s += gelu4(y[t][c]) * m[c];
        s.x = column_sum(s.x); s.y = column_sum(s.y); s.z = column_sum(s.z); s.w = column_sum(s.w);
        if (col == 0) *reinterpret_cast<f32x4*>(gsum + (size_t)node * MPNN_H + 16 * (4 * half + t) + 4 * g) = s;
      }
    }
    if (lane == 0) cnt[node] = csum;
  }
}

// ---------------------------------------------------------------------------
// encoder edge update:  e_ik <- LN3(e_ik + W13 gelu(W12 gelu(P'_i + Q'_j + W11b e_ik) + b12) + b13)
//   (reference :90-95; unmasked).  W11b, W12 in LDS, W13 streamed from L2.  In place on hE.
// ---------------------------------------------------------------------------
template <int KT>
__global__ void __launch_bounds__(MPNN_UPD_WAVES * 64)
k_enc_edge_update(f32x4* __restrict__ hE, const float* __restrict__ PQ, int ldpq, int offP, int offQ,
                  const int* __restrict__ E_idx, const f32x4* __restrict__ W11b, const f32x4* __restrict__ W12,
                  const float* __restrict__ b12, const f32x4* __restrict__ W13, const float* __restrict__ b13,
                  const float* __restrict__ ln_w, const float* __restrict__ ln_b, int N, int L) {
  __shared__ f32x4 lds[2 * 64 * 64];
  lds_fill(lds, W11b, 4096);
  lds_fill(lds + 4096, W12, 4096);
  __syncthreads();
  const f32x4* W1s = lds;
  const f32x4* W2s = lds + 4096;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane >> 4, col = lane & 15;
  constexpr int Kp = KT * 16;

  for (int node = blockIdx.x * MPNN_UPD_WAVES + wave; node < N; node += gridDim.x * MPNN_UPD_WAVES) {
    const int b = node / L;
    int jn[KT];
#pragma unroll
    for (int c = 0; c < KT; ++c) jn[c] = b * L + E_idx[(size_t)node * Kp + 16 * c + col];
    f32x4* x = hE + (size_t)node * (MPNN_NT * KT * 64) + lane;

    f32x4 h1[MPNN_NT][KT];
    {
      const float* Pi = PQ + (size_t)node * ldpq + offP + 4 * g;
#pragma unroll
      for (int t = 0; t < MPNN_NT; ++t) {
        const f32x4 p = *reinterpret_cast<const f32x4*>(Pi + 16 * t);
#pragma unroll
        for (int c = 0; c < KT; ++c)
          h1[t][c] = p + *reinterpret_cast<const f32x4*>(PQ + (size_t)jn[c] * ldpq + offQ + 4 * g + 16 * t);
      }
#pragma unroll
      for (int ti = 0; ti < MPNN_NT; ++ti) {
        f32x4 xin[KT];
#pragma unroll
        for (int c = 0; c < KT; ++c) xin[c] = x[(ti * KT + c) * 64];
#pragma unroll
        for (int to = 0; to < MPNN_NT; ++to) {
          const f32x4 a = W1s[(to * MPNN_NT + ti) * 64 + lane];
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < KT; ++c) h1[to][c] = mfma4(a[r], xin[c][r], h1[to][c]);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      tile_gelu<MPNN_NT, KT>(h1);
    }
    f32x4 h2[MPNN_NT][KT];
#pragma unroll
    for (int t = 0; t < MPNN_NT; ++t) {
      const f32x4 bb = *reinterpret_cast<const f32x4*>(b12 + 16 * t + 4 * g);
#pragma unroll
      for (int c = 0; c < KT; ++c) h2[t][c] = bb;
    }
    tile_gemm<KT, MPNN_NT, 0, 8>(W2s, lane, h1, h2);
    tile_gelu<MPNN_NT, KT>(h2);
    // layer 3: accumulator starts at e + b13 (residual folded in), weights from L2
#pragma unroll
    for (int t = 0; t < MPNN_NT; ++t) {
      const f32x4 bb = *reinterpret_cast<const f32x4*>(b13 + 16 * t + 4 * g);
#pragma unroll
      for (int c = 0; c < KT; ++c) h1[t][c] = x[(t * KT + c) * 64] + bb;
    }
    const f32x4* W13n = W13;
    asm volatile("" : "+s"(W13n));  // opaque per iteration: stops LICM hoisting all 512 weight loads out of the node loop
    tile_gemm<KT, MPNN_NT, 0, 8>(W13n, lane, h2, h1);
    tile_layernorm<KT>(h1, ln_w, ln_b, lane);
#pragma unroll
    for (int t = 0; t < MPNN_NT; ++t)
#pragma unroll
      for (int c = 0; c < KT; ++c) x[(t * KT + c) * 64] = h1[t][c];
  }
}

// ---------------------------------------------------------------------------
// decoder message (reference :478-525 assembly + DecLayer :121-127):
//   input_ik = [h_i | (bw+fw) e_ik | bw h_S[j] | bw h_V[j] + fw h_Venc[j]],  bw/fw from decoding ranks
//   => acc0 = Pa_i + bw (TS[S_j] + PV_j) + fw PVenc_j,   B operand = (bw+fw) e_ik
//   gsum[i] = sum_{k<K} gelu(W2 gelu(acc0 + W1b (..)) + b2)          (no attention mask: :525)
// ---------------------------------------------------------------------------
template <int KT>
__global__ void __launch_bounds__(MPNN_MSG_WAVES * 64)
k_dec_message(const f32x4* __restrict__ hE, const float* __restrict__ PQ, int ldpq, int offPa, int offPV,
              const float* __restrict__ PVenc, int ldenc, int offEnc, const float* __restrict__ TS, int ldts,
              int offTS, const long long* __restrict__ S, const int* __restrict__ rank,
              const int* __restrict__ E_idx, const float* __restrict__ mask, const float* __restrict__ edge_mask,
              const f32x4* __restrict__ W1b, const f32x4* __restrict__ W2, const float* __restrict__ b2, int N,
              int L, int K, float* __restrict__ gsum) {
  __shared__ f32x4 lds[2 * 64 * 64];
  lds_fill(lds, W1b, 4096);
  lds_fill(lds + 4096, W2, 4096);
  __syncthreads();
  const f32x4* W1s = lds;
  const f32x4* W2s = lds + 4096;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane >> 4, col = lane & 15;
  constexpr int Kp = KT * 16;

  for (int node = blockIdx.x * MPNN_MSG_WAVES + wave; node < N; node += gridDim.x * MPNN_MSG_WAVES) {
    const int b = node / L;
    const float mi = mask[node];
    const int ri = rank[node];
    int jn[KT];
    float bw[KT], fw[KT], valid[KT];
#pragma unroll
    for (int c = 0; c < KT; ++c) {
      const int e = 16 * c + col;
      jn[c] = b * L + E_idx[(size_t)node * Kp + e];
      const float att = ri > rank[jn[c]] ? 1.0f : 0.0f;
      float em = 1.0f;
      if (edge_mask && e < K) em = edge_mask[(size_t)node * K + e];
      bw[c] = mi * att * em;
      fw[c] = mi * (1.0f - att) * em;
      valid[c] = e < K ? 1.0f : 0.0f;
    }
    f32x4 h1[MPNN_NT][KT];
    {
      const float* Pi = PQ + (size_t)node * ldpq + offPa + 4 * g;
#pragma unroll
      for (int c = 0; c < KT; ++c) {
        const float* pv = PQ + (size_t)jn[c] * ldpq + offPV + 4 * g;
        const float* pe = PVenc + (size_t)jn[c] * ldenc + offEnc + 4 * g;
        const float* ts = TS + (size_t)S[jn[c]] * ldts + offTS + 4 * g;
#pragma unroll
        for (int t = 0; t < MPNN_NT; ++t) {
          const f32x4 p = *reinterpret_cast<const f32x4*>(Pi + 16 * t);
          const f32x4 v = *reinterpret_cast<const f32x4*>(pv + 16 * t);
          const f32x4 en = *reinterpret_cast<const f32x4*>(pe + 16 * t);
          const f32x4 s = *reinterpret_cast<const f32x4*>(ts + 16 * t);
          h1[t][c] = p + (s + v) * bw[c] + en * fw[c];
        }
      }
      const f32x4* x = hE + (size_t)node * (MPNN_NT * KT * 64) + lane;
#pragma unroll
      for (int ti = 0; ti < MPNN_NT; ++ti) {
        f32x4 xin[KT];
#pragma unroll
        for (int c = 0; c < KT; ++c) xin[c] = x[(ti * KT + c) * 64] * (bw[c] + fw[c]);
#pragma unroll
        for (int to = 0; to < MPNN_NT; ++to) {
          const f32x4 a = W1s[(to * MPNN_NT + ti) * 64 + lane];
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < KT; ++c) h1[to][c] = mfma4(a[r], xin[c][r], h1[to][c]);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      tile_gelu<MPNN_NT, KT>(h1);
    }
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      f32x4 y[4][KT];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const f32x4 bb = *reinterpret_cast<const f32x4*>(b2 + 16 * (4 * half + t) + 4 * g);
#pragma unroll
        for (int c = 0; c < KT; ++c) y[t][c] = bb;
      }
      if (half == 0)
        tile_gemm<KT, MPNN_NT, 0, 4>(W2s, lane, h1, y);
      else
        tile_gemm<KT, MPNN_NT, 4, 8>(W2s, lane, h1, y);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < KT; ++c) s += gelu4(y[t][c]) * valid[c];
        s.x = column_sum(s.x); s.y = column_sum(s.y); s.z = column_sum(s.z); s.w = column_sum(s.w);
        if (col == 0) *reinterpret_cast<f32x4*>(gsum + (size_t)node * MPNN_H + 16 * (4 * half + t) + 4 * g) = s;
      }
    }
  }
}

// ---------------------------------------------------------------------------
// node update + projections: one wavefront per 16 nodes, all weights streamed from L2.
//   dh  = (W3 gsum + cnt b3) / scale                               (:78-81 with W3 hoisted)
//   x   = LN1(h + dh);  y = LN2(x + W_out gelu(W_in x + b_in) + b_out);  h' = mask * y   (:82-88)
//   PQ  = Wproj h' + bproj      (nproj blocks of 128 outputs: the next kernels' P/Q terms)
// do_update = 0 skips the update and only projects h.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_node_update(const float* __restrict__ hV_in, const float* __restrict__ gsum, const float* __restrict__ cnt,
              float cnt_const, const float* __restrict__ mask, const f32x4* __restrict__ W3,
              const float* __restrict__ b3, const float* __restrict__ ln1w, const float* __restrict__ ln1b,
              const f32x4* __restrict__ Win, const float* __restrict__ bin, const f32x4* __restrict__ Wout,
              const float* __restrict__ bout, const float* __restrict__ ln2w, const float* __restrict__ ln2b,
              float inv_scale, int do_update, const f32x4* __restrict__ Wproj, const float* __restrict__ bproj,
              int nproj, int N, float* __restrict__ hV_out, float* __restrict__ PQ) {
  const int lane = threadIdx.x & 63;
  const int tile = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (tile * 16 >= N) return;
  const int g = lane >> 4, col = lane & 15;
  const int node = tile * 16 + col;
  const bool live = node < N;
  const int nd = live ? node : N - 1;

  f32x4 h[MPNN_NT][1];
#pragma unroll
  for (int t = 0; t < MPNN_NT; ++t)
    h[t][0] = *reinterpret_cast<const f32x4*>(hV_in + (size_t)nd * MPNN_H + 16 * t + 4 * g);

  if (do_update) {
    f32x4 gs[MPNN_NT][1], dh[MPNN_NT][1];
    const float cn = cnt ? cnt[nd] : cnt_const;
#pragma unroll
    for (int t = 0; t < MPNN_NT; ++t) {
      gs[t][0] = *reinterpret_cast<const f32x4*>(gsum + (size_t)nd * MPNN_H + 16 * t + 4 * g);
      dh[t][0] = *reinterpret_cast<const f32x4*>(b3 + 16 * t + 4 * g) * cn;
    }
    tile_gemm<1, MPNN_NT, 0, 8>(W3, lane, gs, dh);
#pragma unroll
    for (int t = 0; t < MPNN_NT; ++t) h[t][0] = h[t][0] + dh[t][0] * inv_scale;
    tile_layernorm<1>(h, ln1w, ln1b, lane);
    // FFN
    f32x4 hid[32][1];
#pragma unroll
    for (int t = 0; t < 32; ++t) hid[t][0] = *reinterpret_cast<const f32x4*>(bin + 16 * t + 4 * g);
    tile_gemm<1, MPNN_NT, 0, 32>(Win, lane, h, hid);
    tile_gelu<32, 1>(hid);
    f32x4 o[MPNN_NT][1];
#pragma unroll
    for (int t = 0; t < MPNN_NT; ++t) o[t][0] = h[t][0] + *reinterpret_cast<const f32x4*>(bout + 16 * t + 4 * g);
    tile_gemm<1, 32, 0, 8>(Wout, lane, hid, o);
    tile_layernorm<1>(o, ln2w, ln2b, lane);
    const float mk = mask[nd];
#pragma unroll
    for (int t = 0; t < MPNN_NT; ++t) {
      h[t][0] = o[t][0] * mk;
      if (live) *reinterpret_cast<f32x4*>(hV_out + (size_t)node * MPNN_H + 16 * t + 4 * g) = h[t][0];
    }
  }
  const int ld = nproj * MPNN_H;
  for (int p = 0; p < nproj; ++p) {
    f32x4 y[MPNN_NT][1];
#pragma unroll
    for (int t = 0; t < MPNN_NT; ++t) y[t][0] = *reinterpret_cast<const f32x4*>(bproj + p * MPNN_H + 16 * t + 4 * g);
    tile_gemm<1, MPNN_NT, 0, 8>(Wproj + (size_t)p * 4096, lane, h, y);
    if (live) {
#pragma unroll
      for (int t = 0; t < MPNN_NT; ++t)
        *reinterpret_cast<f32x4*>(PQ + (size_t)node * ld + p * MPNN_H + 16 * t + 4 * g) = y[t][0];
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

// ===========================================================================
// C ABI launchers
// ===========================================================================
static inline int ceil_div(long long a, long long b) { return (int)((a + b - 1) / b); }

static int edge_grid(int N, int waves) {
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) == hipSuccess) {
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
  }
  const int want = ceil_div(N, waves);
  return want < cus ? want : cus;
}

#define KT_DISPATCH(KT, CALL)   \
  switch (KT) {                 \
    case 1: { constexpr int kKT = 1; CALL; } break; \
    case 2: { constexpr int kKT = 2; CALL; } break; \
    case 3: { constexpr int kKT = 3; CALL; } break; \
    default: { constexpr int kKT = 4; CALL; } break; \
  }

extern "C" int mpnn_enc_message(const float* hE_frag, const float* PQ, int ldpq, int offP, int offQ,
                                const int32_t* E_idx32, const float* mask, const float* edge_mask,
                                const float* W1b_packed, const float* W2_packed, const float* b2, int B, int L, int K,
                                float* gsum, float* cnt, void* stream) {
  MPNN_CHECK_ARG(hE_frag && PQ && E_idx32 && mask && W1b_packed && W2_packed && b2 && gsum && cnt,
                 "mpnn_enc_message: null pointer");
  MPNN_CHECK_ARG(B > 0 && L > 0 && K >= 1 && K <= 64 && ldpq >= 128 && (ldpq % 4) == 0 && (offP % 4) == 0 && (offQ % 4) == 0,
                 "mpnn_enc_message: bad sizes");
  const int N = B * L, KT = (K + 15) / 16;
  dim3 grid(edge_grid(N, MPNN_MSG_WAVES)), block(MPNN_MSG_WAVES * 64);
  KT_DISPATCH(KT, hipLaunchKernelGGL((k_enc_message<kKT>), grid, block, 0, (hipStream_t)stream, (const f32x4*)hE_frag,
                                     PQ, ldpq, offP, offQ, E_idx32, mask, edge_mask, (const f32x4*)W1b_packed,
                                     (const f32x4*)W2_packed, b2, N, L, K, gsum, cnt));
  MPNN_LAUNCH_CHECK();
}

extern "C" int mpnn_enc_edge_update(float* hE_frag, const float* PQ, int ldpq, int offP, int offQ,
                                    const int32_t* E_idx32, const float* W11b_packed, const float* W12_packed,
                                    const float* b12, const float* W13_packed, const float* b13, const float* ln_w,
                                    const float* ln_b, int B, int L, int K, void* stream) {
  MPNN_CHECK_ARG(hE_frag && PQ && E_idx32 && W11b_packed && W12_packed && b12 && W13_packed && b13 && ln_w && ln_b,
                 "mpnn_enc_edge_update: null pointer");
  MPNN_CHECK_ARG(B > 0 && L > 0 && K >= 1 && K <= 64 && ldpq >= 128 && (ldpq % 4) == 0 && (offP % 4) == 0 && (offQ % 4) == 0,
                 "mpnn_enc_edge_update: bad sizes");
  const int N = B * L, KT = (K + 15) / 16;
  dim3 grid(edge_grid(N, MPNN_UPD_WAVES)), block(MPNN_UPD_WAVES * 64);
  KT_DISPATCH(KT, hipLaunchKernelGGL((k_enc_edge_update<kKT>), grid, block, 0, (hipStream_t)stream, (f32x4*)hE_frag, PQ,
                                     ldpq, offP, offQ, E_idx32, (const f32x4*)W11b_packed, (const f32x4*)W12_packed,
                                     b12, (const f32x4*)W13_packed, b13, ln_w, ln_b, N, L));
  MPNN_LAUNCH_CHECK();
}

extern "C" int mpnn_dec_message(const float* hE_frag, const float* PQ, int ldpq, int offPa, int offPV,
                                const float* PVenc, int ldenc, int offEnc, const float* TS, int ldts, int offTS,
                                const int64_t* S, const int32_t* rank, const int32_t* E_idx32, const float* mask,
                                const float* edge_mask, const float* W1b_packed, const float* W2_packed,
                                const float* b2, int B, int L, int K, float* gsum, void* stream) {
  MPNN_CHECK_ARG(hE_frag && PQ && PVenc && TS && S && rank && E_idx32 && mask && W1b_packed && W2_packed && b2 && gsum,
                 "mpnn_dec_message: null pointer");
  MPNN_CHECK_ARG(B > 0 && L > 0 && K >= 1 && K <= 64, "mpnn_dec_message: bad sizes");
  MPNN_CHECK_ARG(((ldpq | offPa | offPV | ldenc | offEnc | ldts | offTS) % 4) == 0, "mpnn_dec_message: strides/offsets must be multiples of 4 floats");
  const int N = B * L, KT = (K + 15) / 16;
  dim3 grid(edge_grid(N, MPNN_MSG_WAVES)), block(MPNN_MSG_WAVES * 64);
  KT_DISPATCH(KT, hipLaunchKernelGGL((k_dec_message<kKT>), grid, block, 0, (hipStream_t)stream, (const f32x4*)hE_frag,
                                     PQ, ldpq, offPa, offPV, PVenc, ldenc, offEnc, TS, ldts, offTS, (const long long*)S,
                                     rank, E_idx32, mask, edge_mask, (const f32x4*)W1b_packed, (const f32x4*)W2_packed,
                                     b2, N, L, K, gsum));
  MPNN_LAUNCH_CHECK();
}

extern "C" int mpnn_node_update(const float* hV_in, const float* gsum, const float* cnt, float cnt_const,
                                const float* mask, const float* W3_packed, const float* b3, const float* ln1_w,
                                const float* ln1_b, const float* Win_packed, const float* b_in,
                                const float* Wout_packed, const float* b_out, const float* ln2_w, const float* ln2_b,
                                float scale, int do_update, const float* Wproj_packed, const float* bproj, int nproj,
                                int n_nodes, float* hV_out, float* PQ, void* stream) {
  MPNN_CHECK_ARG(hV_in && n_nodes > 0 && nproj >= 0, "mpnn_node_update: bad arguments");
  if (do_update)
    MPNN_CHECK_ARG(gsum && mask && W3_packed && b3 && ln1_w && ln1_b && Win_packed && b_in && Wout_packed && b_out &&
                       ln2_w && ln2_b && hV_out && scale != 0.f,
                   "mpnn_node_update: null pointer in update");
  if (nproj > 0) MPNN_CHECK_ARG(Wproj_packed && bproj && PQ, "mpnn_node_update: null pointer in projection");
  hipLaunchKernelGGL(k_node_update, dim3(ceil_div(n_nodes, 64)), dim3(256), 0, (hipStream_t)stream, hV_in, gsum, cnt,
                     cnt_const, mask, (const f32x4*)W3_packed, b3, ln1_w, ln1_b, (const f32x4*)Win_packed, b_in,
                     (const f32x4*)Wout_packed, b_out, ln2_w, ln2_b, 1.0f / scale, do_update,
                     (const f32x4*)Wproj_packed, bproj, nproj, n_nodes, hV_out, PQ);
  MPNN_LAUNCH_CHECK();
}

extern "C" int mpnn_occupancy_mlp(const float* occ, const float* w0, const float* b0, const float* W2_packed,
                                  const float* b2, const float* ln_w, const float* ln_b, int n_nodes, float* hV,
                                  void* stream) {
  MPNN_CHECK_ARG(occ && w0 && b0 && W2_packed && b2 && ln_w && ln_b && hV && n_nodes > 0,
                 "mpnn_occupancy_mlp: bad arguments");
  hipLaunchKernelGGL(k_occupancy_mlp, dim3(ceil_div(n_nodes, 64)), dim3(256), 0, (hipStream_t)stream, occ, w0, b0,
                     (const f32x4*)W2_packed, b2, ln_w, ln_b, n_nodes, hV);
  MPNN_LAUNCH_CHECK();
}

extern "C" int mpnn_logits_log_softmax(const float* hV, const float* Wout_packed, const float* bias, int n_letters,
                                       int n_nodes, float* log_probs, void* stream) {
  MPNN_CHECK_ARG(hV && Wout_packed && bias && log_probs && n_nodes > 0 && n_letters >= 1 && n_letters <= 32,
                 "mpnn_logits_log_softmax: bad arguments (n_letters <= 32)");
  hipLaunchKernelGGL(k_logits, dim3(ceil_div(n_nodes, 64)), dim3(256), 0, (hipStream_t)stream, hV,
                     (const f32x4*)Wout_packed, bias, n_letters, n_nodes, log_probs);
  MPNN_LAUNCH_CHECK();
}
