// Per-edge message-passing kernels for gfx950
// (reference: src/mpnn/models/model_utils.py EncLayer :48-95, DecLayer :98-138,
//  decoder input assembly :478-525).
//
// Algebra (exact in real arithmetic, fp32 re-association only):
//   W1 [h_i | e_ik | h_j] = (W1a h_i + b1) + W1b e_ik + W1c h_j
//        -> h_i / h_j terms are per-NODE projections (node kernel); each edge gathers ONE 512-B row
//           of them as the accumulator's initial value, only W1b is per edge.
//   sum_k m_k (W3 g_k + b3) = W3 (sum_k m_k g_k) + (sum_k m_k) b3
//        -> W3 moves to the node kernel; the edge kernel emits gsum and cnt.
//   decoder: input_ik = [h_i | s e_ik | bw h_S[j] | bw h_V[j] + fw h_Venc[j]], bw = s [rank_j < rank_i],
//            fw = s - bw, s = mask_i edge_mask_ik  =>  W1 input = Pa_i + s (Row_ik + W1b e_ik),
//            Row_ik = (rank_j < rank_i) ? (W1c h_S[j] + W1d h_V[j]) : W1d h_Venc[j]      (one gathered row)
//
// Structure: one wavefront owns one residue (KT column tiles of 16 neighbours, fragment layout of
// mpnn_common.h), workgroups are persistent (grid = #CUs) and stride over residues with the two
// per-edge 128x128 weights resident in LDS (128 KiB).  Loads are software-pipelined by hand:
//   * hE k-tiles are double-buffered one tile ahead of their MFMAs,
//   * the NEXT residue's neighbour rows are loaded into each accumulator tile right after the
//     current residue's last read of it, its index/mask scalars one phase earlier,
//   * the edge update streams its third weight matrix from L2 three fragments ahead.
// Two waves per SIMD (8-wave workgroups) let one wave's GELU / reductions run under the other's MFMAs.
#include "mpnn_edge_common.h"
#include "mpnn_hip.h"
#include <stdio.h>

// MPNN_STAMP: diagnostic build only (never shipped): per-phase shader-clock totals of the persistent edge kernels,
// summed over all waves into g_stamp and printed by the launcher at the NEXT launch of the kernel.
#ifdef MPNN_STAMP
__device__ unsigned long long g_stamp[16];
#define STAMP_DECL long long st_t = __builtin_amdgcn_s_memtime(); unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define STAMP(i) do { const long long st_n = __builtin_amdgcn_s_memtime(); st_acc[i] += (unsigned long long)(st_n - st_t); st_t = st_n; } while (0)
#define STAMP_FLUSH(base) do { if ((threadIdx.x & 63) == 0) { for (int i_ = 0; i_ < 8; ++i_) atomicAdd(&g_stamp[(base) + i_], st_acc[i_]); } } while (0)
static void stamp_report(const char* what, int base) {
  unsigned long long h[16];
  if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_stamp), sizeof(h)) != hipSuccess) return;
  unsigned long long tot = 0;
  for (int i = 0; i < 8; ++i) tot += h[base + i];
  if (!tot) return;
  fprintf(stderr, "[stamp %s]", what);
  for (int i = 0; i < 8; ++i) fprintf(stderr, " p%d=%.1f%%", i, 100.0 * h[base + i] / tot);
  fprintf(stderr, "  total %.3e wave-cycles raw7 %llu\n", (double)tot, h[base + 7]);
  unsigned long long z[8] = {0};
  hipMemcpyToSymbol(HIP_SYMBOL(g_stamp), z, sizeof(z), base * sizeof(unsigned long long));
}
#else
#define STAMP_DECL
#define STAMP(i)
#define STAMP_FLUSH(base)
#endif

MPNN_KERNELS_BEGIN

// Tuning constants of the shipped kernels.  The variants that were measured against them and rejected (tile-wise
// kernels at K <= 48, burst prefetches, an LDS fragment read-ahead ring, staggered half-workgroups, other placements
// of the next residue's loads, flat W13 streams) were deleted in round 3; their numbers are in profiles/r02_ab_*.txt
// and their code in the history before that commit.  The only compile-time forks left are MPNN_FAST (the f16-product
// twin), MPNN_STAMP (phase stamps) and the timing-only MPNN_ABL_* ablations listed in mpnn_ablation.h.
#define MPNN_MSG_WAVES 8   // waves per workgroup of the message kernel (two per SIMD)
#define MPNN_UPD_WAVES 8   // ... of the edge update
// where, within the message kernel's first layer, the residue's own P row is requested: k-step MPNN_X0_FIRST of
// MPNN_X0_SPAN (the other k-step buffer is dead there)
#define MPNN_X0_FIRST 2
#define MPNN_X0_SPAN 2
#define MPNN_UPD_RING 4  // W13 lo fragments in flight ahead of their MFMAs (edge update, layer 3; 7 -> 4: 26 -> 6 spilled registers, -2 %)

// layer 1 of every edge MLP: acc += W1b * hE[node]; hE feature tiles stream from HBM one k-step
// (two tiles) ahead of their MFMAs and are split to f16 hi/lo on the fly
template <int KT>
__device__ __forceinline__ void load_kstep(const f32x4* __restrict__ x, int kk, f32x4 (&xt)[2][KT]) {
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int c = 0; c < KT; ++c)
#ifdef MPNN_ABL_NOXLOAD
      xt[u][c] = f32x4{1.f, 2.f, 3.f, 4.f};
#else
      xt[u][c] = x[((2 * kk + u) * KT + c) * 64];
#endif
}

// x0: the tiles of k-step 0, loaded by the caller (persistent kernels issue them one phase early, before the previous
// residue's epilogue: a wave's vector-memory operations complete in issue order, so a load issued behind the epilogue's
// stores and gathers would wait for all of them)
struct NoHook {
  __device__ __forceinline__ void operator()() const {}
};
// last: called at the top of the LAST k-step, when the other k-step buffer is dead: room for loads whose results are
// needed right after layer 1 (the residue's own P row) without holding their registers through the whole layer
template <int KT, typename Hook = NoHook>
__device__ __forceinline__ void layer1_stream(const h8* Ws, const f32x4* __restrict__ x, int lane,
                                              f32x4 (&acc)[MPNN_NT][KT], const f32x4 (&x0)[2][KT], Hook last = Hook()) {
  f32x4 xb[2][2][KT];
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int c = 0; c < KT; ++c) xb[0][u][c] = x0[u][c];
#pragma unroll
  for (int kk = 0; kk < NKK; ++kk) {
    // the next k-step's 2 KT tiles, one load per output tile of this k-step (pinned: hipcc otherwise places them all
    // near the END of the k-step, a few MFMAs before the next k-step waits for them)
    if (kk + 1 == NKK) last();
    BFrag b[KT];
#pragma unroll
    for (int c = 0; c < KT; ++c) split_pair(xb[kk & 1][0][c], xb[kk & 1][1][c], b[c]);
    FENCE();
    split_kstep_hook<KT, NKK, MPNN_NT>(Ws, kk, lane, b, acc, [&](int to) {
      if (kk + 1 < NKK && to < 2 * KT) xb[(kk + 1) & 1][to / KT][to % KT] = x[((2 * (kk + 1) + to / KT) * KT + to % KT) * 64];
    });
    FENCE();
  }
}

// Y += W X with W's fragments in LDS
template <int KT>
__device__ __forceinline__ void lds_gemm(const h8* Ws, int lane, const BFrag (&B)[NKK][KT], f32x4 (&Y)[MPNN_NT][KT]) {
  split_gemm<KT, NKK, 0, MPNN_NT>(Ws, lane, B, Y);
}

template <int KT, typename Hook = NoHook>
__device__ __forceinline__ void layer1_stream(const h8* Ws, const f32x4* __restrict__ x, int lane,
                                              f32x4 (&acc)[MPNN_NT][KT], Hook last = Hook()) {
  f32x4 x0[2][KT];
  load_kstep<KT>(x, 0, x0);
  layer1_stream<KT>(Ws, x, lane, acc, x0, last);
}

// masked neighbour sum of gelu(y) over the residue's edges -> gsum
template <int KT, typename Hook>
__device__ __forceinline__ void reduce_store(f32x4 (&y)[MPNN_NT][KT], const float (&w)[KT], float* __restrict__ out,
                                             int lane, Hook midway) {
  const int g = lane >> 4, col = lane & 15;
#pragma unroll
  for (int t = 0; t < MPNN_NT; ++t) {
    midway(t);  // the next residue's loads, a few per row tile (y[0 .. t) is dead: registers for them)
    f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < KT; ++c) s += gelu4(y[t][c]) * w[c];
    s = column_sum4(s);
    if (col == 0) *reinterpret_cast<f32x4*>(out + 16 * t + 4 * g) = s;
    FENCE();
  }
}

// ---------------------------------------------------------------------------
// message kernel, encoder (DEC = false) and decoder (DEC = true):
//   h1   = gelu(P_i + s_ik (Row_ik + W1b e_ik))
//   gsum = sum_k w_ik gelu(W2 h1 + b2)
//   enc: Row = Q_j, s = 1, w = mask_i mask_j edge_mask_ik, cnt = sum_k w       (:75-81, :467-470)
//   dec: Row by decoding rank (see header), s = mask_i edge_mask_ik, w = [k < K] (:478-525, :121-127)
// ---------------------------------------------------------------------------
// SAVE (training forward): the two pre-activations a1 = P + s (Row + W1b e) and a2 = W2 gelu(a1) + b2 are
// also written out, in the fragment layout of hE, for the backward kernels of mpnn_edge_bwd.hip.
template <int KT, bool DEC, bool SAVE>
__global__ void __launch_bounds__(MPNN_MSG_WAVES * 64)
k_message(const f32x4* __restrict__ hE, const float* __restrict__ Pbuf, int ldp, int offP,
          const float* __restrict__ Abuf, int lda, int offA, const float* __restrict__ Bbuf, int ldb, int offB,
          const int* __restrict__ rank, const int* __restrict__ E_idx, const float* __restrict__ mask,
          const float* __restrict__ edge_mask, const h8* __restrict__ W1b, const h8* __restrict__ W2,
          const float* __restrict__ b2, int N, int L, int K, float* __restrict__ gsum, float* __restrict__ cnt,
          f32x4* __restrict__ sv1, f32x4* __restrict__ sv2, int enc_mod, int w_valid) {
  __shared__ h8 lds[2 * WSPLIT];
  __shared__ __align__(16) float b2s[MPNN_H];  // bias from LDS: a global load would queue behind the row gathers
  lds_fill(lds, W1b, WSPLIT);
  lds_fill(lds + WSPLIT, W2, WSPLIT);
  if (threadIdx.x < MPNN_H) b2s[threadIdx.x] = b2[threadIdx.x];
  __syncthreads();
  const h8* W1s = lds;
  const h8* W2s = lds + WSPLIT;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane >> 4, col = lane & 15;
  constexpr int Kp = KT * 16;
  const ResidueRange rr = residue_range(N, MPNN_MSG_WAVES, wave);
  const int stride = rr.stride, Nend = rr.end;
  int node = rr.first;
  if (node >= Nend) return;

  // per-edge scalars of a residue: gathered row, scale s (dec), reduction weight w
  // shared-structure decoding (DEC, enc_mod < N): rows of the batch beyond the first enc_mod residues are further
  // sequences / orders on the SAME structures; hE, E_idx, edge_mask and the encoder rows Bbuf are read at
  // (residue % enc_mod), everything per sequence (P, A rows, rank, mask, outputs) at the residue itself
  auto setup = [&](int nd, const float* (&row)[KT], float (&s)[KT], float (&w)[KT]) {
    const int b = nd / L;
    const int sh = (DEC && nd >= enc_mod) ? nd % enc_mod : nd;
    const float mi = mask[nd];
    int ri = 0;
    if (DEC) ri = rank[nd];
#pragma unroll
    for (int c = 0; c < KT; ++c) {
      const int e = 16 * c + col;
      int jl = E_idx[(size_t)sh * Kp + e];
#ifdef MPNN_ABL_ROWSAME  // timing-only ablation build (never shipped): all 16 columns gather the residue's own row
      jl = nd - b * L;
#endif
      const int jn = b * L + jl;
      float em = 1.0f;
      if (edge_mask && e < K) em = edge_mask[(size_t)sh * K + e];
      if (DEC) {
        const bool before = ri > rank[jn];
        const int je = (sh / L) * L + jl;
        row[c] = before ? (Abuf + (size_t)jn * lda + offA + 4 * g) : (Bbuf + (size_t)je * ldb + offB + 4 * g);
        s[c] = mi * em;
        w[c] = e < K ? (w_valid ? em : 1.0f) : 0.0f;  // w_valid: sum over valid edges only (sparse variant)
      } else {
        row[c] = Abuf + (size_t)jn * lda + offA + 4 * g;
        s[c] = 1.0f;
        w[c] = e < K ? mi * mask[jn] * em : 0.0f;
      }
    }
  };

  const float* row[KT];
  float s[KT], w[KT];
  setup(node, row, s, w);
  f32x4 h1[MPNN_NT][KT];
#pragma unroll
  for (int t = 0; t < MPNN_NT; ++t)
#pragma unroll
    for (int c = 0; c < KT; ++c) h1[t][c] = row_tile(row[c], t);
  auto edge_block = [&](int nd) {  // a residue's hE block (shared-structure decoding reads it at residue % enc_mod)
    int ne = (DEC && nd >= enc_mod) ? nd % enc_mod : nd;
#ifdef MPNN_ABL_XSAME  // timing-only ablation build (never shipped): every hE read hits L2 (64 residues' blocks)
    ne &= 63;
#endif
    return hE + (size_t)ne * (MPNN_NT * KT * 64) + lane;
  };
  f32x4 x0[2][KT];  // k-step 0 of the residue's hE, always loaded one phase ahead
  load_kstep<KT>(edge_block(node), 0, x0);

  STAMP_DECL;
  for (; node < Nend; node += stride) {
    const int nxt = (node + stride < Nend) ? node + stride : Nend - 1;  // tail: harmless re-load of a valid residue
    STAMP(7);
    f32x4 p[MPNN_NT];  // P_i: needed only after layer 1, loaded during its last k-step
    const float* Pi = Pbuf + (size_t)node * ldp + offP + 4 * g;
    layer1_stream<KT>(W1s, edge_block(node), lane, h1, x0, [&]() {
#pragma unroll
      for (int t = 0; t < MPNN_NT; ++t) p[t] = row_tile(Pi, t);
    });
    STAMP(0);
#pragma unroll
    for (int t = 0; t < MPNN_NT; ++t)
#pragma unroll
      for (int c = 0; c < KT; ++c) h1[t][c] = DEC ? (h1[t][c] * s[c] + p[t]) : (h1[t][c] + p[t]);
    BFrag hb[NKK][KT];
    if (SAVE) {  // the saved pre-activation goes out six stores per k-step of the GELU + split, not as one burst of 24
      f32x4* o = sv1 + (size_t)node * (MPNN_NT * KT * 64) + lane;
#pragma unroll
      for (int kk = 0; kk < NKK; ++kk) {
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int c = 0; c < KT; ++c) __builtin_nontemporal_store(h1[2 * kk + u][c], &o[((2 * kk + u) * KT + c) * 64]);
#pragma unroll
        for (int c = 0; c < KT; ++c) split_pair(gelu4(h1[2 * kk][c]), gelu4(h1[2 * kk + 1][c]), hb[kk][c]);
        FENCE();
      }
    } else {
      tile_to_bfrag<MPNN_NT, KT, true>(h1, hb);
    }
    STAMP(1);
    // next residue's scalars (their loads fly under layer 2)
    const float* rown[KT];
    float sn[KT], wn[KT];
    setup(nxt, rown, sn, wn);
    f32x4 y[MPNN_NT][KT];
#pragma unroll
    for (int t = 0; t < MPNN_NT; ++t) {
      const f32x4 bb = *reinterpret_cast<const f32x4*>(b2s + 16 * t + 4 * g);
#pragma unroll
      for (int c = 0; c < KT; ++c) y[t][c] = bb;
    }
    lds_gemm<KT>(W2s, lane, hb, y);
    STAMP(2);
    // (training: the second saved pre-activation leaves row tile by row tile inside the reduction below)
    // h1 is free: the next residue's gathered rows and its first hE k-step are issued a few loads per row tile of the
    // GELU + reduction below, not as one burst: 30 back-to-back loads (16 rows each) fill the address unit's queue and
    // the wave then waits at the ISSUE of the later ones (measured with MPNN_STAMP: 12 % of the kernel); the k-step
    // starts once half of y has been reduced (registers)
    STAMP(3);
    reduce_store<KT>(y, w, gsum + (size_t)node * MPNN_H, lane, [&](int t) {
      if (SAVE) {
        f32x4* o = sv2 + (size_t)node * (MPNN_NT * KT * 64) + lane;
#pragma unroll
        for (int c = 0; c < KT; ++c) __builtin_nontemporal_store(y[t][c], &o[(t * KT + c) * 64]);
      }
#pragma unroll
      for (int c = 0; c < KT; ++c) h1[t][c] = row_tile(rown[c], t);
      // the k-step's 2 KT tiles over MPNN_X0_SPAN row tiles from MPNN_X0_FIRST on (a dead y tile frees half a k-step
      // tile pair; loads complete in issue order and the k-step is what layer 1 needs FIRST, so it goes out early)
      if (t >= MPNN_X0_FIRST && t < MPNN_X0_FIRST + MPNN_X0_SPAN) {
        const f32x4* xn = edge_block(nxt);
#pragma unroll
        for (int i = 0; i < 2 * KT; ++i)
          if (i * MPNN_X0_SPAN / (2 * KT) == t - MPNN_X0_FIRST) {
#ifdef MPNN_ABL_NOXLOAD
            x0[i / KT][i % KT] = f32x4{1.f, 2.f, 3.f, 4.f};
#else
            x0[i / KT][i % KT] = xn[i * 64];
#endif
          }
      }
    });
    STAMP(6);
    if (!DEC) {
      float csum = 0.f;
#pragma unroll
      for (int c = 0; c < KT; ++c) csum += w[c];
      csum = column_sum(csum);
      if (lane == 0) cnt[node] = csum;
    }
#pragma unroll
    for (int c = 0; c < KT; ++c) {
      s[c] = sn[c];
      w[c] = wn[c];
    }
    STAMP(7);
  }
  STAMP_FLUSH(0);
}

// ---------------------------------------------------------------------------
// encoder edge update, in place on hE (reference :90-95; unmasked):
//   e_ik <- LN3(e_ik + W13 gelu(W12 gelu(P'_i + Q'_j + W11b e_ik) + b12) + b13)
// LDS (all 160 KiB): W11b and W12 (f16 hi + lo, 64 KiB each), the hi halves of W13's first W3_LDS_FRAGS fragments
// (30 KiB) and the four parameter vectors b12, b13, ln_w, ln_b (2 KiB).  What streams from L2 is W13's lo halves
// (32 KiB per residue instead of 64) through a register ring MPNN_UPD_RING steps ahead of their MFMAs, plus the hi
// halves of the last two fragments.
// Order of a wave's vector-memory operations matters (they complete in issue order): the next residue's first hE
// k-step and its gathered rows are issued BEFORE the epilogue's stores, and nothing the epilogue waits for is issued
// behind them (the parameter vectors come from LDS).
// ---------------------------------------------------------------------------
// SAVE (training forward): out of place (hE_out), and a1', a2' and the pre-LayerNorm sum z are written out.
constexpr int W3_LDS_FRAGS = 30;  // of the NKK * 8 = 32 (to, kk) fragments of W13, in step order q = kk * 8 + to
template <int KT, bool SAVE>
__global__ void __launch_bounds__(MPNN_UPD_WAVES * 64)
k_enc_edge_update(f32x4* __restrict__ hE, const float* __restrict__ Pb, int ldp, int offP,
                  const float* __restrict__ Qb, int ldq, int offQ,
                  const int* __restrict__ E_idx, const h8* __restrict__ W11b, const h8* __restrict__ W12,
                  const float* __restrict__ b12, const h8* __restrict__ W13, const float* __restrict__ b13,
                  const float* __restrict__ ln_w, const float* __restrict__ ln_b, int N, int L,
                  f32x4* __restrict__ hE_out, f32x4* __restrict__ sv1, f32x4* __restrict__ sv2,
                  f32x4* __restrict__ svz, unsigned drop_thresh, unsigned drop_seed, float inv_keep,
                  unsigned* __restrict__ rstd_max) {
  __shared__ h8 lds[2 * WSPLIT + W3_LDS_FRAGS * 64];
  __shared__ __align__(16) float prm[4 * MPNN_H];
  lds_fill(lds, W11b, WSPLIT);
  lds_fill(lds + WSPLIT, W12, WSPLIT);
  for (int i = threadIdx.x; i < W3_LDS_FRAGS * 64; i += blockDim.x) {
    const int q = i >> 6, ln = i & 63;
    lds[2 * WSPLIT + i] = W13[(((q & 7) * NKK + (q >> 3)) * 2 + 0) * 64 + ln];
  }
  for (int i = threadIdx.x; i < MPNN_H; i += blockDim.x) {
    prm[i] = b12[i];
    prm[MPNN_H + i] = b13[i];
    prm[2 * MPNN_H + i] = ln_w[i];
    prm[3 * MPNN_H + i] = ln_b[i];
  }
  __syncthreads();
  const h8* W1s = lds;
  const h8* W2s = lds + WSPLIT;
  const h8* W3s = lds + 2 * WSPLIT;
  const float* b12s = prm;
  const float* b13s = prm + MPNN_H;
  const float* lnws = prm + 2 * MPNN_H;
  const float* lnbs = prm + 3 * MPNN_H;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane >> 4, col = lane & 15;
  constexpr int Kp = KT * 16;
  const ResidueRange rr = residue_range(N, MPNN_UPD_WAVES, wave);
  const int stride = rr.stride, Nend = rr.end;
  int node = rr.first;
  if (node >= Nend) return;

  auto setup = [&](int nd, const float* (&row)[KT]) {
    const int b = nd / L;
#pragma unroll
    for (int c = 0; c < KT; ++c)
      row[c] = Qb + (size_t)(b * L + E_idx[(size_t)nd * Kp + 16 * c + col]) * ldq + offQ + 4 * g;
  };

  const float* row[KT];
  setup(node, row);
  f32x4 h2[MPNN_NT][KT];  // carries the next residue's gathered Q' rows across iterations
#pragma unroll
  for (int t = 0; t < MPNN_NT; ++t)
#pragma unroll
    for (int c = 0; c < KT; ++c) h2[t][c] = row_tile(row[c], t);

  STAMP_DECL;
  float rmx = 0.f;
  for (; node < Nend; node += stride) {
    const int nxt = (node + stride < Nend) ? node + stride : Nend - 1;
    STAMP(7);
    const h8* W3u = W13;
    LAUNDER(W3u);
    const auto W3g = GLOBAL(W3u) + lane;
    f32x4* x = hE + (size_t)node * (MPNN_NT * KT * 64) + lane;
    f32x4 p[MPNN_NT];  // P'_i: needed only after layer 1, loaded during its last k-step
    const float* Pi = Pb + (size_t)node * ldp + offP + 4 * g;
    auto load_p = [&]() {
#pragma unroll
      for (int t = 0; t < MPNN_NT; ++t) p[t] = row_tile(Pi, t);
    };
    f32x4 h1[MPNN_NT][KT];
#pragma unroll
    for (int t = 0; t < MPNN_NT; ++t)
#pragma unroll
      for (int c = 0; c < KT; ++c) h1[t][c] = h2[t][c];
    layer1_stream<KT>(W1s, x, lane, h1, load_p);
    STAMP(0);
#pragma unroll
    for (int t = 0; t < MPNN_NT; ++t)
#pragma unroll
      for (int c = 0; c < KT; ++c) h1[t][c] += p[t];
    const size_t soff = (size_t)node * (MPNN_NT * KT * 64) + lane;
    BFrag hb[NKK][KT];
    if (SAVE) {  // saved pre-activation: six stores per k-step of the GELU + split instead of one burst of 24
#pragma unroll
      for (int kk = 0; kk < NKK; ++kk) {
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int c = 0; c < KT; ++c)
            __builtin_nontemporal_store(h1[2 * kk + u][c], &sv1[soff + ((2 * kk + u) * KT + c) * 64]);
#pragma unroll
        for (int c = 0; c < KT; ++c) split_pair(gelu4(h1[2 * kk][c]), gelu4(h1[2 * kk + 1][c]), hb[kk][c]);
        FENCE();
      }
    } else {
      tile_to_bfrag<MPNN_NT, KT, true>(h1, hb);
    }
    STAMP(1);
    const float* rown[KT];
    setup(nxt, rown);
    // ---- layer 2
#pragma unroll
    for (int t = 0; t < MPNN_NT; ++t) {
      const f32x4 bb = *reinterpret_cast<const f32x4*>(b12s + 16 * t + 4 * g);
#pragma unroll
      for (int c = 0; c < KT; ++c) h2[t][c] = bb;
    }
    lds_gemm<KT>(W2s, lane, hb, h2);
    STAMP(2);
    // W13 stream: lo halves (and the hi halves of the fragments that do not fit into LDS), D steps ahead
    constexpr int D = MPNN_UPD_RING;
    constexpr int NQ = NKK * MPNN_NT;
    h8 al[D + 1], ahx[NQ - W3_LDS_FRAGS];
    auto w3_load = [&](int qn) {
      const int f = ((qn & 7) * NKK + (qn >> 3)) * 2;
#ifndef MPNN_FAST
      al[qn % (D + 1)] = W3g[(f + 1) * 64];
#endif
      if (qn >= W3_LDS_FRAGS) ahx[qn - W3_LDS_FRAGS] = W3g[f * 64];
    };
#ifndef MPNN_ABL_W3LDS
#pragma unroll
    for (int q = 0; q < D; ++q) w3_load(q);
#endif
    // residual e -> layer 3's initial accumulator (loads fly under the GELU + split of h2); the training form
    // adds it after the dropout of the MLP output instead
#pragma unroll
    for (int kk = 0; kk < NKK; ++kk) {
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int c = 0; c < KT; ++c) {
          const int t = 2 * kk + u;
#ifdef MPNN_ABL_NORES
          h1[t][c] = f32x4{0.f, 0.f, 0.f, 0.f};
#else
          h1[t][c] = SAVE ? f32x4{0.f, 0.f, 0.f, 0.f} : x[(t * KT + c) * 64];
#endif
          if (SAVE) __builtin_nontemporal_store(h2[t][c], &sv2[soff + (t * KT + c) * 64]);
        }
#pragma unroll
      for (int c = 0; c < KT; ++c) split_pair(gelu4(h2[2 * kk][c]), gelu4(h2[2 * kk + 1][c]), hb[kk][c]);
      FENCE();
    }
    STAMP(3);
    // ---- layer 3.  step q = kk * 8 + to
#ifdef MPNN_ABL_W3LDS  // timing-only ablation build: never shipped
    split_gemm<KT, NKK, 0, MPNN_NT>(W2s, lane, hb, h1);
#else
#pragma unroll
    for (int kk = 0; kk < NKK; ++kk) {
#pragma unroll
      for (int to = 0; to < MPNN_NT; ++to) {
        const int q = kk * MPNN_NT + to, qn = q + D;
        if (qn < NQ) w3_load(qn);
        h8 a_hi;
        if (q < W3_LDS_FRAGS) a_hi = W3s[q * 64 + lane];
        else a_hi = ahx[q - W3_LDS_FRAGS];
#pragma unroll
        for (int c = 0; c < KT; ++c) h1[to][c] = mfma_split(a_hi, al[q % (D + 1)], hb[kk][c], h1[to][c]);
      }
      FENCE();
    }
#endif
    STAMP(4);
    // next residue's Q' rows -> h2 (carried to the next iteration), under the LayerNorm
    // training: the residual is added after the dropout; its tiles are all requested here, in one batch (layer 3's B
    // fragments are dead: registers), not one row tile at a time behind the gathers below
    f32x4 rs[SAVE ? MPNN_NT : 1][KT];
    if (SAVE) {
#pragma unroll
      for (int t = 0; t < MPNN_NT; ++t)
#pragma unroll
        for (int c = 0; c < KT; ++c) rs[SAVE ? t : 0][c] = x[(t * KT + c) * 64];
      FENCE();
    }
#pragma unroll
    for (int t = 0; t < MPNN_NT; ++t) {
#pragma unroll
      for (int c = 0; c < KT; ++c) h2[t][c] = row_tile(rown[c], t);
      const f32x4 bb = *reinterpret_cast<const f32x4*>(b13s + 16 * t + 4 * g);
#pragma unroll
      for (int c = 0; c < KT; ++c) {
        h1[t][c] += bb;
        if (SAVE) {
          if (drop_thresh)
            h1[t][c] = dropout4(h1[t][c], (unsigned)(soff + (t * KT + c) * 64) * 4u, drop_seed, drop_thresh, inv_keep);
          h1[t][c] += rs[SAVE ? t : 0][c];
          __builtin_nontemporal_store(h1[t][c], &svz[soff + (t * KT + c) * 64]);
        }
      }
      FENCE();
    }
    // LayerNorm statistics; then the next residue's first hE k-step is issued - BEFORE this residue's stores (vector
    // memory operations complete in issue order) - and the tiles are normalised and stored one by one
    float mean[KT], rstd[KT];
    tile_layernorm_stats<KT>(h1, mean, rstd);
    if (SAVE) {  // the largest 1 / sigma of the launch: the backward kernel's operand-scale bound (mpnn_enc_edge_update_bwd_fused)
#pragma unroll
      for (int c = 0; c < KT; ++c) rmx = fmaxf(rmx, rstd[c]);
    }
    f32x4* xo = SAVE ? hE_out + soff : x;
#pragma unroll
    for (int t = 0; t < MPNN_NT; ++t) {
#ifndef MPNN_ABL_NOLN
      const f32x4 gm = *reinterpret_cast<const f32x4*>(lnws + 16 * t + 4 * g);
      const f32x4 bt = *reinterpret_cast<const f32x4*>(lnbs + 16 * t + 4 * g);
#pragma unroll
      for (int c = 0; c < KT; ++c) h1[t][c] = (h1[t][c] - mean[c]) * rstd[c] * gm + bt;
#endif
#ifdef MPNN_ABL_NOSTORE
      if (h1[0][0].x == 1234.5f)
#endif
#pragma unroll
      for (int c = 0; c < KT; ++c) __builtin_nontemporal_store(h1[t][c], &xo[(t * KT + c) * 64]);
    }
    STAMP(5);
  }
  if (SAVE && rstd_max) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) rmx = fmaxf(rmx, __shfl_xor(rmx, o));
    if (lane == 0) atomicMax(rstd_max, (unsigned)__float_as_int(rmx));  // a maximum: order-independent
  }
  STAMP_FLUSH(8);
}

// ---------------------------------------------------------------------------
// message kernel, tile-wise form: a wave walks its residue's column tiles (16 neighbours) one at a time.
// Per-wave state shrinks from 2 x 96 to 2 x 32 accumulator registers, so 3-4 waves fit per SIMD and their
// VALU (GELU, f16 splitting), MFMA and memory phases overlap; the price is that every weight fragment is read
// from LDS once per column tile instead of once per residue (LDS ~30 % busy, the VALU stays the bound).
// Works for any K (loop count = ceil(K / 16)).
// ---------------------------------------------------------------------------
#define MPNN_MSGT_WAVES 12

template <bool DEC>
__global__ void __launch_bounds__(MPNN_MSGT_WAVES * 64)
k_message_tiles(const f32x4* __restrict__ hE, const float* __restrict__ Pbuf, int ldp, int offP,
                const float* __restrict__ Abuf, int lda, int offA, const float* __restrict__ Bbuf, int ldb, int offB,
                const int* __restrict__ rank, const int* __restrict__ E_idx, const float* __restrict__ mask,
                const float* __restrict__ edge_mask, const h8* __restrict__ W1b, const h8* __restrict__ W2,
                const float* __restrict__ b2, int N, int L, int K, int KT, float* __restrict__ gsum,
                float* __restrict__ cnt, int w_valid) {
  __shared__ h8 lds[2 * WSPLIT];
  lds_fill(lds, W1b, WSPLIT);
  lds_fill(lds + WSPLIT, W2, WSPLIT);
  __syncthreads();
  const h8* W1s = lds;
  const h8* W2s = lds + WSPLIT;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane >> 4, col = lane & 15;
  const int Kp = KT * 16;
  const ResidueRange rr = residue_range(N, MPNN_MSGT_WAVES, wave);

  for (int node = rr.first; node < rr.end; node += rr.stride) {
    const int b = node / L;
    const float mi = mask[node];
    int ri = 0;
    if (DEC) ri = rank[node];
    const float* Pi = Pbuf + (size_t)node * ldp + offP + 4 * g;
    f32x4 gacc[MPNN_NT][1];
#pragma unroll
    for (int t = 0; t < MPNN_NT; ++t) gacc[t][0] = f32x4{0.f, 0.f, 0.f, 0.f};
    float wsum = 0.f;
#pragma unroll 1
    for (int c = 0; c < KT; ++c) {
      LAUNDER(b2);
      const int e = 16 * c + col;
      const int jn = b * L + E_idx[(size_t)node * Kp + e];
      float em = 1.0f;
      if (edge_mask && e < K) em = edge_mask[(size_t)node * K + e];
      const float* row;
      float sc, w;
      if (DEC) {
        const bool before = ri > rank[jn];
        row = before ? (Abuf + (size_t)jn * lda + offA + 4 * g) : (Bbuf + (size_t)jn * ldb + offB + 4 * g);
        sc = mi * em;
        w = e < K ? (w_valid ? em : 1.0f) : 0.0f;
      } else {
        row = Abuf + (size_t)jn * lda + offA + 4 * g;
        sc = 1.0f;
        w = e < K ? mi * mask[jn] * em : 0.0f;
      }
      wsum += w;
      const f32x4* x = hE + ((size_t)node * MPNN_NT * KT + c) * 64 + lane;  // tile t of this column: x[t * KT * 64]
      f32x4 xt[MPNN_NT];
#pragma unroll
      for (int t = 0; t < MPNN_NT; ++t) xt[t] = x[(size_t)t * KT * 64];
      f32x4 h1[MPNN_NT][1];
#pragma unroll
      for (int t = 0; t < MPNN_NT; ++t) h1[t][0] = row_tile(row, t);
      // ---- layer 1
#pragma unroll
      for (int kk = 0; kk < NKK; ++kk) {
        BFrag bf[1];
        split_pair(xt[2 * kk], xt[2 * kk + 1], bf[0]);
        split_kstep<1, NKK, 0, MPNN_NT>(W1s, kk, lane, bf, h1);
      }
#pragma unroll
      for (int t = 0; t < MPNN_NT; ++t) {
        const f32x4 p = row_tile(Pi, t);
        h1[t][0] = DEC ? (h1[t][0] * sc + p) : (h1[t][0] + p);
      }
      BFrag hb[NKK][1];
      tile_to_bfrag<MPNN_NT, 1, true>(h1, hb);
      // ---- layer 2
      f32x4 y[MPNN_NT][1];
#pragma unroll
      for (int t = 0; t < MPNN_NT; ++t) y[t][0] = *reinterpret_cast<const f32x4*>(b2 + 16 * t + 4 * g);
#pragma unroll
      for (int kk = 0; kk < NKK; ++kk) split_kstep<1, NKK, 0, MPNN_NT>(W2s, kk, lane, hb[kk], y);
#pragma unroll
      for (int t = 0; t < MPNN_NT; ++t) gacc[t][0] += gelu4(y[t][0]) * w;
    }
    float* gout = gsum + (size_t)node * MPNN_H;
#pragma unroll
    for (int t = 0; t < MPNN_NT; ++t) {
      f32x4 sm = gacc[t][0];
      sm = column_sum4(sm);
      if (col == 0) *reinterpret_cast<f32x4*>(gout + 16 * t + 4 * g) = sm;
    }
    if (!DEC) {
      wsum = column_sum(wsum);
      if (lane == 0) cnt[node] = wsum;
    }
  }
}

// ---------------------------------------------------------------------------
// decoder message for the sampler's dependency levels: a level holds only a few hundred residues, so the unit of
// work is one COLUMN TILE (16 neighbours) of one listed residue per wave - three times the parallelism of the
// residue-per-wave kernels and a third of their latency, which is what bounds a level.  Partial neighbour sums
// meet in gsum by fp32 atomics (gsum rows must be zero on entry; mpnn_node_update_subset re-zeroes what it reads).
// Samples share one encoder output: hE, E_idx, edge_mask and the encoder rows Bbuf are indexed with
// (residue % enc_mod), everything else (P, A rows, rank, mask, gsum) with the residue id itself.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(MPNN_MSGT_WAVES * 64)
k_message_level(const f32x4* __restrict__ hE, const float* __restrict__ Pbuf, int ldp, int offP,
                const float* __restrict__ Abuf, int lda, int offA, const float* __restrict__ Bbuf, int ldb, int offB,
                const int* __restrict__ rank, const int* __restrict__ E_idx, const float* __restrict__ mask,
                const float* __restrict__ edge_mask, const h8* __restrict__ W1b, const h8* __restrict__ W2,
                const float* __restrict__ b2, const int* __restrict__ node_list, int n_list, int enc_mod, int L, int K,
                int KT, float* __restrict__ gsum) {
  __shared__ h8 lds[2 * WSPLIT];
#ifndef MPNN_ABL_LEVEL_NOFILL  // timing-only ablation build: never shipped
  lds_fill(lds, W1b, WSPLIT);
  lds_fill(lds + WSPLIT, W2, WSPLIT);
#endif
  __syncthreads();
  const h8* W1s = lds;
  const h8* W2s = lds + WSPLIT;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane >> 4, col = lane & 15;
  const int Kp = KT * 16;
  const int stride = gridDim.x * MPNN_MSGT_WAVES;
  for (int item = blockIdx.x * MPNN_MSGT_WAVES + wave; item < n_list * KT; item += stride) {
    LAUNDER(b2);
    const int node = node_list[item / KT], c = item % KT;
    const int sh = node % enc_mod, b = node / L;
    const int e = 16 * c + col;
    const int jl = E_idx[(size_t)sh * Kp + e];
    const int jn = b * L + jl;
    float em = 1.0f;
    if (edge_mask && e < K) em = edge_mask[(size_t)sh * K + e];
    const bool before = rank[node] > rank[jn];
    const float* row = before ? (Abuf + (size_t)jn * lda + offA + 4 * g)
                              : (Bbuf + (size_t)((sh / L) * L + jl) * ldb + offB + 4 * g);
    const float sc = mask[node] * em;
    const float w = e < K ? 1.0f : 0.0f;
    const float* Pi = Pbuf + (size_t)node * ldp + offP + 4 * g;
    const f32x4* x = hE + ((size_t)sh * MPNN_NT * KT + c) * 64 + lane;
    f32x4 xt[MPNN_NT], h1[MPNN_NT][1];
#pragma unroll
    for (int t = 0; t < MPNN_NT; ++t) {
      xt[t] = x[(size_t)t * KT * 64];
      h1[t][0] = row_tile(row, t);
    }
#pragma unroll
    for (int kk = 0; kk < NKK; ++kk) {
      BFrag bf[1];
      split_pair(xt[2 * kk], xt[2 * kk + 1], bf[0]);
      split_kstep<1, NKK, 0, MPNN_NT>(W1s, kk, lane, bf, h1);
    }
#pragma unroll
    for (int t = 0; t < MPNN_NT; ++t) h1[t][0] = h1[t][0] * sc + row_tile(Pi, t);
    BFrag hb[NKK][1];
    tile_to_bfrag<MPNN_NT, 1, true>(h1, hb);
    f32x4 y[MPNN_NT][1];
#pragma unroll
    for (int t = 0; t < MPNN_NT; ++t) y[t][0] = *reinterpret_cast<const f32x4*>(b2 + 16 * t + 4 * g);
#pragma unroll
    for (int kk = 0; kk < NKK; ++kk) split_kstep<1, NKK, 0, MPNN_NT>(W2s, kk, lane, hb[kk], y);
    // the tile's column sums go to its own row gsum[node][c] (plain stores; the node kernel adds a residue's tiles in
    // tile order: no atomics, no re-zeroing, the same bits on every run)
    float* gout = gsum + ((size_t)node * 4 + c) * MPNN_H + 4 * g;
#pragma unroll
    for (int t = 0; t < MPNN_NT; ++t) {
      f32x4 sm = gelu4(y[t][0]) * w;
      sm = column_sum4(sm);
      if (col == 0) *reinterpret_cast<f32x4*>(gout + 16 * t) = sm;
    }
  }
}

// ---------------------------------------------------------------------------
// One decoder layer for a dependency level in ONE launch: a workgroup (12 waves) owns a tile of LV_NODES listed residues,
// computes their message tiles as k_message_level does - the neighbour sums meet in LDS instead of global
// atomics - and then eight of its waves run the node update on the tile (W3 by output tiles, FFN by hidden eighths,
// projections by halves; cf. k_node_update_level in mpnn_node.hip), with the exchange buffer
// laid over the message weights that are no longer needed.  Saves a launch, the global gsum round trip and its
// re-zeroing per layer and level.
// ---------------------------------------------------------------------------
#ifndef MPNN_LEVEL_RING
#define MPNN_LEVEL_RING 6  // weight fragments in flight per wave in the level kernels' node phase: 64 x 512 samples take
                           // 7.94 / 7.74 / 7.57 / 7.69 / 8.27 ms at 4 / 5 / 6 / 8 / 12 (12 waves per workgroup leave 168
                           // registers a wave: a deeper ring spills)
#endif
// residues per workgroup: 4 x 3 column tiles = one message item per wave (items of one wave run one after the other,
// each with its own chain of dependent loads, so more residues per workgroup only lengthen the launch)
constexpr int LV_NODES = 4;  // at most; small levels use fewer per workgroup (level_nodes_per_wg) to spread over more CUs
// ---- pieces of a level kernel.  A workgroup of MPNN_MSGT_WAVES = 12 waves owns LV_NODES = 4 listed residues; with
// K <= 48 (at most 3 column tiles) that is at most ONE message item (local residue, column tile) per wave.

// a 64 KiB weight matrix on its way into LDS: the loads are issued up front (issue), whatever else the layer has to
// fetch - index chain, neighbour rows, W3 fragments - is issued behind them, and only then are they written (commit)
constexpr int LV_FILL = (WSPLIT + MPNN_MSGT_WAVES * 64 - 1) / (MPNN_MSGT_WAVES * 64);
struct LdsFill {
  h8 v[LV_FILL];
};
__device__ __forceinline__ void lds_fill_issue(LdsFill& f, const h8* __restrict__ src) {
#pragma unroll
  for (int u = 0; u < LV_FILL; ++u) {
    const int i = threadIdx.x + u * (MPNN_MSGT_WAVES * 64);
    if (i < WSPLIT) f.v[u] = src[i];
  }
}
__device__ __forceinline__ void lds_fill_commit(const LdsFill& f, h8* dst) {
#pragma unroll
  for (int u = 0; u < LV_FILL; ++u) {
    const int i = threadIdx.x + u * (MPNN_MSGT_WAVES * 64);
    if (i < WSPLIT) dst[i] = f.v[u];
  }
}

// what a wave knows about its message item before any layer runs: indices, scale / weight, where its hE tile is
struct LevelItem {
  bool has, before;
  int nl, node, c;  // local residue, its id, column tile of this item
  float sc, w;
  size_t rowA, rowB;  // row numbers of the gathered neighbour row in the A buffer / in the encoder rows
  const f32x4* x;     // this lane's slice of the item's hE tile (feature tile t at x[t * xs])
  int xs;
};

__device__ __forceinline__ void
level_item_setup(LevelItem& it, const f32x4* __restrict__ hE, const int* __restrict__ rank, const int* __restrict__ E_idx,
                 const float* __restrict__ mask, const float* __restrict__ edge_mask, const int* __restrict__ node_list,
                 int tile, int lvn, int nlive, int enc_mod, int L, int K, int KT) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = lane & 15;
  const int Kp = KT * 16;
  it.has = wave < nlive * KT;
  const int item = it.has ? wave : 0;
  const int c = item % KT;
  it.c = c;
  it.nl = item / KT;
  it.node = node_list[tile * lvn + it.nl];
  const int sh = it.node % enc_mod, b = it.node / L;
  const int e = 16 * c + col;
  const int jl = E_idx[(size_t)sh * Kp + e];
  const int jn = b * L + jl;
  float em = 1.0f;
  if (edge_mask && e < K) em = edge_mask[(size_t)sh * K + e];
  it.before = rank[it.node] > rank[jn];
  it.rowA = (size_t)jn;
  it.rowB = (size_t)((sh / L) * L + jl);
  it.sc = mask[it.node] * em;
  it.w = e < K ? 1.0f : 0.0f;
  it.x = hE + ((size_t)sh * MPNN_NT * KT + c) * 64 + lane;
  it.xs = KT * 64;
}

// the item's hE tile and its gathered neighbour row for one layer (issued early: both depend on earlier levels only)
__device__ __forceinline__ void
level_rows_load(const LevelItem& it, const float* Abuf, int lda, int offA, const float* __restrict__ Bbuf, int ldb,
                int offB, f32x4 (&xt)[MPNN_NT], f32x4 (&h1)[MPNN_NT][1]) {
  const int g = (threadIdx.x & 63) >> 4;
  const float* row = it.before ? (Abuf + it.rowA * lda + offA + 4 * g) : (Bbuf + it.rowB * ldb + offB + 4 * g);
#pragma unroll
  for (int t = 0; t < MPNN_NT; ++t) {
    xt[t] = it.x[(size_t)t * it.xs];
    h1[t][0] = row_tile(row, t);
  }
}

// message MLP of the item (weights in LDS); the item's column sums go to its own row gsl[LV_NODES c + nl] (plain
// stores: level_node adds a residue's column tiles in tile order - no atomics, the same bits on every run)
__device__ __forceinline__ void
level_message(const LevelItem& it, const h8* W1s, const h8* W2s, const float* __restrict__ b2, const float* Pbuf,
              int ldp, int offP, const f32x4 (&xt)[MPNN_NT], f32x4 (&h1)[MPNN_NT][1], float (*gsl)[MPNN_H]) {
  const int lane = threadIdx.x & 63;
  const int g = lane >> 4, col = lane & 15;
  LAUNDER(b2);
  const float* Pi = Pbuf + (size_t)it.node * ldp + offP + 4 * g;
  f32x4 p[MPNN_NT];
#pragma unroll
  for (int t = 0; t < MPNN_NT; ++t) p[t] = row_tile(Pi, t);
#pragma unroll
  for (int kk = 0; kk < NKK; ++kk) {
    BFrag bf[1];
    split_pair(xt[2 * kk], xt[2 * kk + 1], bf[0]);
    split_kstep<1, NKK, 0, MPNN_NT>(W1s, kk, lane, bf, h1);
  }
#pragma unroll
  for (int t = 0; t < MPNN_NT; ++t) h1[t][0] = h1[t][0] * it.sc + p[t];
  BFrag hb[NKK][1];
  tile_to_bfrag<MPNN_NT, 1, true>(h1, hb);
  f32x4 y[MPNN_NT][1];
#pragma unroll
  for (int t = 0; t < MPNN_NT; ++t) y[t][0] = *reinterpret_cast<const f32x4*>(b2 + 16 * t + 4 * g);
#pragma unroll
  for (int kk = 0; kk < NKK; ++kk) split_kstep<1, NKK, 0, MPNN_NT>(W2s, kk, lane, hb[kk], y);
#pragma unroll
  for (int t = 0; t < MPNN_NT; ++t) {
    f32x4 sm = gelu4(y[t][0]) * it.w;
    sm = column_sum4(sm);
    if (col == 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r) gsl[LV_NODES * it.c + it.nl][16 * t + 4 * g + r] = sm[r];
    }
  }
}

// W3's fragments of a node wave (output tile = wave): NKK (hi, lo) pairs, loaded ahead of the message phase
struct W3Frags {
  h8 hi[NKK], lo[NKK];
};
__device__ __forceinline__ void w3_prefetch(W3Frags& f, const h8* __restrict__ W3) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const h8* Wl = W3 + (size_t)wave * NKK * 2 * 64 + lane;
#pragma unroll
  for (int kk = 0; kk < NKK; ++kk) {
    f.hi[kk] = Wl[(kk * 2 + 0) * 64];
#ifndef MPNN_FAST
    f.lo[kk] = Wl[(kk * 2 + 1) * 64];
#endif
  }
}

// node phase on eight waves: W3 by output tile, the FFN by hidden eighths (64 units: four rows-tiles of W_in, one
// 64-column block of W_out, packed as eight [8][2] matrices by the host), projections by halves.  ex = the exchange
// buffer [NODE_WAVES][8][64] x 16 B = 64 KiB laid over the (dead) first message weight matrix in LDS.
constexpr int LV_NODE_WAVES = 8;
__device__ __forceinline__ void
level_node(h8* lds, float (*gsl)[MPNN_H], const W3Frags& w3, const int* __restrict__ node_list, int n_list, int tile, int lvn,
           const float* __restrict__ mask, const float* hV_in, float cnt_const, const float* __restrict__ b3,
           const float* __restrict__ ln1w, const float* __restrict__ ln1b, const h8* __restrict__ Win,
           const float* __restrict__ bin, const h8* __restrict__ Wout, const float* __restrict__ bout,
           const float* __restrict__ ln2w, const float* __restrict__ ln2b, float inv_scale,
           const h8* __restrict__ Wproj, const float* __restrict__ bproj, int nproj, float* hV_out, float* PQ,
           f32x4 (&h)[MPNN_NT][1]) {  // h: on return the columns' new node features (node waves)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane >> 4, col = lane & 15;
  f32x4 (*ex)[MPNN_NT][64] = reinterpret_cast<f32x4 (*)[MPNN_NT][64]>(lds);
  const bool act = wave < LV_NODE_WAVES;
  const int idx = tile * lvn + col;
  const bool live = col < lvn && idx < n_list;
  const int nd = node_list[live ? idx : n_list - 1];
  f32x4 o[MPNN_NT][1];
  BFrag xb[NKK][1];
  float mk = 1.0f;
  if (act) {
    mk = mask[nd];  // requested with the rows: at its point of use it was one more round trip on the chain
    f32x4 gs[MPNN_NT][1];
#pragma unroll
    for (int t = 0; t < MPNN_NT; ++t) {
      h[t][0] = *reinterpret_cast<const f32x4*>(hV_in + (size_t)nd * MPNN_H + 16 * t + 4 * g);
      gs[t][0] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (col < LV_NODES)  // column tiles 0, 1, 2 of the residue, in that order (rows of absent tiles are zero)
        gs[t][0] = (*reinterpret_cast<const f32x4*>(&gsl[col][16 * t + 4 * g]) +
                    *reinterpret_cast<const f32x4*>(&gsl[LV_NODES + col][16 * t + 4 * g])) +
                   *reinterpret_cast<const f32x4*>(&gsl[2 * LV_NODES + col][16 * t + 4 * g]);
    }
    tile_to_bfrag<MPNN_NT, 1, false>(gs, xb);
    f32x4 d1 = *reinterpret_cast<const f32x4*>(b3 + 16 * wave + 4 * g) * cnt_const;
#pragma unroll
    for (int kk = 0; kk < NKK; ++kk) d1 = mfma_split(w3.hi[kk], w3.lo[kk], xb[kk][0], d1);
    ex[0][wave][lane] = d1;
  }
  __syncthreads();
  if (act) {
#pragma unroll
    for (int t = 0; t < MPNN_NT; ++t) h[t][0] += ex[0][t][lane] * inv_scale;
    tile_layernorm<1>(h, ln1w, ln1b, lane);
    tile_to_bfrag<MPNN_NT, 1, false>(h, xb);
    f32x4 hid[4][1];
#pragma unroll
    for (int t = 0; t < 4; ++t) hid[t][0] = *reinterpret_cast<const f32x4*>(bin + 64 * wave + 16 * t + 4 * g);
#pragma unroll
    for (int t = 0; t < MPNN_NT; ++t) o[t][0] = f32x4{0.f, 0.f, 0.f, 0.f};
    split_gemm_stream<1, NKK, 4, MPNN_LEVEL_RING>(Win + (size_t)(4 * wave) * NKK * 2 * 64 + lane, xb, hid);
    BFrag hb[2][1];
    tile_to_bfrag<4, 1, true>(hid, hb);
    split_gemm_stream<1, 2, MPNN_NT, MPNN_LEVEL_RING>(Wout + (size_t)wave * (WSPLIT / 2) + lane, hb, o);
  }
  __syncthreads();  // every wave has read the W3 exchange
  if (act) {
#pragma unroll
    for (int t = 0; t < MPNN_NT; ++t) ex[wave][t][lane] = o[t][0];
  }
  __syncthreads();
  if (act) {
#pragma unroll
    for (int t = 0; t < MPNN_NT; ++t) {
      f32x4 sum = (ex[0][t][lane] + ex[1][t][lane]) + (ex[2][t][lane] + ex[3][t][lane]);
      sum += (ex[4][t][lane] + ex[5][t][lane]) + (ex[6][t][lane] + ex[7][t][lane]);
      o[t][0] = h[t][0] + *reinterpret_cast<const f32x4*>(bout + 16 * t + 4 * g) + sum;
    }
    tile_layernorm<1>(o, ln2w, ln2b, lane);
#pragma unroll
    for (int t = 0; t < MPNN_NT; ++t) {
      h[t][0] = o[t][0] * mk;
      if (live && wave == 0) *reinterpret_cast<f32x4*>(hV_out + (size_t)nd * MPNN_H + 16 * t + 4 * g) = h[t][0];
    }
    BFrag pb[NKK][1];
    tile_to_bfrag<MPNN_NT, 1, false>(h, pb);
    const int ld = nproj * MPNN_H;
    for (int u = wave; u < 2 * nproj; u += LV_NODE_WAVES) {
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
}

#include "mpnn_sample.h"

// HEAD: the launch of the LAST decoder layer also draws the residues' letters and folds their sequence terms into the
// layers' gathered rows (what mpnn_sample_subset does in a launch of its own): wave 0 still holds the new node
// features in the layout the head's GEMM wants.
template <bool HEAD>
__global__ void __launch_bounds__(MPNN_MSGT_WAVES * 64)
k_dec_layer_level(const f32x4* __restrict__ hE, const float* __restrict__ Pbuf, int ldp, int offP,
                  const float* __restrict__ Abuf, int lda, int offA, const float* __restrict__ Bbuf, int ldb, int offB,
                  const int* __restrict__ rank, const int* __restrict__ E_idx, const float* __restrict__ mask,
                  const float* __restrict__ edge_mask, const h8* __restrict__ W1b, const h8* __restrict__ W2,
                  const float* __restrict__ b2, const int* __restrict__ node_list, int n_list, int enc_mod, int L, int K,
                  int KT, const float* __restrict__ hV_in, float cnt_const, const h8* __restrict__ W3,
                  const float* __restrict__ b3, const float* __restrict__ ln1w, const float* __restrict__ ln1b,
                  const h8* __restrict__ Win, const float* __restrict__ bin, const h8* __restrict__ Wout,
                  const float* __restrict__ bout, const float* __restrict__ ln2w, const float* __restrict__ ln2b,
                  float inv_scale, const h8* __restrict__ Wproj, const float* __restrict__ bproj, int nproj,
                  float* __restrict__ hV_out, float* __restrict__ PQ, int lvn, LevelHead head) {
  __shared__ h8 lds[2 * WSPLIT];
  __shared__ float gsl[16][MPNN_H];
  // the node phase's bias / LayerNorm vectors, copied to LDS while the message phase runs: read from global memory at
  // their point of use each was one more exposed round trip on the level's dependency chain
  constexpr int LPRM_PROJ = 10 * MPNN_H, LPRM_MAXPROJ = 8;  // the launcher checks nproj <= 8
  __shared__ __align__(16) float lprm[LPRM_PROJ + LPRM_MAXPROJ * MPNN_H];
  const int tile = blockIdx.x;
  const int nlive = min(lvn, n_list - tile * lvn);
  // MPNN_ABL_LEVEL_*: timing-only ablation builds (never shipped): what each phase of a level costs
  LdsFill f1, f2;
  lds_fill_issue(f1, W1b);
  lds_fill_issue(f2, W2);
  for (int i = threadIdx.x; i < 16 * MPNN_H; i += blockDim.x) (&gsl[0][0])[i] = 0.f;
  {
    float pv[3];
    const int t = threadIdx.x;  // 768 threads: at most three floats each
    const int nprm = LPRM_PROJ + nproj * MPNN_H;
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      const int i = t + u * (MPNN_MSGT_WAVES * 64);
      pv[u] = 0.f;
      if (i < nprm) {
        const int blk = i >> 7, j = i & 127;
        const float* src = blk == 0 ? b3 : blk == 1 ? ln1w : blk == 2 ? ln1b : blk < 7 ? bin + 128 * (blk - 3)
                         : blk == 7 ? bout : blk == 8 ? ln2w : blk == 9 ? ln2b : bproj + 128 * (blk - 10);
        pv[u] = src[j];
      }
    }
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      const int i = t + u * (MPNN_MSGT_WAVES * 64);
      if (i < nprm) lprm[i] = pv[u];
    }
  }
  LevelItem it;
  level_item_setup(it, hE, rank, E_idx, mask, edge_mask, node_list, tile, lvn, nlive, enc_mod, L, K, KT);
  f32x4 xt[MPNN_NT], h1[MPNN_NT][1];
  level_rows_load(it, Abuf, lda, offA, Bbuf, ldb, offB, xt, h1);
  W3Frags w3;
  if ((threadIdx.x >> 6) < LV_NODE_WAVES) w3_prefetch(w3, W3);
#ifndef MPNN_ABL_LEVEL_NOFILL
  lds_fill_commit(f1, lds);
  lds_fill_commit(f2, lds + WSPLIT);
#endif
  __syncthreads();
#ifndef MPNN_ABL_LEVEL_NOMSG
  if (it.has) level_message(it, lds, lds + WSPLIT, b2, Pbuf, ldp, offP, xt, h1, gsl);
#else
  if (it.has && xt[0].x == 123.f && h1[1][0].x == 7.f) gsl[0][0] = 1.f;
#endif
  __syncthreads();  // neighbour sums complete; the message weights in LDS are dead from here on
#ifndef MPNN_ABL_LEVEL_NONODE
  f32x4 hnew[MPNN_NT][1];
  level_node(lds, gsl, w3, node_list, n_list, tile, lvn, mask, hV_in, cnt_const, lprm, lprm + MPNN_H, lprm + 2 * MPNN_H, Win,
             lprm + 3 * MPNN_H, Wout, lprm + 7 * MPNN_H, lprm + 8 * MPNN_H, lprm + 9 * MPNN_H, inv_scale, Wproj,
             lprm + LPRM_PROJ, nproj, hV_out, PQ, hnew);
  if (HEAD) {
    // gsl (the neighbour sums) was last read at the top of the node phase, three barriers ago
    float (*pr)[33] = reinterpret_cast<float (*)[33]>(&gsl[0][0]);
    int* pick = reinterpret_cast<int*>(&gsl[0][0] + 16 * 33);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (wave == 0) {
      const int col = lane & 15, idx = tile * lvn + col;
      const bool live = col < lvn && idx < n_list;
      const int nd = node_list[live ? idx : n_list - 1];
      sample_columns_h(hnew, head.Wp, head.bias, head.logit_bias, head.n_letters, nd, live, head.inv_temperature,
                       head.seed, head.designable, head.S_in, head.S_out, head.log_probs, pr, pick, lane);
    }
    __syncthreads();
    fold_sequence_rows<MPNN_MSGT_WAVES, 2>(
        [&](int l, int n) { return head.rows.A[l] + (size_t)n * head.rows.ld[l] + head.rows.off[l]; }, head.rows.n_layers,
        head.TS, head.ld_ts, node_list + tile * lvn, nlive, pick, wave, lane);
  }
#else
  if (w3.hi[0][0] == (_Float16)77.f && gsl[1][1] == 5.f) hV_out[0] = 1.f;
#endif
}

MPNN_KERNELS_END

// ===========================================================================
// C ABI launchers
// ===========================================================================
// residues per workgroup of the level kernels: a level is latency-bound, so while the chip has idle CUs every residue
// gets a workgroup of its own (its three message items then run on three SIMDs instead of sharing one with eight
// others); large levels (many samples) share workgroups so that the per-workgroup weight streams stay affordable
#ifndef MPNN_LEVEL_SPREAD
#define MPNN_LEVEL_SPREAD 256  // one workgroup per CU at most for a residue of its own; measured against 128 / 192 / 384
#endif                         // at 16 .. 128 samples of L = 512: +2 .. +5 % up to 48 samples, equal above; 384: -30 %
static int level_nodes_per_wg(int n_list) {
  if (n_list <= MPNN_LEVEL_SPREAD) return 1;
  if (n_list <= 2 * MPNN_LEVEL_SPREAD) return 2;
  return LV_NODES;
}

extern "C" int MPNN_API(mpnn_enc_message)(const float* hE_frag, const float* PQ, int ldpq, int offP, int offQ,
                                const int32_t* E_idx32, const float* mask, const float* edge_mask,
                                const float* W1b_packed, const float* W2_packed, const float* b2, int B, int L, int K,
                                float* gsum, float* cnt, void* stream) {
  MPNN_CHECK_ARG(hE_frag && PQ && E_idx32 && mask && W1b_packed && W2_packed && b2 && gsum && cnt,
                 "mpnn_enc_message: null pointer");
  // ldpq == 0: ONE row serves every node (h_V = 0 before the first layer: P | Q are the bias row)
  MPNN_CHECK_ARG(B > 0 && L > 0 && K >= 1 && K <= 64 && (ldpq >= 128 || ldpq == 0) && mult4(ldpq, offP, offQ),
                 "mpnn_enc_message: bad sizes (strides/offsets must be multiples of 4 floats)");
  const int N = B * L, KT = (K + 15) / 16;
#ifdef MPNN_STAMP
  stamp_report("message: layer1 | gelu1+split | layer2 | save | - | - | reduce + next loads | cnt+loop top", 0);
#endif
  if (KT > 3) {  // four column tiles: the tile-wise kernel (one 16-neighbour tile per wave)
    dim3 grid(edge_grid(N, MPNN_MSGT_WAVES)), block(MPNN_MSGT_WAVES * 64);
    hipLaunchKernelGGL((k_message_tiles<false>), grid, block, 0, (hipStream_t)stream, (const f32x4*)hE_frag, PQ, ldpq,
                       offP, PQ, ldpq, offQ, (const float*)nullptr, 0, 0, (const int*)nullptr, E_idx32, mask,
                       edge_mask, (const h8*)W1b_packed, (const h8*)W2_packed, b2, N, L, K, KT, gsum, cnt, 0);
    MPNN_LAUNCH_CHECK();
  }
  dim3 grid(edge_grid(N, MPNN_MSG_WAVES)), block(MPNN_MSG_WAVES * 64);
  KT_DISPATCH(KT, hipLaunchKernelGGL((k_message<kKT, false, false>), grid, block, 0, (hipStream_t)stream,
                                     (const f32x4*)hE_frag, PQ, ldpq, offP, PQ, ldpq, offQ, (const float*)nullptr, 0, 0,
                                     (const int*)nullptr, E_idx32, mask, edge_mask, (const h8*)W1b_packed,
                                     (const h8*)W2_packed, b2, N, L, K, gsum, cnt, (f32x4*)nullptr, (f32x4*)nullptr, N, 0));
  MPNN_LAUNCH_CHECK();
}

extern "C" int MPNN_API(mpnn_enc_edge_update)(float* hE_frag, const float* PQ, int ldpq, int offP, int offQ,
                                    const int32_t* E_idx32, const float* W11b_packed, const float* W12_packed,
                                    const float* b12, const float* W13_packed, const float* b13, const float* ln_w,
                                    const float* ln_b, int B, int L, int K, void* stream) {
  MPNN_CHECK_ARG(hE_frag && PQ && E_idx32 && W11b_packed && W12_packed && b12 && W13_packed && b13 && ln_w && ln_b,
                 "mpnn_enc_edge_update: null pointer");
  MPNN_CHECK_ARG(B > 0 && L > 0 && K >= 1 && K <= 64 && ldpq >= 128 && mult4(ldpq, offP, offQ),
                 "mpnn_enc_edge_update: bad sizes (strides/offsets must be multiples of 4 floats)");
  const int N = B * L, KT = (K + 15) / 16;
#ifdef MPNN_STAMP
  stamp_report("edge update: layer1 | gelu1+split | layer2 | res load+gelu2+split | layer3 | LN+store | .. | loop top", 8);
#endif
  dim3 grid(edge_grid(N, MPNN_UPD_WAVES)), block(MPNN_UPD_WAVES * 64);
  KT_DISPATCH(KT, hipLaunchKernelGGL((k_enc_edge_update<kKT, false>), grid, block, 0, (hipStream_t)stream,
                                     (f32x4*)hE_frag, PQ, ldpq, offP, PQ, ldpq, offQ, E_idx32, (const h8*)W11b_packed,
                                     (const h8*)W12_packed, b12, (const h8*)W13_packed, b13, ln_w, ln_b, N, L,
                                     (f32x4*)nullptr, (f32x4*)nullptr, (f32x4*)nullptr, (f32x4*)nullptr, 0u, 0u, 1.0f,
                                     (unsigned*)nullptr));
  MPNN_LAUNCH_CHECK();
}

static int launch_dec_message(const char* who, const float* hE_frag, const float* PQ, int ldpq, int offPa, int offA,
                              const float* PVenc, int ldenc, int offEnc, const int32_t* rank, const int32_t* E_idx32,
                              const float* mask, const float* edge_mask, const float* W1b_packed,
                              const float* W2_packed, const float* b2, int B, int n_struct, int L, int K, float* gsum,
                              void* stream, int w_valid = 0) {
  MPNN_CHECK_ARG(hE_frag && PQ && PVenc && rank && E_idx32 && mask && W1b_packed && W2_packed && b2 && gsum,
                 "mpnn_dec_message: null pointer");
  MPNN_CHECK_ARG(B > 0 && L > 0 && K >= 1 && K <= 64 && mult4(ldpq, offPa, offA, ldenc, offEnc),
                 "mpnn_dec_message: bad sizes (strides/offsets must be multiples of 4 floats)");
  MPNN_CHECK_ARG(n_struct >= 1 && n_struct <= B && B % n_struct == 0 && (n_struct == B || K <= 48),
                 "mpnn_dec_message_shared: rows must be a multiple of the structures (and K <= 48)");
  (void)who;
  const int N = B * L, KT = (K + 15) / 16;
  if (KT > 3) {
    dim3 grid(edge_grid(N, MPNN_MSGT_WAVES)), block(MPNN_MSGT_WAVES * 64);
    hipLaunchKernelGGL((k_message_tiles<true>), grid, block, 0, (hipStream_t)stream, (const f32x4*)hE_frag, PQ, ldpq,
                       offPa, PQ, ldpq, offA, PVenc, ldenc, offEnc, rank, E_idx32, mask, edge_mask,
                       (const h8*)W1b_packed, (const h8*)W2_packed, b2, N, L, K, KT, gsum, (float*)nullptr, w_valid);
    MPNN_LAUNCH_CHECK();
  }
  dim3 grid(edge_grid(N, MPNN_MSG_WAVES)), block(MPNN_MSG_WAVES * 64);
  KT_DISPATCH(KT, hipLaunchKernelGGL((k_message<kKT, true, false>), grid, block, 0, (hipStream_t)stream,
                                     (const f32x4*)hE_frag, PQ, ldpq, offPa, PQ, ldpq, offA, PVenc, ldenc, offEnc, rank,
                                     E_idx32, mask, edge_mask, (const h8*)W1b_packed, (const h8*)W2_packed, b2,
                                     N, L, K, gsum, (float*)nullptr, (f32x4*)nullptr, (f32x4*)nullptr, n_struct * L,
                                     w_valid));
  MPNN_LAUNCH_CHECK();
}

extern "C" int MPNN_API(mpnn_dec_message)(const float* hE_frag, const float* PQ, int ldpq, int offPa, int offA,
                                const float* PVenc, int ldenc, int offEnc, const int32_t* rank,
                                const int32_t* E_idx32, const float* mask, const float* edge_mask,
                                const float* W1b_packed, const float* W2_packed, const float* b2, int B, int L, int K,
                                float* gsum, void* stream) {
  return launch_dec_message("mpnn_dec_message", hE_frag, PQ, ldpq, offPa, offA, PVenc, ldenc, offEnc, rank, E_idx32, mask,
                            edge_mask, W1b_packed, W2_packed, b2, B, B, L, K, gsum, stream);
}

// the sparse variant's decoder (protein_mpnn.py:203-217): every edge of the list reads the neighbour's CURRENT row
// (A with the label term where the neighbour is decoded earlier, else B without it) and only the VALID edges
// (edge_mask = 1) enter the neighbour sum; PQ holds [Pa | A | B] per residue
extern "C" int MPNN_API(mpnn_dec_message_valid)(const float* hE_frag, const float* PQ, int ldpq, int offPa, int offA,
                                      int offB, const int32_t* rank, const int32_t* E_idx32, const float* mask,
                                      const float* edge_mask, const float* W1b_packed, const float* W2_packed,
                                      const float* b2, int B, int L, int K, float* gsum, void* stream) {
  MPNN_CHECK_ARG(edge_mask, "mpnn_dec_message_valid: the edge validity mask is required");
  return launch_dec_message("mpnn_dec_message_valid", hE_frag, PQ, ldpq, offPa, offA, PQ, ldpq, offB, rank, E_idx32, mask,
                            edge_mask, W1b_packed, W2_packed, b2, B, B, L, K, gsum, stream, 1);
}

// B rows (sequences / decoding orders) on n_struct structures: row r decodes structure r % n_struct
extern "C" int MPNN_API(mpnn_dec_message_shared)(const float* hE_frag, const float* PQ, int ldpq, int offPa, int offA,
                                       const float* PVenc, int ldenc, int offEnc, const int32_t* rank,
                                       const int32_t* E_idx32, const float* mask, const float* edge_mask,
                                       const float* W1b_packed, const float* W2_packed, const float* b2, int B,
                                       int n_struct, int L, int K, float* gsum, void* stream) {
  return launch_dec_message("mpnn_dec_message_shared", hE_frag, PQ, ldpq, offPa, offA, PVenc, ldenc, offEnc, rank, E_idx32,
                            mask, edge_mask, W1b_packed, W2_packed, b2, B, n_struct, L, K, gsum, stream);
}

// ---- training forward: same kernels, pre-activations saved for mpnn_edge_bwd.hip (K <= 64) ----
#ifndef MPNN_FAST
extern "C" int mpnn_message_train(int dec, const float* hE_frag, const float* P, const float* RowA, const float* RowB,
                                  const int32_t* rank, const int32_t* E_idx32, const float* mask,
                                  const float* edge_mask, const float* W1b_packed, const float* W2_packed,
                                  const float* b2, int B, int L, int K, float* gsum, float* cnt, float* a1_frag,
                                  float* a2_frag, int w_valid, void* stream) {
  MPNN_CHECK_ARG(hE_frag && P && RowA && E_idx32 && mask && W1b_packed && W2_packed && b2 && gsum && a1_frag && a2_frag,
                 "mpnn_message_train: null pointer");
  MPNN_CHECK_ARG(dec ? (RowB && rank) : (cnt != nullptr), "mpnn_message_train: decoder needs RowB and rank, encoder cnt");
  MPNN_CHECK_ARG(!w_valid || (dec && edge_mask), "mpnn_message_train: w_valid is the decoder's form and needs edge_mask");
  MPNN_CHECK_ARG(B > 0 && L > 0 && K >= 1 && K <= 64, "mpnn_message_train: bad sizes (K <= 64)");
  const int N = B * L, KT = (K + 15) / 16;
  dim3 grid(edge_grid(N, MPNN_MSG_WAVES)), block(MPNN_MSG_WAVES * 64);
  if (dec) {
    KT_DISPATCH(KT, hipLaunchKernelGGL((k_message<kKT, true, true>), grid, block, 0, (hipStream_t)stream,
                                       (const f32x4*)hE_frag, P, 128, 0, RowA, 128, 0, RowB, 128, 0, rank, E_idx32,
                                       mask, edge_mask, (const h8*)W1b_packed, (const h8*)W2_packed, b2, N, L, K, gsum,
                                       (float*)nullptr, (f32x4*)a1_frag, (f32x4*)a2_frag, N, w_valid));
  } else {
    KT_DISPATCH(KT, hipLaunchKernelGGL((k_message<kKT, false, true>), grid, block, 0, (hipStream_t)stream,
                                       (const f32x4*)hE_frag, P, 128, 0, RowA, 128, 0, (const float*)nullptr, 0, 0,
                                       (const int*)nullptr, E_idx32, mask, edge_mask, (const h8*)W1b_packed,
                                       (const h8*)W2_packed, b2, N, L, K, gsum, cnt, (f32x4*)a1_frag, (f32x4*)a2_frag, N, 0));
  }
  MPNN_LAUNCH_CHECK();
}
#endif

#ifndef MPNN_FAST
extern "C" int mpnn_enc_edge_update_train(const float* hE_frag, const float* P, const float* Q, const int32_t* E_idx32,
                                          const float* W11b_packed, const float* W12_packed, const float* b12,
                                          const float* W13_packed, const float* b13, const float* ln_w,
                                          const float* ln_b, int B, int L, int K, float* hE_out, float* a1_frag,
                                          float* a2_frag, float* z_frag, float p_drop, uint32_t drop_seed,
                                          float* rstd_max, void* stream) {
  MPNN_CHECK_ARG(hE_frag && P && Q && E_idx32 && W11b_packed && W12_packed && b12 && W13_packed && b13 && ln_w &&
                     ln_b && hE_out && a1_frag && a2_frag && z_frag,
                 "mpnn_enc_edge_update_train: null pointer");
  MPNN_CHECK_ARG(B > 0 && L > 0 && K >= 1 && K <= 64 && hE_out != hE_frag,
                 "mpnn_enc_edge_update_train: bad sizes (K <= 64, out of place)");
  MPNN_CHECK_ARG(p_drop >= 0.f && p_drop < 1.f, "mpnn_enc_edge_update_train: dropout probability must be in [0, 1)");
  const unsigned thresh = p_drop > 0.f ? (unsigned)((double)p_drop * 4294967296.0) : 0u;
  const int N = B * L, KT = (K + 15) / 16;
  dim3 grid(edge_grid(N, MPNN_UPD_WAVES)), block(MPNN_UPD_WAVES * 64);
  KT_DISPATCH(KT, hipLaunchKernelGGL((k_enc_edge_update<kKT, true>), grid, block, 0, (hipStream_t)stream,
                                     (f32x4*)hE_frag, P, 128, 0, Q, 128, 0, E_idx32, (const h8*)W11b_packed,
                                     (const h8*)W12_packed, b12, (const h8*)W13_packed, b13, ln_w, ln_b, N, L,
                                     (f32x4*)hE_out, (f32x4*)a1_frag, (f32x4*)a2_frag, (f32x4*)z_frag, thresh,
                                     (unsigned)drop_seed, 1.0f / (1.0f - p_drop), (unsigned*)rstd_max));
  MPNN_LAUNCH_CHECK();
}
#endif

// ---- autoregressive sampling: decoder message for a listed subset of residues of samples sharing one encoder ----
extern "C" int MPNN_API(mpnn_dec_message_subset)(const float* hE_frag, const float* P, int ldp, int offP, const float* RowA,
                                       int lda, int offA, const float* PVenc, int ldenc, int offEnc,
                                       const int32_t* rank, const int32_t* E_idx32, const float* mask,
                                       const float* edge_mask, const float* W1b_packed, const float* W2_packed,
                                       const float* b2, const int32_t* node_list, int n_list, int enc_nodes, int L,
                                       int K, float* gsum, void* stream) {
  MPNN_CHECK_ARG(hE_frag && P && RowA && PVenc && rank && E_idx32 && mask && W1b_packed && W2_packed && b2 &&
                     node_list && gsum,
                 "mpnn_dec_message_subset: null pointer");
  MPNN_CHECK_ARG(n_list > 0 && L > 0 && enc_nodes > 0 && enc_nodes % L == 0 && K >= 1 && K <= 64 &&
                     mult4(ldp, offP, lda, offA, ldenc, offEnc),
                 "mpnn_dec_message_subset: bad sizes (K <= 64; strides/offsets multiples of 4 floats)");
  const int KT = (K + 15) / 16;
  dim3 grid(edge_grid(n_list * KT, MPNN_MSGT_WAVES)), block(MPNN_MSGT_WAVES * 64);
  hipLaunchKernelGGL(k_message_level, grid, block, 0, (hipStream_t)stream, (const f32x4*)hE_frag, P, ldp, offP, RowA, lda,
                     offA, PVenc, ldenc, offEnc, rank, E_idx32, mask, edge_mask, (const h8*)W1b_packed,
                     (const h8*)W2_packed, b2, node_list, n_list, enc_nodes, L, K, KT, gsum);
  MPNN_LAUNCH_CHECK();
}

// mpnn_dec_message_subset + mpnn_node_update_subset of one decoder layer in one launch (no gsum buffer)
extern "C" int MPNN_API(mpnn_dec_layer_subset)(const float* hE_frag, const float* P, int ldp, int offP, const float* RowA,
                                     int lda, int offA, const float* PVenc, int ldenc, int offEnc,
                                     const int32_t* rank, const int32_t* E_idx32, const float* mask,
                                     const float* edge_mask, const float* W1b_packed, const float* W2_packed,
                                     const float* b2, const int32_t* node_list, int n_list, int enc_nodes, int L, int K,
                                     const float* hV_in, float cnt_const, const float* W3_packed, const float* b3,
                                     const float* ln1_w, const float* ln1_b, const float* Win_packed,
                                     const float* b_in, const float* Wout_packed, const float* b_out,
                                     const float* ln2_w, const float* ln2_b, float scale, const float* Wproj_packed,
                                     const float* bproj, int nproj, float* hV_out, float* PQ, void* stream) {
  MPNN_CHECK_ARG(hE_frag && P && RowA && PVenc && rank && E_idx32 && mask && W1b_packed && W2_packed && b2 &&
                     node_list && hV_in && W3_packed && b3 && ln1_w && ln1_b && Win_packed && b_in && Wout_packed &&
                     b_out && ln2_w && ln2_b && hV_out,
                 "mpnn_dec_layer_subset: null pointer");
  MPNN_CHECK_ARG(n_list > 0 && L > 0 && enc_nodes > 0 && enc_nodes % L == 0 && K >= 1 && K <= 48 && scale != 0.f &&
                     nproj >= 0 && nproj <= 8 && (nproj == 0 || (Wproj_packed && bproj && PQ)) &&
                     mult4(ldp, offP, lda, offA, ldenc, offEnc),
                 "mpnn_dec_layer_subset: bad sizes (K <= 48; at most 8 projection blocks; strides/offsets multiples of 4 floats)");
  const int KT = (K + 15) / 16;
  const int lvn = level_nodes_per_wg(n_list);
  hipLaunchKernelGGL((k_dec_layer_level<false>), dim3(ceil_div(n_list, lvn)), dim3(MPNN_MSGT_WAVES * 64), 0,
                     (hipStream_t)stream, (const f32x4*)hE_frag, P, ldp, offP, RowA, lda, offA, PVenc, ldenc, offEnc, rank,
                     E_idx32, mask, edge_mask, (const h8*)W1b_packed, (const h8*)W2_packed, b2, node_list, n_list,
                     enc_nodes, L, K, KT, hV_in, cnt_const, (const h8*)W3_packed, b3, ln1_w, ln1_b, (const h8*)Win_packed,
                     b_in, (const h8*)Wout_packed, b_out, ln2_w, ln2_b, 1.0f / scale, (const h8*)Wproj_packed, bproj,
                     nproj, hV_out, PQ, lvn, LevelHead{});
  MPNN_LAUNCH_CHECK();
}

// mpnn_dec_layer_subset of the LAST decoder layer (no projections follow) + mpnn_sample_subset on the same list, in
// one launch
extern "C" int MPNN_API(mpnn_dec_layer_sample_subset)(
    const float* hE_frag, const float* P, int ldp, int offP, const float* RowA, int lda, int offA, const float* PVenc,
    int ldenc, int offEnc, const int32_t* rank, const int32_t* E_idx32, const float* mask, const float* edge_mask,
    const float* W1b_packed, const float* W2_packed, const float* b2, const int32_t* node_list, int n_list, int enc_nodes,
    int L, int K, const float* hV_in, float cnt_const, const float* W3_packed, const float* b3, const float* ln1_w,
    const float* ln1_b, const float* Win_packed, const float* b_in, const float* Wout_packed, const float* b_out,
    const float* ln2_w, const float* ln2_b, float scale, float* hV_out, const float* Wlogit_packed, const float* blogit,
    const float* logit_bias, int n_letters, float temperature, uint64_t seed, const float* designable,
    const int64_t* S_in, int64_t* S_out, float* log_probs, const float* TS, int ld_ts, int n_layers,
    float* const* A_rows, const int* A_ld, const int* A_off, void* stream) {
  MPNN_CHECK_ARG(hE_frag && P && RowA && PVenc && rank && E_idx32 && mask && W1b_packed && W2_packed && b2 &&
                     node_list && hV_in && W3_packed && b3 && ln1_w && ln1_b && Win_packed && b_in && Wout_packed &&
                     b_out && ln2_w && ln2_b && hV_out && Wlogit_packed && blogit && designable && S_in && S_out &&
                     log_probs && TS && A_rows && A_ld && A_off,
                 "mpnn_dec_layer_sample_subset: null pointer");
  MPNN_CHECK_ARG(n_list > 0 && L > 0 && enc_nodes > 0 && enc_nodes % L == 0 && K >= 1 && K <= 48 && scale != 0.f &&
                     mult4(ldp, offP, lda, offA, ldenc, offEnc) && n_letters >= 1 && n_letters <= 32 &&
                     temperature > 0.f && n_layers >= 0 && n_layers <= 4,
                 "mpnn_dec_layer_sample_subset: bad sizes (K <= 48; n_letters <= 32; temperature > 0; at most 4 decoder "
                 "layers; strides/offsets multiples of 4 floats)");
  LevelHead head;
  head.Wp = (const f32x4*)Wlogit_packed;
  head.bias = blogit;
  head.logit_bias = logit_bias;
  head.n_letters = n_letters;
  head.inv_temperature = 1.0f / temperature;
  head.seed = (unsigned long long)seed;
  head.designable = designable;
  head.S_in = (const long long*)S_in;
  head.S_out = (long long*)S_out;
  head.log_probs = log_probs;
  head.TS = TS;
  head.ld_ts = ld_ts;
  head.rows.n_layers = n_layers;
  for (int l = 0; l < 4; ++l) {
    head.rows.A[l] = l < n_layers ? A_rows[l] : nullptr;
    head.rows.ld[l] = l < n_layers ? A_ld[l] : 0;
    head.rows.off[l] = l < n_layers ? A_off[l] : 0;
  }
  const int KT = (K + 15) / 16;
  const int lvn = level_nodes_per_wg(n_list);
  hipLaunchKernelGGL((k_dec_layer_level<true>), dim3(ceil_div(n_list, lvn)), dim3(MPNN_MSGT_WAVES * 64), 0,
                     (hipStream_t)stream, (const f32x4*)hE_frag, P, ldp, offP, RowA, lda, offA, PVenc, ldenc, offEnc, rank,
                     E_idx32, mask, edge_mask, (const h8*)W1b_packed, (const h8*)W2_packed, b2, node_list, n_list,
                     enc_nodes, L, K, KT, hV_in, cnt_const, (const h8*)W3_packed, b3, ln1_w, ln1_b, (const h8*)Win_packed,
                     b_in, (const h8*)Wout_packed, b_out, ln2_w, ln2_b, 1.0f / scale, (const h8*)nullptr,
                     (const float*)nullptr, 0, hV_out, (float*)nullptr, lvn, head);
  MPNN_LAUNCH_CHECK();
}

// ---- the sampler's whole level loop in one call ----
// The decoder of a sampling call is ~70 dependency levels x (decoder layers) dependent launches; issued one ctypes call
// at a time from Python they cost ~11 us of host time each, more than the kernels behind them take.  This entry point
// walks the host array of level sizes and enqueues the same launches (mpnn_dec_layer_subset for every layer but the
// last, then mpnn_dec_layer_sample_subset - or mpnn_dec_layer_subset + mpnn_sample_subset with draw_fused = 0) on the
// passed stream: same kernels, same order, same results as the per-level calls; no allocation, no synchronisation.
// (the sampling head has one build: its arithmetic is fp32 in both precision modes)
extern "C" int mpnn_sample_subset(const float* hV, const float* Wout_packed, const float* bias, const float* logit_bias,
                                  int n_letters, const int32_t* node_list, int n_list, float temperature,
                                  uint64_t seed, const float* designable, const int64_t* S_in, int64_t* S_out,
                                  float* log_probs, const float* TS, int ld_ts, int n_layers, float* const* A_rows,
                                  const int* A_ld, const int* A_off, void* stream);

extern "C" int MPNN_API(mpnn_decode_run)(const float* hE_frag, const float* PVenc, int ldenc, const int32_t* rank,
                                         const int32_t* E_idx32, const float* mask, const float* edge_mask,
                                         const int32_t* node_list, const int32_t* level_counts, int n_levels,
                                         int enc_nodes, int L, int K, float cnt_const, const mpnn_level_layer* layers,
                                         int n_layers, const float* Wlogit_packed, const float* logit_b,
                                         const float* logit_bias, int n_letters, float temperature, uint64_t seed,
                                         const float* designable, const int64_t* S_in, int64_t* S_out, float* log_probs,
                                         const float* TS, int ld_ts, int draw_fused, void* stream) {
  MPNN_CHECK_ARG(node_list && level_counts && layers && n_levels >= 0 && n_layers >= 1 && n_layers <= 4,
                 "mpnn_decode_run: need the level list, its host array of sizes and 1..4 layers");
  MPNN_CHECK_ARG(layers[n_layers - 1].nproj == 0, "mpnn_decode_run: the last decoder layer carries no projections");
  float* A_rows[4] = {};
  int A_ld[4] = {}, A_off[4] = {};
  for (int l = 0; l < n_layers; ++l) {
    // a layer's input rows must be the previous layer's outputs (as mpnn_decode_plan demands)
    MPNN_CHECK_ARG(l == 0 || (layers[l].hV_in == layers[l - 1].hV_out && layers[l - 1].nproj >= 2),
                   "mpnn_decode_run: layer l must read the hV rows (and Pa | A projections) layer l-1 writes");
    A_rows[l] = layers[l].RowA;
    A_ld[l] = layers[l].lda;
    A_off[l] = layers[l].offA;
  }
  long long start = 0;
  for (int v = 0; v < n_levels; ++v) {
    const int n_list = level_counts[v];
    MPNN_CHECK_ARG(n_list > 0, "mpnn_decode_run: empty level");
    const int32_t* list = node_list + start;
    start += n_list;
    for (int l = 0; l < n_layers; ++l) {
      const mpnn_level_layer& y = layers[l];
      int rc;
      if (l + 1 < n_layers || !draw_fused) {
        rc = MPNN_API(mpnn_dec_layer_subset)(hE_frag, y.P, y.ldp, y.offP, y.RowA, y.lda, y.offA, PVenc, ldenc, y.offEnc,
                                             rank, E_idx32, mask, edge_mask, y.W1b_packed, y.W2_packed, y.b2, list,
                                             n_list, enc_nodes, L, K, y.hV_in, cnt_const, y.W3_packed, y.b3, y.ln1_w,
                                             y.ln1_b, y.Win_packed, y.b_in, y.Wout8_packed, y.b_out, y.ln2_w, y.ln2_b,
                                             y.scale, y.Wproj_packed, y.bproj, y.nproj, y.hV_out, y.PQ, stream);
      } else {
        rc = MPNN_API(mpnn_dec_layer_sample_subset)(
            hE_frag, y.P, y.ldp, y.offP, y.RowA, y.lda, y.offA, PVenc, ldenc, y.offEnc, rank, E_idx32, mask, edge_mask,
            y.W1b_packed, y.W2_packed, y.b2, list, n_list, enc_nodes, L, K, y.hV_in, cnt_const, y.W3_packed, y.b3,
            y.ln1_w, y.ln1_b, y.Win_packed, y.b_in, y.Wout8_packed, y.b_out, y.ln2_w, y.ln2_b, y.scale, y.hV_out,
            Wlogit_packed, logit_b, logit_bias, n_letters, temperature, seed, designable, S_in, S_out, log_probs, TS,
            ld_ts, n_layers, A_rows, A_ld, A_off, stream);
      }
      if (rc) return rc;
    }
    if (!draw_fused) {
      const int rc = mpnn_sample_subset(layers[n_layers - 1].hV_out, Wlogit_packed, logit_b, logit_bias, n_letters, list,
                                        n_list, temperature, seed, designable, S_in, S_out, log_probs, TS, ld_ts,
                                        n_layers, A_rows, A_ld, A_off, stream);
      if (rc) return rc;
    }
  }
  return 0;
}
