// Backbone featurizer kernels for gfx950 (reference: src/mpnn/models/model_utils.py
// ProteinFeatures, :180-360).  Built with -ffp-contract=off: the neighbour search must
// round every operation exactly like the reference's separate ATen ops do.
#include "mpnn_edge_common.h"

MPNN_KERNELS_BEGIN

// Distance of an atom pair for the RBF features: the hardware square root (v_sqrt_f32, 1 ulp).  The neighbour SEARCH
// needs the correctly rounded sqrtf (its ranks must equal the reference's bit for bit) and keeps it; here the distance
// only enters exp(-((d - mu) / 1.25)^2), where 1 ulp of d (1e-6 A at 20 A) moves a feature by < 2e-6 relative, and
// sqrtf's software refinement (cmp / cndmask / fma: ~20 instructions, 75 roots per residue) was 45 % of the
// featurizer's vector instructions.
__device__ __forceinline__ float rbf_distance(float d2) {
  return __builtin_amdgcn_sqrtf(d2);
}


// ---------------------------------------------------------------------------
// backbone atoms: X [N,4,3] (+noise) -> atoms [N][6][4] = N, CA, C, O, CB, V (xyz + pad)
//                                        ca_soa [B][4][L] = CA.x, CA.y, CA.z, mask
// reference :311-330 (noise, virtual CB), :215-230 (virtual centre)
// ---------------------------------------------------------------------------
__global__ void k_backbone_atoms(const float* __restrict__ X, const float* __restrict__ noise,
                                 const float* __restrict__ mask, int N, int L, float* __restrict__ atoms,
                                 float* __restrict__ ca_soa) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  float p[12];
#pragma unroll
  for (int k = 0; k < 12; ++k) {
    float v = X[(size_t)n * 12 + k];
    if (noise) v = v + noise[(size_t)n * 12 + k];
    p[k] = v;
  }
  const float *Nn = p, *Ca = p + 3, *C = p + 6, *O = p + 9;
  float b[3], c[3], a[3], cb[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    b[k] = Ca[k] - Nn[k];
    c[k] = C[k] - Ca[k];
  }
  a[0] = b[1] * c[2] - b[2] * c[1];
  a[1] = b[2] * c[0] - b[0] * c[2];
  a[2] = b[0] * c[1] - b[1] * c[0];
#pragma unroll
  for (int k = 0; k < 3; ++k) cb[k] = ((-0.58273431f * a[k] + 0.56802827f * b[k]) - 0.54067466f * c[k]) + Ca[k];
  // virtual centre (always computed; consumers ignore it unless asked)
  float ubc[3], unc[3], pl[3], vd[3], v[3];
  {
    float nb = 0.f, nn = 0.f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      ubc[k] = cb[k] - Ca[k];
      unc[k] = Nn[k] - Ca[k];
    }
    nb = sqrtf((ubc[0] * ubc[0] + ubc[1] * ubc[1]) + ubc[2] * ubc[2]) + 1e-6f;
    nn = sqrtf((unc[0] * unc[0] + unc[1] * unc[1]) + unc[2] * unc[2]) + 1e-6f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      ubc[k] = ubc[k] / nb;
      unc[k] = unc[k] / nn;
    }
    pl[0] = unc[1] * ubc[2] - unc[2] * ubc[1];
    pl[1] = unc[2] * ubc[0] - unc[0] * ubc[2];
    pl[2] = unc[0] * ubc[1] - unc[1] * ubc[0];
    const float np_ = sqrtf((pl[0] * pl[0] + pl[1] * pl[1]) + pl[2] * pl[2]) + 1e-6f;
#pragma unroll
    for (int k = 0; k < 3; ++k) pl[k] = pl[k] / np_;
    vd[0] = -(pl[1] * ubc[2] - pl[2] * ubc[1]);
    vd[1] = -(pl[2] * ubc[0] - pl[0] * ubc[2]);
    vd[2] = -(pl[0] * ubc[1] - pl[1] * ubc[0]);
#pragma unroll
    for (int k = 0; k < 3; ++k) v[k] = Ca[k] + 3.06f * vd[k];
  }
  f32x4* out = reinterpret_cast<f32x4*>(atoms + (size_t)n * 24);
  out[0] = f32x4{Nn[0], Nn[1], Nn[2], 0.f};
  out[1] = f32x4{Ca[0], Ca[1], Ca[2], 0.f};
  out[2] = f32x4{C[0], C[1], C[2], 0.f};
  out[3] = f32x4{O[0], O[1], O[2], 0.f};
  out[4] = f32x4{cb[0], cb[1], cb[2], 0.f};
  out[5] = f32x4{v[0], v[1], v[2], 0.f};
  const int bb = n / L, l = n - bb * L;
  float* so = ca_soa + (size_t)bb * 4 * L;
  so[l] = Ca[0];
  so[L + l] = Ca[1];
  so[2 * L + l] = Ca[2];
  so[3 * L + l] = mask[n];
}

// ---------------------------------------------------------------------------
// k-NN top-k: one wavefront per query residue (reference _dist, :270-289)
//   D      = m_i m_j sqrt(((dx^2 + dy^2) + dz^2) + 1e-6)
//   D_adj  = D + (1 - m_i m_j) * rowmax(D)   [+ (1 - [cutoff > D]) * 1e6]
//   output = K smallest D_adj ascending; equal values ordered by column index.
// Selection: candidates pass through the wave 64 at a time as 64-bit keys
// (float bits << 32 | column).  The running best-64 list lives one key per lane,
// sorted; a chunk that cannot beat the current K-th key is skipped with one ballot,
// otherwise it is bitonic-sorted (21 exchange stages) and bitonic-merged (7 stages).
// ---------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long shfl_xor_u64(unsigned long long v, int m) {
  unsigned lo = (unsigned)v, hi = (unsigned)(v >> 32);
  lo = __shfl_xor((int)lo, m);
  hi = __shfl_xor((int)hi, m);
  return ((unsigned long long)hi << 32) | lo;
}

__device__ __forceinline__ unsigned long long shfl_u64(unsigned long long v, int src) {
  unsigned lo = (unsigned)v, hi = (unsigned)(v >> 32);
  lo = __shfl((int)lo, src);
  hi = __shfl((int)hi, src);
  return ((unsigned long long)hi << 32) | lo;
}

__device__ __forceinline__ unsigned long long cmpxchg(unsigned long long v, int lane, int j, bool up) {
  const unsigned long long o = shfl_xor_u64(v, j);
  const bool lower = (lane & j) == 0;
  const bool take_min = (lower == up);
  const unsigned long long mn = v < o ? v : o, mx = v < o ? o : v;
  return take_min ? mn : mx;
}

__device__ __forceinline__ unsigned long long wave_sort64(unsigned long long v, int lane) {
#pragma unroll
  for (int k = 2; k <= 64; k <<= 1) {
#pragma unroll
    for (int j = k >> 1; j > 0; j >>= 1) {
      const bool up = ((lane & k) == 0);
      v = cmpxchg(v, lane, j, up);
    }
  }
  return v;
}

__global__ void __launch_bounds__(256)
k_knn_topk(const float* __restrict__ ca_soa, int B, int L, int K, int Kp, int has_cutoff, float cutoff,
           long long* __restrict__ E_idx64, int* __restrict__ E_idx32, float* __restrict__ D_nb,
           float* __restrict__ edge_mask) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);  // global residue
  if (row >= B * L) return;
  const int b = row / L, i = row - b * L;
  const float* cx = ca_soa + (size_t)b * 4 * L;
  const float *cy = cx + L, *cz = cx + 2 * L, *cm = cx + 3 * L;
  const float xi = cx[i], yi = cy[i], zi = cz[i], mi = cm[i];

  // Streaming selection over chunks of 64 candidates.  Chunks are visited outwards from the one that holds the query
  // itself (c0, c0+1, c0-1, c0+2, ...): residues close in sequence are close in space, so the K-th best distance
  // tightens within the first few chunks and most of the remaining ones are rejected by the ballot.  The result does
  // not depend on the order: keys are totally ordered.
  //
  // Masked columns (m_i m_j != 1) take a value that depends on rowmax(D), which is only known after a full sweep.  The first
  // sweep therefore leaves them out and computes rowmax on the way; they can only matter if rowmax <= the K-th best
  // value found (short or padded-out proteins, masked query rows, cutoff penalties), and then a second sweep with the
  // now known rowmax inserts them.  Fully valid proteins - the common case - make one sweep instead of two.
  const int nchunks = (L + 63) >> 6, c0 = i >> 6;
  float dmax = 0.f;
  bool any_masked = false;
  unsigned long long best = ~0ull;  // sorted ascending over lanes
  auto sweep = [&](bool with_masked) {
    best = ~0ull;
    for (int t = 0; t < 2 * nchunks; ++t) {
      const int cidx = (t & 1) ? c0 + ((t + 1) >> 1) : c0 - (t >> 1);
      if (cidx < 0 || cidx >= nchunks) continue;
      const int j = (cidx << 6) + lane;
      unsigned long long key = ~0ull;
      if (j < L) {
        const float dx = cx[j] - xi, dy = cy[j] - yi, dz = cz[j] - zi;
        const float m2 = mi * cm[j];
        const float d = m2 * sqrtf(((dx * dx + dy * dy) + dz * dz) + 1e-6f);
        if (!with_masked) {
          dmax = fmaxf(dmax, d);
          any_masked = any_masked || (m2 != 1.0f);
        }
        if (with_masked || m2 == 1.0f) {
          float da = d + (1.0f - m2) * dmax;
          if (has_cutoff) da = da + (1.0f - (cutoff > d ? 1.0f : 0.0f)) * 1e6f;
          key = ((unsigned long long)__float_as_uint(da) << 32) | (unsigned)j;
        }
      }
      const unsigned long long thr = shfl_u64(best, K - 1);
      if (__ballot(key < thr) == 0ull) continue;
      key = wave_sort64(key, lane);
      const unsigned long long rev = shfl_u64(key, 63 - lane);
      best = best < rev ? best : rev;  // bitonic: holds the 64 smallest of the union
#pragma unroll
      for (int jj = 32; jj > 0; jj >>= 1) best = cmpxchg(best, lane, jj, true);
    }
  };
  sweep(false);
#pragma unroll
  for (int m = 32; m > 0; m >>= 1) dmax = fmaxf(dmax, __shfl_xor(dmax, m));
  if (__ballot(any_masked) != 0ull) {
    // value of a masked column: d = 0, so D_adj = rowmax (the cutoff term vanishes: cutoff > 0 = d)
    const unsigned long long kth = shfl_u64(best, K - 1);
    if (((unsigned long long)__float_as_uint(dmax) << 32) <= kth) sweep(true);
  }
  const float dsel = __uint_as_float((unsigned)(best >> 32));
  const int jsel = (int)(unsigned)(best & 0xffffffffu);
  if (lane < K) {
    if (E_idx64) E_idx64[(size_t)row * K + lane] = jsel;
    if (D_nb) D_nb[(size_t)row * K + lane] = dsel;
    if (edge_mask) edge_mask[(size_t)row * K + lane] = (dsel < cutoff) ? 1.0f : 0.0f;
  }
  // internal index list, padded to a multiple of 16 with the residue itself (always in range)
  if (lane < Kp) E_idx32[(size_t)row * Kp + lane] = lane < K ? jsel : i;
}

// ---------------------------------------------------------------------------
// occupancy: log1p sum_{j != i, valid, d2 < r2} exp(-d2 / (2 (r/3)^2)) on V-V distances
// one wavefront per residue (reference _compute_occupancy, :232-268)
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_occupancy(const float* __restrict__ atoms, const float* __restrict__ mask, int B, int L, float cutoff,
            float* __restrict__ occ) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= B * L) return;
  const int b = row / L, i = row - b * L;
  const f32x4 vi = reinterpret_cast<const f32x4*>(atoms + (size_t)row * 24)[5];
  const float mi = mask[row];
  const float r2 = cutoff * cutoff;
  const float sigma = cutoff / 3.0f;
  const float denom = 2.0f * (sigma * sigma);
  float acc = 0.f;
  for (int j = lane; j < L; j += 64) {
    const size_t nj = (size_t)b * L + j;
    const f32x4 vj = reinterpret_cast<const f32x4*>(atoms + nj * 24)[5];
    const float dx = vi.x - vj.x, dy = vi.y - vj.y, dz = vi.z - vj.z;
    const float d2 = (dx * dx + dy * dy) + dz * dz;
    const float w = (mi * mask[nj]) * (j != i ? 1.0f : 0.0f) * (d2 < r2 ? 1.0f : 0.0f);
    acc += expf(-d2 / denom) * w;
  }
#pragma unroll
  for (int m = 32; m > 0; m >>= 1) acc += __shfl_xor(acc, m);
  if (lane == 0) occ[row] = log1pf(acc);
}

// ---------------------------------------------------------------------------
// weight packing (see mpnn_common.h): W[out, ld] columns [col0, col0 + in) ->
//   packed[to][ti][lane][r] = W[16 to + (lane&15)][col0 + 16 ti + 4 (lane>>4) + r]
// rows/cols beyond (out_f, in_f) are zero-filled (out_f, in_f need not be multiples of 16)
// ---------------------------------------------------------------------------
__global__ void k_pack_linear(const float* __restrict__ W, int out_f, int in_f, int ld, int col0,
                              float* __restrict__ packed, int nto, int nti) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;  // one float4 each
  if (idx >= nto * nti * 64) return;
  const int lane = idx & 63, ti = (idx >> 6) % nti, to = (idx >> 6) / nti;
  const int row = 16 * to + (lane & 15);
  f32x4 v;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int col = 16 * ti + 4 * (lane >> 4) + r;
    v[r] = (row < out_f && col < in_f) ? W[(size_t)row * ld + col0 + col] : 0.f;
  }
  reinterpret_cast<f32x4*>(packed)[idx] = v;
}

// split-f16 packing (see mpnn_common.h): packed[to][kk][part][lane] is an h8 whose element j is the hi (part 0)
// or lo (part 1) half of W[16 to + (lane&15)][col0 + 16 (2 kk + (j>>2)) + 4 (lane>>4) + (j&3)]
__global__ void k_pack_linear_split(const float* __restrict__ W, int out_f, int in_f, int ld, int col0, int rot,
                                    h8* __restrict__ packed, int nto, int nkk) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;  // one (to, kk, lane)
  if (idx >= nto * nkk * 64) return;
  const int lane = idx & 63, kk = (idx >> 6) % nkk, to = (idx >> 6) / nkk;
  const int row = 16 * to + (lane & 15);
  h8 hi, lo;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int col = 16 * (2 * kk + (j >> 2)) + 4 * (lane >> 4) + (j & 3);
    // rot rotates the input features: packed column `col` is source column (col + rot) mod in_f
    const float w = (row < out_f && col < in_f) ? W[(size_t)row * ld + col0 + (col + rot) % in_f] : 0.f;
    const _Float16 h = (_Float16)w;
    hi[j] = h;
    lo[j] = (_Float16)(w - (float)h);
  }
  packed[((to * nkk + kk) * 2 + 0) * 64 + lane] = hi;
  packed[((to * nkk + kk) * 2 + 1) * 64 + lane] = lo;
}

// positional table: T[d][p] = W[p][d] + b[p]   (one_hot(d) @ W^T + b, :168-177)
__global__ void k_pos_table(const float* __restrict__ W, const float* __restrict__ bias, int nbuckets, int npos,
                            float* __restrict__ T) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= nbuckets * npos) return;
  const int d = idx / npos, p = idx - d * npos;
  T[idx] = W[p * nbuckets + d] + bias[p];
}

// ---------------------------------------------------------------------------
// edge features: one wavefront per residue, KT column tiles of 16 neighbours, split-f16 MFMA.
//   B operand generated on the fly, one k-step = two 16-feature tiles.  Tile order (the packed
//   edge_embedding weight is rotated by 16 columns to match): tiles 0 .. A*A-1 = the 16 RBFs of
//   |atom_a1(i) - atom_a2(j)| in itertools.product order (:344-347), tile A*A = positional table row
//   (:349-356), then zero padding to an even tile count.
//   E   = LayerNorm(W_emb . feat)            (:357-359)
//   h_E = W_e . E + b_e                      (:464)
// Both weights stream from L2 as A fragments; outputs: hE in fragment layout and, optionally, E in the
// reference's natural [N,K,128] layout.
// ---------------------------------------------------------------------------
#ifndef MPNN_FEAT_RING
#define MPNN_FEAT_RING 4  // weight fragments in flight ahead of their MFMAs
#endif
#ifndef MPNN_FEAT_PERSISTENT
#define MPNN_FEAT_PERSISTENT 1  // K <= 48: k_edge_features_ring (weights through LDS, once per eight residues)
#endif
template <int KT, int A, bool SAVE>
__global__ void __launch_bounds__(256, 2)
k_edge_features(const float* __restrict__ atoms, const int* __restrict__ E_idx, const long long* __restrict__ residue_idx,
                const long long* __restrict__ chain_labels, const float* __restrict__ pos_table,
                const float* __restrict__ rbf_mu, const h8* __restrict__ Wemb, const float* __restrict__ ln_w,
                const float* __restrict__ ln_b, const h8* __restrict__ We, const float* __restrict__ be, int N,
                int L, int K, float* __restrict__ E_nat, float* __restrict__ hE, f32x4* __restrict__ E0_frag,
                f32x4* __restrict__ E_frag, const float* __restrict__ emb_bias) {
  constexpr int NT_IN = A * A + 1;            // feature tiles: pairs, then positional
  constexpr int NKK_IN = (NT_IN + 1) / 2;     // k-steps of 32 features
  constexpr int D = MPNN_FEAT_RING;
  const int lane = threadIdx.x & 63;
  const int node = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (node >= N) return;
  const int g = lane >> 4, col = lane & 15;
  const int b = node / L;
  constexpr int Kp = KT * 16;

  int jn[KT];  // neighbour node (global index)
#pragma unroll
  for (int c = 0; c < KT; ++c) jn[c] = b * L + E_idx[(size_t)node * Kp + 16 * c + col];

  // the embedding's bias (the sparse variant's edge_mlp.0 has one, protein_mpnn.py:286-292; the dense one has none)
  f32x4 acc[MPNN_NT][KT];
#pragma unroll
  for (int t = 0; t < MPNN_NT; ++t) {
    f32x4 b0 = f32x4{0.f, 0.f, 0.f, 0.f};
    if (emb_bias) b0 = *reinterpret_cast<const f32x4*>(emb_bias + 16 * t + 4 * (lane >> 4));
#pragma unroll
    for (int c = 0; c < KT; ++c) acc[t][c] = b0;
  }

  float muc[4];  // RBF centres times (1 / 1.25) sqrt(log2 e)
#pragma unroll
  for (int r = 0; r < 4; ++r) muc[r] = rbf_mu[4 * g + r] * 0.96089792702915982f;
  const f32x4* atom_i = reinterpret_cast<const f32x4*>(atoms + (size_t)node * 24);

  // 16 RBFs (this lane: 4 of them) of atom pair p for each column tile
  auto rbf_tile = [&](int pair, f32x4 (&x)[KT]) {
    const f32x4 Ai = atom_i[pair / A];
#pragma unroll
    for (int c = 0; c < KT; ++c) {
      const f32x4 Bj = reinterpret_cast<const f32x4*>(atoms + (size_t)jn[c] * 24)[pair % A];
      const float dx = Ai.x - Bj.x, dy = Ai.y - Bj.y, dz = Ai.z - Bj.z;
      const float d = rbf_distance(((dx * dx + dy * dy) + dz * dz) + 1e-6f);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float z = fmaf(d, 0.96089792702915982f, -muc[r]);  // (d - mu) (1 / 1.25) sqrt(log2 e), one instruction
        x[c][r] = __builtin_amdgcn_exp2f(-(z * z));               // exp(-((d - mu) / 1.25)^2)
      }
    }
  };
  auto pos_tile = [&](f32x4 (&x)[KT]) {
    const long long ri = residue_idx[node], ci = chain_labels[node];
#pragma unroll
    for (int c = 0; c < KT; ++c) {
      const long long off = ri - residue_idx[jn[c]];
      const bool same = (ci - chain_labels[jn[c]]) == 0;
      long long d = off + 32;
      d = d < 0 ? 0 : (d > 64 ? 64 : d);
      if (!same) d = 65;
      x[c] = *reinterpret_cast<const f32x4*>(pos_table + d * 16 + 4 * g);
    }
  };
  auto feature_tile = [&](int tt, f32x4 (&x)[KT]) {
    if (tt < A * A) {
      rbf_tile(tt, x);
    } else if (tt == A * A) {
      pos_tile(x);
    } else {
#pragma unroll
      for (int c = 0; c < KT; ++c) x[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };

  {
    const h8* Wl = Wemb + lane;
    WRing<D> ring;
    wring_prologue<NKK_IN, MPNN_NT, D>(ring, Wl);
#pragma unroll
    for (int kk = 0; kk < NKK_IN; ++kk) {
      f32x4 x0[KT], x1[KT];
      feature_tile(2 * kk, x0);
      feature_tile(2 * kk + 1, x1);
      BFrag bf[KT];
#pragma unroll
      for (int c = 0; c < KT; ++c) split_pair(x0[c], x1[c], bf[c]);
      split_kstep_stream<KT, NKK_IN, MPNN_NT, D>(ring, Wl, kk, bf, acc);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const size_t foff = (size_t)node * (MPNN_NT * KT * 64) + lane;
  if (SAVE) {  // training: pre-LayerNorm embedding for the backward kernel
#pragma unroll
    for (int t = 0; t < MPNN_NT; ++t)
#pragma unroll
      for (int c = 0; c < KT; ++c) __builtin_nontemporal_store(acc[t][c], &E0_frag[foff + (t * KT + c) * 64]);
  }
  // ---- LayerNorm over features (norm_edges)
  tile_layernorm<KT>(acc, ln_w, ln_b, lane);
  if (SAVE) {
#pragma unroll
    for (int t = 0; t < MPNN_NT; ++t)
#pragma unroll
      for (int c = 0; c < KT; ++c) __builtin_nontemporal_store(acc[t][c], &E_frag[foff + (t * KT + c) * 64]);
  }
  if (E_nat) {
#pragma unroll
    for (int c = 0; c < KT; ++c) {
      const int e = 16 * c + col;
      if (e < K) {
        float* dst = E_nat + ((size_t)node * K + e) * MPNN_H + 4 * g;
#pragma unroll
        for (int t = 0; t < MPNN_NT; ++t) *reinterpret_cast<f32x4*>(dst + 16 * t) = acc[t][c];
      }
    }
  }
  // ---- h_E = W_e E + b
  BFrag eb[MPNN_NT / 2][KT];
  tile_to_bfrag<MPNN_NT, KT, false>(acc, eb);
#pragma unroll
  for (int t = 0; t < MPNN_NT; ++t) {
    const f32x4 bb = *reinterpret_cast<const f32x4*>(be + 16 * t + 4 * g);
#pragma unroll
    for (int c = 0; c < KT; ++c) acc[t][c] = bb;
  }
  split_gemm_stream<KT, MPNN_NT / 2, MPNN_NT, D>(We + lane, eb, acc);
  f32x4* out = reinterpret_cast<f32x4*>(hE) + (size_t)node * (MPNN_NT * KT * 64);
#pragma unroll
  for (int t = 0; t < MPNN_NT; ++t)
#pragma unroll
    for (int c = 0; c < KT; ++c) __builtin_nontemporal_store(acc[t][c], &out[(t * KT + c) * 64 + lane]);
}

// ---------------------------------------------------------------------------
// edge features, persistent form (K <= 48): the same arithmetic in the same order as k_edge_features (results are
// bit-identical), but the weights no longer stream from L2 once per RESIDUE (272 KiB per wave: 4.4 GB of L2 reads per
// B32 x L512 launch, the kernel's bound) - a workgroup of eight waves walks its residues eight at a time, all waves on
// the same k-step, and each 16 KiB k-step image of W_emb (8 output tiles x hi/lo) is brought into a three-slot LDS
// ring ONCE per round by LDS-DMA, two k-steps ahead of its MFMAs; W_e (64 KiB) stays resident in LDS.  One barrier
// per k-step: it tells every wave that the slot's image has landed (each wave waits for its own DMA pieces first) and
// that the slot refilled next is no longer being read.  The neighbours' atoms are gathered once per residue into
// registers (the streaming form re-reads them for every atom pair).
// ---------------------------------------------------------------------------
constexpr int FEAT_RING_SLOTS = 3;
constexpr int FEAT_KSTEP_H8 = MPNN_NT * 2 * 64;  // h8 elements of one k-step image (16 KiB)
constexpr int FEAT_WAVES = 8;

template <int KT, int A, bool SAVE>
__global__ void __launch_bounds__(FEAT_WAVES * 64)
k_edge_features_ring(const float* __restrict__ atoms, const int* __restrict__ E_idx,
                     const long long* __restrict__ residue_idx, const long long* __restrict__ chain_labels,
                     const float* __restrict__ pos_table, const float* __restrict__ rbf_mu,
                     const h8* __restrict__ Wemb, const float* __restrict__ ln_w, const float* __restrict__ ln_b,
                     const h8* __restrict__ We, const float* __restrict__ be, int N, int L, int K,
                     float* __restrict__ E_nat, float* __restrict__ hE, f32x4* __restrict__ E0_frag,
                     f32x4* __restrict__ E_frag, const float* __restrict__ emb_bias) {
  constexpr int NT_IN = A * A + 1;
  constexpr int NKK_IN = (NT_IN + 1) / 2;
  constexpr int R = FEAT_RING_SLOTS;
  __shared__ h8 ring[R * FEAT_KSTEP_H8];
  __shared__ h8 wes[WSPLIT];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane >> 4, col = lane & 15;
  constexpr int Kp = KT * 16;

  // rounds of this workgroup: every wave runs all of them (barriers), a wave without a residue in the last round
  // recomputes its previous one and stores nothing
  const ResidueRange rr = residue_range(N, FEAT_WAVES, wave);
  const ResidueRange r0 = residue_range(N, FEAT_WAVES, 0);
  const int n_rounds = r0.first < r0.end ? (r0.end - r0.first + r0.stride - 1) / r0.stride : 0;
  if (n_rounds == 0) return;
  const int n_steps = n_rounds * NKK_IN;

  // k-step image s (flattened over rounds) -> ring slot s % R: 16 pieces of 1 KiB (output tile, hi / lo), two per wave
  auto fill = [&](int s_) {
    const int kk = s_ % NKK_IN;
    h8* slot = ring + (s_ % R) * FEAT_KSTEP_H8;
#pragma unroll
    for (int i = 0; i < 2 * MPNN_NT / FEAT_WAVES; ++i) {
      const int piece = i * FEAT_WAVES + wave, to = piece >> 1, part = piece & 1;
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void*)(Wemb + ((to * NKK_IN + kk) * 2 + part) * 64 + lane),
          (__attribute__((address_space(3))) void*)(slot + piece * 64), 16, 0, 0);
    }
  };
#pragma unroll
  for (int s_ = 0; s_ < R - 1; ++s_)
    if (s_ < n_steps) fill(s_);
  lds_fill(wes, We, WSPLIT);

  float muc[4];  // RBF centres times (1 / 1.25) sqrt(log2 e)
#pragma unroll
  for (int r = 0; r < 4; ++r) muc[r] = rbf_mu[4 * g + r] * 0.96089792702915982f;

  int step = 0;
  for (int round = 0; round < n_rounds; ++round) {
    int node = rr.first + round * rr.stride;
    const bool has = node < rr.end;
    if (!has) node = rr.first < rr.end ? rr.end - 1 - ((rr.end - 1 - rr.first) % rr.stride) : (N - 1);
    node = __builtin_amdgcn_readfirstlane(node);  // wave-uniform: the residue's own atoms come through scalar loads
    const int b = node / L;
    int jn[KT];
#pragma unroll
    for (int c = 0; c < KT; ++c) jn[c] = b * L + E_idx[(size_t)node * Kp + 16 * c + col];
    // this residue's atoms and its neighbours' (A x KT gathers, once)
    f32x4 Ai[A];
    float Bj[A][KT][3];
    {
      const f32x4* atom_i = reinterpret_cast<const f32x4*>(atoms + (size_t)node * 24);
#pragma unroll
      for (int a = 0; a < A; ++a) {
        Ai[a] = atom_i[a];
#pragma unroll
        for (int c = 0; c < KT; ++c) {
          const f32x4 q = reinterpret_cast<const f32x4*>(atoms + (size_t)jn[c] * 24)[a];
          Bj[a][c][0] = q.x; Bj[a][c][1] = q.y; Bj[a][c][2] = q.z;
        }
      }
    }
    f32x4 acc[MPNN_NT][KT];
#pragma unroll
    for (int t = 0; t < MPNN_NT; ++t) {
      f32x4 b0 = f32x4{0.f, 0.f, 0.f, 0.f};
      if (emb_bias) b0 = *reinterpret_cast<const f32x4*>(emb_bias + 16 * t + 4 * g);
#pragma unroll
      for (int c = 0; c < KT; ++c) acc[t][c] = b0;
    }
    auto feature_tile = [&](int tt, f32x4 (&x)[KT]) {
      if (tt < A * A) {
        const f32x4 P = Ai[tt / A];
#pragma unroll
        for (int c = 0; c < KT; ++c) {
          const float dx = P.x - Bj[tt % A][c][0], dy = P.y - Bj[tt % A][c][1], dz = P.z - Bj[tt % A][c][2];
          const float d = rbf_distance(((dx * dx + dy * dy) + dz * dz) + 1e-6f);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float z = fmaf(d, 0.96089792702915982f, -muc[r]);  // (d - mu) (1 / 1.25) sqrt(log2 e), one instruction
            x[c][r] = __builtin_amdgcn_exp2f(-(z * z));               // exp(-((d - mu) / 1.25)^2)
          }
        }
      } else if (tt == A * A) {
        const long long ri = residue_idx[node], ci = chain_labels[node];
#pragma unroll
        for (int c = 0; c < KT; ++c) {
          const long long off = ri - residue_idx[jn[c]];
          const bool same = (ci - chain_labels[jn[c]]) == 0;
          long long d = off + 32;
          d = d < 0 ? 0 : (d > 64 ? 64 : d);
          if (!same) d = 65;
          x[c] = *reinterpret_cast<const f32x4*>(pos_table + d * 16 + 4 * g);
        }
      } else {
#pragma unroll
        for (int c = 0; c < KT; ++c) x[c] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    };
#pragma unroll
    for (int kk = 0; kk < NKK_IN; ++kk, ++step) {
      // B operand of this k-step (no LDS involved) and the slot's hand-over.  Waves 0-3 generate the operand BEFORE
      // the barrier, waves 4-7 (their SIMD partners) AFTER it: between two barriers one partner is in its MFMAs while
      // the other is in its VALU work, instead of both sharing the matrix pipe and then both leaving it idle
      BFrag bf[KT];
      auto operand = [&]() {
        f32x4 x0[KT], x1[KT];
        feature_tile(2 * kk, x0);
        feature_tile(2 * kk + 1, x1);
#pragma unroll
        for (int c = 0; c < KT; ++c) split_pair(x0[c], x1[c], bf[c]);
      };
      const bool early = wave < FEAT_WAVES / 2;
      if (early) operand();
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's pieces of the images in flight have landed
      __syncthreads();
      if (step + R - 1 < n_steps) fill(step + R - 1);  // into the slot the previous k-step read
      if (!early) operand();
      const h8* slot = ring + (step % R) * FEAT_KSTEP_H8;
#pragma unroll
      for (int to = 0; to < MPNN_NT; ++to) {
        const h8 a_hi = slot[(to * 2 + 0) * 64 + lane];
#ifndef MPNN_FAST
        const h8 a_lo = slot[(to * 2 + 1) * 64 + lane];
#pragma unroll
        for (int c = 0; c < KT; ++c) acc[to][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_lo, bf[c].hi, acc[to][c], 0, 0, 0);
#pragma unroll
        for (int c = 0; c < KT; ++c) acc[to][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_hi, bf[c].lo, acc[to][c], 0, 0, 0);
#endif
#pragma unroll
        for (int c = 0; c < KT; ++c) acc[to][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_hi, bf[c].hi, acc[to][c], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    const size_t foff = (size_t)node * (MPNN_NT * KT * 64) + lane;
    if (SAVE && has) {
#pragma unroll
      for (int t = 0; t < MPNN_NT; ++t)
#pragma unroll
        for (int c = 0; c < KT; ++c) __builtin_nontemporal_store(acc[t][c], &E0_frag[foff + (t * KT + c) * 64]);
    }
    tile_layernorm<KT>(acc, ln_w, ln_b, lane);
    if (SAVE && has) {
#pragma unroll
      for (int t = 0; t < MPNN_NT; ++t)
#pragma unroll
        for (int c = 0; c < KT; ++c) __builtin_nontemporal_store(acc[t][c], &E_frag[foff + (t * KT + c) * 64]);
    }
    if (E_nat && has) {
#pragma unroll
      for (int c = 0; c < KT; ++c) {
        const int e = 16 * c + col;
        if (e < K) {
          float* dst = E_nat + ((size_t)node * K + e) * MPNN_H + 4 * g;
#pragma unroll
          for (int t = 0; t < MPNN_NT; ++t) *reinterpret_cast<f32x4*>(dst + 16 * t) = acc[t][c];
        }
      }
    }
    BFrag eb[MPNN_NT / 2][KT];
    tile_to_bfrag<MPNN_NT, KT, false>(acc, eb);
#pragma unroll
    for (int t = 0; t < MPNN_NT; ++t) {
      const f32x4 bb = *reinterpret_cast<const f32x4*>(be + 16 * t + 4 * g);
#pragma unroll
      for (int c = 0; c < KT; ++c) acc[t][c] = bb;
    }
    split_gemm<KT, MPNN_NT / 2, 0, MPNN_NT>(wes, lane, eb, acc);
    if (has) {
      f32x4* out = reinterpret_cast<f32x4*>(hE) + (size_t)node * (MPNN_NT * KT * 64);
#pragma unroll
      for (int t = 0; t < MPNN_NT; ++t)
#pragma unroll
        for (int c = 0; c < KT; ++c) __builtin_nontemporal_store(acc[t][c], &out[(t * KT + c) * 64 + lane]);
    }
  }
}

// ---------------------------------------------------------------------------
// fragment layout <-> natural layout converters (API surfaces that expose h_E)
// ---------------------------------------------------------------------------
__global__ void k_frag_to_natural(const float* __restrict__ hE, int N, int K, int KT, float* __restrict__ nat) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;  // one float4 of the fragment tensor
  const size_t total = (size_t)N * MPNN_NT * KT * 64;
  if (idx >= total) return;
  const int lane = idx & 63;
  const size_t q = idx >> 6;
  const int c = q % KT, t = (q / KT) % MPNN_NT;
  const size_t node = q / ((size_t)KT * MPNN_NT);
  const int e = 16 * c + (lane & 15);
  if (e >= K) return;
  const f32x4 v = reinterpret_cast<const f32x4*>(hE)[idx];
  *reinterpret_cast<f32x4*>(nat + (node * K + e) * MPNN_H + 16 * t + 4 * (lane >> 4)) = v;
}

__global__ void k_natural_to_frag(const float* __restrict__ nat, int N, int K, int KT, float* __restrict__ hE) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t total = (size_t)N * MPNN_NT * KT * 64;
  if (idx >= total) return;
  const int lane = idx & 63;
  const size_t q = idx >> 6;
  const int c = q % KT, t = (q / KT) % MPNN_NT;
  const size_t node = q / ((size_t)KT * MPNN_NT);
  const int e = 16 * c + (lane & 15);
  f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
  if (e < K) v = *reinterpret_cast<const f32x4*>(nat + (node * K + e) * MPNN_H + 16 * t + 4 * (lane >> 4));
  reinterpret_cast<f32x4*>(hE)[idx] = v;
}

// ---------------------------------------------------------------------------
// decoding order -> rank (inverse permutation) and the causal masks (:485-519)
//   mask_bw[i,k] = m_i [rank_j < rank_i] (em_ik),  mask_fw[i,k] = m_i (1 - [..]) (em_ik)
// ---------------------------------------------------------------------------
__global__ void k_order_rank(const long long* __restrict__ order, int B, int L, int* __restrict__ rank) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= B * L) return;
  const int b = idx / L, pos = idx - b * L;
  const long long q = order[idx];
  if (q >= 0 && q < L) rank[b * L + (int)q] = pos;
}

// Dependency level of every residue under a decoding order (sampler scheduling): walking the order once,
//   level[i] = 1 + max{ level[j] : j neighbour of i, rank[j] < rank[i] }.
// One wavefront per (sample, structure) row keeps the row's levels in LDS; lanes = neighbour slots.
// Only the level lookup lv[j] depends on the previous step; the index chain of a step (order -> neighbour list ->
// ranks: three dependent global loads) does not, so it runs three chunks of 16 steps ahead of the sequential part, one
// stage per chunk (software pipeline in registers): 310 -> ~60 us for L = 512.
constexpr int LVC = 16;  // steps per chunk
__global__ void __launch_bounds__(64)
k_decode_levels(const long long* __restrict__ order, const int* __restrict__ rank, const int* __restrict__ E_idx,
                int enc_rows, int L, int K, int Kp, int* __restrict__ level) {
  extern __shared__ int lv[];
  const int row = blockIdx.x, p = row % enc_rows, lane = threadIdx.x;
  const long long* ord = order + (size_t)row * L;
  const int* rk = rank + (size_t)row * L;
  for (int t = lane; t < L; t += 64) lv[t] = 0;
  __builtin_amdgcn_wave_barrier();
  const int nchunk = (L + LVC - 1) / LVC;
  // stage 1 (chunk c): the residues decoded at steps c*16 .. c*16+15, one per lane (-1: past the end or out of range -
  // the host wrapper rejects non-permutations, the kernel stays in bounds regardless)
  auto stage1 = [&](int c) -> int {
    const int t = c * LVC + lane;
    int i = -1;
    if (lane < LVC && t < L) {
      const long long iq = ord[t];
      if (iq >= 0 && iq < L) i = (int)iq;
    }
    return i;
  };
  // stage 2: this lane's neighbour of each of the chunk's residues
  auto stage2 = [&](int iv, int (&j)[LVC]) {
#pragma unroll
    for (int u = 0; u < LVC; ++u) {
      const int i = __shfl(iv, u);
      j[u] = (i >= 0 && lane < K) ? E_idx[((size_t)p * L + i) * Kp + lane] : -1;
    }
  };
  // stage 3: is that neighbour decoded before step t ?  (j < 0 -> no)
  auto stage3 = [&](int c, const int (&j)[LVC], int (&src)[LVC]) {
#pragma unroll
    for (int u = 0; u < LVC; ++u) src[u] = (j[u] >= 0 && rk[j[u]] < c * LVC + u) ? j[u] : -1;
  };
  int i1 = stage1(0), i2, i3 = -1, i4 = -1;       // residues of the chunks in stages 1, 2, 3 and in the sequential part
  int j2[LVC], j3[LVC], s3[LVC], s4[LVC];
  // prologue: fill the pipeline (chunk 0 through stage 3, chunk 1 through stage 2, chunk 2 through stage 1)
  stage2(i1, j2); i2 = i1; i1 = stage1(1);
#pragma unroll
  for (int u = 0; u < LVC; ++u) j3[u] = j2[u];
  i3 = i2; stage3(0, j3, s3);
  stage2(i1, j2); i2 = i1; i1 = stage1(2);
  for (int c = 0; c < nchunk; ++c) {
    // hand chunk c to the sequential part, advance the stages behind it
#pragma unroll
    for (int u = 0; u < LVC; ++u) { s4[u] = s3[u]; j3[u] = j2[u]; }
    i4 = i3; i3 = i2;
    stage3(c + 1, j3, s3);
    stage2(i1, j2); i2 = i1;
    i1 = stage1(c + 3);
#pragma unroll
    for (int u = 0; u < LVC; ++u) {
      const int i = __builtin_amdgcn_readlane(i4, u);
      if (i < 0) continue;  // wave-uniform
      int m = s4[u] >= 0 ? lv[s4[u]] : 0;
      // wave-wide max by DPP (levels are >= 0, lanes shifted in from outside a row contribute 0): four row shifts,
      // then lane 15 of a row into the next row, lane 31 into the upper half; the total sits in lane 63.  Six VALU
      // instructions instead of six LDS-routed shuffles on the one dependent chain of the kernel.
      m = max(m, __builtin_amdgcn_update_dpp(0, m, 0x111, 0xf, 0xf, false));  // row_shr:1
      m = max(m, __builtin_amdgcn_update_dpp(0, m, 0x112, 0xf, 0xf, false));  // row_shr:2
      m = max(m, __builtin_amdgcn_update_dpp(0, m, 0x114, 0xf, 0xf, false));  // row_shr:4
      m = max(m, __builtin_amdgcn_update_dpp(0, m, 0x118, 0xf, 0xf, false));  // row_shr:8
      m = max(m, __builtin_amdgcn_update_dpp(0, m, 0x142, 0xa, 0xf, false));  // row_bcast:15 -> rows 1, 3
      m = max(m, __builtin_amdgcn_update_dpp(0, m, 0x143, 0xc, 0xf, false));  // row_bcast:31 -> rows 2, 3
      m = __builtin_amdgcn_readlane(m, 63);
      if (lane == 0) lv[i] = m + 1;
      // lane 0's LDS store must be visible to the other lanes' loads of the next step (LDS only: the levels go to
      // global memory in one pass at the end - a global store per step would sit in every step's release fence)
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
    }
  }
  for (int t = lane; t < L; t += 64) level[(size_t)row * L + t] = lv[t];
}

__global__ void k_order_masks(const int* __restrict__ rank, const long long* __restrict__ E_idx64,
                              const float* __restrict__ mask, const float* __restrict__ edge_mask, int B, int L,
                              int K, float* __restrict__ mask_bw, float* __restrict__ mask_fw) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)B * L * K) return;
  const size_t node = idx / K;
  const int b = node / L;
  const int j = (int)E_idx64[idx];
  const float att = rank[node] > rank[b * L + j] ? 1.0f : 0.0f;
  float bw = mask[node] * att, fw = mask[node] * (1.0f - att);
  if (edge_mask) {
    bw *= edge_mask[idx];
    fw *= edge_mask[idx];
  }
  mask_bw[idx] = bw;
  mask_fw[idx] = fw;
}

MPNN_KERNELS_END

// ===========================================================================
// C ABI launchers (declared in include/mpnn_hip.h)
// ===========================================================================

#ifndef MPNN_FAST
extern "C" int mpnn_backbone_atoms(const float* X, const float* noise, const float* mask, int B, int L,
                                   float* atoms, float* ca_soa, void* stream) {
  MPNN_CHECK_ARG(X && mask && atoms && ca_soa && B > 0 && L > 0, "mpnn_backbone_atoms: bad arguments");
  const int N = B * L;
  hipLaunchKernelGGL(k_backbone_atoms, dim3(ceil_div(N, 256)), dim3(256), 0, (hipStream_t)stream, X, noise, mask, N,
                     L, atoms, ca_soa);
  MPNN_LAUNCH_CHECK();
}
#endif

#ifndef MPNN_FAST
extern "C" int mpnn_knn_topk(const float* ca_soa, int B, int L, int K, float edge_cutoff, int64_t* E_idx64,
                             int32_t* E_idx32, float* D_neighbors, float* edge_mask, void* stream) {
  MPNN_CHECK_ARG(ca_soa && E_idx32 && B > 0 && L > 0, "mpnn_knn_topk: bad arguments");
  MPNN_CHECK_ARG(K >= 1 && K <= 64 && K <= L, "mpnn_knn_topk: K must satisfy 1 <= K <= min(64, L)");
  const int Kp = 16 * ((K + 15) / 16);
  const int has_cut = edge_cutoff > 0.f;
  MPNN_CHECK_ARG(has_cut || !edge_mask, "mpnn_knn_topk: edge_mask requested without a cutoff");
  hipLaunchKernelGGL(k_knn_topk, dim3(ceil_div((long long)B * L, 4)), dim3(256), 0, (hipStream_t)stream, ca_soa, B, L,
                     K, Kp, has_cut, edge_cutoff, (long long*)E_idx64, E_idx32, D_neighbors, edge_mask);
  MPNN_LAUNCH_CHECK();
}
#endif

#ifndef MPNN_FAST
extern "C" int mpnn_occupancy(const float* atoms, const float* mask, int B, int L, float cutoff, float* occ,
                              void* stream) {
  MPNN_CHECK_ARG(atoms && mask && occ && B > 0 && L > 0 && cutoff > 0.f, "mpnn_occupancy: bad arguments");
  hipLaunchKernelGGL(k_occupancy, dim3(ceil_div((long long)B * L, 4)), dim3(256), 0, (hipStream_t)stream, atoms, mask,
                     B, L, cutoff, occ);
  MPNN_LAUNCH_CHECK();
}
#endif

#ifndef MPNN_FAST
extern "C" int mpnn_pack_linear(const float* W, int out_f, int in_f, int ld, int col0, float* packed, void* stream) {
  MPNN_CHECK_ARG(W && packed && out_f > 0 && in_f > 0 && ld >= col0 + in_f, "mpnn_pack_linear: bad arguments");
  const int nto = (out_f + 15) / 16, nti = (in_f + 15) / 16;
  hipLaunchKernelGGL(k_pack_linear, dim3(ceil_div((long long)nto * nti * 64, 256)), dim3(256), 0, (hipStream_t)stream,
                     W, out_f, in_f, ld, col0, packed, nto, nti);
  MPNN_LAUNCH_CHECK();
}
#endif

#ifndef MPNN_FAST
extern "C" int mpnn_pack_linear_split(const float* W, int out_f, int in_f, int ld, int col0, int rot, void* packed,
                                      void* stream) {
  MPNN_CHECK_ARG(W && packed && out_f > 0 && in_f > 0 && ld >= col0 + in_f && rot >= 0 && rot < in_f,
                 "mpnn_pack_linear_split: bad arguments");
  const int nto = (out_f + 15) / 16, nkk = (in_f + 31) / 32;
  hipLaunchKernelGGL(k_pack_linear_split, dim3(ceil_div((long long)nto * nkk * 64, 256)), dim3(256), 0,
                     (hipStream_t)stream, W, out_f, in_f, ld, col0, rot, (h8*)packed, nto, nkk);
  MPNN_LAUNCH_CHECK();
}
#endif

#ifndef MPNN_FAST
extern "C" int mpnn_pos_table(const float* W, const float* bias, int nbuckets, int npos, float* table, void* stream) {
  MPNN_CHECK_ARG(W && bias && table && nbuckets == 66 && npos == 16, "mpnn_pos_table: expects a 16x66 linear");
  hipLaunchKernelGGL(k_pos_table, dim3(ceil_div(nbuckets * npos, 256)), dim3(256), 0, (hipStream_t)stream, W, bias,
                     nbuckets, npos, table);
  MPNN_LAUNCH_CHECK();
}
#endif

template <int KT>
static int launch_edge_features(int A, const float* atoms, const int* E_idx, const long long* ridx,
                                const long long* chain, const float* pos_table, const float* rbf_mu,
                                const float* Wemb, const float* ln_w, const float* ln_b, const float* We,
                                const float* be, int N, int L, int K, float* E_nat, float* hE, hipStream_t s,
                                float* E0_frag = nullptr, float* E_frag = nullptr, const float* emb_bias = nullptr) {
  // K <= 48: persistent workgroups with the weights staged through LDS; above, the streaming form (its four column
  // tiles leave no registers for the gathered atoms)
  constexpr bool ring = KT <= 3 && MPNN_FEAT_PERSISTENT;
  dim3 grid(ring ? edge_grid(N, FEAT_WAVES) : ceil_div(N, 4)), block(ring ? FEAT_WAVES * 64 : 256);
#define MPNN_FEAT_LAUNCH(AA, SV)                                                                                  \
  if (ring)                                                                                                       \
    hipLaunchKernelGGL((k_edge_features_ring<(KT <= 3 ? KT : 3), AA, SV>), grid, block, 0, s, atoms, E_idx, ridx, chain, \
                       pos_table, rbf_mu, (const h8*)Wemb, ln_w, ln_b, (const h8*)We, be, N, L, K, E_nat, hE,     \
                       (f32x4*)E0_frag, (f32x4*)E_frag, emb_bias);                                                \
  else                                                                                                            \
    hipLaunchKernelGGL((k_edge_features<KT, AA, SV>), grid, block, 0, s, atoms, E_idx, ridx, chain, pos_table, rbf_mu, \
                       (const h8*)Wemb, ln_w, ln_b, (const h8*)We, be, N, L, K, E_nat, hE, (f32x4*)E0_frag,       \
                       (f32x4*)E_frag, emb_bias)
  const bool save = E0_frag != nullptr;
  if (A == 5) {
    if (save) { MPNN_FEAT_LAUNCH(5, true); } else { MPNN_FEAT_LAUNCH(5, false); }
  } else {
    if (save) { MPNN_FEAT_LAUNCH(6, true); } else { MPNN_FEAT_LAUNCH(6, false); }
  }
#undef MPNN_FEAT_LAUNCH
  MPNN_LAUNCH_CHECK();
}

static int edge_features_entry(const float* atoms, const int32_t* E_idx32, const int64_t* residue_idx,
                               const int64_t* chain_labels, const float* pos_table, const float* rbf_mu,
                               const float* Wemb_packed, const float* Wemb_bias, const float* ln_w, const float* ln_b,
                               const float* We_packed, const float* We_bias, int B, int L, int K, int n_atoms,
                               float* E_natural, float* hE_frag, void* stream) {
  MPNN_CHECK_ARG(atoms && E_idx32 && residue_idx && chain_labels && pos_table && rbf_mu && Wemb_packed && ln_w &&
                     ln_b && We_packed && We_bias && hE_frag,
                 "mpnn_edge_features: null pointer");
  MPNN_CHECK_ARG(B > 0 && L > 0 && K >= 1 && K <= 64, "mpnn_edge_features: need 1 <= K <= 64");
  MPNN_CHECK_ARG(n_atoms == 5 || n_atoms == 6, "mpnn_edge_features: n_atoms must be 5 or 6");
  const int KT = (K + 15) / 16, N = B * L;
  hipStream_t s = (hipStream_t)stream;
  const long long *ri = (const long long*)residue_idx, *cl = (const long long*)chain_labels;
#define MPNN_FEAT_CASE(kt)                                                                                          \
  return launch_edge_features<kt>(n_atoms, atoms, E_idx32, ri, cl, pos_table, rbf_mu, Wemb_packed, ln_w, ln_b,      \
                                  We_packed, We_bias, N, L, K, E_natural, hE_frag, s, nullptr, nullptr, Wemb_bias)
  switch (KT) {
    case 1: MPNN_FEAT_CASE(1);
    case 2: MPNN_FEAT_CASE(2);
    case 3: MPNN_FEAT_CASE(3);
    default: MPNN_FEAT_CASE(4);
  }
#undef MPNN_FEAT_CASE
}

extern "C" int MPNN_API(mpnn_edge_features)(const float* atoms, const int32_t* E_idx32, const int64_t* residue_idx,
                                  const int64_t* chain_labels, const float* pos_table, const float* rbf_mu,
                                  const float* Wemb_packed, const float* ln_w, const float* ln_b,
                                  const float* We_packed, const float* We_bias, int B, int L, int K, int n_atoms,
                                  float* E_natural, float* hE_frag, void* stream) {
  return edge_features_entry(atoms, E_idx32, residue_idx, chain_labels, pos_table, rbf_mu, Wemb_packed, nullptr, ln_w,
                             ln_b, We_packed, We_bias, B, L, K, n_atoms, E_natural, hE_frag, stream);
}

// the same kernel with a bias on the embedding before its LayerNorm (the sparse variant's edge_mlp.0)
extern "C" int MPNN_API(mpnn_edge_features_biased)(const float* atoms, const int32_t* E_idx32,
                                         const int64_t* residue_idx, const int64_t* chain_labels,
                                         const float* pos_table, const float* rbf_mu, const float* Wemb_packed,
                                         const float* Wemb_bias, const float* ln_w, const float* ln_b,
                                         const float* We_packed, const float* We_bias, int B, int L, int K,
                                         int n_atoms, float* E_natural, float* hE_frag, void* stream) {
  MPNN_CHECK_ARG(Wemb_bias, "mpnn_edge_features_biased: null bias");
  return edge_features_entry(atoms, E_idx32, residue_idx, chain_labels, pos_table, rbf_mu, Wemb_packed, Wemb_bias, ln_w,
                             ln_b, We_packed, We_bias, B, L, K, n_atoms, E_natural, hE_frag, stream);
}

#ifndef MPNN_FAST
extern "C" int mpnn_edge_features_train(const float* atoms, const int32_t* E_idx32, const int64_t* residue_idx,
                                        const int64_t* chain_labels, const float* pos_table, const float* rbf_mu,
                                        const float* Wemb_packed, const float* Wemb_bias, const float* ln_w,
                                        const float* ln_b, const float* We_packed, const float* We_bias, int B, int L,
                                        int K, int n_atoms, float* E0_frag, float* E_frag, float* hE_frag, void* stream) {
  MPNN_CHECK_ARG(atoms && E_idx32 && residue_idx && chain_labels && pos_table && rbf_mu && Wemb_packed && ln_w &&
                     ln_b && We_packed && We_bias && hE_frag && E0_frag && E_frag,
                 "mpnn_edge_features_train: null pointer");
  MPNN_CHECK_ARG(B > 0 && L > 0 && K >= 1 && K <= 64, "mpnn_edge_features_train: need 1 <= K <= 64");
  MPNN_CHECK_ARG(n_atoms == 5 || n_atoms == 6, "mpnn_edge_features_train: n_atoms must be 5 or 6");
  const int KT = (K + 15) / 16, N = B * L;
  hipStream_t s = (hipStream_t)stream;
  const long long *ri = (const long long*)residue_idx, *cl = (const long long*)chain_labels;
  switch (KT) {
    case 1: return launch_edge_features<1>(n_atoms, atoms, E_idx32, ri, cl, pos_table, rbf_mu, Wemb_packed, ln_w, ln_b, We_packed, We_bias, N, L, K, nullptr, hE_frag, s, E0_frag, E_frag, Wemb_bias);
    case 2: return launch_edge_features<2>(n_atoms, atoms, E_idx32, ri, cl, pos_table, rbf_mu, Wemb_packed, ln_w, ln_b, We_packed, We_bias, N, L, K, nullptr, hE_frag, s, E0_frag, E_frag, Wemb_bias);
    case 3: return launch_edge_features<3>(n_atoms, atoms, E_idx32, ri, cl, pos_table, rbf_mu, Wemb_packed, ln_w, ln_b, We_packed, We_bias, N, L, K, nullptr, hE_frag, s, E0_frag, E_frag, Wemb_bias);
    default: return launch_edge_features<4>(n_atoms, atoms, E_idx32, ri, cl, pos_table, rbf_mu, Wemb_packed, ln_w, ln_b, We_packed, We_bias, N, L, K, nullptr, hE_frag, s, E0_frag, E_frag, Wemb_bias);
  }
}
#endif

#ifndef MPNN_FAST
MPNN_KERNELS_BEGIN
// ---------------------------------------------------------------------------
// Training: the featurizer's raw inputs, materialised once per backward pass in the fragment layout so that the
// generic per-edge weight-gradient kernel (mpnn_edge_wgrad) can contract them with dE0:
//   feature tile T < A*A          : the 16 RBFs of atom pair T                       (reference :334-347)
//   feature tiles A*A .. A*A + 4  : one-hot of the relative-position bucket d (66)    (reference :349-356, :170-176)
// tile T lives in chunk T / 8 (a chunk = 128 features = one hE-shaped array), chunks stored back to back.
// ---------------------------------------------------------------------------
template <int A>
__global__ void __launch_bounds__(256)
k_feature_chunks(const float* __restrict__ atoms, const int* __restrict__ E_idx, const long long* __restrict__ residue_idx,
                 const long long* __restrict__ chain_labels, const float* __restrict__ rbf_mu, int N, int L, int KT,
                 f32x4* __restrict__ F) {
  constexpr int NTILE = A * A + 5;
  constexpr int NCH = (NTILE + 7) / 8;
  const int lane = threadIdx.x & 63;
  const int node = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (node >= N) return;
  const int g = lane >> 4, col = lane & 15;
  const int b = node / L, Kp = KT * 16;
  float muc[4];  // RBF centres times (1 / 1.25) sqrt(log2 e)
#pragma unroll
  for (int r = 0; r < 4; ++r) muc[r] = rbf_mu[4 * g + r] * 0.96089792702915982f;
  const f32x4* atom_i = reinterpret_cast<const f32x4*>(atoms + (size_t)node * 24);
  const long long ri = residue_idx[node], ci = chain_labels[node];
  const size_t chunk = (size_t)N * MPNN_NT * KT * 64;
  for (int c = 0; c < KT; ++c) {
    const int jn = b * L + E_idx[(size_t)node * Kp + 16 * c + col];
    const f32x4* atom_j = reinterpret_cast<const f32x4*>(atoms + (size_t)jn * 24);
    long long d = ri - residue_idx[jn] + 32;
    d = d < 0 ? 0 : (d > 64 ? 64 : d);
    if (ci != chain_labels[jn]) d = 65;
#pragma unroll
    for (int T = 0; T < NCH * 8; ++T) {
      f32x4 x = f32x4{0.f, 0.f, 0.f, 0.f};
      if (T < A * A) {
        const f32x4 Ai = atom_i[T / A], Bj = atom_j[T % A];
        const float dx = Ai.x - Bj.x, dy = Ai.y - Bj.y, dz = Ai.z - Bj.z;
        const float dist = rbf_distance(((dx * dx + dy * dy) + dz * dz) + 1e-6f);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float z = fmaf(dist, 0.96089792702915982f, -muc[r]);
          x[r] = __builtin_amdgcn_exp2f(-(z * z));
        }
      } else if (T < NTILE) {
        const int f0 = 16 * (T - A * A) + 4 * g;
#pragma unroll
        for (int r = 0; r < 4; ++r) x[r] = (f0 + r == (int)d) ? 1.0f : 0.0f;
      }
      F[(size_t)(T / 8) * chunk + ((size_t)node * MPNN_NT + (T % 8)) * KT * 64 + c * 64 + lane] = x;
    }
  }
}

MPNN_KERNELS_END
#endif
#ifndef MPNN_FAST
extern "C" int mpnn_feature_chunks(const float* atoms, const int32_t* E_idx32, const int64_t* residue_idx,
                                   const int64_t* chain_labels, const float* rbf_mu, int B, int L, int K, int n_atoms,
                                   float* chunks_frag, void* stream) {
  MPNN_CHECK_ARG(atoms && E_idx32 && residue_idx && chain_labels && rbf_mu && chunks_frag,
                 "mpnn_feature_chunks: null pointer");
  MPNN_CHECK_ARG(B > 0 && L > 0 && K >= 1 && K <= 64 && (n_atoms == 5 || n_atoms == 6), "mpnn_feature_chunks: bad sizes");
  const int KT = (K + 15) / 16, N = B * L;
  dim3 grid(ceil_div(N, 4)), block(256);
  const long long *ri = (const long long*)residue_idx, *cl = (const long long*)chain_labels;
  if (n_atoms == 5)
    hipLaunchKernelGGL(k_feature_chunks<5>, grid, block, 0, (hipStream_t)stream, atoms, E_idx32, ri, cl, rbf_mu, N, L, KT,
                       (f32x4*)chunks_frag);
  else
    hipLaunchKernelGGL(k_feature_chunks<6>, grid, block, 0, (hipStream_t)stream, atoms, E_idx32, ri, cl, rbf_mu, N, L, KT,
                       (f32x4*)chunks_frag);
  MPNN_LAUNCH_CHECK();
}
#endif

#ifndef MPNN_FAST
extern "C" int mpnn_edges_to_natural(const float* hE_frag, int n_nodes, int K, float* natural, void* stream) {
  MPNN_CHECK_ARG(hE_frag && natural && n_nodes > 0 && K >= 1 && K <= 64, "mpnn_edges_to_natural: bad arguments");
  const int KT = (K + 15) / 16;
  const long long total = (long long)n_nodes * MPNN_NT * KT * 64;
  hipLaunchKernelGGL(k_frag_to_natural, dim3(ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream, hE_frag,
                     n_nodes, K, KT, natural);
  MPNN_LAUNCH_CHECK();
}
#endif

#ifndef MPNN_FAST
extern "C" int mpnn_edges_from_natural(const float* natural, int n_nodes, int K, float* hE_frag, void* stream) {
  MPNN_CHECK_ARG(hE_frag && natural && n_nodes > 0 && K >= 1 && K <= 64, "mpnn_edges_from_natural: bad arguments");
  const int KT = (K + 15) / 16;
  const long long total = (long long)n_nodes * MPNN_NT * KT * 64;
  hipLaunchKernelGGL(k_natural_to_frag, dim3(ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream, natural,
                     n_nodes, K, KT, hE_frag);
  MPNN_LAUNCH_CHECK();
}
#endif

#ifndef MPNN_FAST
extern "C" int mpnn_order_rank(const int64_t* order, int B, int L, int32_t* rank, void* stream) {
  MPNN_CHECK_ARG(order && rank && B > 0 && L > 0, "mpnn_order_rank: bad arguments");
  hipLaunchKernelGGL(k_order_rank, dim3(ceil_div((long long)B * L, 256)), dim3(256), 0, (hipStream_t)stream,
                     (const long long*)order, B, L, rank);
  MPNN_LAUNCH_CHECK();
}
#endif

#ifndef MPNN_FAST
extern "C" int mpnn_decode_levels(const int64_t* order, const int32_t* rank, const int32_t* E_idx32, int n_rows,
                                  int enc_rows, int L, int K, int32_t* level, void* stream) {
  MPNN_CHECK_ARG(order && rank && E_idx32 && level, "mpnn_decode_levels: null pointer");
  MPNN_CHECK_ARG(n_rows > 0 && enc_rows > 0 && n_rows % enc_rows == 0 && L > 0 && L <= 16384 && K >= 1 && K <= 64,
                 "mpnn_decode_levels: bad sizes (L <= 16384, K <= 64)");
  hipLaunchKernelGGL(k_decode_levels, dim3(n_rows), dim3(64), (size_t)L * sizeof(int), (hipStream_t)stream,
                     (const long long*)order, rank, E_idx32, enc_rows, L, K, 16 * ((K + 15) / 16), level);
  MPNN_LAUNCH_CHECK();
}
#endif

#ifndef MPNN_FAST
extern "C" int mpnn_order_masks(const int32_t* rank, const int64_t* E_idx64, const float* mask,
                                const float* edge_mask, int B, int L, int K, float* mask_bw, float* mask_fw,
                                void* stream) {
  MPNN_CHECK_ARG(rank && E_idx64 && mask && mask_bw && mask_fw && B > 0 && L > 0 && K > 0,
                 "mpnn_order_masks: bad arguments");
  hipLaunchKernelGGL(k_order_masks, dim3(ceil_div((long long)B * L * K, 256)), dim3(256), 0, (hipStream_t)stream, rank,
                     (const long long*)E_idx64, mask, edge_mask, B, L, K, mask_bw, mask_fw);
  MPNN_LAUNCH_CHECK();
}
#endif
