// Backward of the per-edge kernels (training step; reference: src/mpnn/models/model_utils.py EncLayer :48-95,
// DecLayer :98-138 differentiated by torch autograd in the reference's training loop).
//
// The training forward (mpnn_message_train / mpnn_enc_edge_update_train, mpnn_edge.hip) saves the MLP
// pre-activations in the fragment layout.  Backward of one MLP is the same register-resident chain as
// forward, run with transposed weights:
//   message:   da2 = dgsum_i w_ik gelu'(a2)          dg1 = W2^T da2        da1 = dg1 gelu'(a1)
//              dP_i = sum_k da1      dRow_ik = s_ik da1      de_ik (+)= W1b^T (s_ik da1)
//   update:    dz = LayerNorm'(dy; z)   dg2 = W13^T dz   da2 = dg2 gelu'(a2)   dg1 = W12^T da2
//              da1 = dg1 gelu'(a1)      dP'_i = sum_k da1     dQ'_j += da1     de_ik = dz + W11b^T da1
// The per-edge gradients da* are written out once (fragment layout, or row-major "natural" rows where a
// scatter to neighbour nodes follows) and feed
//   k_wgrad        dW[fo][fi] (+)= sum_edges A[fo, e] B[fi, e]  (split-f16 MFMA over LDS-transposed tiles)
//   k_gather_rows  dRow_j = sum over the edges that point at j (inverse neighbour list; no atomics)
#include "mpnn_bwd_common.h"

#define MPNN_BWD_WAVES 8

// y <- y * gelu'(a) with a streamed from the saved pre-activations; per-feature sums over the residue's edges
// go to dP (the h_i projection's gradient); the (optionally s-scaled) rows are written row-major for the gather
template <int KT, bool SCALE>
__device__ __forceinline__ void first_layer_grad(f32x4 (&y)[MPNN_NT][KT], float unscale, const f32x4* __restrict__ a,
                                                 const float (&s)[KT], float* __restrict__ dP, float* __restrict__ nat,
                                                 int lane) {
  const int g = lane >> 4, col = lane & 15;
  // the whole saved tensor of the residue in flight at once (the B fragments of the GEMM before are dead: registers),
  // pinned ahead of the arithmetic: tile-by-tile the wave paid one HBM round trip per row tile
  f32x4 av[MPNN_NT][KT];
#pragma unroll
  for (int t = 0; t < MPNN_NT; ++t)
#pragma unroll
    for (int c = 0; c < KT; ++c) av[t][c] = __builtin_nontemporal_load(&a[(t * KT + c) * 64]);
  FENCE();
#pragma unroll
  for (int t = 0; t < MPNN_NT; ++t) {
    f32x4 ps = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < KT; ++c) {
      y[t][c] *= gelu_grad4(av[t][c]) * unscale;
      ps += y[t][c];
      if (SCALE) y[t][c] *= s[c];
      __builtin_nontemporal_store(y[t][c], reinterpret_cast<f32x4*>(nat + (size_t)(16 * c + col) * MPNN_H + 16 * t + 4 * g));
    }
    ps = column_sum4(ps);
    if (col == 0) *reinterpret_cast<f32x4*>(dP + 16 * t + 4 * g) = ps;
    FENCE();
  }
}

// ---------------------------------------------------------------------------
// message backward (encoder DEC = false, decoder DEC = true)
// ---------------------------------------------------------------------------
template <int KT, bool DEC>
__global__ void __launch_bounds__(MPNN_BWD_WAVES * 64)
k_message_bwd(const float* __restrict__ dgsum, const f32x4* __restrict__ a1, const f32x4* __restrict__ a2,
              const int* __restrict__ E_idx, const float* __restrict__ mask, const float* __restrict__ edge_mask,
              const h8* __restrict__ W2T, const h8* __restrict__ W1bT, int N, int L, int K,
              f32x4* __restrict__ da2, float* __restrict__ da1_nat, float* __restrict__ dP, f32x4* __restrict__ dE,
              int accumulate, unsigned* __restrict__ amax, int w_valid) {
  __shared__ h8 lds[2 * WSPLIT];
  __shared__ unsigned amax_l[2];
  lds_fill(lds, W2T, WSPLIT);
  lds_fill(lds + WSPLIT, W1bT, WSPLIT);
  if (threadIdx.x < 2) amax_l[threadIdx.x] = 0u;
  __syncthreads();
  const h8* W2s = lds;
  const h8* W1s = lds + WSPLIT;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane >> 4, col = lane & 15;
  constexpr int Kp = KT * 16;
  const int stride = gridDim.x * MPNN_BWD_WAVES;
  for (int node = blockIdx.x * MPNN_BWD_WAVES + wave; node < N; node += stride) {
    LAUNDER(dgsum);
    const int b = node / L;
    const float mi = mask[node];
    float s[KT], w[KT];
#pragma unroll
    for (int c = 0; c < KT; ++c) {
      const int e = 16 * c + col;
      float em = 1.0f;
      if (edge_mask && e < K) em = edge_mask[(size_t)node * K + e];
      if (DEC) {
        s[c] = mi * em;
        w[c] = e < K ? (w_valid ? em : 1.0f) : 0.0f;  // w_valid: the sum ran over valid edges only (sparse variant)
      } else {
        s[c] = 1.0f;
        w[c] = e < K ? mi * mask[b * L + E_idx[(size_t)node * Kp + e]] * em : 0.0f;
      }
    }
    const size_t off = (size_t)node * (MPNN_NT * KT * 64) + lane;
    f32x4 d[MPNN_NT][KT];
    {
      const float* dg = dgsum + (size_t)node * MPNN_H + 4 * g;
#pragma unroll
      for (int t = 0; t < MPNN_NT; ++t)
#pragma unroll
        for (int c = 0; c < KT; ++c) d[t][c] = __builtin_nontemporal_load(&a2[off + (t * KT + c) * 64]);
      FENCE();
#pragma unroll
      for (int t = 0; t < MPNN_NT; ++t) {
        const f32x4 gs = row_tile(dg, t);
#pragma unroll
        for (int c = 0; c < KT; ++c) {
          d[t][c] = gelu_grad4(d[t][c]) * (gs * w[c]);
          __builtin_nontemporal_store(d[t][c], &da2[off + (t * KT + c) * 64]);
        }
        FENCE();
      }
    }
    BFrag db[NKK][KT];
    float un = grad_to_bfrag<KT>(d, db, &amax_l[0]);
    f32x4 y[MPNN_NT][KT];
    zero_tile<KT>(y);
    split_gemm<KT, NKK, 0, MPNN_NT>(W2s, lane, db, y);
    first_layer_grad<KT, DEC>(y, un, a1 + off, s, dP + (size_t)node * MPNN_H, da1_nat + (size_t)node * Kp * MPNN_H,
                              lane);
    un = grad_to_bfrag<KT>(y, db, &amax_l[1]);
    zero_tile<KT>(d);
    split_gemm<KT, NKK, 0, MPNN_NT>(W1s, lane, db, d);
    if (accumulate) {  // uniform
      f32x4 prev[MPNN_NT][KT];
#pragma unroll
      for (int t = 0; t < MPNN_NT; ++t)
#pragma unroll
        for (int c = 0; c < KT; ++c) prev[t][c] = dE[off + (t * KT + c) * 64];
      FENCE();
#pragma unroll
      for (int t = 0; t < MPNN_NT; ++t)
#pragma unroll
        for (int c = 0; c < KT; ++c) __builtin_nontemporal_store(d[t][c] * un + prev[t][c], &dE[off + (t * KT + c) * 64]);
    } else {
#pragma unroll
      for (int t = 0; t < MPNN_NT; ++t)
#pragma unroll
        for (int c = 0; c < KT; ++c) __builtin_nontemporal_store(d[t][c] * un, &dE[off + (t * KT + c) * 64]);
    }
  }
  __syncthreads();
  if (threadIdx.x < 2) atomicMax(amax + threadIdx.x, amax_l[threadIdx.x]);
}

// ---------------------------------------------------------------------------
// encoder edge-update backward.  dE may alias dy (each wave reads its residue's dy before writing).
// LayerNorm parameter gradients: every wave sums its own residues in LDS, the workgroup adds its waves in wave order
// and stores the partial to ln_part[blockIdx.x][256]; the launcher's k_partials_reduce adds the workgroups in a
// fixed order into dln[0:128] (weight) and dln[128:256] (bias).  No floating-point atomics: bit-reproducible.
// ---------------------------------------------------------------------------
// W11b^T stream of the edge-update backward: fragments in flight ahead of their MFMAs, and the pointer's address space
#ifndef MPNN_BWD_W1_RING
#define MPNN_BWD_W1_RING 3
#endif
#define MPNN_BWD_W1_PTR(p) (p)
template <int KT>
__global__ void __launch_bounds__(MPNN_BWD_WAVES * 64)
k_edge_update_bwd(const f32x4* dy, const f32x4* __restrict__ z, const f32x4* __restrict__ a2,
                  const f32x4* __restrict__ a1, const float* __restrict__ ln_w, const h8* __restrict__ W13T,
                  const h8* __restrict__ W12T, const h8* __restrict__ W11bT, int N, f32x4* __restrict__ dz_out,
                  f32x4* __restrict__ da2_out, float* __restrict__ da1_nat, float* __restrict__ dP, f32x4* dE,
                  float* __restrict__ ln_part, unsigned* __restrict__ amax, unsigned drop_thresh, unsigned drop_seed,
                  float inv_keep) {
  __shared__ h8 lds[2 * WSPLIT];
  __shared__ float lnacc_w[MPNN_BWD_WAVES][2 * MPNN_H];
  __shared__ unsigned amax_l[3];
  if (threadIdx.x < 3) amax_l[threadIdx.x] = 0u;
  lds_fill(lds, W13T, WSPLIT);
  lds_fill(lds + WSPLIT, W12T, WSPLIT);
  for (int i = threadIdx.x; i < MPNN_BWD_WAVES * 2 * MPNN_H; i += blockDim.x) (&lnacc_w[0][0])[i] = 0.f;
  __syncthreads();
  float* lnacc = lnacc_w[threadIdx.x >> 6];
  const h8* W3s = lds;
  const h8* W2s = lds + WSPLIT;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane >> 4, col = lane & 15;
  constexpr int Kp = KT * 16;
  const int stride = gridDim.x * MPNN_BWD_WAVES;
  for (int node = blockIdx.x * MPNN_BWD_WAVES + wave; node < N; node += stride) {
    LAUNDER(ln_w);
    const h8* W1u = W11bT;
    LAUNDER_FLAT(W1u);
    const size_t off = (size_t)node * (MPNN_NT * KT * 64) + lane;
    f32x4 x[MPNN_NT][KT], d[MPNN_NT][KT];
#pragma unroll
    for (int t = 0; t < MPNN_NT; ++t)
#pragma unroll
      for (int c = 0; c < KT; ++c) {
        x[t][c] = __builtin_nontemporal_load(&z[off + (t * KT + c) * 64]);
        d[t][c] = dy[off + (t * KT + c) * 64];
      }
    tile_layernorm_bwd<KT>(x, d, ln_w, lnacc, lane);
    // dz goes two ways: unchanged along the residual (parked in dE until the end), and through the dropout mask of
    // the MLP output (:94) into W13^T / the W13 and b13 gradients (dz_out)
#pragma unroll
    for (int t = 0; t < MPNN_NT; ++t)
#pragma unroll
      for (int c = 0; c < KT; ++c) {
        if (drop_thresh) {  // without dropout dz_out itself serves the residual at the end: one tensor write less
          dE[off + (t * KT + c) * 64] = d[t][c];
          d[t][c] = dropout4(d[t][c], (unsigned)(off + (t * KT + c) * 64) * 4u, drop_seed, drop_thresh, inv_keep);
        }
        dz_out[off + (t * KT + c) * 64] = d[t][c];
      }
    BFrag db[NKK][KT];
    float un = grad_to_bfrag<KT>(d, db, &amax_l[0]);
    // ---- through W13, gelu'(a2)
    zero_tile<KT>(x);
    split_gemm<KT, NKK, 0, MPNN_NT>(W3s, lane, db, x);
    {
      f32x4 av[MPNN_NT][KT];  // db is dead: registers for the whole saved tensor in flight at once
#pragma unroll
      for (int t = 0; t < MPNN_NT; ++t)
#pragma unroll
        for (int c = 0; c < KT; ++c) av[t][c] = __builtin_nontemporal_load(&a2[off + (t * KT + c) * 64]);
      FENCE();
#pragma unroll
      for (int t = 0; t < MPNN_NT; ++t) {
#pragma unroll
        for (int c = 0; c < KT; ++c) {
          x[t][c] *= gelu_grad4(av[t][c]) * un;
          __builtin_nontemporal_store(x[t][c], &da2_out[off + (t * KT + c) * 64]);
        }
        FENCE();
      }
    }
    un = grad_to_bfrag<KT>(x, db, &amax_l[1]);
    // ---- through W12, gelu'(a1)
    zero_tile<KT>(d);
    split_gemm<KT, NKK, 0, MPNN_NT>(W2s, lane, db, d);
    float s1[KT];
#pragma unroll
    for (int c = 0; c < KT; ++c) s1[c] = 1.0f;
    first_layer_grad<KT, false>(d, un, a1 + off, s1, dP + (size_t)node * MPNN_H, da1_nat + (size_t)node * Kp * MPNN_H,
                                lane);
    un = grad_to_bfrag<KT>(d, db, &amax_l[2]);
    // ---- de = dz (residual) + W11b^T da1 ; W11b^T streams from L2
    zero_tile<KT>(x);
    split_gemm_stream<KT, NKK, MPNN_NT, MPNN_BWD_W1_RING>(MPNN_BWD_W1_PTR(W1u) + lane, db, x);
    {
      const f32x4* rsrc = drop_thresh ? dE : dz_out;
      f32x4 res[MPNN_NT][KT];
#pragma unroll
      for (int t = 0; t < MPNN_NT; ++t)
#pragma unroll
        for (int c = 0; c < KT; ++c) res[t][c] = rsrc[off + (t * KT + c) * 64];
      FENCE();
#pragma unroll
      for (int t = 0; t < MPNN_NT; ++t)
#pragma unroll
        for (int c = 0; c < KT; ++c) __builtin_nontemporal_store(x[t][c] * un + res[t][c], &dE[off + (t * KT + c) * 64]);
    }
  }
  __syncthreads();
  if (threadIdx.x < 2 * MPNN_H) {
    float t = lnacc_w[0][threadIdx.x];
#pragma unroll
    for (int w = 1; w < MPNN_BWD_WAVES; ++w) t += lnacc_w[w][threadIdx.x];
    ln_part[(size_t)blockIdx.x * 2 * MPNN_H + threadIdx.x] = t;
  }
  if (threadIdx.x < 3) atomicMax(amax + threadIdx.x, amax_l[threadIdx.x]);  // a maximum: order-independent
}

// ---------------------------------------------------------------------------
// featurizer backward (reference :357-358 norm_edges(edge_embedding(.)) followed by W_e, :464):
//   dE = W_e^T dhE0;   dE0 = LayerNorm'(dE; E0)   ->  dE0 (fragment layout, feeds mpnn_edge_wgrad against the
//   feature chunks), norm_edges weight / bias gradients, and the magnitudes the weight-gradient kernel scales by
//   (amax[0] = max |dhE0|, amax[1] = max |dE0|).
// ---------------------------------------------------------------------------
template <int KT>
__global__ void __launch_bounds__(MPNN_BWD_WAVES * 64)
k_features_bwd(const f32x4* __restrict__ dy, const f32x4* __restrict__ E0, const float* __restrict__ ln_w,
               const h8* __restrict__ WeT, int N, f32x4* __restrict__ dE0, float* __restrict__ ln_part,
               unsigned* __restrict__ amax) {
  __shared__ h8 lds[WSPLIT];
  __shared__ float lnacc_w[MPNN_BWD_WAVES][2 * MPNN_H];
  __shared__ unsigned amax_l[2];
  lds_fill(lds, WeT, WSPLIT);
  for (int i = threadIdx.x; i < MPNN_BWD_WAVES * 2 * MPNN_H; i += blockDim.x) (&lnacc_w[0][0])[i] = 0.f;
  if (threadIdx.x < 2) amax_l[threadIdx.x] = 0u;
  __syncthreads();
  float* lnacc = lnacc_w[threadIdx.x >> 6];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int stride = gridDim.x * MPNN_BWD_WAVES;
  for (int node = blockIdx.x * MPNN_BWD_WAVES + wave; node < N; node += stride) {
    LAUNDER(ln_w);
    const size_t off = (size_t)node * (MPNN_NT * KT * 64) + lane;
    f32x4 x[MPNN_NT][KT], d[MPNN_NT][KT];
#pragma unroll
    for (int t = 0; t < MPNN_NT; ++t)
#pragma unroll
      for (int c = 0; c < KT; ++c) x[t][c] = dy[off + (t * KT + c) * 64];
    BFrag db[NKK][KT];
    const float un = grad_to_bfrag<KT>(x, db, &amax_l[0]);
    zero_tile<KT>(d);
    split_gemm<KT, NKK, 0, MPNN_NT>(lds, lane, db, d);
#pragma unroll
    for (int t = 0; t < MPNN_NT; ++t)
#pragma unroll
      for (int c = 0; c < KT; ++c) {
        d[t][c] *= un;
        x[t][c] = E0[off + (t * KT + c) * 64];
      }
    tile_layernorm_bwd<KT>(x, d, ln_w, lnacc, lane);
    const unsigned mb = tile_absmax_bits<KT>(d);
    if (lane == 0) atomicMax(&amax_l[1], mb);
#pragma unroll
    for (int t = 0; t < MPNN_NT; ++t)
#pragma unroll
      for (int c = 0; c < KT; ++c) __builtin_nontemporal_store(d[t][c], &dE0[off + (t * KT + c) * 64]);
  }
  __syncthreads();
  if (threadIdx.x < 2 * MPNN_H) {
    float t = lnacc_w[0][threadIdx.x];
#pragma unroll
    for (int w = 1; w < MPNN_BWD_WAVES; ++w) t += lnacc_w[w][threadIdx.x];
    ln_part[(size_t)blockIdx.x * 2 * MPNN_H + threadIdx.x] = t;
  }
  if (threadIdx.x < 2) atomicMax(amax + threadIdx.x, amax_l[threadIdx.x]);
}

// ---------------------------------------------------------------------------
// weight gradient of one 128x128 per-edge matrix:
//   dW[fo][fi] += sum_{node, e} A[node, e, fo] * Bt(B[node, e, fi]),   db[fo] += sum A[node, e, fo]
// A: fragment layout (A_NAT = false) or row-major rows [node * Kp + e][128]; B: fragment layout, optionally
// through GELU (the layer input was gelu(a)).  A workgroup (4 waves) stages one residue at a time into LDS as
// f16 hi/lo planes [feature][edge] (edges contiguous = the MFMA k dimension), then each wave accumulates its
// 2 x 8 output tiles with v_mfma_f32_16x16x16_f16 (3 products per tile: lo*hi + hi*lo + hi*hi).
// ---------------------------------------------------------------------------
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
constexpr int WG_PART = MPNN_H * MPNN_H + MPNN_H;  // floats of one workgroup partial (dW | dbias)
// padded edge stride (halves) of the LDS planes: 16 KTM + 4 (104-byte rows at KTM = 3 spread the 8-byte operand reads over
// banks); KTM = 3 for K <= 48 (three workgroups per CU), 4 for K <= 64 (one)

// 4x4 transpose across a lane quad: out[j] of quad lane i = in[i] of quad lane j (two DPP exchange stages)
__device__ __forceinline__ f32x4 quad_transpose(f32x4 v) {
  const int lane = threadIdx.x;
  const bool odd = lane & 1, hi = lane & 2;
  auto xchg = [](float x, int ctrl) {
    return __int_as_float(ctrl == 0xB1 ? __builtin_amdgcn_update_dpp(0, __float_as_int(x), 0xB1, 0xf, 0xf, false)
                                       : __builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x4E, 0xf, 0xf, false));
  };
  // stage 1: lanes i <-> i^1 swap the low index bit
  float r01 = xchg(odd ? v.x : v.y, 0xB1), r23 = xchg(odd ? v.z : v.w, 0xB1);
  if (odd) { v.x = r01; v.z = r23; } else { v.y = r01; v.w = r23; }
  // stage 2: lanes i <-> i^2 swap the high index bit
  float r02 = xchg(hi ? v.x : v.z, 0x4E), r13 = xchg(hi ? v.y : v.w, 0x4E);
  if (hi) { v.x = r02; v.y = r13; } else { v.z = r02; v.w = r13; }
  return v;
}

// x -> f16 hi (truncated) + f16 lo = f16(x - hi): v_cvt_pkrtz for two hi halves, one v_fma_mix per lo half.  The values
// come from DPP moves / loads (never straight from an MFMA) and go to LDS stores.
__device__ __forceinline__ void split4(const f32x4 x, h4& hi, h4& lo) {
  unsigned h[2], l[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    h[j] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(x[2 * j], x[2 * j + 1]));
    float t = x[2 * j];  // lo in place over x[2j], tied as a float (split_pair in mpnn_common.h says why)
    asm("v_fma_mixlo_f16 %0, -%1, 1.0, %0 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "+v"(t) : "v"(h[j]));
    asm("v_fma_mixhi_f16 %0, -%1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(t) : "v"(h[j]), "v"(x[2 * j + 1]));
    l[j] = __builtin_bit_cast(unsigned, t);
  }
  typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
  hi = __builtin_bit_cast(h4, (u32x2_t){h[0], h[1]});
  lo = __builtin_bit_cast(h4, (u32x2_t){l[0], l[1]});
}

#define WGRAD_LOAD(p) __builtin_nontemporal_load(p)
#ifndef MPNN_WGRAD_WAVES
#define MPNN_WGRAD_WAVES 8  // waves per workgroup: 8 (one 16-row tile of dW per wave, ~100 registers, two workgroups =
                            // 16 waves per CU) against 4 (two tiles, ~165 registers, three workgroups = 12 waves):
                            // 3.39 against 3.71 ms per training step - the launch lives on loads in flight
#endif
#ifndef MPNN_WGRAD_WG_PER_CU
#define MPNN_WGRAD_WG_PER_CU (MPNN_WGRAD_WAVES == 8 ? 2 : 3)
#endif
// FEAT = 5 / 6 (atoms per residue): the B operand is not read but COMPUTED - chunk `fa.chunk` (128 features = 8 tiles)
// of the featurizer's raw inputs, tile T = 8 chunk + wave: the 16 RBFs of atom pair T (T < FEAT^2), the one-hot of the
// relative-position bucket (the next five tiles), zero beyond (reference :334-356, :170-176).  Round 2 wrote these
// chunks to memory first (mpnn_feature_chunks: 1 GB per step) and read them back here.  The neighbour ids are loaded
// two residues ahead and the neighbours' atoms one ahead, so neither wait is exposed.
struct FeatArgs {
  const float* atoms;           // [N][6][4]
  const int* E_idx;             // [N][Kp] row-local neighbour ids
  const long long* residue_idx; // [N]
  const long long* chain;       // [N]
  const float* rbf_mu;          // [16]
  int L, chunk;
};

template <bool A_NAT, bool B_GELU, int KTM, int FEAT = 0>
__global__ void __launch_bounds__(64 * MPNN_WGRAD_WAVES)
k_wgrad(const float* __restrict__ A, const f32x4* __restrict__ Bf, const unsigned* __restrict__ amax, int N, int KT,
        float* __restrict__ part, FeatArgs fa) {
  static_assert(FEAT == 0 || MPNN_WGRAD_WAVES == 8, "the feature form gives each wave one tile");
  constexpr int WG_LDK = 16 * KTM + 4;
  __shared__ _Float16 sA[2][MPNN_H][WG_LDK];
  __shared__ _Float16 sB[2][MPNN_H][WG_LDK];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane >> 4, col = lane & 15;
  const int Kp = KT * 16;
  constexpr int TT = 8 / MPNN_WGRAD_WAVES;  // feature tiles (16 rows of dW) per wave
  f32x4 acc[TT][MPNN_NT];
#pragma unroll
  for (int i = 0; i < TT; ++i)
#pragma unroll
    for (int j = 0; j < MPNN_NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 bacc[TT];
#pragma unroll
  for (int i = 0; i < TT; ++i) bacc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  const f32x4* Af = reinterpret_cast<const f32x4*>(A);
  const float sc = pow2_scale_from_bits(*amax), un = pow2_inverse(sc);  // see grad_to_bfrag
  // this thread's share of a residue: feature tiles t = TT wave + {0 .. TT-1}, every column tile
  f32x4 va[TT][KTM], vb[TT][KTM];
  // feature form: this wave's tile, the neighbour ids of the residue whose operands are fetched next, and the raw
  // values its B tile is computed from (neighbour atom of the pair / neighbour's residue number and chain)
  const int T = FEAT ? 8 * fa.chunk + wave : 0;
  constexpr int NPAIR = FEAT * FEAT;
  int jn[KTM];
  f32x4 own = f32x4{0.f, 0.f, 0.f, 0.f};
  long long ri = 0, ci = 0;
  float muc[4] = {0.f, 0.f, 0.f, 0.f};
  if (FEAT) {
#pragma unroll
    for (int r = 0; r < 4; ++r) muc[r] = fa.rbf_mu[4 * g + r] * 0.96089792702915982f;
  }
  auto load_ids = [&](int node) {
#pragma unroll
    for (int c = 0; c < KTM; ++c)
      if (c < KT) jn[c] = (node / fa.L) * fa.L + fa.E_idx[(size_t)node * Kp + 16 * c + col];
  };
  auto feature_raw = [&](int node, int c) {  // -> vb[0][c]: xyz of the pair's neighbour atom | (residue number, chain)
    if (T < NPAIR) {
      vb[0][c] = reinterpret_cast<const f32x4*>(fa.atoms + (size_t)jn[c] * 24)[T % (FEAT ? FEAT : 1)];
    } else if (T < NPAIR + 5) {
      const long long rj = fa.residue_idx[jn[c]], cj = fa.chain[jn[c]];
      vb[0][c] = f32x4{__int_as_float((int)rj), __int_as_float((int)(rj >> 32)), __int_as_float((int)cj),
                       __int_as_float((int)(cj >> 32))};
    }
  };
  auto feature_value = [&](f32x4 raw) -> f32x4 {
    f32x4 x = f32x4{0.f, 0.f, 0.f, 0.f};
    if (T < NPAIR) {
      const float dx = own.x - raw.x, dy = own.y - raw.y, dz = own.z - raw.z;
      const float dist = __builtin_amdgcn_sqrtf(((dx * dx + dy * dy) + dz * dz) + 1e-6f);  // as the forward kernel's
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float z = fmaf(dist, 0.96089792702915982f, -muc[r]);
        x[r] = __builtin_amdgcn_exp2f(-(z * z));
      }
    } else if (T < NPAIR + 5) {
      const long long rj = ((long long)(unsigned)__float_as_int(raw.x)) | ((long long)__float_as_int(raw.y) << 32);
      const long long cj = ((long long)(unsigned)__float_as_int(raw.z)) | ((long long)__float_as_int(raw.w) << 32);
      long long d = ri - rj + 32;
      d = d < 0 ? 0 : (d > 64 ? 64 : d);
      if (ci != cj) d = 65;
      const int f0 = 16 * (T - NPAIR) + 4 * g;
#pragma unroll
      for (int r = 0; r < 4; ++r) x[r] = (f0 + r == (int)d) ? 1.0f : 0.0f;
    }
    return x;
  };
  auto fetch_one = [&](int node, int tt, int c) {
    const int t = TT * wave + tt;
    if (A_NAT)
      va[tt][c] = WGRAD_LOAD(
          reinterpret_cast<const f32x4*>(A + ((size_t)node * Kp + 16 * c + col) * MPNN_H + 16 * t + 4 * g));
    else
      va[tt][c] = WGRAD_LOAD(&Af[((size_t)node * MPNN_NT + t) * KT * 64 + c * 64 + lane]);
    if (FEAT)
      feature_raw(node, c);
    else
      vb[tt][c] = WGRAD_LOAD(&Bf[((size_t)node * MPNN_NT + t) * KT * 64 + c * 64 + lane]);
  };
  auto fetch = [&](int node) {
#pragma unroll
    for (int tt = 0; tt < TT; ++tt)
#pragma unroll
      for (int c = 0; c < KTM; ++c)
        if (c < KT) fetch_one(node, tt, c);
  };
  int node = blockIdx.x;
  const int Nend = N, nstep = gridDim.x;
  // the residue the staging loop works on owns `own_cur` / ri / ci; the values fetched for the next one wait in *_nx
  f32x4 own_nx = own;
  long long ri_nx = 0, ci_nx = 0;
  auto fetch_own = [&](int nd) {
    if (T < NPAIR) own_nx = reinterpret_cast<const f32x4*>(fa.atoms + (size_t)nd * 24)[T / (FEAT ? FEAT : 1)];
    else if (T < NPAIR + 5) { ri_nx = fa.residue_idx[nd]; ci_nx = fa.chain[nd]; }
  };
  if (node < Nend) {
    if (FEAT) { load_ids(node); fetch_own(node); }
    fetch(node);
    if (FEAT && node + nstep < Nend) load_ids(node + nstep);  // ids one residue further than the operands
  }
  for (; node < Nend; node += nstep) {
    if (FEAT) { own = own_nx; ri = ri_nx; ci = ci_nx; }
#ifdef MPNN_ABL_WGRAD_STREAM  // timing-only ablation build (never shipped): the operand streams alone
#pragma unroll
    for (int tt = 0; tt < TT; ++tt)
#pragma unroll
      for (int c = 0; c < KTM; ++c)
        if (c < KT) acc[tt][c] += va[tt][c] * vb[tt][c];
    if (node + nstep < Nend) fetch(node + nstep);
    continue;
#endif
    __syncthreads();  // previous residue's operand reads are done
#pragma unroll
    for (int tt = 0; tt < TT; ++tt) {
      const int t = TT * wave + tt;
#pragma unroll
      for (int c = 0; c < KTM; ++c) {
        if (c < KT) {
          f32x4 a = va[tt][c], b = FEAT ? feature_value(vb[tt][c]) : vb[tt][c];
          if (B_GELU) b = gelu4(b);
          bacc[tt] += a;
          a *= sc;
          // registers hold 4 features of one edge; the MFMA operand rows want 4 edges of one feature: transpose the
          // 4x4 block held by each lane quad (DPP), then one 8-byte LDS write per plane instead of four 2-byte ones
          a = quad_transpose(a);
          b = quad_transpose(b);
          h4 ah, al, bh, bl;
          split4(a, ah, al);
          split4(b, bh, bl);
          const int frow = 16 * t + 4 * g + (col & 3), ecol = 16 * c + 4 * (col >> 2);
          *reinterpret_cast<h4*>(&sA[0][frow][ecol]) = ah;
          *reinterpret_cast<h4*>(&sA[1][frow][ecol]) = al;
          *reinterpret_cast<h4*>(&sB[0][frow][ecol]) = bh;
          *reinterpret_cast<h4*>(&sB[1][frow][ecol]) = bl;
        }
      }
    }
    __syncthreads();
    if (node + nstep < Nend) {  // next residue's loads fly under the MFMAs
      if (FEAT) fetch_own(node + nstep);
      fetch(node + nstep);      // (feature form: with the ids loaded one iteration ago)
      if (FEAT && node + 2 * nstep < Nend) load_ids(node + 2 * nstep);
    }
    for (int c = 0; c < KT; ++c) {
      h4 ah[TT], al[TT];
#pragma unroll
      for (int i = 0; i < TT; ++i) {
        ah[i] = *reinterpret_cast<const h4*>(&sA[0][16 * (TT * wave + i) + col][16 * c + 4 * g]);
        al[i] = *reinterpret_cast<const h4*>(&sA[1][16 * (TT * wave + i) + col][16 * c + 4 * g]);
      }
#pragma unroll
      for (int j = 0; j < MPNN_NT; ++j) {
        const h4 bh = *reinterpret_cast<const h4*>(&sB[0][16 * j + col][16 * c + 4 * g]);
        const h4 bl = *reinterpret_cast<const h4*>(&sB[1][16 * j + col][16 * c + 4 * g]);
#pragma unroll
        for (int i = 0; i < TT; ++i) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x16f16(al[i], bh, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x16f16(ah[i], bl, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x16f16(ah[i], bh, acc[i][j], 0, 0, 0);
        }
      }
    }
  }
  // this workgroup's partial [128 x 128 | 128 bias sums] -> workspace (plain stores; k_wgrad_reduce adds them up:
  // 2^23 same-address-set atomics per launch from 512 workgroups cost more than the whole contraction)
  float* P = part + (size_t)blockIdx.x * WG_PART;
#pragma unroll
  for (int i = 0; i < TT; ++i)
#pragma unroll
    for (int j = 0; j < MPNN_NT; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) P[(16 * (TT * wave + i) + 4 * g + r) * MPNN_H + 16 * j + col] = acc[i][j][r] * un;
#pragma unroll
  for (int tt = 0; tt < TT; ++tt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float v = column_sum(bacc[tt][r]);
      if (col == 0) P[MPNN_H * MPNN_H + 16 * (TT * wave + tt) + 4 * g + r] = v;
    }
}

// dW[i] += sum_w part[w][i]: 16 slices of the workgroup partials per element, met in LDS in slice order (no atomics:
// the same bits on every run).  A thread owns four consecutive elements (16-byte loads: 16 threads cover 256
// contiguous bytes of a partial; the dword form read half-lines and took 13-15 us per launch, 17 launches a step).
__device__ __forceinline__ void wgrad_reduce_body(const float* __restrict__ part, int nparts, size_t stride,
                                                  float* __restrict__ dW, float* __restrict__ db) {
  __shared__ f32x4 sl[16][17];
  const int q = threadIdx.x & 15, slice = threadIdx.x >> 4;
  const int i4 = blockIdx.x * 16 + q;  // quad of elements
  f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
  if (i4 < WG_PART / 4) {
    const f32x4* p = reinterpret_cast<const f32x4*>(part) + i4;
#pragma unroll 8
    for (int w = slice; w < nparts; w += 16) s += p[(size_t)w * (stride / 4)];
  }
  sl[slice][q] = s;
  __syncthreads();
  if (slice == 0 && i4 < WG_PART / 4) {
    f32x4 t = sl[0][q];
#pragma unroll
    for (int k = 1; k < 16; ++k) t += sl[k][q];
    if (i4 < MPNN_H * MPNN_H / 4) {
      f32x4* d = reinterpret_cast<f32x4*>(dW) + i4;
      *d = *d + t;
    } else if (db) {
      f32x4* d = reinterpret_cast<f32x4*>(db) + (i4 - MPNN_H * MPNN_H / 4);
      *d = *d + t;
    }
  }
}
static_assert(WG_PART % 4 == 0 && (MPNN_H * MPNN_H) % 4 == 0, "quads must not straddle the dW / dbias boundary");

__global__ void __launch_bounds__(256)
k_wgrad_reduce(const float* __restrict__ part, int nparts, size_t stride, float* __restrict__ dW,
               float* __restrict__ db) {
  wgrad_reduce_body(part, nparts, stride, dW, db);
}

// ---------------------------------------------------------------------------
// Message backward with BOTH weight-gradient contractions fused in (K <= 48): mpnn_message_bwd + two mpnn_edge_wgrad
// passes in one kernel, so da2 never exists in memory and a1 / hE / the da1 rows are not re-read by separate
// contraction launches (per message MLP and step: 6 instead of 10 edge tensors of traffic).
//
// The chain kernels give a residue to ONE wave (weights in LDS, activations in registers).  A weight gradient is a sum
// over ALL residues of an outer product whose 128 x 128 fp32 accumulator no single wave can hold, so here a residue
// belongs to the WORKGROUP: wave w owns feature tile w (16 of the 128 features) of every tensor, keeps the 16 rows of
// W2^T and W1b^T that produce its tile in REGISTERS (the A operands of the chain's two GEMMs) and its 16 x 128 row
// blocks of dW2 and dW1b as accumulators.  Activations travel between the waves through LDS in two forms:
//   Bf  [k-step][column tile][hi | lo][lane]   the B fragments of the next GEMM (k = features: every wave needs all),
//   sA / sB [hi | lo][feature][edge]           f16 planes with the EDGES contiguous: the MFMA k dimension of the
//                                              contractions dW[o][i] += sum_e A[e][o] B[e][i] (as in k_wgrad).
// Per residue: da2 = gelu'(a2) dgsum w -> (barrier A) -> dg1 = W2^T da2, da1 = dg1 gelu'(a1), dP, the s-scaled rows
// for the gather | dW2 += da2 (x) gelu(a1) -> (barrier B) -> dE (+)= W1b^T rows | dW1b += rows (x) e.
// REGISTERS decide this kernel: 132 hold the weights and accumulators, 48 the four operand tiles of a residue (each
// re-requested for the NEXT residue right after its last use: a workgroup runs in lock-step, nothing else on the CU
// covers an exposed round trip).  The first form spilled a hundred registers, and spill traffic (per-wave scratch:
// 80 MB over the launch, far beyond L2) tripled the kernel's HBM bytes (rocprofv3 FETCH_SIZE 2.9 GB against 1.0 GB) -
// every GEMM / contraction below therefore handles ONE column tile at a time with a fenced two-deep LDS read pipeline,
// and global addresses are a scalar residue base plus a fixed 32-bit lane offset.
// Gradient operands are scaled by an exact power of two before their f16 hi/lo split, launch-wide, from bounds
// (k_fused_scales).  Per-workgroup partial sums [2][128 x 128 + 128] go to the workspace; k_wgrad_reduce adds them in a
// fixed order.
// ---------------------------------------------------------------------------
constexpr int FB_WAVES = 8;
#ifdef MPNN_ABL_FUSED_NOBARRIER  // timing-only ablation build: never shipped
#define FB_BARRIER() FENCE()
#else
#define FB_BARRIER() __syncthreads()
#endif
constexpr int FB_LDK = 16 * 3 + 4;  // padded edge stride (halves) of the planes, as WG_LDK of k_wgrad at KTM = 3

// scales[0] = max |dgsum| (float bits), scales[1] = max_k sum_o |W2[o][k]| (float bits); scales[] zeroed before.
__global__ void __launch_bounds__(256)
k_fused_scales(const float* __restrict__ dgsum, size_t n, const h8* __restrict__ W2T, unsigned* __restrict__ scales) {
  float m = 0.f;
  const f32x4* d4 = reinterpret_cast<const f32x4*>(dgsum);  // n is a multiple of 128
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n / 4; i += (size_t)gridDim.x * 256) {
    const f32x4 v = d4[i];
    m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  __shared__ float wm[4];
  if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
  __syncthreads();
  // one atomic per workgroup (a thousand waves on one address serialise in L2); a maximum: order-independent
  if (threadIdx.x == 0)
    atomicMax(&scales[0], (unsigned)__float_as_int(fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3]))));
  if (blockIdx.x == 0 && threadIdx.x < MPNN_H) {
    // row `col` of output tile `to` of W2^T lives in lanes col, 16 + col, 32 + col, 48 + col of every k-step
    const int to = threadIdx.x >> 4, col = threadIdx.x & 15;
    float sum = 0.f;
    for (int kk = 0; kk < NKK; ++kk)
      for (int g = 0; g < 4; ++g) {
        const h8 hi = W2T[((to * NKK + kk) * 2 + 0) * 64 + 16 * g + col];
        const h8 lo = W2T[((to * NKK + kk) * 2 + 1) * 64 + 16 * g + col];
#pragma unroll
        for (int q = 0; q < 8; ++q) sum += fabsf((float)hi[q] + (float)lo[q]);
      }
    atomicMax(&scales[1], (unsigned)__float_as_int(sum));
  }
}

// k_wgrad_reduce for up to three partial blocks of one workspace in ONE launch (blockIdx.y = which)
struct ReduceJobs {
  const float* part[3];
  float* dW[3];
  float* db[3];
};
__global__ void __launch_bounds__(256)
k_wgrad_reduce_n(ReduceJobs jobs, int nparts, size_t stride) {
  wgrad_reduce_body(jobs.part[blockIdx.y], nparts, stride, jobs.dW[blockIdx.y], jobs.db[blockIdx.y]);
}

template <int KT, bool DEC>
__global__ void __launch_bounds__(FB_WAVES * 64)
k_message_bwd_fused(const float* __restrict__ dgsum, const f32x4* __restrict__ a1, const f32x4* __restrict__ a2,
                    const f32x4* __restrict__ hE, const int* __restrict__ E_idx, const float* __restrict__ mask,
                    const float* __restrict__ edge_mask, const h8* __restrict__ W2T, const h8* __restrict__ W1bT, int N,
                    int L, int K, float* __restrict__ da1_nat, float* __restrict__ dP, f32x4* __restrict__ dE,
                    int accumulate, int w_valid, const unsigned* __restrict__ scales, float* __restrict__ part,
                    unsigned* __restrict__ amax_out) {
  __shared__ h8 Bf[2][NKK][KT][2][64];
  __shared__ _Float16 sA[2][2][MPNN_H][FB_LDK];
  __shared__ _Float16 sB[2][2][MPNN_H][FB_LDK];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane >> 4, col = lane & 15;
  constexpr int Kp = KT * 16;
  // launch-wide power-of-two scales from BOUNDS: |da2| <= 1.13 max |dgsum| (gelu' <= 1.13, w <= 1) and
  // |rows| <= 1.13 |W2^T da2| <= 1.13 max_k sum_o |W2[o][k]| max |da2|.  A bound that is loose by 2^b costs b of the
  // ~36 bits between the scaled maximum and the f16 subnormal step of the lo part.
  const float m1 = __int_as_float((int)scales[0]) * 1.1293f;
  const float m2 = m1 * __int_as_float((int)scales[1]) * 1.1293f;
  const float sc1 = pow2_scale_from_bits((unsigned)__float_as_int(m1)), un1 = pow2_inverse(sc1);
  const float sc2 = pow2_scale_from_bits((unsigned)__float_as_int(m2)), un2 = pow2_inverse(sc2);
  // this wave's rows of the two transposed weights: the A fragments of its output tile, all k-steps
  // (W2^T stays in registers; W1b^T's rows are re-requested from L2 for every residue behind the second phase's
  // arithmetic and live through the third phase only: with both resident the K = 48 instance spilled 40 registers)
  h8 w2h[NKK], w2l[NKK];
#pragma unroll
  for (int kk = 0; kk < NKK; ++kk) {
    w2h[kk] = W2T[((wave * NKK + kk) * 2 + 0) * 64 + lane];
    w2l[kk] = W2T[((wave * NKK + kk) * 2 + 1) * 64 + lane];
  }
  const h8* W1w = W1bT + (size_t)wave * NKK * 2 * 64 + lane;
  // weight-gradient accumulators: this wave's 32 x 64 block of each 128 x 128 matrix, as two 32 x 32 MFMA tiles
  // (rows 32 (wave >> 1) .., columns 64 (wave & 1) + 32 b ..)
  typedef float f32x16 __attribute__((ext_vector_type(16)));
  f32x16 accW2[2], accW1[2];
#pragma unroll
  for (int b = 0; b < 2; ++b)
#pragma unroll
    for (int v = 0; v < 16; ++v) accW2[b][v] = accW1[b][v] = 0.f;
  __shared__ float sacc[MPNN_H];  // db2: one writer lane per word, in residue order (first touched after a barrier)
  if (threadIdx.x < MPNN_H) sacc[threadIdx.x] = 0.f;
  float vmax = 0.f;
  // where this lane's share of a (feature tile, column tile) goes in the planes / in a B fragment
  const int frow = 16 * wave + 4 * g + (col & 3);
  const int bf_kk = wave >> 1, bf_half = (wave & 1) * 4;  // halves [bf_half, bf_half + 4) of k-step bf_kk
  const int G = gridDim.x;

  // the contraction of one stage: acc += sum_e A[rows of this wave's block][e] B[columns of its block][e] (scaled by the
  // stage's sc), on v_mfma_f32_32x32x16_f16: 16 edges per instruction, so the 48 edges of a residue are three steps with
  // nothing padded (the 16x16x16 form this replaces runs at half the matrix pipe's rate: 72 instead of 18 x 2 issue
  // slots per stage and wave), and a 32 x 64 block reads 96 plane rows per step where a 16 x 128 block read 144.
  // Lane (r32, q) holds row / column r32 and edges 8 q .. 8 q + 7 of the step in BOTH operands (any assignment of
  // register elements to k that is the same on both sides is a valid one).
  const int r32 = lane & 31, q32 = lane >> 5;
  const int arow = 32 * (wave >> 1) + r32, brow = 64 * (wave & 1) + r32;
  auto contract = [&](int st, f32x16 (&acc)[2]) {
    auto rd8 = [&](const _Float16* p) {  // eight consecutive halves at an 8-byte aligned address
      const h4 lo4 = *reinterpret_cast<const h4*>(p), hi4 = *reinterpret_cast<const h4*>(p + 4);
      return h8{lo4[0], lo4[1], lo4[2], lo4[3], hi4[0], hi4[1], hi4[2], hi4[3]};
    };
#pragma unroll
    for (int c = 0; c < KT; ++c) {
      const int e0 = 16 * c + 8 * q32;
      const h8 ah = rd8(&sA[st][0][arow][e0]), al = rd8(&sA[st][1][arow][e0]);
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const h8 bh = rd8(&sB[st][0][brow + 32 * b][e0]), bl = rd8(&sB[st][1][brow + 32 * b][e0]);
        acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc[b], 0, 0, 0);
        acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc[b], 0, 0, 0);
        acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[b], 0, 0, 0);
      }
      FENCE();
    }
  };
  // one column tile of this wave's gradient tile (already scaled) -> its halves of the next GEMM's B fragments and its
  // A-plane rows
  auto publish_grad = [&](int st, int c, const f32x4 x) {
    h4 hh, ll;
    split4(x, hh, ll);
    _Float16* bh = reinterpret_cast<_Float16*>(&Bf[st][bf_kk][c][0][lane]) + bf_half;
    _Float16* bl = reinterpret_cast<_Float16*>(&Bf[st][bf_kk][c][1][lane]) + bf_half;
    *reinterpret_cast<h4*>(bh) = hh;
    *reinterpret_cast<h4*>(bl) = ll;
    split4(quad_transpose(x), hh, ll);
    const int ecol = 16 * c + 4 * (col >> 2);
    *reinterpret_cast<h4*>(&sA[st][0][frow][ecol]) = hh;
    *reinterpret_cast<h4*>(&sA[st][1][frow][ecol]) = ll;
  };
  // one column tile of a contraction's second operand -> its B-plane rows
  auto publish_act = [&](int st, int c, const f32x4 v) {
    h4 hh, ll;
    split4(quad_transpose(v), hh, ll);
    const int ecol = 16 * c + 4 * (col >> 2);
    *reinterpret_cast<h4*>(&sB[st][0][frow][ecol]) = hh;
    *reinterpret_cast<h4*>(&sB[st][1][frow][ecol]) = ll;
  };
  // y = sum_k W^T[tile of this wave][k] X[k][column tile c], X from the B fragments of stage st; the next k-step's
  // fragment is read ahead of this one's MFMAs
  auto gemm_tile = [&](int st, int c, const h8 (&wh)[NKK], const h8 (&wl)[NKK]) {
    f32x4 y = f32x4{0.f, 0.f, 0.f, 0.f};
    BFrag b[2];
    b[0].hi = Bf[st][0][c][0][lane];
    b[0].lo = Bf[st][0][c][1][lane];
#pragma unroll
    for (int kk = 0; kk < NKK; ++kk) {
      if (kk + 1 < NKK) {
        b[(kk + 1) & 1].hi = Bf[st][kk + 1][c][0][lane];
        b[(kk + 1) & 1].lo = Bf[st][kk + 1][c][1][lane];
      }
      y = mfma_split(wh[kk], wl[kk], b[kk & 1], y);
      FENCE();
    }
    return y;
  };
  // addresses = a uniform residue base (scalar registers) + a 32-bit lane offset that never changes
  const unsigned tile_off = (unsigned)(wave * KT) * 64u + (unsigned)lane;  // f32x4 elements into a residue's fragments
  auto tile_of = [&](const f32x4* base, int node) {
#ifdef MPNN_ABL_FUSED_SAMELOAD  // timing-only ablation build: never shipped
    node = blockIdx.x;
#endif
    return base + (size_t)__builtin_amdgcn_readfirstlane(node) * (MPNN_NT * KT * 64);
  };
  auto load_tile = [&](f32x4 (&dst)[KT], const f32x4* base, int node) {
    const f32x4* p = tile_of(base, node);
#pragma unroll
    for (int c = 0; c < KT; ++c) dst[c] = __builtin_nontemporal_load(&p[tile_off + (unsigned)c * 64u]);
  };
  f32x4 x2[KT], x1[KT], xe[KT], pe[KT], gs;
  float s[KT], w[KT];
  // branch-free mask reads: out-of-range slots re-read slot K - 1 and are zeroed by the select
  const float* em_base = edge_mask ? edge_mask : mask;  // (any readable address when there is no edge mask)
  const bool has_em = edge_mask != nullptr;
  auto load_masks = [&](int node_) {
    const int node = __builtin_amdgcn_readfirstlane(node_);
    const int b = node / L;
    const float mi = mask[node];
    const float* em_row = em_base + (has_em ? (size_t)node * K : (size_t)0);
    const int* e_row = E_idx + (size_t)node * Kp;
    const float* m_row = mask + (size_t)b * L;
#pragma unroll
    for (int c = 0; c < KT; ++c) {
      const int e = 16 * c + col;
      const bool in = e < K;
      const unsigned ec = (unsigned)(in ? e : K - 1);
      const float em = has_em ? em_row[ec] : 1.0f;
      if (DEC) {
        s[c] = mi * em;
        w[c] = in ? (w_valid ? em : 1.0f) : 0.0f;
      } else {
        s[c] = 1.0f;
        const float mj = m_row[(unsigned)e_row[(unsigned)e]];
        w[c] = in ? mi * mj * em : 0.0f;
      }
    }
  };
  auto load_gs = [&](int node) {
    const float* p = dgsum + (size_t)__builtin_amdgcn_readfirstlane(node) * MPNN_H;
    gs = *reinterpret_cast<const f32x4*>(&p[(unsigned)(16 * wave + 4 * g)]);
  };
  const unsigned row_off = (unsigned)col * MPNN_H + (unsigned)(16 * wave + 4 * g);  // into a residue's da1 rows

  // Operand tiles are requested about two phases ahead of their use and are NOT kept live all the way round the loop
  // (a ring one residue deep for each of the four tensors is 48 registers at every point; this is ~36):
  //   a2, dgsum, a1, masks of the next residue: top of phase 3    (used in its phases 1 / 1+2)
  //   e:                                        top of phase 1    (used in phase 2; requested with the others it
  //                                             costs 14-19 spilled registers and 0.05 ms per step)
  //   dE (accumulate):                          top of phase 2    (used in phase 3)
  int node = blockIdx.x;
  if (node < N) {
    load_masks(node);
    load_tile(x2, a2, node);
    load_gs(node);
    load_tile(x1, a1, node);
  }
  for (; node < N; node += G) {
    const int nxt = node + G;
    const bool more = nxt < N;
    load_tile(xe, hE, node);
    FENCE();
    // ---- phase 1: da2 of this wave's feature tile; its B fragments and planes, the planes of gelu(a1)
    f32x4 bacc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < KT; ++c) {
      f32x4 d2 = gelu_grad4(x2[c]) * (gs * w[c]);
      bacc += d2;
      publish_grad(0, c, d2 * sc1);
      publish_act(0, c, gelu_value_grad4(x1[c]));  // x1 <- gelu'(a1)
      FENCE();
    }
    FB_BARRIER();  // A: stage-0 fragments and planes complete (and stage 1 of the previous residue consumed)
    if (accumulate) load_tile(pe, dE, node);
    FENCE();
    // ---- phase 2: dg1 = W2^T da2; da1, dP, rows; dW2 += da2 (x) gelu(a1)
    f32x4 ps = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < KT; ++c) {
      f32x4 y = gemm_tile(0, c, w2h, w2l);
      y *= x1[c] * un1;
      ps += y;
      if (DEC) y *= s[c];
      {
        float* rb = da1_nat + (size_t)__builtin_amdgcn_readfirstlane(node) * (Kp * MPNN_H);
        __builtin_nontemporal_store(y, reinterpret_cast<f32x4*>(&rb[row_off + (unsigned)(16 * c) * MPNN_H]));
      }
      publish_grad(1, c, y * sc2);
      publish_act(1, c, xe[c]);
      FENCE();
    }
    ps = column_sum4(ps);
    bacc = column_sum4(bacc);
    if (col == 0) {
      float* pb = dP + (size_t)__builtin_amdgcn_readfirstlane(node) * MPNN_H;
      *reinterpret_cast<f32x4*>(&pb[(unsigned)(16 * wave + 4 * g)]) = ps;
      *reinterpret_cast<f32x4*>(&sacc[16 * wave + 4 * g]) += bacc;
    }
    FENCE();
    h8 w1h[NKK], w1l[NKK];
    {
      auto Wp = GLOBAL(W1w);
      LAUNDER(Wp);
#pragma unroll
      for (int kk = 0; kk < NKK; ++kk) {
        w1h[kk] = Wp[(kk * 2 + 0) * 64];
        w1l[kk] = Wp[(kk * 2 + 1) * 64];
      }
    }
    FENCE();
    contract(0, accW2);
    FB_BARRIER();  // B: stage-1 fragments and planes complete (and stage 0 consumed)
    if (more) {
      load_masks(nxt);
      load_tile(x2, a2, nxt);
      load_gs(nxt);
      load_tile(x1, a1, nxt);
    }
    FENCE();
    // ---- phase 3: dE (+)= W1b^T rows; dW1b += rows (x) e
    {
      f32x4* dst = const_cast<f32x4*>(tile_of(dE, node));
#pragma unroll
      for (int c = 0; c < KT; ++c) {
        f32x4 v = gemm_tile(1, c, w1h, w1l) * un2;
        if (accumulate) v += pe[c];
        vmax = fmaxf(fmaxf(vmax, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
        __builtin_nontemporal_store(v, &dst[tile_off + (unsigned)c * 64u]);
        FENCE();
      }
    }
    contract(1, accW1);
  }
  if (amax_out) {  // max |dE| as written: the operand-scale bound of the kernel that takes dE as its dy
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, o));
    if (lane == 0) atomicMax(amax_out, (unsigned)__float_as_int(vmax));  // a maximum: order-independent
  }
  // this workgroup's partial sums: [dW2 | db2] and [dW1b | -]
  float* P = part + (size_t)blockIdx.x * 2 * WG_PART;
  // (a 32 x 32 MFMA tile: register v of lane (r32, q) is row 8 (v / 4) + 4 q + v % 4, column r32)
#pragma unroll
  for (int b = 0; b < 2; ++b)
#pragma unroll
    for (int v = 0; v < 16; ++v) {
      const int o = (32 * (wave >> 1) + 8 * (v >> 2) + 4 * q32 + (v & 3)) * MPNN_H + 64 * (wave & 1) + 32 * b + r32;
      P[o] = accW2[b][v] * un1;
      P[WG_PART + o] = accW1[b][v] * un2;
    }
  __syncthreads();
  if (threadIdx.x < MPNN_H) {
    P[MPNN_H * MPNN_H + threadIdx.x] = sacc[threadIdx.x];
    P[WG_PART + MPNN_H * MPNN_H + threadIdx.x] = 0.f;
  }
}

// ---------------------------------------------------------------------------
// Encoder edge-update backward with its THREE weight-gradient contractions fused in (K <= 48): what
// mpnn_enc_edge_update_bwd + 3 x mpnn_edge_wgrad do, in the workgroup-per-residue form of k_message_bwd_fused - dz and
// da2 never exist in memory, a2 / a1 / hE / the da1 rows are read once (7 instead of 16 edge tensors of traffic).
//   stage 0: dz = LayerNorm'(dy; z), dzd = dropout(dz)      dW13 += dzd (x) gelu(a2)   dg2 = W13^T dzd
//   stage 1: da2 = dg2 gelu'(a2)                            dW12 += da2 (x) gelu(a1)   dg1 = W12^T da2
//   stage 2: da1 = dg1 gelu'(a1) (rows for dQ', dP')        dW11b += da1 (x) e         dE = dz + W11b^T da1
// A wave owns 16 of the 128 features, so the LayerNorm's per-edge sums over the features cross the waves: ONE exchange
// of four partial moments per edge (sum z, centred sum of squares, sum dy g, centred sum dy g z; combined about the
// common mean) through LDS gives mean, rstd, mean(dxhat) and mean(dxhat xhat) together - four barriers per residue (exchange + three stages).  Stage buffers: stage 2 re-uses stage
// 0's (consumed before barrier B).  All three transposed weights stream from L2 (16 rows per wave and GEMM); the three
// 16 x 128 accumulator blocks, the bias and LayerNorm-parameter sums take 112 registers.
// Launch-wide power-of-two operand scales from BOUNDS (k_fused_scales3): |dz| <= 13.3 rstd_max max|g| max|dy| (the forward
// kernel records rstd_max, the producers of dy record max|dy|), then 1.13 ||W^T||_1-row per GEMM.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(128)
k_fused_scales3(const unsigned* __restrict__ dy_amax, const unsigned* __restrict__ rstd_max,
                const float* __restrict__ ln_w, const h8* __restrict__ W13T, const h8* __restrict__ W12T, float inv_keep,
                unsigned* __restrict__ scales) {
  __shared__ float red[3][2];
  const int to = threadIdx.x >> 4, col = threadIdx.x & 15;
  float s13 = 0.f, s12 = 0.f;
  for (int kk = 0; kk < NKK; ++kk)
    for (int g = 0; g < 4; ++g) {
      const size_t i = ((size_t)(to * NKK + kk) * 2) * 64 + 16 * g + col;
      const h8 h3 = W13T[i], l3 = W13T[i + 64], h2 = W12T[i], l2 = W12T[i + 64];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        s13 += fabsf((float)h3[q] + (float)l3[q]);
        s12 += fabsf((float)h2[q] + (float)l2[q]);
      }
    }
  float gm = fabsf(ln_w[threadIdx.x]);
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    s13 = fmaxf(s13, __shfl_xor(s13, o));
    s12 = fmaxf(s12, __shfl_xor(s12, o));
    gm = fmaxf(gm, __shfl_xor(gm, o));
  }
  if ((threadIdx.x & 63) == 0) {
    red[0][threadIdx.x >> 6] = s13;
    red[1][threadIdx.x >> 6] = s12;
    red[2][threadIdx.x >> 6] = gm;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const float S13 = fmaxf(red[0][0], red[0][1]), S12 = fmaxf(red[1][0], red[1][1]), G = fmaxf(red[2][0], red[2][1]);
    const float m0 = 13.3f * __int_as_float((int)rstd_max[0]) * G * __int_as_float((int)dy_amax[0]) * inv_keep;
    const float m1 = 1.1293f * S13 * m0;
    const float m2 = 1.1293f * S12 * m1;
    scales[0] = (unsigned)__float_as_int(m0);
    scales[1] = (unsigned)__float_as_int(m1);
    scales[2] = (unsigned)__float_as_int(m2);
  }
}

constexpr int EU_PART = 2 * WG_PART + 2 * MPNN_H;  // floats of one workgroup's partial sums

template <int KT>
__global__ void __launch_bounds__(FB_WAVES * 64)
k_edge_update_bwd_fused(const f32x4* __restrict__ dy, const f32x4* __restrict__ z, const f32x4* __restrict__ a2,
                        const f32x4* __restrict__ a1, const float* __restrict__ ln_w,
                        const h8* __restrict__ W13T, const h8* __restrict__ W12T, const h8* __restrict__ W11bT, int N,
                        float* __restrict__ da1_nat, float* __restrict__ dP, f32x4* __restrict__ dE,
                        const unsigned* __restrict__ scales, float* __restrict__ part, unsigned* __restrict__ amax_out,
                        unsigned* __restrict__ amax_rows,
                        unsigned drop_thresh, unsigned drop_seed, float inv_keep) {
  __shared__ h8 Bf[2][NKK][KT][2][64];
  __shared__ _Float16 sA[2][MPNN_H][FB_LDK];  // a wave reads only the rows it wrote: one buffer, publish -> contract in order
  __shared__ _Float16 sB[2][2][MPNN_H][FB_LDK];
  __shared__ f32x4 red[FB_WAVES][KT][16];
  __shared__ f32x4 park[FB_WAVES][KT][64];    // dz of the residue (the residual term of dE), out of the registers
  __shared__ float sacc[4][MPNN_H];           // db13 | db12 | dgamma | dbeta: one writer lane per word, in residue order
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane >> 4, col = lane & 15;
  constexpr int Kp = KT * 16;
  float sc[3], un[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    sc[i] = pow2_scale_from_bits(scales[i]);
    un[i] = pow2_inverse(sc[i]);
  }
  f32x4 acc13[MPNN_NT], acc12[MPNN_NT];
#pragma unroll
  for (int j = 0; j < MPNN_NT; ++j) acc13[j] = acc12[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (threadIdx.x < MPNN_H) sacc[0][threadIdx.x] = sacc[1][threadIdx.x] = sacc[2][threadIdx.x] = sacc[3][threadIdx.x] = 0.f;
  // (first touched after the first barrier of the loop)
  auto sacc_add = [&](int q, f32x4 v) {  // per-feature sums over the residue's edges -> the workgroup's running sums
    v = column_sum4(v);
    if (col == 0) {
      f32x4* p = reinterpret_cast<f32x4*>(&sacc[q][16 * wave + 4 * g]);
      *p += v;
    }
  };
  float vmax = 0.f, rmax = 0.f;
  const f32x4 gm = *reinterpret_cast<const f32x4*>(ln_w + 16 * wave + 4 * g);
  const int frow = 16 * wave + 4 * g + (col & 3);
  const int bf_kk = wave >> 1, bf_half = (wave & 1) * 4;
  const int G = gridDim.x;

  // acc[j] += sum_e A[own 16 rows][e] B[tile j][e] on v_mfma_f32_16x16x32_f16: lane (col, g) holds edges 4 g .. 4 g + 3
  // of TWO edge tiles in either operand; a last odd tile is paired with zeros.  (The 16x16x16 form takes the same 16
  // cycles for half the edges, so the zero half costs nothing, and 48 edges are two instruction slots per product
  // instead of three.  Mixing the two forms on one accumulator gave wrong sums on gfx950 - see DESIGN 0b.)  The message
  // kernel's 32 x 32 tiles save more, but read plane rows another wave wrote: this kernel's A planes are
  // single-buffered on the promise that a wave reads only its own.
  auto contract = [&](int st, f32x4 (&acc)[MPNN_NT]) {
    constexpr int NP = (KT + 1) / 2;  // pairs of edge tiles
    auto ld4 = [&](const _Float16* p) { return *reinterpret_cast<const h4*>(p); };
    auto rd8 = [&](const _Float16* r, int p) {  // edges 32 p + 4 g .. and 32 p + 16 + 4 g ..
      const h4 x = ld4(r + 32 * p);
      const h4 y = 2 * p + 1 < KT ? ld4(r + 32 * p + 16) : h4{};
      return h8{x[0], x[1], x[2], x[3], y[0], y[1], y[2], y[3]};
    };
    h8 ah[NP], al[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      ah[p] = rd8(&sA[0][16 * wave + col][4 * g], p);
      al[p] = rd8(&sA[1][16 * wave + col][4 * g], p);
    }
    // B operands one step (an output tile x a pair of edge tiles: three MFMAs) ahead of their use
    h8 bh[2], bl[2];
    auto rd = [&](int s) {
      bh[s & 1] = rd8(&sB[st][0][16 * (s / NP) + col][4 * g], s % NP);
      bl[s & 1] = rd8(&sB[st][1][16 * (s / NP) + col][4 * g], s % NP);
    };
    rd(0);
#pragma unroll
    for (int s = 0; s < MPNN_NT * NP; ++s) {
      const int j = s / NP, p = s % NP;
      if (s + 1 < MPNN_NT * NP) rd(s + 1);
      acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[p], bh[s & 1], acc[j], 0, 0, 0);
      acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[p], bl[s & 1], acc[j], 0, 0, 0);
      acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[p], bh[s & 1], acc[j], 0, 0, 0);
      FENCE();
    }
  };
  auto publish_grad = [&](int st, int c, const f32x4 x) {
    h4 hh, ll;
    split4(x, hh, ll);
    _Float16* bh = reinterpret_cast<_Float16*>(&Bf[st][bf_kk][c][0][lane]) + bf_half;
    _Float16* bl = reinterpret_cast<_Float16*>(&Bf[st][bf_kk][c][1][lane]) + bf_half;
    *reinterpret_cast<h4*>(bh) = hh;
    *reinterpret_cast<h4*>(bl) = ll;
    split4(quad_transpose(x), hh, ll);
    const int ecol = 16 * c + 4 * (col >> 2);
    *reinterpret_cast<h4*>(&sA[0][frow][ecol]) = hh;
    *reinterpret_cast<h4*>(&sA[1][frow][ecol]) = ll;
  };
  auto publish_bf = [&](int st, int c, const f32x4 x) {  // B fragments only (no contraction follows in this kernel)
    h4 hh, ll;
    split4(x, hh, ll);
    _Float16* bh = reinterpret_cast<_Float16*>(&Bf[st][bf_kk][c][0][lane]) + bf_half;
    _Float16* bl = reinterpret_cast<_Float16*>(&Bf[st][bf_kk][c][1][lane]) + bf_half;
    *reinterpret_cast<h4*>(bh) = hh;
    *reinterpret_cast<h4*>(bl) = ll;
  };
  auto publish_act = [&](int st, int c, const f32x4 v) {
    h4 hh, ll;
    split4(quad_transpose(v), hh, ll);
    const int ecol = 16 * c + 4 * (col >> 2);
    *reinterpret_cast<h4*>(&sB[st][0][frow][ecol]) = hh;
    *reinterpret_cast<h4*>(&sB[st][1][frow][ecol]) = ll;
  };
  auto gemm_tile = [&](int st, int c, const h8 (&wh)[NKK], const h8 (&wl)[NKK]) {
    f32x4 y = f32x4{0.f, 0.f, 0.f, 0.f};
    BFrag b[2];
    b[0].hi = Bf[st][0][c][0][lane];
    b[0].lo = Bf[st][0][c][1][lane];
#pragma unroll
    for (int kk = 0; kk < NKK; ++kk) {
      if (kk + 1 < NKK) {
        b[(kk + 1) & 1].hi = Bf[st][kk + 1][c][0][lane];
        b[(kk + 1) & 1].lo = Bf[st][kk + 1][c][1][lane];
      }
      y = mfma_split(wh[kk], wl[kk], b[kk & 1], y);
      FENCE();
    }
    return y;
  };
  // this wave's 16 rows of a transposed weight, from L2 (requested one phase ahead of their GEMM)
  auto load_w = [&](const h8* WT, h8 (&wh)[NKK], h8 (&wl)[NKK]) {
    const h8* base = WT + (size_t)wave * NKK * 2 * 64 + lane;
    auto Wp = GLOBAL(base);
    LAUNDER(Wp);
#pragma unroll
    for (int kk = 0; kk < NKK; ++kk) {
      wh[kk] = Wp[(kk * 2 + 0) * 64];
      wl[kk] = Wp[(kk * 2 + 1) * 64];
    }
  };
  const unsigned tile_off = (unsigned)(wave * KT) * 64u + (unsigned)lane;
  auto tile_of = [&](const f32x4* base, int node) {
#ifdef MPNN_ABL_FUSED_SAMELOAD  // timing-only ablation build: never shipped
    node = blockIdx.x;
#endif
    return base + (size_t)__builtin_amdgcn_readfirstlane(node) * (MPNN_NT * KT * 64);
  };
  auto load_tile = [&](f32x4 (&dst)[KT], const f32x4* base, int node) {
    const f32x4* p = tile_of(base, node);
#pragma unroll
    for (int c = 0; c < KT; ++c) dst[c] = __builtin_nontemporal_load(&p[tile_off + (unsigned)c * 64u]);
  };
  const unsigned row_off = (unsigned)col * MPNN_H + (unsigned)(16 * wave + 4 * g);

  f32x4 zt[KT], dt[KT], t2[KT], t1[KT];
  int node = blockIdx.x;
  if (node < N) {
    load_tile(zt, z, node);
    load_tile(dt, dy, node);
    load_tile(t2, a2, node);
  }
  for (; node < N; node += G) {
    const int nxt = node + G;
    const bool more = nxt < N;
    // ---- LayerNorm exchange: per edge, this wave's 16 features give  sum z,  sum (z - mean_w)^2,  sum dy g,
    //      sum dy g (z - mean_w)  with mean_w the wave's own mean: the waves' partial moments are then combined about the
    //      common mean (Chan et al.) - a plain sum of squares loses the variance when |mean| >> sigma
#pragma unroll
    for (int c = 0; c < KT; ++c) {
      const f32x4 dg = dt[c] * gm;
      f32x4 sm;
      sm.x = quarter_sum((zt[c].x + zt[c].y) + (zt[c].z + zt[c].w));
      const f32x4 zc = zt[c] - sm.x * (1.0f / 16.0f);
      sm.y = quarter_sum((zc.x * zc.x + zc.y * zc.y) + (zc.z * zc.z + zc.w * zc.w));
      sm.z = quarter_sum((dg.x + dg.y) + (dg.z + dg.w));
      sm.w = quarter_sum((dg.x * zc.x + dg.y * zc.y) + (dg.z * zc.z + dg.w * zc.w));
      if (g == 0) red[wave][c][col] = sm;
    }
    load_tile(t1, a1, node);
    FENCE();
    FB_BARRIER();  // L
    // ---- stage 0: dz, its dropout, fragments and planes; planes of gelu(a2); t2 <- gelu'(a2)
    f32x4 bacc13 = f32x4{0.f, 0.f, 0.f, 0.f}, lg = bacc13, lb = bacc13;
#pragma unroll
    for (int c = 0; c < KT; ++c) {
      // lane quarter g combines the partial moments of waves 2g and 2g + 1, the quarters meet by two shuffles
      const f32x4 pa = red[2 * g][c][col], pb = red[2 * g + 1][c][col];
      const float mean = quarter_sum(pa.x + pb.x) * (1.0f / MPNN_H);
      const float da = pa.x * (1.0f / 16.0f) - mean, db = pb.x * (1.0f / 16.0f) - mean;
      const float var = quarter_sum((pa.y + pb.y) + 16.0f * (da * da + db * db)) * (1.0f / MPNN_H);
      const float rstd = rsqrtf(var + MPNN_LN_EPS);
      const float m1 = quarter_sum(pa.z + pb.z) * (1.0f / MPNN_H);
      const float m2 = rstd * quarter_sum((pa.w + pb.w) + (da * pa.z + db * pb.z)) * (1.0f / MPNN_H);
      const f32x4 xh = (zt[c] - mean) * rstd;
      lg += dt[c] * xh;
      lb += dt[c];
      const f32x4 dz = (dt[c] * gm - m1 - xh * m2) * rstd;
      park[wave][c][lane] = dz;
      f32x4 dzd = dz;
      if (drop_thresh)
        dzd = dropout4(dz, ((unsigned)__builtin_amdgcn_readfirstlane(node) * (unsigned)(MPNN_NT * KT * 64) + tile_off +
                            (unsigned)c * 64u) * 4u,
                       drop_seed, drop_thresh, inv_keep);
      bacc13 += dzd;
      publish_grad(0, c, dzd * sc[0]);
      publish_act(0, c, gelu_value_grad4(t2[c]));  // t2 <- gelu'(a2)
      FENCE();
    }
    sacc_add(0, bacc13);
    sacc_add(2, lg);
    sacc_add(3, lb);
    FENCE();
    if (more) {  // z / dy of the next residue (dead from here on): three phases ahead of the exchange that reads them
      load_tile(zt, z, nxt);
      load_tile(dt, dy, nxt);
    }
    FENCE();
    h8 wh[NKK], wl[NKK];
    load_w(W13T, wh, wl);
    FENCE();
    FB_BARRIER();  // A
    // ---- stage 1: dW13 += dzd (x) gelu(a2) first (the A planes are single-buffered), then da2 = (W13^T dzd) gelu'(a2),
    //      its fragments and planes, the planes of gelu(a1); t1 <- gelu'(a1)
    contract(0, acc13);
    f32x4 bacc12 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < KT; ++c) {
      f32x4 y = gemm_tile(0, c, wh, wl);
      y *= t2[c] * un[0];
      bacc12 += y;
      publish_grad(1, c, y * sc[1]);
      publish_act(1, c, gelu_value_grad4(t1[c]));  // t1 <- gelu'(a1)
      FENCE();
    }
    sacc_add(1, bacc12);
    FENCE();
    if (more) load_tile(t2, a2, nxt);  // (dead since the loop above)
    FENCE();
    load_w(W12T, wh, wl);
    FENCE();
    FB_BARRIER();  // B
    // ---- stage 2: dW12 += da2 (x) gelu(a1); da1 = (W12^T da2) gelu'(a1): rows, dP, fragments (stage 0's buffer)
    contract(1, acc12);
    f32x4 ps = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < KT; ++c) {
      f32x4 y = gemm_tile(1, c, wh, wl);
      y *= t1[c] * un[1];
      ps += y;
      rmax = fmaxf(fmaxf(rmax, fmaxf(fabsf(y.x), fabsf(y.y))), fmaxf(fabsf(y.z), fabsf(y.w)));
      {
        float* rb = da1_nat + (size_t)__builtin_amdgcn_readfirstlane(node) * (Kp * MPNN_H);
        __builtin_nontemporal_store(y, reinterpret_cast<f32x4*>(&rb[row_off + (unsigned)(16 * c) * MPNN_H]));
      }
      publish_bf(0, c, y * sc[2]);
      FENCE();
    }
    ps = column_sum4(ps);
    if (col == 0) {
      float* pb = dP + (size_t)__builtin_amdgcn_readfirstlane(node) * MPNN_H;
      *reinterpret_cast<f32x4*>(&pb[(unsigned)(16 * wave + 4 * g)]) = ps;
    }
    FENCE();
    load_w(W11bT, wh, wl);
    FENCE();
    FB_BARRIER();  // C
    // ---- stage 3: dE = dz + W11b^T da1; dW11b += da1 (x) e.  The next residue's z / dy / a2 tiles are requested
    //      BEFORE its GEMM (their registers are dead since stages 0 / 1): requested after it, the LayerNorm exchange at
    //      the top of the loop waited out a full memory round trip per residue
    {
      f32x4* dst = const_cast<f32x4*>(tile_of(dE, node));
#pragma unroll
      for (int c = 0; c < KT; ++c) {
        const f32x4 v = gemm_tile(0, c, wh, wl) * un[2] + park[wave][c][lane];
        vmax = fmaxf(fmaxf(vmax, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
        __builtin_nontemporal_store(v, &dst[tile_off + (unsigned)c * 64u]);
        FENCE();
      }
    }
    FENCE();
  }
  // this workgroup's partial sums: [dW13 | db13] [dW12 | db12] [dgamma | dbeta]
  float* P = part + (size_t)blockIdx.x * EU_PART;
#pragma unroll
  for (int j = 0; j < MPNN_NT; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int o = (16 * wave + 4 * g + r) * MPNN_H + 16 * j + col;
      P[o] = acc13[j][r] * un[0];
      P[WG_PART + o] = acc12[j][r] * un[1];
    }
  __syncthreads();
  if (threadIdx.x < MPNN_H) {
    P[MPNN_H * MPNN_H + threadIdx.x] = sacc[0][threadIdx.x];
    P[WG_PART + MPNN_H * MPNN_H + threadIdx.x] = sacc[1][threadIdx.x];
    P[2 * WG_PART + threadIdx.x] = sacc[2][threadIdx.x];
    P[2 * WG_PART + MPNN_H + threadIdx.x] = sacc[3][threadIdx.x];
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    vmax = fmaxf(vmax, __shfl_xor(vmax, o));
    rmax = fmaxf(rmax, __shfl_xor(rmax, o));
  }
  if (lane == 0) {  // maxima: order-independent
    if (amax_out) atomicMax(amax_out, (unsigned)__float_as_int(vmax));
    atomicMax(amax_rows, (unsigned)__float_as_int(rmax));
  }
}

// ---------------------------------------------------------------------------
// out[j] = sum of the rows listed for node j (inverse neighbour list: rows that gathered node j in forward).
// sel (optional, per list entry): 0 -> outA, 1 -> outB  (decoder: row came from the A or the encoder block)
// In-degrees are very uneven (hub residues are gathered by hundreds of neighbours), so the unit of work is a CHUNK
// of at most 64 list entries: wave w looks its node up in chunk_node (written with the lists), reads its rows as
// coalesced 512-byte lines (8 in flight) in list order and
//   - a node with ONE chunk (in-degree <= 64: nearly all at K = 48) gets its sums by plain stores;
//   - a chunk of a node with several writes its partial sums to part[w], and k_gather_finish adds a node's partials
//     in chunk order.
// Every output row is written exactly once, in a fixed summation order: no atomics, no zero fill, the same bits on
// every run (the round-2 form met 32-entry partials by fp32 atomics).
// ---------------------------------------------------------------------------
constexpr int GATHER_CHUNK = 64;

__global__ void __launch_bounds__(256)
k_gather_rows(const float* __restrict__ rows, const int* __restrict__ inv_ptr, const int* __restrict__ inv_row,
              const unsigned char* __restrict__ sel, const int* __restrict__ chunk_ptr,
              const int* __restrict__ chunk_node, int N, int n_waves, float* __restrict__ outA,
              float* __restrict__ outB, float* __restrict__ part) {
  const int lane = threadIdx.x & 63;
  const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (w >= n_waves || w >= chunk_ptr[N]) return;
  const int j = chunk_node[w];
  const int c0 = chunk_ptr[j], nch = chunk_ptr[j + 1] - c0;
  const int m0 = inv_ptr[j] + (w - c0) * GATHER_CHUNK;
  const int m1 = min(inv_ptr[j + 1], m0 + GATHER_CHUNK);
  // half a wave per row (32 lanes x 16 B = one 512-byte row), two rows per load instruction, 8 instructions in flight
  // (16 in flight - a whole chunk at once - was measured three times slower)
  const int half = lane >> 5, q = lane & 31;
  f32x4 sa = f32x4{0.f, 0.f, 0.f, 0.f}, sb = sa;
  for (int m = m0; m < m1; m += 16) {
    f32x4 v[8];
    int fl[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int mm = m + 2 * u + half;
      v[u] = f32x4{0.f, 0.f, 0.f, 0.f};
      fl[u] = 0;
      if (mm < m1) {
        v[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(rows + (size_t)inv_row[mm] * MPNN_H) + q);
        fl[u] = sel ? sel[mm] : 0;
      }
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (fl[u]) sb += v[u]; else sa += v[u];
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    sa[r] += __shfl_xor(sa[r], 32);
    sb[r] += __shfl_xor(sb[r], 32);
  }
  if (half == 0) {
    if (nch == 1) {
      *(reinterpret_cast<f32x4*>(outA + (size_t)j * MPNN_H) + q) = sa;
      if (outB) *(reinterpret_cast<f32x4*>(outB + (size_t)j * MPNN_H) + q) = sb;
    } else {
      *(reinterpret_cast<f32x4*>(part + (size_t)w * 2 * MPNN_H) + q) = sa;
      if (outB) *(reinterpret_cast<f32x4*>(part + (size_t)w * 2 * MPNN_H + MPNN_H) + q) = sb;
    }
  }
}

// nodes with several chunks: out[j] = part[c0] + part[c0 + 1] + ... in chunk order (half a wave per node)
__global__ void __launch_bounds__(256)
k_gather_finish(const int* __restrict__ chunk_ptr, int N, const float* __restrict__ part, float* __restrict__ outA,
                float* __restrict__ outB) {
  const int j = blockIdx.x * 8 + (threadIdx.x >> 5), q = threadIdx.x & 31;
  if (j >= N) return;
  const int c0 = chunk_ptr[j], c1 = chunk_ptr[j + 1];
  if (c1 - c0 <= 1) return;
  f32x4 sa = f32x4{0.f, 0.f, 0.f, 0.f}, sb = sa;
  for (int c = c0; c < c1; ++c) {
    sa += *(reinterpret_cast<const f32x4*>(part + (size_t)c * 2 * MPNN_H) + q);
    if (outB) sb += *(reinterpret_cast<const f32x4*>(part + (size_t)c * 2 * MPNN_H + MPNN_H) + q);
  }
  *(reinterpret_cast<f32x4*>(outA + (size_t)j * MPNN_H) + q) = sa;
  if (outB) *(reinterpret_cast<f32x4*>(outB + (size_t)j * MPNN_H) + q) = sb;
}

// ===========================================================================
// C ABI launchers
// ===========================================================================
extern "C" int mpnn_message_bwd(int dec, const float* dgsum, const float* a1_frag, const float* a2_frag,
                                const int32_t* E_idx32, const float* mask, const float* edge_mask,
                                const float* W2T_packed, const float* W1bT_packed, int B, int L, int K,
                                float* da2_frag, float* da1_rows, float* dP, float* dE_frag, int accumulate,
                                float* amax2, int w_valid, void* stream) {
  MPNN_CHECK_ARG(dgsum && a1_frag && a2_frag && E_idx32 && mask && W2T_packed && W1bT_packed && da2_frag &&
                     da1_rows && dP && dE_frag && amax2,
                 "mpnn_message_bwd: null pointer");
  MPNN_CHECK_ARG(B > 0 && L > 0 && K >= 1 && K <= 64, "mpnn_message_bwd: bad sizes (K <= 64)");
  MPNN_CHECK_ARG(!w_valid || (dec && edge_mask), "mpnn_message_bwd: w_valid is the decoder's form and needs edge_mask");
  const int N = B * L, KT = (K + 15) / 16;
  dim3 grid(edge_grid(N, MPNN_BWD_WAVES)), block(MPNN_BWD_WAVES * 64);
  if (dec) {
    KT_DISPATCH(KT,
                hipLaunchKernelGGL((k_message_bwd<kKT, true>), grid, block, 0, (hipStream_t)stream, dgsum,
                                   (const f32x4*)a1_frag, (const f32x4*)a2_frag, E_idx32, mask, edge_mask,
                                   (const h8*)W2T_packed, (const h8*)W1bT_packed, N, L, K, (f32x4*)da2_frag, da1_rows,
                                   dP, (f32x4*)dE_frag, accumulate, (unsigned*)amax2, w_valid));
  } else {
    KT_DISPATCH(KT,
                hipLaunchKernelGGL((k_message_bwd<kKT, false>), grid, block, 0, (hipStream_t)stream,
                                   dgsum, (const f32x4*)a1_frag, (const f32x4*)a2_frag, E_idx32, mask, edge_mask,
                                   (const h8*)W2T_packed, (const h8*)W1bT_packed, N, L, K, (f32x4*)da2_frag, da1_rows,
                                   dP, (f32x4*)dE_frag, accumulate, (unsigned*)amax2, 0));
  }
  MPNN_LAUNCH_CHECK();
}

extern "C" int mpnn_message_bwd_fused(int dec, const float* dgsum, const float* a1_frag, const float* a2_frag,
                                     const float* hE_frag, const int32_t* E_idx32, const float* mask,
                                     const float* edge_mask, const float* W2T_packed, const float* W1bT_packed, int B,
                                     int L, int K, float* da1_rows, float* dP, float* dE_frag, int accumulate,
                                     int w_valid, float* dW2, float* db2, float* dW1b, float* dE_absmax,
                                     float* workspace, int workspace_floats, void* stream) {
  MPNN_CHECK_ARG(dgsum && a1_frag && a2_frag && hE_frag && E_idx32 && mask && W2T_packed && W1bT_packed && da1_rows &&
                     dP && dE_frag && dW2 && db2 && dW1b && workspace,
                 "mpnn_message_bwd_fused: null pointer");
  MPNN_CHECK_ARG(B > 0 && L > 0 && K >= 1 && K <= 48, "mpnn_message_bwd_fused: bad sizes (K <= 48; above: mpnn_message_bwd)");
  MPNN_CHECK_ARG(!w_valid || (dec && edge_mask), "mpnn_message_bwd_fused: w_valid is the decoder's form and needs edge_mask");
  const int N = B * L, KT = (K + 15) / 16;
  int nwg = edge_grid(N, 1);
  if (nwg > (workspace_floats - 64) / (2 * WG_PART)) nwg = (workspace_floats - 64) / (2 * WG_PART);
  MPNN_CHECK_ARG(nwg >= 1, "mpnn_message_bwd_fused: workspace holds no partial (64 + 33024 floats per workgroup, one per CU)");
  MPNN_CHECK_ARG((((uintptr_t)dW2 | (uintptr_t)db2 | (uintptr_t)dW1b | (uintptr_t)workspace) & 15) == 0,
                 "mpnn_message_bwd_fused: dW2, db2, dW1b and workspace must be 16-byte aligned");
  unsigned* scales = reinterpret_cast<unsigned*>(workspace);
  workspace += 64;
  dim3 grid(nwg), block(FB_WAVES * 64);
  hipStream_t st = (hipStream_t)stream;
#define MPNN_FB_LAUNCH(kt, DECV)                                                                                     \
  hipLaunchKernelGGL((k_message_bwd_fused<kt, DECV>), grid, block, 0, st, dgsum, (const f32x4*)a1_frag,               \
                     (const f32x4*)a2_frag, (const f32x4*)hE_frag, E_idx32, mask, edge_mask, (const h8*)W2T_packed,  \
                     (const h8*)W1bT_packed, N, L, K, da1_rows, dP, (f32x4*)dE_frag, accumulate, w_valid, scales, workspace, \
                     (unsigned*)dE_absmax)
  if (hipMemsetAsync(scales, 0, 2 * sizeof(unsigned), st) != hipSuccess) {
    mpnn_set_error("mpnn_message_bwd_fused: hipMemsetAsync failed");
    return 1;
  }
  hipLaunchKernelGGL(k_fused_scales, dim3(256), dim3(256), 0, st, dgsum, (size_t)N * MPNN_H, (const h8*)W2T_packed,
                     scales);
  if (dec) {
    if (KT == 1) MPNN_FB_LAUNCH(1, true); else if (KT == 2) MPNN_FB_LAUNCH(2, true); else MPNN_FB_LAUNCH(3, true);
  } else {
    if (KT == 1) MPNN_FB_LAUNCH(1, false); else if (KT == 2) MPNN_FB_LAUNCH(2, false); else MPNN_FB_LAUNCH(3, false);
  }
#undef MPNN_FB_LAUNCH
  ReduceJobs jobs = {{workspace, workspace + WG_PART, nullptr}, {dW2, dW1b, nullptr}, {db2, nullptr, nullptr}};
  hipLaunchKernelGGL(k_wgrad_reduce_n, dim3((WG_PART / 4 + 15) / 16, 2), dim3(256), 0, st, jobs, nwg, (size_t)(2 * WG_PART));
  MPNN_LAUNCH_CHECK();
}

extern "C" int mpnn_enc_edge_update_bwd(const float* dy_frag, const float* z_frag, const float* a2_frag,
                                        const float* a1_frag, const float* ln_w, const float* W13T_packed,
                                        const float* W12T_packed, const float* W11bT_packed, int B, int L, int K,
                                        float* dz_frag, float* da2_frag, float* da1_rows, float* dP, float* dE_frag,
                                        float* dln, float* amax3, float p_drop, uint32_t drop_seed, float* workspace,
                                        int workspace_floats, void* stream) {
  MPNN_CHECK_ARG(dy_frag && z_frag && a2_frag && a1_frag && ln_w && W13T_packed && W12T_packed && W11bT_packed &&
                     dz_frag && da2_frag && da1_rows && dP && dE_frag && dln && amax3 && workspace,
                 "mpnn_enc_edge_update_bwd: null pointer");
  MPNN_CHECK_ARG(B > 0 && L > 0 && K >= 1 && K <= 64, "mpnn_enc_edge_update_bwd: bad sizes (K <= 64)");
  MPNN_CHECK_ARG(p_drop >= 0.f && p_drop < 1.f, "mpnn_enc_edge_update_bwd: dropout probability must be in [0, 1)");
  const unsigned thresh = p_drop > 0.f ? (unsigned)((double)p_drop * 4294967296.0) : 0u;
  const int N = B * L, KT = (K + 15) / 16;
  dim3 grid(edge_grid(N, MPNN_BWD_WAVES)), block(MPNN_BWD_WAVES * 64);
  MPNN_CHECK_ARG(workspace_floats >= (int)grid.x * 2 * MPNN_H,
                 "mpnn_enc_edge_update_bwd: workspace too small (256 floats per workgroup, at most one workgroup per CU)");
  KT_DISPATCH(KT,
              hipLaunchKernelGGL((k_edge_update_bwd<kKT>), grid, block, 0, (hipStream_t)stream,
                                 (const f32x4*)dy_frag, (const f32x4*)z_frag, (const f32x4*)a2_frag,
                                 (const f32x4*)a1_frag, ln_w, (const h8*)W13T_packed, (const h8*)W12T_packed,
                                 (const h8*)W11bT_packed, N, (f32x4*)dz_frag, (f32x4*)da2_frag, da1_rows, dP,
                                 (f32x4*)dE_frag, workspace, (unsigned*)amax3, thresh, (unsigned)drop_seed,
                                 1.0f / (1.0f - p_drop)));
  launch_partials_reduce(workspace, (int)grid.x, 2 * MPNN_H, 2 * MPNN_H, dln, 1, (hipStream_t)stream);
  MPNN_LAUNCH_CHECK();
}

extern "C" int mpnn_enc_edge_update_bwd_fused(const float* dy_frag, const float* z_frag, const float* a2_frag,
                                              const float* a1_frag, const float* ln_w, const float* W13T_packed,
                                              const float* W12T_packed, const float* W11bT_packed, int B, int L, int K,
                                              float* da1_rows, float* da1_absmax, float* dP, float* dE_frag, float* dln,
                                              float* dW13, float* db13, float* dW12, float* db12, const float* dy_absmax,
                                              const float* rstd_max, float* dE_absmax, float p_drop, uint32_t drop_seed,
                                              float* workspace, int workspace_floats, void* stream) {
  MPNN_CHECK_ARG(dy_frag && z_frag && a2_frag && a1_frag && ln_w && W13T_packed && W12T_packed && W11bT_packed &&
                     da1_rows && da1_absmax && dP && dE_frag && dln && dW13 && db13 && dW12 && db12 && dy_absmax &&
                     rstd_max && workspace,
                 "mpnn_enc_edge_update_bwd_fused: null pointer");
  MPNN_CHECK_ARG(B > 0 && L > 0 && K >= 1 && K <= 48,
                 "mpnn_enc_edge_update_bwd_fused: bad sizes (K <= 48; above: mpnn_enc_edge_update_bwd)");
  MPNN_CHECK_ARG(p_drop >= 0.f && p_drop < 1.f, "mpnn_enc_edge_update_bwd_fused: dropout probability must be in [0, 1)");
  const unsigned thresh = p_drop > 0.f ? (unsigned)((double)p_drop * 4294967296.0) : 0u;
  const int N = B * L, KT = (K + 15) / 16;
  int nwg = edge_grid(N, 1);
  if (nwg > (workspace_floats - 64) / EU_PART) nwg = (workspace_floats - 64) / EU_PART;
  MPNN_CHECK_ARG(nwg >= 1, "mpnn_enc_edge_update_bwd_fused: workspace holds no partial (64 + 33280 floats per workgroup, one per CU)");
  MPNN_CHECK_ARG((((uintptr_t)dW13 | (uintptr_t)db13 | (uintptr_t)dW12 | (uintptr_t)db12 | (uintptr_t)workspace) & 15) == 0,
                 "mpnn_enc_edge_update_bwd_fused: dW13, db13, dW12, db12 and workspace must be 16-byte aligned");
  unsigned* scales = reinterpret_cast<unsigned*>(workspace);
  workspace += 64;
  hipStream_t st = (hipStream_t)stream;
  const float inv_keep = 1.0f / (1.0f - p_drop);
  hipLaunchKernelGGL(k_fused_scales3, dim3(1), dim3(128), 0, st, (const unsigned*)dy_absmax, (const unsigned*)rstd_max,
                     ln_w, (const h8*)W13T_packed, (const h8*)W12T_packed, inv_keep, scales);
  dim3 grid(nwg), block(FB_WAVES * 64);
#define MPNN_EU_LAUNCH(kt)                                                                                             \
  hipLaunchKernelGGL((k_edge_update_bwd_fused<kt>), grid, block, 0, st, (const f32x4*)dy_frag, (const f32x4*)z_frag,    \
                     (const f32x4*)a2_frag, (const f32x4*)a1_frag, ln_w, (const h8*)W13T_packed,                         \
                     (const h8*)W12T_packed, (const h8*)W11bT_packed, N, da1_rows, dP, (f32x4*)dE_frag, scales,        \
                     workspace, (unsigned*)dE_absmax, (unsigned*)da1_absmax, thresh, (unsigned)drop_seed, inv_keep)
  if (KT == 1) MPNN_EU_LAUNCH(1); else if (KT == 2) MPNN_EU_LAUNCH(2); else MPNN_EU_LAUNCH(3);
#undef MPNN_EU_LAUNCH
  ReduceJobs jobs = {{workspace, workspace + WG_PART, nullptr}, {dW13, dW12, nullptr}, {db13, db12, nullptr}};
  hipLaunchKernelGGL(k_wgrad_reduce_n, dim3((WG_PART / 4 + 15) / 16, 2), dim3(256), 0, st, jobs, nwg, (size_t)EU_PART);
  launch_partials_reduce(workspace + 2 * WG_PART, nwg, EU_PART, 2 * MPNN_H, dln, 1, st);
  MPNN_LAUNCH_CHECK();
}

extern "C" int mpnn_edge_wgrad(const float* A, int a_rows, const float* A_absmax, const float* B_frag, int b_gelu,
                               int n_nodes, int K, float* dW, float* dbias, float* workspace, int workspace_floats,
                               void* stream) {
  MPNN_CHECK_ARG(A && A_absmax && B_frag && dW && workspace, "mpnn_edge_wgrad: null pointer");
  MPNN_CHECK_ARG((((uintptr_t)dW | (uintptr_t)dbias | (uintptr_t)workspace) & 15) == 0,
                 "mpnn_edge_wgrad: dW, dbias and workspace must be 16-byte aligned");
  MPNN_CHECK_ARG(n_nodes > 0 && K >= 1 && K <= 64, "mpnn_edge_wgrad: bad sizes (K <= 64)");
  const int KT = (K + 15) / 16;
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) == hipSuccess) {
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
  }
  // as many waves per CU as LDS (53 KiB per workgroup at K <= 48, 70 KiB above) and registers allow: the launch is
  // bound by how many operand bytes are in flight
  const int per_cu = KT <= 3 ? MPNN_WGRAD_WG_PER_CU : (MPNN_WGRAD_WAVES == 8 ? 2 : 1);
  int nwg = n_nodes < per_cu * cus ? n_nodes : per_cu * cus;
  if (nwg > workspace_floats / WG_PART) nwg = workspace_floats / WG_PART;
  MPNN_CHECK_ARG(nwg >= 1, "mpnn_edge_wgrad: workspace holds no partial (needs >= 16512 floats, ideally 2 per CU: mpnn_workspace_bytes)");
  dim3 grid(nwg), block(64 * MPNN_WGRAD_WAVES);
  hipStream_t st = (hipStream_t)stream;
  const f32x4* Bf = (const f32x4*)B_frag;
  const unsigned* am = (const unsigned*)A_absmax;
#define MPNN_WGRAD_LAUNCH(AN, BG)                                                                              \
  do {                                                                                                         \
    if (KT <= 3) hipLaunchKernelGGL((k_wgrad<AN, BG, 3>), grid, block, 0, st, A, Bf, am, n_nodes, KT, workspace, FeatArgs{}); \
    else hipLaunchKernelGGL((k_wgrad<AN, BG, 4>), grid, block, 0, st, A, Bf, am, n_nodes, KT, workspace, FeatArgs{});        \
  } while (0)
  if (a_rows) {
    if (b_gelu) MPNN_WGRAD_LAUNCH(true, true); else MPNN_WGRAD_LAUNCH(true, false);
  } else {
    if (b_gelu) MPNN_WGRAD_LAUNCH(false, true); else MPNN_WGRAD_LAUNCH(false, false);
  }
#undef MPNN_WGRAD_LAUNCH
  hipLaunchKernelGGL(k_wgrad_reduce, dim3((WG_PART / 4 + 15) / 16), dim3(256), 0, st, workspace, nwg, (size_t)WG_PART, dW,
                     dbias);
  MPNN_LAUNCH_CHECK();
}

// Weight gradient of the featurizer's edge embedding: dF[c][o][i] += sum_{node, e} dE0[node, e, o] F_c[node, e, i] for
// the chunks c of the raw feature vector (RBFs of the atom pairs, then the one-hot position bucket), the chunk's values
// computed inside the contraction kernel (k_wgrad, FEAT form) instead of read from memory.
extern "C" int mpnn_feature_wgrad(const float* dE0_frag, const float* dE0_absmax, const float* atoms,
                                  const int32_t* E_idx32, const int64_t* residue_idx, const int64_t* chain_labels,
                                  const float* rbf_mu, int B, int L, int K, int n_atoms, float* dF, float* dbias,
                                  float* workspace, int workspace_floats, void* stream) {
  MPNN_CHECK_ARG(dE0_frag && dE0_absmax && atoms && E_idx32 && residue_idx && chain_labels && rbf_mu && dF && workspace,
                 "mpnn_feature_wgrad: null pointer");
  MPNN_CHECK_ARG(B > 0 && L > 0 && K >= 1 && K <= 64 && (n_atoms == 5 || n_atoms == 6), "mpnn_feature_wgrad: bad sizes");
  MPNN_CHECK_ARG((((uintptr_t)dF | (uintptr_t)dbias | (uintptr_t)workspace) & 15) == 0,
                 "mpnn_feature_wgrad: dF, dbias and workspace must be 16-byte aligned");
  const int KT = (K + 15) / 16, n_nodes = B * L;
  const int nch = (n_atoms * n_atoms + 5 + 7) / 8;
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) == hipSuccess) {
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
  }
  const int per_cu = KT <= 3 ? MPNN_WGRAD_WG_PER_CU : (MPNN_WGRAD_WAVES == 8 ? 2 : 1);
  int nwg = n_nodes < per_cu * cus ? n_nodes : per_cu * cus;
  if (nwg > workspace_floats / WG_PART) nwg = workspace_floats / WG_PART;
  MPNN_CHECK_ARG(nwg >= 1, "mpnn_feature_wgrad: workspace holds no partial (mpnn_workspace_bytes(\"mpnn_edge_wgrad\"))");
  dim3 grid(nwg), block(64 * MPNN_WGRAD_WAVES);
  hipStream_t st = (hipStream_t)stream;
  const unsigned* am = (const unsigned*)dE0_absmax;
  FeatArgs fa{atoms, E_idx32, (const long long*)residue_idx, (const long long*)chain_labels, rbf_mu, L, 0};
  for (int c = 0; c < nch; ++c) {  // the launches share the workspace: each chunk's partials are reduced before the next
    fa.chunk = c;
#define MPNN_FEAT_LAUNCH(KTM, AA)                                                                                \
    hipLaunchKernelGGL((k_wgrad<false, false, KTM, AA>), grid, block, 0, st, dE0_frag, (const f32x4*)nullptr, am, \
                       n_nodes, KT, workspace, fa)
    if (n_atoms == 5) { if (KT <= 3) MPNN_FEAT_LAUNCH(3, 5); else MPNN_FEAT_LAUNCH(4, 5); }
    else              { if (KT <= 3) MPNN_FEAT_LAUNCH(3, 6); else MPNN_FEAT_LAUNCH(4, 6); }
#undef MPNN_FEAT_LAUNCH
    hipLaunchKernelGGL(k_wgrad_reduce, dim3((WG_PART / 4 + 15) / 16), dim3(256), 0, st, workspace, nwg, (size_t)WG_PART,
                       dF + (size_t)c * MPNN_H * MPNN_H, c == 0 ? dbias : nullptr);
  }
  MPNN_LAUNCH_CHECK();
}

// ---------------------------------------------------------------------------
// Inverse neighbour lists for mpnn_gather_rows: for every node j the edges (i, k) with E_idx[i][k] = j, ordered by
// (i, k) - what a stable sort of the edge list by its target gives (torch.argsort(stable=True): ~0.3 ms of merge-sort
// and index launches per step).  A residue's K neighbours are distinct, so node j holds at most one edge of each
// centre i of its protein: slot[j][i] (one byte, 0xFF = none) is a complete, collision-free map written with plain
// byte stores, and reading row j in order of i yields the list already sorted - no atomics, no sort, the same result
// on every run.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_inv_mark(const int* __restrict__ E, long long n_edges, int L, int K, int Kp, unsigned char* __restrict__ slot) {
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= n_edges) return;
  const int p = (int)(e / K), k = (int)(e - (long long)p * K);
  const int b = p / L;
  const int j = E[(size_t)p * Kp + k];  // local to the protein
  // an index outside the protein is dropped here and shows as a short list (status word of mpnn_inverse_lists)
  if ((unsigned)j < (unsigned)L) slot[((size_t)b * L + j) * L + (p - b * L)] = (unsigned char)k;
}

// FILL = false: deg[n] -> inv_ptr[n + 1];  FILL = true: the list of node n -> inv_row[inv_ptr[n] ...]
template <bool FILL>
__global__ void __launch_bounds__(256)
k_inv_rows(const unsigned char* __restrict__ slot, int N, int L, int Kp, int* __restrict__ inv_ptr,
           int* __restrict__ inv_row) {
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= N) return;
  const unsigned char* row = slot + (size_t)n * L;
  const int first = (n / L) * L;  // first node of this protein
  int base = FILL ? inv_ptr[n] : 0;
  for (int p0 = 0; p0 < L; p0 += 64) {
    const int p = p0 + lane;
    const unsigned v = p < L ? row[p] : 0xFFu;
    const unsigned long long m = __ballot(v != 0xFFu);
    if (FILL && v != 0xFFu)
      inv_row[base + __popcll(m & ((1ull << lane) - 1ull))] = (first + p) * Kp + (int)v;
    base += __popcll(m);
  }
  if (!FILL && lane == 0) inv_ptr[n + 1] = base;
}

// chunks of a node's list (mpnn_gather_rows): at least one, so that every node's output row gets written
__device__ __forceinline__ int gather_chunks(int deg) { return deg <= GATHER_CHUNK ? 1 : (deg + GATHER_CHUNK - 1) / GATHER_CHUNK; }

// one workgroup: inv_ptr[n + 1] holds deg[n] on entry and sum_{i <= n} deg[i] on exit (inv_ptr[0] = 0);
// chunk_ptr likewise for the chunk counts; status (optional) = 1 when the lists do not hold n_edges entries (an index
// out of range, or one neighbour listed twice by a residue: the slot map keeps one edge per (target, centre))
__global__ void __launch_bounds__(1024)
k_inv_scan(int N, int* __restrict__ inv_ptr, int* __restrict__ chunk_ptr, long long n_edges, int* __restrict__ status) {
  __shared__ int sd[1024], sc[1024];
  const int t = threadIdx.x;
  const int per = (N + 1023) / 1024;
  const int n0 = min(N, t * per), n1 = min(N, n0 + per);
  int d = 0, c = 0;
  for (int n = n0; n < n1; ++n) {
    const int deg = inv_ptr[n + 1];
    d += deg;
    c += gather_chunks(deg);
  }
  sd[t] = d;
  sc[t] = c;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {  // inclusive scan over the threads' sums
    const int ad = t >= off ? sd[t - off] : 0, ac = t >= off ? sc[t - off] : 0;
    __syncthreads();
    sd[t] += ad;
    sc[t] += ac;
    __syncthreads();
  }
  int rd = sd[t] - d, rc = sc[t] - c;  // exclusive
  if (t == 0) { inv_ptr[0] = 0; chunk_ptr[0] = 0; }
  for (int n = n0; n < n1; ++n) {
    const int deg = inv_ptr[n + 1];
    rd += deg;
    rc += gather_chunks(deg);
    inv_ptr[n + 1] = rd;
    chunk_ptr[n + 1] = rc;
  }
  if (status && t == 1023) status[0] = (long long)sd[1023] != n_edges ? 1 : 0;
}

// chunk_node[c] = the node chunk c belongs to
__global__ void __launch_bounds__(256)
k_inv_chunk_nodes(const int* __restrict__ chunk_ptr, int N, int* __restrict__ chunk_node) {
  const int n = blockIdx.x * 256 + threadIdx.x;
  if (n >= N) return;
  for (int c = chunk_ptr[n]; c < chunk_ptr[n + 1]; ++c) chunk_node[c] = n;
}

// sel[m] = 1 where the edge's centre does not come after its target in the decoding order (the decoder read the
// encoder block for it), in list order
__global__ void __launch_bounds__(256)
k_inv_sel(const int* __restrict__ rank, const int* __restrict__ inv_ptr, const int* __restrict__ inv_row, int N, int Kp,
          unsigned char* __restrict__ sel) {
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= N) return;
  const int rn = rank[n], m1 = inv_ptr[n + 1];
  for (int m = inv_ptr[n] + lane; m < m1; m += 64) sel[m] = rank[inv_row[m] / Kp] <= rn ? 1 : 0;
}

extern "C" int mpnn_gather_rows(const float* rows, const int32_t* inv_ptr, const int32_t* inv_row,
                                const unsigned char* sel, const int32_t* chunk_ptr, const int32_t* chunk_node,
                                int n_nodes, int max_chunks, float* outA, float* outB, float* partials, void* stream) {
  MPNN_CHECK_ARG(rows && inv_ptr && inv_row && chunk_ptr && chunk_node && outA && partials,
                 "mpnn_gather_rows: null pointer");
  MPNN_CHECK_ARG(n_nodes > 0 && max_chunks > 0 && (!sel || outB), "mpnn_gather_rows: a selector needs the second output");
  hipLaunchKernelGGL(k_gather_rows, dim3((max_chunks + 3) / 4), dim3(256), 0, (hipStream_t)stream, rows, inv_ptr, inv_row,
                     sel, chunk_ptr, chunk_node, n_nodes, max_chunks, outA, outB, partials);
  hipLaunchKernelGGL(k_gather_finish, dim3((n_nodes + 7) / 8), dim3(256), 0, (hipStream_t)stream, chunk_ptr, n_nodes,
                     partials, outA, outB);
  MPNN_LAUNCH_CHECK();
}

extern "C" int mpnn_inverse_lists(const int32_t* E_idx32, int B, int L, int K, unsigned char* slot_map, int32_t* inv_ptr,
                                  int32_t* chunk_ptr, int32_t* chunk_node, int32_t* inv_row, int32_t* status,
                                  void* stream) {
  MPNN_CHECK_ARG(E_idx32 && slot_map && inv_ptr && chunk_ptr && chunk_node && inv_row, "mpnn_inverse_lists: null pointer");
  MPNN_CHECK_ARG(B > 0 && L > 0 && K >= 1 && K <= 64 && K <= L, "mpnn_inverse_lists: bad sizes (1 <= K <= min(64, L))");
  MPNN_CHECK_ARG((long long)B * L * K < (1ll << 31) && (long long)B * L < (1ll << 31) / 64,
                 "mpnn_inverse_lists: more than 2^31 edges");
  hipStream_t st = (hipStream_t)stream;
  const int N = B * L, Kp = 16 * ((K + 15) / 16);
  const long long n_edges = (long long)N * K;
  if (hipMemsetAsync(slot_map, 0xFF, (size_t)N * L, st) != hipSuccess) {
    mpnn_set_error("mpnn_inverse_lists: hipMemsetAsync failed");
    return 1;
  }
  hipLaunchKernelGGL(k_inv_mark, dim3((unsigned)((n_edges + 255) / 256)), dim3(256), 0, st, E_idx32, n_edges, L, K, Kp,
                     slot_map);
  hipLaunchKernelGGL((k_inv_rows<false>), dim3((N + 3) / 4), dim3(256), 0, st, slot_map, N, L, Kp, inv_ptr, inv_row);
  hipLaunchKernelGGL(k_inv_scan, dim3(1), dim3(1024), 0, st, N, inv_ptr, chunk_ptr, n_edges, status);
  hipLaunchKernelGGL(k_inv_chunk_nodes, dim3((N + 255) / 256), dim3(256), 0, st, chunk_ptr, N, chunk_node);
  hipLaunchKernelGGL((k_inv_rows<true>), dim3((N + 3) / 4), dim3(256), 0, st, slot_map, N, L, Kp, inv_ptr, inv_row);
  MPNN_LAUNCH_CHECK();
}

extern "C" int mpnn_inverse_sel(const int32_t* rank, const int32_t* inv_ptr, const int32_t* inv_row, int n_nodes, int K,
                                unsigned char* sel, void* stream) {
  MPNN_CHECK_ARG(rank && inv_ptr && inv_row && sel, "mpnn_inverse_sel: null pointer");
  MPNN_CHECK_ARG(n_nodes > 0 && K >= 1 && K <= 64, "mpnn_inverse_sel: bad sizes");
  hipLaunchKernelGGL(k_inv_sel, dim3((n_nodes + 3) / 4), dim3(256), 0, (hipStream_t)stream, rank, inv_ptr, inv_row, n_nodes,
                     16 * ((K + 15) / 16), sel);
  MPNN_LAUNCH_CHECK();
}

extern "C" int mpnn_edge_features_bwd(const float* dhE_frag, const float* E0_frag, const float* ln_w,
                                      const float* WeT_packed, int B, int L, int K, float* dE0_frag, float* dln,
                                      float* amax2, float* workspace, int workspace_floats, void* stream) {
  MPNN_CHECK_ARG(dhE_frag && E0_frag && ln_w && WeT_packed && dE0_frag && dln && amax2 && workspace,
                 "mpnn_edge_features_bwd: null pointer");
  MPNN_CHECK_ARG(B > 0 && L > 0 && K >= 1 && K <= 64, "mpnn_edge_features_bwd: bad sizes (K <= 64)");
  const int N = B * L, KT = (K + 15) / 16;
  dim3 grid(edge_grid(N, MPNN_BWD_WAVES)), block(MPNN_BWD_WAVES * 64);
  MPNN_CHECK_ARG(workspace_floats >= (int)grid.x * 2 * MPNN_H,
                 "mpnn_edge_features_bwd: workspace too small (256 floats per workgroup, at most one workgroup per CU)");
  KT_DISPATCH(KT,
              hipLaunchKernelGGL((k_features_bwd<kKT>), grid, block, 0, (hipStream_t)stream,
                                 (const f32x4*)dhE_frag, (const f32x4*)E0_frag, ln_w, (const h8*)WeT_packed, N,
                                 (f32x4*)dE0_frag, workspace, (unsigned*)amax2));
  launch_partials_reduce(workspace, (int)grid.x, 2 * MPNN_H, 2 * MPNN_H, dln, 1, (hipStream_t)stream);
  MPNN_LAUNCH_CHECK();
}
