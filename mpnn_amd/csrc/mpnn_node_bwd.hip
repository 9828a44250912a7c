// Node-level backward pieces (training step).
//
// k_node_wgrad: weight / bias gradient of an nn.Linear applied to per-residue features,
//     dW[o][i] += sum_n dY[n][o] X[n][i],   db[o] += sum_n dY[n][o]          (autograd of reference :78-93, :151-155)
// The shape is "tiny output, long reduction" (128..512 x 128..512 outputs over ~10^4 residues), which library GEMMs
// serve badly (measured 91 us per call through hipBLASLt's 32x32 macro-tile at 10240 residues).  Here the reduction
// is split over workgroups; each wave keeps 2 x 8 output tiles in registers, feeds split-f16 MFMAs straight from
// global memory, and stores its partial tile to a workspace that a second small kernel adds up in a fixed order.
#include "mpnn_bwd_common.h"
#include <mutex>

constexpr int NW_PART = 128 * 128 + 128;  // floats of one workgroup partial (tile | bias sums)

// up to WG_JOBS weight gradients over the same rows in one launch (all linears of one node update): the job table
// travels by value; blockIdx.y enumerates the 128x128 output blocks of all jobs (job j owns blk0[j] .. blk0[j+1])
constexpr int WG_JOBS = 12;
struct WgradJobs {
  const float* dY[WG_JOBS];
  const float* X[WG_JOBS];
  float* dW[WG_JOBS];
  float* db[WG_JOBS];
  int ldy[WG_JOBS], out_f[WG_JOBS], ldx[WG_JOBS], in_f[WG_JOBS], ldw[WG_JOBS], ny[WG_JOBS], blk0[WG_JOBS + 1];
  int n;
};

namespace {

__device__ __forceinline__ int find_job(const WgradJobs& jb, int blk) {
  int j = 0;
  while (j + 1 < jb.n && blk >= jb.blk0[j + 1]) ++j;
  return j;
}

// The contraction on the f16 matrix pipe: 32 rows per MFMA k-step, three split products per tile (a_lo b_hi + a_hi b_lo
// + a_hi b_hi, as everywhere else in the library) = 48 v_mfma_f32_16x16x32_f16 per 32 rows and wave (round 1 used 128
// v_mfma_f32_16x16x4_f32, which run at the fp32 vector rate and made this kernel compute-bound: 46 us per launch at
// 10 240 rows).  The gradient operand is tiny (loss / 2000): each 32-row batch is scaled by an
// exact power of two taken from the wave's largest |dY| before the split, and its partial tile is scaled back when it
// is added to the fp32 accumulators.  Operand addressing: k = 8 g + i, i.e. lane (col, g) holds rows
// n + 8 g .. n + 8 g + 7 of its feature column (single dwords at row stride: latency, not bandwidth, is to be hidden).
__global__ void __launch_bounds__(256)
k_node_wgrad_split(WgradJobs jb, int N, int rows_per_wg, float* __restrict__ part) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane >> 4, col = lane & 15;
  const int job = find_job(jb, blockIdx.y);
  const int local = blockIdx.y - jb.blk0[job];
  const float* __restrict__ dY = jb.dY[job];
  const float* __restrict__ X = jb.X[job];
  const int ldy = jb.ldy[job], out_f = jb.out_f[job], ldx = jb.ldx[job], in_f = jb.in_f[job];
  const int o0 = (local % jb.ny[job]) * 128 + 32 * wave, i0 = (local / jb.ny[job]) * 128;
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
  for (int n = r0; n < r1; n += 32) {
    float a[2][8], b[8][8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int row = n + 8 * g + i;
      const bool rv = row < r1;
      const float* dy = dY + (size_t)row * ldy + o0 + col;
      const float* x = X + (size_t)row * ldx + i0 + col;
#pragma unroll
      for (int q = 0; q < 2; ++q) a[q][i] = (rv && ov[q]) ? dy[16 * q] : 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) b[j][i] = (rv && iv[j]) ? x[16 * j] : 0.f;
    }
    __builtin_amdgcn_sched_barrier(0);
    float m = 0.f;
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        bs[q] += a[q][i];
        m = fmaxf(m, fabsf(a[q][i]));
      }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    const float sc = pow2_scale_from_bits((unsigned)__float_as_int(m)), un = pow2_inverse(sc);
    BFrag af[2], bf[8];
#pragma unroll
    for (int q = 0; q < 2; ++q)
      split_pair(f32x4{a[q][0], a[q][1], a[q][2], a[q][3]} * sc, f32x4{a[q][4], a[q][5], a[q][6], a[q][7]} * sc, af[q]);
#pragma unroll
    for (int j = 0; j < 8; ++j)
      split_pair(f32x4{b[j][0], b[j][1], b[j][2], b[j][3]}, f32x4{b[j][4], b[j][5], b[j][6], b[j][7]}, bf[j]);
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        f32x4 t = f32x4{0.f, 0.f, 0.f, 0.f};
        t = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[q].lo, bf[j].hi, t, 0, 0, 0);
        t = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[q].hi, bf[j].lo, t, 0, 0, 0);
        t = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[q].hi, bf[j].hi, t, 0, 0, 0);
        acc[q][j] += t * un;
      }
  }
  float* P = part + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * NW_PART;
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

// dW block blockIdx.y += sum over row chunks of the partial tiles: 4 slices of the chunks per element, met in LDS in
// slice order (no atomics: the same bits on every run)
__global__ void __launch_bounds__(256)
k_node_wgrad_reduce(const float* __restrict__ part, int nx, WgradJobs jb) {
  __shared__ float sl[4][64];
  const int el = threadIdx.x & 63, slice = threadIdx.x >> 6;
  const int e = blockIdx.x * 64 + el;  // element of the 128x128 (+128 bias) partial
  const int blk = blockIdx.y;
  const int job = find_job(jb, blk);
  const int local = blk - jb.blk0[job];
  const int by = local % jb.ny[job], bz = local / jb.ny[job];
  const int out_f = jb.out_f[job], in_f = jb.in_f[job], ldw = jb.ldw[job];
  float* __restrict__ dW = jb.dW[job];
  float* __restrict__ db = jb.db[job];
  float s = 0.f;
  if (e < NW_PART) {
    const float* P = part + (size_t)blk * nx * NW_PART + e;
    for (int w = slice; w < nx; w += 4) s += P[(size_t)w * NW_PART];
  }
  sl[slice][el] = s;
  __syncthreads();
  if (slice != 0 || e >= NW_PART) return;
  s = ((sl[0][el] + sl[1][el]) + sl[2][el]) + sl[3][el];
  if (e < 128 * 128) {
    const int o = by * 128 + e / 128, i = bz * 128 + e % 128;
    if (o < out_f && i < in_f) dW[(size_t)o * ldw + i] += s;
  } else if (db && bz == 0) {
    const int o = by * 128 + (e - 128 * 128);
    if (o < out_f) db[o] += s;
  }
}

// ---------------------------------------------------------------------------
// Node update, training form.  Same maths and weight streaming as k_node_update (mpnn_node.hip):
//   x1pre = h + drop1((W3 gsum + cnt b3) / scale);  x1 = LN1(x1pre);  a = W_in x1 + b_in;
//   x2pre = x1 + drop2(W_out gelu(a) + b_out);  h' = mask LN2(x2pre);  PQ_p = Wproj_p h' + bproj_p
// plus: x1pre, a, x2pre are stored for the backward kernel, the two node-level dropouts of the reference's layers
// (:82, :86, :128, :135) use the counter-based keep decision of mpnn_edge_common.h (ids: node*128+feature, the second
// dropout with seed ^ 0x5bd1e995), and the projections are written block-major PQ[p][node][128].
// ---------------------------------------------------------------------------
constexpr int WSPLIT_N = MPNN_NT * NKK * 2 * 64;

__device__ __forceinline__ void stage_matrix(const h8* __restrict__ g, h8* l) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < WSPLIT_N / 256; ++i) {
    const int chunk = i * 4 + wave;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + chunk * 64 + lane),
                                     (__attribute__((address_space(3))) void*)(l + chunk * 64), 16, 0, 0);
  }
}

struct DropCfg {
  unsigned thresh, seed;
  float inv_keep;
};

__global__ void __launch_bounds__(256)
k_node_update_train(const float* __restrict__ hV_in, const float* __restrict__ gsum, const float* __restrict__ cnt,
                    float cnt_const, const float* __restrict__ mask, const h8* __restrict__ W3,
                    const float* __restrict__ b3_g, const float* __restrict__ ln1w_g, const float* __restrict__ ln1b_g,
                    const h8* __restrict__ Win, const float* __restrict__ bin_g, const h8* __restrict__ Wout,
                    const float* __restrict__ bout_g, const float* __restrict__ ln2w_g, const float* __restrict__ ln2b_g,
                    float inv_scale, const h8* __restrict__ Wproj, const float* __restrict__ bproj_g, int nproj, int N,
                    float* __restrict__ hV_out, float* __restrict__ PQ, float* __restrict__ s_x1pre,
                    float* __restrict__ s_a, float* __restrict__ s_x2pre, DropCfg dc) {
  __shared__ h8 lds[2 * WSPLIT_N];
  // bias / LayerNorm vectors in LDS, copied once (as in k_node_update: their per-stage global reads were exposed)
  constexpr int PRM_PROJ = 11 * MPNN_H;  // b3 ln1w ln1b | bin (512) | bout ln2w ln2b | bproj (<= 8 blocks)
  __shared__ __align__(16) float prm[PRM_PROJ + 8 * MPNN_H];
  for (int i = threadIdx.x; i < MPNN_H; i += blockDim.x) {
    prm[i] = b3_g[i]; prm[MPNN_H + i] = ln1w_g[i]; prm[2 * MPNN_H + i] = ln1b_g[i];
    prm[7 * MPNN_H + i] = bout_g[i]; prm[8 * MPNN_H + i] = ln2w_g[i]; prm[9 * MPNN_H + i] = ln2b_g[i];
  }
  for (int i = threadIdx.x; i < 4 * MPNN_H; i += blockDim.x) prm[3 * MPNN_H + i] = bin_g[i];
  for (int i = threadIdx.x; i < nproj * MPNN_H; i += blockDim.x) prm[PRM_PROJ + i] = bproj_g[i];
  const float* b3 = prm;
  const float* ln1w = prm + MPNN_H;
  const float* ln1b = prm + 2 * MPNN_H;
  const float* bin = prm + 3 * MPNN_H;
  const float* bout = prm + 7 * MPNN_H;
  const float* ln2w = prm + 8 * MPNN_H;
  const float* ln2b = prm + 9 * MPNN_H;
  const float* bproj = prm + PRM_PROJ;
  const int lane = threadIdx.x & 63;
  const int tile = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int g = lane >> 4, col = lane & 15;
  const int node = tile * 16 + col;
  const bool live = node < N;
  const int nd = live ? node : N - 1;
  const int nmat = 9 + nproj;
  auto matrix = [&](int k) -> const h8* {
    if (k == 0) return W3;
    if (k < 9) return (((k - 1) & 1) ? Wout : Win) + (size_t)((k - 1) >> 1) * WSPLIT_N;
    return Wproj + (size_t)(k - 9) * WSPLIT_N;
  };
  int k = 0;
  stage_matrix(matrix(0), lds);
  auto prefetch = [&]() {
    if (k + 1 < nmat) stage_matrix(matrix(k + 1), lds + ((k + 1) & 1) * WSPLIT_N);
  };
  auto retire = [&]() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    ++k;
  };
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  f32x4 h[MPNN_NT][1], gs[MPNN_NT][1], dh[MPNN_NT][1];
  const float cn = cnt ? cnt[nd] : cnt_const;
#pragma unroll
  for (int t = 0; t < MPNN_NT; ++t) {
    h[t][0] = *reinterpret_cast<const f32x4*>(hV_in + (size_t)nd * MPNN_H + 16 * t + 4 * g);
    gs[t][0] = *reinterpret_cast<const f32x4*>(gsum + (size_t)nd * MPNN_H + 16 * t + 4 * g);
    dh[t][0] = *reinterpret_cast<const f32x4*>(b3 + 16 * t + 4 * g) * cn;
  }
  BFrag xb[NKK][1];
  tile_to_bfrag<MPNN_NT, 1, false>(gs, xb);
  prefetch();
  split_gemm_stream<1, NKK, MPNN_NT, 3>(lds + (k & 1) * WSPLIT_N + lane, xb, dh);
  retire();
#pragma unroll
  for (int t = 0; t < MPNN_NT; ++t) {
    f32x4 d1 = dh[t][0] * inv_scale;
    if (dc.thresh) d1 = dropout4(d1, (unsigned)nd * 128u + 16 * t + 4 * g, dc.seed, dc.thresh, dc.inv_keep);
    h[t][0] += d1;
    if (live) *reinterpret_cast<f32x4*>(s_x1pre + (size_t)node * MPNN_H + 16 * t + 4 * g) = h[t][0];
  }
  tile_layernorm<1>(h, ln1w, ln1b, lane);
  tile_to_bfrag<MPNN_NT, 1, false>(h, xb);
  f32x4 o[MPNN_NT][1];
#pragma unroll
  for (int t = 0; t < MPNN_NT; ++t) o[t][0] = *reinterpret_cast<const f32x4*>(bout + 16 * t + 4 * g);
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    f32x4 hid[MPNN_NT][1];
#pragma unroll
    for (int t = 0; t < MPNN_NT; ++t) hid[t][0] = *reinterpret_cast<const f32x4*>(bin + 128 * q + 16 * t + 4 * g);
    prefetch();
    split_gemm_stream<1, NKK, MPNN_NT, 3>(lds + (k & 1) * WSPLIT_N + lane, xb, hid);
    retire();
    if (live) {
#pragma unroll
      for (int t = 0; t < MPNN_NT; ++t)
        *reinterpret_cast<f32x4*>(s_a + (size_t)node * 512 + 128 * q + 16 * t + 4 * g) = hid[t][0];
    }
    BFrag hb[NKK][1];
    tile_to_bfrag<MPNN_NT, 1, true>(hid, hb);
    prefetch();
    split_gemm_stream<1, NKK, MPNN_NT, 3>(lds + (k & 1) * WSPLIT_N + lane, hb, o);
    retire();
  }
#pragma unroll
  for (int t = 0; t < MPNN_NT; ++t) {
    if (dc.thresh)
      o[t][0] = dropout4(o[t][0], (unsigned)nd * 128u + 16 * t + 4 * g, dc.seed ^ 0x5bd1e995u, dc.thresh, dc.inv_keep);
    o[t][0] += h[t][0];
    if (live) *reinterpret_cast<f32x4*>(s_x2pre + (size_t)node * MPNN_H + 16 * t + 4 * g) = o[t][0];
  }
  tile_layernorm<1>(o, ln2w, ln2b, lane);
  const float mk = mask[nd];
#pragma unroll
  for (int t = 0; t < MPNN_NT; ++t) {
    h[t][0] = o[t][0] * mk;
    if (live) *reinterpret_cast<f32x4*>(hV_out + (size_t)node * MPNN_H + 16 * t + 4 * g) = h[t][0];
  }
  BFrag pb[NKK][1];
  tile_to_bfrag<MPNN_NT, 1, false>(h, pb);
  for (int p = 0; p < nproj; ++p) {
    f32x4 y[MPNN_NT][1];
#pragma unroll
    for (int t = 0; t < MPNN_NT; ++t) y[t][0] = *reinterpret_cast<const f32x4*>(bproj + p * MPNN_H + 16 * t + 4 * g);
    prefetch();
    split_gemm_stream<1, NKK, MPNN_NT, 3>(lds + (k & 1) * WSPLIT_N + lane, pb, y);
    retire();
    if (live) {
#pragma unroll
      for (int t = 0; t < MPNN_NT; ++t)
        *reinterpret_cast<f32x4*>(PQ + ((size_t)p * N + node) * MPNN_H + 16 * t + 4 * g) = y[t][0];
    }
  }
}

// ---------------------------------------------------------------------------
// Node update backward.  Streams the TRANSPOSED matrices (projections, then W_out^T / W_in^T per hidden quarter,
// then W3^T) through the same LDS ring; gradients are power-of-two scaled before the f16 split (grad_to_bfrag).
//   dy    = mask (dY + sum_p Wproj_p^T dPQ_p)            dx2pre = LN2'(dy; x2pre)        do = drop2'(dx2pre)
//   da_q  = (W_out_q^T do) gelu'(a_q)                    dx1 = dx2pre + sum_q W_in_q^T da_q
//   dx1pre = LN1'(dx1; x1pre) = d hV_in                  ddh = drop1'(dx1pre) / scale    dgsum = W3^T ddh
// Written out for the weight gradients (mpnn_node_wgrad): x1, gelu(a), da, do, ddh.  LayerNorm parameter
// gradients and db3 = sum_n cnt_n ddh_n: every wave sums its own 16 nodes in LDS, the workgroup adds its waves in wave
// order and stores vec_part[blockIdx.x][640]; the launcher's k_partials_reduce adds the workgroups in a fixed order
// into dvec[0:640] (ln1.w, ln1.b, ln2.w, ln2.b, b3).  No floating-point atomics.
// ---------------------------------------------------------------------------
struct ProjGrads {  // d PQ_p [n][128] of each projection, wherever its producer left it (NULL: no gradient, zero)
  const float* p[8];
};

__global__ void __launch_bounds__(256)
k_node_update_bwd(const float* __restrict__ dY, ProjGrads dPQ, const float* __restrict__ s_x1pre,
                  const float* __restrict__ s_a, const float* __restrict__ s_x2pre, const float* __restrict__ cnt,
                  float cnt_const, const float* __restrict__ mask, const float* __restrict__ ln1w,
                  const float* __restrict__ ln1b, const float* __restrict__ ln2w, float inv_scale,
                  const h8* __restrict__ WprojT, int nproj, const h8* __restrict__ WoutT, const h8* __restrict__ WinT,
                  const h8* __restrict__ W3T, int N, float* __restrict__ d_hV_in, float* __restrict__ d_gsum,
                  float* __restrict__ o_x1, float* __restrict__ o_hid, float* __restrict__ o_da,
                  float* __restrict__ o_do, float* __restrict__ o_ddh, float* __restrict__ vec_part, DropCfg dc) {
  __shared__ h8 lds[2 * WSPLIT_N];
  __shared__ float acc1_w[4][2 * MPNN_H], acc2_w[4][2 * MPNN_H], accb3_w[4][MPNN_H];
  float* acc1 = acc1_w[threadIdx.x >> 6];
  float* acc2 = acc2_w[threadIdx.x >> 6];
  float* accb3 = accb3_w[threadIdx.x >> 6];
  const int lane = threadIdx.x & 63;
  const int tile = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int g = lane >> 4, col = lane & 15;
  const int node = tile * 16 + col;
  const bool live = node < N;
  const int nd = live ? node : N - 1;
  const int nmat = nproj + 9;
  auto matrix = [&](int k) -> const h8* {
    if (k < nproj) return WprojT + (size_t)k * WSPLIT_N;
    k -= nproj;
    if (k < 8) return ((k & 1) ? WinT : WoutT) + (size_t)(k >> 1) * WSPLIT_N;
    return W3T;
  };
  int k = 0;
  stage_matrix(matrix(0), lds);
  auto prefetch = [&]() {
    if (k + 1 < nmat) stage_matrix(matrix(k + 1), lds + ((k + 1) & 1) * WSPLIT_N);
  };
  auto retire = [&]() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    ++k;
  };
  for (int i = threadIdx.x; i < 4 * 2 * MPNN_H; i += 256) {
    (&acc1_w[0][0])[i] = 0.f;
    (&acc2_w[0][0])[i] = 0.f;
    if (i < 4 * MPNN_H) (&accb3_w[0][0])[i] = 0.f;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  f32x4 dy[MPNN_NT][1], x[MPNN_NT][1], tmp[MPNN_NT][1];
  BFrag bf[NKK][1];
#pragma unroll
  for (int t = 0; t < MPNN_NT; ++t)
    dy[t][0] = dY ? *reinterpret_cast<const f32x4*>(dY + (size_t)nd * MPNN_H + 16 * t + 4 * g) : f32x4{0.f, 0.f, 0.f, 0.f};
  for (int p = 0; p < nproj; ++p) {
#pragma unroll
    for (int t = 0; t < MPNN_NT; ++t)
      x[t][0] = dPQ.p[p] ? *reinterpret_cast<const f32x4*>(dPQ.p[p] + (size_t)nd * MPNN_H + 16 * t + 4 * g)
                         : f32x4{0.f, 0.f, 0.f, 0.f};
    const float un = grad_to_bfrag<1>(x, bf, nullptr);
    zero_tile<1>(tmp);
    prefetch();
    split_gemm_stream<1, NKK, MPNN_NT, 3>(lds + (k & 1) * WSPLIT_N + lane, bf, tmp);
    retire();
#pragma unroll
    for (int t = 0; t < MPNN_NT; ++t) dy[t][0] += tmp[t][0] * un;
  }
  const float mk = live ? mask[nd] : 0.f;  // rows past N contribute nothing to any sum below
#pragma unroll
  for (int t = 0; t < MPNN_NT; ++t) {
    dy[t][0] *= mk;
    x[t][0] = *reinterpret_cast<const f32x4*>(s_x2pre + (size_t)nd * MPNN_H + 16 * t + 4 * g);
  }
  tile_layernorm_bwd<1>(x, dy, ln2w, acc2, lane);  // dy <- dx2pre
  f32x4 dx1[MPNN_NT][1], dob[MPNN_NT][1];
#pragma unroll
  for (int t = 0; t < MPNN_NT; ++t) {
    dx1[t][0] = dy[t][0];
    dob[t][0] = dy[t][0];
    if (dc.thresh)
      dob[t][0] = dropout4(dob[t][0], (unsigned)nd * 128u + 16 * t + 4 * g, dc.seed ^ 0x5bd1e995u, dc.thresh, dc.inv_keep);
    if (live) *reinterpret_cast<f32x4*>(o_do + (size_t)node * MPNN_H + 16 * t + 4 * g) = dob[t][0];
    x[t][0] = *reinterpret_cast<const f32x4*>(s_x1pre + (size_t)nd * MPNN_H + 16 * t + 4 * g);
  }
  tile_layernorm<1>(x, ln1w, ln1b, lane);  // x1, the FFN's input (operand of the W_in gradient)
  if (live) {
#pragma unroll
    for (int t = 0; t < MPNN_NT; ++t) *reinterpret_cast<f32x4*>(o_x1 + (size_t)node * MPNN_H + 16 * t + 4 * g) = x[t][0];
  }
  BFrag dobf[NKK][1];
  const float un_o = grad_to_bfrag<1>(dob, dobf, nullptr);
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    zero_tile<1>(tmp);
    prefetch();
    split_gemm_stream<1, NKK, MPNN_NT, 3>(lds + (k & 1) * WSPLIT_N + lane, dobf, tmp);  // W_out_q^T do
    retire();
#pragma unroll
    for (int t = 0; t < MPNN_NT; ++t) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(s_a + (size_t)nd * 512 + 128 * q + 16 * t + 4 * g);
      tmp[t][0] *= gelu_grad4(a) * un_o;
      if (live) {
        *reinterpret_cast<f32x4*>(o_da + (size_t)node * 512 + 128 * q + 16 * t + 4 * g) = tmp[t][0];
        *reinterpret_cast<f32x4*>(o_hid + (size_t)node * 512 + 128 * q + 16 * t + 4 * g) = gelu4(a);
      }
    }
    const float un = grad_to_bfrag<1>(tmp, bf, nullptr);
    zero_tile<1>(x);
    prefetch();
    split_gemm_stream<1, NKK, MPNN_NT, 3>(lds + (k & 1) * WSPLIT_N + lane, bf, x);  // W_in_q^T da_q
    retire();
#pragma unroll
    for (int t = 0; t < MPNN_NT; ++t) dx1[t][0] += x[t][0] * un;
  }
#pragma unroll
  for (int t = 0; t < MPNN_NT; ++t)
    x[t][0] = *reinterpret_cast<const f32x4*>(s_x1pre + (size_t)nd * MPNN_H + 16 * t + 4 * g);
  tile_layernorm_bwd<1>(x, dx1, ln1w, acc1, lane);  // dx1 <- dx1pre
  const float cn = cnt ? cnt[nd] : cnt_const;
#pragma unroll
  for (int t = 0; t < MPNN_NT; ++t) {
    if (live) *reinterpret_cast<f32x4*>(d_hV_in + (size_t)node * MPNN_H + 16 * t + 4 * g) = dx1[t][0];
    f32x4 dd = dx1[t][0];
    if (dc.thresh) dd = dropout4(dd, (unsigned)nd * 128u + 16 * t + 4 * g, dc.seed, dc.thresh, dc.inv_keep);
    dd *= inv_scale;
    tmp[t][0] = dd;
    if (live) *reinterpret_cast<f32x4*>(o_ddh + (size_t)node * MPNN_H + 16 * t + 4 * g) = dd;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float v = column_sum(dd[r] * cn);
      if (col == 0) accb3[16 * t + 4 * g + r] += v;
    }
  }
  const float un = grad_to_bfrag<1>(tmp, bf, nullptr);
  zero_tile<1>(x);
  prefetch();
  split_gemm_stream<1, NKK, MPNN_NT, 3>(lds + (k & 1) * WSPLIT_N + lane, bf, x);  // W3^T ddh
  retire();
  if (live) {
#pragma unroll
    for (int t = 0; t < MPNN_NT; ++t)
      *reinterpret_cast<f32x4*>(d_gsum + (size_t)node * MPNN_H + 16 * t + 4 * g) = x[t][0] * un;
  }
  __syncthreads();
  float* vp = vec_part + (size_t)blockIdx.x * 5 * MPNN_H;
  for (int i = threadIdx.x; i < 2 * MPNN_H; i += 256) {
    vp[i] = ((acc1_w[0][i] + acc1_w[1][i]) + acc1_w[2][i]) + acc1_w[3][i];
    vp[2 * MPNN_H + i] = ((acc2_w[0][i] + acc2_w[1][i]) + acc2_w[2][i]) + acc2_w[3][i];
    if (i < MPNN_H) vp[4 * MPNN_H + i] = ((accb3_w[0][i] + accb3_w[1][i]) + accb3_w[2][i]) + accb3_w[3][i];
  }
}

}  // namespace

// ---------------------------------------------------------------------------
// Batched split packing of 128x128 blocks (training re-packs every weight each step: one launch per kernel call
// instead of one per matrix).  Block b: out[o][i] = W_b[o * ld_b + i], or W_b[i * ld_b + o] when transposed; the
// descriptors travel as a by-value kernel argument.
// ---------------------------------------------------------------------------
constexpr int PACK_MAX = 16;
struct PackBatch {
  const float* W[PACK_MAX];
  int ld[PACK_MAX];
  int transposed[PACK_MAX];
};

namespace {
__global__ void __launch_bounds__(256)
k_pack_blocks(PackBatch pb, h8* __restrict__ packed) {
  const int b = blockIdx.y;
  const int idx = blockIdx.x * 256 + threadIdx.x;  // (to, kk, lane): 8 * 4 * 64
  const int lane = idx & 63, kk = (idx >> 6) & 3, to = idx >> 8;
  const int row = 16 * to + (lane & 15);
  const float* W = pb.W[b];
  const int ld = pb.ld[b], tr = pb.transposed[b];
  h8 hi, lo;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int col = 16 * (2 * kk + (j >> 2)) + 4 * (lane >> 4) + (j & 3);
    const float w = tr ? W[(size_t)col * ld + row] : W[(size_t)row * ld + col];
    const _Float16 h = (_Float16)w;
    hi[j] = h;
    lo[j] = (_Float16)(w - (float)h);
  }
  h8* dst = packed + (size_t)b * WSPLIT_N;
  dst[((to * 4 + kk) * 2 + 0) * 64 + lane] = hi;
  dst[((to * 4 + kk) * 2 + 1) * 64 + lane] = lo;
}
}  // namespace

extern "C" int mpnn_pack_blocks(const float* const* blocks, const int* ld, const int* transposed, int n, float* packed,
                                void* stream) {
  MPNN_CHECK_ARG(blocks && ld && transposed && packed && n > 0, "mpnn_pack_blocks: bad arguments");
  for (int i0 = 0; i0 < n; i0 += PACK_MAX) {
    PackBatch pb;
    const int m = n - i0 < PACK_MAX ? n - i0 : PACK_MAX;
    for (int i = 0; i < PACK_MAX; ++i) {
      pb.W[i] = blocks[i0 + (i < m ? i : 0)];
      pb.ld[i] = ld[i0 + (i < m ? i : 0)];
      pb.transposed[i] = transposed[i0 + (i < m ? i : 0)];
    }
    MPNN_CHECK_ARG(pb.W[0] != nullptr, "mpnn_pack_blocks: null block");
    hipLaunchKernelGGL(k_pack_blocks, dim3(8, m), dim3(256), 0, (hipStream_t)stream, pb,
                       (h8*)packed + (size_t)i0 * WSPLIT_N);
  }
  MPNN_LAUNCH_CHECK();
}

// ---------------------------------------------------------------------------
// Sequence term of a decoder layer in training (reference :478-483: h_S = W_s(S) enters the third column block of
// the layer's W1): with TS = W_s W1c^T a table of n_letters rows, the term of residue n is TS[S[n]].  Three tiny
// kernels in plain fp32 FMAs, each sum in a fixed order (the operands are gradient-sized: no f16 splitting, no
// scaling):   TS[l][o] = sum_i W_s[l][i] W1c[o][i];   A[n] = A0[n] + TS[S[n]];
//             dW_s[l][i] = sum_o dTS[l][o] W1c[o][i],  dW1c[o][i] = sum_l dTS[l][o] W_s[l][i]
// (dTS arrives as mpnn_node_wgrad wrote it from one-hot rows: [feature o][letter l], row stride ld_dts).
// ---------------------------------------------------------------------------
namespace {
__global__ void __launch_bounds__(MPNN_H)
k_seq_table(const float* __restrict__ Ws, const float* __restrict__ W1c, int ld, float* __restrict__ TS) {
  __shared__ float ws[MPNN_H];
  const int l = blockIdx.x, o = threadIdx.x;
  ws[o] = Ws[(size_t)l * MPNN_H + o];
  __syncthreads();
  const float* w = W1c + (size_t)o * ld;
  float acc = 0.f;
#pragma unroll 16
  for (int i = 0; i < MPNN_H; ++i) acc = fmaf(ws[i], w[i], acc);
  TS[(size_t)l * MPNN_H + o] = acc;
}

__global__ void __launch_bounds__(256)
k_add_table_rows(const f32x4* __restrict__ A0, const f32x4* __restrict__ TS, const long long* __restrict__ S, int n_letters,
                 long long n4, f32x4* __restrict__ A) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;  // (row, quad of features)
  if (idx >= n4) return;
  const long long row = idx >> 5;
  const long long s = S[row];
  f32x4 v = A0[idx];
  if (s >= 0 && s < n_letters) v += TS[s * (MPNN_H / 4) + (idx & 31)];
  A[idx] = v;
}

__global__ void __launch_bounds__(MPNN_H)
k_seq_table_bwd(const float* __restrict__ dTSt, int ld_dts, const float* __restrict__ Ws, const float* __restrict__ W1c,
                int ld, int n_letters, float* __restrict__ dWs, float* __restrict__ dW1c) {
  const int i = threadIdx.x;
  if ((int)blockIdx.x < n_letters) {  // dW_s row l
    const int l = blockIdx.x;
    float acc = 0.f;
#pragma unroll 16
    for (int o = 0; o < MPNN_H; ++o) acc = fmaf(dTSt[(size_t)o * ld_dts + l], W1c[(size_t)o * ld + i], acc);
    dWs[(size_t)l * MPNN_H + i] = acc;
  } else {                            // dW1c row o
    const int o = blockIdx.x - n_letters;
    float acc = 0.f;
#pragma unroll 8
    for (int l = 0; l < n_letters; ++l) acc = fmaf(dTSt[(size_t)o * ld_dts + l], Ws[(size_t)l * MPNN_H + i], acc);
    dW1c[(size_t)o * MPNN_H + i] = acc;
  }
}
}  // namespace

extern "C" int mpnn_seq_table(const float* Ws, const float* W1c, int ld_w1c, int n_letters, float* TS, void* stream) {
  MPNN_CHECK_ARG(Ws && W1c && TS && n_letters > 0 && ld_w1c >= MPNN_H, "mpnn_seq_table: bad arguments");
  hipLaunchKernelGGL(k_seq_table, dim3(n_letters), dim3(MPNN_H), 0, (hipStream_t)stream, Ws, W1c, ld_w1c, TS);
  MPNN_LAUNCH_CHECK();
}

extern "C" int mpnn_add_table_rows(const float* A0, const float* TS, const int64_t* S, int n_letters, int n_rows,
                                   float* A, void* stream) {
  MPNN_CHECK_ARG(A0 && TS && S && A && n_letters > 0 && n_rows > 0, "mpnn_add_table_rows: bad arguments");
  const long long n4 = (long long)n_rows * (MPNN_H / 4);
  hipLaunchKernelGGL(k_add_table_rows, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const f32x4*)A0, (const f32x4*)TS, (const long long*)S, n_letters, n4, (f32x4*)A);
  MPNN_LAUNCH_CHECK();
}

extern "C" int mpnn_seq_table_bwd(const float* dTS_t, int ld_dts, const float* Ws, const float* W1c, int ld_w1c,
                                  int n_letters, float* dWs, float* dW1c, void* stream) {
  MPNN_CHECK_ARG(dTS_t && Ws && W1c && dWs && dW1c && n_letters > 0 && ld_dts >= n_letters && ld_w1c >= MPNN_H,
                 "mpnn_seq_table_bwd: bad arguments");
  hipLaunchKernelGGL(k_seq_table_bwd, dim3(n_letters + MPNN_H), dim3(MPNN_H), 0, (hipStream_t)stream, dTS_t, ld_dts, Ws,
                     W1c, ld_w1c, n_letters, dWs, dW1c);
  MPNN_LAUNCH_CHECK();
}

static DropCfg make_drop(float p, uint32_t seed) {
  DropCfg d;
  d.thresh = p > 0.f ? (unsigned)((double)p * 4294967296.0) : 0u;
  d.seed = seed;
  d.inv_keep = 1.0f / (1.0f - p);
  return d;
}

extern "C" int mpnn_node_update_train(const float* hV_in, const float* gsum, const float* cnt, float cnt_const,
                                      const float* mask, const float* W3_packed, const float* b3, const float* ln1_w,
                                      const float* ln1_b, const float* Win_packed, const float* b_in,
                                      const float* Wout_packed, const float* b_out, const float* ln2_w,
                                      const float* ln2_b, float scale, const float* Wproj_packed, const float* bproj,
                                      int nproj, int n_nodes, float* hV_out, float* PQ_blocks, float* x1pre, float* a_pre,
                                      float* x2pre, float p_drop, uint32_t drop_seed, void* stream) {
  MPNN_CHECK_ARG(hV_in && gsum && mask && W3_packed && b3 && ln1_w && ln1_b && Win_packed && b_in && Wout_packed &&
                     b_out && ln2_w && ln2_b && hV_out && x1pre && a_pre && x2pre && n_nodes > 0 && nproj >= 0,
                 "mpnn_node_update_train: bad arguments");
  MPNN_CHECK_ARG(nproj >= 0 && nproj <= 8 && (nproj == 0 || (Wproj_packed && bproj && PQ_blocks)),
                 "mpnn_node_update_train: projections need Wproj, bproj, PQ (at most 8 blocks)");
  MPNN_CHECK_ARG(p_drop >= 0.f && p_drop < 1.f, "mpnn_node_update_train: dropout probability must be in [0, 1)");
  hipLaunchKernelGGL(k_node_update_train, dim3((n_nodes + 63) / 64), dim3(256), 0, (hipStream_t)stream, hV_in, gsum, cnt,
                     cnt_const, mask, (const h8*)W3_packed, b3, ln1_w, ln1_b, (const h8*)Win_packed, b_in,
                     (const h8*)Wout_packed, b_out, ln2_w, ln2_b, 1.0f / scale, (const h8*)Wproj_packed, bproj, nproj,
                     n_nodes, hV_out, PQ_blocks, x1pre, a_pre, x2pre, make_drop(p_drop, drop_seed));
  MPNN_LAUNCH_CHECK();
}

static int node_update_bwd_launch(const float* dY, const ProjGrads& pg, const float* x1pre, const float* a_pre,
                                  const float* x2pre, const float* cnt, float cnt_const, const float* mask,
                                  const float* ln1_w, const float* ln1_b, const float* ln2_w, float scale,
                                  const float* WprojT_packed, int nproj, const float* WoutT_packed,
                                  const float* WinT_packed, const float* W3T_packed, int n_nodes, float* d_hV_in,
                                  float* d_gsum, float* x1, float* hid, float* d_a, float* d_o, float* d_dh,
                                  float* dvec640, float p_drop, uint32_t drop_seed, float* workspace,
                                  int workspace_floats, void* stream) {
  MPNN_CHECK_ARG(x1pre && a_pre && x2pre && mask && ln1_w && ln1_b && ln2_w && WoutT_packed && WinT_packed &&
                     W3T_packed && d_hV_in && d_gsum && x1 && hid && d_a && d_o && d_dh && dvec640 && workspace &&
                     n_nodes > 0,
                 "mpnn_node_update_bwd: bad arguments");
  const int n_wg = (n_nodes + 63) / 64;
  MPNN_CHECK_ARG((long long)workspace_floats >= (long long)n_wg * 5 * MPNN_H,
                 "mpnn_node_update_bwd: workspace too small (640 floats per 64 nodes)");
  MPNN_CHECK_ARG(nproj >= 0 && nproj <= 8 && (nproj == 0 || WprojT_packed),
                 "mpnn_node_update_bwd: 0 .. 8 projections, with their transposed weights");
  MPNN_CHECK_ARG(p_drop >= 0.f && p_drop < 1.f, "mpnn_node_update_bwd: dropout probability must be in [0, 1)");
  hipLaunchKernelGGL(k_node_update_bwd, dim3((n_nodes + 63) / 64), dim3(256), 0, (hipStream_t)stream, dY, pg,
                     x1pre, a_pre, x2pre, cnt, cnt_const, mask, ln1_w, ln1_b, ln2_w, 1.0f / scale,
                     (const h8*)WprojT_packed, nproj, (const h8*)WoutT_packed, (const h8*)WinT_packed,
                     (const h8*)W3T_packed, n_nodes, d_hV_in, d_gsum, x1, hid, d_a, d_o, d_dh, workspace,
                     make_drop(p_drop, drop_seed));
  launch_partials_reduce(workspace, n_wg, 5 * MPNN_H, 5 * MPNN_H, dvec640, 1, (hipStream_t)stream);
  MPNN_LAUNCH_CHECK();
}

extern "C" int mpnn_node_update_bwd(const float* dY, const float* dPQ_blocks, const float* x1pre, const float* a_pre,
                                    const float* x2pre, const float* cnt, float cnt_const, const float* mask,
                                    const float* ln1_w, const float* ln1_b, const float* ln2_w, float scale,
                                    const float* WprojT_packed, int nproj, const float* WoutT_packed,
                                    const float* WinT_packed, const float* W3T_packed, int n_nodes, float* d_hV_in,
                                    float* d_gsum, float* x1, float* hid, float* d_a, float* d_o, float* d_dh,
                                    float* dvec640, float p_drop, uint32_t drop_seed, float* workspace,
                                    int workspace_floats, void* stream) {
  MPNN_CHECK_ARG(nproj >= 0 && nproj <= 8 && (nproj == 0 || dPQ_blocks) && n_nodes > 0,
                 "mpnn_node_update_bwd: 0 .. 8 projections, with dPQ");
  ProjGrads pg = {};
  for (int p = 0; p < nproj; ++p) pg.p[p] = dPQ_blocks + (size_t)p * n_nodes * MPNN_H;
  return node_update_bwd_launch(dY, pg, x1pre, a_pre, x2pre, cnt, cnt_const, mask, ln1_w, ln1_b, ln2_w, scale,
                                WprojT_packed, nproj, WoutT_packed, WinT_packed, W3T_packed, n_nodes, d_hV_in, d_gsum, x1,
                                hid, d_a, d_o, d_dh, dvec640, p_drop, drop_seed, workspace, workspace_floats, stream);
}

// the same with the projections' gradients given one by one (a HOST array of nproj device pointers, NULL = no gradient):
// autograd hands them over as separate tensors, and stacking them first is a copy of nproj x n x 128 floats per layer
extern "C" int mpnn_node_update_bwd_each(const float* dY, const float* const* dPQ_each, const float* x1pre,
                                         const float* a_pre, const float* x2pre, const float* cnt, float cnt_const,
                                         const float* mask, const float* ln1_w, const float* ln1_b, const float* ln2_w,
                                         float scale, const float* WprojT_packed, int nproj, const float* WoutT_packed,
                                         const float* WinT_packed, const float* W3T_packed, int n_nodes, float* d_hV_in,
                                         float* d_gsum, float* x1, float* hid, float* d_a, float* d_o, float* d_dh,
                                         float* dvec640, float p_drop, uint32_t drop_seed, float* workspace,
                                         int workspace_floats, void* stream) {
  MPNN_CHECK_ARG(nproj >= 0 && nproj <= 8 && (nproj == 0 || dPQ_each), "mpnn_node_update_bwd_each: 0 .. 8 projections");
  ProjGrads pg = {};
  for (int p = 0; p < nproj; ++p) pg.p[p] = dPQ_each[p];
  return node_update_bwd_launch(dY, pg, x1pre, a_pre, x2pre, cnt, cnt_const, mask, ln1_w, ln1_b, ln2_w, scale,
                                WprojT_packed, nproj, WoutT_packed, WinT_packed, W3T_packed, n_nodes, d_hV_in, d_gsum, x1,
                                hid, d_a, d_o, d_dh, dvec640, p_drop, drop_seed, workspace, workspace_floats, stream);
}

static int launch_node_wgrads(const WgradJobs& jb, int n_rows, float* workspace, int workspace_floats, hipStream_t st) {
  const int nblk = jb.blk0[jb.n];
  const int max_x = workspace_floats / NW_PART / nblk;
  MPNN_CHECK_ARG(max_x >= 1, "mpnn_node_wgrad: workspace too small (16512 floats per 128x128 block and row chunk)");
  int rows = 32;  // rows per workgroup: as many row chunks as the workspace and ~256 workgroups per block allow
  while ((n_rows + rows - 1) / rows > (max_x < 256 ? max_x : 256)) rows *= 2;
  const int nx = (n_rows + rows - 1) / rows;
  hipLaunchKernelGGL(k_node_wgrad_split, dim3(nx, nblk), dim3(256), 0, st, jb, n_rows, rows, workspace);
  hipLaunchKernelGGL(k_node_wgrad_reduce, dim3((NW_PART + 63) / 64, nblk), dim3(256), 0, st, workspace, nx, jb);
  MPNN_LAUNCH_CHECK();
}

extern "C" int mpnn_node_wgrad(const float* dY, int ldy, int out_f, const float* X, int ldx, int in_f, int n_rows,
                               float* dW, int ldw, float* dbias, float* workspace, int workspace_floats, void* stream) {
  MPNN_CHECK_ARG(dY && X && dW && workspace, "mpnn_node_wgrad: null pointer");
  MPNN_CHECK_ARG(n_rows > 0 && out_f > 0 && in_f > 0 && ldy >= out_f && ldx >= in_f && ldw >= in_f,
                 "mpnn_node_wgrad: bad sizes");
  WgradJobs jb = {};
  jb.n = 1;
  jb.dY[0] = dY; jb.X[0] = X; jb.dW[0] = dW; jb.db[0] = dbias;
  jb.ldy[0] = ldy; jb.out_f[0] = out_f; jb.ldx[0] = ldx; jb.in_f[0] = in_f; jb.ldw[0] = ldw;
  jb.ny[0] = (out_f + 127) / 128;
  jb.blk0[0] = 0;
  jb.blk0[1] = jb.ny[0] * ((in_f + 127) / 128);
  return launch_node_wgrads(jb, n_rows, workspace, workspace_floats, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------
// mpnn_node_wgrad for callers that cannot allocate (HIP-graph capture, serving loops, harnesses that forbid hipMalloc):
// the split-K partials go to a file-scope device array instead of a caller workspace.  Each GPU holds its own copy of
// the array (NOALLOC_PARTS x 16512 floats = 8.06 MiB) at its own address until the process ends; the address is looked
// up once per device (hipGetSymbolAddress) and only that pointer is ever handed to kernels or runtime calls.
// Nothing in the array has to start at zero: k_node_wgrad overwrites every partial that k_node_wgrad_reduce reads.
// Calls that overlap on one GPU (two streams, two host threads) share the array and race.
// ---------------------------------------------------------------------------
constexpr int NOALLOC_PARTS = 128;   // partial tiles: 128x128 output blocks x row chunks
constexpr int NOALLOC_MAX_DEVICES = 64;
static __device__ __align__(256) float scratch[(size_t)NOALLOC_PARTS * NW_PART];
static float* g_scratch_ptr[NOALLOC_MAX_DEVICES];
static std::mutex g_scratch_mutex;

static int scratch_of_current_device(float** out) {
  int dev = -1;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) {
    mpnn_set_error(hipGetErrorString(e));
    return (int)e;
  }
  MPNN_CHECK_ARG(dev >= 0 && dev < NOALLOC_MAX_DEVICES, "mpnn_node_wgrad_noalloc: device ordinal out of range");
  std::lock_guard<std::mutex> lock(g_scratch_mutex);
  if (!g_scratch_ptr[dev]) {
    float* p = nullptr;
    e = hipGetSymbolAddress((void**)&p, HIP_SYMBOL(scratch));  // this device's copy of the array
    if (e != hipSuccess || !p) {
      mpnn_set_error(e != hipSuccess ? hipGetErrorString(e) : "mpnn_node_wgrad_noalloc: no address for the scratch array");
      return e != hipSuccess ? (int)e : -1;
    }
    g_scratch_ptr[dev] = p;
  }
  *out = g_scratch_ptr[dev];
  return 0;
}

extern "C" int mpnn_node_wgrad_noalloc(const float* dY, int ldy, int out_f, const float* X, int ldx, int in_f, int n_rows,
                                       float* dW, int ldw, float* dbias, void* stream) {
  MPNN_CHECK_ARG(dY && X && dW, "mpnn_node_wgrad_noalloc: null pointer");
  MPNN_CHECK_ARG(n_rows > 0 && out_f > 0 && in_f > 0 && ldy >= out_f && ldx >= in_f && ldw >= in_f,
                 "mpnn_node_wgrad_noalloc: bad sizes");
  const long long blocks = (long long)((out_f + 127) / 128) * ((in_f + 127) / 128);
  MPNN_CHECK_ARG(blocks <= NOALLOC_PARTS && n_rows <= (1 << 24),
                 "mpnn_node_wgrad_noalloc: beyond the built-in scratch (ceil(out_f/128) x ceil(in_f/128) <= 128 output "
                 "blocks, n_rows <= 2^24); use mpnn_node_wgrad with a caller workspace");
  float* ws = nullptr;
  const int rc = scratch_of_current_device(&ws);
  if (rc) return rc;
  WgradJobs jb = {};
  jb.n = 1;
  jb.dY[0] = dY; jb.X[0] = X; jb.dW[0] = dW; jb.db[0] = dbias;
  jb.ldy[0] = ldy; jb.out_f[0] = out_f; jb.ldx[0] = ldx; jb.in_f[0] = in_f; jb.ldw[0] = ldw;
  jb.ny[0] = (out_f + 127) / 128;
  jb.blk0[0] = 0;
  jb.blk0[1] = (int)blocks;
  return launch_node_wgrads(jb, n_rows, ws, NOALLOC_PARTS * NW_PART, (hipStream_t)stream);
}

extern "C" int mpnn_node_wgrad_batch(int n_jobs, const float* const* dY, const int* ldy, const int* out_f,
                                     const float* const* X, const int* ldx, const int* in_f, float* const* dW,
                                     const int* ldw, float* const* dbias, int n_rows, float* workspace,
                                     int workspace_floats, void* stream) {
  MPNN_CHECK_ARG(dY && ldy && out_f && X && ldx && in_f && dW && ldw && dbias && workspace && n_rows > 0,
                 "mpnn_node_wgrad_batch: null pointer");
  MPNN_CHECK_ARG(n_jobs >= 1 && n_jobs <= WG_JOBS, "mpnn_node_wgrad_batch: 1 .. 12 jobs per call");
  WgradJobs jb = {};
  jb.n = n_jobs;
  jb.blk0[0] = 0;
  for (int j = 0; j < n_jobs; ++j) {
    MPNN_CHECK_ARG(dY[j] && X[j] && dW[j] && out_f[j] > 0 && in_f[j] > 0 && ldy[j] >= out_f[j] && ldx[j] >= in_f[j] &&
                       ldw[j] >= in_f[j],
                   "mpnn_node_wgrad_batch: bad job");
    jb.dY[j] = dY[j]; jb.X[j] = X[j]; jb.dW[j] = dW[j]; jb.db[j] = dbias[j];
    jb.ldy[j] = ldy[j]; jb.out_f[j] = out_f[j]; jb.ldx[j] = ldx[j]; jb.in_f[j] = in_f[j]; jb.ldw[j] = ldw[j];
    jb.ny[j] = (out_f[j] + 127) / 128;
    jb.blk0[j + 1] = jb.blk0[j] + jb.ny[j] * ((in_f[j] + 127) / 128);
  }
  return launch_node_wgrads(jb, n_rows, workspace, workspace_floats, (hipStream_t)stream);
}
