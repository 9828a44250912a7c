// Sampling head shared by k_sample (mpnn_node.hip) and the one-launch-per-level decoder k_decode_level
// (mpnn_edge.hip).  Device code only; include inside MPNN_KERNELS_BEGIN / MPNN_KERNELS_END.
//   logits = W_out h + b;  log_probs = log_softmax(logits)  (the model's distribution, what a teacher-forced
//   forward returns at this position);  S ~ softmax((logits + bias) / temperature)  by inverse CDF with a
//   counter-based uniform u(seed, residue);  fixed residues (designable[n] == 0) keep S_in[n].
//   Then the residue's sequence term is folded into the decoder's gathered rows: A_l[n] += TS[S][128 l ..] for
//   every decoder layer l (reference h_S = W_s(S) enters W1 through h_ES, :478-479, :513).
#pragma once

struct SampleRows {
  float* A[4];
  int ld[4], off[4];
  int n_layers;
};

// everything the head needs beyond the residues' features (k_dec_layer_level<true>: the last decoder layer's launch
// also draws)
struct LevelHead {
  const f32x4* Wp;
  const float *bias, *logit_bias;
  int n_letters;
  float inv_temperature;
  unsigned long long seed;
  const float* designable;
  const long long* S_in;
  long long* S_out;
  float* log_probs;
  const float* TS;
  int ld_ts;
  SampleRows rows;
};

__device__ __forceinline__ float uniform01(unsigned long long seed, unsigned long long ctr) {
  unsigned long long z = seed + 0x9E3779B97F4A7C15ull * (ctr + 1);  // splitmix64
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  return (float)(z >> 40) * (1.0f / 16777216.0f);  // 24 bits -> [0, 1)
}

// One wave: logits, log-probs and the draw for the residue `nd` of this lane's column (col = lane & 15; `live`
// columns write).  pr: 16 x 33 floats of LDS scratch, pick: 16 ints; pick[col] <- the letter.
// h: the residues' final node features as the lane holds them (features 16 t + 4 g .. + 3 of column col)
__device__ __forceinline__ void sample_columns_h(const f32x4 (&h)[MPNN_NT][1], const f32x4* __restrict__ Wp,
                                                 const float* __restrict__ bias, const float* __restrict__ logit_bias,
                                                 int n_letters, int nd, bool live, float inv_temperature,
                                                 unsigned long long seed, const float* __restrict__ designable,
                                                 const long long* __restrict__ S_in, long long* __restrict__ S_out,
                                                 float* __restrict__ log_probs, float (*pr)[33], int* pick, int lane) {
  const int g = lane >> 4, col = lane & 15;
  f32x4 y[2][1];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int letter = 16 * t + 4 * g + r;
      y[t][0][r] = letter < n_letters ? bias[letter] : 0.f;
    }
#ifndef MPNN_ABL_SAMPLE_NOGEMM
  tile_gemm<1, MPNN_NT, 0, 2>(Wp, lane, h, y);
#endif
  // model log-probs
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
  // tempered, biased weights for the draw
  float z[2][4], zm = -INFINITY;
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int letter = 16 * t + 4 * g + r;
      if (letter < n_letters) {
        if (live) log_probs[(size_t)nd * n_letters + letter] = y[t][0][r] - lse;
        z[t][r] = (y[t][0][r] + (logit_bias ? logit_bias[letter] : 0.f)) * inv_temperature;
        zm = fmaxf(zm, z[t][r]);
      }
    }
  zm = fmaxf(zm, __shfl_xor(zm, 16));
  zm = fmaxf(zm, __shfl_xor(zm, 32));
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int letter = 16 * t + 4 * g + r;
      if (letter < n_letters) pr[col][letter] = expf(z[t][r] - zm);
    }
  __builtin_amdgcn_s_waitcnt(0);
  __builtin_amdgcn_wave_barrier();
  if (g == 0) {
    int s = 0;
#ifdef MPNN_ABL_SAMPLE_NOLOOP
    if (false) {
#else
    if (designable[nd] != 0.f) {
#endif
      float tot = 0.f;
      for (int a = 0; a < n_letters; ++a) tot += pr[col][a];
      const float u = uniform01(seed, (unsigned long long)nd) * tot;
      float acc = 0.f;
      s = n_letters - 1;
      for (int a = 0; a < n_letters; ++a) {
        acc += pr[col][a];
        if (u < acc) { s = a; break; }
      }
    } else {
      s = (int)S_in[nd];
    }
    pick[col] = s;
    if (live) S_out[nd] = s;
  }
}

__device__ __forceinline__ void sample_columns(const float* __restrict__ hV, const f32x4* __restrict__ Wp,
                                               const float* __restrict__ bias, const float* __restrict__ logit_bias,
                                               int n_letters, int nd, bool live, float inv_temperature,
                                               unsigned long long seed, const float* __restrict__ designable,
                                               const long long* __restrict__ S_in, long long* __restrict__ S_out,
                                               float* __restrict__ log_probs, float (*pr)[33], int* pick, int lane) {
  const int g = lane >> 4;
  f32x4 h[MPNN_NT][1];
#pragma unroll
  for (int t = 0; t < MPNN_NT; ++t) h[t][0] = *reinterpret_cast<const f32x4*>(hV + (size_t)nd * MPNN_H + 16 * t + 4 * g);
  sample_columns_h(h, Wp, bias, logit_bias, n_letters, nd, live, inv_temperature, seed, designable, S_in, S_out, log_probs,
                   pr, pick, lane);
}

// Fold the sequence term of the chosen letters into every decoder layer's A rows (2 floats per lane per row); the
// rows belong to these residues alone until later levels gather them, so plain read-modify-write; the
// nlive x n_layers row updates are dealt round-robin to the NW waves, each with all its loads in flight before
// the first store.  MAXU >= ceil(max nlive * 4 / NW).
// row_of(l, n) -> pointer to residue n's A row of decoder layer l.
template <int NW, int MAXU, typename RowOf>
__device__ __forceinline__ void fold_sequence_rows(RowOf row_of, int n_layers, const float* __restrict__ TS, int ld_ts,
                                                   const int* __restrict__ nodes, int nlive, const int* pick,
                                                   int wave, int lane) {
  const int nupd = nlive * n_layers;
  float2 v[MAXU], add[MAXU];
  float2* dst[MAXU];
#pragma unroll
  for (int i = 0; i < MAXU; ++i) {
    const int u = wave + NW * i;
    if (u < nupd) {
      const int c = u % nlive, l = u / nlive;
      const int n = nodes[c], s = pick[c];
      dst[i] = reinterpret_cast<float2*>(row_of(l, n)) + lane;
      v[i] = *dst[i];
      add[i] = *(reinterpret_cast<const float2*>(TS + (size_t)s * ld_ts + l * MPNN_H) + lane);
    }
  }
#pragma unroll
  for (int i = 0; i < MAXU; ++i) {
    if (wave + NW * i < nupd) {
      v[i].x += add[i].x;
      v[i].y += add[i].y;
      *dst[i] = v[i];
    }
  }
}
