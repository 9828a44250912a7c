// Helpers shared by the forward (mpnn_edge.hip) and backward (mpnn_edge_bwd.hip) per-edge kernels.
#pragma once
#include "mpnn_common.h"
#include <type_traits>

namespace {
// make a (uniform) pointer opaque once per loop iteration: keeps LICM from hoisting hundreds of
// loop-invariant weight / bias loads out of the persistent loop and pinning them in registers
// The opaque value is a ZERO OFFSET added to the pointer, not the pointer itself: laundering the pointer also hides
// where it came from (a kernel argument = global memory), and hipcc then emits flat_load for everything read through
// it.  Flat loads count on lgkmcnt as well as vmcnt, so every "wait for my LDS read" behind one waits for an L2 / HBM
// round trip too (measured on the edge update's W13 ring).  GLOBAL(p) states the address space outright.
// LAUNDER_FLAT: the pointer itself goes through the asm; reads through it become flat_load.  Kept for ONE stream, the
// edge-update backward's W11b^T ring, where the flat form measured FASTER (1.665 against 1.81 ms per training step).
#define LAUNDER_FLAT(p) asm volatile("" : "+s"(p))
#define LAUNDER(p)                                    \
  do {                                                \
    int z__;                                          \
    asm volatile("s_mov_b32 %0, 0" : "=s"(z__));      \
    (p) += z__;                                       \
  } while (0)
#define GLOBAL(p) ((const __attribute__((address_space(1))) std::remove_cv_t<std::remove_pointer_t<decltype(p)>>*)(p))
#define FENCE() __builtin_amdgcn_sched_barrier(0)

__device__ __forceinline__ void lds_fill(h8* dst, const h8* __restrict__ src, int n) {
  // batches of 8 loads in flight per thread before their LDS writes (a plain copy loop waits for every load in turn,
  // which is what a short-lived workgroup - the sampler's level kernels - spends most of its time on)
  const int stride = blockDim.x;
  for (int i0 = threadIdx.x; i0 < n; i0 += 8 * stride) {
    h8 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (i0 + u * stride < n) v[u] = src[i0 + u * stride];
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (i0 + u * stride < n) dst[i0 + u * stride] = v[u];
  }
}

__device__ __forceinline__ f32x4 row_tile(const float* row, int t) {
#ifdef MPNN_ABL_NOROW  // timing-only ablation
  return f32x4{0.1f, 0.2f, 0.3f, 0.4f};
#endif
  return *reinterpret_cast<const f32x4*>(row + 16 * t);
}

// Dropout keep-decision of one fragment-layout element (training; reference EncLayer.dropout3, :94).  Counter-based:
// the same (seed, element id) gives the same decision in the forward and the backward kernel, nothing is stored.
// element id = ((node * 8 + t) * KT + c) * 256 + lane * 4 + r  (its float index in the fragment array, mod 2^32)
__device__ __forceinline__ bool dropout_keep(unsigned id, unsigned seed, unsigned thresh) {
  unsigned h = (id * 0x9E3779B1u) ^ seed;
  h ^= h >> 16;
  h *= 0x85EBCA6Bu;
  h ^= h >> 13;
  h *= 0xC2B2AE35u;
  h ^= h >> 16;
  return h >= thresh;  // thresh = p * 2^32
}
__device__ __forceinline__ f32x4 dropout4(f32x4 v, unsigned id0, unsigned seed, unsigned thresh, float inv_keep) {
  f32x4 o;
#pragma unroll
  for (int r = 0; r < 4; ++r) o[r] = dropout_keep(id0 + r, seed, thresh) ? v[r] * inv_keep : 0.f;
  return o;
}

// Residue range of one wave of a persistent edge kernel.  XCD-aware: blocks b and b + 8 share an XCD (observed
// round-robin dispatch; a speed assumption only), so each of the 8 block classes walks its own contiguous eighth of
// the residues - the neighbour rows a residue gathers belong to its own protein, which then stays in that XCD's
// 4 MiB L2 instead of every XCD streaming every protein's rows.
struct ResidueRange {
  int first, end, stride;
};
__device__ __forceinline__ ResidueRange residue_range(int N, int waves, int wave) {
  const int G = gridDim.x, b = blockIdx.x;
  if (G >= 8 && (G & 7) == 0) {
    const int chunk = (N + 7) >> 3;
    const int lo = (b & 7) * chunk;
    const int hi = lo + chunk < N ? lo + chunk : N;
    return ResidueRange{lo + (b >> 3) * waves + wave, hi, (G >> 3) * waves};
  }
  return ResidueRange{b * waves + wave, N, G * waves};
}

constexpr int NKK = MPNN_NT / 2;       // k-steps (32 features) per 128-wide layer
constexpr int WSPLIT = 8 * NKK * 2 * 64;  // h8 elements of one packed 128x128 matrix (64 KiB)

}  // namespace

// ---- host side -------------------------------------------------------------
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

static bool mult4(int a, int b = 0, int c = 0, int d = 0, int e = 0, int f = 0) {
  return ((a | b | c | d | e | f) & 3) == 0;
}

