// Output head backward and the label-smoothed loss (SURVEY section 8f "next" N3).
//   log_probs = log_softmax(W_out h + b)                         (reference model_utils.py :527-528)
//   loss_smoothed(S, log_probs, mask)                            (reference train.py :63-73)
// The forward of the head is mpnn_logits_log_softmax (mpnn_node.hip).  Everything here is tiny (N x 21 against the
// N x 48 x 128 edge tensors): the point is one launch each instead of a dozen ATen launches, and no host sync.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <math.h>
#include "mpnn_common.h"

namespace {

constexpr int HB_NODES = 64;    // nodes per workgroup
constexpr int HB_MAXL = 32;     // letters (padded)
constexpr int HB_PART = HB_MAXL * MPNN_H + HB_MAXL;  // floats of one workgroup's partial (dW | db)

// d_logits = d_logp - softmax * sum(d_logp);  d_hV = d_logits W;  dW += d_logits^T hV;  db += sum_n d_logits
__global__ void __launch_bounds__(256)
k_head_bwd(const float* __restrict__ d_logp, const float* __restrict__ logp, const float* __restrict__ hV,
           const float* __restrict__ W, int N, int n_letters, float* __restrict__ d_hV, float* __restrict__ part) {
  __shared__ float dl[HB_NODES][HB_MAXL + 1];
  __shared__ float Wl[HB_MAXL][MPNN_H];
  __shared__ float hv[HB_NODES][MPNN_H + 4];
  const int tid = threadIdx.x;
  const int n0 = blockIdx.x * HB_NODES;
  for (int i = tid; i < HB_MAXL * MPNN_H; i += 256) {
    const int a = i / MPNN_H;
    Wl[a][i % MPNN_H] = a < n_letters ? W[i] : 0.f;
  }
  for (int i = tid; i < HB_NODES * (MPNN_H / 4); i += 256) {
    const int n = i / (MPNN_H / 4), q = i % (MPNN_H / 4);
    f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
    if (n0 + n < N) v = *reinterpret_cast<const f32x4*>(hV + (size_t)(n0 + n) * MPNN_H + 4 * q);
    hv[n][4 * q] = v.x; hv[n][4 * q + 1] = v.y; hv[n][4 * q + 2] = v.z; hv[n][4 * q + 3] = v.w;
  }
  {  // four threads per node share its letters
    const int n = tid >> 2, q = tid & 3;
    const bool live = n0 + n < N;
    float g[HB_MAXL / 4], p[HB_MAXL / 4], s = 0.f;
#pragma unroll
    for (int u = 0; u < HB_MAXL / 4; ++u) {
      const int a = q + 4 * u;
      g[u] = p[u] = 0.f;
      if (live && a < n_letters) {
        g[u] = d_logp[(size_t)(n0 + n) * n_letters + a];
        p[u] = __expf(logp[(size_t)(n0 + n) * n_letters + a]);
      }
      s += g[u];
    }
    s += __shfl_xor(s, 1);
    s += __shfl_xor(s, 2);
#pragma unroll
    for (int u = 0; u < HB_MAXL / 4; ++u) dl[n][q + 4 * u] = g[u] - p[u] * s;
  }
  __syncthreads();
  {  // d_hV: thread = (node, 32-feature block)
    const int n = tid >> 2, f0 = (tid & 3) * 32;
    float acc[32];
#pragma unroll
    for (int j = 0; j < 32; ++j) acc[j] = 0.f;
    for (int a = 0; a < n_letters; ++a) {
      const float d = dl[n][a];
#pragma unroll
      for (int j = 0; j < 32; ++j) acc[j] = fmaf(d, Wl[a][f0 + j], acc[j]);
    }
    if (n0 + n < N) {
#pragma unroll
      for (int j = 0; j < 32; j += 4)
        *reinterpret_cast<f32x4*>(d_hV + (size_t)(n0 + n) * MPNN_H + f0 + j) =
            f32x4{acc[j], acc[j + 1], acc[j + 2], acc[j + 3]};
    }
  }
  {  // dW partial of this workgroup: thread = (feature, letter parity)
    const int f = tid & (MPNN_H - 1), a0 = tid >> 7;
    float acc[HB_MAXL / 2];
#pragma unroll
    for (int u = 0; u < HB_MAXL / 2; ++u) acc[u] = 0.f;
    for (int n = 0; n < HB_NODES; ++n) {
      const float h = hv[n][f];
#pragma unroll
      for (int u = 0; u < HB_MAXL / 2; ++u) acc[u] = fmaf(dl[n][a0 + 2 * u], h, acc[u]);
    }
    // this workgroup's partial [HB_MAXL x 128 | HB_MAXL]: the launcher's k_partials_reduce adds them in a fixed order
    float* P = part + (size_t)blockIdx.x * HB_PART;
#pragma unroll
    for (int u = 0; u < HB_MAXL / 2; ++u) P[(size_t)(a0 + 2 * u) * MPNN_H + f] = acc[u];
  }
  if (tid < HB_MAXL) {
    float s = 0.f;
    for (int n = 0; n < HB_NODES; ++n) s += dl[n][tid];
    part[(size_t)blockIdx.x * HB_PART + HB_MAXL * MPNN_H + tid] = s;
  }
}

// loss[n] = -sum_a t[n,a] logp[n,a],  t = (onehot(S) + weight / A) / (1 + weight);   loss_sum += sum_n loss mask * scale;
// d_logp[n,a] = -t[n,a] mask[n] scale   (the gradient of loss_sum)
__global__ void __launch_bounds__(256)
k_loss_smoothed(const float* __restrict__ logp, const long long* __restrict__ S, const float* __restrict__ mask, int N,
                int n_letters, float weight, float scale, float* __restrict__ loss, float* __restrict__ loss_part,
                float* __restrict__ d_logp) {
  __shared__ float part[4];
  const int n = blockIdx.x * 256 + threadIdx.x;
  float contrib = 0.f;
  if (n < N) {
    const int s = (int)S[n];
    const float m = mask[n];
    const float base = weight / (float)n_letters;
    float tot = 0.f;  // the reference normalises by the actual sum of the smoothed row (train.py:69)
    for (int a = 0; a < n_letters; ++a) tot += (a == s ? 1.f : 0.f) + base;
    float l = 0.f;
    for (int a = 0; a < n_letters; ++a) {
      const float t = ((a == s ? 1.f : 0.f) + base) / tot;
      l -= t * logp[(size_t)n * n_letters + a];
      if (d_logp) d_logp[(size_t)n * n_letters + a] = -t * m * scale;
    }
    loss[n] = l;
    contrib = l * m * scale;
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) contrib += __shfl_xor(contrib, o);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = contrib;
  __syncthreads();
  if (threadIdx.x == 0) loss_part[blockIdx.x] = (part[0] + part[1]) + (part[2] + part[3]);
}

// out[0] = sum of part[0 .. n) in a fixed tree (one workgroup: strided per-thread sums, then an LDS tree): the scalar
// sums of the step (loss, squared gradient norm) without floating-point atomics
__global__ void __launch_bounds__(1024)
k_sum_ordered(const float* __restrict__ part, int n, float* __restrict__ out) {
  __shared__ float sh[1024];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 1024) s += part[i];
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int o = 512; o >= 1; o >>= 1) {
    if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = sh[0];
}

// element-wise probes of the two device helpers every MLP on the path is built from (unit tests of the arithmetic):
//   mode 0: y = gelu(x)          (mpnn_common.h; reference nn.GELU(), :70, :115, :149)
//   mode 1: y = hi + lo of the f16 hi/lo split the matrix-pipe kernels feed their MFMAs with (x must be |x| < 65504)
//   mode 2: y = the ONE f16 rounding of the f16-products mode (round to nearest even; +-inf from |x| >= 65520 on)
__global__ void k_probe(const float* __restrict__ x, int n, int mode, float* __restrict__ y) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int i8 = i * 8;
  if (i8 >= n) return;
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = i8 + j < n ? x[i8 + j] : 0.f;
  if (mode == 0) {
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = gelu(v[j]);
  } else if (mode == 2) {
    const h8 h = f16_rne8(v);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (float)h[j];
  } else {
    BFrag b;
    split_pair(f32x4{v[0], v[1], v[2], v[3]}, f32x4{v[4], v[5], v[6], v[7]}, b);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (float)b.hi[j] + (float)b.lo[j];
  }
#pragma unroll
  for (int j = 0; j < 8; ++j)
    if (i8 + j < n) y[i8 + j] = v[j];
}

// ---------------------------------------------------------------------------
// Multi-tensor AdamW and gradient-norm clipping (reference train.py:217 AdamW(lr, weight_decay=1e-2), :389-393
// clip_grad_norm_ + optimizer.step): every parameter tensor of the model in one launch per 48 tensors, the clip
// coefficient read from device memory so the step never synchronises with the host.
// ---------------------------------------------------------------------------
constexpr int MT_MAX = 48;  // tensors per launch (the table travels as a kernel argument)
struct MultiTensor {
  float* p[MT_MAX];
  const float* g[MT_MAX];
  float* m[MT_MAX];
  float* v[MT_MAX];
  int numel[MT_MAX];
  int n;
};
constexpr int MT_CHUNK = 2048;  // elements per workgroup

// ss_part[blockIdx.y * gridDim.x + blockIdx.x] = sum of squares of this workgroup's chunk of gradient elements
// (0 past a tensor's end); k_sum_ordered adds the partials
__global__ void __launch_bounds__(256)
k_grad_sumsq(MultiTensor t, float* __restrict__ ss_part) {
  __shared__ float part[4];
  const int ti = blockIdx.y, n = t.numel[ti];
  const int i0 = blockIdx.x * MT_CHUNK;
  const float* __restrict__ g = t.g[ti];
  float s = 0.f;
  for (int i = i0 + threadIdx.x; i < min(n, i0 + MT_CHUNK); i += 256) s += g[i] * g[i];
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) ss_part[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = (part[0] + part[1]) + (part[2] + part[3]);
}

// torch.nn.utils.clip_grad_norm_: g *= min(1, max_norm / (sqrt(sumsq) + 1e-6))
__device__ __forceinline__ float clip_coef(const float* sumsq, float max_norm) {
  if (!sumsq || !(max_norm > 0.f)) return 1.0f;
  const float c = max_norm / (sqrtf(*sumsq) + 1e-6f);
  // a NaN norm propagates, as torch's clamp(nan, max = 1) does: every gradient becomes NaN, an obvious failure (an
  // infinite norm gives 0 there and here: finite gradients are zeroed, the infinite ones become NaN)
  return c < 1.0f || c != c ? c : 1.0f;
}

__global__ void __launch_bounds__(256)
k_grad_scale(MultiTensor t, const float* __restrict__ sumsq, float max_norm) {
  const int ti = blockIdx.y, n = t.numel[ti];
  const int i0 = blockIdx.x * MT_CHUNK;
  if (i0 >= n) return;
  const float c = clip_coef(sumsq, max_norm);
  if (c == 1.0f) return;  // (false for NaN: the scaling below then runs)
  float* g = t.p[ti];  // the gradient travels in the p slot for this kernel
  for (int i = i0 + threadIdx.x; i < min(n, i0 + MT_CHUNK); i += 256) g[i] *= c;
}

// torch.optim.AdamW (decoupled weight decay, no amsgrad, not maximize):
//   p *= 1 - lr wd;  m = b1 m + (1 - b1) g;  v = b2 v + (1 - b2) g^2;
//   p -= (lr / bc1) m / (sqrt(v) / sqrt(bc2) + eps),   bc_i = 1 - b_i^step
__global__ void __launch_bounds__(256)
k_adamw(MultiTensor t, float lr, float b1, float b2, float eps, float wd, float bc1, float bc2_sqrt,
        const float* __restrict__ sumsq, float max_norm) {
  const int ti = blockIdx.y, n = t.numel[ti];
  const int i0 = blockIdx.x * MT_CHUNK;
  if (i0 >= n) return;
  const float c = clip_coef(sumsq, max_norm);
  float* __restrict__ p = t.p[ti];
  const float* __restrict__ g = t.g[ti];
  float* __restrict__ m = t.m[ti];
  float* __restrict__ v = t.v[ti];
  const float step_size = lr / bc1;
  for (int i = i0 + threadIdx.x; i < min(n, i0 + MT_CHUNK); i += 256) {
    const float gi = g[i] * c;
    float pi = p[i] * (1.0f - lr * wd);
    const float mi = b1 * m[i] + (1.0f - b1) * gi;
    const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    pi -= step_size * (mi / (sqrtf(vi) / bc2_sqrt + eps));
    p[i] = pi;
  }
}

static int fill_table(MultiTensor& t, int k0, int n, float* const* p, const float* const* g, float* const* m,
                      float* const* v, const int* numel, int* max_numel) {
  t.n = n - k0 < MT_MAX ? n - k0 : MT_MAX;
  *max_numel = 0;
  for (int j = 0; j < t.n; ++j) {
    t.p[j] = p ? p[k0 + j] : nullptr;
    t.g[j] = g ? g[k0 + j] : nullptr;
    t.m[j] = m ? m[k0 + j] : nullptr;
    t.v[j] = v ? v[k0 + j] : nullptr;
    t.numel[j] = numel[k0 + j];
    if (numel[k0 + j] > *max_numel) *max_numel = numel[k0 + j];
  }
  return t.n;
}

// *flag |= bit when any element is outside (-limit, limit) or is NaN (limit = +inf: a pure finiteness check)
__global__ void k_range_flag(const float* __restrict__ x, long long n, float limit, int* __restrict__ flag, int bit) {
  bool bad = false;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    bad |= !(fabsf(x[i]) < limit);
  if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(flag, bit);
}

}  // namespace

static long long sumsq_partials(int n, const int* numel) {
  long long total = 0;
  for (int k0 = 0; k0 < n; k0 += MT_MAX) {
    int mx = 0;
    const int m = n - k0 < MT_MAX ? n - k0 : MT_MAX;
    for (int j = 0; j < m; ++j) mx = numel[k0 + j] > mx ? numel[k0 + j] : mx;
    total += (long long)((mx + MT_CHUNK - 1) / MT_CHUNK) * m;
  }
  return total;
}

extern "C" int mpnn_grad_sumsq(int n, const float* const* grads, const int* numel, float* sumsq, float* workspace,
                               int workspace_floats, void* stream) {
  MPNN_CHECK_ARG(n > 0 && grads && numel && sumsq && workspace, "mpnn_grad_sumsq: bad arguments");
  for (int j = 0; j < n; ++j) MPNN_CHECK_ARG(grads[j] && numel[j] > 0, "mpnn_grad_sumsq: null or empty tensor");
  const long long need = sumsq_partials(n, numel);
  MPNN_CHECK_ARG(need <= workspace_floats,
                 "mpnn_grad_sumsq: workspace too small (one float per 2048 elements of the largest tensor of each group of "
                 "48 tensors, times the tensors of the group)");
  long long off = 0;
  for (int k0 = 0; k0 < n; k0 += MT_MAX) {
    MultiTensor t;
    int mx;
    fill_table(t, k0, n, nullptr, grads, nullptr, nullptr, numel, &mx);
    const int gx = (mx + MT_CHUNK - 1) / MT_CHUNK;
    hipLaunchKernelGGL(k_grad_sumsq, dim3(gx, t.n), dim3(256), 0, (hipStream_t)stream, t, workspace + off);
    off += (long long)gx * t.n;
  }
  hipLaunchKernelGGL(k_sum_ordered, dim3(1), dim3(1024), 0, (hipStream_t)stream, workspace, (int)need, sumsq);
  MPNN_LAUNCH_CHECK();
}

extern "C" int mpnn_grad_scale(int n, float* const* grads, const int* numel, const float* sumsq, float max_norm,
                               void* stream) {
  MPNN_CHECK_ARG(n > 0 && grads && numel && sumsq && max_norm > 0.f, "mpnn_grad_scale: bad arguments");
  for (int j = 0; j < n; ++j) MPNN_CHECK_ARG(grads[j] && numel[j] > 0, "mpnn_grad_scale: null or empty tensor");
  for (int k0 = 0; k0 < n; k0 += MT_MAX) {
    MultiTensor t;
    int mx;
    fill_table(t, k0, n, grads, nullptr, nullptr, nullptr, numel, &mx);
    hipLaunchKernelGGL(k_grad_scale, dim3((mx + MT_CHUNK - 1) / MT_CHUNK, t.n), dim3(256), 0, (hipStream_t)stream, t, sumsq,
                       max_norm);
  }
  MPNN_LAUNCH_CHECK();
}

extern "C" int mpnn_adamw_step(int n, float* const* params, const float* const* grads, float* const* exp_avg,
                               float* const* exp_avg_sq, const int* numel, float lr, float beta1, float beta2, float eps,
                               float weight_decay, int step, const float* grad_sumsq, float max_norm, void* stream) {
  MPNN_CHECK_ARG(n > 0 && params && grads && exp_avg && exp_avg_sq && numel && step >= 1, "mpnn_adamw_step: bad arguments");
  MPNN_CHECK_ARG(beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f && eps >= 0.f && lr >= 0.f,
                 "mpnn_adamw_step: bad hyper-parameters");
  for (int j = 0; j < n; ++j)
    MPNN_CHECK_ARG(params[j] && grads[j] && exp_avg[j] && exp_avg_sq[j] && numel[j] > 0, "mpnn_adamw_step: null or empty tensor");
  const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
  for (int k0 = 0; k0 < n; k0 += MT_MAX) {
    MultiTensor t;
    int mx;
    fill_table(t, k0, n, params, grads, exp_avg, exp_avg_sq, numel, &mx);
    hipLaunchKernelGGL(k_adamw, dim3((mx + MT_CHUNK - 1) / MT_CHUNK, t.n), dim3(256), 0, (hipStream_t)stream, t, lr, beta1,
                       beta2, eps, weight_decay, (float)bc1, (float)sqrt(bc2), grad_sumsq, max_norm);
  }
  MPNN_LAUNCH_CHECK();
}

extern "C" int mpnn_range_flag(const float* x, long long n, float limit, int32_t* flag, int bit, void* stream) {
  MPNN_CHECK_ARG(x && flag && n > 0 && bit != 0, "mpnn_range_flag: bad arguments");
  const long long blocks = (n + 255) / 256;
  hipLaunchKernelGGL(k_range_flag, dim3((unsigned)(blocks < 1024 ? blocks : 1024)), dim3(256), 0, (hipStream_t)stream, x, n,
                     limit, flag, bit);
  MPNN_LAUNCH_CHECK();
}

extern "C" int mpnn_probe_elementwise(const float* x, int n, int mode, float* y, void* stream) {
  MPNN_CHECK_ARG(x && y && n > 0 && mode >= 0 && mode <= 2, "mpnn_probe_elementwise: bad arguments");
  hipLaunchKernelGGL(k_probe, dim3((n + 2047) / 2048), dim3(256), 0, (hipStream_t)stream, x, n, mode, y);
  MPNN_LAUNCH_CHECK();
}

extern "C" int mpnn_logits_log_softmax_bwd(const float* d_log_probs, const float* log_probs, const float* hV,
                                           const float* W_out, int n_letters, int n_nodes, float* d_hV, float* dW,
                                           float* db, float* workspace, int workspace_floats, void* stream) {
  MPNN_CHECK_ARG(d_log_probs && log_probs && hV && W_out && d_hV && dW && workspace, "mpnn_logits_log_softmax_bwd: null pointer");
  MPNN_CHECK_ARG(n_nodes > 0 && n_letters >= 1 && n_letters <= HB_MAXL, "mpnn_logits_log_softmax_bwd: bad sizes");
  const int n_wg = (n_nodes + HB_NODES - 1) / HB_NODES;
  MPNN_CHECK_ARG((long long)workspace_floats >= (long long)n_wg * HB_PART,
                 "mpnn_logits_log_softmax_bwd: workspace too small (4128 floats per 64 nodes)");
  hipLaunchKernelGGL(k_head_bwd, dim3(n_wg), dim3(256), 0, (hipStream_t)stream, d_log_probs, log_probs, hV, W_out, n_nodes,
                     n_letters, d_hV, workspace);
  launch_partials_reduce(workspace, n_wg, HB_PART, n_letters * MPNN_H, dW, 1, (hipStream_t)stream);
  if (db) launch_partials_reduce(workspace + HB_MAXL * MPNN_H, n_wg, HB_PART, n_letters, db, 1, (hipStream_t)stream);
  MPNN_LAUNCH_CHECK();
}

extern "C" int mpnn_loss_smoothed(const float* log_probs, const int64_t* S, const float* mask, int n_nodes,
                                  int n_letters, float weight, float scale, float* loss, float* loss_sum,
                                  float* d_log_probs, float* workspace, int workspace_floats, void* stream) {
  MPNN_CHECK_ARG(log_probs && S && mask && loss && loss_sum && workspace, "mpnn_loss_smoothed: null pointer");
  MPNN_CHECK_ARG(n_nodes > 0 && n_letters >= 1, "mpnn_loss_smoothed: bad sizes");
  const int n_wg = (n_nodes + 255) / 256;
  MPNN_CHECK_ARG(workspace_floats >= n_wg, "mpnn_loss_smoothed: workspace too small (one float per 256 nodes)");
  hipLaunchKernelGGL(k_loss_smoothed, dim3(n_wg), dim3(256), 0, (hipStream_t)stream, log_probs, (const long long*)S, mask,
                     n_nodes, n_letters, weight, scale, loss, workspace, d_log_probs);
  hipLaunchKernelGGL(k_sum_ordered, dim3(1), dim3(1024), 0, (hipStream_t)stream, workspace, n_wg, loss_sum);
  MPNN_LAUNCH_CHECK();
}
