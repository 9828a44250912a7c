// Reproducer of the second compiler finding in DESIGN.md section 0b: hipcc (ROCm 7.2, gfx950) puts no wait states
// between a v_mfma_f32_16x16x32_f16 and a v_mfma_f32_16x16x16_f16 that takes its result as the accumulator (either
// order), and the hardware forwards an accumulator only between two MFMAs of ONE form: the second adds to a stale value.
//   /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -o /tmp/mixed tools/ubench/mixed_mfma_repro.hip && /tmp/mixed
// prints, per variant, how many of the 256 outputs differ from the sum of the two products taken in separate kernels
// (operands are small integers: every sum is exact in fp32).  Measured on an MI355X:
//   16x16x32 x3 -> 16x16x16 x3 (plain builtins)   128 of 256 outputs wrong      s_nop 1 between: 128    s_nop 3 (+ the
//   16x16x16 x3 -> 16x16x32 x3 (plain builtins)   124 of 256 outputs wrong      compiler's own s_nop 0): 0   s_nop 7: 0
//   the same chains written as asm with C = D throughout: 125 / 124 wrong up to s_nop 3, 0 with s_nop 7
// Six MFMAs of one form in a row are right (every variant's two halves are), so it is the change of form that needs the
// wait.  tools/hazard_scan.py reports such a pair as an X finding; the library never mixes the forms on one accumulator
// (k_edge_update_bwd_fused pairs an odd edge tile with zeros on the 16x16x32 form instead, at the same cycle count).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

// ORDER 0: 3 x (16x16x32) then 3 x (16x16x16) on one accumulator; ORDER 1: the reverse.  NOPS < 0: nothing between the
// two groups (what the compiler makes of the plain source); NOPS >= 0: s_nop NOPS between them.
template <int ORDER, int NOPS>
__global__ void k_mixed(const h8* a8, const h8* b8, const h4* a4, const h4* b4, f32x4* out) {
  const int l = threadIdx.x;
  h8 x8 = a8[l], y8 = b8[l];
  h4 x4 = a4[l], y4 = b4[l];
  f32x4 acc = out[l];  // (zeros: a loaded accumulator, so that the chain runs in place, C = D, as in a real kernel)
  asm volatile("" : "+v"(x8), "+v"(y8), "+v"(x4), "+v"(y4), "+v"(acc));  // every operand in registers before the chain
  auto wide = [&] {
#pragma unroll
    for (int i = 0; i < 3; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(x8, y8, acc, 0, 0, 0);
  };
  auto narrow = [&] {
#pragma unroll
    for (int i = 0; i < 3; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x16f16(x4, y4, acc, 0, 0, 0);
  };
  if (ORDER == 0) wide(); else narrow();
  if (NOPS >= 0) asm volatile("s_nop %1" : "+v"(acc) : "n"(NOPS));
  if (ORDER == 0) narrow(); else wide();
  out[l] = acc;
}
// the same chain written out with the accumulator in ONE register quad (C = D in every instruction), as the register
// allocator laid it out in the kernel that failed: NOPS < 0 nothing between the forms, else s_nop NOPS
template <int ORDER, int NOPS>
__global__ void k_mixed_asm(const h8* a8, const h8* b8, const h4* a4, const h4* b4, f32x4* out) {
  const int l = threadIdx.x;
  const h8 x8 = a8[l], y8 = b8[l];
  const h4 x4 = a4[l], y4 = b4[l];
  f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#define WIDE3 "v_mfma_f32_16x16x32_f16 %0, %1, %2, %0\n\tv_mfma_f32_16x16x32_f16 %0, %1, %2, %0\n\tv_mfma_f32_16x16x32_f16 %0, %1, %2, %0\n\t"
#define NARROW3 "v_mfma_f32_16x16x16_f16 %0, %3, %4, %0\n\tv_mfma_f32_16x16x16_f16 %0, %3, %4, %0\n\tv_mfma_f32_16x16x16_f16 %0, %3, %4, %0\n\t"
#define TAIL "s_nop 7\n\ts_nop 7\n\ts_nop 7" : "+v"(acc) : "v"(x8), "v"(y8), "v"(x4), "v"(y4), "n"(NOPS < 0 ? 0 : NOPS)
  if (ORDER == 0) {
    if (NOPS < 0) asm volatile("s_nop 7\n\t" WIDE3 NARROW3 TAIL);
    else asm volatile("s_nop 7\n\t" WIDE3 "s_nop %5\n\t" NARROW3 TAIL);
  } else {
    if (NOPS < 0) asm volatile("s_nop 7\n\t" NARROW3 WIDE3 TAIL);
    else asm volatile("s_nop 7\n\t" NARROW3 "s_nop %5\n\t" WIDE3 TAIL);
  }
  out[l] = acc;
}
template <int WIDE>
__global__ void k_one(const h8* a8, const h8* b8, const h4* a4, const h4* b4, f32x4* out) {
  const int l = threadIdx.x;
  f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
  if (WIDE) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8[l], b8[l], acc, 0, 0, 0);
  else acc = __builtin_amdgcn_mfma_f32_16x16x16f16(a4[l], b4[l], acc, 0, 0, 0);
  out[l] = acc;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main() {
  std::vector<_Float16> h(64 * 24);
  srand(7);
  for (auto& v : h) v = (_Float16)(float)(rand() % 7 - 3);
  _Float16* d;
  f32x4* o;
  CK(hipMalloc(&d, h.size() * sizeof(_Float16)));
  CK(hipMalloc(&o, 64 * sizeof(f32x4)));
  CK(hipMemcpy(d, h.data(), h.size() * sizeof(_Float16), hipMemcpyHostToDevice));
  const h8 *a8 = (const h8*)d, *b8 = a8 + 64;
  const h4 *a4 = (const h4*)(b8 + 64), *b4 = a4 + 64;
  std::vector<float> w(256), n(256), r(256);
  k_one<1><<<1, 64>>>(a8, b8, a4, b4, o);
  CK(hipMemcpy(w.data(), o, 1024, hipMemcpyDeviceToHost));
  k_one<0><<<1, 64>>>(a8, b8, a4, b4, o);
  CK(hipMemcpy(n.data(), o, 1024, hipMemcpyDeviceToHost));
  int worst = 0;
  auto run = [&](const char* tag, auto kern) {
    if (hipMemset(o, 0, 1024) != hipSuccess) return;
    kern<<<1, 64>>>(a8, b8, a4, b4, o);
    if (hipMemcpy(r.data(), o, 1024, hipMemcpyDeviceToHost) != hipSuccess) { printf("%s: copy failed\n", tag); return; }
    int bad = 0;
    for (int i = 0; i < 256; ++i) bad += r[i] != 3.f * (w[i] + n[i]);
    printf("%-34s %3d of 256 outputs wrong\n", tag, bad);
    if (bad > worst) worst = bad;
  };
  run("16x16x32 x3 -> 16x16x16 x3", k_mixed<0, -1>);
  run("16x16x32 x3 -> s_nop 0 -> x16 x3", k_mixed<0, 0>);
  run("16x16x32 x3 -> s_nop 1 -> x16 x3", k_mixed<0, 1>);
  run("16x16x32 x3 -> s_nop 3 -> x16 x3", k_mixed<0, 3>);
  run("16x16x32 x3 -> s_nop 7 -> x16 x3", k_mixed<0, 7>);
  run("16x16x16 x3 -> 16x16x32 x3", k_mixed<1, -1>);
  run("16x16x16 x3 -> s_nop 0 -> x32 x3", k_mixed<1, 0>);
  run("16x16x16 x3 -> s_nop 3 -> x32 x3", k_mixed<1, 3>);
  run("16x16x16 x3 -> s_nop 7 -> x32 x3", k_mixed<1, 7>);
  run("asm x32 x3 -> x16 x3, C = D", k_mixed_asm<0, -1>);
  run("asm x32 x3 -> s_nop 0 -> x16 x3", k_mixed_asm<0, 0>);
  run("asm x32 x3 -> s_nop 1 -> x16 x3", k_mixed_asm<0, 1>);
  run("asm x32 x3 -> s_nop 3 -> x16 x3", k_mixed_asm<0, 3>);
  run("asm x32 x3 -> s_nop 7 -> x16 x3", k_mixed_asm<0, 7>);
  run("asm x16 x3 -> x32 x3, C = D", k_mixed_asm<1, -1>);
  run("asm x16 x3 -> s_nop 0 -> x32 x3", k_mixed_asm<1, 0>);
  run("asm x16 x3 -> s_nop 1 -> x32 x3", k_mixed_asm<1, 1>);
  run("asm x16 x3 -> s_nop 3 -> x32 x3", k_mixed_asm<1, 3>);
  run("asm x16 x3 -> s_nop 7 -> x32 x3", k_mixed_asm<1, 7>);
  return 0;
}
