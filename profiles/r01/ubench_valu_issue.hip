// micro-benchmark: issue cost of v_fma_f32 vs v_pk_fma_f32 vs v_mul/v_cndmask/v_rcp on gfx950 (one wave per SIMD)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float float2v __attribute__((ext_vector_type(2)));
template <int MODE> __global__ void k(float* out, int iters, float a, float b) {
  float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
  float2v p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7};
  float2v av = {a, a}, bv = {b, b};
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) {  // 8 independent scalar fma
      x0 = fmaf(x0, a, b); x1 = fmaf(x1, a, b); x2 = fmaf(x2, a, b); x3 = fmaf(x3, a, b);
      x4 = fmaf(x4, a, b); x5 = fmaf(x5, a, b); x6 = fmaf(x6, a, b); x7 = fmaf(x7, a, b);
    } else if (MODE == 1) {  // 4 independent packed fma (same 8 flop-pairs)
      p0 = __builtin_elementwise_fma(p0, av, bv); p1 = __builtin_elementwise_fma(p1, av, bv);
      p2 = __builtin_elementwise_fma(p2, av, bv); p3 = __builtin_elementwise_fma(p3, av, bv);
    } else if (MODE == 2) {  // 1 dependent chain scalar
      x0 = fmaf(x0, a, b); x0 = fmaf(x0, a, b); x0 = fmaf(x0, a, b); x0 = fmaf(x0, a, b);
      x0 = fmaf(x0, a, b); x0 = fmaf(x0, a, b); x0 = fmaf(x0, a, b); x0 = fmaf(x0, a, b);
    } else if (MODE == 3) {  // 8 rcp
      x0 = __builtin_amdgcn_rcpf(x0); x1 = __builtin_amdgcn_rcpf(x1); x2 = __builtin_amdgcn_rcpf(x2); x3 = __builtin_amdgcn_rcpf(x3);
      x4 = __builtin_amdgcn_rcpf(x4); x5 = __builtin_amdgcn_rcpf(x5); x6 = __builtin_amdgcn_rcpf(x6); x7 = __builtin_amdgcn_rcpf(x7);
    } else if (MODE == 4) {  // 8 med3
      x0 = __builtin_amdgcn_fmed3f(x0, a, b); x1 = __builtin_amdgcn_fmed3f(x1, a, b); x2 = __builtin_amdgcn_fmed3f(x2, a, b); x3 = __builtin_amdgcn_fmed3f(x3, a, b);
      x4 = __builtin_amdgcn_fmed3f(x4, a, b); x5 = __builtin_amdgcn_fmed3f(x5, a, b); x6 = __builtin_amdgcn_fmed3f(x6, a, b); x7 = __builtin_amdgcn_fmed3f(x7, a, b);
    }
    asm volatile("" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
    asm volatile("" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3));
  }
  out[blockIdx.x * 64 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
}
template <int MODE> void run(const char* name, int blocks, float* d) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000;
  k<MODE><<<blocks, 64>>>(d, 100, 1.0001f, 0.5f);
  hipEventRecord(e0); k<MODE><<<blocks, 64>>>(d, iters, 1.0001f, 0.5f); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double ns_per_iter = ms * 1e6 / iters;
  printf("%-28s blocks=%5d  %.2f ns per 8-op iteration  (= %.1f cycles @2.4GHz, %.2f cycles/op)\n", name, blocks, ns_per_iter, ns_per_iter * 2.4, ns_per_iter * 2.4 / 8);
}
int main() {
  float* d; hipMalloc(&d, 64 * 8192 * 4);
  for (int blocks : {1024, 4096}) {
    run<0>("8 x v_fma_f32 (indep)", blocks, d);
    run<1>("4 x v_pk_fma_f32 (indep)", blocks, d);
    run<2>("8 x v_fma_f32 (dependent)", blocks, d);
    run<3>("8 x v_rcp_f32", blocks, d);
    run<4>("8 x v_med3_f32", blocks, d);
  }
  return 0;
}
