// micro-benchmark (round 3): what one contact row-triple of the Gauss-Seidel sweep costs a lone wave, and why.
//   hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -mllvm -enable-misched=false -ffp-contract=off -I. profiles/r03/ubench_sweep.hip -o /tmp/ubench_sweep && /tmp/ubench_sweep
// Variants: the production sweep (ContactSet::sweeps: LDS records, prefetched one contact ahead); the same arithmetic on a record
// held in registers (no LDS at all); the LDS traffic alone (reads + write, no arithmetic). 1 or 64 active lanes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include "pyflyt_amd/csrc/uav_vehicles.hpp"
using namespace pf;

template <int MODE>
__global__ void __launch_bounds__(64) k(float* out, unsigned long long* cyc, int n, int iters, int active) {
  __shared__ __attribute__((aligned(16))) float ws[64 * 9 * 20];
  const int tid = threadIdx.x;
  if (tid >= active) return;
  ContactSet S;
  const m3 R = rot_from_quat(quat{0.01f * tid, 0.02f, 0.0f, 1.0f});
  S.begin((lds_fptr)ws + tid * 9 * 20, R, v3{0, 0, 0}, 37.0f, v3{0.1f, 0.0f, -1.0f}, v3{0.3f, -0.2f, 0.1f}, 7e4f, 0, 0, 7e4f, 0, 4e4f, 1e-3f, 240.0f, 0.0f);
  for (int c = 0; c < n; ++c) S.add(v3{(c & 1) ? 0.045f : -0.045f, (c & 2) ? 0.045f : -0.045f, -0.01f}, 0.0015f);
  const unsigned long long t0 = __builtin_readcyclecounter();
  if (MODE == 0) {
    S.sweeps(iters, 0.5f);
  } else if (MODE == 1) {  // same arithmetic, the record in registers
    const pf_f4v r0 = S.W4[0], r1 = S.W4[1], r2 = S.W4[2], r3 = S.W4[3];
    pf_f4v r4 = S.W4[4];
    __shared__ pf_f4v sink[64];
    for (int it = 0; it < iters; ++it)
      for (int c = 0; c < n; ++c) {
        S.row3(r0, r1, r2, r3, r4, 0.5f, (lds_f4ptr)&sink[tid]);
        r4.x += 1e-9f;
      }
  } else {  // LDS traffic alone
    pf_f4v acc{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it)
      for (int c = 0; c < n; ++c) {
        lds_f4ptr r = S.W4 + 5 * c;
        acc += r[0] + r[1] + r[2] + r[3] + r[4];
        r[4] = acc;
      }
    S.vc.x += acc.x;
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  out[tid] = S.vc.x + S.vc.y + S.vc.z + S.w.x + S.w.y + S.w.z;
  if (tid == 0) cyc[0] = t1 - t0;
}
template <int MODE> void run(const char* name, int n, int iters, int active) {
  float* d; unsigned long long* c; hipMalloc(&d, 64 * 4); hipMalloc(&c, 8);
  k<MODE><<<1, 64>>>(d, c, n, iters, active); hipDeviceSynchronize();
  k<MODE><<<1, 64>>>(d, c, n, iters, active); hipDeviceSynchronize();
  unsigned long long h; hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
  printf("%-34s contacts %d sweeps %3d active lanes %2d: %8llu cycles = %6.1f per contact row-triple\n", name, n, iters, active, h, (double)h / (n * iters));
  hipFree(d); hipFree(c);
}
int main() {
  for (int active : {1, 64}) {
    run<0>("production sweep (LDS, prefetch)", 4, 100, active);
    run<0>("production sweep (LDS, prefetch)", 8, 100, active);
    run<1>("same arithmetic, registers only", 4, 100, active);
    run<2>("LDS traffic only", 4, 100, active);
  }
  return 0;
}
