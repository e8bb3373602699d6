// Does a kernel that merely RESERVES scratch (private segment > 0, touched on a never-taken path) dispatch more slowly than the same
// kernel without? 1024 workgroups of one wave, ~8 us of dependent FMAs per wave (the shape of the env kernels), timed in a graph.
//   hipcc --offload-arch=gfx950 -O3 -o scratch_dispatch scratch_dispatch.hip && ./scratch_dispatch
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int SCRATCH_WORDS>
__global__ void __launch_bounds__(64) work(float* out, const int iters, const int never) {
  float x = threadIdx.x * 1e-3f + blockIdx.x;
  for (int i = 0; i < iters; ++i) x = fmaf(x, 0.999f, 0.001f);
  if (SCRATCH_WORDS > 0) {
    volatile float spill[SCRATCH_WORDS > 0 ? SCRATCH_WORDS : 1];
    if (never) {  // (never taken: the reservation is what is measured)
      for (int i = 0; i < SCRATCH_WORDS; ++i) spill[i] = x + i;
      x = spill[never % SCRATCH_WORDS];
    }
  }
  out[blockIdx.x * 64 + threadIdx.x] = x;
}

template <int W>
static int run(const char* name, float* out, hipStream_t s, int grid) {
  const int steps = 200;
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
  for (int i = 0; i < steps; ++i) hipLaunchKernelGGL(work<W>, dim3(grid), dim3(64), 0, s, out, 2600, 0);
  CK(hipStreamEndCapture(s, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  float best = 1e9f;
  for (int r = 0; r < 5; ++r) {
    CK(hipEventRecord(a, s)); CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(b, s)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    if (ms < best) best = ms;
  }
  printf("%-28s grid %5d: %.2f us per launch\n", name, grid, best * 1e3f / steps);
  return 0;
}

int main() {
  float* out; CK(hipMalloc(&out, sizeof(float) * 64 * 8192));
  hipStream_t s; CK(hipStreamCreate(&s));
  for (int grid : {1024, 4096}) {
    if (run<0>("no scratch", out, s, grid)) return 1;
    if (run<6>("24 B of scratch reserved", out, s, grid)) return 1;
    if (run<64>("256 B of scratch reserved", out, s, grid)) return 1;
    if (run<0>("no scratch (again)", out, s, grid)) return 1;
  }
  return 0;
}
