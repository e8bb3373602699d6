// Does a wave that runs a long stretch of straight-line code ONCE pay for fetching it? 1 024 one-wave workgroups (one per SIMD),
// each executing N dependent v_fma_f32 (8 bytes each) laid out as straight-line code (N x 8 B of instructions) or as a loop over a
// 64-instruction body (512 B: always in the instruction cache); timed in a graph of back-to-back launches.
//   issue floor: N x 4 clocks (a wave64 VALU instruction on a 16-lane SIMD); dependent issue measured at ~6.5 clocks on a lone wave
//   hipcc --offload-arch=gfx950 -O3 -o icache_cold icache_cold.hip && ./icache_cold
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

#define F1 asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
#define F4 F1 F1 F1 F1
#define F16 F4 F4 F4 F4
#define F64 F16 F16 F16 F16
#define F256 F64 F64 F64 F64
#define F1024 F256 F256 F256 F256
// two independent chains interleaved: issue every 4 clocks if nothing else is in the way
#define G1 asm volatile("v_fma_f32 %0, %0, %2, %3\n\tv_fma_f32 %1, %1, %2, %3" : "+v"(x), "+v"(y) : "v"(a), "v"(b));
#define G4 G1 G1 G1 G1
#define G16 G4 G4 G4 G4
#define G64 G16 G16 G16 G16
#define G256 G64 G64 G64 G64
#define G512 G256 G256

template <int KIND>
__global__ void __launch_bounds__(64, 1) code(float* out, const float a, const float b, const int trips) {
  float x = threadIdx.x, y = blockIdx.x;
  if (KIND == 0) { F1024 F1024 F1024 F1024 }                          // 4 096 dependent, straight line: 32 KB
  if (KIND == 1) { for (int i = 0; i < trips; ++i) { F64 } }          // 4 096 dependent, a 512-byte loop body (trips = 64)
  if (KIND == 2) { G512 G512 G512 G512 }                              // 2 x 2 048 independent pairs, straight line: 32 KB
  if (KIND == 3) { for (int i = 0; i < trips; ++i) { G16 G16 } }      // the same in a 512-byte loop body (trips = 64)
  if (KIND == 4) { F1024 F1024 }                                      // 2 048 dependent, straight line: 16 KB
  if (KIND == 5) { for (int i = 0; i < trips / 2; ++i) { F64 } }      // 2 048 dependent, loop
  out[blockIdx.x * 64 + threadIdx.x] = x + y;
}

template <int KIND>
static int run(const char* name, float* out, hipStream_t s, int waves) {
  const int steps = 200;
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
  for (int i = 0; i < steps; ++i) hipLaunchKernelGGL((code<KIND>), dim3(waves), dim3(64), 0, s, out, 0.999f, 0.001f, 64);
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
  printf("%-52s waves %5d: %6.2f us per launch\n", name, waves, best * 1e3f / steps);
  CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
  return 0;
}

int main() {
  float* out; CK(hipMalloc(&out, sizeof(float) * 64 * 16384));
  hipStream_t s; CK(hipStreamCreate(&s));
  for (int waves : {256, 1024}) {
    if (run<0>("4096 dependent fma, straight line (32 KB)", out, s, waves)) return 1;
    if (run<1>("4096 dependent fma, 512 B loop", out, s, waves)) return 1;
    if (run<2>("2 x 2048 independent fma, straight line (32 KB)", out, s, waves)) return 1;
    if (run<3>("2 x 2048 independent fma, 512 B loop", out, s, waves)) return 1;
    if (run<4>("2048 dependent fma, straight line (16 KB)", out, s, waves)) return 1;
    if (run<5>("2048 dependent fma, 512 B loop", out, s, waves)) return 1;
  }
  return 0;
}
