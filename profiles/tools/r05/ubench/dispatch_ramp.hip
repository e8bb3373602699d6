// What does a launch of 1 024 one-wave workgroups cost beyond the life of its waves? Each wave spins on the shader clock for a given
// time (so its life is known), the launch is timed in a graph of back-to-back launches: launch - life = dispatch ramp + launch gap.
// Variants: registers per wave (the allocation granule the dispatcher has to find), LDS per workgroup, kernel-argument bytes,
// waves per workgroup.
//   hipcc --offload-arch=gfx950 -O3 -o dispatch_ramp dispatch_ramp.hip && ./dispatch_ramp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

struct Big { float v[96]; };  // (the env kernels pass ~100 dwords of constants by value)

template <int VGPRS, int LDS_BYTES, int WPB, int BIG = 0>
__global__ void __launch_bounds__(64 * WPB, VGPRS > 256 ? 1 : 2) spin(float* out, long long* stamps, const long long clocks, const Big big, const int use_big) {
  __shared__ float lds[LDS_BYTES > 0 ? LDS_BYTES / 4 : 1];
  float x = threadIdx.x;
  if (BIG) {  // every kernel-argument line is needed before anything else (the env kernels' prologue: eleven s_loads, one wait)
#pragma unroll
    for (int i = 0; i < 96; ++i) x += big.v[i];
  }
  const long long t0 = BIG ? (long long)__builtin_amdgcn_s_memrealtime() + (long long)(x == 12345.0f) : (long long)__builtin_amdgcn_s_memrealtime();
  if (stamps != nullptr && (threadIdx.x & 63) == 0) stamps[blockIdx.x * WPB + (threadIdx.x >> 6)] = t0;
  if (VGPRS >= 128) asm volatile("v_mov_b32 v120, 0" ::: "v120");
  if (VGPRS >= 256) asm volatile("v_mov_b32 v250, 0" ::: "v250");
  if (VGPRS >= 512) asm volatile("v_accvgpr_write_b32 a250, 0" ::: "a250");
  if (LDS_BYTES > 0) lds[threadIdx.x] = x;
  while ((long long)__builtin_amdgcn_s_memrealtime() - t0 < clocks) x = fmaf(x, 0.999f, 0.001f);
  if (use_big) x += big.v[use_big & 63];
  if (LDS_BYTES > 0) x += lds[(threadIdx.x + 1) & 63];
  out[blockIdx.x * 64 * WPB + threadIdx.x] = x;
}

template <int VGPRS, int LDS_BYTES, int WPB, int BIG = 0>
static int run(const char* name, float* out, long long* stamps, hipStream_t s, int waves, float life_us) {
  const int steps = 200;
  const long long clocks = (long long)(life_us * 100.0f);  // s_memrealtime: 100 MHz
  Big big{};
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
  for (int i = 0; i < steps; ++i) hipLaunchKernelGGL((spin<VGPRS, LDS_BYTES, WPB, BIG>), dim3(waves / WPB), dim3(64 * WPB), 0, s, out, i == steps - 1 ? stamps : nullptr, clocks, big, 0);
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
  static long long h[16384];
  CK(hipMemcpy(h, stamps, sizeof(long long) * waves, hipMemcpyDeviceToHost));
  std::sort(h, h + waves);
  printf("%-34s waves %5d life %5.1f us: %6.2f us per launch (+%.2f); wave entries after the first: median %.2f p90 %.2f last %.2f us\n", name, waves, life_us,
         best * 1e3f / steps, best * 1e3f / steps - life_us, (h[waves / 2] - h[0]) * 0.01, (h[waves * 9 / 10] - h[0]) * 0.01, (h[waves - 1] - h[0]) * 0.01);
  CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
  return 0;
}

int main() {
  float* out; CK(hipMalloc(&out, sizeof(float) * 64 * 16384));
  long long* stamps; CK(hipMalloc(&stamps, sizeof(long long) * 16384));
  hipStream_t s; CK(hipStreamCreate(&s));
  for (float life : {0.0f, 4.0f, 8.0f}) {
    for (int waves : {256, 1024}) {
      if (run<64, 0, 1>("64 regs, no LDS", out, stamps, s, waves, life)) return 1;
      if (run<256, 0, 1>("256 regs, no LDS", out, stamps, s, waves, life)) return 1;
      if (run<512, 0, 1>("512 regs, no LDS", out, stamps, s, waves, life)) return 1;
      if (run<512, 16384, 1>("512 regs, 16 KB LDS", out, stamps, s, waves, life)) return 1;
      if (run<256, 16384, 1>("256 regs, 16 KB LDS", out, stamps, s, waves, life)) return 1;
      if (run<256, 16384, 4>("256 regs, 64 KB LDS, 4 waves/WG", out, stamps, s, waves, life)) return 1;
      if (run<512, 16384, 1, 1>("512 regs, 16 KB LDS, 384 B of args", out, stamps, s, waves, life)) return 1;
    }
  }
  return 0;
}
