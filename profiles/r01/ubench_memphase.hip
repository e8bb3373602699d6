// memphase.hip -- how long do the memory phases of one env-step launch take on their own?
// Kernel A: per lane load 7 state float4 + 1 action float4, touch them, store 7 state float4 + 21 obs floats
// + reward + 2 flags (the hover step's traffic: 330 B/lane), no physics. Kernel B: the same with ~N dependent FMAs
// per lane between load and store (a stand-in compute phase). 100 launches per hipGraph, 65 536 and 4 096 lanes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int FMAS>
__global__ void __launch_bounds__(64) step_like(float4* __restrict__ state, const float4* __restrict__ act, float* __restrict__ obs,
                                                float* __restrict__ rew, unsigned char* __restrict__ t, unsigned char* __restrict__ u, int n) {
  const int i = blockIdx.x * 64 + threadIdx.x;
  if (i >= n) return;
  float4 g[7];
#pragma unroll
  for (int k = 0; k < 7; ++k) g[k] = state[(size_t)k * n + i];
  float4 a = act[i];
  float x = g[0].x + a.x, y = g[1].y + a.y;
#pragma unroll 8
  for (int k = 0; k < FMAS; ++k) { x = fmaf(x, 0.999f, y); y = fmaf(y, 1.001f, -x * 1e-3f); }
#pragma unroll
  for (int k = 0; k < 7; ++k) { g[k].x += x * 1e-9f; g[k].w += y * 1e-9f; state[(size_t)k * n + i] = g[k]; }
#pragma unroll
  for (int k = 0; k < 21; ++k) __builtin_nontemporal_store(g[k % 7].y + k, &obs[(size_t)k * n + i]);  // coalesced: bytes only, the real kernel transposes through LDS
  rew[i] = x; t[i] = 0; u[i] = y > 1e30f;
}

template <int FMAS>
static int run(int n, const char* name) {
  float4 *state, *act; float *obs, *rew; unsigned char *t, *u;
  CK(hipMalloc(&state, sizeof(float4) * 7 * n)); CK(hipMemset(state, 0, sizeof(float4) * 7 * n));
  CK(hipMalloc(&act, sizeof(float4) * n)); CK(hipMemset(act, 0, sizeof(float4) * n));
  CK(hipMalloc(&obs, sizeof(float) * 21 * n)); CK(hipMalloc(&rew, 4 * n)); CK(hipMalloc(&t, n)); CK(hipMalloc(&u, n));
  hipStream_t s; CK(hipStreamCreate(&s));
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
  for (int k = 0; k < 100; ++k) hipLaunchKernelGGL(step_like<FMAS>, dim3((n + 63) / 64), dim3(64), 0, s, state, act, obs, rew, t, u, n);
  CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  for (int k = 0; k < 3; ++k) CK(hipGraphLaunch(ge, s));
  CK(hipStreamSynchronize(s));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0, s));
  for (int k = 0; k < 20; ++k) CK(hipGraphLaunch(ge, s));
  CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  printf("%-28s n=%7d: %6.2f us per launch\n", name, n, ms * 1e3f / 2000.f);
  return 0;
}
int main() {
  for (int n : {4096, 65536, 524288}) {
    run<0>(n, "memory only");
    run<1200>(n, "memory + 2400 dependent FMAs");
    run<2400>(n, "memory + 4800 dependent FMAs");
  }
  return 0;
}
