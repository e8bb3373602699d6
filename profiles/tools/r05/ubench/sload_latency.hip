// What does a scalar load of a kernel argument cost a wave that has to wait for it? 1 024 one-wave workgroups, a 704-byte by-value
// argument block; each wave times, with s_memtime, (a) nothing (the stamp pair alone), (b) the first touch of a line no earlier load
// of the wave has fetched, (c) sixteen dependent load + wait round trips to lines already fetched, (d) values read
// from the lanes of a VGPR (v_readlane_b32) instead. Reports medians over the waves of the last of 50 back-to-back launches.
//   hipcc --offload-arch=gfx950 -O3 -o sload_latency sload_latency.hip && ./sload_latency
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

struct Big { float w[176]; };

#define STAMP(t) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory")
#define LW(off) asm volatile("s_load_dword %0, %1, " #off "\n\ts_waitcnt lgkmcnt(0)\n\ts_add_u32 %2, %2, %0" : "=&s"(x), "+s"(kp), "+s"(acc) :: "memory");

__global__ void __launch_bounds__(64, 1) k(const Big a, unsigned long long* out, float* sink) {
  unsigned long long kp = 0;
#if defined(__HIP_DEVICE_COMPILE__)
  kp = (unsigned long long)__builtin_amdgcn_kernarg_segment_ptr();
#endif
  unsigned long long t0, t1, t2, t3, t4, t5, t6, t7;
  uint32_t x, acc = 0;
  STAMP(t0);
  STAMP(t1);
  LW(0x200)   // a line nothing has touched
  STAMP(t2);
  LW(0x0) LW(0x40) LW(0x80) LW(0xc0)   // four more cold lines, one after the other
  STAMP(t3);
  LW(0x4) LW(0x44) LW(0x84) LW(0xc4) LW(0x8) LW(0x48) LW(0x88) LW(0xc8) LW(0xc) LW(0x4c) LW(0x8c) LW(0xcc)   // twelve round trips to lines this wave has fetched
  STAMP(t4);
  LW(0x204) LW(0x208) LW(0x20c) LW(0x210)   // the line touched once, a while ago
  STAMP(t5);
  float v = a.w[threadIdx.x];  // (one vector load: lane i holds word i)
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(v));
  STAMP(t6);
  float r = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) r += __builtin_amdgcn_readlane(__float_as_int(v), i) * (float)(i + 1);
  asm volatile("" :: "v"(r));
  STAMP(t7);
  if (threadIdx.x == 0) {
    unsigned long long* o = out + (size_t)blockIdx.x * 8;
    o[0] = t1 - t0; o[1] = t2 - t1; o[2] = t3 - t2; o[3] = t4 - t3; o[4] = t5 - t4; o[5] = t6 - t5; o[6] = t7 - t6;
  }
  if (acc == 0x12345u) sink[0] = r;
}

int main() {
  const int waves = 1024;
  unsigned long long* out; CK(hipMalloc(&out, sizeof(unsigned long long) * 8 * waves));
  float* sink; CK(hipMalloc(&sink, 64));
  Big a; for (int i = 0; i < 176; ++i) a.w[i] = 1.0f + i;
  hipStream_t s; CK(hipStreamCreate(&s));
  for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(k, dim3(waves), dim3(64), 0, s, a, out, sink);
  CK(hipStreamSynchronize(s));
  std::vector<unsigned long long> h(8 * waves);
  CK(hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost));
  const char* names[7] = {"stamp pair alone", "first touch of a cold line (+ stamp)", "4 more cold lines, one after the other (+ stamp)",
                          "12 load + wait round trips, lines fetched before (+ stamp)", "4 round trips to the line touched once (+ stamp)",
                          "one vector load of the block, waited for (+ stamp)", "16 v_readlane_b32 + 16 v_fmac (+ stamp)"};
  printf("s_memtime ticks (100 MHz on gfx950 if constant-rate, else shader clocks -- compare with the stamp pair):\n");
  for (int j = 0; j < 7; ++j) {
    std::vector<unsigned long long> v(waves);
    for (int w = 0; w < waves; ++w) v[w] = h[(size_t)w * 8 + j];
    std::sort(v.begin(), v.end());
    printf("  %-52s median %6llu  p10 %6llu  p90 %6llu\n", names[j], v[waves / 2], v[waves / 10], v[waves * 9 / 10]);
  }
  return 0;
}
