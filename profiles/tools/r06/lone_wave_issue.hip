// What a LONE wave (one wave on its SIMD: the 65 536-lane launches of every kernel here) pays per instruction, by kind and by
// dependence. Evidence script, not product: hipcc --offload-arch=gfx950 -O2 lone_wave_issue.hip -o lone_wave_issue && ./lone_wave_issue
// Each body is N copies of a short pattern between two s_memtime reads (shader clocks; s_memrealtime around them for ns); one workgroup of
// one wave, and the same with 1 024 workgroups (one wave on every SIMD of the chip) to see whether neighbours change the figure.
// `lone_wave_issue 200`: every reading is the last of 200 back-to-back launches (clocks ramped). Results: profiles/r06/lone_wave_issue.txt.
// (The patterns are ONE asm statement each: between two `asm volatile` statements of which the second reads what the first wrote the
//  compiler inserts an s_nop -- which is what round 5's ubench/icache_cold.hip timed as a "dependent instruction".)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <algorithm>

#define BODY_KERNEL(NAME, PATTERN, PER)                                                                              \
  __global__ void __launch_bounds__(64, 1) NAME(unsigned long long* out, float seed, float b_, float c_) {                              \
    float a0 = seed, a1 = seed + 1.f, a2 = seed + 2.f, a3 = seed + 3.f, a4 = seed + 4.f, a5 = seed + 5.f,          \
          a6 = seed + 6.f, a7 = seed + 7.f, b = b_, c = c_;                                       \
    double d0 = seed, d1 = seed + 1., d2 = 0.75;                                                               \
    unsigned long long t0, t1, r0, r1;                                                                               \
    asm volatile("s_nop 0" ::: "memory");                                                                            \
    asm volatile("s_memrealtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(r0)::"memory");                                   \
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");                                       \
    asm volatile(".rept " #PER "\n" PATTERN "\n.endr"                                                               \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(d0), "+v"(d1) \
                 : "v"(b), "v"(c), "v"(d2)                                                                           \
                 : "s20", "s21", "s22", "s23", "vcc", "scc", "a0", "a1", "memory");                                  \
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");                                       \
    asm volatile("s_memrealtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(r1)::"memory");                                   \
    float r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)(d0 + d1);                                              \
    if (threadIdx.x == 0) { out[blockIdx.x] = t1 - t0; out[1024 + blockIdx.x] = r1 - r0; }                           \
    if (r == 123.456f) out[blockIdx.x] = 0;                                                                          \
  }

// operands: %0-%7 a0-a7, %8 d0, %9 d1, %10 b, %11 c, %12 d2
// --- plain fp32
BODY_KERNEL(k_fma_dep1, "v_fma_f32 %0, %0, %10, %11", 256)
BODY_KERNEL(k_fma_ind2, "v_fma_f32 %0, %0, %10, %11\n v_fma_f32 %1, %1, %10, %11", 128)
BODY_KERNEL(k_fma_ind4, "v_fma_f32 %0, %0, %10, %11\n v_fma_f32 %1, %1, %10, %11\n v_fma_f32 %2, %2, %10, %11\n v_fma_f32 %3, %3, %10, %11", 64)
BODY_KERNEL(k_fma_ind8, "v_fma_f32 %0, %0, %10, %11\n v_fma_f32 %1, %1, %10, %11\n v_fma_f32 %2, %2, %10, %11\n v_fma_f32 %3, %3, %10, %11\n v_fma_f32 %4, %4, %10, %11\n v_fma_f32 %5, %5, %10, %11\n v_fma_f32 %6, %6, %10, %11\n v_fma_f32 %7, %7, %10, %11", 32)
BODY_KERNEL(k_add_dep1, "v_add_f32 %0, %0, %10", 256)
BODY_KERNEL(k_add_ind2, "v_add_f32 %0, %0, %10\n v_add_f32 %1, %1, %10", 128)
BODY_KERNEL(k_mul_add_dep, "v_mul_f32 %0, %0, %10\n v_add_f32 %0, %0, %11", 128)
BODY_KERNEL(k_fma_dep1_4096, "v_fma_f32 %0, %0, %10, %11", 4096)
BODY_KERNEL(k_fma_ind2_4096, "v_fma_f32 %0, %0, %10, %11\n v_fma_f32 %1, %1, %10, %11", 2048)
BODY_KERNEL(k_fma_ind4_4096, "v_fma_f32 %0, %0, %10, %11\n v_fma_f32 %1, %1, %10, %11\n v_fma_f32 %2, %2, %10, %11\n v_fma_f32 %3, %3, %10, %11", 1024)
BODY_KERNEL(k_fma_dep1_16k, "v_fma_f32 %0, %0, %10, %11", 16384)
BODY_KERNEL(k_fma_ind2_16k, "v_fma_f32 %0, %0, %10, %11\n v_fma_f32 %1, %1, %10, %11", 8192)
// --- packed fp32 (register pairs: a0:a1 is not guaranteed adjacent -- use the doubles' pairs as raw 64-bit registers)
BODY_KERNEL(k_pk_dep1, "v_pk_fma_f32 %8, %8, %12, %12", 256)
BODY_KERNEL(k_pk_ind2, "v_pk_fma_f32 %8, %8, %12, %12\n v_pk_fma_f32 %9, %9, %12, %12", 128)
BODY_KERNEL(k_pk_then_fma_dep, "v_pk_fma_f32 %8, %8, %12, %12\n v_fma_f32 %0, %0, %10, %11", 128)
// --- fp64
BODY_KERNEL(k_f64_dep1, "v_fma_f64 %8, %8, %12, %12", 256)
BODY_KERNEL(k_f64_ind2, "v_fma_f64 %8, %8, %12, %12\n v_fma_f64 %9, %9, %12, %12", 128)
BODY_KERNEL(k_f64_with_f32, "v_fma_f64 %8, %8, %12, %12\n v_fma_f32 %0, %0, %10, %11", 128)
BODY_KERNEL(k_f64_with_3f32, "v_fma_f64 %8, %8, %12, %12\n v_fma_f32 %0, %0, %10, %11\n v_fma_f32 %1, %1, %10, %11\n v_fma_f32 %2, %2, %10, %11", 64)
// --- transcendental (quarter rate)
BODY_KERNEL(k_rcp_dep1, "v_rcp_f32 %0, %0", 256)
BODY_KERNEL(k_rcp_ind2, "v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1", 128)
BODY_KERNEL(k_rcp_then_use, "v_rcp_f32 %0, %0\n v_fma_f32 %0, %0, %10, %11", 128)
BODY_KERNEL(k_rcp_then_ind, "v_rcp_f32 %0, %0\n v_fma_f32 %1, %1, %10, %11", 128)
BODY_KERNEL(k_rcp_then_3ind, "v_rcp_f32 %0, %0\n v_fma_f32 %1, %1, %10, %11\n v_fma_f32 %2, %2, %10, %11\n v_fma_f32 %3, %3, %10, %11", 64)
BODY_KERNEL(k_sin_dep1, "v_sin_f32 %0, %0", 256)
// --- scalar instructions between dependent vector ones
BODY_KERNEL(k_fma_dep_salu, "v_fma_f32 %0, %0, %10, %11\n s_add_u32 s20, s20, 1", 128)
BODY_KERNEL(k_fma_dep_2salu, "v_fma_f32 %0, %0, %10, %11\n s_add_u32 s20, s20, 1\n s_add_u32 s21, s21, 1", 85)
BODY_KERNEL(k_salu_dep1, "s_add_u32 s20, s20, 1", 256)
BODY_KERNEL(k_salu_ind2, "s_add_u32 s20, s20, 1\n s_add_u32 s21, s21, 1", 128)
BODY_KERNEL(k_fma_dep_nop, "v_fma_f32 %0, %0, %10, %11\n s_nop 0", 128)
BODY_KERNEL(k_fma_ind2_salu2, "v_fma_f32 %0, %0, %10, %11\n s_add_u32 s20, s20, 1\n v_fma_f32 %1, %1, %10, %11\n s_add_u32 s21, s21, 1", 64)
// --- compare -> mask -> select (vcc round trip), readlane, accvgpr
BODY_KERNEL(k_cmp_cndmask, "v_cmp_lt_f32 vcc, %0, %10\n v_cndmask_b32 %0, %0, %11, vcc", 128)
BODY_KERNEL(k_readlane_use, "v_readlane_b32 s20, %0, 3\n v_fma_f32 %0, %0, s20, %11", 128)
BODY_KERNEL(k_readfirst_use, "v_readfirstlane_b32 s20, %0\n v_add_f32 %0, s20, %0", 128)
BODY_KERNEL(k_acc_roundtrip, "v_accvgpr_write_b32 a0, %0\n v_accvgpr_read_b32 %0, a0", 128)
BODY_KERNEL(k_acc_read_ind, "v_accvgpr_read_b32 %1, a0\n v_fma_f32 %0, %0, %10, %11", 128)
BODY_KERNEL(k_dpp_dep, "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf", 256)
BODY_KERNEL(k_mov_dep, "v_mov_b32 %0, %0", 256)

struct T { const char* name; void (*k)(unsigned long long*, float, float, float); int n; };
#define E(NAME, n) {#NAME, NAME, n}

int main(int argc, char** argv) {
  const int chain = argc > 1 ? atoi(argv[1]) : 1;
  std::vector<T> tests = {
      E(k_fma_dep1, 256), E(k_fma_dep1_4096, 4096), E(k_fma_ind2_4096, 4096), E(k_fma_ind4_4096, 4096), E(k_fma_dep1_16k, 16384), E(k_fma_ind2_16k, 16384), E(k_fma_ind2, 256), E(k_fma_ind4, 256), E(k_fma_ind8, 256), E(k_add_dep1, 256), E(k_add_ind2, 256),
      E(k_mul_add_dep, 256), E(k_pk_dep1, 256), E(k_pk_ind2, 256), E(k_pk_then_fma_dep, 256), E(k_f64_dep1, 256), E(k_f64_ind2, 256),
      E(k_f64_with_f32, 256), E(k_f64_with_3f32, 256), E(k_rcp_dep1, 256), E(k_rcp_ind2, 256), E(k_rcp_then_use, 256), E(k_rcp_then_ind, 256),
      E(k_rcp_then_3ind, 256), E(k_sin_dep1, 256), E(k_fma_dep_salu, 256), E(k_fma_dep_2salu, 255), E(k_salu_dep1, 256), E(k_salu_ind2, 256),
      E(k_fma_dep_nop, 256), E(k_fma_ind2_salu2, 256), E(k_cmp_cndmask, 256), E(k_readlane_use, 256), E(k_readfirst_use, 256),
      E(k_acc_roundtrip, 256), E(k_acc_read_ind, 256), E(k_dpp_dep, 256), E(k_mov_dep, 256)};
  unsigned long long* d;
  hipMalloc(&d, 2048 * sizeof(unsigned long long));
  std::vector<unsigned long long> h(2048);
  printf("launches back to back per reading: %d\n", chain);
  printf("%-22s %10s %10s   (shader clocks per instruction; n instructions between the two s_memtime)\n", "pattern", "1 wave", "1024 waves");
  for (auto& t : tests) {
    double res[2], ns[2];
    for (int g = 0; g < 2; ++g) {
      const int grid = g ? 1024 : 1;
      unsigned long long best = ~0ull;
      for (int it = 0; it < 5; ++it) {
        // (sustained: the launch that is read is the last of `chain` back-to-back ones -- clocks ramped, every SIMD busy throughout)
        for (int c = 0; c < chain; ++c) hipLaunchKernelGGL(t.k, dim3(grid), dim3(64), 0, 0, d, 1.0f + it, 0.999f, 0.001f);
        hipDeviceSynchronize();
        hipMemcpy(h.data(), d, 2048 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        // median over the workgroups
        std::vector<unsigned long long> v(h.begin(), h.begin() + grid);
        std::sort(v.begin(), v.end());
        if (v[grid / 2] < best) {
          best = v[grid / 2];
          std::vector<unsigned long long> w(h.begin() + 1024, h.begin() + 1024 + grid);
          std::sort(w.begin(), w.end());
          ns[g] = (double)w[grid / 2] * 10.0 / t.n;  // s_memrealtime: 100 MHz
        }
      }
      res[g] = (double)best / t.n;
    }
    printf("%-22s %10.2f %10.2f   %8.2f %8.2f ns\n", t.name, res[0], res[1], ns[0], ns[1]);
  }
  return 0;
}
