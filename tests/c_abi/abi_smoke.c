/* abi_smoke.c -- drives libpyflyt_amd.so through include/pyflyt_amd.h from plain C (HIP runtime for the
 * device buffers, no Python, no torch): the boundary a foreign-language binding sees.
 *   usage: abi_smoke <params.bin> <n_lanes> <n_steps> <out.bin>
 * params.bin = the raw bytes of a filled pf_params (written by the test from pyflyt_amd/params.py);
 * out.bin receives obs | reward | terminated | truncated of the last step for comparison with the
 * Python path (same seed => bit-identical). */
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/pyflyt_amd.h"

#define CHECK(x) do { int rc_ = (x); if (rc_) { fprintf(stderr, "%s -> %d: %s\n", #x, rc_, pf_last_error(ctx)); return 2; } } while (0)
#define HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s -> %s\n", #x, hipGetErrorString(e_)); return 3; } } while (0)

int main(int argc, char** argv) {
  pf_ctx* ctx = NULL;
  if (argc != 5) return 1;
  if (pf_abi_version() != PF_ABI_VERSION || pf_sizeof_params() != sizeof(pf_params) || pf_sizeof_buffers() != sizeof(pf_buffers)) {
    fprintf(stderr, "header / library mismatch\n");
    return 1;
  }
  pf_params P;
  FILE* f = fopen(argv[1], "rb");
  if (!f || fread(&P, sizeof(P), 1, f) != 1) { fprintf(stderr, "cannot read %s\n", argv[1]); return 1; }
  fclose(f);
  const int n = atoi(argv[2]), steps = atoi(argv[3]);
  CHECK(pf_ctx_create(&P, n, 0, 0, &ctx));
  const int groups = pf_state_groups(ctx), D = pf_obs_dim(ctx);
  pf_buffers b;
  memset(&b, 0, sizeof(b));
  float* actions;
  HIP(hipMalloc((void**)&b.state, sizeof(float) * 4 * (size_t)groups * n));
  HIP(hipMemset(b.state, 0, sizeof(float) * 4 * (size_t)groups * n));
  HIP(hipMalloc((void**)&b.obs, sizeof(float) * (size_t)D * n));
  HIP(hipMalloc((void**)&b.reward, sizeof(float) * n));
  HIP(hipMalloc((void**)&b.terminated, n));
  HIP(hipMalloc((void**)&b.truncated, n));
  HIP(hipMalloc((void**)&actions, sizeof(float) * 4 * n));
  b.actions = actions;
  CHECK(pf_env_reset(ctx, &b, NULL, NULL));
  for (int k = 0; k < steps; ++k) {
    CHECK(pf_sample_actions(ctx, actions, (uint32_t)k, NULL));
    CHECK(pf_env_step(ctx, &b, NULL));
  }
  HIP(hipDeviceSynchronize());
  const size_t nb = sizeof(float) * (size_t)D * n + sizeof(float) * n + 2 * (size_t)n;
  unsigned char* host = (unsigned char*)malloc(nb);
  size_t o = 0;
  HIP(hipMemcpy(host + o, b.obs, sizeof(float) * (size_t)D * n, hipMemcpyDeviceToHost)); o += sizeof(float) * (size_t)D * n;
  HIP(hipMemcpy(host + o, b.reward, sizeof(float) * n, hipMemcpyDeviceToHost)); o += sizeof(float) * n;
  HIP(hipMemcpy(host + o, b.terminated, n, hipMemcpyDeviceToHost)); o += n;
  HIP(hipMemcpy(host + o, b.truncated, n, hipMemcpyDeviceToHost)); o += n;
  f = fopen(argv[4], "wb");
  if (!f || fwrite(host, 1, nb, f) != nb) return 1;
  fclose(f);
  printf("abi_smoke ok: %d lanes, %d steps, obs_dim %d, %d state groups\n", n, steps, D, groups);
  pf_ctx_destroy(ctx);
  return 0;
}
