// Test probe (not product): the cascaded controller's double-precision helpers (pyflyt_amd/csrc/quadx_control_d.hpp: rcp_d, sqrt_pos_d,
// atan2_d with the unit vector it hands back) evaluated on the device for a list of arguments. tests/test_gpu_fp64_math.py compiles
// this file with hipcc on the GPU box, calls probe() through ctypes and compares with numpy's float64 functions.
#include <hip/hip_runtime.h>
#include "../../pyflyt_amd/csrc/quadx_control_d.hpp"

__global__ void probe_kernel(const double* y, const double* x, double* at, double* cs, double* sn, double* rc, double* sq, double* as, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double c, s;
  at[i] = pf::atan2_d(y[i], x[i], &c, &s);
  cs[i] = c; sn[i] = s;
  rc[i] = pf::rcp_d(x[i]);
  sq[i] = pf::sqrt_pos_d(__builtin_fabs(x[i]));
  // asin the way quad_ctl_inputs_d takes it (|s| < 0.99999): s = y / hypot(x, y) is some value in (-1, 1)
  const double h = y[i] / __builtin_sqrt(x[i] * x[i] + y[i] * y[i] + 1e-300), hs = h > 0.99998 ? 0.99998 : (h < -0.99998 ? -0.99998 : h);
  as[i] = pf::atan2_d(hs, pf::sqrt_pos_d((1.0 - hs) * (1.0 + hs)));
  sn[i] = s;
}

extern "C" int probe(const double* y, const double* x, double* at, double* cs, double* sn, double* rc, double* sq, double* as, int n) {
  double* d[8];
  const size_t b = (size_t)n * sizeof(double);
  for (int k = 0; k < 8; ++k)
    if (hipMalloc(&d[k], b) != hipSuccess) return 1;
  if (hipMemcpy(d[0], y, b, hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(d[1], x, b, hipMemcpyHostToDevice) != hipSuccess) return 2;
  hipLaunchKernelGGL(probe_kernel, dim3((n + 255) / 256), dim3(256), 0, 0, d[0], d[1], d[2], d[3], d[4], d[5], d[6], d[7], n);
  if (hipDeviceSynchronize() != hipSuccess) return 3;
  double* out[6] = {at, cs, sn, rc, sq, as};
  for (int k = 0; k < 6; ++k)
    if (hipMemcpy(out[k], d[2 + k], b, hipMemcpyDeviceToHost) != hipSuccess) return 4;
  for (int k = 0; k < 8; ++k) (void)hipFree(d[k]);
  return 0;
}
