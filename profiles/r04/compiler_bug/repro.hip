// Reproducer for the exec-restore placement bug (ROCm 7.2.0 hipcc, gfx950): two explicit instantiations of the product's QuadX
// kernel for two waves per SIMD (256-register budget, the general contact solver's out-of-line call in the tick).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -mllvm -enable-misched=false -ffp-contract=off \
//         --cuda-device-only -S profiles/r04/compiler_bug/repro.hip -o /tmp/repro.s && python tools/isa_exec_check.py /tmp/repro.s
// prints the vector instructions the register allocator placed in front of a join block's `s_or_b64 exec, exec, ...`.
#include <hip/hip_runtime.h>
#include "../../../pyflyt_amd/csrc/quadx_fast.hpp"
template __global__ void pf::quadx_m0_env_kernel<PF_TASK_WAYPOINTS, PF_NOISE_OFF, 64, 2, true, false, false, 2>(
    const pf::QuadK, const pf_buffers, const pf_params*, const int, const uint64_t, const int, const uint8_t*, const int, const uint32_t);
template __global__ void pf::quadx_m0_env_kernel<PF_TASK_HOVER, PF_NOISE_PHILOX, 64, 1, true, false, false, 2>(
    const pf::QuadK, const pf_buffers, const pf_params*, const int, const uint64_t, const int, const uint8_t*, const int, const uint32_t);
