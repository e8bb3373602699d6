// pyflyt_amd.hip -- kernels + C ABI (include/pyflyt_amd.h) of the MI355X-native batched UAV step.
//
// Execution model: one wavefront lane per drone, one 64-lane wavefront per workgroup (no
// inter-wave synchronisation anywhere), persistent state as float4 groups [group][lane][4] so that
// every state access is a 16 B/lane, 1 KiB/wave coalesced global_load/store_dwordx4. The whole env
// step (env_step_ratio x ticks_per_control physics ticks, controller, reward, termination,
// auto-reset with its settle ticks) stays in registers; the row-major observation tile is
// transposed through LDS so that the [n][obs_dim] output is written with full-width stores.
// No MFMA: there is no dense contraction on this path (3x3 work only).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <type_traits>

#include "../../include/pyflyt_amd.h"
#include "uav_vehicles.hpp"
#include "quadx_fast.hpp"
#include "fixedwing_fast.hpp"
#include "dogfight.hpp"
#include "rocket.hpp"

namespace pf {

constexpr int kWave = 64;
constexpr int kMaxObs = 40;  // 13 + 4 + 6 + 3*4 = 35 (Fixedwing), 13 + 4 + 4 + 4*4 = 37 (QuadX with yaw targets)

enum { OP_STEP = 0, OP_RESET = 1 };
// Aviary-level kernels (where bodies land and stay landed): 30 KB of LDS for the contact solve -- every lane of a wave of
// quadrotors (the incident face's 4 vertices + the sentinel, 24 floats each) in one round, six worst-case airframes (48
// vertices) side by side
constexpr int kAviaryContactFloats = 64 * 5 * kContactWords;

// ------------------------------------------------------------------ per-task side block
// 12 floats per lane in state groups G_TGT..G_TGT+2:
//   waypoint tasks : the 4 x 3 target positions (waypoint_handler.py:70-83)
//   MA hover       : spawn position (3), spawn quaternion (4), the action of the previous call (4)
struct SideBlock {
  float t[4][3];
  float yaw[4];  // QuadX-Waypoints yaw targets (waypoint_handler.py:85-89), state group G_TGT + 3
  int n_left;
  PF_DEV void load(const float4* S, size_t n, size_t i, int g) {
    float4 a = S[(size_t)(g + 0) * n + i], b = S[(size_t)(g + 1) * n + i], c = S[(size_t)(g + 2) * n + i];
    t[0][0] = a.x; t[0][1] = a.y; t[0][2] = a.z; t[1][0] = a.w;
    t[1][1] = b.x; t[1][2] = b.y; t[2][0] = b.z; t[2][1] = b.w;
    t[2][2] = c.x; t[3][0] = c.y; t[3][1] = c.z; t[3][2] = c.w;
  }
  PF_DEV void store(float4* S, size_t n, size_t i, int g) const {
    S[(size_t)(g + 0) * n + i] = float4{t[0][0], t[0][1], t[0][2], t[1][0]};
    S[(size_t)(g + 1) * n + i] = float4{t[1][1], t[1][2], t[2][0], t[2][1]};
    S[(size_t)(g + 2) * n + i] = float4{t[2][2], t[3][0], t[3][1], t[3][2]};
  }
  PF_DEV void pop() {  // waypoint_handler.py:181-188
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
      for (int c = 0; c < 3; ++c) t[k][c] = t[k + 1][c];
    yaw[0] = yaw[1]; yaw[1] = yaw[2]; yaw[2] = yaw[3];
    n_left -= 1;
  }
};
PF_DEV float wrap_pi(float e) {  // waypoint_handler.py:147-149
  e = e > kPi ? e - 2.0f * kPi : e;
  return e < -kPi ? e + 2.0f * kPi : e;
}

// One Aviary.step out of line: used only by the rarely taken SAME_STEP settle loop so that the hot
// loop below keeps the single inlined copy.
template <class VEH, int MODE_T>
__device__ __noinline__ void aviary_step_outlined(VEH* V, const pf_params* P, const float* sp, Noise* nz, int flat_base) {
  V->template aviary_step<MODE_T>(*P, sp, *nz, flat_base);
}

// ------------------------------------------------------------------ the generic fused env kernel
// Every (vehicle, task, flight mode) the specialised QuadX mode-0 kernel (quadx_fast.hpp) does not
// cover. Flat control flow: reset preamble, ONE loop of Aviary steps with a single per-lane
// predicate (stepping lanes run env_step_ratio iterations with the env logic, lanes that are being
// reset run their settle iterations through the same loop body), epilogue.
// `tmpl`: a settled spawn state (GROUPS float4s) computed once per context when the settle phase
// cannot depend on the lane (Fixedwing: the settle throttle command is 0, so motor noise has
// nothing to scale; any vehicle with noise off) -- a reset is then a copy.
// roll_steps > 0: pf_rollout -- that many env steps in this one launch, the lane's state resident in registers between them (what a
// relaunch would re-derive from the stored groups is re-derived by VEH::relaunch, so the trajectory is the one of roll_steps x
// (pf_sample_actions + pf_env_step), bit for bit); step k's actions are B.actions[k] or drawn here with pf_sample_actions' keys
// (step index step0 + k), its outputs go to slot k of the trajectory buffers. 0: one step (pf_env_step / pf_env_reset).
template <class VEH, int TASK, int MODE_T>
__global__ void __launch_bounds__(kWave) env_kernel(const pf_params P, const pf_buffers B, const int n,
                                                    const uint64_t lane0, const int op, const uint8_t* mask,
                                                    const float4* __restrict__ tmpl, const pf_params* __restrict__ Pdev,
                                                    const int roll_steps, const uint32_t step0) {
  __shared__ __attribute__((aligned(16))) float tile[kWave * kMaxObs];
  __shared__ __attribute__((aligned(16))) float ktab[VEH::TABLE_FLOATS];
  __shared__ float wpose[kWave * 8];  // shared worlds: each lane's pose and contact bit, exchanged once per tick
  __shared__ float wvel[TASK == PF_TASK_MA_HOVER ? kWave * kPairVelStride : 1];  // ... and the new velocities for the pair stage
  const int tid = threadIdx.x;
  VEH::fill_table(ktab, Pdev, tid);
  __syncthreads();
  const int wave_base = blockIdx.x * kWave;
  const int lane = wave_base + tid;
  const bool valid = lane < n;
  const size_t li = valid ? lane : n - 1;
  const size_t N = (size_t)n;
  const float4* Sin = reinterpret_cast<const float4*>(B.state);
  float4* Sout = reinterpret_cast<float4*>(B.state);
  const int mode = (MODE_T == kRuntimeMode) ? P.flight_mode : MODE_T;
  constexpr bool kSide = (TASK == PF_TASK_WAYPOINTS || TASK == PF_TASK_MA_HOVER);

  static_assert(kWave * kMaxObs >= kContactSlotFloats, "the contact solver's LDS regions alias the observation tile: at least one worst-case region");
  VEH V;
  V.b.pdev = Pdev;
  V.b.cws = (lds_fptr)tile;  // (idle during the physics ticks)
  V.b.contact_regions(P, kWave * kMaxObs);
  V.bind(ktab);
  if (TASK == PF_TASK_MA_HOVER && P.agents_per_world > 1) { V.b.wpose_ = wpose; V.b.wvel_ = wvel; V.b.wtid = tid; V.b.wA = P.agents_per_world; }
  SideBlock tg;
  float new_dist;
  int4 ints;
  V.load(Sin, N, li, mode, new_dist, ints);
  if (kSide) tg.load(Sin, N, li, VEH::G_TGT);
  const bool kYaw = (TASK == PF_TASK_WAYPOINTS) && P.use_yaw_targets != 0;
  tg.yaw[0] = tg.yaw[1] = tg.yaw[2] = tg.yaw[3] = 0.0f;
  float yaw_err0 = 0.0f;  // waypoint_handler.py:156
  if (kYaw) { float4 y = Sin[(size_t)(VEH::G_TGT + 3) * N + li]; tg.yaw[0] = y.x; tg.yaw[1] = y.y; tg.yaw[2] = y.z; tg.yaw[3] = y.w; }
  float4 ma_past = float4{0.f, 0.f, 0.f, 0.f};  // MA hover: self.past_actions (ma_quadx_base_env.py:326)
  if (TASK == PF_TASK_MA_HOVER) ma_past = Sin[(size_t)(VEH::G_TGT + 3) * N + li];
  int step_count = ints.x, flags = ints.y;
  uint32_t rng_ctr = (uint32_t)ints.z;
  // QuadX Hover / Waypoints: a reset's draws are keyed by the event counter at the lane's PREVIOUS reset (oracle/uav_oracle.c:
  // orc_env_reset; quadx_fast.hpp: QuadSpare). The key word lives in group 7's fourth word (flight modes -1, 0) or, where groups
  // 7-11 hold the cascade's memories, in group 11's third; this kernel prepares nothing ahead: it writes the word back with the
  // "spare valid" bit clear.
  constexpr bool REKEY = std::is_same<VEH, QuadX>::value && (TASK == PF_TASK_HOVER || TASK == PF_TASK_WAYPOINTS);
  const bool key_in11 = REKEY && mode > 0;
  uint32_t reset_kw = 0u;
  bool reset_kw_dirty = false;
  if (REKEY) {
    const float4 gk = Sin[(size_t)(key_in11 ? 11 : 7) * N + li];
    reset_kw = (uint32_t)__float_as_int(key_in11 ? gk.z : gk.w) & 0x7fffffffu;
  }
  tg.n_left = ints.w;
  bool term = (flags & PF_F_TERMINATED) != 0, trunc = (flags & PF_F_TRUNCATED) != 0;
  float old_dist = new_dist;

  Noise nz;
  nz.mode = P.noise_mode; nz.n = n; nz.lane = (int)li;
  nz.k0 = (uint32_t)P.seed; nz.k1 = (uint32_t)(P.seed >> 32);
  nz.c0 = (uint32_t)(lane0 + li); nz.nmot = (float)P.n_motors; nz.cached = -1; nz.xi = nullptr;

  bool active = false, do_reset = false, wave_all = false;  // (per env step: set at the top of the step loop below)

  float sp[6] = {0, 0, 0, 0, 0, 0};
  float act4[4] = {0, 0, 0, 0};   // action slots of the observation
  float reward = 0.0f;
  bool pop_pending = false;
  bool rpy_valid = false;
  const int D = (P.angle_repr ? 13 : 12) + 4 + VEH::AUX +
                (TASK == PF_TASK_WAYPOINTS ? (kYaw ? 4 : 3) * P.num_targets : (TASK == PF_TASK_MA_HOVER ? 3 : 0));

  // env.reset() up to (not including) the settle phase: quadx_base_env.py:149-206,
  // ma_quadx_base_env.py:206-241. Returns with `sp` = the mode's default setpoint.
  auto begin_reset = [&]() {
    if (tmpl != nullptr) {
      float nd_;
      int4 i_;
      V.load(tmpl, 1, 0, 7, nd_, i_);  // settled spawn state incl. controller memories (mode 7 == every group)
      V.b.rpy = euler_from_quat_fast(V.b.q);
    } else if (TASK == PF_TASK_MA_HOVER) {
      const float pose[7] = {tg.t[0][0], tg.t[0][1], tg.t[0][2], tg.t[1][0], tg.t[1][1], tg.t[1][2], tg.t[2][0]};
      V.reset(P, pose, sp);
      V.set_mode(mode, sp);
    } else {
      V.reset(P, nullptr, sp);
      V.set_mode(mode, sp);
    }
    rpy_valid = true;
    step_count = 0; term = false; trunc = false; flags = 0; pop_pending = false;
    if (TASK != PF_TASK_MA_HOVER) act4[0] = act4[1] = act4[2] = act4[3] = 0.0f;
    nz.begin_event(REKEY ? reset_kw : rng_ctr, 1u, B.xi_reset);
    if (REKEY) { reset_kw = (rng_ctr + 1u) & 0x7fffffffu; reset_kw_dirty = true; }  // (the NEXT reset's key: the counter as this reset leaves it -- strictly increasing)
    if (TASK == PF_TASK_WAYPOINTS) {  // waypoint_handler.py:53-83
      const int nt = P.num_targets;
      tg.n_left = nt;
      new_dist = INFINITY; old_dist = INFINITY;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (i < nt) {
          float theta, phi, dist;
          if (P.noise_mode == PF_NOISE_INJECT && B.u_targets != nullptr) {
            theta = B.u_targets[(size_t)i * N + li];
            phi = B.u_targets[(size_t)(nt + i) * N + li];
            dist = B.u_targets[(size_t)(2 * nt + i) * N + li];
          } else {
            theta = (2.0f * kPi) * nz.uniform(i, 2u);
            phi = (2.0f * kPi) * nz.uniform(nt + i, 2u);
            dist = fmaf(P.dome * 0.9f - 1.0f, nz.uniform(2 * nt + i, 2u), 1.0f);
          }
          float st, ct, sph, cph;
          sincosf(theta, &st, &ct);
          sincosf(phi, &sph, &cph);
          float zz = __builtin_fabsf(dist * cph);
          tg.t[i][0] = dist * sph * ct; tg.t[i][1] = dist * sph * st; tg.t[i][2] = zz > P.min_height ? zz : P.min_height;
        }
      }
      if (kYaw) {  // waypoint_handler.py:85-89: uniform(-pi, pi), drawn after all the positions
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (i < nt)
            tg.yaw[i] = (P.noise_mode == PF_NOISE_INJECT && B.u_targets != nullptr) ? B.u_targets[(size_t)(3 * nt + i) * N + li]
                                                                                    : fmaf(2.0f * kPi, nz.uniform(3 * nt + i, 2u), -kPi);
      }
    }
  };
  // compute_state's waypoint bookkeeping (waypoint_handler.py:135-142); ||R^T d|| = ||d||
  auto wp_distance = [&]() {
    if (TASK != PF_TASK_WAYPOINTS) return;
    if (pop_pending) { tg.pop(); pop_pending = false; }
    float dx = tg.t[0][0] - V.b.p.x, dy = tg.t[0][1] - V.b.p.y, dz = tg.t[0][2] - V.b.p.z;
    old_dist = new_dist;
    new_dist = sqrtf(fmaf(dx, dx, fmaf(dy, dy, dz * dz)));
    if (kYaw) yaw_err0 = __builtin_fabsf(wrap_pi(tg.yaw[0] - V.b.rpy.z));
  };
  // compute_term_trunc_reward: quadx_base_env.py:251-267, quadx_hover_env.py:117-138,
  // quadx_waypoints_env.py:177-204, fixedwing_waypoints_env.py:169-190, ma_quadx_hover_env.py:168-205
  auto term_trunc_reward = [&]() {
    if (step_count > P.max_steps) trunc = true;
    if (TASK == PF_TASK_MA_HOVER) {
      if (V.b.contact_step) { reward -= 100.0f; flags |= PF_F_INFO_COLLISION; term = true; }
      if (sqrtf(dot(V.b.p, V.b.p)) > P.dome) { reward -= 100.0f; flags |= PF_F_INFO_OOB; term = true; }
      if (!P.sparse_reward) {
        v3 d{V.b.p.x - tg.t[0][0], V.b.p.y - tg.t[0][1], V.b.p.z - tg.t[0][2]};
        float lin = sqrtf(dot(d, d));
        float ang = sqrtf(fmaf(V.b.rpy.x, V.b.rpy.x, V.b.rpy.y * V.b.rpy.y));
        reward -= lin + ang * 0.1f;
        reward += 1.0f;
      }
      return;
    }
    if (V.b.contact_step) { reward = -100.0f; flags |= PF_F_INFO_COLLISION; term = true; }
    if (sqrtf(dot(V.b.p, V.b.p)) > P.dome) { reward = -100.0f; flags |= PF_F_INFO_OOB; term = true; }
    if (TASK == PF_TASK_HOVER) {
      if (!P.sparse_reward) {
        v3 d{V.b.p.x, V.b.p.y, V.b.p.z - 1.0f};
        float lin = sqrtf(dot(d, d));
        float yaw_rate = __builtin_fabsf(V.b.wb.z);
        reward -= 0.01f * (yaw_rate * yaw_rate);
        float ang = sqrtf(fmaf(V.b.rpy.x, V.b.rpy.x, V.b.rpy.y * V.b.rpy.y));
        reward -= lin + ang;
        reward += 1.0f;
      }
    } else if (TASK == PF_TASK_WAYPOINTS) {
      if (!P.sparse_reward) {
        float progress = (isinf(old_dist + new_dist)) ? 0.0f : old_dist - new_dist;
        reward += __builtin_fmaxf(3.0f * progress, 0.0f);
        reward += P.wp_dist_reward / new_dist;
        if (P.wp_yaw_penalty != 0.0f) {
          float yaw_rate = __builtin_fabsf(V.b.wb.z);
          reward -= P.wp_yaw_penalty * (yaw_rate * yaw_rate);
        }
      }
      if (new_dist < P.goal_reach_distance && (!kYaw || yaw_err0 < P.goal_reach_angle)) {  // waypoint_handler.py:167-179
        reward = 100.0f;
        pop_pending = true;  // the observation of this step still shows the reached target
        if ((tg.n_left - 1) == 0) { trunc = true; flags |= PF_F_INFO_COMPLETE; }
      }
    }
  };
  // flattened observation row of this lane -> LDS tile (Appendix A of SURVEY.md)
  auto write_obs_row = [&]() {
    if (!rpy_valid) { V.b.rpy = euler_from_quat_fast(V.b.q); rpy_valid = true; }
    float* row = tile + tid * D;
    int k = 0;
    row[k++] = V.b.wb.x; row[k++] = V.b.wb.y; row[k++] = V.b.wb.z;
    quat qe = canon_quat(V.b.q);
    if (P.angle_repr) { row[k++] = qe.x; row[k++] = qe.y; row[k++] = qe.z; row[k++] = qe.w; }
    else { row[k++] = V.b.rpy.x; row[k++] = V.b.rpy.y; row[k++] = V.b.rpy.z; }
    row[k++] = V.b.vb.x; row[k++] = V.b.vb.y; row[k++] = V.b.vb.z;
    row[k++] = V.b.p.x; row[k++] = V.b.p.y; row[k++] = V.b.p.z;
    float aux[VEH::AUX];
    V.aux(aux);
    if (TASK == PF_TASK_MA_HOVER) {  // ma_quadx_hover_env.py:141-166: aux, past action, start_pos
#pragma unroll
      for (int a = 0; a < VEH::AUX; ++a) row[k++] = aux[a];
      row[k++] = ma_past.x; row[k++] = ma_past.y; row[k++] = ma_past.z; row[k++] = ma_past.w;
      row[k++] = tg.t[0][0]; row[k++] = tg.t[0][1]; row[k++] = tg.t[0][2];
    } else {
      row[k++] = act4[0]; row[k++] = act4[1]; row[k++] = act4[2]; row[k++] = act4[3];
#pragma unroll
      for (int a = 0; a < VEH::AUX; ++a) row[k++] = aux[a];
    }
    if (TASK == PF_TASK_WAYPOINTS) {
      m3 Re = rot_from_quat(qe);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (i < P.num_targets) {
          v3 d = mulT(Re, v3{tg.t[i][0] - V.b.p.x, tg.t[i][1] - V.b.p.y, tg.t[i][2] - V.b.p.z});
          bool live = i < tg.n_left;
          row[k++] = live ? d.x : 0.0f; row[k++] = live ? d.y : 0.0f; row[k++] = live ? d.z : 0.0f;
          if (kYaw) row[k++] = live ? wrap_pi(tg.yaw[i] - V.b.rpy.z) : 0.0f;  // waypoint_handler.py:144-153
        }
      }
    }
  };
  // tile -> global with full-width stores; LDS-only sync (one wave per workgroup, see quadx_fast.hpp)
  auto lds_sync = [&]() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
  };
  auto flush_tile = [&](float* out) {
    lds_sync();
    if (wave_all) {
      const int rows = min(kWave, n - wave_base);
      const int total = rows * D;
      float* g = out + (size_t)wave_base * D;
      stream_tile(tile, g, total, tid);
    } else if (active) {  // partial (masked reset): this lane writes its own row
      float* g = out + (size_t)lane * D;
      const float* row = tile + tid * D;
      for (int k = 0; k < D; ++k) g[k] = row[k];
    }
    lds_sync();
  };

  const int n_env_steps = roll_steps > 0 ? roll_steps : 1;
  for (int ks = 0; ks < n_env_steps; ++ks) {
  const size_t toff = (size_t)ks * N;  // this env step's slot in the trajectory buffers, in lanes
  if (ks > 0) {  // what the next launch would start from: the stored groups, re-derived
    V.relaunch(mode, flags);
    rpy_valid = false;
    old_dist = new_dist;
    yaw_err0 = 0.0f;
    reward = 0.0f;
#pragma unroll
    for (int k = 0; k < 6; ++k) sp[k] = 0.0f;
    act4[0] = act4[1] = act4[2] = act4[3] = 0.0f;
  }
  if (op == OP_RESET) {
    do_reset = (mask == nullptr) || (mask[li] != 0);
    // (shared worlds: a mask that names some agents of a world resets the world)
    if (TASK == PF_TASK_MA_HOVER && P.agents_per_world > 1) do_reset = widen_to_world(do_reset && valid, tid, P.agents_per_world);
    active = do_reset;
  } else {
    do_reset = (P.autoreset == PF_AUTORESET_NEXT_STEP) && (term || trunc);
    active = true;
  }
  active = active && valid;
  do_reset = do_reset && active;
  wave_all = __all(active || !valid);

  // ---------------------------------------------------------------- what each lane runs
  bool settling = false;  // in the settle phase of a reset (no env logic after an Aviary step)
  int my_its = 0;         // Aviary steps this lane still has to run in the loop below
  float4 a = float4{0.f, 0.f, 0.f, 0.f};
  if (roll_steps > 0 && B.actions == nullptr) {  // pf_sample_actions' draw for (lane, step0 + ks): every lane's, restarting or not
    f4 u = uniform4(philox4x32((uint32_t)P.seed, (uint32_t)(P.seed >> 32), (uint32_t)(lane0 + li), step0 + (uint32_t)ks, 0u, 3u));
    a = float4{fmaf(P.action_high[0] - P.action_low[0], u.a, P.action_low[0]), fmaf(P.action_high[1] - P.action_low[1], u.b, P.action_low[1]),
               fmaf(P.action_high[2] - P.action_low[2], u.c, P.action_low[2]), fmaf(P.action_high[3] - P.action_low[3], u.d, P.action_low[3])};
    if (B.actions_out != nullptr && valid) reinterpret_cast<float4*>(B.actions_out)[toff + li] = a;
  } else if (active && !do_reset) {
    a = reinterpret_cast<const float4*>(B.actions)[toff + li];
  }
  // (wave-uniform guard: most waves of most launches restart nobody. It also takes the reset's divergent region out of the path into
  //  the Aviary-step loop: with `if (do_reset) ... else if (active) ...` the allocator's copies of the zeroed PID memories landed in
  //  front of the join block's exec restore in the QuadX-Hover instantiations -- tools/isa_exec_check.py, 25 of round 5's 186 sites)
  if (__builtin_expect(__any(do_reset), 0)) {
    if (do_reset) {
      begin_reset();
      settling = true;
      my_its = (tmpl != nullptr) ? 0 : P.settle_steps;
    }
  }
  if (active && !do_reset) {
    if (TASK == PF_TASK_MA_HOVER) {  // past <- current, current <- action (ma_quadx_base_env.py:326-332)
      ma_past = float4{tg.t[2][1], tg.t[2][2], tg.t[3][0], tg.t[3][1]};
      tg.t[2][1] = a.x; tg.t[2][2] = a.y; tg.t[3][0] = a.z; tg.t[3][1] = a.w;
      reward = 0.0f;
      term = false; trunc = false;  // per-call flags (ma_quadx_base_env.py:336-337)
      my_its = P.env_step_ratio;    // no early exit in the multi-agent base (:342-361)
    } else {
      act4[0] = a.x; act4[1] = a.y; act4[2] = a.z; act4[3] = a.w;
      reward = -0.1f;
      my_its = (term || trunc) ? 0 : P.env_step_ratio;  // quadx_base_env.py:289-290
    }
    sp[0] = a.x; sp[1] = a.y; sp[2] = a.z;
    sp[3] = P.throttle_remap ? fmaf(a.w, 0.5f, 0.5f) : a.w;  // fixedwing_base_env.py:260
    nz.begin_event(rng_ctr, 0u, B.xi);
  }

  // Shared world (pz_envs: every agent's drone in ONE Bullet world): the A lanes of a world sit next to each other in the
  // wave; before every tick they exchange pose and contact bit through LDS, test their collision boxes against each other
  // (15 axes, in the peer's frame, behind a bounding-sphere test) and OR the world's contact bits into the gate of the
  // rotational drag. One Aviary.step = control + ticks_per_control x (exchange, tick).
  const int A = (TASK == PF_TASK_MA_HOVER && P.agents_per_world > 1) ? P.agents_per_world : 1;
  auto world_aviary_step = [&](int flat_base) {
    V.b.contact_step = false;
    V.template control<MODE_T>(P, sp);
    for (int t = 0; t < P.ticks_per_control; ++t) {
      world_exchange(V.b, wpose, tid, A, P.bound_radius, Pdev);  // (shared_world.hpp)
      if constexpr (TASK == PF_TASK_MA_HOVER) V.template tick<true>(P, nz.get(flat_base + t));  // (with the contact response between the drones)
      else V.tick(P, nz.get(flat_base + t));
    }
    V.b.peer_contact = false;
    V.b.rpy = euler_from_quat_fast(V.b.q);
  };
  int it = 0;
  while (__any(my_its > 0)) {
    if (my_its > 0) {
      if (A > 1) world_aviary_step(it * P.ticks_per_control);
      else V.template aviary_step<MODE_T>(P, sp, nz, it * P.ticks_per_control);
      rpy_valid = true;
      my_its -= 1;
      if (!settling) {
        wp_distance();
        term_trunc_reward();
        if (TASK != PF_TASK_MA_HOVER && (term || trunc)) my_its = 0;
      }
    }
    it += 1;
  }
  const bool stepped = active && !settling && op == OP_STEP;
  const float out_reward = stepped ? reward : 0.0f;
  const bool out_term = stepped && term, out_trunc = stepped && trunc;
  if (stepped) {
    step_count += 1; rng_ctr += 1;  // quadx_base_env.py:299
    if (V.nonfinite()) flags |= PF_F_NONFINITE;  // NaN / Inf guard (see quadx_fast.hpp)
  }

  // ---------------------------------------------------------------- SAME_STEP auto-reset (rare path)
  if (P.autoreset == PF_AUTORESET_SAME_STEP) {
    const bool same = stepped && (term || trunc);
    if (__any(same)) {
      if (B.final_obs != nullptr) {  // terminal observation, before the state is re-initialised
        if (active) write_obs_row();
        flush_tile(B.final_obs + toff * D);
      }
      if (B.final_info != nullptr && same) {  // gymnasium's final_info: the episode's flags / targets left, pre-reset
        B.final_info[2 * (toff + li) + 0] = (flags & ~(PF_F_TERMINATED | PF_F_TRUNCATED | PF_F_CONTACT)) | (term ? PF_F_TERMINATED : 0) |
                                            (trunc ? PF_F_TRUNCATED : 0) | (V.b.contact_now ? PF_F_CONTACT : 0);
        B.final_info[2 * (toff + li) + 1] = tg.n_left - (pop_pending ? 1 : 0);
      }
      if (same) {
        begin_reset();
        settling = true;
        if (tmpl == nullptr) {
          // through copies: handing &V itself to an out-of-line call would pin the whole vehicle
          // state in scratch memory for the entire kernel (measured: 1.9 KB/lane, 2x slower)
          VEH Vc = V;
          Noise nc = nz;
          float spc[6] = {sp[0], sp[1], sp[2], sp[3], sp[4], sp[5]};
          for (int s = 0; s < P.settle_steps; ++s) aviary_step_outlined<VEH, MODE_T>(&Vc, Pdev, spc, &nc, s * P.ticks_per_control);
          V = Vc;
        }
      }
    }
  }
  if (active && settling) {  // end_reset: compute_state after the settle steps (quadx_base_env.py:212)
    wp_distance();
    rng_ctr += 1;
  }

  // ---------------------------------------------------------------- outputs: obs tile first, state after
  if (active) write_obs_row();
  flush_tile(B.obs + toff * D);
  if (active) {
    if (pop_pending) { tg.pop(); pop_pending = false; }
    flags = (flags & ~(PF_F_TERMINATED | PF_F_TRUNCATED | PF_F_CONTACT)) | (term ? PF_F_TERMINATED : 0) |
            (trunc ? PF_F_TRUNCATED : 0) | (V.b.contact_now ? PF_F_CONTACT : 0);
    if (ks == n_env_steps - 1) {  // the state goes back to HBM once per launch
      V.store(Sout, N, li, mode, new_dist, int4{step_count, flags, (int)rng_ctr, tg.n_left});
      if constexpr (REKEY) {
        if (key_in11) Sout[(size_t)11 * N + li] = float4{V.zE[0], V.zE[1], __int_as_float((int)reset_kw), 0.0f};
        else if (reset_kw_dirty) Sout[(size_t)7 * N + li] = float4{0.0f, 0.0f, 0.0f, __int_as_float((int)reset_kw)};
      }
      if (kSide) tg.store(Sout, N, li, VEH::G_TGT);
      if (kYaw) Sout[(size_t)(VEH::G_TGT + 3) * N + li] = float4{tg.yaw[0], tg.yaw[1], tg.yaw[2], tg.yaw[3]};
      if (TASK == PF_TASK_MA_HOVER) Sout[(size_t)(VEH::G_TGT + 3) * N + li] = ma_past;
    }
    if (op == OP_STEP) {  // a NEXT_STEP reset call reports (r=0, not done), gymnasium's convention
      B.reward[toff + li] = out_reward;
      B.terminated[toff + li] = out_term ? 1 : 0;
      B.truncated[toff + li] = out_trunc ? 1 : 0;
    }
  }
  }  // (env steps of this launch)
}

// Settled spawn state for contexts whose settle phase is lane-independent (see env_kernel).
template <class VEH>
__global__ void settle_template_kernel(const pf_params P, float4* tmpl, const pf_params* __restrict__ Pdev) {
  __shared__ __attribute__((aligned(16))) float ktab[VEH::TABLE_FLOATS];
  VEH::fill_table(ktab, Pdev, threadIdx.x);
  __syncthreads();
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  __shared__ __attribute__((aligned(16))) float cws[kAviaryContactFloats];
  VEH V;
  V.b.pdev = Pdev;
  V.b.cws = (lds_fptr)cws;
  V.b.contact_regions(P, kAviaryContactFloats);
  V.bind(ktab);
  float sp[6] = {0, 0, 0, 0, 0, 0};
  V.reset(P, nullptr, sp);
  V.set_mode(P.flight_mode, sp);
  Noise nz;
  nz.mode = PF_NOISE_OFF; nz.n = 1; nz.lane = 0; nz.k0 = nz.k1 = nz.c0 = 0; nz.nmot = 0.f; nz.cached = -1; nz.xi = nullptr;
  nz.begin_event(0u, 1u, nullptr);
  for (int s = 0; s < P.settle_steps; ++s) V.template aviary_step<kRuntimeMode>(P, sp, nz, 0);
  V.store(tmpl, 1, 0, 7, INFINITY, int4{0, 0, 0, 0});
}

// ------------------------------------------------------------------ Aviary-level kernels
template <class VEH>
__global__ void __launch_bounds__(kWave) aviary_reset_kernel(const pf_params P, const pf_buffers B, const int n,
                                                             const float* pose) {
  const int lane = blockIdx.x * kWave + threadIdx.x;
  if (lane >= n) return;
  VEH V;
  V.b.pdev = nullptr;  // (no tick in this kernel)
  V.b.cws = nullptr;
  float sp[8];
  V.reset(P, pose ? pose + (size_t)lane * 7 : nullptr, sp, B.start_vel ? B.start_vel + (size_t)lane * 3 : nullptr);
  float4* S = reinterpret_cast<float4*>(B.state);
  V.store(S, (size_t)n, (size_t)lane, /*mode=*/7, INFINITY, int4{0, 0, 0, 0});
  if (B.out_state) {
    float* o = B.out_state + (size_t)lane * 12;
    o[0] = V.b.wb.x; o[1] = V.b.wb.y; o[2] = V.b.wb.z; o[3] = V.b.rpy.x; o[4] = V.b.rpy.y; o[5] = V.b.rpy.z;
    o[6] = V.b.vb.x; o[7] = V.b.vb.y; o[8] = V.b.vb.z; o[9] = V.b.p.x; o[10] = V.b.p.y; o[11] = V.b.p.z;
  }
  if (B.out_link_pos) {
#pragma unroll
    for (int k = 0; k < VEH::WIND_LINKS; ++k) {
      v3 lp = V.link_pos(P, k);
      float* o = B.out_link_pos + ((size_t)lane * VEH::WIND_LINKS + k) * 3;
      o[0] = lp.x; o[1] = lp.y; o[2] = lp.z;
    }
  }
  if (B.out_aux) {
    float aux[VEH::AUX];
    V.aux(aux);
    for (int k = 0; k < VEH::AUX; ++k) B.out_aux[(size_t)lane * VEH::AUX + k] = aux[k];
  }
}

template <class VEH>
__global__ void __launch_bounds__(kWave) aviary_set_mode_kernel(const pf_params P, const pf_buffers B, const int n,
                                                                const int sp_dim, const int new_mode, float* sp_out) {
  const int lane = blockIdx.x * kWave + threadIdx.x;
  if (lane >= n) return;
  VEH V;
  V.b.pdev = nullptr;  // (no tick in this kernel)
  V.b.cws = nullptr;
  float nd;
  int4 ints;
  // load everything (old mode 7 == all groups), re-initialise the controllers, store everything
  V.load(reinterpret_cast<const float4*>(B.state), (size_t)n, (size_t)lane, 7, nd, ints);
  V.b.rpy = euler_from_quat(V.b.q);
  float sp[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (sp_out)
    for (int k = 0; k < 8; ++k)
      if (k < sp_dim) sp[k] = sp_out[(size_t)lane * sp_dim + k];
  V.set_mode(B.modes ? B.modes[lane] : new_mode, sp);
  V.store(reinterpret_cast<float4*>(B.state), (size_t)n, (size_t)lane, 7, nd, ints);
  if (sp_out)
    for (int k = 0; k < 8; ++k)
      if (k < sp_dim) sp_out[(size_t)lane * sp_dim + k] = sp[k];
  (void)P;
}

template <class VEH>
__global__ void __launch_bounds__(kWave) aviary_step_kernel(const pf_params P, const pf_buffers B, const int n,
                                                            const uint64_t lane0, const int n_steps,
                                                            const pf_params* __restrict__ Pdev) {
  __shared__ __attribute__((aligned(16))) float ktab[VEH::TABLE_FLOATS];
  VEH::fill_table(ktab, Pdev, threadIdx.x);
  __syncthreads();
  const int lane = blockIdx.x * kWave + threadIdx.x;
  if (lane >= n) return;
  const size_t li = lane, N = n;
  __shared__ __attribute__((aligned(16))) float cws[kAviaryContactFloats];  // the contact solver's LDS regions (uav_vehicles.hpp)
  VEH V;
  V.b.pdev = Pdev;
  V.b.cws = (lds_fptr)cws;
  V.b.contact_regions(P, kAviaryContactFloats);
  V.bind(ktab);
  float nd;
  int4 ints;
  const int mode = B.modes ? B.modes[li] : P.flight_mode;
  V.load(reinterpret_cast<const float4*>(B.state), N, li, mode, nd, ints);
  V.b.rpy = euler_from_quat_fast(V.b.q);
  uint32_t rng_ctr = (uint32_t)ints.z;
  Noise nz;
  nz.mode = P.noise_mode; nz.n = n; nz.lane = lane;
  nz.k0 = (uint32_t)P.seed; nz.k1 = (uint32_t)(P.seed >> 32);
  nz.c0 = (uint32_t)(lane0 + li); nz.nmot = (float)P.n_motors; nz.cached = -1; nz.xi = nullptr;
  float sp[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const int spn = P.vehicle == PF_ROCKET ? 7 : ((P.vehicle == PF_FIXEDWING && mode == -1) ? 6 : 4);
#pragma unroll
  for (int k = 0; k < 8; ++k)
    if (k < spn) sp[k] = B.setpoints[li * spn + k];
  bool contact = false;
  const int ratio = B.ctrl_ratio ? B.ctrl_ratio[li] : 0;
  const bool armed = !B.armed || B.armed[li] != 0;  // aviary.py:423-438
  for (int s = 0; s < n_steps; ++s) {
    nz.begin_event(rng_ctr, 0u, B.xi ? B.xi + (size_t)s * P.ticks_per_control * N : nullptr);
    if (!armed) {  // no control, no forces, no state read-back: gravity only (aviary.py:510-521 skip it, Bullet does not)
      V.b.contact_step = false;
      for (int t = 0; t < P.ticks_per_control; ++t) V.tick_unarmed(P);
    } else if (ratio > 0 || B.modes) {  // this drone's own control rate (period ratio * dt) and / or flight mode
      const int rr = ratio > 0 ? ratio : P.ticks_per_control;
      V.b.contact_step = false;
      for (int t = 0; t < P.ticks_per_control; ++t) {
        if (t % rr == 0) {
          if (t > 0) V.b.rpy = euler_from_quat_fast(V.b.q);
          V.template control<kRuntimeMode>(P, sp, ratio > 0 ? ratio * P.dt : 0.0f, B.modes ? mode : kNoModeOverride);
        }
        V.tick(P, nz.get(t));
      }
      V.b.rpy = euler_from_quat_fast(V.b.q);
    } else {
      V.template aviary_step<kRuntimeMode>(P, sp, nz, 0);
    }
    rng_ctr += 1;
    contact = V.b.contact_step;
  }
  int flags = (ints.y & ~PF_F_CONTACT) | (V.b.contact_now ? PF_F_CONTACT : 0);
  V.store(reinterpret_cast<float4*>(B.state), N, li, mode, nd, int4{ints.x, flags, (int)rng_ctr, ints.w});
  if (B.out_state && armed) {
    float4* o = reinterpret_cast<float4*>(B.out_state + li * 12);
    o[0] = float4{V.b.wb.x, V.b.wb.y, V.b.wb.z, V.b.rpy.x};
    o[1] = float4{V.b.rpy.y, V.b.rpy.z, V.b.vb.x, V.b.vb.y};
    o[2] = float4{V.b.vb.z, V.b.p.x, V.b.p.y, V.b.p.z};
  }
  if (B.out_aux && armed) {
    float aux[VEH::AUX];
    V.aux(aux);
    for (int k = 0; k < VEH::AUX; ++k) B.out_aux[li * VEH::AUX + k] = aux[k];
  }
  if (B.out_contact) B.out_contact[li] = contact ? 1 : 0;
}

// One physics tick of Aviary.step (pf_aviary_tick): the wind-field protocol needs the host between
// ticks. QuadX carries the motor commands of the step's control tick in state group 12.
template <class VEH>
__global__ void __launch_bounds__(kWave) aviary_tick_kernel(const pf_params P, const pf_buffers B, const int n,
                                                            const uint64_t lane0, const int tick_index,
                                                            const pf_params* __restrict__ Pdev) {
  __shared__ __attribute__((aligned(16))) float ktab[VEH::TABLE_FLOATS];
  VEH::fill_table(ktab, Pdev, threadIdx.x);
  __syncthreads();
  const int lane = blockIdx.x * kWave + threadIdx.x;
  if (lane >= n) return;
  const size_t li = lane, N = n;
  constexpr bool kQuad = VEH::AUX == 4;
  constexpr int kCmdGroup = 12;
  __shared__ __attribute__((aligned(16))) float cws[kAviaryContactFloats];  // the contact solver's LDS regions (uav_vehicles.hpp)
  VEH V;
  V.b.pdev = Pdev;
  V.b.cws = (lds_fptr)cws;
  V.b.contact_regions(P, kAviaryContactFloats);
  V.bind(ktab);
  float nd;
  int4 ints;
  const int mode = B.modes ? B.modes[li] : P.flight_mode;
  float4* S = reinterpret_cast<float4*>(B.state);
  V.load(S, N, li, mode, nd, ints);
  V.b.rpy = euler_from_quat_fast(V.b.q);
  uint32_t rng_ctr = (uint32_t)ints.z;
  Noise nz;
  nz.mode = P.noise_mode; nz.n = n; nz.lane = lane;
  nz.k0 = (uint32_t)P.seed; nz.k1 = (uint32_t)(P.seed >> 32);
  nz.c0 = (uint32_t)(lane0 + li); nz.nmot = (float)P.n_motors; nz.cached = -1; nz.xi = nullptr;
  nz.begin_event(rng_ctr, 0u, B.xi);
  float sp[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const int spn = P.vehicle == PF_ROCKET ? 7 : ((P.vehicle == PF_FIXEDWING && mode == -1) ? 6 : 4);
#pragma unroll
  for (int k = 0; k < 8; ++k)
    if (k < spn) sp[k] = B.setpoints[li * spn + k];
  const int ratio = B.ctrl_ratio ? B.ctrl_ratio[li] : P.ticks_per_control;
  const bool armed = !B.armed || B.armed[li] != 0;
  if (!armed) {
    V.tick_unarmed(P);
  } else {
  if (tick_index % ratio == 0 || !kQuad) {
    V.template control<kRuntimeMode>(P, sp, B.ctrl_ratio ? ratio * P.dt : 0.0f, B.modes ? mode : kNoModeOverride);  // Fixedwing: stateless mixing, recomputed every tick
  } else {
    const float4 c = S[(size_t)kCmdGroup * N + li];
    V.set_cmd(c);
  }
  const float xi = nz.get(P.noise_mode == PF_NOISE_INJECT ? 0 : tick_index);
  V.tick(P, xi, B.wind ? B.wind + li * (size_t)(VEH::WIND_LINKS * 3) : nullptr);
  }
  V.b.rpy = euler_from_quat_fast(V.b.q);
  if (tick_index == P.ticks_per_control - 1) rng_ctr += 1;
  int flags = (ints.y & ~PF_F_CONTACT) | (V.b.contact_now ? PF_F_CONTACT : 0);
  V.store(S, N, li, mode, nd, int4{ints.x, flags, (int)rng_ctr, ints.w});
  if (kQuad) S[(size_t)kCmdGroup * N + li] = V.get_cmd();
  if (B.out_state && armed) {
    float4* o = reinterpret_cast<float4*>(B.out_state + li * 12);
    o[0] = float4{V.b.wb.x, V.b.wb.y, V.b.wb.z, V.b.rpy.x};
    o[1] = float4{V.b.rpy.y, V.b.rpy.z, V.b.vb.x, V.b.vb.y};
    o[2] = float4{V.b.vb.z, V.b.p.x, V.b.p.y, V.b.p.z};
  }
  if (B.out_aux && armed) {
    float aux[VEH::AUX];
    V.aux(aux);
    for (int k = 0; k < VEH::AUX; ++k) B.out_aux[li * VEH::AUX + k] = aux[k];
  }
  if (B.out_contact) B.out_contact[li] = V.b.contact_now ? 1 : 0;
  if (B.out_link_pos) {
#pragma unroll
    for (int k = 0; k < VEH::WIND_LINKS; ++k) {
      v3 lp = V.link_pos(P, k);
      float* o = B.out_link_pos + (li * VEH::WIND_LINKS + k) * 3;
      o[0] = lp.x; o[1] = lp.y; o[2] = lp.z;
    }
  }
}

// applyExternalForce / applyExternalTorque on the base link (LINK_FRAME) + stepSimulation, n_ticks times
// (pf_body_tick): the free-body tick by itself, for the integrator's known-answer tests.
template <class VEH>
__global__ void __launch_bounds__(kWave) body_tick_kernel(const pf_params P, const pf_buffers B, const int n, const int n_ticks,
                                                          const pf_params* __restrict__ Pdev) {
  const int lane = blockIdx.x * kWave + threadIdx.x;
  if (lane >= n) return;
  const size_t li = lane, N = n;
  __shared__ __attribute__((aligned(16))) float cws[kAviaryContactFloats];
  VEH V;
  V.b.pdev = Pdev;
  V.b.cws = (lds_fptr)cws;
  V.b.contact_regions(P, kAviaryContactFloats);
  float nd;
  int4 ints;
  float4* S = reinterpret_cast<float4*>(B.state);
  V.load(S, N, li, 7, nd, ints);
  const float* wr = B.wrench + li * 6;
  const v3 F{wr[0], wr[1], wr[2]}, tau{wr[3], wr[4], wr[5]};
  bool contact = false;
  for (int t = 0; t < n_ticks; ++t) {
    V.b.tick(P, F, tau);
    contact |= V.b.contact_now;
  }
  V.b.rpy = euler_from_quat_fast(V.b.q);
  int flags = (ints.y & ~PF_F_CONTACT) | (V.b.contact_now ? PF_F_CONTACT : 0);
  V.store(S, N, li, 7, nd, int4{ints.x, flags, ints.z, ints.w});
  if (B.out_state) {
    float4* o = reinterpret_cast<float4*>(B.out_state + li * 12);
    o[0] = float4{V.b.wb.x, V.b.wb.y, V.b.wb.z, V.b.rpy.x};
    o[1] = float4{V.b.rpy.y, V.b.rpy.z, V.b.vb.x, V.b.vb.y};
    o[2] = float4{V.b.vb.z, V.b.p.x, V.b.p.y, V.b.p.z};
  }
  if (B.out_contact) B.out_contact[li] = contact ? 1 : 0;
}

__global__ void __launch_bounds__(256) sample_actions_kernel(const pf_params P, float* actions, const int n,
                                                             const uint64_t lane0, const uint32_t step_index) {
  const int lane = blockIdx.x * 256 + threadIdx.x;
  if (lane >= n) return;
  f4 u = uniform4(philox4x32((uint32_t)P.seed, (uint32_t)(P.seed >> 32), (uint32_t)(lane0 + lane), step_index, 0u, 3u));
  float4 a{fmaf(P.action_high[0] - P.action_low[0], u.a, P.action_low[0]),
           fmaf(P.action_high[1] - P.action_low[1], u.b, P.action_low[1]),
           fmaf(P.action_high[2] - P.action_low[2], u.c, P.action_low[2]),
           fmaf(P.action_high[3] - P.action_low[3], u.d, P.action_low[3])};
  reinterpret_cast<float4*>(actions)[lane] = a;
}

}  // namespace pf

// ====================================================================== C ABI
struct pf_ctx {
  pf_params P;
  int n;
  int device;
  uint64_t lane0;
  char err[256];
  // hot-path specialisation (quadx_fast.hpp)
  bool fast;
  uint32_t* launch_ctr;  // device, one word per workgroup of the specialised QuadX kernel: env steps taken so far -- the cadence its waves refill their spares on (quadx_fast.hpp: QuadSpare)
  bool one_wave; // the batch is at most one wave per SIMD of the device (the 512-register instantiations' condition)
  bool lean;     // the batch is at most one wave per SIMD of the device: the specialised QuadX kernel's 512-register instantiation
  pf::QuadK K;
  pf_params* P_dev;  // device copy of P for the rarely-taken floor paths (contact detection and response) and the LDS constant tables
  float4* tmpl;      // settled spawn state for lane-independent resets (env_kernel), or null
  // Fixedwing-Waypoints specialisation (fixedwing_fast.hpp)
  bool fast_fw;
  pf::FwK FK;
  pf::FwTable* surf_dev;  // pre-combined surface + body constants (scalar-loaded per tick)
  bool df_fast;           // dogfight: the aircraft run on the specialised Fixedwing tick (dogfight.hpp: DfFastVeh)
};
static thread_local char g_err[256] = "";

static int fail(pf_ctx* ctx, int code, const char* msg) {
  snprintf(ctx ? ctx->err : g_err, 256, "%s", msg);
  if (ctx) snprintf(g_err, 256, "%s", msg);
  return code;
}
static int hip_fail(pf_ctx* ctx, hipError_t e, const char* where) {
  char buf[256];
  snprintf(buf, sizeof(buf), "%s: %s", where, hipGetErrorString(e));
  return fail(ctx, (int)e, buf);
}
#define PF_HIP(ctx, call)                                   \
  do {                                                      \
    hipError_t e__ = (call);                                \
    if (e__ != hipSuccess) return hip_fail(ctx, e__, #call); \
  } while (0)

template <int TASK>
static void launch_fast(pf_ctx* ctx, const pf_buffers* b, int op, const uint8_t* mask, hipStream_t s) {
  const int grid = (ctx->n + 64 * pf::kQuadWPB - 1) / (64 * pf::kQuadWPB);
  // (the one-wave-per-SIMD instantiation -- quadx_fast.hpp, WPS -- where the batch is no more than that and the kernel has it; since
  //  round 6 the PettingZoo task with independent lanes as well; since round 5 the cascaded flight modes: their fp64 controller needs the 512 registers -- 408 B of stack per lane under 256)
#define PF_FAST4(NZ, CR, MD, SH) do { constexpr int W1 = ((CR) && !(SH)) ? 1 : 2; \
    if (W1 == 1 && ctx->lean) hipLaunchKernelGGL((pf::quadx_m0_env_kernel<TASK, NZ, 64, 0, CR, MD, SH, W1>), dim3(grid), dim3(64 * pf::kQuadWPB), 0, s, ctx->K, *b, ctx->P_dev, ctx->n, ctx->lane0, op, mask, 1, 0u, ctx->launch_ctr); \
    else hipLaunchKernelGGL((pf::quadx_m0_env_kernel<TASK, NZ, 64, 0, CR, MD, SH, 2>), dim3(grid), dim3(64 * pf::kQuadWPB), 0, s, ctx->K, *b, ctx->P_dev, ctx->n, ctx->lane0, op, mask, 1, 0u, ctx->launch_ctr); } while (0)
  // (flight modes other than 0: the MODES instantiation, contact response compiled in -- quadk_from_params)
  // (shared worlds: PF_TASK_MA_HOVER with the contact response on -- quadk_from_params)
#define PF_FAST3(NZ, CR, MD) do { if (TASK == PF_TASK_MA_HOVER && CR && ctx->K.apw > 1) PF_FAST4(NZ, CR, MD, (TASK == PF_TASK_MA_HOVER && CR)); else PF_FAST4(NZ, CR, MD, false); } while (0)
#define PF_FAST(NZ) do { if (ctx->K.mode != 0) PF_FAST3(NZ, true, true); else if (ctx->P.contact_response) PF_FAST3(NZ, true, false); else PF_FAST3(NZ, false, false); } while (0)
  if (ctx->P.noise_mode == PF_NOISE_PHILOX) PF_FAST(PF_NOISE_PHILOX);
  else if (ctx->P.noise_mode == PF_NOISE_INJECT) PF_FAST(PF_NOISE_INJECT);
  else PF_FAST(PF_NOISE_OFF);
#undef PF_FAST
#undef PF_FAST3
#undef PF_FAST4
}
template <int TASK>
static void launch_rollout(pf_ctx* ctx, const pf_buffers* b, int k_steps, uint32_t step0, hipStream_t s) {
  const int grid = (ctx->n + 64 * pf::kQuadWPB - 1) / (64 * pf::kQuadWPB);
#define PF_ROLL4(NZ, R, CR, MD, SH) do { constexpr int W1 = ((CR) && !(SH)) ? 1 : 2; \
    if (W1 == 1 && ctx->lean) hipLaunchKernelGGL((pf::quadx_m0_env_kernel<TASK, NZ, 64, R, CR, MD, SH, W1>), dim3(grid), dim3(64 * pf::kQuadWPB), 0, s, ctx->K, *b, ctx->P_dev, ctx->n, ctx->lane0, 0, (const uint8_t*)nullptr, k_steps, step0, ctx->launch_ctr); \
    else hipLaunchKernelGGL((pf::quadx_m0_env_kernel<TASK, NZ, 64, R, CR, MD, SH, 2>), dim3(grid), dim3(64 * pf::kQuadWPB), 0, s, ctx->K, *b, ctx->P_dev, ctx->n, ctx->lane0, 0, (const uint8_t*)nullptr, k_steps, step0, ctx->launch_ctr); } while (0)
#define PF_ROLL3(NZ, R, CR, MD) do { if (TASK == PF_TASK_MA_HOVER && CR && ctx->K.apw > 1) PF_ROLL4(NZ, R, CR, MD, (TASK == PF_TASK_MA_HOVER && CR)); else PF_ROLL4(NZ, R, CR, MD, false); } while (0)
#define PF_ROLL(NZ, R) do { if (ctx->K.mode != 0) PF_ROLL3(NZ, R, true, true); else if (ctx->P.contact_response) PF_ROLL3(NZ, R, true, false); else PF_ROLL3(NZ, R, false, false); } while (0)
  if (b->actions == nullptr) {
    if (ctx->P.noise_mode == PF_NOISE_PHILOX) PF_ROLL(PF_NOISE_PHILOX, 1);
    else PF_ROLL(PF_NOISE_OFF, 1);
  } else {
    if (ctx->P.noise_mode == PF_NOISE_PHILOX) PF_ROLL(PF_NOISE_PHILOX, 2);
    else PF_ROLL(PF_NOISE_OFF, 2);
  }
#undef PF_ROLL
#undef PF_ROLL3
#undef PF_ROLL4
}
static void launch_fast_fw(pf_ctx* ctx, const pf_buffers* b, int op, const uint8_t* mask, hipStream_t s) {
  const int grid = (ctx->n + 63) / 64;
  // (the one-wave-per-SIMD instantiation -- fixedwing_fast.hpp, WPS -- where the batch is no more than that)
#define PF_FAST(NZ) do { if (ctx->one_wave) hipLaunchKernelGGL((pf::fixedwing_wp_env_kernel<NZ, 0, 1>), dim3(grid), dim3(64), 0, s, ctx->FK, ctx->surf_dev, *b, ctx->P_dev, ctx->tmpl, ctx->n, ctx->lane0, op, mask, 1, 0u); \
    else hipLaunchKernelGGL((pf::fixedwing_wp_env_kernel<NZ, 0, 2>), dim3(grid), dim3(64), 0, s, ctx->FK, ctx->surf_dev, *b, ctx->P_dev, ctx->tmpl, ctx->n, ctx->lane0, op, mask, 1, 0u); } while (0)
  if (ctx->P.noise_mode == PF_NOISE_PHILOX) PF_FAST(PF_NOISE_PHILOX);
  else if (ctx->P.noise_mode == PF_NOISE_INJECT) PF_FAST(PF_NOISE_INJECT);
  else PF_FAST(PF_NOISE_OFF);
#undef PF_FAST
}
static void launch_rollout_fw(pf_ctx* ctx, const pf_buffers* b, int k_steps, uint32_t step0, hipStream_t s) {
  const int grid = (ctx->n + 63) / 64;
#define PF_ROLL(NZ, R) do { if (ctx->one_wave) hipLaunchKernelGGL((pf::fixedwing_wp_env_kernel<NZ, R, 1>), dim3(grid), dim3(64), 0, s, ctx->FK, ctx->surf_dev, *b, ctx->P_dev, ctx->tmpl, ctx->n, ctx->lane0, 0, (const uint8_t*)nullptr, k_steps, step0); \
    else hipLaunchKernelGGL((pf::fixedwing_wp_env_kernel<NZ, R, 2>), dim3(grid), dim3(64), 0, s, ctx->FK, ctx->surf_dev, *b, ctx->P_dev, ctx->tmpl, ctx->n, ctx->lane0, 0, (const uint8_t*)nullptr, k_steps, step0); } while (0)
  if (b->actions == nullptr) {
    if (ctx->P.noise_mode == PF_NOISE_PHILOX) PF_ROLL(PF_NOISE_PHILOX, 1);
    else PF_ROLL(PF_NOISE_OFF, 1);
  } else {
    if (ctx->P.noise_mode == PF_NOISE_PHILOX) PF_ROLL(PF_NOISE_PHILOX, 2);
    else PF_ROLL(PF_NOISE_OFF, 2);
  }
#undef PF_ROLL
}
template <class VEH, int TASK>
static void launch_env_t(pf_ctx* ctx, const pf_buffers* b, int op, const uint8_t* mask, hipStream_t s, int roll_steps = 0, uint32_t step0 = 0u) {
  const int grid = (ctx->n + pf::kWave - 1) / pf::kWave;
  if (ctx->P.vehicle == PF_QUADX && ctx->P.flight_mode == 0)
    hipLaunchKernelGGL((pf::env_kernel<VEH, TASK, 0>), dim3(grid), dim3(pf::kWave), 0, s, ctx->P, *b, ctx->n, ctx->lane0, op, mask, ctx->tmpl, ctx->P_dev, roll_steps, step0);
  else
    hipLaunchKernelGGL((pf::env_kernel<VEH, TASK, pf::kRuntimeMode>), dim3(grid), dim3(pf::kWave), 0, s, ctx->P, *b, ctx->n, ctx->lane0, op, mask, ctx->tmpl, ctx->P_dev, roll_steps, step0);
}
extern "C" {

int pf_abi_version(void) { return PF_ABI_VERSION; }
size_t pf_sizeof_params(void) { return sizeof(pf_params); }
size_t pf_sizeof_buffers(void) { return sizeof(pf_buffers); }
const char* pf_last_error(const pf_ctx* ctx) { return ctx ? ctx->err : g_err; }

int pf_ctx_create(const pf_params* params, int n_lanes, int device, uint64_t lane_offset, pf_ctx** out) {
  if (!params || !out || n_lanes <= 0) return fail(nullptr, PF_ERR_ARG, "pf_ctx_create: bad argument");
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count == 0)
    return fail(nullptr, PF_ERR_NO_DEVICE, "pf_ctx_create: no HIP device (this library has no CPU fallback)");
  if (device < 0 || device >= count) return fail(nullptr, PF_ERR_ARG, "pf_ctx_create: bad device index");
  const pf_params& P = *params;
  if (P.vehicle != PF_QUADX && P.vehicle != PF_FIXEDWING && P.vehicle != PF_ROCKET) return fail(nullptr, PF_ERR_ARG, "unknown vehicle");
  if (P.vehicle == PF_ROCKET && P.task != PF_TASK_NONE)
    return fail(nullptr, PF_ERR_UNSUPPORTED, "the Rocket is available through the Aviary-level entry points only (Rocket-Landing needs a resting contact)");
  if (P.vehicle == PF_ROCKET && P.flight_mode != 0) return fail(nullptr, PF_ERR_ARG, "rocket flight_mode must be 0");
  if (P.task == PF_TASK_WAYPOINTS && (P.num_targets < 1 || P.num_targets > 4))
    return fail(nullptr, PF_ERR_UNSUPPORTED, "num_targets must be in 1..4");
  if (P.vehicle == PF_QUADX && (P.flight_mode < -1 || P.flight_mode > 7)) return fail(nullptr, PF_ERR_ARG, "quadx flight_mode must be in -1..7");
  if (P.vehicle == PF_FIXEDWING && (P.flight_mode < -1 || P.flight_mode > 0)) return fail(nullptr, PF_ERR_ARG, "fixedwing flight_mode must be -1 or 0");
  if (P.vehicle == PF_FIXEDWING && (P.task == PF_TASK_HOVER || P.task == PF_TASK_MA_HOVER)) return fail(nullptr, PF_ERR_UNSUPPORTED, "no fixedwing hover task in the reference");
  if (P.task == PF_TASK_DOGFIGHT) {  // ma_fixedwing_dogfight_env.py
    if (P.vehicle != PF_FIXEDWING) return fail(nullptr, PF_ERR_UNSUPPORTED, "the dogfight task flies fixedwing aircraft (ma_fixedwing_base_env.py:201)");
    if (P.df_team_size < 1 || 2 * P.df_team_size > pf::kDfMaxAgents || P.agents_per_world != 2 * P.df_team_size)
      return fail(nullptr, PF_ERR_ARG, "dogfight: agents_per_world must be 2 * df_team_size, at most 8");
    if (P.autoreset != PF_AUTORESET_OFF) return fail(nullptr, PF_ERR_ARG, "the multi-agent env has no auto-reset (PettingZoo parallel API)");
    if (P.angle_repr != 0) return fail(nullptr, PF_ERR_UNSUPPORTED, "the dogfight env observes Euler angles (ma_fixedwing_dogfight_env.py:92)");
    if (P.n_surf != PF_MAX_SURF || P.n_motors != 1) return fail(nullptr, PF_ERR_ARG, "dogfight: a five-surface, one-motor airframe");
    if (P.df_action_dim != 0 && P.df_action_dim != 4 && P.df_action_dim != 6) return fail(nullptr, PF_ERR_ARG, "dogfight: df_action_dim is 4 or 6");
  }
  if (P.agents_per_world > 1) {
    if (!((P.vehicle == PF_QUADX && P.task == PF_TASK_MA_HOVER) || P.task == PF_TASK_DOGFIGHT))
      return fail(nullptr, PF_ERR_UNSUPPORTED, "agents_per_world > 1 (a shared world) exists for the PettingZoo tasks only (QuadX hover, fixedwing dogfight)");
    if ((P.task != PF_TASK_DOGFIGHT && 64 % P.agents_per_world != 0) || n_lanes % P.agents_per_world != 0)
      return fail(nullptr, PF_ERR_ARG, "agents_per_world must divide 64 (the lanes of a world share a wavefront) and the lane count");
    // (the pair stage keeps one deepest-contact slot per agent of a world in registers: shared_world.hpp, pair_stage_dev)
    if (P.agents_per_world > 8) return fail(nullptr, PF_ERR_UNSUPPORTED, "a shared world holds at most 8 agents (ORC_MAX_WORLD in the oracle)");
    for (int k = 0; k < P.n_boxes; ++k)
      if (P.boxes[k].kind != 0 || P.boxes[k].yaw != 0.0f)
        return fail(nullptr, PF_ERR_UNSUPPORTED, "shared worlds test plain box colliders against each other (cf2x); this airframe has cylinders / yawed boxes");
  }
  if (P.use_yaw_targets && !(P.vehicle == PF_QUADX && P.task == PF_TASK_WAYPOINTS))
    return fail(nullptr, PF_ERR_UNSUPPORTED, "use_yaw_targets exists for QuadX-Waypoints only (fixedwing_waypoints_env.py:77 hard-wires False)");
  if (P.task == PF_TASK_MA_HOVER && P.autoreset != PF_AUTORESET_OFF) return fail(nullptr, PF_ERR_ARG, "the multi-agent env has no auto-reset (PettingZoo parallel API)");
  if (P.task != PF_TASK_NONE && P.vehicle == PF_FIXEDWING && P.flight_mode != 0) return fail(nullptr, PF_ERR_UNSUPPORTED, "fixedwing env uses flight_mode 0");
  // quat_integrate()'s polynomial range: |w| dt / 2 <= pi/8 given the per-coordinate clamp
  if (1.7320508f * P.max_coord_vel * P.dt * 0.5f > 0.3926991f + 1e-6f)
    return fail(nullptr, PF_ERR_UNSUPPORTED, "max_coord_vel * dt too large for the exponential-map polynomial");
  if (P.ticks_per_control < 1 || P.env_step_ratio < 0 || P.n_boxes > PF_MAX_BOXES) return fail(nullptr, PF_ERR_ARG, "bad loop constants");
  // the contact model's parameters (params.py: build_params checks the same for Python callers; a C caller gets the same answer
  // here): a negative residual threshold is sqrt -> NaN and ends every solve after one sweep, negative distances shrink the slab,
  // a manifold size other than 4 or 8 would be mapped silently
  if (P.contact_manifold_points != 4 && P.contact_manifold_points != 8) return fail(nullptr, PF_ERR_ARG, "contact_manifold_points must be 4 or 8");
  if (P.contact_response && P.contact_iters < 1) return fail(nullptr, PF_ERR_ARG, "contact_iters must be at least 1");
  if (!(P.contact_residual_threshold >= 0.0f) || !(P.contact_report_distance >= 0.0f) || !(P.contact_break_distance >= 0.0f) ||
      !(P.contact_margin >= 0.0f) || !(P.contact_slop >= 0.0f) || !(P.contact_erp >= 0.0f) || !(P.contact_friction >= 0.0f) ||
      !(P.contact_restitution >= 0.0f))
    return fail(nullptr, PF_ERR_ARG, "the contact model's distances, threshold, erp, friction and restitution must be >= 0");
  pf_ctx* c = new (std::nothrow) pf_ctx;
  if (!c) return fail(nullptr, PF_ERR_ARG, "out of host memory");
  c->P = P; c->n = n_lanes; c->device = device; c->lane0 = lane_offset; c->err[0] = 0; c->launch_ctr = nullptr;
  {  // the airframe's worst-case contact count (collider vertices), see pf_params.contact_max_points
    int pts = 0;
    for (int k = 0; k < P.n_boxes; ++k) pts += P.boxes[k].kind == 1 ? 16 : (P.contact_manifold_points >= 8 ? 8 : 4);
    c->P.contact_max_points = pts < 1 ? 1 : (pts > PF_MAX_CONTACTS ? PF_MAX_CONTACTS : pts);
  }
  c->P_dev = nullptr; c->tmpl = nullptr; c->surf_dev = nullptr;
  c->fast = pf::quadk_from_params(P, c->K) && getenv("PF_DISABLE_FAST") == nullptr;
  {
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess) cus = 0;
    const long waves = ((long)n_lanes + 63) / 64;
    // (the lean instantiations solve floor contacts in registers, four slots = the incident face: quadx_fast.hpp, quad_floor_solve)
    c->one_wave = cus > 0 && waves <= 4L * cus && getenv("PF_NO_LEAN_KERNEL") == nullptr;  // (4 SIMDs per CU)
    c->lean = c->one_wave && P.contact_manifold_points < 8;
  }
  pf::FwTable fsurf;
  c->fast_fw = pf::fwk_from_params(P, c->FK, fsurf) && getenv("PF_DISABLE_FAST") == nullptr;
  c->df_fast = P.task == PF_TASK_DOGFIGHT && pf::fw_table_from_params(P, fsurf) && getenv("PF_DISABLE_FAST") == nullptr;
  {  // device copy of the parameter block (LDS constant tables, the out-of-line floor test)
    int cur = -1;
    (void)hipGetDevice(&cur);
    (void)hipSetDevice(device);
    // (behind the block: what the specialised QuadX kernel's in-register floor solve reads -- quadx_fast.hpp: quad_solve_consts)
    hipError_t e = hipMalloc((void**)&c->P_dev, pf::kQuadSolveOffset + sizeof(float) * pf::kQuadSolveWords);
    if (e == hipSuccess) e = hipMemcpy(c->P_dev, &c->P, sizeof(pf_params), hipMemcpyHostToDevice);
    if (e == hipSuccess) {
      float sw[pf::kQuadSolveWords] = {0};
      if (c->fast) pf::quad_solve_words(c->P, sw);
      e = hipMemcpy(reinterpret_cast<char*>(c->P_dev) + pf::kQuadSolveOffset, sw, sizeof(sw), hipMemcpyHostToDevice);
    }
    if (e == hipSuccess && c->fast) {  // (quadx_fast.hpp: launch_ctr)
      const size_t words = (size_t)pf::kCtrStride * (((size_t)n_lanes + 63) / 64);
      e = hipMalloc((void**)&c->launch_ctr, sizeof(uint32_t) * words);
      if (e == hipSuccess) e = hipMemset(c->launch_ctr, 0, sizeof(uint32_t) * words);
    }
    if (e == hipSuccess && (c->fast_fw || c->df_fast)) {
      e = hipMalloc((void**)&c->surf_dev, sizeof(fsurf));
      if (e == hipSuccess) e = hipMemcpy(c->surf_dev, &fsurf, sizeof(fsurf), hipMemcpyHostToDevice);
    }
    if (cur >= 0) (void)hipSetDevice(cur);
    if (e != hipSuccess) { if (c->P_dev) hipFree(c->P_dev); if (c->launch_ctr) hipFree(c->launch_ctr); delete c; return hip_fail(nullptr, e, "pf_ctx_create: device parameter block"); }
  }
  if (!c->fast && (P.task == PF_TASK_HOVER || P.task == PF_TASK_WAYPOINTS) &&
      (P.vehicle == PF_FIXEDWING || P.noise_mode == PF_NOISE_OFF)) {
    // the settle phase cannot depend on the lane (fixedwing: throttle command 0 during settle, so the
    // motor noise scales nothing; or noise off): settle once here, resets copy the result
    int cur = -1;
    hipGetDevice(&cur);
    hipSetDevice(device);
    const int groups = P.vehicle == PF_QUADX ? pf::QuadX::GROUPS : pf::Fixedwing::GROUPS;
    hipError_t e = hipMalloc((void**)&c->tmpl, sizeof(float4) * groups);
    if (e == hipSuccess) e = hipMemset(c->tmpl, 0, sizeof(float4) * groups);
    if (e == hipSuccess) {
      if (P.vehicle == PF_QUADX) hipLaunchKernelGGL(pf::settle_template_kernel<pf::QuadX>, dim3(1), dim3(64), 0, 0, c->P, c->tmpl, c->P_dev);
      else hipLaunchKernelGGL(pf::settle_template_kernel<pf::Fixedwing>, dim3(1), dim3(64), 0, 0, c->P, c->tmpl, c->P_dev);
      e = hipDeviceSynchronize();
    }
    if (cur >= 0) hipSetDevice(cur);
    if (e != hipSuccess) { if (c->tmpl) hipFree(c->tmpl); if (c->surf_dev) hipFree(c->surf_dev); hipFree(c->P_dev); delete c; return hip_fail(nullptr, e, "pf_ctx_create: settle template"); }
  }
  *out = c;
  return PF_OK;
}
void pf_ctx_destroy(pf_ctx* ctx) {
  if (!ctx) return;
  if (ctx->P_dev) hipFree(ctx->P_dev);
  if (ctx->launch_ctr) hipFree(ctx->launch_ctr);
  if (ctx->tmpl) hipFree(ctx->tmpl);
  if (ctx->surf_dev) hipFree(ctx->surf_dev);
  delete ctx;
}
int pf_state_groups(const pf_ctx* ctx) {
  if (ctx->P.task == PF_TASK_DOGFIGHT) return pf::kDfGroups;
  // (the specialised QuadX kernel in a cascaded flight mode, no shared world: eleven more groups, the float32 remainders of its fp64
  //  rigid-body state and PID memories -- quadx_fast.hpp: QuadStateD)
  if (ctx->fast && ctx->K.mode != 0 && ctx->K.apw == 1) return pf::QuadX::GROUPS + 11;  // (16-19 state, 20-21 rate PID, 22-26 cascade)
  return ctx->P.vehicle == PF_QUADX ? pf::QuadX::GROUPS : (ctx->P.vehicle == PF_ROCKET ? pf::Rocket::GROUPS : pf::Fixedwing::GROUPS);
}
int pf_obs_dim(const pf_ctx* ctx) {
  const pf_params& P = ctx->P;
  if (P.task == PF_TASK_DOGFIGHT) return 19 + (P.df_action_dim == 6 ? 6 : 4) + (P.agents_per_world - 1) * 14;  // ma_fixedwing_dogfight_env.py:128-160
  int aux = P.vehicle == PF_QUADX ? 4 : 6;
  return (P.angle_repr ? 13 : 12) + 4 + aux + (P.task == PF_TASK_WAYPOINTS ? (P.use_yaw_targets ? 4 : 3) * P.num_targets : (P.task == PF_TASK_MA_HOVER ? 3 : 0));
}
int pf_n_lanes(const pf_ctx* ctx) { return ctx->n; }
int pf_ctx_is_specialised(const pf_ctx* ctx) { return ctx->fast ? 1 : (((ctx->fast_fw && ctx->tmpl) || ctx->df_fast) ? 2 : 0); }

static int ensure_device(pf_ctx* ctx) {
  int cur = -1;
  PF_HIP(ctx, hipGetDevice(&cur));
  if (cur != ctx->device) PF_HIP(ctx, hipSetDevice(ctx->device));
  return PF_OK;
}

static int launch_env(pf_ctx* ctx, const pf_buffers* b, int op, const uint8_t* mask, void* stream) {
  if (!ctx || !b || !b->state || !b->obs) return fail(ctx, PF_ERR_ARG, "pf_env_*: state and obs buffers are required");
  const pf_params& P = ctx->P;
  if (P.task == PF_TASK_NONE) return fail(ctx, PF_ERR_ARG, "pf_env_*: context has no env task (use the pf_aviary_* calls)");
  if (op == pf::OP_STEP && (!b->actions || !b->reward || !b->terminated || !b->truncated))
    return fail(ctx, PF_ERR_ARG, "pf_env_step: actions/reward/terminated/truncated buffers are required");
  if (P.noise_mode == PF_NOISE_INJECT && ((op == pf::OP_STEP && !b->xi) || !b->xi_reset))
    if (!(op == pf::OP_STEP && P.autoreset == PF_AUTORESET_OFF && b->xi))
      return fail(ctx, PF_ERR_ARG, "PF_NOISE_INJECT needs xi (step) and xi_reset (reset/auto-reset)");
  int rc = ensure_device(ctx);
  if (rc) return rc;
  hipStream_t s = (hipStream_t)stream;
  if (P.task == PF_TASK_DOGFIGHT) {
    const int lpw = (64 / P.agents_per_world) * P.agents_per_world;  // whole worlds per wave
    const dim3 grid((ctx->n + lpw - 1) / lpw);
#define PF_DF(AA, VV) hipLaunchKernelGGL((pf::dogfight_env_kernel<AA, VV>), grid, dim3(64), 0, s, ctx->P, *b, ctx->n, ctx->lane0, op, mask, ctx->P_dev, ctx->surf_dev)
#define PF_DFA(VV) switch (P.agents_per_world) { case 2: PF_DF(2, VV); break; case 4: PF_DF(4, VV); break; case 6: PF_DF(6, VV); break; default: PF_DF(8, VV); break; }
    if (ctx->df_fast) { PF_DFA(pf::DfFastVeh) } else { PF_DFA(pf::DfGenericVeh) }
#undef PF_DFA
#undef PF_DF
  } else if (ctx->fast) {
    if (P.task == PF_TASK_HOVER) launch_fast<PF_TASK_HOVER>(ctx, b, op, mask, s);
    else if (P.task == PF_TASK_MA_HOVER) launch_fast<PF_TASK_MA_HOVER>(ctx, b, op, mask, s);
    else launch_fast<PF_TASK_WAYPOINTS>(ctx, b, op, mask, s);
  } else if (ctx->fast_fw && ctx->tmpl) {
    launch_fast_fw(ctx, b, op, mask, s);
  } else if (P.vehicle == PF_QUADX) {
    if (P.task == PF_TASK_HOVER) launch_env_t<pf::QuadX, PF_TASK_HOVER>(ctx, b, op, mask, s);
    else if (P.task == PF_TASK_MA_HOVER) launch_env_t<pf::QuadX, PF_TASK_MA_HOVER>(ctx, b, op, mask, s);
    else launch_env_t<pf::QuadX, PF_TASK_WAYPOINTS>(ctx, b, op, mask, s);
  } else {
    launch_env_t<pf::Fixedwing, PF_TASK_WAYPOINTS>(ctx, b, op, mask, s);
  }
  PF_HIP(ctx, hipGetLastError());
  return PF_OK;
}
int pf_env_reset(pf_ctx* ctx, const pf_buffers* b, const uint8_t* mask, void* stream) {
  return launch_env(ctx, b, pf::OP_RESET, mask, stream);
}
int pf_env_step(pf_ctx* ctx, const pf_buffers* b, void* stream) { return launch_env(ctx, b, pf::OP_STEP, nullptr, stream); }

int pf_aviary_reset(pf_ctx* ctx, const pf_buffers* b, void* stream) {
  if (!ctx || !b || !b->state) return fail(ctx, PF_ERR_ARG, "pf_aviary_reset: state buffer required");
  int rc = ensure_device(ctx);
  if (rc) return rc;
  const int grid = (ctx->n + pf::kWave - 1) / pf::kWave;
  hipStream_t s = (hipStream_t)stream;
  const float* pose = b->start_pose;
  if (ctx->P.vehicle == PF_QUADX)
    hipLaunchKernelGGL(pf::aviary_reset_kernel<pf::QuadX>, dim3(grid), dim3(pf::kWave), 0, s, ctx->P, *b, ctx->n, pose);
  else if (ctx->P.vehicle == PF_ROCKET)
    hipLaunchKernelGGL(pf::aviary_reset_kernel<pf::Rocket>, dim3(grid), dim3(pf::kWave), 0, s, ctx->P, *b, ctx->n, pose);
  else
    hipLaunchKernelGGL(pf::aviary_reset_kernel<pf::Fixedwing>, dim3(grid), dim3(pf::kWave), 0, s, ctx->P, *b, ctx->n, pose);
  ctx->P.flight_mode = 0;  // drone.reset() -> set_mode(0) (quadx.py:224, fixedwing.py:196)
  PF_HIP(ctx, hipGetLastError());
  return PF_OK;
}
int pf_aviary_set_mode(pf_ctx* ctx, const pf_buffers* b, int mode, float* setpoints_out, void* stream) {
  if (!ctx || !b || !b->state) return fail(ctx, PF_ERR_ARG, "pf_aviary_set_mode: state buffer required");
  if (ctx->P.vehicle == PF_QUADX && (mode < -1 || mode > 7)) return fail(ctx, PF_ERR_ARG, "`mode` must be between -1 and 7");
  if (ctx->P.vehicle == PF_FIXEDWING && (mode < -1 || mode > 0)) return fail(ctx, PF_ERR_ARG, "`mode` must be between -1 and 0");
  if (ctx->P.vehicle == PF_ROCKET && mode != 0) return fail(ctx, PF_ERR_ARG, "`mode` must be 0 (rocket.py:238-247)");
  if (b->modes && ctx->P.vehicle != PF_QUADX) return fail(ctx, PF_ERR_UNSUPPORTED, "per-drone flight modes are supported for QuadX only (Fixedwing modes differ in setpoint width)");
  int rc = ensure_device(ctx);
  if (rc) return rc;
  const int grid = (ctx->n + pf::kWave - 1) / pf::kWave;
  hipStream_t s = (hipStream_t)stream;
  const int sp_dim = ctx->P.vehicle == PF_ROCKET ? 7 : ((ctx->P.vehicle == PF_FIXEDWING && mode == -1) ? 6 : 4);  // fixedwing.py:221-224, rocket.py:228
  if (ctx->P.vehicle == PF_QUADX)
    hipLaunchKernelGGL(pf::aviary_set_mode_kernel<pf::QuadX>, dim3(grid), dim3(pf::kWave), 0, s, ctx->P, *b, ctx->n, sp_dim, mode, setpoints_out);
  else if (ctx->P.vehicle == PF_ROCKET)
    hipLaunchKernelGGL(pf::aviary_set_mode_kernel<pf::Rocket>, dim3(grid), dim3(pf::kWave), 0, s, ctx->P, *b, ctx->n, sp_dim, mode, setpoints_out);
  else
    hipLaunchKernelGGL(pf::aviary_set_mode_kernel<pf::Fixedwing>, dim3(grid), dim3(pf::kWave), 0, s, ctx->P, *b, ctx->n, sp_dim, mode, setpoints_out);
  ctx->P.flight_mode = mode;
  PF_HIP(ctx, hipGetLastError());
  return PF_OK;
}
int pf_aviary_step(pf_ctx* ctx, const pf_buffers* b, int n_steps, void* stream) {
  if (!ctx || !b || !b->state || !b->setpoints) return fail(ctx, PF_ERR_ARG, "pf_aviary_step: state and setpoints required");
  if (n_steps < 1) return fail(ctx, PF_ERR_ARG, "pf_aviary_step: n_steps must be >= 1");
  if (ctx->P.noise_mode == PF_NOISE_INJECT && !b->xi) return fail(ctx, PF_ERR_ARG, "PF_NOISE_INJECT needs xi");
  int rc = ensure_device(ctx);
  if (rc) return rc;
  const int grid = (ctx->n + pf::kWave - 1) / pf::kWave;
  hipStream_t s = (hipStream_t)stream;
  if (ctx->P.vehicle == PF_QUADX)
    hipLaunchKernelGGL(pf::aviary_step_kernel<pf::QuadX>, dim3(grid), dim3(pf::kWave), 0, s, ctx->P, *b, ctx->n, ctx->lane0, n_steps, ctx->P_dev);
  else if (ctx->P.vehicle == PF_ROCKET)
    hipLaunchKernelGGL(pf::aviary_step_kernel<pf::Rocket>, dim3(grid), dim3(pf::kWave), 0, s, ctx->P, *b, ctx->n, ctx->lane0, n_steps, ctx->P_dev);
  else
    hipLaunchKernelGGL(pf::aviary_step_kernel<pf::Fixedwing>, dim3(grid), dim3(pf::kWave), 0, s, ctx->P, *b, ctx->n, ctx->lane0, n_steps, ctx->P_dev);
  PF_HIP(ctx, hipGetLastError());
  return PF_OK;
}
int pf_aviary_tick(pf_ctx* ctx, const pf_buffers* b, int tick_index, void* stream) {
  if (!ctx || !b || !b->state || !b->setpoints) return fail(ctx, PF_ERR_ARG, "pf_aviary_tick: state and setpoints buffers are required");
  if (tick_index < 0 || tick_index >= ctx->P.ticks_per_control) return fail(ctx, PF_ERR_ARG, "pf_aviary_tick: tick_index must be in [0, ticks_per_control)");
  if (ctx->P.noise_mode == PF_NOISE_INJECT && !b->xi) return fail(ctx, PF_ERR_ARG, "pf_aviary_tick: PF_NOISE_INJECT needs b->xi");
  int rc = ensure_device(ctx);
  if (rc) return rc;
  const int grid = (ctx->n + pf::kWave - 1) / pf::kWave;
  hipStream_t s = (hipStream_t)stream;
  if (ctx->P.vehicle == PF_QUADX)
    hipLaunchKernelGGL(pf::aviary_tick_kernel<pf::QuadX>, dim3(grid), dim3(pf::kWave), 0, s, ctx->P, *b, ctx->n, ctx->lane0, tick_index, ctx->P_dev);
  else if (ctx->P.vehicle == PF_ROCKET)
    hipLaunchKernelGGL(pf::aviary_tick_kernel<pf::Rocket>, dim3(grid), dim3(pf::kWave), 0, s, ctx->P, *b, ctx->n, ctx->lane0, tick_index, ctx->P_dev);
  else
    hipLaunchKernelGGL(pf::aviary_tick_kernel<pf::Fixedwing>, dim3(grid), dim3(pf::kWave), 0, s, ctx->P, *b, ctx->n, ctx->lane0, tick_index, ctx->P_dev);
  PF_HIP(ctx, hipGetLastError());
  return PF_OK;
}
int pf_wind_links(const pf_ctx* ctx) {
  return ctx->P.vehicle == PF_QUADX ? pf::QuadX::WIND_LINKS : (ctx->P.vehicle == PF_ROCKET ? pf::Rocket::WIND_LINKS : pf::Fixedwing::WIND_LINKS);
}
int pf_sample_actions(pf_ctx* ctx, float* actions, uint32_t step_index, void* stream) {
  if (!ctx || !actions) return fail(ctx, PF_ERR_ARG, "pf_sample_actions: bad argument");
  int rc = ensure_device(ctx);
  if (rc) return rc;
  hipLaunchKernelGGL(pf::sample_actions_kernel, dim3((ctx->n + 255) / 256), dim3(256), 0, (hipStream_t)stream, ctx->P, actions, ctx->n, ctx->lane0, step_index);
  PF_HIP(ctx, hipGetLastError());
  return PF_OK;
}

int pf_rollout(pf_ctx* ctx, const pf_buffers* b, int k_steps, uint32_t step_index0, void* stream) {
  if (!ctx || !b || !b->state || !b->obs || !b->reward || !b->terminated || !b->truncated)
    return fail(ctx, PF_ERR_ARG, "pf_rollout: state, obs, reward, terminated and truncated buffers are required");
  if (k_steps < 1) return fail(ctx, PF_ERR_ARG, "pf_rollout: k_steps must be >= 1");
  const pf_params& P = ctx->P;
  const bool fw = ctx->fast_fw && ctx->tmpl;
  if (P.noise_mode == PF_NOISE_INJECT) return fail(ctx, PF_ERR_UNSUPPORTED, "pf_rollout: PF_NOISE_INJECT is a per-step protocol; use pf_env_step");
  if (P.task == PF_TASK_NONE) return fail(ctx, PF_ERR_UNSUPPORTED, "pf_rollout: this context has no env task");
  if (P.task == PF_TASK_DOGFIGHT) {
    // the dogfight, state-resident on either aircraft model: dogfight_env_kernel<.., ROLLOUT = true> (four-wide actions sampled on
    // device, or the given sequence of either width)
    if (!b->actions && P.df_action_dim == 6) return fail(ctx, PF_ERR_UNSUPPORTED, "pf_rollout: on-device sampling draws four-wide actions; pass the six-wide sequence in b->actions");
    int rc = ensure_device(ctx);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    const int lpw = (64 / P.agents_per_world) * P.agents_per_world;  // whole worlds per wave
    dim3 grid((ctx->n + lpw - 1) / lpw);
#define PF_DFR(AA, VV) hipLaunchKernelGGL((pf::dogfight_env_kernel<AA, VV, true>), grid, dim3(64), 0, s, ctx->P, *b, ctx->n, ctx->lane0, 0, \
                                          (const uint8_t*)nullptr, ctx->P_dev, ctx->surf_dev, k_steps, step_index0)
#define PF_DFRA(VV) switch (P.agents_per_world) { case 2: PF_DFR(2, VV); break; case 4: PF_DFR(4, VV); break; case 6: PF_DFR(6, VV); break; default: PF_DFR(8, VV); break; }
    if (ctx->df_fast) { PF_DFRA(pf::DfFastVeh) } else { PF_DFRA(pf::DfGenericVeh) }
#undef PF_DFRA
#undef PF_DFR
    PF_HIP(ctx, hipGetLastError());
    return PF_OK;
  }
  if (!fw && !ctx->fast) {
    // Every other task (the generic env kernel: tilted multi-agent spawns, airframes and flight modes outside the specialised
    // envelopes): state-resident as well since round 4 -- env_kernel's roll_steps, one launch for the k_steps.
    int rc = ensure_device(ctx);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    if (P.vehicle == PF_QUADX) {
      if (P.task == PF_TASK_HOVER) launch_env_t<pf::QuadX, PF_TASK_HOVER>(ctx, b, pf::OP_STEP, nullptr, s, k_steps, step_index0);
      else if (P.task == PF_TASK_MA_HOVER) launch_env_t<pf::QuadX, PF_TASK_MA_HOVER>(ctx, b, pf::OP_STEP, nullptr, s, k_steps, step_index0);
      else launch_env_t<pf::QuadX, PF_TASK_WAYPOINTS>(ctx, b, pf::OP_STEP, nullptr, s, k_steps, step_index0);
    } else {
      launch_env_t<pf::Fixedwing, PF_TASK_WAYPOINTS>(ctx, b, pf::OP_STEP, nullptr, s, k_steps, step_index0);
    }
    PF_HIP(ctx, hipGetLastError());
    return PF_OK;
  }
  // (the PettingZoo task has no auto-reset: finished agents are culled by the caller, their drones fly on in the shared world)
  if (P.autoreset == PF_AUTORESET_OFF && P.task != PF_TASK_MA_HOVER)
    return fail(ctx, PF_ERR_UNSUPPORTED, "pf_rollout: needs an auto-reset mode (finished lanes would idle for the rest of the launch)");
  int rc = ensure_device(ctx);
  if (rc) return rc;
  hipStream_t s = (hipStream_t)stream;
  if (fw) launch_rollout_fw(ctx, b, k_steps, step_index0, s);
  else if (P.task == PF_TASK_HOVER) launch_rollout<PF_TASK_HOVER>(ctx, b, k_steps, step_index0, s);
  else if (P.task == PF_TASK_MA_HOVER) launch_rollout<PF_TASK_MA_HOVER>(ctx, b, k_steps, step_index0, s);
  else launch_rollout<PF_TASK_WAYPOINTS>(ctx, b, k_steps, step_index0, s);
  PF_HIP(ctx, hipGetLastError());
  return PF_OK;
}
int pf_body_tick(pf_ctx* ctx, const pf_buffers* b, int n_ticks, void* stream) {
  if (!ctx || !b || !b->state || !b->wrench) return fail(ctx, PF_ERR_ARG, "pf_body_tick: state and wrench buffers are required");
  if (n_ticks < 1) return fail(ctx, PF_ERR_ARG, "pf_body_tick: n_ticks must be >= 1");
  if (ctx->P.vehicle == PF_ROCKET) return fail(ctx, PF_ERR_UNSUPPORTED, "pf_body_tick: the Rocket's mass properties change per tick; QuadX / Fixedwing only");
  int rc = ensure_device(ctx);
  if (rc) return rc;
  const int grid = (ctx->n + pf::kWave - 1) / pf::kWave;
  hipStream_t s = (hipStream_t)stream;
  if (ctx->P.vehicle == PF_QUADX)
    hipLaunchKernelGGL(pf::body_tick_kernel<pf::QuadX>, dim3(grid), dim3(pf::kWave), 0, s, ctx->P, *b, ctx->n, n_ticks, ctx->P_dev);
  else
    hipLaunchKernelGGL(pf::body_tick_kernel<pf::Fixedwing>, dim3(grid), dim3(pf::kWave), 0, s, ctx->P, *b, ctx->n, n_ticks, ctx->P_dev);
  PF_HIP(ctx, hipGetLastError());
  return PF_OK;
}

#ifdef PF_PHASE_TRACE
// diagnostic variant only (profiles/tools/solver_trace.py): read (and clear) the contact solver's call statistics
int pf_debug_solver_trace(unsigned long long* out) {
  hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(pf::g_solver_trace), sizeof(unsigned long long) * 8, 0, hipMemcpyDeviceToHost);
  unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (e == hipSuccess) e = hipMemcpyToSymbol(HIP_SYMBOL(pf::g_solver_trace), z, sizeof(z), 0, hipMemcpyHostToDevice);
  return (int)e;
}
// diagnostic variant only: (waves that were not calm, waves that took the calm test) since the last call; clears
int pf_debug_calm_trace(unsigned long long* out) {
  hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(pf::g_calm_trace), sizeof(unsigned long long) * 2, 0, hipMemcpyDeviceToHost);
  unsigned long long z[2] = {0, 0};
  if (e == hipSuccess) e = hipMemcpyToSymbol(HIP_SYMBOL(pf::g_calm_trace), z, sizeof(z), 0, hipMemcpyHostToDevice);
  return (int)e;
}
// diagnostic variant only (profiles/tools/phase_trace.py): copy out the per-wave phase stamps of the last quadx_m0 launch
int pf_debug_phase_trace(unsigned long long* out, int n_words) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(pf::g_phase_trace), sizeof(unsigned long long) * (size_t)n_words, 0, hipMemcpyDeviceToHost);
}
#endif

}  // extern "C"
