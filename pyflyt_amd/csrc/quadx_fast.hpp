// quadx_fast.hpp -- the hand-tuned hot kernel: QuadX, flight mode 0 (rate PID + thrust), Hover,
// Waypoints and multi-agent Hover tasks. This is the kernel BASELINE.json's metric is quoted on.
//
// What makes it lean (measured with rocprofv3 PMC, profiles/):
//   * a compact, vehicle-specific constant block (QuadK, ~60 dwords) instead of the generic
//     pf_params, so every constant lives in SGPRs (no SGPR->VGPR spill traffic);
//   * structural facts the reference hard-codes (quadx.py:94-101,130-137: +z thrust axis, torque
//     signs, the +-1 motor map; identical motors) are compile-time, not data;
//   * no libm: v_rcp/v_rsq/v_sqrt, polynomial atan2, half-angle algebra for the observation
//     quaternion, polynomial exponential map;
//   * the reset's 10 settle Aviary steps collapse to a 20-tick vertical recurrence when the spawn
//     pose is level and at rest (always true for the Hover/Waypoints envs), so lanes that
//     auto-reset cost ~1/4 of an env step instead of 3x one;
//   * flat control flow (one predicate per Aviary step) so the register allocator emits no PHI
//     copies; obs tile flushed before the state stores with an LDS-only sync; non-temporal obs
//     stores; the step's Philox call issued while the state loads are in flight.
//   (Tried and rejected, profiles/README.md: under-filled 16/32-lane waves, 128-VGPR variants.)
#pragma once
#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "../../include/pyflyt_amd.h"
#include "uav_device.hpp"
#include "uav_vehicles.hpp"  // contact_solve_dev
#include "quadx_control_d.hpp"
#include "shared_world.hpp"

namespace pf {

struct QuadK {
  // integrator
  float dt, half_dt, gravity_z, vmax, inv_mass;
  float I[3], iI[3];           // diagonal inertia and its inverse (cf2x.urdf:14)
  float use_gyro;              // 1.0 / 0.0
  float bound_radius;          // gate of the out-of-line floor code (incl. the speculative contact margin)
  float bound_radius0, slop;   // the bare bounding radius and the allowed overlap: "can a contact constraint act this tick?"
  float margin;                // fresh contact points: no vertex above it is one (pf_params.contact_margin)
  float brk, rd;               // persisting points / reports reach up to brk (contact_break_distance); fresh reports from rd on (contact_report_distance)
  float c_erp;                 // contact_erp (the rest of the contact model is QuadSolveC's: only the in-register solve reads it)
  int32_t c_all8;              // 1: every vertex of the box is a candidate (contact_manifold_points = 8)
  float box_h[3], plane_xy, plane_z;  // the collision box's half extents and the slab's: kernel-argument SGPRs, because with random
                                      // actions some lane of nearly every wave is near the floor in nearly every tick -- as scalar
                                      // loads inside that block they cost a memory round trip per tick (+0.7 us per env step)
  // motors (identical): motors.py:131-138,182-193
  float m_a, m_noise, fmax, tmax;
  float ryf[4], rxf[4];        // r_y*fmax, -r_x*fmax per motor (torque arms)
  float drag[3], pqr;          // boring_bodies.py:63, cf2x.yaml:11
  // the same constants with the inverse inertia / inverse mass folded in (diagonal inertia, COM at the base
  // origin): the tick computes angular and linear ACCELERATIONS directly
  float ryfI[4], rxfI[4], tmaxI;   // ryf * iI.x, rxf * iI.y, tmax * iI.z
  float pqI[3];                    // pqr * iI[k]
  float gyI[3];                    // iI.x (I.z - I.y), iI.y (I.x - I.z), iI.z (I.y - I.x): -(w x I w) / I, 0 without the gyro term
  float dragM[3], fmaxM;           // drag[k] * inv_mass, fmax * inv_mass
  // rate PID (cf2x.yaml:13-19): kp, ki*T, kd/T, lim
  float kp[3], kiT[3], kdT[3], lim[3];
  // env
  float start_pos[3], start_quat[4];
  float dome2, goal_reach, min_height, dome09m1;
  float wp_dist_reward, wp_yaw_penalty;
  int32_t task_sparse, angle_repr, num_targets, max_steps, env_step_ratio, settle_steps, tpc;
  int32_t noise_mode, autoreset, fast_settle;
  uint32_t seed_lo, seed_hi;
  int32_t apw;                   // agents per world (PF_TASK_MA_HOVER with a shared world), else 1
  int32_t mode;                  // flight mode -1 .. 7 (quadx.py:233-373); 0 for the MODES = false instantiations
  int32_t use_yaw;               // quadx_waypoints_env.py:40
  float goal_angle;              // :42
  float act_lo[4], act_span[4];  // action box (quadx_base_env.py:80-102): low, high - low (pf_rollout's on-device sampling)
  // "this lane cannot come within reach of the floor during this env step / this Aviary step" (the calm tests in the kernel): the
  // horizon T and T (T + dt) / 2 for the env step and for one Aviary step, 4 fmax / m, the square of the bound t* the motor state
  // cannot grow beyond once it is below it, the largest drag / m
  float calm_T, calm_TT, calm_T2, calm_TT2, calm_kt, calm_t2, calm_c;
  int32_t calm_on;
};

// Fill QuadK from the ABI struct; returns false when the configuration needs the generic kernel.
inline bool quadk_from_params(const pf_params& P, QuadK& K) {
  if (P.vehicle != PF_QUADX || P.flight_mode < -1 || P.flight_mode > 7) return false;
  // the cascaded modes' instantiations exist with the contact response compiled in only (the default; the detection-only
  // opt-out of a cascaded-mode env runs on the generic kernel)
  if (P.flight_mode != 0 && !P.contact_response) return false;
  if (P.flight_mode != 0 && P.ticks_per_control * P.dt != P.control_period) return false;
  if (P.task != PF_TASK_HOVER && P.task != PF_TASK_WAYPOINTS && P.task != PF_TASK_MA_HOVER) return false;
  if (P.has_com_offset) return false;
  // shared worlds (the PettingZoo task): the lanes of a world exchange poses through LDS before every tick; whole worlds per
  // wave, and a dead drone must come to rest on the floor
  if (P.agents_per_world > 1 && (P.task != PF_TASK_MA_HOVER || 64 % P.agents_per_world != 0 || !P.contact_response)) return false;
  {  // QuadHot::derive() builds the rotation with scale 2 instead of btMatrix3x3::setRotation's 2/|q|^2: unit spawn quaternion only
    const float q2 = P.start_quat[0] * P.start_quat[0] + P.start_quat[1] * P.start_quat[1] + P.start_quat[2] * P.start_quat[2] + P.start_quat[3] * P.start_quat[3];
    if (!(q2 > 1.0f - 1e-6f && q2 < 1.0f + 1e-6f)) return false;
  }
  if (P.I_own[1] != 0.f || P.I_own[2] != 0.f || P.I_own[4] != 0.f) return false;  // diagonal inertia only
  for (int k = 0; k < 6; ++k)
    if (P.I_pa[k] != 0.f) return false;
  const float ref_map[4][4] = {{-1, -1, -1, 1}, {1, 1, -1, 1}, {1, -1, 1, 1}, {-1, 1, 1, 1}};
  for (int i = 0; i < 4; ++i) {
    for (int j = 0; j < 4; ++j)
      if (P.motor_map[i][j] != ref_map[i][j]) return false;
    if (P.motor_dt_over_tau[i] != P.motor_dt_over_tau[0] || P.motor_noise[i] != P.motor_noise[0] ||
        P.motor_fmax[i] != P.motor_fmax[0]) return false;
    if (P.thrust_unit[i][0] != 0.f || P.thrust_unit[i][1] != 0.f || P.thrust_unit[i][2] != 1.f) return false;
  }
  if (!(P.motor_tmax[0] == P.motor_tmax[1] && P.motor_tmax[2] == P.motor_tmax[3] && P.motor_tmax[0] == -P.motor_tmax[2] &&
        P.motor_tmax[0] <= 0.f)) return false;
  if (P.n_boxes != 1 || P.boxes[0].kind != 0 || P.num_targets > 4) return false;
  // (the tick's direct lowest-vertex floor test assumes the one collision box is centred on the base origin and not yawed)
  if (P.boxes[0].c[0] != 0.f || P.boxes[0].c[1] != 0.f || P.boxes[0].c[2] != 0.f || P.boxes[0].yaw != 0.f) return false;
  if (P.ticks_per_control != 2 || P.env_step_ratio > 4 || P.env_step_ratio < 1) return false;
  if ((P.settle_steps * 2) % 4 != 0 || P.settle_steps * 2 > 24) return false;
  K.dt = P.dt; K.half_dt = 0.5f * P.dt; K.gravity_z = P.gravity_z; K.vmax = P.max_coord_vel; K.inv_mass = P.inv_mass;
  K.I[0] = P.I_own[0]; K.I[1] = P.I_own[3]; K.I[2] = P.I_own[5];
  K.iI[0] = P.I_inv[0]; K.iI[1] = P.I_inv[3]; K.iI[2] = P.I_inv[5];
  K.use_gyro = P.use_gyro_term ? 1.f : 0.f;
  // gate of the floor code: within one bounding radius of the floor, widened by the farthest a contact point or report reaches
  K.bound_radius = P.bound_radius + fmaxf(fmaxf(P.contact_margin, P.contact_break_distance), P.contact_report_distance);
  K.bound_radius0 = P.bound_radius; K.slop = P.contact_slop; K.margin = P.contact_margin;
  K.brk = P.contact_break_distance; K.rd = P.contact_report_distance;
  K.c_erp = P.contact_erp;
  K.c_all8 = P.contact_manifold_points >= 8 ? 1 : 0;
  for (int k = 0; k < 3; ++k) K.box_h[k] = P.boxes[0].h[k];
  K.plane_xy = P.plane_half_xy; K.plane_z = P.plane_half_z;
  K.m_a = P.motor_dt_over_tau[0]; K.m_noise = P.motor_noise[0]; K.fmax = P.motor_fmax[0]; K.tmax = P.motor_tmax[2];
  for (int i = 0; i < 4; ++i) { K.ryf[i] = P.motor_r[i][1] * P.motor_fmax[0]; K.rxf[i] = -P.motor_r[i][0] * P.motor_fmax[0]; }
  for (int k = 0; k < 3; ++k) {
    K.drag[k] = P.drag_const[k];
    K.kp[k] = P.pid[0].kp[k]; K.kiT[k] = P.pid[0].ki[k] * P.control_period; K.kdT[k] = P.pid[0].kd[k] * P.inv_control_period;
    K.lim[k] = P.pid[0].lim[k];
    K.start_pos[k] = P.start_pos[k];
  }
  K.pqr = P.drag_coef_pqr;
  for (int i = 0; i < 4; ++i) { K.ryfI[i] = K.ryf[i] * K.iI[0]; K.rxfI[i] = K.rxf[i] * K.iI[1]; }
  K.tmaxI = K.tmax * K.iI[2];
  for (int k = 0; k < 3; ++k) { K.pqI[k] = K.pqr * K.iI[k]; K.dragM[k] = K.drag[k] * K.inv_mass; }
  const float gy = P.use_gyro_term ? 1.f : 0.f;
  K.gyI[0] = gy * K.iI[0] * (K.I[2] - K.I[1]); K.gyI[1] = gy * K.iI[1] * (K.I[0] - K.I[2]); K.gyI[2] = gy * K.iI[2] * (K.I[1] - K.I[0]);
  K.fmaxM = K.fmax * K.inv_mass;
  for (int k = 0; k < 4; ++k) K.start_quat[k] = P.start_quat[k];
  K.dome2 = P.dome * P.dome; K.goal_reach = P.goal_reach_distance; K.min_height = P.min_height;
  K.dome09m1 = P.dome * 0.9f - 1.0f;
  K.wp_dist_reward = P.wp_dist_reward; K.wp_yaw_penalty = P.wp_yaw_penalty;
  K.task_sparse = P.sparse_reward; K.angle_repr = P.angle_repr; K.num_targets = P.num_targets; K.max_steps = P.max_steps;
  K.env_step_ratio = P.env_step_ratio; K.settle_steps = P.settle_steps; K.tpc = P.ticks_per_control;
  K.noise_mode = P.noise_mode; K.autoreset = P.autoreset;
  K.seed_lo = (uint32_t)P.seed; K.seed_hi = (uint32_t)(P.seed >> 32);
  for (int k = 0; k < 4; ++k) { K.act_lo[k] = P.action_low[k]; K.act_span[k] = P.action_high[k] - P.action_low[k]; }
  K.mode = P.flight_mode;
  K.apw = P.agents_per_world > 1 ? P.agents_per_world : 1;
  {
    const int n_ticks = P.env_step_ratio * P.ticks_per_control;
    K.calm_T = n_ticks * P.dt;
    K.calm_TT = 0.5f * K.calm_T * (K.calm_T + P.dt);
    K.calm_T2 = P.ticks_per_control * P.dt;
    K.calm_TT2 = 0.5f * K.calm_T2 * (K.calm_T2 + P.dt);
    // the motor state follows t' = ((1 - a) t + a pwm) s with pwm in [0.05, 1], 0 < a <= 1, s = 1 + xi m_noise <= smax:
    // t' <= f(t) = ((1 - a) t + a) smax, an increasing contraction when (1 - a) smax < 1, with fixed point
    // t* = a smax / (1 - (1 - a) smax): t <= M and M >= t* give f(t) <= M, so the state never exceeds max(t_now, t*) however
    // many ticks follow. xi = num_motors + z (the reference's np_random.normal(*shape) quirk: mean 4), |z| <= 4.85 for the
    // device's normals (Box-Muller with the radius from a 16-bit uniform: uav_device.hpp) -> |xi| < 9
    const float smax = P.noise_mode == PF_NOISE_OFF ? 1.0f : 1.0f + 9.0f * __builtin_fabsf(P.motor_noise[0]);
    const float am = P.motor_dt_over_tau[0], contr = (1.0f - am) * smax;
    const float tstar = contr < 1.0f ? am * smax / (1.0f - contr) : INFINITY;
    K.calm_kt = 4.0f * K.fmaxM;
    K.calm_t2 = fmaxf(tstar * tstar, 1.0f);
    K.calm_c = fmaxf(fmaxf(__builtin_fabsf(K.dragM[0]), __builtin_fabsf(K.dragM[1])), __builtin_fabsf(K.dragM[2]));
    // (injected noise is unbounded; mode -1 hands the action to the motors unclipped; a shared world has the pair stage in its tick)
    K.calm_on = (P.contact_response && P.noise_mode != PF_NOISE_INJECT && P.flight_mode != -1 && K.apw == 1 && P.motor_dt_over_tau[0] <= 1.0f &&
                 K.calm_t2 < 1e6f && getenv("PF_NO_CALM_PATH") == nullptr) ? 1 : 0;
  }
  K.use_yaw = (P.task == PF_TASK_WAYPOINTS && P.use_yaw_targets) ? 1 : 0;
  K.goal_angle = P.goal_reach_angle;
  // level spawn at rest, far enough above the floor that the settle free-fall cannot touch it
  const float fall = 0.5f * 9.81f * (P.settle_steps * P.ticks_per_control * P.dt) * (P.settle_steps * P.ticks_per_control * P.dt);
  K.fast_settle = (P.start_quat[0] == 0.f && P.start_quat[1] == 0.f && P.start_vel[0] == 0.f && P.start_vel[1] == 0.f &&
                   P.start_vel[2] == 0.f && P.start_pos[2] - 2.0f * fall - 0.05f > K.bound_radius && P.gravity_z < 0.f)
                      ? 1 : 0;
  return K.fast_settle != 0;  // the hot kernel only implements the level-spawn settle recurrence
}

// Full 15-axis box test against the ground box, kept out of line: it runs only for waves that have
// a lane within one bounding radius of the floor.
__device__ __noinline__ PF_RARE_TEXT bool quad_floor_contact(float px, float py, float pz, quat q, float hx, float hy, float hz,
                                                float plane_xy, float plane_z, float rd) {
  m3 R = rot_from_quat(q);
  const float ha[3] = {hx, hy, hz};
  const float hb[3] = {plane_xy + rd, plane_xy + rd, plane_z + rd};  // (reported from the gap rd on: the slab enlarged by it)
  return box_overlaps_aabb(v3{px, py, pz}, R, ha, v3{0.f, 0.f, -plane_z}, hb);
}

// Controller memories of the outer loops (flight modes 1-7, quadx.py:437-479): state groups 7-11, the generic QuadX layout
// (uav_vehicles.hpp: QuadX::load / store).
// The tick's flight-path constants in VECTOR registers, for the calm ticks of the kernels that carry the contact response. With
// the solver's call in the kernel, the scalar register allocator parks some thirty of these constants in the lanes of a spill
// VGPR and fetches each one back with a v_readlane in front of every use -- 32 extra instructions per tick, 0.6 us per env step
// (profiles/README.md, r03). A uniform value held in a vector register costs the VALU instruction that reads it nothing.
struct QuadKV {
  float m_noise, m_a, ryfI[4], rxfI[4], tmaxI, pqI[3], gyI[3], dragM[3], fmaxM, bound_radius, plane_z, gravity_z, dt, vmax, half_dt;
};
PF_DEV float in_vgpr(float s) {
  float v;
  asm("v_mov_b32 %0, %1" : "=v"(v) : "s"(s));  // (opaque to the compiler: it cannot fold the copy back into the scalar register)
  return v;
}
PF_DEV QuadKV quadkv_from(const QuadK& K) {
  QuadKV V;
  V.m_noise = in_vgpr(K.m_noise); V.m_a = in_vgpr(K.m_a); V.tmaxI = in_vgpr(K.tmaxI); V.fmaxM = in_vgpr(K.fmaxM);
  for (int i = 0; i < 4; ++i) { V.ryfI[i] = in_vgpr(K.ryfI[i]); V.rxfI[i] = in_vgpr(K.rxfI[i]); }
  for (int i = 0; i < 3; ++i) { V.pqI[i] = in_vgpr(K.pqI[i]); V.gyI[i] = in_vgpr(K.gyI[i]); V.dragM[i] = in_vgpr(K.dragM[i]); }
  V.bound_radius = in_vgpr(K.bound_radius); V.plane_z = in_vgpr(K.plane_z); V.gravity_z = in_vgpr(K.gravity_z);
  V.dt = in_vgpr(K.dt); V.vmax = in_vgpr(K.vmax); V.half_dt = in_vgpr(K.half_dt);
  return V;
}
template <class MT>
struct QuadCascT {
  MT I1[3], E1[3];                // ang_pos
  MT I2[2], E2[2], I3[2], E3[2];  // lin_vel, lin_pos
  MT zI[2], zE[2];                // z_vel, z_pos
  PF_DEV void zero() {
#pragma unroll
    for (int k = 0; k < 3; ++k) I1[k] = E1[k] = (MT)0;
#pragma unroll
    for (int k = 0; k < 2; ++k) I2[k] = E2[k] = I3[k] = E3[k] = zI[k] = zE[k] = (MT)0;
  }
  // groups g0 .. g0 + 4 of the state: the eighteen memories, packed as the generic vehicle's groups 7-11 (+ the key word)
  PF_DEV void unpack(const float4 g7, const float4 g8, const float4 g9, const float4 g10, const float4 g11, const bool add) {
    const MT v[18] = {(MT)g7.x, (MT)g7.y, (MT)g7.z, (MT)g7.w, (MT)g8.x, (MT)g8.y, (MT)g8.z, (MT)g8.w, (MT)g9.x, (MT)g9.y, (MT)g9.z, (MT)g9.w,
                      (MT)g10.x, (MT)g10.y, (MT)g10.z, (MT)g10.w, (MT)g11.x, (MT)g11.y};
    MT* const d[18] = {&I1[0], &I1[1], &I1[2], &E1[0], &E1[1], &E1[2], &I2[0], &I2[1], &E2[0], &E2[1], &I3[0], &I3[1], &E3[0], &E3[1], &zI[0], &zI[1], &zE[0], &zE[1]};
#pragma unroll
    for (int k = 0; k < 18; ++k) *d[k] = add ? *d[k] + v[k] : v[k];
  }
  PF_DEV void load(const float4* S, size_t n, size_t i) {
    unpack(S[7 * n + i], S[8 * n + i], S[9 * n + i], S[10 * n + i], S[11 * n + i], false);
  }
  PF_DEV void store(float4* S, size_t n, size_t i, const uint32_t key_word = 0u) const {  // key_word: quadx_fast.hpp, QuadSpare
    S[7 * n + i] = float4{(float)I1[0], (float)I1[1], (float)I1[2], (float)E1[0]};
    S[8 * n + i] = float4{(float)E1[1], (float)E1[2], (float)I2[0], (float)I2[1]};
    S[9 * n + i] = float4{(float)E2[0], (float)E2[1], (float)I3[0], (float)I3[1]};
    S[10 * n + i] = float4{(float)E3[0], (float)E3[1], (float)zI[0], (float)zI[1]};
    S[11 * n + i] = float4{(float)zE[0], (float)zE[1], __int_as_float((int)key_word), 0.0f};
  }
  PF_DEV void through_hilo() {  // (MT = double; see QuadStateD::hilo)
    auto h = [](const MT x) { const float f = (float)x; return (MT)((double)f + (double)(float)((double)x - (double)f)); };
#pragma unroll
    for (int k = 0; k < 3; ++k) { I1[k] = h(I1[k]); E1[k] = h(E1[k]); }
#pragma unroll
    for (int k = 0; k < 2; ++k) { I2[k] = h(I2[k]); E2[k] = h(E2[k]); I3[k] = h(I3[k]); E3[k] = h(E3[k]); zI[k] = h(zI[k]); zE[k] = h(zE[k]); }
  }
  // (MT = double) the remainders: each memory minus its float32 rounding, the same packing in groups g0 .. g0 + 4
  PF_DEV void store_lo(float4* S, size_t n, size_t i, const int g0) const {
    auto lo = [](const MT x) { return (float)((double)x - (double)(float)x); };
    S[(size_t)(g0 + 0) * n + i] = float4{lo(I1[0]), lo(I1[1]), lo(I1[2]), lo(E1[0])};
    S[(size_t)(g0 + 1) * n + i] = float4{lo(E1[1]), lo(E1[2]), lo(I2[0]), lo(I2[1])};
    S[(size_t)(g0 + 2) * n + i] = float4{lo(E2[0]), lo(E2[1]), lo(I3[0]), lo(I3[1])};
    S[(size_t)(g0 + 3) * n + i] = float4{lo(E3[0]), lo(E3[1]), lo(zI[0]), lo(zI[1])};
    S[(size_t)(g0 + 4) * n + i] = float4{lo(zE[0]), lo(zE[1]), 0.0f, 0.0f};
  }
};
typedef QuadCascT<float> QuadCasc;

// ------------------------------------------------------------------------------------------
// The floor contact solve of THIS kernel's airframe, in registers (round 4). What quadk_from_params guarantees -- one collision box
// centred on the base origin, not yawed; centre of mass at the base origin; diagonal inertia -- makes the general solver's
// machinery (uav_vehicles.hpp: contact_solve_impl: parameter block through the scalar cache, records in LDS regions dealt out by a
// wave prefix sum, a count pass, sentinel records) unnecessary: with the manifold reduced to the incident face there are at most
// four contact points. In the env tasks a solve is a LONE lane on a lone wave and the launch waits for it (1.1 calls per 65 536-lane
// Waypoints launch; 19 k clocks each through the general solver: such a launch took twice as long as its other 1023 waves), and a
// lone wave issues ONE instruction of any kind per 4-5 clocks, dependent or not (profiles/r06/lone_wave_issue.txt; round 5's
// "~6.5 clocks per dependent instruction" was the instruction plus an s_nop): what counts is the NUMBER of instructions per row.
//   * rows r = (slot, direction); directions normal (+z), +x, +y. With the twist as (v, w~ = sqrt(I) w_body) and
//     J~_r = sqrt(I^-1) (a_body x e_r,body): row velocity u_r = v . e_r + w~ . J~_r, and the rows couple through
//     A_rs = J~_r . J~_s + [e_r = e_s] / m (the Delassus matrix; A_rr = the inverse effective mass);
//   * one or two contacts (nineteen solves in twenty): the sweep works on the rows themselves instead of on the twist. With impulses
//     in velocity units (lambda' = lambda A_rr, as in contact_solve_impl) and E_r = l'_r + target_r - u_r, the impulse row r would
//     take unclamped:  l'_new = clamp(E_r);  dl = l'_new - l'_r;  E_s -= (A_sr / A_rr) dl for the other rows -- a dependent chain
//     of THREE per row and 3 N - 1 independent multiply-adds, and dl itself is the row's velocity change, the quantity the residual
//     exit bounds. The twist is rebuilt from the impulses once, at the end. (Three and four contacts keep the twist form: their
//     coupling matrices, 72 / 132 entries, overflow the architectural registers into AGPR copies and scratch.)
//   * slots are DENSE over the wave: the N candidate vertices of the incident face that some lane of the wave has as a contact, in
//     vertex order (which they are is wave-uniform: scalar indices), N = 1 .. 4 a template parameter -- no skip branches in the
//     sweep, no work on vertices nobody touches with; a lane that lacks one of the wave's vertices has an inert slot there (zero
//     coupling, target -FLT_MAX: its rows move nothing), and the sweep order within a lane stays the vertex order;
//   * in a wave where several lanes solve side by side, a lane whose sweep met the residual bound is frozen (its dl forced to
//     zero) while the others go on; a lone lane runs the loop without that bookkeeping.
// The same model and the same sweep order as contact_solve_impl / the oracle (include/pyflyt_amd.h at pf_params.contact_response);
// the arithmetic is arranged differently, so the results differ in the last digits (the parity tests' impact tolerance).
// Returns the new (v, w) and the deepest penetration net of the slop. act: this lane asks; reach: how far above the face a vertex
// may be (margin / breaking distance); check_rim: test the vertices against the slab's rim as well (wave-uniform).
struct QuadFloorOut { v3 v, w; float deepest; };
// What only the solve reads of the contact model, fetched INSIDE the rare path (scalar cache): as members of the kernel-argument
// block these fourteen words were loaded in every wave's prologue and parked in the lanes of a spill VGPR for the solve that one
// wave in a thousand runs (round 5: 25 v_writelane and a scalar-load wait in front of every wave's resets).
struct QuadSolveC {
  float inv_dt, rest, mu, res, inv_mass;
  float sqI[3], siI[3];  // sqrt of the diagonal inertia and of its inverse: quad_floor_solve works on w~ = sqrt(I) w_body
  int iters;             // sweeps at most
};
// (sixteen words behind the device copy of the parameter block, written by pf_ctx_create: one scalar load; computing the derived
//  ones in the kernel -- a division, seven square roots -- made the solving wave, the one the launch waits for, 0.3 us longer)
constexpr size_t kQuadSolveOffset = (sizeof(pf_params) + 63) / 64 * 64;
constexpr int kQuadSolveWords = 16;
inline void quad_solve_words(const pf_params& P, float (&w)[kQuadSolveWords]) {  // host side
  w[0] = 1.0f / P.dt; w[1] = P.contact_restitution; w[2] = P.contact_friction; w[3] = sqrtf(P.contact_residual_threshold); w[4] = P.inv_mass;
  const float I[3] = {P.I_own[0], P.I_own[3], P.I_own[5]}, iI[3] = {P.I_inv[0], P.I_inv[3], P.I_inv[5]};
  for (int k = 0; k < 3; ++k) { w[5 + k] = sqrtf(I[k]); w[8 + k] = sqrtf(iI[k]); }
  const int32_t it = P.contact_iters;
  memcpy(&w[11], &it, 4);
  w[12] = w[13] = w[14] = w[15] = 0.0f;
}
PF_DEV QuadSolveC quad_solve_consts(const pf_params* Pg) {
  QuadSolveC c;
  typedef const float __attribute__((address_space(4)))* kfptr;
  const kfptr q = (kfptr)(uintptr_t)(reinterpret_cast<const char*>(Pg) + kQuadSolveOffset);
  c.inv_dt = q[0]; c.rest = q[1]; c.mu = q[2]; c.res = q[3]; c.inv_mass = q[4];
  c.sqI[0] = q[5]; c.sqI[1] = q[6]; c.sqI[2] = q[7]; c.siI[0] = q[8]; c.siI[1] = q[9]; c.siI[2] = q[10];
  c.iters = __float_as_int(q[11]);
  return c;
}
struct QuadFace {  // the incident face of the collision box at this pose (uav_vehicles.hpp: box_contact_vertices)
  bool use_y, use_z;
  float fs;
  PF_DEV void of(const m3& R) {
    // the axis with the largest |z component|, first on a tie. (The sign is selected among the COMPARISONS, not among the matrix
    // entries: a select between two members turns into a load through a selected address, which kept two rows of the caller's
    // rotation matrix in scratch memory.)
    const float ax = __builtin_fabsf(R.m20), ay = __builtin_fabsf(R.m21), az = __builtin_fabsf(R.m22);
    use_y = ay > ax; use_z = az > __builtin_fmaxf(ax, ay);
    const bool nx = R.m20 < 0.0f, ny = R.m21 < 0.0f, nz = R.m22 < 0.0f;
    fs = (use_z ? nz : (use_y ? ny : nx)) ? 1.0f : -1.0f;  // the face on the + side looks down when the axis points down
  }
  // vertex c (0 .. 3, wave-uniform) of the face in the body frame: the face axis carries fs, the other two axes, in index order, the bits of c
  PF_DEV v3 vertex(const int c, const float hx, const float hy, const float hz) const {
    const float b0 = (c & 1) ? 1.0f : -1.0f, b1 = (c & 2) ? 1.0f : -1.0f;
    const float sx = use_z ? b0 : (use_y ? b0 : fs);
    const float sy = use_z ? b1 : (use_y ? fs : b0);
    const float sz = use_z ? fs : b1;
    return v3{sx * hx, sy * hy, sz * hz};
  }
};
PF_DEV bool quad_vertex_touches(const QuadK& Kc, const v3 a, const m3& R, const v3 p, const float reach, const bool check_rim, float& z) {
  z = fmaf(a.x, R.m20, fmaf(a.y, R.m21, fmaf(a.z, R.m22, p.z)));
  bool is = z <= reach && z >= -2.0f * Kc.plane_z;
  if (check_rim) {
    const float x = fmaf(a.x, R.m00, fmaf(a.y, R.m01, fmaf(a.z, R.m02, p.x))), y = fmaf(a.x, R.m10, fmaf(a.y, R.m11, fmaf(a.z, R.m12, p.y)));
    is = is && __builtin_fabsf(x) <= Kc.plane_xy && __builtin_fabsf(y) <= Kc.plane_xy;
  }
  return is;
}
// cand: the wave's contact vertices, 4 bits per dense slot (wave-uniform)
template <int N>
PF_DEV QuadFloorOut quad_floor_solve_n(const QuadK& Kc, const QuadSolveC& Sc, const bool act, const float reach, const bool check_rim, const v3 p, const m3 R, const QuadFace F,
                                       const v3 v_in, const v3 w_in, const uint32_t cand) {
#ifdef PF_PHASE_TRACE
  const unsigned long long pf_q0 = __builtin_readcyclecounter();
#endif
  float Jt[N][3][3];     // J~ of every row
  // N <= 2 (nineteen solves in twenty): the row-velocity form below. N = 3, 4: its coupling matrix (45 / 78 entries) on top of this
  // kernel's own live state overflows the 256 architectural registers into AGPR copies and scratch -- those keep the twist form
  // (v, w~ updated row by row: 19 registers per contact, a dependent chain of eight per row).
  constexpr bool ROWFORM = N <= 2;
  constexpr int NR = ROWFORM ? 3 * N : 1;
  float B[NR][NR];       // B[s][r] = A_sr / A_rr: what row s's velocity changes by when row r's changes by one
  float kk[N][3];        // 1 / A_rr (effective mass)
  float u[N][3], lam[N][3], tg[N], fx[N], fy[N];
  bool on[N];
  const float hx = Kc.box_h[0], hy = Kc.box_h[1], hz = Kc.box_h[2];
  // (the model's constants in vector registers: as scalars they had been parked in the lanes of a spill VGPR and came back with a
  //  v_readlane in front of every use inside the sweep)
  const float im = in_vgpr(Sc.inv_mass), mu = in_vgpr(Sc.mu);
  const v3 wb = mulT(R, w_in);
  const v3 wt0{wb.x * Sc.sqI[0], wb.y * Sc.sqI[1], wb.z * Sc.sqI[2]};
  float deepest = 0.0f;
#pragma unroll
  for (int k = 0; k < N; ++k) {
    const v3 a = F.vertex((int)((cand >> (4 * k)) & 3u), hx, hy, hz);
    float z;
    const bool is = quad_vertex_touches(Kc, a, R, p, reach, check_rim, z) && act;
    on[k] = is;
    // row order: normal (+z), friction +x, friction +y; the world axis of the row in the body frame = a row of R, passed entry by
    // entry (a v3 built from three members of the matrix is a 12-byte copy out of it: that too kept those rows in scratch).
    // A lane that does not have this vertex gets an inert slot: zero coupling, target -FLT_MAX, zero impulse -- its rows move nothing.
#define PF_QROW(D_, EX_, EY_, EZ_, VAX_)                                                         \
    { const v3 j_{fmaf(a.y, EZ_, -(a.z * (EY_))), fmaf(a.z, EX_, -(a.x * (EZ_))), fmaf(a.x, EY_, -(a.y * (EX_)))};  \
      const v3 jt_{j_.x * Sc.siI[0], j_.y * Sc.siI[1], j_.z * Sc.siI[2]};                        \
      Jt[k][D_][0] = is ? jt_.x : 0.0f; Jt[k][D_][1] = is ? jt_.y : 0.0f; Jt[k][D_][2] = is ? jt_.z : 0.0f;  \
      kk[k][D_] = frcp(im + dot(jt_, jt_));                                                      \
      u[k][D_] = fmaf(wt0.x, jt_.x, fmaf(wt0.y, jt_.y, fmaf(wt0.z, jt_.z, VAX_)));               \
      lam[k][D_] = 0.0f; }
    PF_QROW(0, R.m20, R.m21, R.m22, v_in.z)
    PF_QROW(1, R.m00, R.m01, R.m02, v_in.x)
    PF_QROW(2, R.m10, R.m11, R.m12, v_in.y)
#undef PF_QROW
    const float depth = -z;
    // normal row: may close the gap down to the slop, no more; otherwise towards restitution x approach speed
    const float t = depth < Kc.slop ? (depth - Kc.slop) * Sc.inv_dt : (u[k][0] < 0.0f ? -Sc.rest * u[k][0] : 0.0f);
    tg[k] = is ? t : -3.4028235e38f;
    // friction cone in velocity units: |l'_x| <= mu (A_xx / A_zz) l'_z
    fx[k] = mu * kk[k][0] * frcp(kk[k][1]); fy[k] = mu * kk[k][0] * frcp(kk[k][2]);
    deepest = is ? __builtin_fmaxf(deepest, depth) : deepest;
  }
  // the couplings A_sr = J~_s . J~_r + [same direction] / m, scaled by the moving row's effective mass (zero towards / from an inert
  // slot: its J~ is zero, and so is made the translational part -- which otherwise couples all rows of the same direction)
#pragma unroll
  for (int cs = 0; cs < (ROWFORM ? N : 0); ++cs)
#pragma unroll
    for (int cr = 0; cr < N; ++cr) {
      const bool both = on[cs] && on[cr];
#pragma unroll
      for (int ds = 0; ds < 3; ++ds)
#pragma unroll
        for (int dr = 0; dr < 3; ++dr) {
          if (cs == cr && ds == dr) continue;  // (the row itself: not stored)
          const float jj = fmaf(Jt[cs][ds][0], Jt[cr][dr][0], fmaf(Jt[cs][ds][1], Jt[cr][dr][1], Jt[cs][ds][2] * Jt[cr][dr][2]));
          const float a_sr = ds == dr ? jj + im : jj;
          B[ROWFORM ? 3 * cs + ds : 0][ROWFORM ? 3 * cr + dr : 0] = both ? a_sr * kk[cr][dr] : 0.0f;
        }
    }
  // The row-velocity form keeps, per row, E_r = l'_r + target_r - u_r: the impulse the row would take if it were not clamped.
  // Moving row r by dl leaves E_r where it is (l'_r and u_r both move by dl) and shifts every other E_s by -B_sr dl, so a row is
  //   l'_new = clamp(E_r);  dl = l'_new - l'_r;  E_s -= B_sr dl  (s != r)
  // -- a dependent chain of THREE per row (clamp, subtract, multiply-add into the next row's E).
  if (ROWFORM) {
#pragma unroll
    for (int k = 0; k < N; ++k) { u[k][0] = tg[k] - u[k][0]; u[k][1] = -u[k][1]; u[k][2] = -u[k][2]; }  // (u now holds E)
  }
  bool any_on = false;
#pragma unroll
  for (int k = 0; k < N; ++k) any_on = any_on || on[k];
  bool done = !any_on;
#ifdef PF_PHASE_TRACE
  const unsigned long long pf_q1 = __builtin_readcyclecounter();
  int pf_sweeps = 0;
#endif
  float res = 0.0f;
  v3 vc = v_in, wt = wt0;  // (the twist form's running twist)
  float ki[N][3];          // (the twist form: A_rr, for the residual)
  if (!ROWFORM) {
#pragma unroll
    for (int k = 0; k < N; ++k)
#pragma unroll
      for (int d = 0; d < 3; ++d) { ki[k][d] = on[k] ? frcp(kk[k][d]) : 0.0f; kk[k][d] = on[k] ? kk[k][d] : 0.0f; }
  }
  // one sweep; FREEZE: several lanes solve side by side, a done lane's rows must move nothing
  auto sweep = [&](auto freeze_tag) {
    constexpr bool FREEZE = decltype(freeze_tag)::value;
    float r0 = 0.0f, r1 = 0.0f;
#pragma unroll
    for (int c = 0; c < N; ++c) {
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        if (ROWFORM) {
          float nl;
          if (d == 0) nl = __builtin_fmaxf(u[c][0], 0.0f);
          else { const float lim = (d == 1 ? fx[c] : fy[c]) * lam[c][0]; nl = med3(u[c][d], -lim, lim); }
          float dl = nl - lam[c][d];
          if (FREEZE) dl = done ? 0.0f : dl;
          lam[c][d] = FREEZE ? lam[c][d] + dl : nl;
#pragma unroll
          for (int cs = 0; cs < N; ++cs)
#pragma unroll
            for (int ds = 0; ds < 3; ++ds)
              if (!(cs == c && ds == d)) u[cs][ds] = fmaf(-B[ROWFORM ? 3 * cs + ds : 0][ROWFORM ? 3 * c + d : 0], dl, u[cs][ds]);
          if ((3 * c + d) & 1) r1 = __builtin_fmaxf(r1, __builtin_fabsf(dl)); else r0 = __builtin_fmaxf(r0, __builtin_fabsf(dl));
        } else {  // impulses in impulse units here (lam), the twist carried along
          const float vax = d == 0 ? vc.z : (d == 1 ? vc.x : vc.y);
          const float ur = fmaf(wt.x, Jt[c][d][0], fmaf(wt.y, Jt[c][d][1], fmaf(wt.z, Jt[c][d][2], vax)));
          float nl;
          if (d == 0) nl = __builtin_fmaxf(fmaf(tg[c] - ur, kk[c][0], lam[c][0]), 0.0f);
          else { const float lim = mu * lam[c][0]; nl = med3(fmaf(-ur, kk[c][d], lam[c][d]), -lim, lim); }
          float dl = nl - lam[c][d];
          if (FREEZE) dl = done ? 0.0f : dl;
          lam[c][d] = FREEZE ? lam[c][d] + dl : nl;
          if (d == 0) vc.z = fmaf(im, dl, vc.z); else if (d == 1) vc.x = fmaf(im, dl, vc.x); else vc.y = fmaf(im, dl, vc.y);
          wt = v3{fmaf(dl, Jt[c][d][0], wt.x), fmaf(dl, Jt[c][d][1], wt.y), fmaf(dl, Jt[c][d][2], wt.z)};
          const float rv = __builtin_fabsf(dl) * ki[c][d];
          if ((3 * c + d) & 1) r1 = __builtin_fmaxf(r1, rv); else r0 = __builtin_fmaxf(r0, rv);
        }
      }
    }
    res = __builtin_fmaxf(r0, r1);
  };
  const bool lone = __popcll(__ballot(any_on)) <= 1;  // (wave-uniform)
  if (lone && ROWFORM && N == 1) {
    // ONE contact on a lone lane -- four solves in five -- with the sweep loop written out. What the compiler made of the loop below
    // was 49 instructions a sweep for a dependent chain of ten: twelve v_readlane / four v_writelane of scalars it had spilled around
    // the loop (the trip counter among them), copies of the loop-carried impulses, a compare-select-compare for the exit test -- 325
    // clocks a sweep on the one wave the whole launch waits for (profiles/r05/solver_trace.txt). The same instructions on the same
    // values in the same order, 22 a sweep: bit-identical impulses, the same number of sweeps.
    float e0 = u[0][0], e1 = u[0][1], e2 = u[0][2], l0 = 0.0f, l1 = 0.0f, l2 = 0.0f, n0_, n1_, n2_, lim_, d0_, d1_, d2_;
    // (round 6, second pass: two sweeps per trip, the impulses alternating between two register sets -- no copy of the new impulse over
    //  the old one --, the count-down's borrow as the exit test: 20 instructions a sweep; profiles/tools/r06/gen_solve.py writes the text)
    int cnt = Sc.iters - 1;
    const unsigned long long onm = __ballot(any_on);
    asm volatile(
        "1:\n\t"
        "v_max_f32 %[n0], 0, %[e0]\n\t"
        "v_sub_f32 %[d0], %[n0], %[l0]\n\t"
        "v_fma_f32 %[e1], -%[b10], %[d0], %[e1]\n\t"
        "v_fma_f32 %[e2], -%[b20], %[d0], %[e2]\n\t"
        "v_mul_f32 %[lim], %[fx], %[n0]\n\t"
        "v_med3_f32 %[n1], %[e1], -%[lim], %[lim]\n\t"
        "v_sub_f32 %[d1], %[n1], %[l1]\n\t"
        "v_fma_f32 %[e2], -%[b21], %[d1], %[e2]\n\t"
        "v_fma_f32 %[e0], -%[b01], %[d1], %[e0]\n\t"
        "v_mul_f32 %[lim], %[fy], %[n0]\n\t"
        "v_med3_f32 %[n2], %[e2], -%[lim], %[lim]\n\t"
        "v_sub_f32 %[d2], %[n2], %[l2]\n\t"
        "v_fma_f32 %[e0], -%[b02], %[d2], %[e0]\n\t"
        "v_fma_f32 %[e1], -%[b12], %[d2], %[e1]\n\t"
        "v_max3_f32 %[lim], |%[d0]|, |%[d1]|, |%[d2]|\n\t"
        "v_cmp_lt_f32 vcc, %[bound], %[lim]\n\t"
        "s_and_b64 vcc, vcc, %[on]\n\t"
        "s_sub_u32 %[cnt], %[cnt], 1\n\t"
        "s_cbranch_vccz 3f\n\t"
        "s_cbranch_scc1 3f\n\t"
        "v_max_f32 %[l0], 0, %[e0]\n\t"
        "v_sub_f32 %[d0], %[l0], %[n0]\n\t"
        "v_fma_f32 %[e1], -%[b10], %[d0], %[e1]\n\t"
        "v_fma_f32 %[e2], -%[b20], %[d0], %[e2]\n\t"
        "v_mul_f32 %[lim], %[fx], %[l0]\n\t"
        "v_med3_f32 %[l1], %[e1], -%[lim], %[lim]\n\t"
        "v_sub_f32 %[d1], %[l1], %[n1]\n\t"
        "v_fma_f32 %[e2], -%[b21], %[d1], %[e2]\n\t"
        "v_fma_f32 %[e0], -%[b01], %[d1], %[e0]\n\t"
        "v_mul_f32 %[lim], %[fy], %[l0]\n\t"
        "v_med3_f32 %[l2], %[e2], -%[lim], %[lim]\n\t"
        "v_sub_f32 %[d2], %[l2], %[n2]\n\t"
        "v_fma_f32 %[e0], -%[b02], %[d2], %[e0]\n\t"
        "v_fma_f32 %[e1], -%[b12], %[d2], %[e1]\n\t"
        "v_max3_f32 %[lim], |%[d0]|, |%[d1]|, |%[d2]|\n\t"
        "v_cmp_lt_f32 vcc, %[bound], %[lim]\n\t"
        "s_and_b64 vcc, vcc, %[on]\n\t"
        "s_sub_u32 %[cnt], %[cnt], 1\n\t"
        "s_cbranch_vccz 2f\n\t"
        "s_cbranch_scc0 1b\n\t"
        "s_branch 2f\n\t"
        "3:\n\t"
        "v_mov_b32 %[l0], %[n0]\n\t"
        "v_mov_b32 %[l1], %[n1]\n\t"
        "v_mov_b32 %[l2], %[n2]\n\t"
        "2:"
        : [e0] "+v"(e0), [e1] "+v"(e1), [e2] "+v"(e2), [l0] "+v"(l0), [l1] "+v"(l1), [l2] "+v"(l2), [cnt] "+s"(cnt), [n0] "=&v"(n0_), [n1] "=&v"(n1_), [n2] "=&v"(n2_),
          [lim] "=&v"(lim_), [d0] "=&v"(d0_), [d1] "=&v"(d1_), [d2] "=&v"(d2_)
        : [b10] "v"(B[ROWFORM ? 1 : 0][0]), [b20] "v"(B[ROWFORM ? 2 : 0][0]), [b01] "v"(B[0][ROWFORM ? 1 : 0]), [b21] "v"(B[ROWFORM ? 2 : 0][ROWFORM ? 1 : 0]),
          [b02] "v"(B[0][ROWFORM ? 2 : 0]), [b12] "v"(B[ROWFORM ? 1 : 0][ROWFORM ? 2 : 0]), [fx] "v"(fx[0]), [fy] "v"(fy[0]), [bound] "s"(Sc.res), [on] "s"(onm)
        : "vcc", "scc");
    lam[0][0] = l0; lam[0][1] = l1; lam[0][2] = l2;
#ifdef PF_PHASE_TRACE
    pf_sweeps = Sc.iters - 1 - cnt;
#endif
  } else if (lone && ROWFORM && N == 2) {
    // TWO contacts on a lone lane (an edge impact: one solve in six, and the ones that run into the sweep cap): the same treatment, six
    // rows, 54 instructions a sweep for the 107 the compiler's loop took (447 clocks: profiles/r05/solver_trace.txt). Each row's five
    // coupling updates start with the NEXT row's, the one the dependent chain waits for.
    // (round 6, second pass) The six rows' velocities as three aligned register pairs, every row's five coupling updates as three
    // v_pk_fma_f32: two whole pairs, and the pair the row itself sits in with (-0) x 1.0 on its own half -- e + (-0) = e for every e,
    // the zeros included, so the row passes through untouched. The step d of a row is the low half of a (d, 1.0) pair. A 32-bit
    // instruction cannot name half of a 64-bit asm operand, hence FIXED registers for those ten (v232 .. v241: the allocator vacates
    // them around this block, on the rare path). Two sweeps per trip as above. 42 instructions a sweep for round 6's first 63: the same
    // operations on the same values (a packed fma is two fmas), bit-identical impulses. profiles/tools/r06/gen_solve.py writes the text.
#define BB(s_, r_) B[ROWFORM ? (s_) : 0][ROWFORM ? (r_) : 0]
    f2 P0 = f2{u[0][0], u[0][1]}, P1 = f2{u[0][2], u[N > 1 ? 1 : 0][0]}, P2 = f2{u[N > 1 ? 1 : 0][1], u[N > 1 ? 1 : 0][2]};
    float l_[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, n_[6], t_, lim_;
    int cnt = Sc.iters - 1;
    const unsigned long long onm = __ballot(any_on);
    asm volatile(
        "v_mov_b32 v239, 1.0\n\t"
        "v_mov_b32 v241, 1.0\n\t"
        "1:\n\t"
        "v_max_f32 %[n0], 0, v232\n\t"
        "v_sub_f32 v238, %[n0], %[l0]\n\t"
        "v_pk_fma_f32 v[232:233], %[bh0], v[238:239], v[232:233] op_sel:[0,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"
        "v_pk_fma_f32 v[234:235], %[bw01], v[238:239], v[234:235] op_sel_hi:[1,0,1] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"
        "v_pk_fma_f32 v[236:237], %[bw02], v[238:239], v[236:237] op_sel_hi:[1,0,1] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"
        "v_mul_f32 %[lim], %[fx0], %[n0]\n\t"
        "v_med3_f32 %[n1], v233, -%[lim], %[lim]\n\t"
        "v_sub_f32 v240, %[n1], %[l1]\n\t"
        "v_pk_fma_f32 v[234:235], %[bw11], v[240:241], v[234:235] op_sel_hi:[1,0,1] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"
        "v_pk_fma_f32 v[232:233], %[bh1], v[240:241], v[232:233] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"
        "v_pk_fma_f32 v[236:237], %[bw12], v[240:241], v[236:237] op_sel_hi:[1,0,1] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"
        "v_max_f32 %[t], |v238|, |v240|\n\t"
        "v_mul_f32 %[lim], %[fy0], %[n0]\n\t"
        "v_med3_f32 %[n2], v234, -%[lim], %[lim]\n\t"
        "v_sub_f32 v238, %[n2], %[l2]\n\t"
        "v_pk_fma_f32 v[234:235], %[bh2], v[238:239], v[234:235] op_sel:[0,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"
        "v_pk_fma_f32 v[232:233], %[bw20], v[238:239], v[232:233] op_sel_hi:[1,0,1] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"
        "v_pk_fma_f32 v[236:237], %[bw22], v[238:239], v[236:237] op_sel_hi:[1,0,1] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"
        "v_max_f32 %[n3], 0, v235\n\t"
        "v_sub_f32 v240, %[n3], %[l3]\n\t"
        "v_pk_fma_f32 v[236:237], %[bw32], v[240:241], v[236:237] op_sel_hi:[1,0,1] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"
        "v_pk_fma_f32 v[232:233], %[bw30], v[240:241], v[232:233] op_sel_hi:[1,0,1] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"
        "v_pk_fma_f32 v[234:235], %[bh3], v[240:241], v[234:235] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"
        "v_max3_f32 %[t], %[t], |v238|, |v240|\n\t"
        "v_mul_f32 %[lim], %[fx1], %[n3]\n\t"
        "v_med3_f32 %[n4], v236, -%[lim], %[lim]\n\t"
        "v_sub_f32 v238, %[n4], %[l4]\n\t"
        "v_pk_fma_f32 v[236:237], %[bh4], v[238:239], v[236:237] op_sel:[0,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"
        "v_pk_fma_f32 v[232:233], %[bw40], v[238:239], v[232:233] op_sel_hi:[1,0,1] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"
        "v_pk_fma_f32 v[234:235], %[bw41], v[238:239], v[234:235] op_sel_hi:[1,0,1] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"
        "v_mul_f32 %[lim], %[fy1], %[n3]\n\t"
        "v_med3_f32 %[n5], v237, -%[lim], %[lim]\n\t"
        "v_sub_f32 v240, %[n5], %[l5]\n\t"
        "v_pk_fma_f32 v[232:233], %[bw50], v[240:241], v[232:233] op_sel_hi:[1,0,1] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"
        "v_pk_fma_f32 v[234:235], %[bw51], v[240:241], v[234:235] op_sel_hi:[1,0,1] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"
        "v_pk_fma_f32 v[236:237], %[bh5], v[240:241], v[236:237] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"
        "v_max3_f32 %[t], %[t], |v238|, |v240|\n\t"
        "v_cmp_lt_f32 vcc, %[bound], %[t]\n\t"
        "s_and_b64 vcc, vcc, %[on]\n\t"
        "s_sub_u32 %[cnt], %[cnt], 1\n\t"
        "s_cbranch_vccz 3f\n\t"
        "s_cbranch_scc1 3f\n\t"
        "v_max_f32 %[l0], 0, v232\n\t"
        "v_sub_f32 v238, %[l0], %[n0]\n\t"
        "v_pk_fma_f32 v[232:233], %[bh0], v[238:239], v[232:233] op_sel:[0,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"
        "v_pk_fma_f32 v[234:235], %[bw01], v[238:239], v[234:235] op_sel_hi:[1,0,1] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"
        "v_pk_fma_f32 v[236:237], %[bw02], v[238:239], v[236:237] op_sel_hi:[1,0,1] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"
        "v_mul_f32 %[lim], %[fx0], %[l0]\n\t"
        "v_med3_f32 %[l1], v233, -%[lim], %[lim]\n\t"
        "v_sub_f32 v240, %[l1], %[n1]\n\t"
        "v_pk_fma_f32 v[234:235], %[bw11], v[240:241], v[234:235] op_sel_hi:[1,0,1] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"
        "v_pk_fma_f32 v[232:233], %[bh1], v[240:241], v[232:233] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"
        "v_pk_fma_f32 v[236:237], %[bw12], v[240:241], v[236:237] op_sel_hi:[1,0,1] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"
        "v_max_f32 %[t], |v238|, |v240|\n\t"
        "v_mul_f32 %[lim], %[fy0], %[l0]\n\t"
        "v_med3_f32 %[l2], v234, -%[lim], %[lim]\n\t"
        "v_sub_f32 v238, %[l2], %[n2]\n\t"
        "v_pk_fma_f32 v[234:235], %[bh2], v[238:239], v[234:235] op_sel:[0,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"
        "v_pk_fma_f32 v[232:233], %[bw20], v[238:239], v[232:233] op_sel_hi:[1,0,1] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"
        "v_pk_fma_f32 v[236:237], %[bw22], v[238:239], v[236:237] op_sel_hi:[1,0,1] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"
        "v_max_f32 %[l3], 0, v235\n\t"
        "v_sub_f32 v240, %[l3], %[n3]\n\t"
        "v_pk_fma_f32 v[236:237], %[bw32], v[240:241], v[236:237] op_sel_hi:[1,0,1] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"
        "v_pk_fma_f32 v[232:233], %[bw30], v[240:241], v[232:233] op_sel_hi:[1,0,1] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"
        "v_pk_fma_f32 v[234:235], %[bh3], v[240:241], v[234:235] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"
        "v_max3_f32 %[t], %[t], |v238|, |v240|\n\t"
        "v_mul_f32 %[lim], %[fx1], %[l3]\n\t"
        "v_med3_f32 %[l4], v236, -%[lim], %[lim]\n\t"
        "v_sub_f32 v238, %[l4], %[n4]\n\t"
        "v_pk_fma_f32 v[236:237], %[bh4], v[238:239], v[236:237] op_sel:[0,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"
        "v_pk_fma_f32 v[232:233], %[bw40], v[238:239], v[232:233] op_sel_hi:[1,0,1] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"
        "v_pk_fma_f32 v[234:235], %[bw41], v[238:239], v[234:235] op_sel_hi:[1,0,1] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"
        "v_mul_f32 %[lim], %[fy1], %[l3]\n\t"
        "v_med3_f32 %[l5], v237, -%[lim], %[lim]\n\t"
        "v_sub_f32 v240, %[l5], %[n5]\n\t"
        "v_pk_fma_f32 v[232:233], %[bw50], v[240:241], v[232:233] op_sel_hi:[1,0,1] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"
        "v_pk_fma_f32 v[234:235], %[bw51], v[240:241], v[234:235] op_sel_hi:[1,0,1] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"
        "v_pk_fma_f32 v[236:237], %[bh5], v[240:241], v[236:237] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"
        "v_max3_f32 %[t], %[t], |v238|, |v240|\n\t"
        "v_cmp_lt_f32 vcc, %[bound], %[t]\n\t"
        "s_and_b64 vcc, vcc, %[on]\n\t"
        "s_sub_u32 %[cnt], %[cnt], 1\n\t"
        "s_cbranch_vccz 2f\n\t"
        "s_cbranch_scc0 1b\n\t"
        "s_branch 2f\n\t"
        "3:\n\t"
        "v_mov_b32 %[l0], %[n0]\n\t"
        "v_mov_b32 %[l1], %[n1]\n\t"
        "v_mov_b32 %[l2], %[n2]\n\t"
        "v_mov_b32 %[l3], %[n3]\n\t"
        "v_mov_b32 %[l4], %[n4]\n\t"
        "v_mov_b32 %[l5], %[n5]\n\t"
        "2:"
        : [p0] "+{v[232:233]}"(P0), [p1] "+{v[234:235]}"(P1), [p2] "+{v[236:237]}"(P2), [l0] "+v"(l_[0]), [l1] "+v"(l_[1]), [l2] "+v"(l_[2]), [l3] "+v"(l_[3]), [l4] "+v"(l_[4]), [l5] "+v"(l_[5]),
          [n0] "=&v"(n_[0]), [n1] "=&v"(n_[1]), [n2] "=&v"(n_[2]), [n3] "=&v"(n_[3]), [n4] "=&v"(n_[4]), [n5] "=&v"(n_[5]), [cnt] "+s"(cnt), [t] "=&v"(t_), [lim] "=&v"(lim_)
        : [bh0] "v"(f2{0.0f, BB(1, 0)}), [bw01] "v"(f2{BB(2, 0), BB(3, 0)}), [bw02] "v"(f2{BB(4, 0), BB(5, 0)}),
          [bh1] "v"(f2{BB(0, 1), 0.0f}), [bw11] "v"(f2{BB(2, 1), BB(3, 1)}), [bw12] "v"(f2{BB(4, 1), BB(5, 1)}),
          [bh2] "v"(f2{0.0f, BB(3, 2)}), [bw20] "v"(f2{BB(0, 2), BB(1, 2)}), [bw22] "v"(f2{BB(4, 2), BB(5, 2)}),
          [bh3] "v"(f2{BB(2, 3), 0.0f}), [bw30] "v"(f2{BB(0, 3), BB(1, 3)}), [bw32] "v"(f2{BB(4, 3), BB(5, 3)}),
          [bh4] "v"(f2{0.0f, BB(5, 4)}), [bw40] "v"(f2{BB(0, 4), BB(1, 4)}), [bw41] "v"(f2{BB(2, 4), BB(3, 4)}),
          [bh5] "v"(f2{BB(4, 5), 0.0f}), [bw50] "v"(f2{BB(0, 5), BB(1, 5)}), [bw51] "v"(f2{BB(2, 5), BB(3, 5)}),
          [fx0] "v"(fx[0]), [fy0] "v"(fy[0]), [fx1] "v"(fx[N > 1 ? 1 : 0]), [fy1] "v"(fy[N > 1 ? 1 : 0]), [bound] "s"(Sc.res), [on] "s"(onm)
        : "vcc", "scc", "v238", "v239", "v240", "v241");
#undef BB
#pragma unroll
    for (int d = 0; d < 3; ++d) { lam[0][d] = l_[d]; lam[N > 1 ? 1 : 0][d] = l_[3 + d]; }
#ifdef PF_PHASE_TRACE
    pf_sweeps = Sc.iters - 1 - cnt;
#endif
  } else if (lone) {
    for (int it = 0; it < Sc.iters; ++it) {
#ifdef PF_PHASE_TRACE
      pf_sweeps = it + 1;
#endif
      sweep(std::false_type{});
      if (__ballot(any_on && res > Sc.res) == 0ull) break;
    }
  } else {
    for (int it = 0; it < Sc.iters; ++it) {
#ifdef PF_PHASE_TRACE
      pf_sweeps = it + 1;
#endif
      sweep(std::true_type{});
      done = done || !(res > Sc.res);
      if (__ballot(!done) == 0ull) break;
    }
  }
  // the row-velocity form: the twist from the impulses (impulse = l' / A_rr)
  if (ROWFORM) {
#pragma unroll
    for (int c = 0; c < N; ++c) {
      const float lz = on[c] ? lam[c][0] * kk[c][0] : 0.0f, lx = on[c] ? lam[c][1] * kk[c][1] : 0.0f, ly = on[c] ? lam[c][2] * kk[c][2] : 0.0f;
      vc = v3{fmaf(im, lx, vc.x), fmaf(im, ly, vc.y), fmaf(im, lz, vc.z)};
      wt = v3{fmaf(lz, Jt[c][0][0], fmaf(lx, Jt[c][1][0], fmaf(ly, Jt[c][2][0], wt.x))),
              fmaf(lz, Jt[c][0][1], fmaf(lx, Jt[c][1][1], fmaf(ly, Jt[c][2][1], wt.y))),
              fmaf(lz, Jt[c][0][2], fmaf(lx, Jt[c][1][2], fmaf(ly, Jt[c][2][2], wt.z)))};
    }
  }
  const v3 wbn{wt.x * Sc.siI[0], wt.y * Sc.siI[1], wt.z * Sc.siI[2]};
#ifdef PF_PHASE_TRACE
  {  // (diagnostic build: the same counters as the general solver's -- calls, clocks in the records / in the sweeps, contacts, lanes, sweeps)
    const unsigned long long pf_q2 = __builtin_readcyclecounter();
    const int first = __ffsll((long long)__ballot(1)) - 1;
    int nc = 0;
#pragma unroll
    for (int k = 0; k < N; ++k) nc += (int)on[k];
    const int lanes = __popcll(__ballot(any_on));
    for (int o = 32; o > 0; o >>= 1) nc = max(nc, __shfl_xor(nc, o));
    if ((int)(threadIdx.x & 63u) == first) {
      trace_add(&g_solver_trace[0], 1ull);
      trace_add(&g_solver_trace[6], pf_q1 - pf_q0);
      trace_add(&g_solver_trace[2], pf_q2 - pf_q1);
      trace_add(&g_solver_trace[3], (unsigned long long)nc);
      trace_add(&g_solver_trace[4], (unsigned long long)lanes);
      trace_add(&g_solver_trace[5], (unsigned long long)pf_sweeps);
    }
  }
#endif
  QuadFloorOut o;
  o.v = any_on ? vc : v_in;
  o.w = any_on ? mul(R, wbn) : w_in;
  o.deepest = any_on ? __builtin_fmaxf(deepest - Kc.slop, 0.0f) : 0.0f;
  return o;
}
// (the rotation matrix entry by entry: as an m3 argument the 36-byte copy kept two rows of the caller's matrix in scratch memory)
PF_DEV QuadFloorOut quad_floor_solve(const QuadK& Kc, const pf_params* Pfull, const bool act, const float reach, const bool check_rim, const v3 p,
                                     const float r00, const float r01, const float r02, const float r10, const float r11, const float r12,
                                     const float r20, const float r21, const float r22, const v3 v_in, const v3 w_in) {
  const m3 R{r00, r01, r02, r10, r11, r12, r20, r21, r22};
  const QuadSolveC Sc = quad_solve_consts(Pfull);
  QuadFace F;
  F.of(R);
  // which of the face's four vertices does some lane of the wave have as a contact? (wave-uniform: the dense slots)
  uint32_t cand = 0u;
  int n = 0;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    float z;
    const bool is = act && quad_vertex_touches(Kc, F.vertex(c, Kc.box_h[0], Kc.box_h[1], Kc.box_h[2]), R, p, reach, check_rim, z);
    if (__any(is)) { cand |= (uint32_t)c << (4 * n); n += 1; }
  }
  switch (n) {
    case 1: return quad_floor_solve_n<1>(Kc, Sc, act, reach, check_rim, p, R, F, v_in, w_in, cand);
    case 2: return quad_floor_solve_n<2>(Kc, Sc, act, reach, check_rim, p, R, F, v_in, w_in, cand);
    case 3: return quad_floor_solve_n<3>(Kc, Sc, act, reach, check_rim, p, R, F, v_in, w_in, cand);
    case 4: return quad_floor_solve_n<4>(Kc, Sc, act, reach, check_rim, p, R, F, v_in, w_in, cand);
    default: break;
  }
  return QuadFloorOut{v_in, w_in, 0.0f};
}


// The rigid-body state of the cascaded flight modes in fp64 (round 6; the MODES instantiations without a shared world). The outer
// loops of modes 4-7 differentiate positions and velocities (k_d / T = 60 per control tick) and hand the result down three more PIDs:
// whatever is rounded to float32 on the way -- state, derived quantities, PID internals -- comes back amplified a thousandfold over an
// episode (tests/tools/fp32_rounding_sites.py: the fp64 oracle with ONLY the state rounded after every tick replays the mode-7 fixture
// 2.2e-4 away from itself, with float32 parameters and fp64 arithmetic 2.5e-5, with float32 motors alone 3e-5). Round 5 moved the
// controller to fp64 (quadx_control_d.hpp); the state it read was still float32: 1.7e-3 on that fixture. Here the master copy of
// (p, q, v, w) is fp64 -- state groups 0-3 hold its float32 rounding as before, groups 16-19 the remainders -- and the tick's
// arithmetic from the motor thrusts on (which stay float32) is the oracle's (oracle/uav_oracle.c: rigid_tick_vel / rigid_tick_pos)
// for this airframe: centre of mass on the base origin, diagonal inertia. The float32 members of QuadHot are kept as the ROUNDED VIEW
// of it: the floor tests, the contact solve, the observation and the task's tests read them as before. fp64 vector instructions issue
// at the float32 rate on gfx950, the mode-0 instantiations do not contain this code, and no BASELINE config runs these modes.
struct QuadStateD {
  double p[3], q[4], v[3], w[3];
  double R[9], wb[3], vb[3];  // update_state (quadx.py:512-535): rotation matrix of q, R^T w, R^T v
  double rI[3], rE[3];        // the rate PID's memories (QuadHot::I / E are their float32 view)
  double thr[4], pwm[4];      // the motors' states (QuadHot::t01 / t23 their view) and commands
  PF_DEV void derive() {  // btMatrix3x3::setRotation
    const double x = q[0], y = q[1], z = q[2], ww = q[3];
    const double d = x * x + y * y + z * z + ww * ww, s = 2.0 * rcp_d(d);  // (quadx_control_d.hpp: rcp_d)
    const double xs = x * s, ys = y * s, zs = z * s;
    const double wx = ww * xs, wy = ww * ys, wz = ww * zs, xx = x * xs, xy = x * ys, xz = x * zs, yy = y * ys, yz = y * zs, zz = z * zs;
    R[0] = 1.0 - (yy + zz); R[1] = xy - wz; R[2] = xz + wy; R[3] = xy + wz; R[4] = 1.0 - (xx + zz); R[5] = yz - wx;
    R[6] = xz - wy; R[7] = yz + wx; R[8] = 1.0 - (xx + yy);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      wb[k] = R[k] * w[0] + R[3 + k] * w[1] + R[6 + k] * w[2];
      vb[k] = R[k] * v[0] + R[3 + k] * v[1] + R[6 + k] * v[2];
    }
  }
  // (the controller's inputs: wb / vb are what derive() left -- the same formula on the same q, v, w as quad_ctl_inputs_d's, which the
  //  first build evaluated a second time, 60 fp64 instructions a control update --, the Euler angles from q)
  PF_DEV QuadCtlIn ctl_inputs() const {
    QuadCtlIn o;
#pragma unroll
    for (int k = 0; k < 3; ++k) { o.wb[k] = wb[k]; o.vb[k] = vb[k]; o.p[k] = p[k]; }
    quad_ctl_euler_d(q, rcp_d(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]), o);
    return o;
  }
  // what survives a store / load through the state groups: the float32 rounding + the float32 rounding of the remainder (48 of a
  // double's 53 bits). pf_rollout passes its resident copy through this after every env step, so that it equals k launches bit for bit.
  PF_DEV static double hilo(const double x) { const float h = (float)x; return (double)h + (double)(float)(x - (double)h); }
  PF_DEV void through_hilo() {
#pragma unroll
    for (int k = 0; k < 3; ++k) { p[k] = hilo(p[k]); v[k] = hilo(v[k]); w[k] = hilo(w[k]); rI[k] = hilo(rI[k]); rE[k] = hilo(rE[k]); }
#pragma unroll
    for (int k = 0; k < 4; ++k) { q[k] = hilo(q[k]); thr[k] = hilo(thr[k]); }
    derive();
  }
};

struct QuadHot {
  v3 p; quat q;
  // angular / linear velocity as (w, v) element pairs, motor throttle and pwm as (0, 1), (2, 3) pairs: the operands of the
  // packed-fp32 chains below live in aligned register pairs for the whole kernel (members of vector type, not pairs
  // assembled from scalar members: those turned into overlapping vector loads that kept part of this struct in scratch)
  f2 wvx, wvy, wvz;
  f2 t01, t23;
  float I[3], E[3];
  m3 R; v3 wb, vb;
  f2 pw01, pw23;
  PF_DEV v3 w() const { return v3{wvx.x, wvy.x, wvz.x}; }
  PF_DEV v3 v() const { return v3{wvx.y, wvy.y, wvz.y}; }
  PF_DEV void set_wv(v3 w_, v3 v_) { wvx = f2{w_.x, v_.x}; wvy = f2{w_.y, v_.y}; wvz = f2{w_.z, v_.z}; }
  PF_DEV float thr(int i) const { return i == 0 ? t01.x : i == 1 ? t01.y : i == 2 ? t23.x : t23.y; }
  bool contact_now, contact_step;
  // shared world (dogfight.hpp: world_exchange): a contact point anywhere in the world after the previous tick (quadx.py:509),
  // this tick's drone-drone verdict for this body. Both stay false outside tick<.., SHARED = true>.
  bool world_contact = false, peer_contact = false;
  bool world_touch = false;  // some pair of this world is within reach of the contact response between drones this tick
  const float* wpose_ = nullptr;  // SHARED: the wave's pose / velocity exchange arrays, this lane, agents per world
  float* wvel_ = nullptr;
  float* rec_ = nullptr;          // the pair stage's contact records (the observation tile, as a generic pointer)
  int wtid = 0, wA = 1;
  lds_fptr cws;  // the wave's LDS regions for the contact solver (aliased onto the observation tile, idle during the ticks)
  int cws_floats;

  PF_DEV void derive() {
    // btMatrix3x3::setRotation scales by 2/|q|^2; q leaves quat_integrate()/the spawn normalised to
    // 1 ulp, so the factor is 2 to fp32 rounding and the reciprocal (a quarter-rate op) is skipped.
    // Packed fp32 where two results share a shape (uav_device.hpp: f2): the off-diagonal entries come in
    // +- pairs, and R^T w / R^T v are the same three-term chains on (w, v) element pairs.
    const float xs = q.x + q.x, ys = q.y + q.y, zs = q.z + q.z;
    const float xy = q.x * ys, xz = q.x * zs, yz = q.y * zs;
    const float dx = fmaf(-q.x, xs, 1.0f), dy = fmaf(-q.y, ys, 1.0f);  // 1 - xx, 1 - yy
    const f2 qw = f2{-q.w, q.w};
    const f2 r0011 = fma2(sp2(-q.z), sp2(zs), f2{dy, dx});  // (m00, m11)
    const f2 r0110 = fma2(qw, sp2(zs), sp2(xy));            // (m01, m10)
    const f2 r2002 = fma2(qw, sp2(ys), sp2(xz));            // (m20, m02)
    const f2 r1221 = fma2(qw, sp2(xs), sp2(yz));            // (m12, m21)
    const float m22 = fmaf(-q.y, ys, dx);
    R = m3{r0011.x, r0110.x, r2002.y, r0110.y, r0011.y, r1221.x, r2002.x, r1221.y, m22};
    f2 bx = sp2(R.m20) * wvz, by = sp2(R.m21) * wvz, bz = sp2(R.m22) * wvz;
    bx = fma2(sp2(R.m10), wvy, bx); by = fma2(sp2(R.m11), wvy, by); bz = fma2(sp2(R.m12), wvy, bz);
    bx = fma2(sp2(R.m00), wvx, bx); by = fma2(sp2(R.m01), wvx, by); bz = fma2(sp2(R.m02), wvx, bz);
    wb = v3{bx.x, by.x, bz.x};  // mulT(R, w)
    vb = v3{bx.y, by.y, bz.y};  // mulT(R, v)
  }
  // update_control (quadx.py:401-493). MODES = false: flight mode 0 only (rate PID + thrust, :437-438,472,482-493), the
  // instantiation BASELINE's metric is quoted on; MODES = true: K.mode selects -1 .. 7 at run time (wave-uniform branches).
  template <bool MODES, class CT = QuadCasc>
  PF_DEV void control(const QuadK& K, const pf_params_kptr Pk, CT& C, float s0, float s1, float s2, float s3, QuadStateD* D = nullptr) {
    // (HAS_D: the fp64 state is there exactly when the memories are doubles. A compile-time fact, never a test of the pointer: a private
    //  address compared with null is an address that has been LOOKED AT, and the whole QuadStateD then stays in scratch memory -- 352
    //  bytes a lane, every access a memory round trip: 52 us per env step in round 6's first cascaded-mode build)
    constexpr bool HAS_D = std::is_same<CT, QuadCascT<double>>::value;
    if (MODES) {
      if (K.mode == -1) {  // motor commands as they are (quadx.py:427-429): no clipping
        pw01 = f2{s0, s1}; pw23 = f2{s2, s3};
        if constexpr (HAS_D) { D->pwm[0] = s0; D->pwm[1] = s1; D->pwm[2] = s2; D->pwm[3] = s3; }
        return;
      }
      if (K.mode != 0) {
        // the cascaded modes: state derivation and every PID in fp64, shared with the generic vehicle (quadx_control_d.hpp: why --
        // round 4's float32 outer loops on polynomial atan2 / asin sat 5 x further from the reference than the generic kernel)
        const float sp4[4] = {s0, s1, s2, s3};
        float pw[4];
        if constexpr (HAS_D) {  // (state and memories in fp64)
          const QuadCtlIn in = D->ctl_inputs();
          quad_cascade_d(Pk, K.mode, (double)Pk->control_period, in, QuadMemT<double>{D->rI, D->rE, C.I1, C.E1, C.I2, C.E2, C.I3, C.E3, C.zI, C.zE}, sp4, D->pwm);
#pragma unroll
          for (int k = 0; k < 3; ++k) { I[k] = (float)D->rI[k]; E[k] = (float)D->rE[k]; }
#pragma unroll
          for (int k = 0; k < 4; ++k) pw[k] = (float)D->pwm[k];
        } else {
          const QuadCtlIn in = quad_ctl_inputs(q, v(), w(), p);
          quad_cascade_d(Pk, K.mode, (double)Pk->control_period, in, QuadMemD{I, E, C.I1, C.E1, C.I2, C.E2, C.I3, C.E3, C.zI, C.zE}, sp4, pw);
        }
        pw01 = f2{pw[0], pw[1]}; pw23 = f2{pw[2], pw[3]};
        return;
      }
    }
    const float st[3] = {wb.x, wb.y, wb.z};
    const float sp[3] = {s0, s1, s2};
    float a[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {  // pid.py:81-94
      float e = sp[k] - st[k];
      I[k] = med3(fmaf(K.kiT[k], e, I[k]), -K.lim[k], K.lim[k]);
      const float de = e - E[k];
      E[k] = e;
      a[k] = med3(fmaf(K.kdT[k], de, fmaf(K.kp[k], e, I[k])), -K.lim[k], K.lim[k]);
    }
    float z = med3(s3, 0.0f, 1.0f);
    // motor mixing (quadx.py:96-105): ((z -+ a0) -+ a1) -+ a2, motors (0, 1) and (2, 3) as packed pairs
    const f2 a0 = f2{-a[0], a[0]}, a1 = f2{-a[1], a[1]};
    f2 p01 = sp2(z) + a0, p23 = sp2(z) - a0;
    p01 = p01 + a1; p23 = p23 + a1;
    p01 = p01 - sp2(a[2]); p23 = p23 + sp2(a[2]);
    float hi = __builtin_fmaxf(__builtin_fmaxf(p01.x, p01.y), __builtin_fmaxf(p23.x, p23.y));
    float lo = __builtin_fminf(__builtin_fminf(p01.x, p01.y), __builtin_fminf(p23.x, p23.y));
    if (hi != lo) {
      float pmax = __builtin_fminf(hi, 1.0f), pmin = __builtin_fmaxf(lo, 0.05f);
      float ka = (pmin - lo) * frcp(pmax - lo), ks = (hi - pmax) * frcp(hi - pmin);
      // pwm = fma(ka, pmax - pwm, fma(-ks, pwm - pmin, pwm))
      const f2 u01 = p01 - sp2(pmin), u23 = p23 - sp2(pmin);
      const f2 d01 = sp2(pmax) - p01, d23 = sp2(pmax) - p23;
      const f2 g01 = fma2(sp2(-ks), u01, p01), g23 = fma2(sp2(-ks), u23, p23);
      p01 = fma2(sp2(ka), d01, g01); p23 = fma2(sp2(ka), d23, g23);
    }
    pw01 = f2{med3(p01.x, 0.05f, 1.0f), med3(p01.y, 0.05f, 1.0f)};
    pw23 = f2{med3(p23.x, 0.05f, 1.0f), med3(p23.y, 0.05f, 1.0f)};
  }
  // one physics tick: update_physics (quadx.py:495-510) + stepSimulation + update_state (:512-535)
  // CR: contact RESPONSE compiled in (pf_params.contact_response). The env tasks of this kernel end the episode in the Aviary
  // step that reports a floor contact, so the response can only alter that terminal observation; it is a template switch
  // because even its never-taken call site costs the hot loop (+0.6 us per env step at 65 536 lanes: one more divergent region
  // and its PHI copies per tick, profiles/r02), and with random actions some lane of nearly every wave is near the floor.
  // K: the constants of the flight path (QuadK itself, in scalar registers -- or QuadKV, the same fields copied to vector
  // registers for the calm ticks); Kc: the constants of the rare floor code, always the kernel argument.
  // NOFLOOR: the ticks of a calm wave (quadx_m0_env_kernel: no lane can come within a bounding radius of the floor during this env
  // step, and none holds a contact point): `near` is false in every lane and contact_now stays false, so the detection, the gate's
  // select and its three multiplications by 1.0 are left out -- x * 1.0f is x: the same bits, a dozen issue slots a tick fewer.
  template <bool CR, bool SHARED = false, class KT = QuadK, bool INL = false, bool COLD = true, bool NOFLOOR = false>
  PF_DEV void tick(const KT& K, const QuadK& Kc, float xi, const pf_params* Pfull) {
    static_assert(!NOFLOOR || (!CR && !SHARED), "NOFLOOR is the calm tick: no contact response, no shared world");
    const float s = fmaf(xi, K.m_noise, 1.0f);
    float k[4];
    {  // motors.py:110-195, t = fma(a, pwm - thr, thr) * noise; motors (0, 1) and (2, 3) as packed pairs
      const f2 d01 = pw01 - t01, d23 = pw23 - t23;
      t01 = fma2(sp2(K.m_a), d01, t01); t23 = fma2(sp2(K.m_a), d23, t23);
      t01 = t01 * sp2(s); t23 = t23 * sp2(s);
      k[0] = t01.x * __builtin_fabsf(t01.x); k[1] = t01.y * __builtin_fabsf(t01.y);
      k[2] = t23.x * __builtin_fabsf(t23.x); k[3] = t23.y * __builtin_fabsf(t23.y);
    }
    // body-frame angular acceleration (torque / inertia, constants pre-divided): motor arms and reaction torque,
    // rotational drag gated on "no contact in the world" (quadx.py:502-510), gyroscopic term -(w x I w) / I
    const float pqf = (contact_now || (SHARED && world_contact)) ? 0.0f : 1.0f;
    v3 wdb{fmaf(K.ryfI[0], k[0], fmaf(K.ryfI[1], k[1], fmaf(K.ryfI[2], k[2], K.ryfI[3] * k[3]))),
           fmaf(K.rxfI[0], k[0], fmaf(K.rxfI[1], k[1], fmaf(K.rxfI[2], k[2], K.rxfI[3] * k[3]))),
           K.tmaxI * ((k[2] + k[3]) - (k[0] + k[1]))};
    if (NOFLOOR) {
      wdb.x = fmaf(-K.pqI[0], wb.x * __builtin_fabsf(wb.x), wdb.x);
      wdb.y = fmaf(-K.pqI[1], wb.y * __builtin_fabsf(wb.y), wdb.y);
      wdb.z = fmaf(-K.pqI[2], wb.z * __builtin_fabsf(wb.z), wdb.z);
    } else {
    wdb.x = fmaf(-K.pqI[0], pqf * (wb.x * __builtin_fabsf(wb.x)), wdb.x);
    wdb.y = fmaf(-K.pqI[1], pqf * (wb.y * __builtin_fabsf(wb.y)), wdb.y);
    wdb.z = fmaf(-K.pqI[2], pqf * (wb.z * __builtin_fabsf(wb.z)), wdb.z);
    }
    wdb.x = fmaf(-K.gyI[0], wb.y * wb.z, wdb.x);
    wdb.y = fmaf(-K.gyI[1], wb.z * wb.x, wdb.y);
    wdb.z = fmaf(-K.gyI[2], wb.x * wb.y, wdb.z);
    // body-frame specific force (force / mass): body drag (boring_bodies.py:113-119) + thrust along +z
    v3 Fm{-K.dragM[0] * (vb.x * __builtin_fabsf(vb.x)), -K.dragM[1] * (vb.y * __builtin_fabsf(vb.y)),
          fmaf(-K.dragM[2], vb.z * __builtin_fabsf(vb.z), K.fmaxM * ((k[0] + k[1]) + (k[2] + k[3])))};
    // collision detection at the pre-integration pose. The single collision box is centred on the base origin, so its
    // lowest vertex sits at low = p.z - (|R20| hx + |R21| hy + |R22| hz), and away from the slab's rim the 15-axis verdict
    // "penetration >= 0" IS low <= 0 (the slab's top-face normal is the only axis that can separate a box from what is
    // locally a half-space; the direct form also avoids the cancellation of (p.z + 5) - (5 + ext) in fp32). The out-of-line
    // 15-axis test runs only within one bounding radius of the rim.
    bool near = !NOFLOOR && ((p.z - K.bound_radius) <= 0.0f) && ((p.z + K.bound_radius) >= -2.0f * K.plane_z);  // (not once it has fallen through, contact_response off)
    const bool persisted = contact_now;  // contact points left by the previous tick persist up to the breaking distance
    contact_now = false;
    float low = INFINITY;
    if (!NOFLOOR && __any(near)) {
      if (near) {
        const float hx = Kc.box_h[0], hy = Kc.box_h[1], hz = Kc.box_h[2];
        const float pxy = Kc.plane_xy, pz = K.plane_z;
        low = p.z - fmaf(__builtin_fabsf(R.m20), hx, fmaf(__builtin_fabsf(R.m21), hy, __builtin_fabsf(R.m22) * hz));
        const bool inside = (__builtin_fabsf(p.x) + Kc.bound_radius0 < pxy) && (__builtin_fabsf(p.y) + Kc.bound_radius0 < pxy) && (low > -pz);
        const float rdx = persisted ? Kc.brk : Kc.rd;
        contact_now = low <= rdx;
        if (!inside) contact_now = quad_floor_contact(p.x, p.y, p.z, q, hx, hy, hz, pxy, pz, rdx);
      }
    }
    if (SHARED) contact_now = contact_now || peer_contact;  // drone-drone hits enter contact_array[drone.Id] too (aviary.py:523-525)
    // world-frame angular and linear acceleration, R wdb and R Fm + g, row by row on (wdb, Fm) element pairs; then the
    // semi-implicit velocity update on (w, v) pairs. (fma(x, y, -0) == x * y for every x, y: the angular half of the last
    // row has no gravity term.)
    {
      const f2 X = f2{wdb.x, Fm.x}, Y = f2{wdb.y, Fm.y}, Z = f2{wdb.z, Fm.z};
      f2 u0 = sp2(R.m02) * Z, u1 = sp2(R.m12) * Z, u2 = fma2(sp2(R.m22), Z, f2{-0.0f, K.gravity_z});
      u0 = fma2(sp2(R.m01), Y, u0); u1 = fma2(sp2(R.m11), Y, u1); u2 = fma2(sp2(R.m21), Y, u2);
      u0 = fma2(sp2(R.m00), X, u0); u1 = fma2(sp2(R.m10), X, u1); u2 = fma2(sp2(R.m20), X, u2);
      const f2 nx = fma2(u0, sp2(K.dt), wvx), ny = fma2(u1, sp2(K.dt), wvy), nz = fma2(u2, sp2(K.dt), wvz);
      wvx = f2{med3(nx.x, -K.vmax, K.vmax), med3(nx.y, -K.vmax, K.vmax)};
      wvy = f2{med3(ny.x, -K.vmax, K.vmax), med3(ny.y, -K.vmax, K.vmax)};
      wvz = f2{med3(nz.x, -K.vmax, K.vmax), med3(nz.y, -K.vmax, K.vmax)};
    }
    // contact response (the constraint solve of stepSimulation) for lanes within one bounding radius of the floor:
    // out of line, with its constants read from the device parameter block inside the rare path
    // (can a constraint act at all this tick? conservative bound on the lowest vertex's height after the tick; when it stays
    //  above the allowed overlap every constraint is slack, the solve would return the velocities unchanged: Body::contact_may_act)
    // contact response between the drones of a shared world (shared_world.hpp: pair_stage_dev), one stage before the ground's:
    // publish the new velocity, the world's first lane resolves the contacts, take back velocity and position-level shift
    v3 shift{0.0f, 0.0f, 0.0f};
    if (SHARED && CR) {
      float* o = wvel_ + wtid * kPairVelStride;
      o[0] = wvx.y; o[1] = wvy.y; o[2] = wvz.y; o[3] = wvx.x; o[4] = wvy.x; o[5] = wvz.x; o[6] = 0.0f; o[7] = 0.0f; o[8] = 0.0f;
      lds_sync_wave();
      if (__any(world_touch)) {
        pair_stage_dev(Pfull, wpose_, wvel_, rec_, cws_floats, wtid, wA, world_touch);
        set_wv(v3{o[3], o[4], o[5]}, v3{o[0], o[1], o[2]});
        shift = v3{o[6], o[7], o[8]};
      }
    }
    float lift = 0.0f;
    if (CR) {
      bool act = false;
      if (near) {  // (low: the exact height of the lowest vertex, from the detection above)
        // (and no vertex is a contact unless the lowest one is within the margin; 1e-6: `low` and the solver's vertex heights
        //  are the same quantity rounded differently)
        const float vlow = wvz.y - fsqrt(dot(w(), w())) * Kc.bound_radius0;
        act = ((fmaf(K.dt, vlow, low + Kc.slop) < 0.0f) || (low < -Kc.slop)) && (low <= (persisted ? Kc.brk : Kc.margin) + 1e-6f);
      }
      // (unlikely: a wave solves a floor contact in a few ticks of an episode. The hint is also what keeps the register allocator's
      //  spill burst around the out-of-line call INSIDE this cold block: without it the caller-saved registers were stored at the top of
      //  the join block in front of it, ahead of its exec restore -- tools/isa_exec_check.py, 161 of round 5's 186 repaired sites.
      //  COLD = false: the Hover task's two-waves-per-SIMD instantiations, which never had such a site and lose 2-3 % at 524 288
      //  lanes with the hint -- 75 spilled registers against 43, profiles/r06/ab_plain_build_same_box.txt)
      if (COLD ? __builtin_expect(__any(act), 0) : (long)__any(act)) {
        // (INL: inline, in the instantiations sized for one wave per SIMD -- 512 registers: no call, so no stack, and a launch
        //  whose waves carry scratch memory dispatches 0.4 us slower; nothing pinned to callee-saved registers: the solve of this
        //  airframe in registers, quad_floor_solve. Otherwise out of line, the general solver: within the 256 registers of two
        //  waves per SIMD the inlined solve spills. profiles/README.md, r03 / r04)
        if (INL) {  // (the launcher picks these instantiations only with the manifold reduced to the incident face: pf_ctx_create)
          const bool at_rim = !((__builtin_fabsf(p.x) + Kc.bound_radius0 < Kc.plane_xy) && (__builtin_fabsf(p.y) + Kc.bound_radius0 < Kc.plane_xy));
          const QuadFloorOut o = quad_floor_solve(Kc, Pfull, act, persisted ? Kc.brk : Kc.margin, __any(act && at_rim), p, R.m00, R.m01, R.m02, R.m10, R.m11, R.m12,
                                                      R.m20, R.m21, R.m22, v(), w());
          set_wv(o.w, o.v);
          lift = Kc.c_erp * o.deepest;
        } else {
          const ContactOut o = contact_solve_dev(Pfull, cws, need_cap_of(act, cws_floats, persisted), p, q, v(), w());
          set_wv(o.w, o.v);  // (unchanged for a lane that did not ask or has no contact vertex)
          lift = Kc.c_erp * o.deepest;  // (already net of the slop)
        }
      }
    }
    if (SHARED) p = v3{fmaf(K.dt, wvx.y, p.x) + shift.x, fmaf(K.dt, wvy.y, p.y) + shift.y, fmaf(K.dt, wvz.y, p.z) + lift + shift.z};
    else p = v3{fmaf(K.dt, wvx.y, p.x), fmaf(K.dt, wvy.y, p.y), CR ? fmaf(K.dt, wvz.y, p.z) + lift : fmaf(K.dt, wvz.y, p.z)};
    q = quat_integrate(q, w(), K.half_dt);
    derive();
    if (!NOFLOOR) contact_step |= contact_now;
  }
  // this struct's float32 members <- the rounding of the fp64 state (and derive() on them)
  PF_DEV void view_of(const QuadStateD& D) {
    p = v3{(float)D.p[0], (float)D.p[1], (float)D.p[2]};
    q = quat{(float)D.q[0], (float)D.q[1], (float)D.q[2], (float)D.q[3]};
    set_wv(v3{(float)D.w[0], (float)D.w[1], (float)D.w[2]}, v3{(float)D.v[0], (float)D.v[1], (float)D.v[2]});
    derive();
  }
  // The same physics tick with the rigid-body state in fp64 (QuadStateD): motors in float32 as above, then the oracle's arithmetic --
  // rigid_tick_vel / rigid_tick_pos for a body with its centre of mass on the base origin and a diagonal inertia -- on doubles;
  // collision detection and the contact solve on the float32 view, as in tick<>; a lane whose velocities the solve changed takes them
  // over (an impact is float32 either way: the parity tests' impact tolerance).
  template <bool CR, bool INL>
  PF_DEV void tick_d(QuadStateD& D, const QuadK& K, float xi, const pf_params* Pfull) {
    double kd[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {  // motors.py:131-138: first-order lag, then the noise as a fraction of the state
      D.thr[i] += (double)K.m_a * (D.pwm[i] - D.thr[i]);
      D.thr[i] += (double)xi * D.thr[i] * (double)K.m_noise;
      kd[i] = D.thr[i] * __builtin_fabs(D.thr[i]);
    }
    t01 = f2{(float)D.thr[0], (float)D.thr[1]}; t23 = f2{(float)D.thr[2], (float)D.thr[3]};
    const double k0 = kd[0], k1 = kd[1], k2 = kd[2], k3 = kd[3];
    const double pqf = contact_now ? 0.0 : 1.0;
    // body-frame angular acceleration and specific force (the constants are tick<>'s: torque / inertia, force / mass)
    double wdb[3] = {(double)K.ryfI[0] * k0 + (double)K.ryfI[1] * k1 + (double)K.ryfI[2] * k2 + (double)K.ryfI[3] * k3,
                     (double)K.rxfI[0] * k0 + (double)K.rxfI[1] * k1 + (double)K.rxfI[2] * k2 + (double)K.rxfI[3] * k3,
                     (double)K.tmaxI * ((k2 + k3) - (k0 + k1))};
#pragma unroll
    for (int i = 0; i < 3; ++i) wdb[i] -= (double)K.pqI[i] * pqf * (D.wb[i] * __builtin_fabs(D.wb[i]));
    wdb[0] -= (double)K.gyI[0] * (D.wb[1] * D.wb[2]);
    wdb[1] -= (double)K.gyI[1] * (D.wb[2] * D.wb[0]);
    wdb[2] -= (double)K.gyI[2] * (D.wb[0] * D.wb[1]);
    const double Fm[3] = {-(double)K.dragM[0] * (D.vb[0] * __builtin_fabs(D.vb[0])), -(double)K.dragM[1] * (D.vb[1] * __builtin_fabs(D.vb[1])),
                          -(double)K.dragM[2] * (D.vb[2] * __builtin_fabs(D.vb[2])) + (double)K.fmaxM * ((k0 + k1) + (k2 + k3))};
    // collision detection at the pre-integration pose, on the float32 view (tick<>'s code)
    bool near = ((p.z - K.bound_radius) <= 0.0f) && ((p.z + K.bound_radius) >= -2.0f * K.plane_z);
    const bool persisted = contact_now;
    contact_now = false;
    float low = INFINITY;
    if (__any(near)) {
      if (near) {
        const float hx = K.box_h[0], hy = K.box_h[1], hz = K.box_h[2];
        const float pxy = K.plane_xy, pz = K.plane_z;
        low = p.z - fmaf(__builtin_fabsf(R.m20), hx, fmaf(__builtin_fabsf(R.m21), hy, __builtin_fabsf(R.m22) * hz));
        const bool inside = (__builtin_fabsf(p.x) + K.bound_radius0 < pxy) && (__builtin_fabsf(p.y) + K.bound_radius0 < pxy) && (low > -pz);
        const float rdx = persisted ? K.brk : K.rd;
        contact_now = low <= rdx;
        if (!inside) contact_now = quad_floor_contact(p.x, p.y, p.z, q, hx, hy, hz, pxy, pz, rdx);
      }
    }
    // world-frame accelerations, semi-implicit velocity update with the per-coordinate clamp
    const double vmax = K.vmax, dt = K.dt;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const double wd = D.R[3 * i] * wdb[0] + D.R[3 * i + 1] * wdb[1] + D.R[3 * i + 2] * wdb[2];
      const double a = D.R[3 * i] * Fm[0] + D.R[3 * i + 1] * Fm[1] + D.R[3 * i + 2] * Fm[2] + (i == 2 ? (double)K.gravity_z : 0.0);
      const double wn = D.w[i] + wd * dt, vn = D.v[i] + a * dt;
      D.w[i] = wn < -vmax ? -vmax : (wn > vmax ? vmax : wn);
      D.v[i] = vn < -vmax ? -vmax : (vn > vmax ? vmax : vn);
    }
    float lift = 0.0f;
    if (CR) {
      bool act = false;
      const v3 vf{(float)D.v[0], (float)D.v[1], (float)D.v[2]}, wf{(float)D.w[0], (float)D.w[1], (float)D.w[2]};
      if (near) {
        const float vlow = vf.z - fsqrt(dot(wf, wf)) * K.bound_radius0;
        act = ((fmaf(K.dt, vlow, low + K.slop) < 0.0f) || (low < -K.slop)) && (low <= (persisted ? K.brk : K.margin) + 1e-6f);
      }
      if (__builtin_expect(__any(act), 0)) {
        v3 nv, nw;
        if (INL) {
          const bool at_rim = !((__builtin_fabsf(p.x) + K.bound_radius0 < K.plane_xy) && (__builtin_fabsf(p.y) + K.bound_radius0 < K.plane_xy));
          const QuadFloorOut o = quad_floor_solve(K, Pfull, act, persisted ? K.brk : K.margin, __any(act && at_rim), p, R.m00, R.m01, R.m02, R.m10, R.m11, R.m12,
                                                      R.m20, R.m21, R.m22, vf, wf);
          nv = o.v; nw = o.w; lift = K.c_erp * o.deepest;
        } else {
          const ContactOut o = contact_solve_dev(Pfull, cws, need_cap_of(act, cws_floats, persisted), p, q, vf, wf);
          nv = o.v; nw = o.w; lift = K.c_erp * o.deepest;
        }
        // (unchanged for a lane that did not ask or has no contact vertex: such a lane keeps its fp64 velocities)
        if (nv.x != vf.x || nv.y != vf.y || nv.z != vf.z || nw.x != wf.x || nw.y != wf.y || nw.z != wf.z) {
          D.v[0] = nv.x; D.v[1] = nv.y; D.v[2] = nv.z; D.w[0] = nw.x; D.w[1] = nw.y; D.w[2] = nw.z;
        }
      }
    }
    // x += v dt; q <- exp(w dt / 2) q with the world-frame w, normalised (rigid_tick_pos: Taylor branch below 1e-3 rad/s, the pi/4 cap)
#pragma unroll
    for (int i = 0; i < 3; ++i) D.p[i] += dt * D.v[i];
    D.p[2] += (double)lift;
    {
      // sin(h) / ang = (dt / 2) sinc(h) and cos(h) for the half angle h = ang dt / 2 <= pi / 8, as series (to h^15 / h^16: 1e-18 of the
      // leading term at the cap; the reference's own Taylor branch below 1e-3 rad/s is the first two terms of the same series) --
      // the library's sin / cos carry an argument reduction this range never needs, 300 instructions a tick. Both are functions of
      // h^2 = (dt / 2)^2 |w|^2: no square root; the cap (ang dt > pi / 4, which three components clamped to vmax = 100 rad/s never reach
      // at 240 Hz) is the same test on the squares
      const double ww2 = D.w[0] * D.w[0] + D.w[1] * D.w[1] + D.w[2] * D.w[2];
      const double cap = 0.25 * 3.14159265358979323846;
      const double h2 = (ww2 * (dt * dt) > cap * cap) ? 0.25 * (cap * cap) : 0.25 * (dt * dt) * ww2;
      const double sinc = 1.0 + h2 * (-1.0 / 6.0 + h2 * (1.0 / 120.0 + h2 * (-1.0 / 5040.0 + h2 * (1.0 / 362880.0 + h2 * (-1.0 / 39916800.0 +
                          h2 * (1.0 / 6227020800.0 + h2 * (-1.0 / 1307674368000.0)))))));
      const double cw = 1.0 + h2 * (-0.5 + h2 * (1.0 / 24.0 + h2 * (-1.0 / 720.0 + h2 * (1.0 / 40320.0 + h2 * (-1.0 / 3628800.0 + h2 * (1.0 / 479001600.0 +
                        h2 * (-1.0 / 87178291200.0 + h2 * (1.0 / 20922789888000.0))))))));
      const double kq = 0.5 * dt * sinc;
      const double ax = D.w[0] * kq, ay = D.w[1] * kq, az = D.w[2] * kq;
      const double n0 = cw * D.q[0] + ax * D.q[3] + ay * D.q[2] - az * D.q[1], n1 = cw * D.q[1] + ay * D.q[3] + az * D.q[0] - ax * D.q[2],
                   n2 = cw * D.q[2] + az * D.q[3] + ax * D.q[1] - ay * D.q[0], n3 = cw * D.q[3] - ax * D.q[0] - ay * D.q[1] - az * D.q[2];
      // (|n|^2 = |q|^2 (cos^2 h + sin^2 h) = 1 to a few 1e-16 below the cap: 1 / sqrt(1 + e) = 1 - e/2 + 3 e^2 / 8 is exact to 1e-30;
      //  anything further from 1 -- the cap, a hand-written state -- through the square root)
      const double nn = n0 * n0 + n1 * n1 + n2 * n2 + n3 * n3, en = nn - 1.0;
      double inv = __builtin_fma(en, __builtin_fma(en, 0.375, -0.5), 1.0);
      if (!(__builtin_fabs(en) < 1e-6)) inv = rcp_d(sqrt_pos_d(nn));
      D.q[0] = n0 * inv; D.q[1] = n1 * inv; D.q[2] = n2 * inv; D.q[3] = n3 * inv;
    }
    D.derive();
    view_of(D);
    contact_step |= contact_now;
  }
};

// Shared world for the quadrotor of this kernel (one collision box centred on the base origin: quadk_from_params): the generic
// world_exchange (shared_world.hpp) with the box test written out for that one box pair and everything inline -- an out-of-line
// call in the tick loop forces every value that lives across it into the callee-saved half of the register file (the first
// version of this instantiation spilled 300-550 bytes per lane to scratch memory).
PF_DEV void quad_world_exchange(QuadHot& b, float* wpose, const int tid, const int A, const QuadK& K) {
  const int wbase = (tid / A) * A, wlocal = tid - wbase;
  float* me = wpose + tid * 8;
  me[0] = b.p.x; me[1] = b.p.y; me[2] = b.p.z; me[3] = b.q.x; me[4] = b.q.y; me[5] = b.q.z; me[6] = b.q.w;
  me[7] = b.contact_now ? 1.0f : 0.0f;
  lds_sync_wave();
  bool world = false, peer = false, nearp = false;
  // (gates only: the farthest a report or a contact point between two drones can reach -- the exact tests decide)
  const float far = fmaxf(fmaxf(K.margin, K.brk), K.rd);
  const float rr = 2.0f * K.bound_radius0 + 1.7320508f * far, rr2 = rr * rr;
  const float rp = 2.0f * K.bound_radius0 + 2.0f * far, rp2 = rp * rp;  // within reach of the contact response between drones
  const float h[3] = {K.box_h[0], K.box_h[1], K.box_h[2]};
  for (int j = 1; j < A; ++j) {
    int jj = wlocal + j;
    jj = jj >= A ? jj - A : jj;
    const float* o = wpose + (wbase + jj) * 8;
    world |= o[7] != 0.0f;
    const v3 d{b.p.x - o[0], b.p.y - o[1], b.p.z - o[2]};
    const float d2 = dot(d, d);
    nearp |= d2 <= rp2;
    const bool touch = d2 <= rr2;  // bounding spheres touch: this drone's box in the peer's box frame, 15 axes
    if (__any(touch)) {
      if (touch) {
        const m3 Rb = rot_from_quat(quat{o[3], o[4], o[5], o[6]});
        const m3& Ra = b.R;
        const m3 Rrel{Rb.m00 * Ra.m00 + Rb.m10 * Ra.m10 + Rb.m20 * Ra.m20, Rb.m00 * Ra.m01 + Rb.m10 * Ra.m11 + Rb.m20 * Ra.m21, Rb.m00 * Ra.m02 + Rb.m10 * Ra.m12 + Rb.m20 * Ra.m22,
                      Rb.m01 * Ra.m00 + Rb.m11 * Ra.m10 + Rb.m21 * Ra.m20, Rb.m01 * Ra.m01 + Rb.m11 * Ra.m11 + Rb.m21 * Ra.m21, Rb.m01 * Ra.m02 + Rb.m11 * Ra.m12 + Rb.m21 * Ra.m22,
                      Rb.m02 * Ra.m00 + Rb.m12 * Ra.m10 + Rb.m22 * Ra.m20, Rb.m02 * Ra.m01 + Rb.m12 * Ra.m11 + Rb.m22 * Ra.m21, Rb.m02 * Ra.m02 + Rb.m12 * Ra.m12 + Rb.m22 * Ra.m22};
        // (reported from the gap rd on -- up to the breaking distance when either drone holds contact points: ONE test per pair,
        //  the lower-indexed drone's box in the frame of the other's, enlarged -- shared_world.hpp: peers_overlap_dev)
        const float rdx = (b.contact_now || o[7] != 0.0f) ? K.brk : K.rd;
        const float hb[3] = {h[0] + rdx, h[1] + rdx, h[2] + rdx};
        if (wlocal < jj) {
          peer |= box_overlaps_aabb(mulT(Rb, d), Rrel, h, v3{0.f, 0.f, 0.f}, hb);
        } else {  // the peer is a: its box in MY frame, Ra^T Rb = Rrel^T
          const m3 RrelT{Rrel.m00, Rrel.m10, Rrel.m20, Rrel.m01, Rrel.m11, Rrel.m21, Rrel.m02, Rrel.m12, Rrel.m22};
          peer |= box_overlaps_aabb(mulT(Ra, v3{-d.x, -d.y, -d.z}), RrelT, h, v3{0.f, 0.f, 0.f}, hb);
        }
      }
    }
  }
  b.world_contact = world;
  b.peer_contact = peer;
  b.world_touch = widen_to_world(nearp, tid, A);
  lds_sync_wave();
}

// Per-wave phase timeline (profiles/tools/phase_trace.py): a diagnostic build switch, -DPF_PHASE_TRACE. Each wave of the
// one-step-per-launch instantiation stamps the shader clock (s_memtime) at its phase boundaries and its first lane writes the
// stamps to a device array the variant library exports; the product build has none of it.
#ifdef PF_PHASE_TRACE
constexpr int kPhaseStamps = 13;
__device__ unsigned long long g_phase_trace[4096 * kPhaseStamps];
__device__ unsigned long long g_calm_trace[2];  // (waves that were not calm over the env step; Aviary steps of theirs that were not calm)
#define PF_STAMP(i) do { if (ROLL == 0) { pf_ts[i] = __builtin_readcyclecounter(); } } while (0)
#else
#define PF_STAMP(i) do { } while (0)
#endif

// What the next env.reset() of a lane needs that is random -- the settled spawn state (z, vz, motor state after the settle phase's
// noisy ticks) and, in the Waypoints task, the targets -- prepared AHEAD ("spare"). For the Hover / Waypoints tasks a reset's draws are
// keyed by the event counter at the lane's PREVIOUS reset (oracle/uav_oracle.c: orc_env_reset), which is known for a whole episode:
// instead of every wave serving the two or three of its lanes that restart in a given env step -- 1.7 us (Hover) / 2.8 us
// (Waypoints) of every wave's life for a SIMD utilisation of 4 % -- a wave prepares the spares of ALL its lanes that have used
// theirs up once every kSpareEvery env steps (episodes last 16 steps at the least under the action space's own draws, 28 on
// average: a lane cannot need two within that), and a reset copies. A lane without a valid spare (the first reset of a context,
// a hand-written state, an episode shorter than the refill period) generates on the spot with the same code: the results do not
// depend on when they were computed. State: group 7 = (z, vz, thr, key word), groups 8-11 = 4 x (x, y, z, yaw) of the targets;
// key word = the key in bits 0-30, bit 31 = "the values are valid". (The cascaded flight modes, which own groups 7-11, keep the key
// in group 11's third word and generate at reset; so do the injected-noise and shared-world instantiations and the generic kernel.)
constexpr uint32_t kSpareValid = 0x80000000u;
constexpr uint32_t kSpareEvery = 16u;
// The refill cadence's counter words (quadx_m0_env_kernel: launch_ctr), one per workgroup, each on a 128-byte line of its own: packed
// sixteen to a line they cost a 4 096-lane launch 0.15 us (7.60 against 7.45 us; 65 536 lanes: no difference -- profiles/r06).
constexpr int kCtrStride = 32;
struct QuadSpare {
  float z, vz, thr;
  float4 t[4];
};

// ------------------------------------------------------------------------------------------
// The kernel. Control flow is deliberately flat -- a prologue, one uniform-trip-count loop over the
// env step's Aviary steps with a single per-lane predicate, an epilogue -- because every extra
// divergent region costs the register allocator a round of PHI copies per iteration (the first,
// state-machine shaped version spent ~300 of its ~1 150 instructions per Aviary step on v_mov).
// Specialised at compile time on the task and the noise source; requires (checked by
// quadk_from_params) ticks_per_control == 2, env_step_ratio <= 4, a level spawn at rest and
// settle_ticks a multiple of 4 and <= 24.
// ROLLOUT: pf_rollout -- the same env step k_steps times in one launch with the lane's state resident in
// registers (the loop below is then a real loop; for the one-launch-per-step instantiation it has a
// compile-time trip count of one and disappears). Every step still writes its observation / reward /
// flags, to trajectory buffers [k_steps][n][..]; the stores of step s drain while step s+1 computes.
// ROLL: 0 = one env step per launch (pf_env_step / pf_env_reset); 1 = pf_rollout with on-device action sampling -- the
// loop then contains NO vector-memory load, so nothing in it ever waits on vmcnt (on gfx9 stores count in vmcnt too: a
// load in the loop would make every step wait for the previous step's observation stores to be acknowledged);
// 2 = pf_rollout over a given action sequence (prefetched one step ahead; pays that wait).
// One wavefront per workgroup: no barrier, no inter-wave traffic. (Multi-wave workgroups were measured in round 3 and dropped --
// profiles/r06/experiments/ab_switches.patch has the switch; the index arithmetic below keeps the general form.)
constexpr int kQuadWPB = 1;
// MODES: false = flight mode 0 only; true = the flight mode is K.mode, -1 .. 7 (cascaded PIDs; their memories in state groups
// 7-11 and, at reset, the z PIDs inside the settle recurrence).
// SHARED (PF_TASK_MA_HOVER only): the K.apw agents of an env share one world (pose / contact exchange before every tick).
// WPS: the waves per SIMD the register budget is sized for. 2 (256 registers): every batch. 1 (512 registers, the contact solve
// inlined: no stack): chosen by the launcher for batches of at most one wave per SIMD, where a second resident wave would have
// nothing to run -- 11.6 -> 11.2 us per hover step at 65 536 lanes; at 524 288 lanes, 71 us against 57 us with two waves resident.
// (the shared-world instantiations need more than 256 registers whatever they are asked for -- the compiler's "desired occupancy
//  was 2, final occupancy is 1" in rounds 4 / 5 --: their budget says so)
template <int TASK, int NOISE, int LPW, int ROLL, bool CR, bool MODES = false, bool SHARED = false, int WPS = 2>
__global__ void __launch_bounds__(64 * kQuadWPB, SHARED ? 1 : WPS) quadx_m0_env_kernel(const QuadK K, const pf_buffers B, const pf_params* __restrict__ Pfull,
                                                             const int n, const uint64_t lane0, const int op,
                                                             const uint8_t* __restrict__ mask, const int k_steps, const uint32_t step0,
                                                             uint32_t* launch_ctr) {
  constexpr bool ROLLOUT = ROLL != 0;
  // (see QuadSpare) REKEY: a reset's draws are keyed by the counter at the previous reset; SPARE: ... and prepared ahead. launch_ctr: one
  // word per workgroup (kCtrStride apart) in device memory, the env steps this context has taken -- the refill cadence is a function of that count alone.
  // It is DEVICE state (round 5 passed the host's count as a kernel argument, which a HIP-graph capture bakes in: a captured
  // single step replayed for ever either never refilled or refilled in every launch -- ADVICE r05): every workgroup reads its own
  // word with the state groups and writes it back advanced by the steps it took, so all workgroups of a launch see the same count.
  constexpr bool REKEY = TASK == PF_TASK_HOVER || TASK == PF_TASK_WAYPOINTS;
  constexpr bool SPARE = REKEY && !MODES && !SHARED && NOISE != PF_NOISE_INJECT;
  typedef QuadSpare Sp;
  constexpr bool GIVEN = ROLL == 2;
  constexpr int kMaxD = 13 + 4 + 4 + 16;  // attitude + 4 targets x (delta, yaw error)
  constexpr int kSettleMax = 24;  // settle ticks served by the cooperative generator (3 Philox calls)
  __shared__ __attribute__((aligned(16))) float tile_all[kQuadWPB * LPW * kMaxD];
  __shared__ __attribute__((aligned(16))) float sxi_all[kQuadWPB * 64 * kSettleMax];
  __shared__ int spos_all[kQuadWPB * 64];
  __shared__ float wpose_all[SHARED ? kQuadWPB * 64 * 8 : 1];  // shared worlds: pose + contact bit of every lane
  __shared__ float wvel_all[SHARED ? kQuadWPB * 64 * kPairVelStride : 1];  // ... and the new velocities for the pair stage
  __shared__ uint32_t sctr_all[kQuadWPB * 64];
  __shared__ __attribute__((aligned(16))) float splds_all[(ROLLOUT && SPARE) ? kQuadWPB * 64 * 20 : 4];  // pf_rollout: the lanes' spares (a row is its lane's own)
  const int wid = kQuadWPB > 1 ? (int)(threadIdx.x >> 6) : 0;
  const int tid = kQuadWPB > 1 ? (int)(threadIdx.x & 63u) : (int)threadIdx.x;
  float* const tile = tile_all + wid * (LPW * kMaxD);
  float* const sxi = sxi_all + wid * (64 * kSettleMax);
  int* const spos = spos_all + wid * 64;
  uint32_t* const sctr = sctr_all + wid * 64;
  float* const wpose = wpose_all + (SHARED ? wid * 64 * 8 : 0);
  const int apw = SHARED ? K.apw : 1;
  const int wave_base = (blockIdx.x * kQuadWPB + wid) * LPW;
  if (kQuadWPB > 1 && wave_base >= n) return;  // (no barrier anywhere: a surplus wave simply leaves)
  const int lane = wave_base + tid;
  const bool valid = (tid < LPW) && (lane < n);
  const size_t li = valid ? (size_t)lane : (size_t)(n - 1);
  const size_t N = (size_t)n;
  const float4* Sin = reinterpret_cast<const float4*>(B.state);
  float4* Sout = reinterpret_cast<float4*>(B.state);
#ifdef PF_PHASE_TRACE
  unsigned long long pf_ts[kPhaseStamps];
  pf_ts[11] = __builtin_amdgcn_s_memrealtime();  // 100 MHz wall clock: aligns the waves of different CUs
#endif
  PF_STAMP(0);

  QuadHot V;
  // (the cascaded flight modes without a shared world: the master copy of the rigid-body state and of every PID memory in fp64; state
  //  groups 16-19 the state's remainders, 20-21 the rate PID's, 22-26 the cascade's)
  constexpr bool DSTATE = MODES && !SHARED;
  QuadStateD SD;  // (dead unless DSTATE)
  static_assert(LPW * kMaxD >= kContactSlotFloats, "the contact solver's LDS regions alias the observation tile: at least one worst-case region");
  V.cws = (lds_fptr)tile;
  V.cws_floats = LPW * kMaxD;
  if (SHARED) { V.wpose_ = wpose; V.wvel_ = wvel_all + wid * 64 * kPairVelStride; V.rec_ = tile; V.wtid = tid; V.wA = apw; }
  QuadCascT<typename std::conditional<DSTATE, double, float>::type> C;  // (MODES only; dead otherwise. DSTATE: the memories in fp64, their remainders in groups 22-26)
  const pf_params_kptr Pk = uniform_params(Pfull);
  // (see the calm test below)
  constexpr bool CALM = CR && !SHARED && !MODES && NOISE != PF_NOISE_INJECT;
  QuadKV KV{};
  if (CALM) KV = quadkv_from(K);
  float tgt[4][3];
  float new_dist, old_dist;
  int step_count, flags, n_left;
  uint32_t rng_ctr;
  f8 zn;  // this step's motor-noise normals (Philox call 0 of the event)
  uint32_t call0 = 0u;     // (SPARE) env steps this context had taken before this launch (launch_ctr)
  uint32_t rkw = 0u;       // (REKEY) the reset key word: key in bits 0-30, bit 31 = the spare's values are valid
  bool sp_dirty = false;   // ... changed in this launch: goes back to the state
  uint32_t reset_key_now = 0u;  // the key of the reset at hand (do_resets)
  Sp spv{};                // (SPARE, one step per launch) the lane's spare as loaded / as refilled
  float* const splds = splds_all + ((ROLLOUT && SPARE) ? (wid * 64 + tid) * 20 : 0);  // (SPARE, pf_rollout) ... its row in LDS
  float4 a_pre = float4{0.f, 0.f, 0.f, 0.f};  // (one step per launch) this step's action, requested with the state
  const float4 f4z = float4{0.f, 0.f, 0.f, 0.f};
  float4 cm0 = f4z, cm1 = f4z, cm2 = f4z, cm3 = f4z, cm4 = f4z, cm5 = f4z, cm6 = f4z, cm7 = f4z, cm8 = f4z, cm9 = f4z;  // (MODES, a cascaded mode) state groups 7-11 and 22-26 as loaded
  {
    // the int group is requested first: loads complete in order, so the step's Philox call (which
    // needs only the event counter) runs while the rest of the state is still in flight
    float4 gi = Sin[6 * N + li];
    float4 g0 = Sin[0 * N + li], g1 = Sin[1 * N + li], g2 = Sin[2 * N + li], g3 = Sin[3 * N + li], g4 = Sin[4 * N + li],
           g5 = Sin[5 * N + li];
    // (SPARE: the lane's spare with the state groups -- requested behind the int group's arrival it came 0.4 us late for the resets)
    float4 gs7 = float4{0.f, 0.f, 0.f, 0.f};
    if (SPARE) gs7 = Sin[7 * N + li];
    if (SPARE) {
      // (a VECTOR load behind the state groups, in their order: as a scalar load it would sit in every lgkmcnt wait that follows. The
      //  address goes through an opaque copy so that the compiler does not recognise it as uniform, and comes back as a GLOBAL
      //  pointer: through a generic one this was a flat_load, which may complete out of order with the global loads -- every wait
      //  behind it became vmcnt(0), 0.35 us in front of the step's Philox call)
      typedef const uint32_t __attribute__((address_space(1))) * gu32ptr;
      uintptr_t ca = reinterpret_cast<uintptr_t>(launch_ctr + kCtrStride * blockIdx.x);
      asm volatile("" : "+v"(ca));
      call0 = *reinterpret_cast<gu32ptr>(ca);
    }
    // the action is first needed after the resets, a microsecond from here: requested where it is used (inside the stepping
    // lanes' branch) every wave sat out its whole memory latency there; requested behind the state groups it is long there
    if (ROLL == 0 && op == 0) a_pre = reinterpret_cast<const float4*>(B.actions)[li];
    float4 dl0 = float4{0.f, 0.f, 0.f, 0.f}, dl1 = dl0, dl2 = dl0, dl3 = dl0;
    float4 dl4 = dl0, dl5 = dl0;
    if (DSTATE) { dl0 = Sin[16 * N + li]; dl1 = Sin[17 * N + li]; dl2 = Sin[18 * N + li]; dl3 = Sin[19 * N + li]; dl4 = Sin[20 * N + li]; dl5 = Sin[21 * N + li]; }
    // (the cascade's memories -- groups 7-11, DSTATE: their remainders in 22-26 -- requested HERE, with the state groups: asked for where
    //  they are unpacked they were a second and a third memory round trip in front of the first Aviary step, 4.5 us by the phase trace)
    if (MODES) {  // (whatever the mode: a wave-uniform branch around the requests costs a spill of its own; mode -1 ignores the words)
      cm0 = Sin[7 * N + li]; cm1 = Sin[8 * N + li]; cm2 = Sin[9 * N + li]; cm3 = Sin[10 * N + li]; cm4 = Sin[11 * N + li];
      if (DSTATE) { cm5 = Sin[22 * N + li]; cm6 = Sin[23 * N + li]; cm7 = Sin[24 * N + li]; cm8 = Sin[25 * N + li]; cm9 = Sin[26 * N + li]; }
    }
    if (CR && blockIdx.x < kRareTextPrefetchBlocks) rare_text_prefetch(tid);  // (behind the state loads: one wait for both)
    rng_ctr = (uint32_t)__float_as_int(gi.z);
    PF_STAMP(1);  // (the int group has arrived)
    if (REKEY) {
      // the key word (and the spare) of the lanes that will need them, requested as soon as the flags are here: a lane that restarts
      // in this launch, every lane of a launch that refills the spares or resets in the same step; pf_rollout: every lane, once
      const bool in11 = MODES && K.mode > 0;
      // (where the key shares group 11 with the cascade's memories it is rewritten with them by every launch: always read)
      // (SPARE: every lane, every launch -- 16 B a lane in Hover, 80 B in Waypoints that only the restarting lanes use. Requested
      //  by those lanes alone, once their flags are here, the words came from HBM and not from the cache the state groups of the
      //  previous launch still sit in: 2.8 us in front of the reset, as long as generating them had taken -- profiles/r05)
      const bool need_sp = SPARE || ROLLOUT || in11 || op == 1 || K.autoreset == PF_AUTORESET_SAME_STEP ||
                           (__float_as_int(gi.y) & (PF_F_TERMINATED | PF_F_TRUNCATED)) != 0;
      if (need_sp) {
        const float4 gk = SPARE ? gs7 : (in11 ? cm4 : (MODES ? cm0 : Sin[(size_t)7 * N + li]));
        rkw = (uint32_t)__float_as_int(in11 ? gk.z : gk.w);
        // pf_env_reset of EVERY lane (null mask): the spares in the state are not trusted -- a state buffer may come from another
        // context (other seed, lane offset, spawn pose, settle length, dome, number of targets) with bit 31 set; this reset generates
        // on the spot and the refill below prepares fresh ones (include/pyflyt_amd.h at pf_env_reset)
        if (SPARE && op == 1 && mask == nullptr) rkw &= ~kSpareValid;
        if (SPARE) {
          spv.z = gk.x; spv.vz = gk.y; spv.thr = gk.z;
          // (the targets' four groups behind the int group: in front of it, with the state groups, they cost the Waypoints launch
          //  0.2 us -- 16.9 against 16.7 us --, the settled state's one group gained the Hover launch 0.13)
          if (TASK == PF_TASK_WAYPOINTS) { spv.t[0] = Sin[8 * N + li]; spv.t[1] = Sin[9 * N + li]; spv.t[2] = Sin[10 * N + li]; spv.t[3] = Sin[11 * N + li]; }
          if (ROLLOUT) {
            float4* r4 = reinterpret_cast<float4*>(splds);
            r4[0] = float4{spv.z, spv.vz, spv.thr, 0.0f};
            if (TASK == PF_TASK_WAYPOINTS) { r4[1] = spv.t[0]; r4[2] = spv.t[1]; r4[3] = spv.t[2]; r4[4] = spv.t[3]; }
          }
        }
      }
    }
    if (NOISE == PF_NOISE_PHILOX) {
      if (op == 0) zn = normal8(philox4x32(K.seed_lo, K.seed_hi, (uint32_t)(lane0 + li), rng_ctr, 0u, 0u));
    }
    PF_STAMP(2);  // (the step's Philox call done)
    V.p = v3{g0.x, g0.y, g0.z}; new_dist = g0.w;
    V.q = quat{g1.x, g1.y, g1.z, g1.w};
    V.set_wv(v3{g2.w, g3.x, g3.y}, v3{g2.x, g2.y, g2.z});
    V.t01 = f2{g3.z, g3.w}; V.t23 = f2{g4.x, g4.y};
    V.I[0] = g4.z; V.I[1] = g4.w; V.I[2] = g5.x;
    V.E[0] = g5.y; V.E[1] = g5.z; V.E[2] = g5.w;
    if (DSTATE) {  // master = stored float32 rounding + stored remainder (g16: p, q.x | g17: q.yzw, v.x | g18: v.yz, w.xy | g19: w.z)
      SD.p[0] = (double)g0.x + (double)dl0.x; SD.p[1] = (double)g0.y + (double)dl0.y; SD.p[2] = (double)g0.z + (double)dl0.z;
      SD.q[0] = (double)g1.x + (double)dl0.w; SD.q[1] = (double)g1.y + (double)dl1.x; SD.q[2] = (double)g1.z + (double)dl1.y; SD.q[3] = (double)g1.w + (double)dl1.z;
      SD.v[0] = (double)g2.x + (double)dl1.w; SD.v[1] = (double)g2.y + (double)dl2.x; SD.v[2] = (double)g2.z + (double)dl2.y;
      SD.w[0] = (double)g2.w + (double)dl2.z; SD.w[1] = (double)g3.x + (double)dl2.w; SD.w[2] = (double)g3.y + (double)dl3.x;
      // (g20: the rate PID's I, E.x | g21: E.yz -- the remainders of groups 4 / 5's words)
      SD.rI[0] = (double)g4.z + (double)dl4.x; SD.rI[1] = (double)g4.w + (double)dl4.y; SD.rI[2] = (double)g5.x + (double)dl4.z;
      SD.rE[0] = (double)g5.y + (double)dl4.w; SD.rE[1] = (double)g5.z + (double)dl5.x; SD.rE[2] = (double)g5.w + (double)dl5.y;
      // (the motor states' remainders in the free words: g19.yzw, g21.z)
      SD.thr[0] = (double)g3.z + (double)dl3.y; SD.thr[1] = (double)g3.w + (double)dl3.z; SD.thr[2] = (double)g4.x + (double)dl3.w; SD.thr[3] = (double)g4.y + (double)dl5.z;
#pragma unroll
      for (int k = 0; k < 4; ++k) SD.pwm[k] = 0.05;
      SD.derive();
    }
    step_count = __float_as_int(gi.x); flags = __float_as_int(gi.y);
    n_left = __float_as_int(gi.w);
    if (TASK == PF_TASK_WAYPOINTS || TASK == PF_TASK_MA_HOVER) {  // MA: spawn pos (3), spawn quat (4), current action (4)
      float4 a = Sin[12 * N + li], b = Sin[13 * N + li], c = Sin[14 * N + li];
      tgt[0][0] = a.x; tgt[0][1] = a.y; tgt[0][2] = a.z; tgt[1][0] = a.w;
      tgt[1][1] = b.x; tgt[1][2] = b.y; tgt[2][0] = b.z; tgt[2][1] = b.w;
      tgt[2][2] = c.x; tgt[3][0] = c.y; tgt[3][1] = c.z; tgt[3][2] = c.w;
    }
  }
  old_dist = new_dist;
  if (MODES && K.mode > 0) {
    C.unpack(cm0, cm1, cm2, cm3, cm4, false);
    if constexpr (DSTATE) C.unpack(cm5, cm6, cm7, cm8, cm9, true);  // (+ the remainders)
  } else C.zero();
  // MA hover (ma_quadx_base_env.py:139-150,326-332): the action of the previous call, observed this call
  float4 ma_past = float4{0.f, 0.f, 0.f, 0.f};
  if (TASK == PF_TASK_MA_HOVER) ma_past = Sin[15 * N + li];
  // waypoints with yaw targets (waypoint_handler.py:85-89): the four yaw targets ride in group 15
  const bool kYaw = (TASK == PF_TASK_WAYPOINTS) && K.use_yaw != 0;
  float ytg[4] = {0.f, 0.f, 0.f, 0.f};
  float yaw_err0 = 0.0f;  // |yaw error| to the next target as of the last compute_state (waypoint_handler.py:156)
  if (kYaw) { float4 y = Sin[15 * N + li]; ytg[0] = y.x; ytg[1] = y.y; ytg[2] = y.z; ytg[3] = y.w; }
  // yaw of getEulerFromQuaternion from the rotation matrix derive() holds (gimbal branch: the library definition)
  auto yaw_now = [&]() {
    if (__builtin_fabsf(V.R.m20) >= 0.99999f) return euler_from_quat(V.q).z;
    return fast_atan2(V.R.m10, V.R.m00);
  };
  auto wrap_pi = [](float e) {  // waypoint_handler.py:147-149
    e = e > kPi ? e - 2.0f * kPi : e;
    return e < -kPi ? e + 2.0f * kPi : e;
  };
  V.contact_now = (flags & PF_F_CONTACT) != 0;
  V.contact_step = false;
  V.derive();
#ifdef PF_PHASE_TRACE
  asm volatile("" ::"v"(V.wb.x), "v"(V.vb.z));
#endif
  PF_STAMP(3);  // (every state group has arrived and is unpacked)
  bool term = (flags & PF_F_TERMINATED) != 0, trunc = (flags & PF_F_TRUNCATED) != 0;

  bool active;
  if (op == 1) active = (mask == nullptr) || (mask[li] != 0);
  else active = true;
  active = active && valid;
  if (SHARED && op == 1) active = widen_to_world(active, tid, apw) && valid;  // a mask that names some agents of a world resets the world

  float act0 = 0.f, act1 = 0.f, act2 = 0.f, act3 = 0.f;
  float reward = 0.0f;
  bool pop_pending = false;
  bool was_reset = false;
  const int D = (K.angle_repr ? 13 : 12) + 8 + (TASK == PF_TASK_WAYPOINTS ? (kYaw ? 4 : 3) * K.num_targets : (TASK == PF_TASK_MA_HOVER ? 3 : 0));
  const int settle_ticks = K.settle_steps * 2;

  auto pop_target = [&]() {
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
      for (int c = 0; c < 3; ++c) tgt[k][c] = tgt[k + 1][c];
    ytg[0] = ytg[1]; ytg[1] = ytg[2]; ytg[2] = ytg[3];
    n_left -= 1;
  };
  // One wave per workgroup: LDS operations of a wave execute in issue order, so the row writes only
  // have to be retired (lgkmcnt) and not reordered by the compiler before the tile is read back --
  // no s_barrier and, unlike __syncthreads(), no wait on outstanding global stores.
  auto lds_sync = [&]() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0), vmcnt/expcnt untouched
    __builtin_amdgcn_wave_barrier();
  };
  // What a resetting lane draws, generated cooperatively: settle_ticks motor-noise normals (3 Philox calls per lane, stream 1) and,
  // in the Waypoints task, the targets (waypoint_handler.py:53-89: 3 calls, 4 with yaw targets, stream 2, and per target two sine /
  // cosine pairs). Instead of the two or three resetting lanes of a wave walking through their calls serially while the other lanes
  // idle (0.85 us + 1.4 us of a Waypoints wave's life, r03 / r04), the (lane, call) pairs are dealt out over all 64 lanes. Since
  // round 5 ONE pass serves both streams -- a lane evaluates one Philox call and turns it into eight normals or four uniforms; two
  // passes paid for the Philox latency (four quarter-rate multiplies per round) and an LDS round trip twice --; pass 2, one (lane,
  // target) pair per lane, turns the uniforms into targets, the same arithmetic as the per-lane path, through the observation tile
  // (idle here); the resetting lane picks up its noise from sxi and its 4 x (x, y, z, yaw) from the tile. Wave-uniform call.
  // (Injected draws -- B.xi_reset, B.u_targets -- keep the per-lane paths.)
  const bool coop_targets = (TASK == PF_TASK_WAYPOINTS) && !((NOISE == PF_NOISE_INJECT) && (B.u_targets != nullptr));
  auto prepare_reset_draws = [&](const bool reset_now, const uint32_t key) {
    const bool want_noise = NOISE == PF_NOISE_PHILOX;
    if (!want_noise && !coop_targets) return;
    const unsigned long long m = __ballot(reset_now);
    if (m == 0ull) return;
    const int r = __popcll(m);
    // (the seed through an opaque copy made HERE: the compiler had hoisted the Philox key schedule of this rare path -- twenty
    //  scalar additions, each spilled to a VGPR lane -- in front of the early return, into every wave's life)
    uint32_t seed_lo = K.seed_lo, seed_hi = K.seed_hi;
    asm volatile("" : "+s"(seed_lo), "+s"(seed_hi));
    if (reset_now) {
      spos[__popcll(m & ((1ull << tid) - 1ull))] = tid;
      sctr[tid] = key;
    }
    lds_sync();
    const int ncall = want_noise ? (settle_ticks + 7) >> 3 : 0;                // stream 1: eight normals per call
    const int nt = K.num_targets, ntc = coop_targets ? (kYaw ? 4 : 3) : 0;     // stream 2: four uniforms per call
    const int per = ncall + ntc;
    float* const U = tile;             // [lane][16]: the uniforms of calls 0 .. 3
    float* const TG = tile + 64 * 16;  // [lane][16]: four targets x (x, y, z, yaw)
    for (int base = 0; base < r * per; base += 64) {
      const int j = base + tid;
      if (j < r * per) {
        const int which = j / per, c = j - which * per;
        const int src = spos[which];
        const bool noise_call = c < ncall;
        const int call = noise_call ? c : c - ncall;
        const u32x4 x = philox4x32(seed_lo, seed_hi, (uint32_t)(lane0 + (uint64_t)(wave_base + src)), sctr[src], (uint32_t)call, noise_call ? 1u : 2u);
        if (noise_call) {
          const f8 z = normal8(x);
#pragma unroll
          for (int e = 0; e < 8; ++e) sxi[src * kSettleMax + call * 8 + e] = 4.0f + z.v[e];
        } else {
          const f4 u = uniform4(x);
          float* o = U + src * 16 + call * 4;
          o[0] = u.a; o[1] = u.b; o[2] = u.c; o[3] = u.d;
        }
      }
    }
    lds_sync();
    if (coop_targets) {
      for (int base = 0; base < r * nt; base += 64) {
        const int j = base + tid;
        if (j < r * nt) {
          const int which = j / nt, i = j - which * nt;
          const int src = spos[which];
          const float* u = U + src * 16;
          const float theta = u[i], phi = u[nt + i], dist = fmaf(K.dome09m1, u[2 * nt + i], 1.0f);  // theta, phi in turns
          float st, ct, sph, cph;
          sincos_turns(theta, st, ct);
          sincos_turns(phi, sph, cph);
          const float zz = __builtin_fabsf(dist * cph);
          float* o = TG + src * 16 + 4 * i;
          o[0] = dist * sph * ct; o[1] = dist * sph * st; o[2] = zz > K.min_height ? zz : K.min_height;
          o[3] = kYaw ? fmaf(2.0f * kPi, u[3 * nt + i], -kPi) : 0.0f;
        }
      }
      lds_sync();
    }
  };
  // The settle phase as a vertical recurrence (see reset_lane): (z, vz, thr) in -> out, pwm_s = the motor command it ends on.
  // Its noise comes from sxi (prepare_reset_draws) / B.xi_reset. Called by reset_lane and, for the lanes whose next episode is
  // prepared ahead, by make_spare.
  auto settle_run = [&](float& z, float& vz, float& thr, float& pwm_s) {
    pwm_s = 0.05f;  // mode 0: rate error 0 -> command 0 -> clipped to 0.05
    const float z_hold = z;
    C.zero();  // set_mode: fresh PID objects; drone.reset(): the z PIDs too (quadx.py:206,222-231)
    auto settle_control = [&]() {
      if (!MODES || K.mode == 0) return;
      if (K.mode == -1) { pwm_s = 0.0f; return; }  // motor commands = setpoint = 0, no clipping (quadx.py:427-429)
      const double T = (double)Pk->control_period, iT = rcp_d(T);  // (fp64 like the flight's own cascade: quadx_control_d.hpp)
      double zc = (K.mode == 1 || K.mode == 5 || K.mode == 6) ? 0.0 : (double)z_hold;
      if (!(K.mode == 1 || K.mode == 5 || K.mode == 6))
        zc = pid1d(Pk->zpid[1].kp[0], Pk->zpid[1].ki[0], Pk->zpid[1].kd[0], Pk->zpid[1].lim[0], T, iT, C.zI[1], C.zE[1], (double)z, zc);
      zc = pid1d(Pk->zpid[0].kp[0], Pk->zpid[0].ki[0], Pk->zpid[0].kd[0], Pk->zpid[0].lim[0], T, iT, C.zI[0], C.zE[0], (double)vz, zc);
      pwm_s = (float)clampd(clampd(zc, 0.0, 1.0), 0.05, 1.0);
    };
    auto settle_tick = [&](float xi) {
      float s = fmaf(xi, K.m_noise, 1.0f);
      thr = fmaf(K.m_a, pwm_s - thr, thr) * s;
      float kk = thr * __builtin_fabsf(thr);
      float Fz = fmaf(-K.drag[2], vz * __builtin_fabsf(vz), 4.0f * (K.fmax * kk));
      float az = fmaf(K.inv_mass, Fz, K.gravity_z);
      vz = med3(fmaf(az, K.dt, vz), -K.vmax, K.vmax);
      z = fmaf(K.dt, vz, z);
    };
    auto settle_noise = [&](int t) {  // four draws per read
      float4 x;
      if (NOISE == PF_NOISE_PHILOX) x = reinterpret_cast<const float4*>(sxi + tid * kSettleMax)[t >> 2];
      else if (NOISE == PF_NOISE_INJECT) x = float4{B.xi_reset[(size_t)(t + 0) * N + li], B.xi_reset[(size_t)(t + 1) * N + li],
                                                    B.xi_reset[(size_t)(t + 2) * N + li], B.xi_reset[(size_t)(t + 3) * N + li]};
      else x = float4{0.f, 0.f, 0.f, 0.f};
      return x;
    };
    if (!MODES || K.mode == 0) {
      // Mode 0: the motor command is the constant 0.05, so the throttle recurrence does not depend on the vertical state and the
      // 20 ticks are TWO dependency chains -- throttle (3 instructions per tick) and climb rate (3-4 per tick):
      //   vz' = clamp(vz + dt (F/m + g)) with F/m + g = 4 fmax/m t|t| + g - drag/m vz|vz|
      //       = clamp(fma(-dt drag/m, vz|vz|, vz + e)),  e = fma(4 dt fmax/m, t|t|, dt g).
      // The chains are interleaved statement by statement (the build runs with the machine scheduler off: statement order is the
      // schedule; a lone wave pays per INSTRUCTION, 4-5 clocks each whether dependent or not -- profiles/r06/lone_wave_issue.txt --, so
      // what the interleaving buys is the wait states behind a compare / a packed result filled with work instead of s_nop), the throttle
      // chain running one chunk of four ticks ahead of the climb-rate chain. (The few lanes of a wave that reset are what the
      // whole wave waits for: the serial 10-instructions-per-tick version was 0.85 us of every env step.)
      const float A4 = 4.0f * (K.fmaxM * K.dt), G = K.gravity_z * K.dt, Dd = K.dragM[2] * K.dt;
      float e0, e1, e2, e3;
      {  // throttle chain, chunk 0
        const float4 x = settle_noise(0);
        const float s0 = fmaf(x.x, K.m_noise, 1.0f), s1 = fmaf(x.y, K.m_noise, 1.0f), s2 = fmaf(x.z, K.m_noise, 1.0f), s3 = fmaf(x.w, K.m_noise, 1.0f);
        thr = fmaf(K.m_a, 0.05f - thr, thr) * s0; e0 = fmaf(A4, thr * __builtin_fabsf(thr), G);
        thr = fmaf(K.m_a, 0.05f - thr, thr) * s1; e1 = fmaf(A4, thr * __builtin_fabsf(thr), G);
        thr = fmaf(K.m_a, 0.05f - thr, thr) * s2; e2 = fmaf(A4, thr * __builtin_fabsf(thr), G);
        thr = fmaf(K.m_a, 0.05f - thr, thr) * s3; e3 = fmaf(A4, thr * __builtin_fabsf(thr), G);
      }
      // one tick of each chain, interleaved: T = throttle chain (chunk k + 1), V = climb-rate chain (chunk k)
#define PF_SETTLE_PAIR(S_, E_IN, E_OUT)                                      \
      { const float dl_ = 0.05f - thr;               /* T */                  \
        const float d_ = vz + (E_IN);                /* V */                  \
        const float tn_ = fmaf(K.m_a, dl_, thr);     /* T */                  \
        const float q_ = vz * __builtin_fabsf(vz);   /* V */                  \
        thr = tn_ * (S_);                            /* T */                  \
        const float vn_ = fmaf(-Dd, q_, d_);         /* V */                  \
        const float kk_ = thr * __builtin_fabsf(thr);/* T */                  \
        vz = med3(vn_, -K.vmax, K.vmax);             /* V */                  \
        (E_OUT) = fmaf(A4, kk_, G);                  /* T */                  \
        z = fmaf(K.dt, vz, z); }                     /* V */
      for (int t = 4; t < settle_ticks; t += 4) {
        const float4 x = settle_noise(t);
        const float s0 = fmaf(x.x, K.m_noise, 1.0f), s1 = fmaf(x.y, K.m_noise, 1.0f), s2 = fmaf(x.z, K.m_noise, 1.0f), s3 = fmaf(x.w, K.m_noise, 1.0f);
        float n0, n1, n2, n3;
        PF_SETTLE_PAIR(s0, e0, n0)
        PF_SETTLE_PAIR(s1, e1, n1)
        PF_SETTLE_PAIR(s2, e2, n2)
        PF_SETTLE_PAIR(s3, e3, n3)
        e0 = n0; e1 = n1; e2 = n2; e3 = n3;
      }
#undef PF_SETTLE_PAIR
      {  // climb-rate chain, last chunk
        const float ee[4] = {e0, e1, e2, e3};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float d_ = vz + ee[k], q_ = vz * __builtin_fabsf(vz);
          vz = med3(fmaf(-Dd, q_, d_), -K.vmax, K.vmax);
          z = fmaf(K.dt, vz, z);
        }
      }
    } else {
      for (int t = 0; t < settle_ticks; t += 4) {  // four ticks per trip, the mode's z PIDs once per Aviary step
        const float4 x = settle_noise(t);
        settle_control(); settle_tick(x.x); settle_tick(x.y);  // (two ticks per control: quadk_from_params)
        settle_control(); settle_tick(x.z); settle_tick(x.w);
      }
    }
  };
  // (DSTATE) the same vertical recurrence in fp64, the z PIDs on fp64 memories: what the oracle's settle Aviary steps compute for a level
  // spawn at rest. (A float32 settle leaves the episode's first state 5e-7 off the reference's, and a cascaded controller that has just
  // been handed its setpoint turns that into 1e-4 within four env steps: profiles/tools/r06/dbg_m7.py.)
  auto settle_run_d = [&](double& z, double& vz, double& thr, double& pwm_s) {
    pwm_s = 0.05;
    const double z_hold = z;
    C.zero();
    const double T = (double)Pk->control_period, iT = rcp_d(T);
    auto ctl = [&]() {
      if (K.mode == -1) { pwm_s = 0.0; return; }
      double zc = (K.mode == 1 || K.mode == 5 || K.mode == 6) ? 0.0 : z_hold;
      if (!(K.mode == 1 || K.mode == 5 || K.mode == 6))
        zc = pid1d(Pk->zpid[1].kp[0], Pk->zpid[1].ki[0], Pk->zpid[1].kd[0], Pk->zpid[1].lim[0], T, iT, C.zI[1], C.zE[1], z, zc);
      zc = pid1d(Pk->zpid[0].kp[0], Pk->zpid[0].ki[0], Pk->zpid[0].kd[0], Pk->zpid[0].lim[0], T, iT, C.zI[0], C.zE[0], vz, zc);
      pwm_s = clampd(clampd(zc, 0.0, 1.0), 0.05, 1.0);
    };
    auto tk = [&](const float xi) {
      thr += (double)K.m_a * (pwm_s - thr);
      thr += (double)xi * thr * (double)K.m_noise;
      const double kk = thr * __builtin_fabs(thr);
      const double az = -(double)K.dragM[2] * (vz * __builtin_fabs(vz)) + (double)K.fmaxM * (4.0 * kk) + (double)K.gravity_z;
      const double vn = vz + az * (double)K.dt;
      vz = vn < -(double)K.vmax ? -(double)K.vmax : (vn > (double)K.vmax ? (double)K.vmax : vn);
      z += (double)K.dt * vz;
    };
    for (int t = 0; t < settle_ticks; t += 4) {
      float4 x;
      if (NOISE == PF_NOISE_PHILOX) x = reinterpret_cast<const float4*>(sxi + tid * kSettleMax)[t >> 2];
      else if (NOISE == PF_NOISE_INJECT) x = float4{B.xi_reset[(size_t)(t + 0) * N + li], B.xi_reset[(size_t)(t + 1) * N + li],
                                                    B.xi_reset[(size_t)(t + 2) * N + li], B.xi_reset[(size_t)(t + 3) * N + li]};
      else x = float4{0.f, 0.f, 0.f, 0.f};
      ctl(); tk(x.x); tk(x.y);
      ctl(); tk(x.z); tk(x.w);
    }
  };
  // env.reset() for this lane: begin_reset + waypoint sampling + set_mode(0) + the settle phase
  // (quadx_base_env.py:149-212). Level spawn at rest under the mode-0 default setpoint
  // (quadx.py:276-278): rate error 0 -> cmd 0 -> pwm 0.05 on all four motors (quadx.py:488 branch
  // skipped, :493 clip); equal thrusts cancel every torque exactly, so the settle ticks are a
  // vertical (z, vz, throttle) recurrence.
  auto reset_lane = [&](const bool have_pre, const Sp& pre) {
    // MA hover: the agent's own spawn pose from the side block (level by quadk_from_params' contract:
    // the host passes the least level / lowest agent as the parameter block's start pose)
    const float sx = TASK == PF_TASK_MA_HOVER ? tgt[0][0] : K.start_pos[0], sy = TASK == PF_TASK_MA_HOVER ? tgt[0][1] : K.start_pos[1];
    float thr = 0.f, vz = 0.f, z = TASK == PF_TASK_MA_HOVER ? tgt[0][2] : K.start_pos[2];
    // Flight modes other than 0 (MODES): set_mode's default setpoint holds the spawn pose (quadx.py:233-373), every attitude,
    // rate and lateral error is exactly zero on a level spawn at rest, so the cascade reduces to its z PIDs -- run here once per
    // Aviary step on the vertical state -- and four equal motor commands: still a vertical recurrence.
    float pwm_s = 0.05f;
    double zd = z, vzd = 0.0, thrd = 0.0, pwmd = 0.05;  // (DSTATE)
    if (SPARE && have_pre) { z = pre.z; vz = pre.vz; thr = pre.thr; C.zero(); }  // (the next episode's settled state, computed ahead: make_spare)
    else if (DSTATE) { settle_run_d(zd, vzd, thrd, pwmd); z = (float)zd; vz = (float)vzd; thr = (float)thrd; pwm_s = (float)pwmd; }
    else settle_run(z, vz, thr, pwm_s);
    V.p = v3{sx, sy, z};
    if (TASK == PF_TASK_MA_HOVER) V.q = quat{tgt[1][0], tgt[1][1], tgt[1][2], tgt[2][0]};
    else V.q = quat{K.start_quat[0], K.start_quat[1], K.start_quat[2], K.start_quat[3]};
    V.set_wv(v3{0.f, 0.f, 0.f}, v3{0.f, 0.f, vz});
    V.t01 = V.t23 = sp2(thr); V.pw01 = V.pw23 = sp2(pwm_s);
#pragma unroll
    for (int k = 0; k < 3; ++k) { V.I[k] = 0.f; V.E[k] = 0.f; }
    V.contact_now = false; V.contact_step = false;
    V.derive();
    if (DSTATE) {  // (the settle recurrence ran in fp64: settle_run_d)
      SD.p[0] = V.p.x; SD.p[1] = V.p.y; SD.p[2] = zd; SD.q[0] = V.q.x; SD.q[1] = V.q.y; SD.q[2] = V.q.z; SD.q[3] = V.q.w;
      SD.v[0] = 0.0; SD.v[1] = 0.0; SD.v[2] = vzd; SD.w[0] = 0.0; SD.w[1] = 0.0; SD.w[2] = 0.0;
#pragma unroll
      for (int k = 0; k < 4; ++k) { SD.thr[k] = thrd; SD.pwm[k] = pwmd; }
#pragma unroll
      for (int k = 0; k < 3; ++k) { SD.rI[k] = 0.0; SD.rE[k] = 0.0; }
      SD.derive();
    }
    step_count = 0; term = false; trunc = false; flags = 0; pop_pending = false;
    act0 = act1 = act2 = act3 = 0.f;  // (MA hover: the action memories live in the side block and survive resets)
    if (TASK == PF_TASK_WAYPOINTS) {  // waypoint_handler.py:53-83
      const int nt = K.num_targets;
      n_left = nt;
      if (SPARE && have_pre) {  // prepared ahead (make_spare)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (i < nt) {
            tgt[i][0] = pre.t[i].x; tgt[i][1] = pre.t[i].y; tgt[i][2] = pre.t[i].z;
            if (kYaw) ytg[i] = pre.t[i].w;
          }
        }
      } else if (coop_targets) {  // sampled by prepare_reset_draws(): this lane's 4 x (x, y, z, yaw)
        const float4* t4 = reinterpret_cast<const float4*>(tile + 64 * 16 + tid * 16);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (i < nt) {
            const float4 t = t4[i];
            tgt[i][0] = t.x; tgt[i][1] = t.y; tgt[i][2] = t.z;
            if (kYaw) ytg[i] = t.w;
          }
        }
      } else {
      f4 u0, u1, u2, u3 = f4{0.f, 0.f, 0.f, 0.f};
      const bool inj = (NOISE == PF_NOISE_INJECT) && (B.u_targets != nullptr);
      if (!inj) {
        u0 = uniform4(philox4x32(K.seed_lo, K.seed_hi, (uint32_t)(lane0 + li), reset_key_now, 0u, 2u));
        u1 = uniform4(philox4x32(K.seed_lo, K.seed_hi, (uint32_t)(lane0 + li), reset_key_now, 1u, 2u));
        u2 = uniform4(philox4x32(K.seed_lo, K.seed_hi, (uint32_t)(lane0 + li), reset_key_now, 2u, 2u));
        if (kYaw) u3 = uniform4(philox4x32(K.seed_lo, K.seed_hi, (uint32_t)(lane0 + li), reset_key_now, 3u, 2u));
      }
      auto u = [&](int flat) { return pick4(flat < 4 ? u0 : (flat < 8 ? u1 : (flat < 12 ? u2 : u3)), (uint32_t)flat & 3u); };
      if (kYaw) {  // waypoint_handler.py:85-89: uniform(-pi, pi), drawn after all the positions
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (i < nt) ytg[i] = inj ? B.u_targets[(size_t)(3 * nt + i) * N + li] : fmaf(2.0f * kPi, u(3 * nt + i), -kPi);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (i < nt) {
          float theta, phi, dist;  // theta, phi in turns
          if (inj) {
            theta = B.u_targets[(size_t)i * N + li] * (0.5f / kPi);  // injected as angles; turns below
            phi = B.u_targets[(size_t)(nt + i) * N + li] * (0.5f / kPi);
            dist = B.u_targets[(size_t)(2 * nt + i) * N + li];
          } else {
            theta = u(i);
            phi = u(nt + i);
            dist = fmaf(K.dome09m1, u(2 * nt + i), 1.0f);
          }
          float st, ct, sph, cph;
          sincos_turns(theta, st, ct);
          sincos_turns(phi, sph, cph);
          float zz = __builtin_fabsf(dist * cph);
          tgt[i][0] = dist * sph * ct; tgt[i][1] = dist * sph * st; tgt[i][2] = zz > K.min_height ? zz : K.min_height;
        }
      }
      }
      // end_reset's compute_state: distance to the first target (old distance = inf)
      float dx = tgt[0][0] - V.p.x, dy = tgt[0][1] - V.p.y, dz = tgt[0][2] - V.p.z;
      old_dist = INFINITY;
      new_dist = fsqrt(fmaf(dx, dx, fmaf(dy, dy, dz * dz)));
      if (kYaw) yaw_err0 = __builtin_fabsf(wrap_pi(ytg[0] - yaw_now()));
    }
    rng_ctr += 1;
    if (REKEY) { rkw = rng_ctr & ~kSpareValid; sp_dirty = true; }  // the NEXT reset's key: the counter as this reset leaves it (strictly increasing); its spare is still to be made
    was_reset = true;
  };
  // The next episode's random part for the lanes that ask (wave-uniform call): the draws keyed by `key`, the settle recurrence, the
  // targets -- the very code a reset without a spare runs.
  auto make_spare = [&](const bool me, const uint32_t key, Sp& out) {
    prepare_reset_draws(me, key);
    if (me) {
      float z = K.start_pos[2], vz = 0.f, thr = 0.f, pw = 0.05f;
      settle_run(z, vz, thr, pw);
      out.z = z; out.vz = vz; out.thr = thr;
      if (TASK == PF_TASK_WAYPOINTS) {
        const float4* t4 = reinterpret_cast<const float4*>(tile + 64 * 16 + tid * 16);
#pragma unroll
        for (int i = 0; i < 4; ++i) out.t[i] = t4[i];
      }
    }
  };
  auto spare_get = [&]() -> Sp {  // (SPARE) this lane's spare: the registers it was loaded into, or its LDS row (pf_rollout)
    if (!ROLLOUT) return spv;
    Sp x;
    const float4* r4 = reinterpret_cast<const float4*>(splds);
    const float4 a0 = r4[0];
    x.z = a0.x; x.vz = a0.y; x.thr = a0.z;
    if (TASK == PF_TASK_WAYPOINTS) { x.t[0] = r4[1]; x.t[1] = r4[2]; x.t[2] = r4[3]; x.t[3] = r4[4]; }
    return x;
  };
  auto spare_put = [&](const Sp& x) {
    if (!ROLLOUT) { spv = x; return; }
    float4* r4 = reinterpret_cast<float4*>(splds);
    r4[0] = float4{x.z, x.vz, x.thr, 0.0f};
    if (TASK == PF_TASK_WAYPOINTS) { r4[1] = x.t[0]; r4[2] = x.t[1]; r4[3] = x.t[2]; r4[4] = x.t[3]; }
  };
  // A wave's resets: the lanes with a valid spare copy it, the others generate (cooperatively) from their key. Wave-uniform call.
  auto do_resets = [&](const bool r) {
    const bool have = SPARE && r && (rkw & kSpareValid) != 0u;
    reset_key_now = REKEY ? (rkw & ~kSpareValid) : rng_ctr;
    prepare_reset_draws(r && !have, reset_key_now);
    // (MODES: behind a wave-uniform guard -- the fp64 state a reset writes otherwise meets the flying lanes' at this block's join, and
    //  the allocator put those copies in front of the exec restore: tools/isa_exec_check.py. The mode-0 instantiations keep their shape)
    if (!MODES || __any(r)) {
      if (r) {
        if (SPARE && __builtin_expect(have, 1)) reset_lane(true, spare_get());
        else reset_lane(false, Sp{});
      }
    }
  };
  // The spares of the lanes that have used theirs up, once every kSpareEvery env steps (and with every explicit reset). Wave-uniform.
  auto refill_spares = [&](const bool now) {
    if (!SPARE || !now) return;
    const bool want = active && (rkw & kSpareValid) == 0u;
    if (!__any(want)) return;
    Sp g{};
    make_spare(want, rkw & ~kSpareValid, g);
    if (want) { spare_put(g); rkw |= kSpareValid; sp_dirty = true; }
  };
  // observation row (Appendix A of SURVEY.md) -> LDS tile. The attitude quaternion is
  // getQuaternionFromEuler(getEulerFromQuaternion(q)) (quadx_base_env.py:243), computed without trig.
  auto write_obs_row = [&]() {
    float* row = tile + tid * D;
    // (the Euler-angle arguments are entries of the rotation matrix derive() holds for the unit q:
    //  -2(xz - wy) = -R20, 2(yz + wx) = R21, w2-x2-y2+z2 = R22, 2(xy + wz) = R10, w2+x2-y2-z2 = R00)
    float sarg = -V.R.m20;
    quat qe;
    v3 rpy;
    const float obs_yaw = kYaw ? yaw_now() : 0.0f;
    if (__builtin_fabsf(sarg) >= 0.99999f) {  // gimbal-lock branch of pybullet, rare: library trig
      rpy = euler_from_quat(V.q);
      qe = quat_from_euler(rpy);
    } else {
      float ar = V.R.m21, br = V.R.m22;
      float ay = V.R.m10, by = V.R.m00;
      float hr = frsq(fmaf(ar, ar, br * br)), hy = frsq(fmaf(ay, ay, by * by));
      float cr, sr, cp, sp, cy, sy;
      half_angle(br * hr, ar * hr, cr, sr);
      half_angle(fsqrt((1.0f - sarg) * (1.0f + sarg)), sarg, cp, sp);
      half_angle(by * hy, ay * hy, cy, sy);
      quat t = quat_from_half_angles(cr, sr, cp, sp, cy, sy);
      float inv = frsq(fmaf(t.x, t.x, fmaf(t.y, t.y, fmaf(t.z, t.z, t.w * t.w))));
      qe = quat{t.x * inv, t.y * inv, t.z * inv, t.w * inv};
      if (!K.angle_repr) rpy = v3{fast_atan2(ar, br), fast_asin(sarg), fast_atan2(ay, by)};
    }
    int k = 0;
    row[k++] = V.wb.x; row[k++] = V.wb.y; row[k++] = V.wb.z;
    if (K.angle_repr) { row[k++] = qe.x; row[k++] = qe.y; row[k++] = qe.z; row[k++] = qe.w; }
    else { row[k++] = rpy.x; row[k++] = rpy.y; row[k++] = rpy.z; }
    row[k++] = V.vb.x; row[k++] = V.vb.y; row[k++] = V.vb.z;
    row[k++] = V.p.x; row[k++] = V.p.y; row[k++] = V.p.z;
    if (TASK == PF_TASK_MA_HOVER) {  // ma_quadx_hover_env.py:141-166: aux, past action, start_pos
      row[k++] = V.t01.x; row[k++] = V.t01.y; row[k++] = V.t23.x; row[k++] = V.t23.y;
      row[k++] = ma_past.x; row[k++] = ma_past.y; row[k++] = ma_past.z; row[k++] = ma_past.w;
      row[k++] = tgt[0][0]; row[k++] = tgt[0][1]; row[k++] = tgt[0][2];
    } else {
      row[k++] = act0; row[k++] = act1; row[k++] = act2; row[k++] = act3;
      row[k++] = V.t01.x; row[k++] = V.t01.y; row[k++] = V.t23.x; row[k++] = V.t23.y;
    }
    if (TASK == PF_TASK_WAYPOINTS) {
      m3 Re = rot_from_quat(qe);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (i < K.num_targets) {
          v3 d = mulT(Re, v3{tgt[i][0] - V.p.x, tgt[i][1] - V.p.y, tgt[i][2] - V.p.z});
          bool live = i < n_left;
          row[k++] = live ? d.x : 0.0f; row[k++] = live ? d.y : 0.0f; row[k++] = live ? d.z : 0.0f;
          if (kYaw) row[k++] = live ? wrap_pi(ytg[i] - obs_yaw) : 0.0f;  // waypoint_handler.py:144-153
        }
      }
    }
  };
  const bool wave_all = __all(active || !valid);
  auto flush_tile = [&](float* out) {
    lds_sync();
    if (wave_all) {
      const int rows = min(LPW, n - wave_base);
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

  const int KS = ROLLOUT ? k_steps : 1;
  float4 a_nxt = float4{0.f, 0.f, 0.f, 0.f};
  if (GIVEN) a_nxt = reinterpret_cast<const float4*>(B.actions)[li];
  for (int it = 0; it < KS; ++it) {
  const size_t toff = ROLLOUT ? (size_t)it * N : (size_t)0;  // this step's slot in the trajectory buffers (lanes)
  // ---------------------------------------------------------------- reset (NEXT_STEP / explicit)
  bool do_reset;
  if (op == 1) do_reset = active;
  else do_reset = (K.autoreset == PF_AUTORESET_NEXT_STEP) && (term || trunc) && active;
  act0 = act1 = act2 = act3 = 0.f;
  reward = 0.0f;
  was_reset = false;
  do_resets(do_reset);
  // (the counter word is first needed HERE, behind the resets: read where the loop starts, its wait sat in front of them -- the load is
  //  the last of the prologue's, and a lone wave of a 4 096-lane batch waited 0.45 us for it. The same word in every lane: made
  //  wave-uniform for the control flow below)
  const uint32_t call0_u = SPARE ? (uint32_t)__builtin_amdgcn_readfirstlane((int)call0) : 0u;
  refill_spares(op == 1 || ((call0_u + (uint32_t)it) % kSpareEvery) == kSpareEvery - 1u);
  PF_STAMP(4);  // (NEXT_STEP resets done)

  // ---------------------------------------------------------------- the env step
  const bool stepping = active && !was_reset && op == 0;
  float sp0 = 0.f, sp1 = 0.f, sp2 = 0.f, sp3 = 0.f;
  float4 a_roll = float4{0.f, 0.f, 0.f, 0.f};
  if (ROLLOUT) {  // this step's action for every lane: given sequence (prefetched one step ahead) or sampled
    if (GIVEN) {
      a_roll = a_nxt;
      if (it + 1 < KS) a_nxt = reinterpret_cast<const float4*>(B.actions)[toff + N + li];
    } else {  // == sample_actions_kernel(step0 + it): same Philox key, same arithmetic
      f4 u = uniform4(philox4x32(K.seed_lo, K.seed_hi, (uint32_t)(lane0 + li), step0 + (uint32_t)it, 0u, 3u));
      a_roll = float4{fmaf(K.act_span[0], u.a, K.act_lo[0]), fmaf(K.act_span[1], u.b, K.act_lo[1]),
                      fmaf(K.act_span[2], u.c, K.act_lo[2]), fmaf(K.act_span[3], u.d, K.act_lo[3])};
    }
    if (!GIVEN && B.actions_out != nullptr && active) {
      float* ao = B.actions_out + 4 * (toff + li);
      __builtin_nontemporal_store(a_roll.x, ao + 0); __builtin_nontemporal_store(a_roll.y, ao + 1);
      __builtin_nontemporal_store(a_roll.z, ao + 2); __builtin_nontemporal_store(a_roll.w, ao + 3);
    }
  }
  if (stepping) {
    const float4 a = ROLLOUT ? a_roll : a_pre;
    act0 = a.x; act1 = a.y; act2 = a.z; act3 = a.w;
    sp0 = a.x; sp1 = a.y; sp2 = a.z; sp3 = a.w;
    reward = -0.1f;
    if (TASK == PF_TASK_MA_HOVER) {  // past <- current, current <- action (ma_quadx_base_env.py:326-332)
      ma_past = float4{tgt[2][1], tgt[2][2], tgt[3][0], tgt[3][1]};
      tgt[2][1] = a.x; tgt[2][2] = a.y; tgt[3][0] = a.z; tgt[3][1] = a.w;
      reward = 0.0f;
      term = false; trunc = false;  // per-call flags (:336-337); no early exit from the inner loop (:342-361)
    }
  }
  bool go = stepping && !(term || trunc);  // quadx_base_env.py:289-290
  // Calm waves. The contact response's call site costs the tick loop about 0.2 us per tick even when it is never taken (the
  // values that live across it are pinned to the callee-saved registers, one more divergent region per tick), and in the env
  // tasks it almost never is: the drones fly at z = 1 and the few that sink to the floor are a handful per launch. A wave none of
  // whose lanes can come within reach of the floor during this env step runs the ticks instantiated without the call -- the
  // same arithmetic (no lane near: no solve, lift = 0), so the results are identical bit for bit. The bound: over the step's
  // duration T the vertical acceleration is at most A = g + thrust + drag in magnitude, with thrust <= 4 fmax / m x (largest
  // motor state^2, grown by the largest noise factor), drag <= c_max |v|^2 and |v| <= |v0| + (g + thrust) T (the drag only
  // dissipates); the body sinks by at most |vz0| T + A T (T + dt) / 2 (semi-implicit Euler, the velocity clamp only shrinks).
  // (instantiated where the env benchmarks live -- flight mode 0, noise drawn on device or off; in the cascaded-mode and
  //  injected-noise instantiations the second copy of the ticks cost registers they do not have: stack spills)
  // NF: this instantiation's calm ticks without the floor detection and the rotational-drag gate (tick<.., NOFLOOR>, see there)
  // (the one-wave-per-SIMD instantiations only: with two waves' 256 registers the Hover rollout without the detection came out of the
  //  compiler with 31 vector copies in front of an exec restore -- tools/isa_exec_check.py stopped the build; at 524 288 lanes the one-step
  //  launch had gained 0.7 %)
  // (the PettingZoo task with independent lanes as well: its facade's env.step() 10.93 -> 10.32 us captured, 10.68 -> 9.99 eager -- same box,
  //  profiles/r06/ab_calm_tick_nofloor_same_box.txt)
  constexpr bool NF = WPS == 1 && (TASK == PF_TASK_HOVER || TASK == PF_TASK_MA_HOVER || (TASK == PF_TASK_WAYPOINTS && ROLLOUT));
  // NFX: ... and the collision's consequence evaluated in the not-calm branch instead of behind the join (a calm wave skips it: Hover 9.264 ->
  // 9.250 us, 7.26 -> 7.21 at 4 096 lanes, pf_rollout 5.42 -> 5.34; bit-identical)
  constexpr bool NFX = NF && CALM;
  bool calm = false;
  // the same bound over any horizon T (TT = T (T + dt) / 2): how far a lane can sink within it
  auto sink_within = [&](const float T, const float TT) {
    const float tm = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(V.t01.x), __builtin_fabsf(V.t01.y)), __builtin_fmaxf(__builtin_fabsf(V.t23.x), __builtin_fabsf(V.t23.y)));
    const float a_nd = fmaf(K.calm_kt, __builtin_fmaxf(tm * tm, K.calm_t2), __builtin_fabsf(K.gravity_z));
    const float u = fsqrt(fmaf(V.wvx.y, V.wvx.y, fmaf(V.wvy.y, V.wvy.y, V.wvz.y * V.wvz.y))) + a_nd * T;
    return fmaf(__builtin_fabsf(V.wvz.y), T, fmaf(K.calm_c, u * u, a_nd) * TT);
  };
  if (CALM && K.calm_on) {
    // (... and holds no contact point: the calm ticks leave the rotational-drag gate out as well -- tick<.., NOFLOOR>)
    calm = __all(!go || ((!NF || !V.contact_now) && V.p.z - fmaf(sink_within(K.calm_T, K.calm_TT), 1.01f, 1e-3f) > K.bound_radius));  // (wave-uniform; NaN compares false: not calm)
#ifdef PF_PHASE_TRACE
    if (!calm && tid == 0) trace_add(&g_calm_trace[0], 1ull);  // waves that keep the call site this step (rare: no contention)
#endif
  }
  for (int s = 0; s < K.env_step_ratio; ++s) {
    if (!__any(go)) break;
    // a wave that is not calm over the whole env step (some lane is low) still is over most of its Aviary steps: the same bound
    // over the two ticks ahead (a lane sinks a few millimetres in them) -- evaluated in those waves only
    bool calm_s = calm;
    if (CALM && K.calm_on && !calm) {
      calm_s = __all(!go || ((!NF || !V.contact_now) && V.p.z - fmaf(sink_within(K.calm_T2, K.calm_TT2), 1.01f, 1e-3f) > K.bound_radius));
#ifdef PF_PHASE_TRACE
      if (!calm_s && tid == 0) trace_add(&g_calm_trace[1], 1ull);  // Aviary steps that keep the call site
#endif
    }
    if (go) {
      float xi0, xi1;
      if (NOISE == PF_NOISE_PHILOX) { xi0 = 4.0f + pick8(zn, (uint32_t)(2 * s)); xi1 = 4.0f + pick8(zn, (uint32_t)(2 * s + 1)); }
      else if (NOISE == PF_NOISE_INJECT) { xi0 = B.xi[(size_t)(2 * s) * N + li]; xi1 = B.xi[(size_t)(2 * s + 1) * N + li]; }
      else { xi0 = 0.f; xi1 = 0.f; }
      // one Aviary.step (aviary.py:480-531): control on the first tick, pwm held on the second
      V.contact_step = false;
      V.template control<MODES>(K, Pk, C, sp0, sp1, sp2, sp3, DSTATE ? &SD : nullptr);
      if (SHARED) {
        // one world for the agents of an env: pose / contact exchange before every tick. Every lane of a world is in here
        // together: the PettingZoo task has no early exit from the inner loop and resets whole worlds.
        quad_world_exchange(V, wpose, tid, apw, K);
        V.template tick<CR, true>(K, K, xi0, Pfull);
        quad_world_exchange(V, wpose, tid, apw, K);
        V.template tick<CR, true>(K, K, xi1, Pfull);
        V.peer_contact = false;
      } else if (CALM && __builtin_expect(calm_s, 1)) {
        // (NOFLOOR where it measured faster, same box, bit-identical over 400 steps x 65 536 lanes: Hover 9.60 -> 9.25 us per step, 7.71 ->
        //  7.26 at 4 096 lanes, pf_rollout 5.80 -> 5.35; QuadX-Waypoints' pf_rollout 7.50 -> 7.02 -- but its one-step launch 15.72 -> 16.19:
        //  without the detection its allocation spills 169 scalar registers for 145; profiles/r06/ab_calm_tick_nofloor_same_box.txt)
        V.template tick<false, false, QuadKV, false, true, NF>(KV, K, xi0, Pfull);
        V.template tick<false, false, QuadKV, false, true, NF>(KV, K, xi1, Pfull);
      } else if (CALM && WPS == 1) {
        // (not calm: the same ticks with the floor code in them. Their flight-path constants from the vector registers as well --
        //  the floor code's own constants stay scalar, Kc)
        V.template tick<CR, false, QuadKV, true>(KV, K, xi0, Pfull);
        V.template tick<CR, false, QuadKV, true>(KV, K, xi1, Pfull);
        // (NFX: the collision's consequence HERE, where a contact can have been reported -- the calm ticks cannot report one; it commutes with
        //  what stands between here and its place below: quadx_base_env.py:258-267 assign the same reward, the PettingZoo task's penalties add)
        if (NFX && V.contact_step) {
          if (TASK == PF_TASK_MA_HOVER) reward -= 100.0f; else reward = -100.0f;
          flags |= PF_F_INFO_COLLISION; term = true;
        }
      } else if (DSTATE) {
        V.template tick_d<CR, WPS == 1>(SD, K, xi0, Pfull);
        V.template tick_d<CR, WPS == 1>(SD, K, xi1, Pfull);
      } else {
        V.template tick<CR, false, QuadK, WPS == 1, !(TASK == PF_TASK_HOVER && WPS == 2)>(K, K, xi0, Pfull);
        V.template tick<CR, false, QuadK, WPS == 1, !(TASK == PF_TASK_HOVER && WPS == 2)>(K, K, xi1, Pfull);
      }
      // compute_state side effects + compute_term_trunc_reward
      if (TASK == PF_TASK_WAYPOINTS) {  // waypoint_handler.py:135-142; ||R^T d|| = ||d||
        if (pop_pending) { pop_target(); pop_pending = false; }
        float dx = tgt[0][0] - V.p.x, dy = tgt[0][1] - V.p.y, dz = tgt[0][2] - V.p.z;
        old_dist = new_dist;
        new_dist = fsqrt(fmaf(dx, dx, fmaf(dy, dy, dz * dz)));
        if (kYaw) yaw_err0 = __builtin_fabsf(wrap_pi(ytg[0] - yaw_now()));
      }
      if (step_count > K.max_steps) trunc = true;                                          // quadx_base_env.py:254
      if (TASK == PF_TASK_MA_HOVER) {  // ma_quadx_hover_env.py:168-205: additive penalties, no early exit
        if (!NFX && V.contact_step) { reward -= 100.0f; flags |= PF_F_INFO_COLLISION; term = true; }
        if (dot(V.p, V.p) > K.dome2) { reward -= 100.0f; flags |= PF_F_INFO_OOB; term = true; }
        if (!K.task_sparse) {
          float dx = V.p.x - tgt[0][0], dy = V.p.y - tgt[0][1], dz = V.p.z - tgt[0][2];
          float lin = fsqrt(fmaf(dx, dx, fmaf(dy, dy, dz * dz)));
          // roll, pitch of getEulerFromQuaternion from the rotation matrix derive() just built (unit q):
          // -2(xz - wy) = -R20, 2(yz + wx) = R21, w^2 - x^2 - y^2 + z^2 = R22
          float sarg = -V.R.m20;
          bool gim = __builtin_fabsf(sarg) >= 0.99999f;
          float roll = gim ? 0.0f : fast_atan2(V.R.m21, V.R.m22);
          float pitch = gim ? __builtin_copysignf(0.5f * kPi, sarg) : fast_asin(sarg);
          float ang = fsqrt(fmaf(roll, roll, pitch * pitch));
          reward -= fmaf(ang, 0.1f, lin);
          reward += 1.0f;
        }
      } else {
      if (!NFX && V.contact_step) { reward = -100.0f; flags |= PF_F_INFO_COLLISION; term = true; } // :258-261
      if (dot(V.p, V.p) > K.dome2) { reward = -100.0f; flags |= PF_F_INFO_OOB; term = true; } // :264-267
      }
      if (TASK == PF_TASK_HOVER) {
        if (!K.task_sparse) {  // quadx_hover_env.py:120-138
          float dz = V.p.z - 1.0f;
          float lin = fsqrt(fmaf(V.p.x, V.p.x, fmaf(V.p.y, V.p.y, dz * dz)));
          // roll, pitch of getEulerFromQuaternion (gimbal branch: roll = 0, |pitch| = pi/2)
          // (from the rotation matrix derive() just built, unit q: -2(xz - wy) = -R20, 2(yz + wx) = R21,
          //  w^2 - x^2 - y^2 + z^2 = R22)
          float sarg = -V.R.m20;
          bool gim = __builtin_fabsf(sarg) >= 0.99999f;
          float roll = gim ? 0.0f : fast_atan2(V.R.m21, V.R.m22);
          float pitch = gim ? __builtin_copysignf(0.5f * kPi, sarg) : fast_asin(sarg);
          float ang = fsqrt(fmaf(roll, roll, pitch * pitch));
          reward = fmaf(-0.01f * V.wb.z, V.wb.z, reward);
          reward -= lin + ang;
          reward += 1.0f;
        }
      } else if (TASK == PF_TASK_WAYPOINTS) {
        if (!K.task_sparse) {  // quadx_waypoints_env.py:183-192
          float progress = (isinf(old_dist + new_dist)) ? 0.0f : old_dist - new_dist;
          reward += __builtin_fmaxf(3.0f * progress, 0.0f);
          reward = fmaf(K.wp_dist_reward, frcp(new_dist), reward);
          reward = fmaf(-K.wp_yaw_penalty * V.wb.z, V.wb.z, reward);
        }
        if (new_dist < K.goal_reach && (!kYaw || yaw_err0 < K.goal_angle)) {  // :195-204 (waypoint_handler.py:167-179); the observation of this step still shows the target
          reward = 100.0f;
          pop_pending = true;
          if (n_left - 1 == 0) { trunc = true; flags |= PF_F_INFO_COMPLETE; }
        }
      }
      if (TASK != PF_TASK_MA_HOVER) go = !(term || trunc);
    }
  }
  PF_STAMP(5);  // (the env step's Aviary steps done)
  const float out_reward = stepping ? reward : 0.0f;
  const bool out_term = stepping && term, out_trunc = stepping && trunc;
  if (stepping) {
    step_count += 1; rng_ctr += 1;  // quadx_base_env.py:299
    // NaN / Inf guard (the reference would carry a NaN on silently, e.g. 0 * inf in
    // quadx.py:490-491 when the largest motor command equals the clipped minimum): any non-finite state
    // word poisons the sum
    const float chk = ((V.p.x + V.p.y) + (V.p.z + V.q.x)) + ((V.q.y + V.q.z) + (V.q.w + V.wvx.y)) + ((V.wvy.y + V.wvz.y) + (V.wvx.x + V.wvy.x)) +
                      ((V.wvz.x + V.t01.x) + (V.t01.y + V.t23.x)) + ((V.t23.y + V.I[0]) + (V.I[1] + V.I[2]));
    if (!(__builtin_fabsf(chk) < INFINITY)) flags |= PF_F_NONFINITE;
  }

  // ---------------------------------------------------------------- SAME_STEP auto-reset
  if (K.autoreset == PF_AUTORESET_SAME_STEP) {
    const bool same = stepping && (term || trunc);
    if (__any(same)) {
      if (B.final_obs != nullptr) {  // terminal observation, before the state is re-initialised
        if (active) write_obs_row();
        flush_tile(B.final_obs + toff * D);
      }
      if (B.final_info != nullptr && same) {  // gymnasium's final_info: the episode's flags / targets left, pre-reset
        B.final_info[2 * (toff + li) + 0] = (flags & ~(PF_F_TERMINATED | PF_F_TRUNCATED | PF_F_CONTACT)) | (term ? PF_F_TERMINATED : 0) |
                                            (trunc ? PF_F_TRUNCATED : 0) | (V.contact_now ? PF_F_CONTACT : 0);
        B.final_info[2 * (toff + li) + 1] = n_left - (pop_pending ? 1 : 0);
      }
      do_resets(same);
    }
  }

  // ---------------------------------------------------------------- outputs
  // observation tile first, persistent state after it: nothing waits on the state stores
  PF_STAMP(6);
  if (active) write_obs_row();
  PF_STAMP(7);  // (observation row computed and written to LDS)
  flush_tile(B.obs + toff * D);
  PF_STAMP(8);  // (observation tile stores issued)
  if (active) {
    if (pop_pending) { pop_target(); pop_pending = false; }
    flags = (flags & ~(PF_F_TERMINATED | PF_F_TRUNCATED | PF_F_CONTACT)) | (term ? PF_F_TERMINATED : 0) |
            (trunc ? PF_F_TRUNCATED : 0) | (V.contact_now ? PF_F_CONTACT : 0);
    if (op == 0) {  // a NEXT_STEP reset call reports (r=0, not done), gymnasium's convention
      B.reward[toff + li] = out_reward;
      B.terminated[toff + li] = out_term ? 1 : 0;
      B.truncated[toff + li] = out_trunc ? 1 : 0;
    }
  }
  if (ROLLOUT && DSTATE) {  // (the resident fp64 copies as a launch boundary would leave them)
    SD.through_hilo();
    if constexpr (DSTATE) C.through_hilo();
  }
  // the next step's motor-noise normals (keyed by the event counter this step left behind)
  if (ROLLOUT && NOISE == PF_NOISE_PHILOX && it + 1 < KS)
    zn = normal8(philox4x32(K.seed_lo, K.seed_hi, (uint32_t)(lane0 + li), rng_ctr, 0u, 0u));
  }  // for it
  if (SPARE && op == 0 && tid == 0) launch_ctr[kCtrStride * blockIdx.x] = call0 + (uint32_t)KS;
  if (active) {  // the persistent state goes back to HBM once per launch
    Sout[0 * N + li] = float4{V.p.x, V.p.y, V.p.z, new_dist};
    Sout[1 * N + li] = float4{V.q.x, V.q.y, V.q.z, V.q.w};
    Sout[2 * N + li] = float4{V.wvx.y, V.wvy.y, V.wvz.y, V.wvx.x};
    Sout[3 * N + li] = float4{V.wvy.x, V.wvz.x, V.t01.x, V.t01.y};
    Sout[4 * N + li] = float4{V.t23.x, V.t23.y, V.I[0], V.I[1]};
    Sout[5 * N + li] = float4{V.I[2], V.E[0], V.E[1], V.E[2]};
    Sout[6 * N + li] = float4{__int_as_float(step_count), __int_as_float(flags), __int_as_float((int)rng_ctr), __int_as_float(n_left)};
    if (MODES && K.mode > 0) C.store(Sout, N, li, REKEY ? (rkw & ~kSpareValid) : 0u);
    if (DSTATE) {  // the remainders of the fp64 state: master - its float32 rounding (groups 0-3 above hold the roundings: V is the view of SD)
      const float e0 = (float)(SD.p[0] - (double)V.p.x), e1 = (float)(SD.p[1] - (double)V.p.y), e2 = (float)(SD.p[2] - (double)V.p.z);
      const float e3 = (float)(SD.q[0] - (double)V.q.x), e4 = (float)(SD.q[1] - (double)V.q.y), e5 = (float)(SD.q[2] - (double)V.q.z), e6 = (float)(SD.q[3] - (double)V.q.w);
      const float e7 = (float)(SD.v[0] - (double)V.wvx.y), e8 = (float)(SD.v[1] - (double)V.wvy.y), e9 = (float)(SD.v[2] - (double)V.wvz.y);
      const float e10 = (float)(SD.w[0] - (double)V.wvx.x), e11 = (float)(SD.w[1] - (double)V.wvy.x), e12 = (float)(SD.w[2] - (double)V.wvz.x);
      Sout[16 * N + li] = float4{e0, e1, e2, e3};
      Sout[17 * N + li] = float4{e4, e5, e6, e7};
      Sout[18 * N + li] = float4{e8, e9, e10, e11};
      Sout[19 * N + li] = float4{e12, (float)(SD.thr[0] - (double)V.t01.x), (float)(SD.thr[1] - (double)V.t01.y), (float)(SD.thr[2] - (double)V.t23.x)};
      // (the PID memories' remainders: the rate PID's behind the state's, the cascade's in the packing of groups 7-11)
      Sout[20 * N + li] = float4{(float)(SD.rI[0] - (double)V.I[0]), (float)(SD.rI[1] - (double)V.I[1]), (float)(SD.rI[2] - (double)V.I[2]), (float)(SD.rE[0] - (double)V.E[0])};
      Sout[21 * N + li] = float4{(float)(SD.rE[1] - (double)V.E[1]), (float)(SD.rE[2] - (double)V.E[2]), (float)(SD.thr[3] - (double)V.t23.y), 0.0f};
      if constexpr (DSTATE) { if (K.mode > 0) C.store_lo(Sout, N, li, 22); }
    }
    if (REKEY && !(MODES && K.mode > 0) && (sp_dirty || ROLLOUT)) {  // group 7: the spare's settled state + the key word; 8-11: its targets
      const Sp x = SPARE ? spare_get() : Sp{};
      Sout[7 * N + li] = float4{x.z, x.vz, x.thr, __int_as_float((int)(SPARE ? rkw : (rkw & ~kSpareValid)))};
      // (pf_rollout writes its LDS rows back whole: what a consumed spare leaves behind is then the same stale words as after k launches)
      if (SPARE && TASK == PF_TASK_WAYPOINTS && (ROLLOUT || (rkw & kSpareValid) != 0u)) {
        Sout[8 * N + li] = x.t[0]; Sout[9 * N + li] = x.t[1]; Sout[10 * N + li] = x.t[2]; Sout[11 * N + li] = x.t[3];
      }
    }
    if (TASK == PF_TASK_MA_HOVER) Sout[15 * N + li] = ma_past;
    if (kYaw) Sout[15 * N + li] = float4{ytg[0], ytg[1], ytg[2], ytg[3]};
    if (TASK == PF_TASK_WAYPOINTS || TASK == PF_TASK_MA_HOVER) {
      Sout[12 * N + li] = float4{tgt[0][0], tgt[0][1], tgt[0][2], tgt[1][0]};
      Sout[13 * N + li] = float4{tgt[1][1], tgt[1][2], tgt[2][0], tgt[2][1]};
      Sout[14 * N + li] = float4{tgt[2][2], tgt[3][0], tgt[3][1], tgt[3][2]};
    }
  }
#ifdef PF_PHASE_TRACE
  PF_STAMP(9);  // (state stores issued)
  __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0): every store acknowledged
  PF_STAMP(10);
  pf_ts[12] = __builtin_amdgcn_s_memrealtime();
  if (ROLL == 0 && tid == 0 && wid == 0 && blockIdx.x < 4096) {
#pragma unroll
    for (int i = 0; i < kPhaseStamps; ++i) g_phase_trace[blockIdx.x * kPhaseStamps + i] = pf_ts[i];
  }
#endif
}

}  // namespace pf
