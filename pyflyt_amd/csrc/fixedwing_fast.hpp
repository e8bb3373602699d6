// fixedwing_fast.hpp -- the specialised kernel for BASELINE.json's config[3]: Fixedwing, flight mode 0
// (fixedwing.py:229-259), Waypoints task (fixedwing_waypoints_env.py), the reference airframe's
// structure (fixedwing.py:80-168, fixedwing.urdf).
//
// The generic env_kernel<Fixedwing> spent 13 700 VALU instructions per wave per env step (8 physics
// ticks; rocprofv3 PMC, profiles/README.md) -- issue-bound like every kernel here, so the lever is the
// instruction count. What this kernel folds away:
//   * structure the reference hard-codes is compile-time: every surface's forward unit is +x, the lift
//     unit is +z (ailerons, h-tail, main wing) or +y (v-tail), so the dot products, the force
//     re-assembly and r x f lose their zero terms; the control-surface mixing [0,0,1,2,1,3] x
//     [+,-,+,-,-,+] (fixedwing.py:143-144) is applied once per env step; the motor sits at the base
//     origin and pushes along +x;
//   * per-surface constants are pre-combined on the host (FwSurf: the flap-shifted zero-lift and stall
//     angles are affine in the deflection) and fetched per surface with scalar loads -- 16 SGPRs live
//     for one surface at a time instead of 130 constants at once (the first generic version spilled
//     1.9 KB/lane, the second staged them through an LDS table);
//   * the post-stall branches (lifting_surfaces.py:409-448) run only in waves where some lane is
//     stalled (wave-uniform test), side-symmetric so one interpolation serves both signs;
//   * the four lift-+z surfaces are evaluated two at a time in packed fp32 (v_pk_fma_f32 / v_pk_mul_f32 /
//     v_pk_add_f32: two IEEE operations per issue slot, and at one wave per SIMD every instruction is
//     one slot): ailerons together, h-tail with the main wing. The element-wise arithmetic and the
//     order the five surfaces accumulate in are those of the one-at-a-time evaluation, so results are
//     bit-identical to it; the transcendental and select steps have no packed form and stay scalar;
//     (the kernel sits exactly at the 256-VGPR limit of two waves per SIMD: packing the body integration as well, or moving
//     the packed chains' constants from SGPR pairs into VGPRs, each saved ~100 instructions per tick on paper and LOST 8 % in
//     the rollout measurement -- the first spill inside the tick loop is a scratch load, and on gfx9 its vmcnt wait also waits
//     for every observation store in flight; profiles/README.md, round 2)
//   * one gyroscopic inertia (I_pa + I_own) instead of two products, rotation scale 2 for the unit
//     quaternion, Euler angles only in the epilogue;
//   * resets copy the context's settled spawn state (the settle throttle command is 0, so the motor
//     noise scales nothing: pyflyt_amd.hip, settle_template_kernel).
#pragma once
#include "../../include/pyflyt_amd.h"
#include "uav_device.hpp"
#include "uav_vehicles.hpp"  // contact_solve_dev
#include "shared_world.hpp"  // pair_stage_dev (the dogfight's shared worlds)
#include <cstddef>
#include <type_traits>

namespace pf {

struct FwSurf {  // 16 floats, one s_load_dwordx16 per surface per tick
  float rx, ry, rz;        // link COM in the base frame (fixedwing.urdf:58,84,110,136,162)
  float cl3d;              // Cl_alpha_3D (lifting_surfaces.py:228-232)
  float a0b, aPb, aNb;     // alpha_0 / alpha_stall_P / alpha_stall_N bases [rad]
  float tau_eta;           // a0 = a0b - tau_eta * defl                       (:386-394)
  float c1;                // aP = aPb + c1 * defl, aN = aNb + c1 * defl, c1 = (flap_to_chord - 1) * tau_eta
  float ipa;               // 1 / (pi * aspect)
  float exp_term, cd0;     // :436, :402
  float defl_lim;          // deflection limit [rad]
  float dt_tau;            // dt / tau of the actuator (:277)
  float hra;               // 0.5 * rho * area
  float chord;
};

struct FwSurf2 {  // two lift-+z surfaces side by side, every field a (first, second) pair: two s_load_dwordx16 per tick
  f2 rx, ry, rz, cl3d, a0b, aPb, aNb, tau_eta, c1, ipa, exp_term, cd0, defl_lim, dt_tau, hra, chord;
};

struct FwBody {  // 32 floats behind the surface rows, two s_load_dwordx16 per tick
  float dt, half_dt, gravity_z, vmax, inv_mass;
  float H[6], iI[6];       // gyroscopic inertia I_pa (+ I_own if use_gyro) and inverse inertia, symmetric 6
  float com[3];
  float bound_radius;
  float m_a, m_noise, fmax, tmax;   // the single motor (fixedwing.py:147-168)
  float slab_xy, slab_bottom;       // the ground slab's half extent and the height of its bottom face
  float pad[5];
};
struct FwTable {
  FwSurf2 pair[2];  // (left aileron, right aileron), (horizontal tail, main wing): fixedwing.py:80-141 ids (0,1), (2,4)
  FwSurf vtail;     // id 3, lift unit +y
  FwBody body;
};
static_assert(sizeof(FwSurf) == 64 && sizeof(FwSurf2) == 128 && sizeof(FwBody) == 128, "constant table rows are whole s_load_dwordx16 units");

struct FwK {  // what stays in kernel-argument SGPRs for the whole kernel: env constants only
  float dome2, goal_reach, min_height, dome09m1, wp_dist_reward;
  int32_t task_sparse, angle_repr, num_targets, max_steps, env_step_ratio, throttle_remap;
  int32_t noise_mode, autoreset;
  uint32_t seed_lo, seed_hi;
  float act_lo[4], act_span[4];  // action box (fixedwing_base_env.py:78-80): low, high - low (pf_rollout's on-device sampling)
};

// The airframe structure FwHot::tick folds away, and its constant table; false -> the generic Fixedwing vehicle is needed.
// (Shared by the Fixedwing-Waypoints kernel below and the dogfight kernel, dogfight.hpp.)
inline bool fw_table_from_params(const pf_params& P, FwTable& T) {
  FwSurf S[5];
  FwBody& Bd = T.body;
  if (P.vehicle != PF_FIXEDWING || P.flight_mode != 0) return false;
  if (P.n_surf != 5 || P.n_motors != 1 || P.ticks_per_control != 2) return false;
  const int ids[6] = {0, 0, 1, 2, 1, 3};
  const float sg[6] = {1.f, -1.f, 1.f, -1.f, -1.f, 1.f};
  for (int k = 0; k < 6; ++k)
    if (P.assist_ids[k] != ids[k] || P.assist_signs[k] != sg[k]) return false;
  for (int i = 0; i < 5; ++i) {
    const pf_surface& s = P.surf[i];
    const float lz = (i == 3) ? 0.f : 1.f, ly = (i == 3) ? 1.f : 0.f;
    if (s.drag[0] != 1.f || s.drag[1] != 0.f || s.drag[2] != 0.f) return false;
    if (s.lift[0] != 0.f || s.lift[1] != ly || s.lift[2] != lz) return false;
    // torque unit = lift x forward (lifting_surfaces.py:236)
    if (s.torque[0] != 0.f || s.torque[1] != lz || s.torque[2] != -ly) return false;
  }
  if (P.motor_r[0][0] != 0.f || P.motor_r[0][1] != 0.f || P.motor_r[0][2] != 0.f) return false;
  if (P.thrust_unit[0][0] != 1.f || P.thrust_unit[0][1] != 0.f || P.thrust_unit[0][2] != 0.f) return false;
  for (int k = 0; k < P.n_boxes; ++k)
    if (P.boxes[k].kind != 0) return false;  // fw_floor_contact tests boxes only
  Bd.dt = P.dt; Bd.half_dt = 0.5f * P.dt; Bd.gravity_z = P.gravity_z; Bd.vmax = P.max_coord_vel; Bd.inv_mass = P.inv_mass;
  for (int k = 0; k < 6; ++k) { Bd.H[k] = P.I_pa[k] + (P.use_gyro_term ? P.I_own[k] : 0.f); Bd.iI[k] = P.I_inv[k]; }
  for (int k = 0; k < 3; ++k) Bd.com[k] = P.has_com_offset ? P.com[k] : 0.f;
  // floor-code gate: one bounding radius + the farthest a contact point or a report reaches
  Bd.bound_radius = P.bound_radius + fmaxf(fmaxf(P.contact_margin, P.contact_break_distance), P.contact_report_distance);
  Bd.m_a = P.motor_dt_over_tau[0]; Bd.m_noise = P.motor_noise[0]; Bd.fmax = P.motor_fmax[0]; Bd.tmax = P.motor_tmax[0];
  Bd.slab_xy = P.plane_half_xy; Bd.slab_bottom = -2.0f * P.plane_half_z;
  for (int k = 0; k < 5; ++k) Bd.pad[k] = 0.f;
  for (int i = 0; i < 5; ++i) {
    const pf_surface& s = P.surf[i];
    FwSurf& o = S[i];
    o.rx = s.r[0]; o.ry = s.r[1]; o.rz = s.r[2];
    o.cl3d = s.Cl_alpha_3D; o.a0b = s.alpha_0_base; o.aPb = s.alpha_stall_P_base; o.aNb = s.alpha_stall_N_base;
    o.tau_eta = s.aero_tau_eta; o.c1 = (s.flap_to_chord - 1.0f) * s.aero_tau_eta;
    o.ipa = s.inv_pi_aspect; o.exp_term = s.exp_term; o.cd0 = s.Cd_0; o.defl_lim = s.deflection_limit_rad;
    o.dt_tau = s.dt_over_tau; o.hra = s.half_rho_area; o.chord = s.chord;
  }
  const int pa[2] = {0, 2}, pb[2] = {1, 4};
  for (int k = 0; k < 2; ++k) {
    const FwSurf &a = S[pa[k]], &b = S[pb[k]];
    FwSurf2& o = T.pair[k];
    o.rx = f2{a.rx, b.rx}; o.ry = f2{a.ry, b.ry}; o.rz = f2{a.rz, b.rz}; o.cl3d = f2{a.cl3d, b.cl3d};
    o.a0b = f2{a.a0b, b.a0b}; o.aPb = f2{a.aPb, b.aPb}; o.aNb = f2{a.aNb, b.aNb}; o.tau_eta = f2{a.tau_eta, b.tau_eta};
    o.c1 = f2{a.c1, b.c1}; o.ipa = f2{a.ipa, b.ipa}; o.exp_term = f2{a.exp_term, b.exp_term}; o.cd0 = f2{a.cd0, b.cd0};
    o.defl_lim = f2{a.defl_lim, b.defl_lim}; o.dt_tau = f2{a.dt_tau, b.dt_tau}; o.hra = f2{a.hra, b.hra}; o.chord = f2{a.chord, b.chord};
  }
  T.vtail = S[3];
  return true;
}

// Fill FwK / FwTable from the ABI struct for the Fixedwing-Waypoints kernel; false -> the configuration needs the generic kernel.
inline bool fwk_from_params(const pf_params& P, FwK& K, FwTable& T) {
  if (P.task != PF_TASK_WAYPOINTS || !fw_table_from_params(P, T)) return false;
  if (P.env_step_ratio < 1 || P.env_step_ratio > 4 || P.num_targets < 1 || P.num_targets > 4) return false;
  if (P.wp_yaw_penalty != 0.f) return false;
  K.dome2 = P.dome * P.dome; K.goal_reach = P.goal_reach_distance; K.min_height = P.min_height;
  K.dome09m1 = P.dome * 0.9f - 1.0f; K.wp_dist_reward = P.wp_dist_reward;
  K.task_sparse = P.sparse_reward; K.angle_repr = P.angle_repr; K.num_targets = P.num_targets; K.max_steps = P.max_steps;
  K.env_step_ratio = P.env_step_ratio; K.throttle_remap = P.throttle_remap;
  K.noise_mode = P.noise_mode; K.autoreset = P.autoreset;
  K.seed_lo = (uint32_t)P.seed; K.seed_hi = (uint32_t)(P.seed >> 32);
  for (int k = 0; k < 4; ++k) { K.act_lo[k] = P.action_low[k]; K.act_span[k] = P.action_high[k] - P.action_low[k]; }
  return true;
}

// 15-axis box tests of the airframe's collision boxes against the ground box; out of line, runs only
// in waves that have a lane within one bounding radius of the floor.
// (the parameter block through the scalar cache: inside an out-of-line function the plain pointer lives in VGPRs and its fields
//  were flat loads at full memory latency, six boxes one after the other, in every tick of every wave that has a low flyer --
//  the tail of the kernel's time in a population that is crashing; see uav_vehicles.hpp: uniform_params)
// persisted: the body held contact points after the previous tick -- reported up to the breaking distance (pf_params.contact_break_distance),
// a fresh pair from contact_report_distance on: the slab enlarged by that gap.
__device__ __noinline__ PF_RARE_TEXT bool fw_floor_contact(float px, float py, float pz, m3 R, const pf_params* Pg, bool persisted) {
  const pf_params_kptr P = uniform_params(Pg);
  const float rd_kept = P->contact_break_distance, rd_fresh = P->contact_report_distance;
  const float rd = persisted ? rd_kept : rd_fresh;
  const float hb[3] = {P->plane_half_xy + rd, P->plane_half_xy + rd, P->plane_half_z + rd};
  const v3 cb{0.0f, 0.0f, -P->plane_half_z};
  bool hit = false;
  const int nb = P->n_boxes;
  for (int k = 0; k < nb; ++k) {
    const float bh[3] = {P->boxes[k].h[0], P->boxes[k].h[1], P->boxes[k].h[2]};
    v3 c = v3{px, py, pz} + mul(R, v3{P->boxes[k].c[0], P->boxes[k].c[1], P->boxes[k].c[2]});
    hit |= box_overlaps_aabb(c, R, bh, cb, hb);
  }
  return hit;
}

typedef const FwTable __attribute__((address_space(4))) * fw_tab_cptr;
typedef const FwSurf __attribute__((address_space(4))) * fw_surf_cptr;
typedef const FwSurf2 __attribute__((address_space(4))) * fw_surf2_cptr;
typedef const FwBody __attribute__((address_space(4))) * fw_body_cptr;
PF_DEV FwBody fw_load_body(fw_body_cptr p) {
  FwBody b;
  b.dt = p->dt; b.half_dt = p->half_dt; b.gravity_z = p->gravity_z; b.vmax = p->vmax; b.inv_mass = p->inv_mass;
#pragma unroll
  for (int k = 0; k < 6; ++k) { b.H[k] = p->H[k]; b.iI[k] = p->iI[k]; }
#pragma unroll
  for (int k = 0; k < 3; ++k) b.com[k] = p->com[k];
  b.bound_radius = p->bound_radius; b.m_a = p->m_a; b.m_noise = p->m_noise; b.fmax = p->fmax; b.tmax = p->tmax;
  b.slab_xy = p->slab_xy; b.slab_bottom = p->slab_bottom;
  return b;
}
PF_DEV FwSurf fw_load_surf(fw_surf_cptr p) {  // uniform address in the constant address space -> s_load_dwordx16
  FwSurf S;
  S.rx = p->rx; S.ry = p->ry; S.rz = p->rz; S.cl3d = p->cl3d; S.a0b = p->a0b; S.aPb = p->aPb; S.aNb = p->aNb;
  S.tau_eta = p->tau_eta; S.c1 = p->c1; S.ipa = p->ipa; S.exp_term = p->exp_term; S.cd0 = p->cd0;
  S.defl_lim = p->defl_lim; S.dt_tau = p->dt_tau; S.hra = p->hra; S.chord = p->chord;
  return S;
}

PF_DEV FwSurf2 fw_load_surf2(fw_surf2_cptr p) {  // two s_load_dwordx16, every field an aligned SGPR pair
  FwSurf2 S;
  S.rx = p->rx; S.ry = p->ry; S.rz = p->rz; S.cl3d = p->cl3d; S.a0b = p->a0b; S.aPb = p->aPb; S.aNb = p->aNb;
  S.tau_eta = p->tau_eta; S.c1 = p->c1; S.ipa = p->ipa; S.exp_term = p->exp_term; S.cd0 = p->cd0;
  S.defl_lim = p->defl_lim; S.dt_tau = p->dt_tau; S.hra = p->hra; S.chord = p->chord;
  return S;
}
// The whole constant table in VECTOR registers, for the one-wave-per-SIMD instantiation (512 registers): loaded once per launch
// instead of five scalar loads -- each with its own wait, and nothing to hide it behind at one wave per SIMD -- in every tick, and a
// uniform value in a vector register costs the instruction that reads it nothing (no constant-bus limit, no SGPR -> VGPR copy in
// front of a packed operand; quadx_fast.hpp: QuadKV).
struct FwTableV {
  FwSurf2 pair[2];   // the two lift-+z surface pairs
  FwSurf vtail;
  FwBody body;
};
// (Measured in round 5 and dropped -- profiles/r06/experiments/ab_switches.patch has the switches: the v-tail and body rows by scalar
//  loads in the tick, 22.1 us against 21.5, sixteen scalars spilled around the pairs; the same two rows read from LDS in the tick,
//  22.2 against 21.4, the reads' latency shows; the whole table by 28 broadcast global loads, 21.1 against 20.7 through LDS. Round 6,
//  profiles/r06/experiments/fixedwing_x3.patch: the v-tail as a third stream next to the two pairs -- 23.0 against 20.7, the third
//  stream's temporaries push another 45 values per tick through AGPR copies; and the table shrunk by what a mirror-symmetric airframe
//  makes redundant (aileron constants as splats, zero products of inertia): 21.15 against 20.65 per step, 15.63 against 16.0 per
//  rollout step -- inside what the register allocator moves from one neutral edit to the next.)
typedef float fw_f4v __attribute__((ext_vector_type(4)));
typedef const fw_f4v __attribute__((address_space(3))) * fw_lds_f4ptr;
// Into the vector registers through LDS: every lane fetches two WORDS of the table with the state groups (two coalesced requests
// instead of 28 broadcast ones, which took the wave 0.7 us to issue -- the int group, first in line, sat ready the while), the words
// go to LDS once they are here, behind the step's Philox call, and 28 broadcast ds_read_b128 bring them to every lane.
// (Measured, r05: the two word loads in FRONT of the state groups and the LDS stage in front of the Philox call -- so that the LDS reads
//  complete under it -- is slower, 20.7 -> 21.3 us: every wave of the launch asks for the same four lines at the same moment, the
//  requests queue at one L2 channel, and behind them, in order, waits the wave's own state. Where they are, that queue drains under
//  the state groups' latency and the Philox call. profiles/tools/r05/g22.sh)
PF_DEV FwTableV fw_table_from_lds(lds_fptr base) {
  const fw_lds_f4ptr q = (fw_lds_f4ptr)base;
  constexpr int kN = (int)(sizeof(FwTable) / 16);
  fw_f4v r[kN];
#pragma unroll
  for (int i = 0; i < kN; ++i) r[i] = q[i];
  FwTableV V;
  __builtin_memcpy(&V.pair[0], r, kN * 16);
  return V;
}
static_assert(sizeof(FwTable) == 112 * 4, "the table's 112 words: lanes 0-63 fetch the first 64, lanes 0-47 the rest");
static_assert(offsetof(FwTableV, vtail) == offsetof(FwTable, vtail) && offsetof(FwTableV, body) == offsetof(FwTable, body), "FwTableV starts with FwTable");
struct FwPairOut { f2 fp, fn, ty; };  // per surface: force along +x, along the lift unit (+z), and the r x f + moment part of tau.y

struct FwHot {
  v3 p; quat q; v3 v, w;
  float act[5];
  float thr;
  float cmd[6];
  m3 R; v3 wb, vb;
  bool contact_now, contact_step;
  lds_fptr cws;  // the wave's LDS regions for the contact solver (aliased onto the observation tile, idle during the ticks)
  int cws_floats = 64 * 35;
  // shared worlds (dogfight.hpp): this tick's drone-drone verdict, ORed into the contact report; the world-global bit is only
  // exchanged (the rotational-drag gate it feeds exists on the quadrotor, quadx.py:509)
  bool peer_contact = false, world_contact = false;
  bool world_touch = false;  // world_exchange's verdict: some pair of this world is within reach of the contact response between aircraft
  bool woken = false;        // world_exchange: a wreck at rest that a moving aircraft has come within reach of
  // shared worlds, the pair stage (shared_world.hpp: pair_stage_dev): the wave's pose / velocity exchange arrays, the LDS for its
  // contact records (the observation tile, as a generic pointer), this lane, aircraft per world
  const float* wpose_ = nullptr;
  float* wvel_ = nullptr;
  float* prec_ = nullptr;
  int wtid = 0, wA = 1;

  PF_DEV void derive() {  // unit quaternion (quat_integrate / the settled template): scale 2
    const float xs = q.x + q.x, ys = q.y + q.y, zs = q.z + q.z;
    const float xy = q.x * ys, xz = q.x * zs, yz = q.y * zs;
    const float dx = fmaf(-q.x, xs, 1.0f), dy = fmaf(-q.y, ys, 1.0f);  // 1 - xx, 1 - yy
    R = m3{fmaf(-q.z, zs, dy), fmaf(-q.w, zs, xy), fmaf(q.w, ys, xz), fmaf(q.w, zs, xy), fmaf(-q.z, zs, dx), fmaf(-q.w, xs, yz),
           fmaf(-q.w, ys, xz), fmaf(q.w, xs, yz), fmaf(-q.y, ys, dx)};
    wb = mulT(R, w);
    vb = mulT(R, v);
  }
  // One lifting surface (lifting_surfaces.py:73-110 local velocity, :266-498 coefficients and forces).
  // LIFT_Y: the vertical tail (lift unit +y, torque unit -z); otherwise lift +z, torque +y.
  // `a`: this surface's actuation after the first-order lag. Accumulates into F, tau (base frame,
  // torque about the base origin).
  template <bool LIFT_Y>
  PF_DEV void surface(const FwSurf S, const float a, v3& F, v3& tau) const {
    const float vx = fmaf(wb.y, S.rz, fmaf(-wb.z, S.ry, vb.x));
    const float vy = fmaf(wb.z, S.rx, fmaf(-wb.x, S.rz, vb.y));
    const float vz = fmaf(wb.x, S.ry, fmaf(-wb.y, S.rx, vb.z));
    const float V2 = fmaf(vx, vx, fmaf(vy, vy, vz * vz));
    const float la = LIFT_Y ? vy : vz, fa = vx;
    const float h2 = fmaf(la, la, fa * fa);
    const float ih = frsq(h2);
    const bool still = !(h2 > 0.0f);
    const float ca = still ? 1.0f : fa * ih, sa = still ? 0.0f : -la * ih;
    const float alpha = fast_atan2(-la, fa);  // :342-345
    const float defl = a * S.defl_lim;        // :386
    const float a0 = fmaf(-S.tau_eta, defl, S.a0b);
    const float aP = fmaf(S.c1, defl, S.aPb), aN = fmaf(S.c1, defl, S.aNb);
    const bool linear = (aN < alpha) && (alpha < aP);
    const float Cl_lin = S.cl3d * (alpha - a0);
    float ai = Cl_lin * S.ipa;  // :397-399
    const bool any_stall = __any(!linear);
    if (any_stall) {  // :409-425, two-point np.interp between the stall angle and +-pi/2
      const bool pos = alpha > 0.0f;
      const float as = pos ? aP : aN;
      const float ai_stall = S.cl3d * (as - a0) * S.ipa;
      const float edge = pos ? 0.5f * kPi : -0.5f * kPi;
      // pos: alpha<=aP -> ai_stall, alpha>=pi/2 -> 0, else ai_stall*(1-(alpha-aP)/(pi/2-aP))
      // neg: alpha<=-pi/2 -> 0, alpha>=aN -> ai_stall, else ai_stall*(alpha+pi/2)/(aN+pi/2)
      const float tt = (edge - alpha) * frcp(edge - as);
      const float ais = ai_stall * med3(tt, 0.0f, 1.0f);
      ai = linear ? ai : ais;
    }
    const float x = a0 + ai;
    const float ae = alpha - x;
    float sx, cx;
    sincos_small(x, sx, cx);
    const float se = fmaf(sa, cx, -(ca * sx)), ce = fmaf(ca, cx, sa * sx);
    // :397-406
    float CT = S.cd0 * ce;
    float CN = fmaf(CT, se, Cl_lin) * frcp(ce);
    float Cl = Cl_lin;
    float Cd = fmaf(CN, se, CT * ce);
    float CM = -CN * fmaf(0.175f * (2.0f / kPi), ae, 0.075f);  // 0.25 - 0.175 (1 - 2 ae / pi)
    if (any_stall) {  // :427-448
      const float Cd90 = fmaf(-4.26e-2f, defl * defl, fmaf(2.1e-1f, defl, 1.98f));
      const float CNs = Cd90 * se * (frcp(fmaf(0.44f, __builtin_fabsf(se), 0.56f)) - S.exp_term);
      const float CTs = 0.5f * S.cd0 * ce;
      const float Cls = fmaf(CNs, ce, -(CTs * se));
      const float Cds = fmaf(CNs, se, CTs * ce);
      const float CMs = -CNs * fmaf(0.175f * (2.0f / kPi), __builtin_fabsf(ae), 0.075f);
      Cl = linear ? Cl : Cls; Cd = linear ? Cd : Cds; CM = linear ? CM : CMs;
    }
    // :485-498
    const float QA = S.hra * V2;
    const float L = Cl * QA, D = Cd * QA;
    const float fn = fmaf(L, ca, D * sa), fp = fmaf(L, sa, -(D * ca));  // along the lift unit / along +x
    const float tm = QA * CM * S.chord;
    F.x += fp;
    if (LIFT_Y) {
      F.y += fn;
      tau.x = fmaf(-S.rz, fn, tau.x);
      tau.y = fmaf(S.rz, fp, tau.y);
      tau.z += fmaf(S.rx, fn, fmaf(-S.ry, fp, -tm));
    } else {
      F.z += fn;
      tau.x = fmaf(S.ry, fn, tau.x);
      tau.y += fmaf(S.rz, fp, fmaf(-S.rx, fn, tm));
      tau.z = fmaf(-S.ry, fp, tau.z);
    }
  }
  // Two lift-+z surfaces at once: surface<false> element by element (same operations on each element),
  // the multiply-add chains in packed fp32. The statement ORDER is the schedule (the build runs with the
  // machine scheduler off): a packed result read by the very next VALU instruction costs a wait state
  // (s_nop) and a compare needs two instructions before the select that reads its mask (gfx940 hazards),
  // so independent work is placed into every such gap by hand -- the Horner chains of atan2 and of
  // sin / cos are interleaved with the actuator lag, the stall angles and the selects.
  PF_DEV FwPairOut surface_pair(const FwSurf2 S, f2& a, const f2 cmd2) const {
    const f2 wbx = sp2(wb.x), wby = sp2(wb.y), wbz = sp2(wb.z);
    const f2 da = cmd2 - a;
    const f2 t1 = fma2(-wbz, S.ry, sp2(vb.x));
    const f2 t2 = fma2(-wbx, S.rz, sp2(vb.y));
    const f2 t3 = fma2(-wby, S.rx, sp2(vb.z));
    a = fma2(S.dt_tau, da, a);  // lifting_surfaces.py:277
    const f2 vx = fma2(wby, S.rz, t1);
    const f2 vy = fma2(wbz, S.rx, t2);
    const f2 vz = fma2(wbx, S.ry, t3);
    const f2 la = vz, fa = vx;
    const f2 defl = a * S.defl_lim;
    const f2 fa2 = fa * fa;
    const f2 vz2 = vz * vz;
    const f2 h2 = fma2(la, la, fa2);
    const f2 V2t = fma2(vy, vy, vz2);
    const f2 a0 = fma2(-S.tau_eta, defl, S.a0b);
    const f2 V2 = fma2(vx, vx, V2t);
    // fast_atan2_pair(-la, fa), inlined so its chain can be interleaved
    const f2 y = -la;
    const float ax0 = __builtin_fabsf(fa.x), ay0 = __builtin_fabsf(y.x), ax1 = __builtin_fabsf(fa.y), ay1 = __builtin_fabsf(y.y);
    const float mx0 = __builtin_fmaxf(ax0, ay0), mx1 = __builtin_fmaxf(ax1, ay1);
    const float ih0 = frsq(h2.x), ih1 = frsq(h2.y);
    const float rc0 = frcp(mx0), rc1 = frcp(mx1);
    const float mn0 = __builtin_fminf(ax0, ay0), mn1 = __builtin_fminf(ax1, ay1);
    const bool zero0 = mx0 == 0.0f, zero1 = mx1 == 0.0f;
    const bool still0 = !(h2.x > 0.0f), still1 = !(h2.y > 0.0f);
    const bool steep0 = ay0 > ax0, steep1 = ay1 > ax1;
    const bool back0 = fa.x < 0.0f, back1 = fa.y < 0.0f;
    const f2 ih = f2{ih0, ih1};
    f2 t = f2{mn0, mn1} * f2{rc0, rc1};
    const f2 cu = fa * ih, su = y * ih;
    const f2 aP = fma2(S.c1, defl, S.aPb);
    t = f2{zero0 ? 0.0f : t.x, zero1 ? 0.0f : t.y};
    const f2 aN = fma2(S.c1, defl, S.aNb);
    const f2 s = t * t;
    const f2 QA = S.hra * V2;
    f2 p = fma2(s, sp2(0.0029035410843789577f), sp2(-0.016282962635159492f));
    const float ca0 = still0 ? 1.0f : cu.x;
    p = fma2(s, p, sp2(0.04303929582238197f));
    const float ca1 = still1 ? 1.0f : cu.y;
    p = fma2(s, p, sp2(-0.07533670216798782f));
    const float sa0 = still0 ? 0.0f : su.x;
    p = fma2(s, p, sp2(0.10654674470424652f));
    const float sa1 = still1 ? 0.0f : su.y;
    p = fma2(s, p, sp2(-0.14207133650779724f));
    const f2 ca = f2{ca0, ca1};
    const f2 sa = f2{sa0, sa1};
    const f2 dd = defl * defl;
    p = fma2(s, p, sp2(0.19993053376674652f));
    const f2 c9a = fma2(sp2(2.1e-1f), defl, sp2(1.98f));
    p = fma2(s, p, sp2(-0.3333309292793274f));
    const f2 Cd90 = fma2(sp2(-4.26e-2f), dd, c9a);  // (used by the post-stall branch only)
    p = fma2(s, p, sp2(1.0f));
    const f2 hcd = sp2(0.5f) * S.cd0;
    f2 r = p * t;
    const f2 rq = sp2(0.5f * kPi) - r;
    r = f2{steep0 ? rq.x : r.x, steep1 ? rq.y : r.y};
    const f2 rh = sp2(kPi) - r;
    r = f2{back0 ? rh.x : r.x, back1 ? rh.y : r.y};
    const f2 alpha = f2{__builtin_copysignf(r.x, y.x), __builtin_copysignf(r.y, y.y)};
    const bool lin0 = (aN.x < alpha.x) && (alpha.x < aP.x), lin1 = (aN.y < alpha.y) && (alpha.y < aP.y);
    const f2 am = alpha - a0;
    const bool any_stall = __any(!(lin0 && lin1));
    const f2 Cl_lin = S.cl3d * am;
    f2 ai = Cl_lin * S.ipa;
    if (any_stall) {  // :409-425 (see surface<>)
      const bool pos0 = alpha.x > 0.0f, pos1 = alpha.y > 0.0f;
      const f2 as = f2{pos0 ? aP.x : aN.x, pos1 ? aP.y : aN.y};
      const f2 edge = f2{pos0 ? 0.5f * kPi : -0.5f * kPi, pos1 ? 0.5f * kPi : -0.5f * kPi};
      const f2 asm0 = as - a0;
      const f2 den = edge - as;
      const f2 num = edge - alpha;
      const f2 aist = S.cl3d * asm0;
      const f2 rden = f2{frcp(den.x), frcp(den.y)};
      const f2 ai_stall = aist * S.ipa;
      const f2 tt = num * rden;
      const f2 ais = ai_stall * f2{med3(tt.x, 0.0f, 1.0f), med3(tt.y, 0.0f, 1.0f)};
      ai = f2{lin0 ? ai.x : ais.x, lin1 ? ai.y : ais.y};
    }
    const f2 x = a0 + ai;
    // sincos_small(x): the two Horner chains side by side
    const f2 tq = x * x;
    const f2 ae = alpha - x;
    f2 ps = fma2(tq, sp2(-2.5052108e-8f), sp2(2.7557319e-6f));
    f2 pc = fma2(tq, sp2(2.0876757e-9f), sp2(-2.7557319e-7f));
    ps = fma2(tq, ps, sp2(-1.9841270e-4f));
    pc = fma2(tq, pc, sp2(2.4801587e-5f));
    ps = fma2(tq, ps, sp2(8.3333333e-3f));
    pc = fma2(tq, pc, sp2(-1.3888889e-3f));
    ps = fma2(tq, ps, sp2(-1.6666667e-1f));
    pc = fma2(tq, pc, sp2(4.1666667e-2f));
    ps = fma2(tq, ps, sp2(1.0f));
    pc = fma2(tq, pc, sp2(-0.5f));
    const f2 cmk = fma2(sp2(0.175f * (2.0f / kPi)), ae, sp2(0.075f));
    const f2 cx = fma2(tq, pc, sp2(1.0f));
    const f2 sx = x * ps;
    const f2 cas = ca * sx;
    const f2 sas = sa * sx;
    const f2 se = fma2(sa, cx, -cas), ce = fma2(ca, cx, sas);
    // :397-406
    const float rce0 = frcp(ce.x), rce1 = frcp(ce.y);
    const f2 CT = S.cd0 * ce;
    const f2 CTc = CT * ce;
    const f2 cnn = fma2(CT, se, Cl_lin);
    const f2 CN = cnn * f2{rce0, rce1};
    f2 Cl = Cl_lin;
    f2 Cd = fma2(CN, se, CTc);
    f2 CM = -CN * cmk;
    if (any_stall) {  // :427-448
      const f2 dn = fma2(sp2(0.44f), f2{__builtin_fabsf(se.x), __builtin_fabsf(se.y)}, sp2(0.56f));
      const f2 c9s = Cd90 * se;
      const f2 CTs = hcd * ce;
      const f2 cms = fma2(sp2(0.175f * (2.0f / kPi)), f2{__builtin_fabsf(ae.x), __builtin_fabsf(ae.y)}, sp2(0.075f));
      const f2 rdn = f2{frcp(dn.x), frcp(dn.y)};
      const f2 cts = CTs * se;
      const f2 ctc = CTs * ce;
      const f2 CNs = c9s * (rdn - S.exp_term);
      const f2 Cls = fma2(CNs, ce, -cts);
      const f2 Cds = fma2(CNs, se, ctc);
      const f2 CMs = -CNs * cms;
      Cl = f2{lin0 ? Cl.x : Cls.x, lin1 ? Cl.y : Cls.y};
      Cd = f2{lin0 ? Cd.x : Cds.x, lin1 ? Cd.y : Cds.y};
      CM = f2{lin0 ? CM.x : CMs.x, lin1 ? CM.y : CMs.y};
    }
    // :485-498
    const f2 L = Cl * QA, D = Cd * QA;
    const f2 qc = QA * CM;
    const f2 dsa = D * sa, dca = D * ca;
    const f2 tm = qc * S.chord;
    FwPairOut o;
    o.fn = fma2(L, ca, dsa);
    o.fp = fma2(L, sa, -dca);
    const f2 ty0 = fma2(-S.rx, o.fn, tm);
    o.ty = fma2(S.rz, o.fp, ty0);
    return o;
  }
  // Both surface pairs at once: surface_pair statement by statement on the aileron pair (k = 0) and on the (h-tail, main wing) pair
  // (k = 1) alternately. The two evaluations are independent, so every instruction has an independent neighbour: the
  // packed-result / compare-select wait states of the one-pair version are filled by the other pair instead of by hand or by s_nop
  // (that is the whole gain: a lone wave issues one instruction per 4-5 clocks dependent or not, profiles/r06/lone_wave_issue.txt --
  // round 5's "7.5 clocks per dependent instruction" was an fma plus the s_nop the compiler puts between inline-asm statements; the two
  // pairs one after the other on the same table are 0.63 us per step slower, profiles/r06/ab_fixedwing_x1_vs_x2_same_box.txt). Per element
  // the same operations in the same order as surface_pair: bit-identical results. Needs both pairs' constants at once (64 values):
  // the one-wave-per-SIMD instantiation holds them in vector registers (FwTableV).
#define PF_X2(...) { constexpr int k = 0; __VA_ARGS__ } { constexpr int k = 1; __VA_ARGS__ }
  PF_DEV void surface_pair_x2(const FwSurf2 (&S)[2], f2 (&a)[2], const f2 (&cmd2)[2], FwPairOut (&o)[2]) const {
    const f2 wbx = sp2(wb.x), wby = sp2(wb.y), wbz = sp2(wb.z);
    f2 da[2], t1[2], t2[2], t3[2], vx[2], vy[2], vz[2], defl[2], fa2[2], vz2[2], h2[2], V2t[2], a0[2], V2[2], y[2];
    PF_X2(da[k] = cmd2[k] - a[k];)
    PF_X2(t1[k] = fma2(-wbz, S[k].ry, sp2(vb.x));)
    PF_X2(t2[k] = fma2(-wbx, S[k].rz, sp2(vb.y));)
    PF_X2(t3[k] = fma2(-wby, S[k].rx, sp2(vb.z));)
    PF_X2(a[k] = fma2(S[k].dt_tau, da[k], a[k]);)  // lifting_surfaces.py:277
    PF_X2(vx[k] = fma2(wby, S[k].rz, t1[k]);)
    PF_X2(vy[k] = fma2(wbz, S[k].rx, t2[k]);)
    PF_X2(vz[k] = fma2(wbx, S[k].ry, t3[k]);)
    PF_X2(defl[k] = a[k] * S[k].defl_lim;)
    PF_X2(fa2[k] = vx[k] * vx[k];)
    PF_X2(vz2[k] = vz[k] * vz[k];)
    PF_X2(h2[k] = fma2(vz[k], vz[k], fa2[k]);)
    PF_X2(V2t[k] = fma2(vy[k], vy[k], vz2[k]);)
    PF_X2(a0[k] = fma2(-S[k].tau_eta, defl[k], S[k].a0b);)
    PF_X2(V2[k] = fma2(vx[k], vx[k], V2t[k]);)
    PF_X2(y[k] = -vz[k];)
    // fast_atan2_pair(-la, fa) with la = vz, fa = vx
    float ax0[2], ay0[2], ax1[2], ay1[2], mx0[2], mx1[2], ih0[2], ih1[2], rc0[2], rc1[2], mn0[2], mn1[2];
    bool zero0[2], zero1[2], still0[2], still1[2], steep0[2], steep1[2], back0[2], back1[2];
    PF_X2(ax0[k] = __builtin_fabsf(vx[k].x); ay0[k] = __builtin_fabsf(y[k].x); ax1[k] = __builtin_fabsf(vx[k].y); ay1[k] = __builtin_fabsf(y[k].y);)
    PF_X2(mx0[k] = __builtin_fmaxf(ax0[k], ay0[k]); mx1[k] = __builtin_fmaxf(ax1[k], ay1[k]);)
    PF_X2(ih0[k] = frsq(h2[k].x); ih1[k] = frsq(h2[k].y);)
    PF_X2(rc0[k] = frcp(mx0[k]); rc1[k] = frcp(mx1[k]);)
    PF_X2(mn0[k] = __builtin_fminf(ax0[k], ay0[k]); mn1[k] = __builtin_fminf(ax1[k], ay1[k]);)
    PF_X2(zero0[k] = mx0[k] == 0.0f; zero1[k] = mx1[k] == 0.0f;)
    PF_X2(still0[k] = !(h2[k].x > 0.0f); still1[k] = !(h2[k].y > 0.0f);)
    PF_X2(steep0[k] = ay0[k] > ax0[k]; steep1[k] = ay1[k] > ax1[k];)
    PF_X2(back0[k] = vx[k].x < 0.0f; back1[k] = vx[k].y < 0.0f;)
    f2 ih[2], t[2], cu[2], su[2], aP[2], aN[2], ss[2], QA[2], p[2], ca[2], sa[2], dd[2], c9a[2], Cd90[2], hcd[2], r[2], rq[2], rh[2], alpha[2], am[2], Cl_lin[2], ai[2];
    PF_X2(ih[k] = f2{ih0[k], ih1[k]};)
    PF_X2(t[k] = f2{mn0[k], mn1[k]} * f2{rc0[k], rc1[k]};)
    PF_X2(cu[k] = vx[k] * ih[k];)
    PF_X2(su[k] = y[k] * ih[k];)
    PF_X2(aP[k] = fma2(S[k].c1, defl[k], S[k].aPb);)
    PF_X2(t[k] = f2{zero0[k] ? 0.0f : t[k].x, zero1[k] ? 0.0f : t[k].y};)
    PF_X2(aN[k] = fma2(S[k].c1, defl[k], S[k].aNb);)
    PF_X2(ss[k] = t[k] * t[k];)
    PF_X2(QA[k] = S[k].hra * V2[k];)
    PF_X2(p[k] = fma2(ss[k], sp2(0.0029035410843789577f), sp2(-0.016282962635159492f));)
    PF_X2(p[k] = fma2(ss[k], p[k], sp2(0.04303929582238197f));)
    PF_X2(p[k] = fma2(ss[k], p[k], sp2(-0.07533670216798782f));)
    PF_X2(ca[k] = f2{still0[k] ? 1.0f : cu[k].x, still1[k] ? 1.0f : cu[k].y};)
    PF_X2(p[k] = fma2(ss[k], p[k], sp2(0.10654674470424652f));)
    PF_X2(sa[k] = f2{still0[k] ? 0.0f : su[k].x, still1[k] ? 0.0f : su[k].y};)
    PF_X2(p[k] = fma2(ss[k], p[k], sp2(-0.14207133650779724f));)
    PF_X2(dd[k] = defl[k] * defl[k];)
    PF_X2(p[k] = fma2(ss[k], p[k], sp2(0.19993053376674652f));)
    PF_X2(c9a[k] = fma2(sp2(2.1e-1f), defl[k], sp2(1.98f));)
    PF_X2(p[k] = fma2(ss[k], p[k], sp2(-0.3333309292793274f));)
    PF_X2(Cd90[k] = fma2(sp2(-4.26e-2f), dd[k], c9a[k]);)  // (used by the post-stall branch only)
    PF_X2(p[k] = fma2(ss[k], p[k], sp2(1.0f));)
    PF_X2(hcd[k] = sp2(0.5f) * S[k].cd0;)
    PF_X2(r[k] = p[k] * t[k];)
    PF_X2(rq[k] = sp2(0.5f * kPi) - r[k];)
    PF_X2(r[k] = f2{steep0[k] ? rq[k].x : r[k].x, steep1[k] ? rq[k].y : r[k].y};)
    PF_X2(rh[k] = sp2(kPi) - r[k];)
    PF_X2(r[k] = f2{back0[k] ? rh[k].x : r[k].x, back1[k] ? rh[k].y : r[k].y};)
    PF_X2(alpha[k] = f2{__builtin_copysignf(r[k].x, y[k].x), __builtin_copysignf(r[k].y, y[k].y)};)
    bool lin0[2], lin1[2];
    PF_X2(lin0[k] = (aN[k].x < alpha[k].x) && (alpha[k].x < aP[k].x); lin1[k] = (aN[k].y < alpha[k].y) && (alpha[k].y < aP[k].y);)
    PF_X2(am[k] = alpha[k] - a0[k];)
    // (one wave-uniform test for both pairs: the post-stall code only SELECTS per element, so running it for a pair none of whose
    //  elements is stalled changes nothing)
    const bool any_stall = __any(!(lin0[0] && lin1[0] && lin0[1] && lin1[1]));
    PF_X2(Cl_lin[k] = S[k].cl3d * am[k];)
    PF_X2(ai[k] = Cl_lin[k] * S[k].ipa;)
    if (any_stall) {  // :409-425 (see surface<>)
      bool pos0[2], pos1[2];
      f2 as[2], edge[2], asm0[2], den[2], num[2], aist[2], rden[2], ai_stall[2], tt[2], ais[2];
      PF_X2(pos0[k] = alpha[k].x > 0.0f; pos1[k] = alpha[k].y > 0.0f;)
      PF_X2(as[k] = f2{pos0[k] ? aP[k].x : aN[k].x, pos1[k] ? aP[k].y : aN[k].y};)
      PF_X2(edge[k] = f2{pos0[k] ? 0.5f * kPi : -0.5f * kPi, pos1[k] ? 0.5f * kPi : -0.5f * kPi};)
      PF_X2(asm0[k] = as[k] - a0[k];)
      PF_X2(den[k] = edge[k] - as[k];)
      PF_X2(num[k] = edge[k] - alpha[k];)
      PF_X2(aist[k] = S[k].cl3d * asm0[k];)
      PF_X2(rden[k] = f2{frcp(den[k].x), frcp(den[k].y)};)
      PF_X2(ai_stall[k] = aist[k] * S[k].ipa;)
      PF_X2(tt[k] = num[k] * rden[k];)
      PF_X2(ais[k] = ai_stall[k] * f2{med3(tt[k].x, 0.0f, 1.0f), med3(tt[k].y, 0.0f, 1.0f)};)
      PF_X2(ai[k] = f2{lin0[k] ? ai[k].x : ais[k].x, lin1[k] ? ai[k].y : ais[k].y};)
    }
    f2 x[2], tq[2], ae[2], ps[2], pc[2], cmk[2], cx[2], sx[2], cas[2], sas[2], se[2], ce[2], CT[2], CTc[2], cnn[2], CN[2], Cl[2], Cd[2], CM[2];
    float rce0[2], rce1[2];
    PF_X2(x[k] = a0[k] + ai[k];)
    // sincos_small(x): the two Horner chains side by side
    PF_X2(tq[k] = x[k] * x[k];)
    PF_X2(ae[k] = alpha[k] - x[k];)
    PF_X2(ps[k] = fma2(tq[k], sp2(-2.5052108e-8f), sp2(2.7557319e-6f));)
    PF_X2(pc[k] = fma2(tq[k], sp2(2.0876757e-9f), sp2(-2.7557319e-7f));)
    PF_X2(ps[k] = fma2(tq[k], ps[k], sp2(-1.9841270e-4f));)
    PF_X2(pc[k] = fma2(tq[k], pc[k], sp2(2.4801587e-5f));)
    PF_X2(ps[k] = fma2(tq[k], ps[k], sp2(8.3333333e-3f));)
    PF_X2(pc[k] = fma2(tq[k], pc[k], sp2(-1.3888889e-3f));)
    PF_X2(ps[k] = fma2(tq[k], ps[k], sp2(-1.6666667e-1f));)
    PF_X2(pc[k] = fma2(tq[k], pc[k], sp2(4.1666667e-2f));)
    PF_X2(ps[k] = fma2(tq[k], ps[k], sp2(1.0f));)
    PF_X2(pc[k] = fma2(tq[k], pc[k], sp2(-0.5f));)
    PF_X2(cmk[k] = fma2(sp2(0.175f * (2.0f / kPi)), ae[k], sp2(0.075f));)
    PF_X2(cx[k] = fma2(tq[k], pc[k], sp2(1.0f));)
    PF_X2(sx[k] = x[k] * ps[k];)
    PF_X2(cas[k] = ca[k] * sx[k];)
    PF_X2(sas[k] = sa[k] * sx[k];)
    PF_X2(se[k] = fma2(sa[k], cx[k], -cas[k]); ce[k] = fma2(ca[k], cx[k], sas[k]);)
    // :397-406
    PF_X2(rce0[k] = frcp(ce[k].x); rce1[k] = frcp(ce[k].y);)
    PF_X2(CT[k] = S[k].cd0 * ce[k];)
    PF_X2(CTc[k] = CT[k] * ce[k];)
    PF_X2(cnn[k] = fma2(CT[k], se[k], Cl_lin[k]);)
    PF_X2(CN[k] = cnn[k] * f2{rce0[k], rce1[k]};)
    PF_X2(Cl[k] = Cl_lin[k];)
    PF_X2(Cd[k] = fma2(CN[k], se[k], CTc[k]);)
    PF_X2(CM[k] = -CN[k] * cmk[k];)
    if (any_stall) {  // :427-448
      f2 dn[2], c9s[2], CTs[2], cms[2], rdn[2], cts[2], ctc[2], CNs[2], Cls[2], Cds[2], CMs[2];
      PF_X2(dn[k] = fma2(sp2(0.44f), f2{__builtin_fabsf(se[k].x), __builtin_fabsf(se[k].y)}, sp2(0.56f));)
      PF_X2(c9s[k] = Cd90[k] * se[k];)
      PF_X2(CTs[k] = hcd[k] * ce[k];)
      PF_X2(cms[k] = fma2(sp2(0.175f * (2.0f / kPi)), f2{__builtin_fabsf(ae[k].x), __builtin_fabsf(ae[k].y)}, sp2(0.075f));)
      PF_X2(rdn[k] = f2{frcp(dn[k].x), frcp(dn[k].y)};)
      PF_X2(cts[k] = CTs[k] * se[k];)
      PF_X2(ctc[k] = CTs[k] * ce[k];)
      PF_X2(CNs[k] = c9s[k] * (rdn[k] - S[k].exp_term);)
      PF_X2(Cls[k] = fma2(CNs[k], ce[k], -cts[k]);)
      PF_X2(Cds[k] = fma2(CNs[k], se[k], ctc[k]);)
      PF_X2(CMs[k] = -CNs[k] * cms[k];)
      PF_X2(Cl[k] = f2{lin0[k] ? Cl[k].x : Cls[k].x, lin1[k] ? Cl[k].y : Cls[k].y};)
      PF_X2(Cd[k] = f2{lin0[k] ? Cd[k].x : Cds[k].x, lin1[k] ? Cd[k].y : Cds[k].y};)
      PF_X2(CM[k] = f2{lin0[k] ? CM[k].x : CMs[k].x, lin1[k] ? CM[k].y : CMs[k].y};)
    }
    // :485-498
    f2 L[2], D[2], qc[2], dsa[2], dca[2], tm[2], ty0[2];
    PF_X2(L[k] = Cl[k] * QA[k]; D[k] = Cd[k] * QA[k];)
    PF_X2(qc[k] = QA[k] * CM[k];)
    PF_X2(dsa[k] = D[k] * sa[k]; dca[k] = D[k] * ca[k];)
    PF_X2(tm[k] = qc[k] * S[k].chord;)
    PF_X2(o[k].fn = fma2(L[k], ca[k], dsa[k]);)
    PF_X2(o[k].fp = fma2(L[k], sa[k], -dca[k]);)
    PF_X2(ty0[k] = fma2(-S[k].rx, o[k].fn, tm[k]);)
    PF_X2(o[k].ty = fma2(S[k].rz, o[k].fp, ty0[k]);)
  }
#undef PF_X2
  // one lift-+z surface's force and torque into the body totals (the tail of surface<false>)
  PF_DEV static void accumulate(const float ry, const float fp, const float fn, const float ty, v3& F, v3& tau) {
    F.x += fp;
    F.z += fn;
    tau.x = fmaf(ry, fn, tau.x);
    tau.y += ty;
    tau.z = fmaf(-ry, fp, tau.z);
  }
  // one physics tick: update_physics (fixedwing.py:261-264) + stepSimulation + update_state (:266-291)
  // FLOOR = false: the same tick for a wave none of whose aircraft can come within reach of the ground during the ticks ahead (the
  // kernel's calm test: every velocity component is clamped to max_coord_vel, so an aircraft sinks at most that x the time ahead) --
  // no floor test, no contact solve, neither of the two out-of-line calls in the tick loop. Bit-identical for such a wave: the
  // floor code would have found `near` false for every lane.
  // SHARED (the dogfight): the contact response BETWEEN the aircraft of the world, one stage before the ground's -- publish the new
  // velocity, the world's first lane resolves the contacts, take back velocity and position-level shift.
  // TAB: fw_tab_cptr -- the constant table through the scalar cache, one row at a time (16 SGPRs live); FwTableV -- the table in
  // vector registers (the one-wave-per-SIMD instantiation): no load in the tick, both surface pairs evaluated side by side.
  template <bool FLOOR = true, bool SHARED = false, class TAB = fw_tab_cptr>
  PF_DEV void tick(const TAB& tab, const float xi, const pf_params* Pfull) {
#ifdef PF_FW_TICK_TRACE  // (per-tick clocks through global atomics: inflates the timeline, its own switch since r05)
    const unsigned long long pf_f0 = __builtin_readcyclecounter();
#endif
    v3 F{0.f, 0.f, 0.f}, tau{0.f, 0.f, 0.f};
    if constexpr (std::is_same<TAB, FwTableV>::value) {
      const FwSurf& Sv = tab.vtail;
      f2 a[2] = {f2{act[0], act[1]}, f2{act[2], act[4]}};
      const f2 c2[2] = {f2{cmd[0], cmd[1]}, f2{cmd[2], cmd[4]}};
      FwPairOut o[2];
      surface_pair_x2(tab.pair, a, c2, o);
      act[0] = a[0].x; act[1] = a[0].y; act[2] = a[1].x; act[4] = a[1].y;
      // the reference's accumulation order: ailerons (0, 1), h-tail (2), v-tail (3), main wing (4)
      accumulate(tab.pair[0].ry.x, o[0].fp.x, o[0].fn.x, o[0].ty.x, F, tau);
      accumulate(tab.pair[0].ry.y, o[0].fp.y, o[0].fn.y, o[0].ty.y, F, tau);
      accumulate(tab.pair[1].ry.x, o[1].fp.x, o[1].fn.x, o[1].ty.x, F, tau);
      __builtin_amdgcn_sched_barrier(0);
      const FwBody& Kb = tab.body;
      act[3] = fmaf(Sv.dt_tau, cmd[3] - act[3], act[3]);
      surface<true>(Sv, act[3], F, tau);
      accumulate(tab.pair[1].ry.y, o[1].fp.y, o[1].fn.y, o[1].ty.y, F, tau);
      __builtin_amdgcn_sched_barrier(0);
#ifdef PF_FW_TICK_TRACE  // (the surfaces' share of the tick: this branch had no counter in round 5 -- its timeline read 0.000 us)
      asm volatile("" ::"v"(F.x), "v"(tau.y));
      if ((threadIdx.x & 63u) == 0u) { trace_add(&g_solver_trace[0], 1ull); trace_add(&g_solver_trace[1], __builtin_readcyclecounter() - pf_f0); }
#endif
      tick_body<FLOOR, SHARED>(Kb, F, tau, xi, Pfull);
    } else {
      tick_scalar_table<FLOOR, SHARED>(tab, F, tau, xi, Pfull);
    }
  }
  template <bool FLOOR, bool SHARED>
  PF_DEV void tick_scalar_table(fw_tab_cptr tab, v3& F, v3& tau, const float xi, const pf_params* Pfull) {
#ifdef PF_FW_TICK_TRACE  // (per-tick clocks through global atomics: inflates the timeline, its own switch since r05)
    const unsigned long long pf_f0 = __builtin_readcyclecounter();
#endif
    // An opaque zero offset per tick keeps the per-surface constant loads inside the tick (16 SGPRs at
    // a time, see the file header) instead of hoisted and spilled; the scheduling barriers keep the
    // five surface bodies from being interleaved (which cost 255 VGPRs and scratch).
    uint32_t zoff;
    asm volatile("s_mov_b32 %0, 0" : "=s"(zoff));
    fw_tab_cptr tk = tab + zoff;
    FwPairOut tw;  // (h-tail, main wing): evaluated together, accumulated as surfaces 2 and 4 around the v-tail
    float ry4;
    {  // ailerons
      const FwSurf2 S = fw_load_surf2(&tk->pair[0]);
      f2 a = f2{act[0], act[1]};
      const FwPairOut o = surface_pair(S, a, f2{cmd[0], cmd[1]});
      act[0] = a.x; act[1] = a.y;
      accumulate(S.ry.x, o.fp.x, o.fn.x, o.ty.x, F, tau);
      accumulate(S.ry.y, o.fp.y, o.fn.y, o.ty.y, F, tau);
    }
    __builtin_amdgcn_sched_barrier(0);
    {  // horizontal tail + main wing
      const FwSurf2 S = fw_load_surf2(&tk->pair[1]);
      f2 a = f2{act[2], act[4]};
      tw = surface_pair(S, a, f2{cmd[2], cmd[4]});
      act[2] = a.x; act[4] = a.y;
      accumulate(S.ry.x, tw.fp.x, tw.fn.x, tw.ty.x, F, tau);
      ry4 = S.ry.y;
    }
    __builtin_amdgcn_sched_barrier(0);
    {  // vertical tail
      const FwSurf S = fw_load_surf(&tk->vtail);
      act[3] = fmaf(S.dt_tau, cmd[3] - act[3], act[3]);
      surface<true>(S, act[3], F, tau);
    }
    accumulate(ry4, tw.fp.y, tw.fn.y, tw.ty.y, F, tau);
    __builtin_amdgcn_sched_barrier(0);
    const FwBody K = fw_load_body(&tk->body);
#ifdef PF_FW_TICK_TRACE  // (per-tick clocks through global atomics: inflates the timeline, its own switch since r05)
    asm volatile("" ::"v"(F.x), "v"(tau.y));
    if ((threadIdx.x & 63u) == 0u) { trace_add(&g_solver_trace[0], 1ull); trace_add(&g_solver_trace[1], __builtin_readcyclecounter() - pf_f0); }
#endif
    tick_body<FLOOR, SHARED>(K, F, tau, xi, Pfull);
  }
  // the rest of the tick: motor, collision detection, the free-base tick, contact response, update_state
  template <bool FLOOR, bool SHARED>
  PF_DEV void tick_body(const FwBody& K, v3 F, v3 tau, const float xi, const pf_params* Pfull) {
#ifdef PF_FW_TICK_TRACE  // (per-tick clocks through global atomics: inflates the timeline, its own switch since r05)
    const unsigned long long pf_f1 = __builtin_readcyclecounter();
#endif
    {  // motor (motors.py:110-195), at the base origin along +x
      float t = fmaf(K.m_a, cmd[5] - thr, thr);
      t = fmaf(xi * t, K.m_noise, t);
      thr = t;
      const float k = t * __builtin_fabsf(t);
      F.x = fmaf(k, K.fmax, F.x);
      tau.x = fmaf(k, K.tmax, tau.x);
    }
    // collision detection at the pre-integration pose
    // (within reach of the ground slab: one bounding radius (+ contact margin) of its top face, not below its bottom face,
    //  not beyond its rim -- an aircraft that has flown off the 30 m slab and keeps falling is "under the floor" for seconds,
    //  and the six 15-axis box tests per tick for such lanes were ~20 % of this kernel's instructions)
    const bool near = FLOOR && ((p.z - K.bound_radius) <= 0.0f) && ((p.z + K.bound_radius) >= K.slab_bottom) &&
                      (__builtin_fabsf(p.x) - K.bound_radius <= K.slab_xy) && (__builtin_fabsf(p.y) - K.bound_radius <= K.slab_xy);
    const bool persisted = contact_now;  // contact points left by the previous tick persist up to the breaking distance
    contact_now = false;
    if (FLOOR) {
      if (__any(near)) {
        if (near) contact_now = fw_floor_contact(p.x, p.y, p.z, R, Pfull, persisted);
      }
    }
    contact_now = contact_now || peer_contact;
    // free-base multibody tick, composite of point masses: COM offset, full symmetric inertia
    const v3 com{K.com[0], K.com[1], K.com[2]};
    tau = tau - cross(com, F);
    const v3 h = symmul(K.H, wb);
    const v3 wdb = symmul(K.iI, tau - cross(wb, h));
    const v3 wd = mul(R, wdb);
    v3 a = K.inv_mass * mul(R, F);
    a.z += K.gravity_z;
    const v3 cw = mul(R, com);
    a = a - cross(wd, cw) - cross(w, cross(w, cw));
    w = v3{med3(fmaf(wd.x, K.dt, w.x), -K.vmax, K.vmax), med3(fmaf(wd.y, K.dt, w.y), -K.vmax, K.vmax), med3(fmaf(wd.z, K.dt, w.z), -K.vmax, K.vmax)};
    v = v3{med3(fmaf(a.x, K.dt, v.x), -K.vmax, K.vmax), med3(fmaf(a.y, K.dt, v.y), -K.vmax, K.vmax), med3(fmaf(a.z, K.dt, v.z), -K.vmax, K.vmax)};
    v3 shift{0.0f, 0.0f, 0.0f};
    if (SHARED) {
      if (wvel_ != nullptr && Pfull->contact_response) {  // (wave-uniform)
        float* o = wvel_ + wtid * kPairVelStride;
        o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = w.x; o[4] = w.y; o[5] = w.z; o[6] = 0.0f; o[7] = 0.0f; o[8] = 0.0f;
        lds_sync_wave();
        if (__any(world_touch)) {
          pair_stage_dev(Pfull, wpose_, wvel_, prec_, cws_floats, wtid, wA, world_touch);
          v = v3{o[0], o[1], o[2]}; w = v3{o[3], o[4], o[5]};
          shift = v3{o[6], o[7], o[8]};
        }
      }
    }
    float lift = 0.0f;  // contact response (see quadx_fast.hpp / uav_vehicles.hpp:contact_solve_dev)
    bool act = false;  // can a contact constraint act at all this tick? (Body::contact_may_act)
    if (FLOOR && near) {
      const float r0 = Pfull->bound_radius, slop = Pfull->contact_slop;
      const float low = p.z - r0, vlow = v.z - fsqrt(dot(w, w)) * r0;
      act = ((fmaf(K.dt, vlow, low + slop) < 0.0f) || (low < -slop)) && (low <= (persisted ? Pfull->contact_break_distance : Pfull->contact_margin));  // (no vertex can be within reach otherwise)
    }
    if (FLOOR) {
      if (__any(act)) {
        const ContactOut o = contact_solve_dev(Pfull, cws, need_cap_of(act && Pfull->contact_response, cws_floats, persisted), p, q, v, w);
        v = o.v; w = o.w;  // (unchanged for a lane that did not ask or has no contact vertex)
        lift = Pfull->contact_erp * o.deepest;  // (already net of the slop)
      }
    }
    if (SHARED) p = v3{fmaf(K.dt, v.x, p.x) + shift.x, fmaf(K.dt, v.y, p.y) + shift.y, fmaf(K.dt, v.z, p.z) + lift + shift.z};
    else p = v3{fmaf(K.dt, v.x, p.x), fmaf(K.dt, v.y, p.y), FLOOR ? fmaf(K.dt, v.z, p.z) + lift : fmaf(K.dt, v.z, p.z)};
    q = quat_integrate(q, w, K.half_dt);
    derive();
    contact_step |= contact_now;
#ifdef PF_FW_TICK_TRACE  // (per-tick clocks through global atomics: inflates the timeline, its own switch since r05)
    asm volatile("" ::"v"(wb.x), "v"(vb.z));
    if ((threadIdx.x & 63u) == 0u) {  // (diagnostic build: per tick and wave, clocks in the five surfaces / in the rest of the tick)
      const unsigned long long pf_f2 = __builtin_readcyclecounter();
      trace_add(&g_solver_trace[7], pf_f2 - pf_f1);
    }
#endif
  }
};

// Same flat shape as quadx_m0_env_kernel (quadx_fast.hpp): prologue, one loop over the env step's
// Aviary steps with a single per-lane predicate, epilogue. State groups: g0 p+dist, g1 q, g2 v+w.x,
// g3 w.yz+act0,1, g4 act2..4+throttle, g5 ints, g6..8 the 4x3 targets (Fixedwing::load/store layout).
// ROLL: as in quadx_m0_env_kernel -- 0 one env step per launch, 1 pf_rollout with on-device action sampling (no vector-memory
// load in the loop), 2 pf_rollout over a given action sequence.
// WPS (as in quadx_m0_env_kernel): 2 = two waves per SIMD, 256 registers, any batch; 1 = one wave per SIMD, 512 registers (the
// overflow lives in AGPRs, not on the stack), chosen for batches of at most one wave per SIMD, where a second resident wave has
// nothing to run. The calm-wave tick (second tick instantiation without the floor code's call sites) needs that budget: under 256
// registers it spilled another 128 B per lane inside the tick loop (23.9 -> 28.3 us per step at 65 536 lanes, profiles/README.md, r04).
// (the calm-wave split -- a second tick instantiation without the floor code, as in quadx_fast.hpp -- stays off in both register
//  budgets: 23.9 -> 28.3 us under 256 registers (spills inside the tick loop, r04), 23.9 against 22.3 under 512 (the second
//  instantiation's AGPR copies, r05). The code path is kept: the dogfight's aircraft use tick<false>.)
template <int NOISE, int ROLL, int WPS = 2>
__global__ void __launch_bounds__(64, WPS) fixedwing_wp_env_kernel(const FwK K, const FwTable* table_g, const pf_buffers B,
                                                                 const pf_params* __restrict__ Pfull, const float4* __restrict__ tmpl,
                                                                 const int n, const uint64_t lane0, const int op,
                                                                 const uint8_t* __restrict__ mask, const int k_steps, const uint32_t step0) {
  constexpr bool CALM = false;
  constexpr bool ROLLOUT = ROLL != 0;
  constexpr bool GIVEN = ROLL == 2;
  constexpr int kMaxD = 13 + 4 + 6 + 12;
  __shared__ __attribute__((aligned(16))) float tile[64 * kMaxD];
  __shared__ __attribute__((aligned(16))) float ktab[112];  // (WPS == 1) the constant table on its way to the vector registers
  __shared__ int spos[64];        // the cooperative waypoint sampling's (lane, counter) exchange
  __shared__ uint32_t sctr[64];
  static_assert(64 * kMaxD >= 2 * 64 * 16, "prepare_targets stages uniforms and targets in the observation tile");
  const int tid = threadIdx.x;
  const int wave_base = blockIdx.x * 64;
  const int lane = wave_base + tid;
  const bool valid = lane < n;
  const size_t li = valid ? (size_t)lane : (size_t)(n - 1);
  const size_t N = (size_t)n;
  const float4* Sin = reinterpret_cast<const float4*>(B.state);
  float4* Sout = reinterpret_cast<float4*>(B.state);
  fw_tab_cptr surf = (fw_tab_cptr)(uintptr_t)table_g;

#ifdef PF_PHASE_TRACE
  unsigned long long pf_ts[kPhaseStamps];
  pf_ts[11] = __builtin_amdgcn_s_memrealtime();
#endif
  PF_STAMP(0);
  FwHot V;
  static_assert(64 * kMaxD >= kContactSlotFloats, "the contact solver's LDS regions alias the observation tile: at least one worst-case region");
  // the calm test's constants (one scalar load of the table's body row): reach of the floor code, how far an aircraft can sink
  // over an env step / over an Aviary step (max_coord_vel x ticks x dt, with a margin for the rounding of the position updates)
  const float Kc_bound_radius = surf->body.bound_radius;
  const float calm_sink_env = surf->body.vmax * surf->body.dt * (float)(2 * K.env_step_ratio) * 1.001f + 1e-3f;
  const float calm_sink_av = surf->body.vmax * surf->body.dt * 2.0f * 1.001f + 1e-3f;
  V.cws = (lds_fptr)tile;
  // (WPS == 1: the constant table in vector registers for the whole launch -- FwTableV)
  FwTableV TV{};
  float tgt[4][3];
  float new_dist, old_dist;
  int step_count, flags, n_left;
  uint32_t rng_ctr;
  f8 zn;
  float4 a_pre = float4{0.f, 0.f, 0.f, 0.f};  // (one step per launch) this step's action, requested with the state (quadx_fast.hpp)
  {
    float4 gi = Sin[5 * N + li];
    float4 g0 = Sin[0 * N + li], g1 = Sin[1 * N + li], g2 = Sin[2 * N + li], g3 = Sin[3 * N + li], g4 = Sin[4 * N + li];
    if (ROLL == 0 && op == 0) a_pre = reinterpret_cast<const float4*>(B.actions)[li];
    // (the rare code's prefetch, uav_vehicles.hpp: only where a second wave shares the SIMD -- with one wave per SIMD every wave takes
    //  the same time and the prefetching ones were the launch's slowest: 21.1 -> 20.7 us without)
    if (WPS != 1 && blockIdx.x < kRareTextPrefetchBlocks) rare_text_prefetch((int)threadIdx.x);
    // (the constant table BEHIND the state groups: loads return in order, and in front of them its 28 requests held the int group --
    //  which the Philox call below waits for -- back by 0.7 us; profiles/r05/phase_fixedwing.txt)
    float tw0 = 0.0f, tw1 = 0.0f;
    if (WPS == 1) {
      const float* tg = reinterpret_cast<const float*>(table_g);
      tw0 = tg[tid];
      tw1 = tg[64 + (tid < 48 ? tid : 47)];
    }
    rng_ctr = (uint32_t)__float_as_int(gi.z);
    PF_STAMP(1);
    if (NOISE == PF_NOISE_PHILOX) {
      if (op == 0) zn = normal8(philox4x32(K.seed_lo, K.seed_hi, (uint32_t)(lane0 + li), rng_ctr, 0u, 0u));
    }
    if (WPS == 1) {  // (one wave per workgroup; LDS operations of a wave complete in issue order)
      ktab[tid] = tw0;
      ktab[64 + (tid < 48 ? tid : 47)] = tw1;
      TV = fw_table_from_lds((lds_fptr)ktab);
    }
    PF_STAMP(2);
    V.p = v3{g0.x, g0.y, g0.z}; new_dist = g0.w;
    V.q = quat{g1.x, g1.y, g1.z, g1.w};
    V.v = v3{g2.x, g2.y, g2.z};
    V.w = v3{g2.w, g3.x, g3.y};
    V.act[0] = g3.z; V.act[1] = g3.w; V.act[2] = g4.x; V.act[3] = g4.y; V.act[4] = g4.z; V.thr = g4.w;
    step_count = __float_as_int(gi.x); flags = __float_as_int(gi.y); n_left = __float_as_int(gi.w);
    float4 a = Sin[6 * N + li], b = Sin[7 * N + li], c = Sin[8 * N + li];
    tgt[0][0] = a.x; tgt[0][1] = a.y; tgt[0][2] = a.z; tgt[1][0] = a.w;
    tgt[1][1] = b.x; tgt[1][2] = b.y; tgt[2][0] = b.z; tgt[2][1] = b.w;
    tgt[2][2] = c.x; tgt[3][0] = c.y; tgt[3][1] = c.z; tgt[3][2] = c.w;
  }
  old_dist = new_dist;
  V.contact_now = (flags & PF_F_CONTACT) != 0;
  V.contact_step = false;
  V.derive();
#ifdef PF_PHASE_TRACE
  asm volatile("" ::"v"(V.wb.x), "v"(V.vb.z));
#endif
  PF_STAMP(3);
  bool term = (flags & PF_F_TERMINATED) != 0, trunc = (flags & PF_F_TRUNCATED) != 0;

  bool active;
  if (op == 1) active = (mask == nullptr) || (mask[li] != 0);
  else active = true;
  active = active && valid;

  float act0 = 0.f, act1 = 0.f, act2 = 0.f, act3 = 0.f;
  float reward = 0.0f;
  bool pop_pending = false;
  bool was_reset = false;
  const int D = (K.angle_repr ? 13 : 12) + 4 + 6 + 3 * K.num_targets;

  auto pop_target = [&]() {
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
      for (int c = 0; c < 3; ++c) tgt[k][c] = tgt[k + 1][c];
    n_left -= 1;
  };
  auto lds_sync = [&]() {  // one wave per workgroup: see quadx_fast.hpp
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
  };
  // Waypoint sampling (waypoint_handler.py:53-83), cooperatively (quadx_fast.hpp: prepare_targets): a resetting lane needs three
  // Philox calls and, per target, two sine / cosine pairs -- some 700 instructions that the one or two resetting lanes of a wave walked
  // through while the other lanes idled, in about every second wave of a launch, and a launch lasts as long as its slowest wave.
  // Dealt out over the 64 lanes instead: pass 1, one (lane, call) pair per lane -> the uniforms, through the observation tile (idle
  // here); pass 2, one (lane, target) pair per lane -> the target, the same arithmetic on the same numbers as the per-lane path
  // (which injected draws, B.u_targets, keep). Wave-uniform call.
  const bool coop_targets = !((NOISE == PF_NOISE_INJECT) && (B.u_targets != nullptr));
  auto prepare_targets = [&](bool reset_now) {
    if (!coop_targets) return;
    const unsigned long long m = __ballot(reset_now);
    if (m == 0ull) return;
    const int r = __popcll(m);
    if (reset_now) {
      spos[__popcll(m & ((1ull << tid) - 1ull))] = tid;
      sctr[tid] = rng_ctr;
    }
    lds_sync();
    const int nt = K.num_targets;
    float* const U = tile;             // [lane][16]: the uniforms of calls 0 .. 2
    float* const TG = tile + 64 * 16;  // [lane][16]: four targets x (x, y, z, -)
    for (int base = 0; base < r * 3; base += 64) {
      const int j = base + tid;
      if (j < r * 3) {
        const int which = j / 3, call = j - which * 3;
        const int src = spos[which];
        const f4 u = uniform4(philox4x32(K.seed_lo, K.seed_hi, (uint32_t)(lane0 + (uint64_t)(wave_base + src)), sctr[src], (uint32_t)call, 2u));
        float* o = U + src * 16 + call * 4;
        o[0] = u.a; o[1] = u.b; o[2] = u.c; o[3] = u.d;
      }
    }
    lds_sync();
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
        o[0] = dist * sph * ct; o[1] = dist * sph * st; o[2] = zz > K.min_height ? zz : K.min_height; o[3] = 0.0f;
      }
    }
    lds_sync();
  };
  // env.reset() for this lane (fixedwing_base_env.py:136-211 begin_reset/end_reset,
  // fixedwing_waypoints_env.py:101-114): settled spawn state from the template, fresh waypoints.
  auto reset_lane = [&]() {
    const float4 t0 = tmpl[0], t1 = tmpl[1], t2 = tmpl[2], t3 = tmpl[3], t4 = tmpl[4];
    V.p = v3{t0.x, t0.y, t0.z};
    V.q = quat{t1.x, t1.y, t1.z, t1.w};
    V.v = v3{t2.x, t2.y, t2.z};
    V.w = v3{t2.w, t3.x, t3.y};
    V.act[0] = t3.z; V.act[1] = t3.w; V.act[2] = t4.x; V.act[3] = t4.y; V.act[4] = t4.z; V.thr = t4.w;
    V.contact_now = false; V.contact_step = false;
    V.derive();
    step_count = 0; term = false; trunc = false; flags = 0; pop_pending = false;
    act0 = act1 = act2 = act3 = 0.f;
    const int nt = K.num_targets;  // waypoint_handler.py:53-83
    n_left = nt;
    if (coop_targets) {  // sampled by prepare_targets(): this lane's 4 x (x, y, z, -)
      const float4* t4 = reinterpret_cast<const float4*>(tile + 64 * 16 + tid * 16);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (i < nt) {
          const float4 t = t4[i];
          tgt[i][0] = t.x; tgt[i][1] = t.y; tgt[i][2] = t.z;
        }
      }
    } else {
    f4 u0, u1, u2;
    const bool inj = (NOISE == PF_NOISE_INJECT) && (B.u_targets != nullptr);
    if (!inj) {
      u0 = uniform4(philox4x32(K.seed_lo, K.seed_hi, (uint32_t)(lane0 + li), rng_ctr, 0u, 2u));
      u1 = uniform4(philox4x32(K.seed_lo, K.seed_hi, (uint32_t)(lane0 + li), rng_ctr, 1u, 2u));
      u2 = uniform4(philox4x32(K.seed_lo, K.seed_hi, (uint32_t)(lane0 + li), rng_ctr, 2u, 2u));
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
          auto u = [&](int flat) { return pick4(flat < 4 ? u0 : (flat < 8 ? u1 : u2), (uint32_t)flat & 3u); };
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
    float dx = tgt[0][0] - V.p.x, dy = tgt[0][1] - V.p.y, dz = tgt[0][2] - V.p.z;
    old_dist = INFINITY;
    new_dist = fsqrt(fmaf(dx, dx, fmaf(dy, dy, dz * dz)));
    rng_ctr += 1;
    was_reset = true;
  };
  // observation row (fixedwing_base_env.py:75-92,213-224; fixedwing_waypoints_env.py:116-167)
  auto write_obs_row = [&]() {
    float* row = tile + tid * D;
    // (the Euler-angle arguments are entries of the rotation matrix derive() holds for the unit q:
    //  -2(xz - wy) = -R20, 2(yz + wx) = R21, w2-x2-y2+z2 = R22, 2(xy + wz) = R10, w2+x2-y2-z2 = R00)
    float sarg = -V.R.m20;
    quat qe;
    v3 rpy;
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
    row[k++] = act0; row[k++] = act1; row[k++] = act2; row[k++] = act3;
    row[k++] = V.act[0]; row[k++] = V.act[1]; row[k++] = V.act[2]; row[k++] = V.act[3]; row[k++] = V.act[4];
    row[k++] = V.thr;
    m3 Re = rot_from_quat(qe);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (i < K.num_targets) {
        v3 d = mulT(Re, v3{tgt[i][0] - V.p.x, tgt[i][1] - V.p.y, tgt[i][2] - V.p.z});
        bool live = i < n_left;
        row[k++] = live ? d.x : 0.0f; row[k++] = live ? d.y : 0.0f; row[k++] = live ? d.z : 0.0f;
      }
    }
  };
  const bool wave_all = __all(active || !valid);
  auto flush_tile = [&](float* out) {
    lds_sync();
    if (wave_all) {
      const int rows = min(64, n - wave_base);
      const int total = rows * D;
      float* g = out + (size_t)wave_base * D;
      stream_tile(tile, g, total, tid);
    } else if (active) {
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
  prepare_targets(do_reset);
  if (do_reset) reset_lane();
  PF_STAMP(4);

  // ---------------------------------------------------------------- the env step
  const bool stepping = active && !was_reset && op == 0;
#pragma unroll
  for (int k = 0; k < 6; ++k) V.cmd[k] = 0.f;
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
    const float thr_sp = K.throttle_remap ? fmaf(a.w, 0.5f, 0.5f) : a.w;  // fixedwing_base_env.py:260
    // update_control, mode 0 (fixedwing.py:143-144,246-250): constant over the env step
    V.cmd[0] = a.x; V.cmd[1] = -a.x; V.cmd[2] = a.y; V.cmd[3] = -a.z; V.cmd[4] = -a.y; V.cmd[5] = thr_sp;
    reward = -0.1f;
  }
  bool go = stepping && !(term || trunc);  // fixedwing_base_env.py:262-263
  // Calm waves (as in quadx_fast.hpp): no aircraft of the wave can come within reach of the ground slab during the env step -- each
  // velocity component is clamped to max_coord_vel after every tick, so an aircraft sinks at most max_coord_vel x the time ahead
  // -- and the wave runs ticks instantiated without the floor code (no out-of-line call in the tick loop). A wave that is not calm
  // over the env step asks again per Aviary step, over its two ticks.
  const float calm_reach = Kc_bound_radius;
  const bool calm_env = CALM && __all(!go || (V.p.z - calm_reach > calm_sink_env));
  for (int s = 0; s < K.env_step_ratio; ++s) {
    if (!__any(go)) break;
    const bool calm_s = calm_env || (CALM && __all(!go || (V.p.z - calm_reach > calm_sink_av)));
    if (go) {
      float xi0, xi1;
      if (NOISE == PF_NOISE_PHILOX) { xi0 = 1.0f + pick8(zn, (uint32_t)(2 * s)); xi1 = 1.0f + pick8(zn, (uint32_t)(2 * s + 1)); }
      else if (NOISE == PF_NOISE_INJECT) { xi0 = B.xi[(size_t)(2 * s) * N + li]; xi1 = B.xi[(size_t)(2 * s + 1) * N + li]; }
      else { xi0 = 0.f; xi1 = 0.f; }
      V.contact_step = false;
      if (WPS == 1) {
        if (calm_s) {
#pragma unroll 1
          for (int t = 0; t < 2; ++t) V.template tick<false, false, FwTableV>(TV, t == 0 ? xi0 : xi1, Pfull);
        } else {
#pragma unroll 1
          for (int t = 0; t < 2; ++t) V.template tick<true, false, FwTableV>(TV, t == 0 ? xi0 : xi1, Pfull);
        }
      } else if (calm_s) {
#pragma unroll 1
        for (int t = 0; t < 2; ++t) V.template tick<false>(surf, t == 0 ? xi0 : xi1, Pfull);
      } else {
#pragma unroll 1
        for (int t = 0; t < 2; ++t) V.template tick<true>(surf, t == 0 ? xi0 : xi1, Pfull);
      }
      // compute_state side effects + compute_term_trunc_reward
      if (pop_pending) { pop_target(); pop_pending = false; }
      float dx = tgt[0][0] - V.p.x, dy = tgt[0][1] - V.p.y, dz = tgt[0][2] - V.p.z;
      old_dist = new_dist;
      new_dist = fsqrt(fmaf(dx, dx, fmaf(dy, dy, dz * dz)));
      if (step_count > K.max_steps) trunc = true;                                             // fixedwing_base_env.py:229
      if (V.contact_step) { reward = -100.0f; flags |= PF_F_INFO_COLLISION; term = true; }    // :233-236
      if (dot(V.p, V.p) > K.dome2) { reward = -100.0f; flags |= PF_F_INFO_OOB; term = true; } // :239-242
      if (!K.task_sparse) {  // fixedwing_waypoints_env.py:174-178
        float progress = (isinf(old_dist + new_dist)) ? 0.0f : old_dist - new_dist;
        reward += __builtin_fmaxf(3.0f * progress, 0.0f);
        reward = fmaf(K.wp_dist_reward, frcp(new_dist), reward);
      }
      if (new_dist < K.goal_reach) {  // :181-190
        reward = 100.0f;
        pop_pending = true;
        if (n_left - 1 == 0) { trunc = true; flags |= PF_F_INFO_COMPLETE; }
      }
      go = !(term || trunc);
    }
  }
  PF_STAMP(5);
  const float out_reward = stepping ? reward : 0.0f;
  const bool out_term = stepping && term, out_trunc = stepping && trunc;
  if (stepping) {
    step_count += 1; rng_ctr += 1;
    // NaN / Inf guard: any non-finite state word poisons the sum (see quadx_fast.hpp)
    const float chk = ((V.p.x + V.p.y) + (V.p.z + V.q.x)) + ((V.q.y + V.q.z) + (V.q.w + V.v.x)) + ((V.v.y + V.v.z) + (V.w.x + V.w.y)) +
                      ((V.w.z + V.thr) + (V.act[0] + V.act[1])) + ((V.act[2] + V.act[3]) + V.act[4]);
    if (!(__builtin_fabsf(chk) < INFINITY)) flags |= PF_F_NONFINITE;
  }

  // ---------------------------------------------------------------- SAME_STEP auto-reset
  if (K.autoreset == PF_AUTORESET_SAME_STEP) {
    const bool same = stepping && (term || trunc);
    if (__any(same)) {
      if (B.final_obs != nullptr) {
        if (active) write_obs_row();
        flush_tile(B.final_obs + toff * D);
      }
      if (B.final_info != nullptr && same) {  // gymnasium's final_info: the episode's flags / targets left, pre-reset
        B.final_info[2 * (toff + li) + 0] = (flags & ~(PF_F_TERMINATED | PF_F_TRUNCATED | PF_F_CONTACT)) | (term ? PF_F_TERMINATED : 0) |
                                            (trunc ? PF_F_TRUNCATED : 0) | (V.contact_now ? PF_F_CONTACT : 0);
        B.final_info[2 * (toff + li) + 1] = n_left - (pop_pending ? 1 : 0);
      }
      prepare_targets(same);
      if (same) reset_lane();
    }
  }

  // ---------------------------------------------------------------- outputs
  PF_STAMP(6);
  if (active) write_obs_row();
  PF_STAMP(7);
  flush_tile(B.obs + toff * D);
  PF_STAMP(8);
  if (active) {
    if (pop_pending) { pop_target(); pop_pending = false; }
    flags = (flags & ~(PF_F_TERMINATED | PF_F_TRUNCATED | PF_F_CONTACT)) | (term ? PF_F_TERMINATED : 0) |
            (trunc ? PF_F_TRUNCATED : 0) | (V.contact_now ? PF_F_CONTACT : 0);
    if (op == 0) {
      B.reward[toff + li] = out_reward;
      B.terminated[toff + li] = out_term ? 1 : 0;
      B.truncated[toff + li] = out_trunc ? 1 : 0;
    }
  }
  // the next step's motor-noise normals (keyed by the event counter this step left behind)
  if (ROLLOUT && NOISE == PF_NOISE_PHILOX && it + 1 < KS)
    zn = normal8(philox4x32(K.seed_lo, K.seed_hi, (uint32_t)(lane0 + li), rng_ctr, 0u, 0u));
  }  // for it
  if (active) {  // the persistent state goes back to HBM once per launch
    Sout[0 * N + li] = float4{V.p.x, V.p.y, V.p.z, new_dist};
    Sout[1 * N + li] = float4{V.q.x, V.q.y, V.q.z, V.q.w};
    Sout[2 * N + li] = float4{V.v.x, V.v.y, V.v.z, V.w.x};
    Sout[3 * N + li] = float4{V.w.y, V.w.z, V.act[0], V.act[1]};
    Sout[4 * N + li] = float4{V.act[2], V.act[3], V.act[4], V.thr};
    Sout[5 * N + li] = float4{__int_as_float(step_count), __int_as_float(flags), __int_as_float((int)rng_ctr), __int_as_float(n_left)};
    Sout[6 * N + li] = float4{tgt[0][0], tgt[0][1], tgt[0][2], tgt[1][0]};
    Sout[7 * N + li] = float4{tgt[1][1], tgt[1][2], tgt[2][0], tgt[2][1]};
    Sout[8 * N + li] = float4{tgt[2][2], tgt[3][0], tgt[3][1], tgt[3][2]};
  }
#ifdef PF_PHASE_TRACE
  PF_STAMP(9);
  __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0): every store acknowledged
  PF_STAMP(10);
  pf_ts[12] = __builtin_amdgcn_s_memrealtime();
  if (ROLL == 0 && tid == 0 && blockIdx.x < 4096) {
#pragma unroll
    for (int i = 0; i < kPhaseStamps; ++i) g_phase_trace[blockIdx.x * kPhaseStamps + i] = pf_ts[i];
  }
#endif
}

}  // namespace pf
