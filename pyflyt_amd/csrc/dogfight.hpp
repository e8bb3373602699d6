// dogfight.hpp -- MAFixedwingDogfightEnv (pz_envs/fixedwing_envs/ma_fixedwing_dogfight_env.py,
// ma_fixedwing_base_env.py) as one fused launch per env step.
//
// One lane = one Acrowing aircraft (the generic Fixedwing vehicle of uav_vehicles.hpp with the acrowing parameter block);
// the A = 2 team_size <= 8 aircraft of a world are adjacent lanes of one wavefront, so everything the reference couples
// them through goes through LDS:
//   * physics: pose + contact bit before every tick (drone-drone box tests, the world-global contact gate -- world_exchange,
//     shared with the PettingZoo QuadX task);
//   * the env's update_states() after every Aviary step (:305-307 of the base env): each lane publishes its attitude (nose
//     position shifted to the body centre, :318), computes ITS ROW of the n x n engagement matrices (distance, angle, in
//     range / chasing / hit), publishes the row so that the others can read their column (hits received, the transposed
//     reward terms), then its health, then -- after the collision / out-of-bounds overrides -- its health again for the
//     element-wise team-win rule (:682-690).
// Arithmetic differences from the reference, both inside the fp32 tolerance of the parity tests: the engagement angle is
// atan2(|sep x fwd|, sep . fwd) instead of arccos(sep . fwd / |sep|) (the same angle, but arccos loses half the digits near
// 0 -- exactly where the cone of fire is); the diagonal of the matrices is never formed (the reference fills it with NaN
// and zeroes it afterwards).
//
// State groups (float4 each, [group][lane]): 0-5 the Fixedwing vehicle's (body, surfaces, ints); 6 (health, accumulated
// reward, received hits, dogfight flags); 7 current action; 8 past action; 9-10 this lane's row of current_distances;
// 11-12 its row of current_angles; 13 (spawn x, y, z, roll); 14 (spawn pitch, yaw, -, -); 15 (current action 4, 5; past action
// 4, 5: six-wide actions only, df_action_dim).
#pragma once
#include "../../include/pyflyt_amd.h"
#include "uav_device.hpp"
#include "uav_vehicles.hpp"
#include "fixedwing_fast.hpp"
#include "shared_world.hpp"

namespace pf {

constexpr int kDfMaxAgents = 8;
constexpr int kDfMaxObs = 23 + (kDfMaxAgents - 1) * 14;  // 121
constexpr int kDfGroups = 16;
enum {  // group 6, word 3
  DF_ALIVE = 1,      // still in self.agents
  DF_ACC_TERM = 2,   // accumulated_terminations / _truncations (sticky over the episode)
  DF_ACC_TRUNC = 4,
  DF_INACTIVE = 8,   // dead, on the ground and at rest (:505-510)
  DF_INFO_DEAD = 16, DF_INFO_COLLISION = 32, DF_INFO_OOB = 64, DF_INFO_TEAM_WIN = 128,
  DF_FROZEN = 256,   // df_freeze_wrecks: stopped where it hit the ground
  // a wreck that has really come to rest: DF_INACTIVE, on the ground and |w| small for kDfRestUpdates consecutive updates (the
  // count lives in bits 9-12). `inactive` by itself -- recomputed by the reference on every update (:505-510) -- can hold for a
  // single 120 Hz sample at the apex of a bounce or a rock; such an aircraft keeps moving, here as there.
  DF_REST_SHIFT = 9, DF_REST_MASK = 15 << 9,
  DF_AT_REST = 8192
};
constexpr int kDfRestUpdates = 8;

// ---------------------------------------------------------------- the aircraft
// Two interchangeable vehicles behind the kernel: the generic Fixedwing of uav_vehicles.hpp (any five-surface airframe the
// parameter block describes; surface constants in an LDS table) and the specialised tick of fixedwing_fast.hpp (FwHot: the
// reference airframes' fixed structure folded away, packed-fp32 surfaces, constants through the scalar cache; 2.3x fewer
// instructions). pf_ctx_create picks the second whenever fw_table_from_params accepts the airframe (acrowing does).
struct DfGenericVeh : Fixedwing {
  PF_DEV void attach(const FwTable*) {}
  PF_DEV void tick(const pf_params& P, float xi) { Fixedwing::template tick<true>(P, xi); }  // (with the pair stage: Body::tick<SHARED>)
  PF_DEV void share(const float* wpose, float* wvel, float*, int tid, int A) { b.wpose_ = wpose; b.wvel_ = wvel; b.wtid = tid; b.wA = A; }
};
struct DfFastBody : FwHot {
  v3 rpy;
  const pf_params* pdev;
  PF_DEV void contact_regions(const pf_params&, int floats) { cws_floats = floats; }
};
struct DfFastVeh {
  static constexpr int TABLE_FLOATS = 4;
  static PF_DEV void fill_table(float*, const pf_params*, int) {}
  PF_DEV void bind(const float*) {}
  DfFastBody b;
  fw_tab_cptr tab;
  PF_DEV void attach(const FwTable* t) { tab = (fw_tab_cptr)(uintptr_t)t; }
  // state groups 0-5: the generic Fixedwing's layout (Fixedwing::load / store)
  PF_DEV void load(const float4* S, size_t n, size_t i, int, float& new_dist, int4& ints) {
    const float4 g0 = S[0 * n + i], g1 = S[1 * n + i], g2 = S[2 * n + i], g3 = S[3 * n + i], g4 = S[4 * n + i], gi = S[5 * n + i];
    b.p = v3{g0.x, g0.y, g0.z}; new_dist = g0.w;
    b.q = quat{g1.x, g1.y, g1.z, g1.w};
    b.v = v3{g2.x, g2.y, g2.z};
    b.w = v3{g2.w, g3.x, g3.y};
    b.act[0] = g3.z; b.act[1] = g3.w; b.act[2] = g4.x; b.act[3] = g4.y; b.act[4] = g4.z; b.thr = g4.w;
    ints = int4{__float_as_int(gi.x), __float_as_int(gi.y), __float_as_int(gi.z), __float_as_int(gi.w)};
    b.contact_now = (ints.y & PF_F_CONTACT) != 0;
    b.contact_step = false;
#pragma unroll
    for (int k = 0; k < 6; ++k) b.cmd[k] = 0.0f;
    b.derive();
    b.rpy = v3{0.0f, 0.0f, 0.0f};
  }
  PF_DEV void store(float4* S, size_t n, size_t i, int, float new_dist, int4 ints) const {
    S[0 * n + i] = float4{b.p.x, b.p.y, b.p.z, new_dist};
    S[1 * n + i] = float4{b.q.x, b.q.y, b.q.z, b.q.w};
    S[2 * n + i] = float4{b.v.x, b.v.y, b.v.z, b.w.x};
    S[3 * n + i] = float4{b.w.y, b.w.z, b.act[0], b.act[1]};
    S[4 * n + i] = float4{b.act[2], b.act[3], b.act[4], b.thr};
    S[5 * n + i] = float4{__int_as_float(ints.x), __int_as_float(ints.y), __int_as_float(ints.z), __int_as_float(ints.w)};
  }
  PF_DEV void reset(const pf_params&, const float* pose, float sp[6], const float* vel) {  // fixedwing.py:194-204
    b.p = v3{pose[0], pose[1], pose[2]};
    b.q = quat{pose[3], pose[4], pose[5], pose[6]};
    b.v = v3{vel[0], vel[1], vel[2]};
    b.w = v3{0.0f, 0.0f, 0.0f};
    b.contact_now = false; b.contact_step = false;
#pragma unroll
    for (int k = 0; k < 5; ++k) b.act[k] = 0.0f;
    b.thr = 0.0f;
#pragma unroll
    for (int k = 0; k < 6; ++k) { sp[k] = 0.0f; b.cmd[k] = 0.0f; }
    b.derive();
    b.rpy = euler_from_quat_fast(b.q);
  }
  template <int MODE_T>
  PF_DEV void control(const pf_params&, const float sp[6]) {  // mode 0, fixedwing.py:143-144,246-250: ids [0,0,1,2,1,3], signs [+,-,+,-,-,+]
    b.cmd[0] = sp[0]; b.cmd[1] = -sp[0]; b.cmd[2] = sp[1]; b.cmd[3] = -sp[2]; b.cmd[4] = -sp[1]; b.cmd[5] = sp[3];
  }
  PF_DEV void tick(const pf_params&, float xi) { b.template tick<true, true>(tab, xi, b.pdev); }
  PF_DEV void share(const float* wpose, float* wvel, float* rec, int tid, int A) { b.wpose_ = wpose; b.wvel_ = wvel; b.prec_ = rec; b.wtid = tid; b.wA = A; }
  PF_DEV void aux(float* o) const {
#pragma unroll
    for (int k = 0; k < 5; ++k) o[k] = b.act[k];
    o[5] = b.thr;
  }
  PF_DEV bool nonfinite() const {
    const float chk = ((b.p.x + b.p.y) + (b.p.z + b.q.x)) + ((b.q.y + b.q.z) + (b.q.w + b.v.x)) + ((b.v.y + b.v.z) + (b.w.x + b.w.y)) +
                      ((b.w.z + b.thr) + (b.act[0] + b.act[1])) + ((b.act[2] + b.act[3]) + b.act[4]);
    return !(__builtin_fabsf(chk) < INFINITY);
  }
};

// sin / cos of an angle in [-pi, pi] (Euler angles): sincos_turns' exact quadrant reduction, abs error < 1e-7
PF_DEV void sincos_angle(float a, float& s, float& c) {
  const float u = a * (0.5f / kPi);
  sincos_turns(u - __builtin_floorf(u), s, c);
}
// tanh(x) = 1 - 2 / (exp(2x) + 1): v_exp_f32 + v_rcp_f32, abs error ~1e-7 (saturates cleanly: exp -> inf gives 1, -> 0 gives -1)
PF_DEV float fast_tanh(float x) { return 1.0f - 2.0f * frcp(__builtin_amdgcn_exp2f(x * 2.885390081777927f) + 1.0f); }

// per-lane exchange record of update_states(): 0-2 w_b, 3-5 rpy, 6-8 v_b, 9-11 body-centre position, 12-14 ground velocity,
// 15 health after this update's hits, 16 inactive, 17 health after the overrides, 18 hit bits of this lane's row (as int),
// 19 unused, 20-27 this lane's row of the masked 1 / (angle + 0.1)
constexpr int kDfRec = 28;

// A: aircraft per world (2 team_size), a template parameter so that the per-row loops and register arrays have their exact size
// and the observation tile (the biggest LDS user: 64 x (23 + 14 (A - 1)) floats) does not limit the workgroups per CU for the
// common 2 v 2 case.
// ROLLOUT (pf_rollout, round 4): k_steps env steps in ONE launch with every aircraft's state resident in registers -- the PettingZoo
// loop of tests/test_pz_envs.py:71-93 without the state's round trip through HBM and without a launch per step. Every step still
// writes its observation / reward / flags, to trajectory buffers [k_steps][n][..]; actions: the given sequence b.actions
// [k_steps][n][AD], or sampled on device with pf_sample_actions' keys (four-wide). Bit-identical to k_steps x (pf_sample_actions +
// pf_env_step): the same code, instantiated with the loop.
template <int A, class VEH, bool ROLLOUT = false>
__global__ void __launch_bounds__(64) dogfight_env_kernel(const pf_params P, const pf_buffers B, const int n, const uint64_t lane0,
                                                          const int op, const uint8_t* mask, const pf_params* __restrict__ Pdev,
                                                          const FwTable* table_g, const int k_steps = 1, const uint32_t step0 = 0u) {
  constexpr int Dmax = 25 + (A - 1) * 14;  // (six-wide actions: two more past-action entries)
  constexpr int kTile = 64 * Dmax > kContactSlotFloats ? 64 * Dmax : kContactSlotFloats;  // (at least one worst-case solver region)
  const int AD = P.df_action_dim == 6 ? 6 : 4;
  const int D = 19 + AD + (A - 1) * 14;
  __shared__ __attribute__((aligned(16))) float tile[kTile];
  __shared__ __attribute__((aligned(16))) float ktab[VEH::TABLE_FLOATS];
  __shared__ float wpose[64 * 8];
  __shared__ float rec[64 * kDfRec];
  const int tid = threadIdx.x;
  VEH::fill_table(ktab, Pdev, tid);
  __syncthreads();
  // a wave holds floor(64 / A) whole worlds (A = 6: ten worlds, four idle lanes): worlds never straddle a wave
  constexpr int T = A / 2;
  constexpr int LPW = (64 / A) * A;
  if (tid >= LPW) return;  // (one wave per workgroup and no barrier below: the idle lanes simply leave)
  const int wave_base = blockIdx.x * LPW;
  const int lane = wave_base + tid;
  const bool valid = lane < n;
  const size_t li = valid ? lane : n - 1;
  const size_t N = (size_t)n;
  const float4* Sin = reinterpret_cast<const float4*>(B.state);
  float4* Sout = reinterpret_cast<float4*>(B.state);
  const int wbase = (tid / A) * A, wlocal = tid - wbase;
  const int my_team = wlocal >= T ? 1 : 0;

  VEH V;
  V.attach(table_g);
  V.b.pdev = Pdev;
  V.b.cws = (lds_fptr)tile;
  V.b.contact_regions(P, kTile);  // as many solver regions as fit the idle observation tile
  // the pair stage's exchange arrays: the velocities in update_states()' exchange records (idle during the ticks), its contact
  // records in the observation tile like the ground solve's (the two run one after the other)
  V.share(wpose, rec, tile, tid, A);
  V.bind(ktab);
  float nd_unused;
  int4 ints;
  V.load(Sin, N, li, 0, nd_unused, ints);
  int step_count = ints.x, flags = ints.y;
  uint32_t rng_ctr = (uint32_t)ints.z;
  const float4 g6 = Sin[6 * N + li];
  float health = g6.x, acc = g6.y;
  int received_hits = __float_as_int(g6.z), df = __float_as_int(g6.w);
  float4 cur_a4 = Sin[7 * N + li], past_a4 = Sin[8 * N + li];
  float4 a45 = Sin[15 * N + li];  // current action 4, 5; past action 4, 5 (six-wide actions)
  float cur_d[A], cur_ang[A];
  {
    const float4 a = Sin[9 * N + li], b = Sin[10 * N + li], c = Sin[11 * N + li], d = Sin[12 * N + li];
    const float rd[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w}, ra[8] = {c.x, c.y, c.z, c.w, d.x, d.y, d.z, d.w};
#pragma unroll
    for (int j = 0; j < A; ++j) { cur_d[j] = rd[j]; cur_ang[j] = ra[j]; }
  }
  float4 sp_a = Sin[13 * N + li], sp_b = Sin[14 * N + li];

  Noise nz;
  nz.mode = P.noise_mode; nz.n = n; nz.lane = (int)li;
  nz.k0 = (uint32_t)P.seed; nz.k1 = (uint32_t)(P.seed >> 32);
  nz.c0 = (uint32_t)(lane0 + li); nz.nmot = (float)P.n_motors; nz.cached = -1; nz.xi = nullptr;

  // (a mask that names some aircraft of a world resets the world)
  const bool do_reset = widen_to_world(op == 1 && valid && ((mask == nullptr) || (mask[li] != 0)), tid, A);
  const bool active = valid && (op == 0 || do_reset);
  float sp[6] = {0, 0, 0, 0, 0, 0};

  // One Aviary.step of the shared world: control + ticks_per_control x (exchange, tick)
  auto world_aviary_step = [&](int flat_base) {
    V.b.contact_step = false;
    V.template control<0>(P, sp);
    // A wreck that has come to rest -- dead, on the ground, in contact, linear and angular velocity small for kDfRestUpdates
    // consecutive updates (DF_AT_REST) -- is not integrated any further (the reference keeps stepping it in Bullet, where it
    // stays where it is; a world of wrecks would otherwise run the contact solve on every lane in every tick). It keeps its
    // resting contact: the collision verdict of :667-670 stays up. A momentary `inactive` does NOT stop the integration.
    // (round 5: a wreck at rest that a moving aircraft comes within reach of is woken -- world_exchange -- and is integrated again,
    //  pair stage included, until it has come to rest anew; wrecks frozen by the opt-in df_freeze_wrecks stay where they are)
    bool wreck = (df & (DF_AT_REST | DF_FROZEN)) != 0;
    for (int t = 0; t < P.ticks_per_control; ++t) {
      world_exchange(V.b, wpose, tid, A, P.bound_radius, Pdev, wreck, rec, (df & DF_FROZEN) != 0);
      if (wreck && V.b.woken) { wreck = false; df &= ~(DF_AT_REST | DF_REST_MASK); }  // (a frozen wreck is never woken)
      if (!wreck) V.tick(P, nz.get(flat_base + t));
    }
    if (wreck) V.b.contact_step = V.b.contact_now;
    if (P.df_freeze_wrecks && V.b.contact_step && V.b.p.z < P.bound_radius + P.contact_break_distance) {  // (opt-in) stops where it hits the ground
      V.b.v = v3{0.f, 0.f, 0.f}; V.b.w = v3{0.f, 0.f, 0.f};
      V.b.derive();
      df |= DF_FROZEN;  // no further integration; update_states() finds it inactive (dead, low, at rest) one update later
    }
    V.b.peer_contact = false;
    V.b.rpy = euler_from_quat_fast(V.b.q);
  };

  // update_states(): _compute_observation (:466-549) + _compute_term_trunc_rew_info (:651-722)
  auto update_states = [&](const bool write_obs) {
    float* me = rec + tid * kDfRec;
    // ---- own attitude, rotation (rz ry rx) and nose direction (ma_fixedwing_base_env.py:336-405)
    float sr, cr, spp, cp, sy, cy;
    sincos_angle(V.b.rpy.x, sr, cr); sincos_angle(V.b.rpy.y, spp, cp); sincos_angle(V.b.rpy.z, sy, cy);
    const float R00 = cy * cp, R01 = cy * spp * sr - sy * cr, R02 = cy * spp * cr + sy * sr;
    const float R10 = sy * cp, R11 = sy * spp * sr + cy * cr, R12 = sy * spp * cr - cy * sr;
    const float R20 = -spp, R21 = cp * sr, R22 = cp * cr;
    const v3 fwd{cy * cp, sy * cp, -spp};
    const v3 pc{fmaf(-0.35f, fwd.x, V.b.p.x), fmaf(-0.35f, fwd.y, V.b.p.y), fmaf(-0.35f, fwd.z, V.b.p.z)};  // :318
    const v3 vb = V.b.vb, wb = V.b.wb;
    const v3 gv{R00 * vb.x + R01 * vb.y + R02 * vb.z, R10 * vb.x + R11 * vb.y + R12 * vb.z, R20 * vb.x + R21 * vb.y + R22 * vb.z};  // :365
    me[0] = wb.x; me[1] = wb.y; me[2] = wb.z; me[3] = V.b.rpy.x; me[4] = V.b.rpy.y; me[5] = V.b.rpy.z;
    me[6] = vb.x; me[7] = vb.y; me[8] = vb.z; me[9] = pc.x; me[10] = pc.y; me[11] = pc.z; me[12] = gv.x; me[13] = gv.y; me[14] = gv.z;
    lds_sync_wave();
    // ---- this lane's row of the engagement matrices (:313-344)
    float prev_d[A], prev_ang[A], iaa[A];
    int hit_bits = 0, inr_bits = 0, chase_bits = 0;
#pragma unroll
    for (int j = 0; j < A; ++j) {
      prev_d[j] = cur_d[j]; prev_ang[j] = cur_ang[j]; iaa[j] = 0.0f;
      if (j < A && j != wlocal) {
        const float* o = rec + (wbase + j) * kDfRec;
        const v3 sep{o[9] - pc.x, o[10] - pc.y, o[11] - pc.z};
        const float dist = fsqrt(dot(sep, sep));
        const v3 cx = cross(sep, fwd);
        const float ang = fast_atan2(fsqrt(dot(cx, cx)), dot(sep, fwd));  // == arccos(sep . fwd / |sep|), well conditioned near 0
        const bool in_range = dist < P.df_lethal_distance, chasing = ang < 0.5f * kPi;
        const bool ff = (j >= T ? 1 : 0) != my_team;
        cur_d[j] = dist; cur_ang[j] = ang;
        if (in_range) inr_bits |= 1 << j;
        if (chasing) chase_bits |= 1 << j;
        if ((ang < P.df_lethal_angle) && in_range && chasing && ff) hit_bits |= 1 << j;
        if (ff && in_range && chasing) iaa[j] = 1.0f / (ang + 0.1f);
      } else if (j < A) {
        cur_d[j] = 0.0f; cur_ang[j] = 0.0f;
      }
    }
    me[18] = __int_as_float(hit_bits);
#pragma unroll
    for (int j = 0; j < A; ++j) me[20 + j] = iaa[j];
    lds_sync_wave();
    // ---- hits received, health (:499-503), inactive (:505-510)
    int rec_hits = 0, team_hits = 0;
    for (int j = 0; j < A; ++j) {
      const int hb = __float_as_int(rec[(wbase + j) * kDfRec + 18]);
      rec_hits += (hb >> wlocal) & 1;
      if ((j >= T ? 1 : 0) == my_team) team_hits += __builtin_popcount(hb);
    }
    received_hits += rec_hits;
    health = __builtin_fmaxf(fmaf(-P.df_damage_per_hit, (float)rec_hits, health), 0.0f);
    const bool inactive = (health <= 0.0f) && (pc.z < 2.0f) && (dot(vb, vb) < 0.01f);
    df = inactive ? (df | DF_INACTIVE) : (df & ~DF_INACTIVE);
    {  // consecutive updates at rest (see DF_AT_REST)
      int rest = (df & DF_REST_MASK) >> DF_REST_SHIFT;
      const bool still = inactive && V.b.contact_now && dot(wb, wb) < 0.01f;
      rest = still ? (rest < 15 ? rest + 1 : 15) : 0;
      df = (df & ~DF_REST_MASK) | (rest << DF_REST_SHIFT);
      if (rest >= kDfRestUpdates) df |= DF_AT_REST;
    }
    me[15] = health; me[16] = inactive ? 1.0f : 0.0f;
    lds_sync_wave();
    // ---- observation (:519-549, pop_obs_by_id :724-752): written on the last update of the call only
    if (write_obs) {
      float* row = tile + tid * D;
      int k = 0;
      row[k++] = wb.x; row[k++] = wb.y; row[k++] = wb.z;
      row[k++] = V.b.rpy.x; row[k++] = V.b.rpy.y; row[k++] = V.b.rpy.z;
      row[k++] = vb.x; row[k++] = vb.y; row[k++] = vb.z;
      row[k++] = pc.x; row[k++] = pc.y; row[k++] = pc.z;
      float aux[6];
      V.aux(aux);
#pragma unroll
      for (int a = 0; a < 6; ++a) row[k++] = aux[a];
      row[k++] = health;
      row[k++] = past_a4.x; row[k++] = past_a4.y; row[k++] = past_a4.z; row[k++] = past_a4.w;
      if (AD == 6) { row[k++] = a45.z; row[k++] = a45.w; }
      for (int j = 0; j < A; ++j) {
        if (j == wlocal) continue;
        const float* o = rec + (wbase + j) * kDfRec;
        if (o[16] != 0.0f) continue;  // inactive aircraft are dropped, the rest moves up (:523-526)
        row[k++] = o[0]; row[k++] = o[1]; row[k++] = o[2];
        row[k++] = o[3] - V.b.rpy.x; row[k++] = o[4] - V.b.rpy.y; row[k++] = o[5] - V.b.rpy.z;  // :357-359
        // the other's ground velocity in the own body frame (gv_j @ R_i) minus the own body velocity (:368-375)
        row[k++] = o[12] * R00 + o[13] * R10 + o[14] * R20 - vb.x;
        row[k++] = o[12] * R01 + o[13] * R11 + o[14] * R21 - vb.y;
        row[k++] = o[12] * R02 + o[13] * R12 + o[14] * R22 - vb.z;
        const v3 sep{o[9] - pc.x, o[10] - pc.y, o[11] - pc.z};  // sep @ R_i (:378)
        row[k++] = sep.x * R00 + sep.y * R10 + sep.z * R20;
        row[k++] = sep.x * R01 + sep.y * R11 + sep.z * R21;
        row[k++] = sep.x * R02 + sep.y * R12 + sep.z * R22;
        row[k++] = o[15];
        row[k++] = ((j >= T ? 1 : 0) == my_team) ? 1.0f : 0.0f;
      }
      for (; k < D; ++k) row[k] = 0.0f;
    }
    // ---- rewards (:551-649)
    float e = 0.0f;
#pragma unroll
    for (int j = 0; j < A; ++j) {
      if (j < A && j != wlocal) {
        const bool ff = (j >= T ? 1 : 0) != my_team;
        const bool in_range = (inr_bits >> j) & 1, chasing = (chase_bits >> j) & 1;
        const float* o = rec + (wbase + j) * kDfRec;
        const int hb_j = __float_as_int(o[18]);
        const float hit_ij = (float)((hit_bits >> j) & 1), hit_ji = (float)((hb_j >> wlocal) & 1);
        if (!P.sparse_reward) {
          const float dd = __builtin_fmaxf(prev_d[j] - cur_d[j], 0.0f);
          if (!in_range && chasing && ff) e = fmaf(4.0f, dd, e);
          float da = (in_range && ff) ? prev_ang[j] - cur_ang[j] : 0.0f;
          da = da < 0.0f ? da * P.df_aggressiveness : da;
          e = fmaf(30.0f, da, e);
          e = fmaf(3.0f, iaa[j] - (1.0f - P.df_aggressiveness) * o[20 + wlocal], e);
        }
        e = fmaf(20.0f, hit_ij - (1.0f - P.df_aggressiveness) * hit_ji, e);
      }
    }
    e = fmaf(P.df_cooperativeness, (float)team_hits, e);  // :609-617
    const float dist_origin = fsqrt(dot(pc, pc));
    float bnd = 0.0f;
    if (!P.sparse_reward) {
      bnd += fast_tanh(fmaf(0.1f, pc.z, -1.0f));
      bnd -= fast_tanh(fmaf(0.0025f, dist_origin, -1.0f));
#pragma unroll
      for (int j = 0; j < A; ++j)
        if (j < A && j != wlocal && cur_d[j] < 5.0f) bnd -= 10.0f * (5.0f - cur_d[j]);
    }
    acc += e + bnd;
    if (step_count > P.max_steps) df |= DF_ACC_TRUNC;
    if (health <= 1e-3f) df |= DF_ACC_TERM | DF_INFO_DEAD;
    if (V.b.contact_step) { df |= DF_ACC_TERM | DF_INFO_COLLISION; acc = -1000.0f; health = 0.0f; }
    if (dist_origin > P.dome) { df |= DF_ACC_TERM | DF_INFO_OOB; acc = -1000.0f; health = 0.0f; }
    me[17] = health;
    lds_sync_wave();
    // ---- team wins (:682-690): element-wise -- member k of a team wins when member k of the other team is out and
    // someone of its own team is still up
    bool any_up = false;
    for (int j = 0; j < A; ++j)
      if ((j >= T ? 1 : 0) == my_team) any_up |= rec[(wbase + j) * kDfRec + 17] > 0.0f;
    const int opp = (1 - my_team) * T + (wlocal - my_team * T);
    if (rec[(wbase + opp) * kDfRec + 17] <= 0.0f && any_up) { df |= DF_ACC_TERM | DF_INFO_TEAM_WIN; acc = 300.0f; }
    lds_sync_wave();
  };

  float out_reward = 0.0f;
  bool out_term = false, out_trunc = false;
  const int KS = ROLLOUT ? k_steps : 1;
  for (int it = 0; it < KS; ++it) {
  const size_t toff = ROLLOUT ? (size_t)it * N : (size_t)0;  // this step's slot in the trajectory buffers (lanes)
  out_reward = 0.0f; out_term = false; out_trunc = false;
  if (op == 1) {
    // ---------------------------------------------------------------- reset (dogfight :215-322, base env :160-234)
    if (do_reset) {
      if (P.df_sample_spawn) {  // _get_start_pos_orn (:176-213): the world's draws, keyed by its first lane
        Noise wz = nz;
        wz.c0 = (uint32_t)(lane0 + li - (size_t)wlocal);
        wz.begin_event(rng_ctr, 2u, nullptr);
        const float u0 = wz.uniform(0, 2u), ur = wz.uniform(1 + wlocal, 2u), uh = wz.uniform(1 + A + wlocal, 2u), uy = wz.uniform(1 + 2 * A + wlocal, 2u);
        const float rad = fmaf(kPi / (float)T, (float)wlocal, 2.0f * kPi * u0);
        const float radius = fmaf(P.df_spawn_max_radius - P.df_spawn_min_radius, ur, P.df_spawn_min_radius);
        const float height = fmaf(P.df_spawn_max_radius - P.df_spawn_min_radius, uh, P.df_spawn_min_radius);  // (sic: the radius bounds, :197-201)
        float s, c;
        sincosf(rad, &s, &c);
        sp_a = float4{radius * c, radius * s, height, 0.0f};
        sp_b = float4{0.0f, fmaf(uy, kPi / 8.0f, rad), 0.0f, 0.0f};
      }
      const v3 rpy0{sp_a.w, sp_b.x, sp_b.y};
      const quat q0 = quat_from_euler(rpy0);
      float sr, cr, spp, cp, sy, cy;
      sincosf(rpy0.x, &sr, &cr); sincosf(rpy0.y, &spp, &cp); sincosf(rpy0.z, &sy, &cy);
      const float pose[7] = {sp_a.x, sp_a.y, sp_a.z, q0.x, q0.y, q0.z, q0.w};
      const float vel[3] = {20.0f * cy * cp, 20.0f * sy * cp, -20.0f * spp};  // :216-222
      V.reset(P, pose, sp, vel);
      step_count = 0; flags = 0;
      health = 1.0f; acc = 0.0f; received_hits = 0; df = DF_ALIVE;  // (clears DF_FROZEN too)
#pragma unroll
      for (int j = 0; j < A; ++j) { cur_d[j] = 0.0f; cur_ang[j] = 0.0f; }
      nz.begin_event(rng_ctr, 1u, B.xi_reset);
    }
    // (whole worlds reset together -- widen_to_world above -- so the lanes that exchange data through LDS always take this
    //  branch together; there is one wave per workgroup and no s_barrier, lanes that are not being reset simply idle)
    if (do_reset) {
      for (int s = 0; s < P.settle_steps; ++s) world_aviary_step(s * P.ticks_per_control);  // ma_fixedwing_base_env.py:232-233
      update_states(true);  // :234; whatever it accumulates is popped by the first step
      rng_ctr += 1;
    }
  } else {
    // ---------------------------------------------------------------- step (ma_fixedwing_base_env.py:272-334)
    float4 a;
    float a4 = 0.0f, a5 = 0.0f;
    if (ROLLOUT && B.actions == nullptr) {  // == sample_actions_kernel(step0 + it): same Philox key, same arithmetic (four-wide)
      const f4 u = uniform4(philox4x32((uint32_t)P.seed, (uint32_t)(P.seed >> 32), (uint32_t)(lane0 + li), step0 + (uint32_t)it, 0u, 3u));
      a = float4{fmaf(P.action_high[0] - P.action_low[0], u.a, P.action_low[0]), fmaf(P.action_high[1] - P.action_low[1], u.b, P.action_low[1]),
                 fmaf(P.action_high[2] - P.action_low[2], u.c, P.action_low[2]), fmaf(P.action_high[3] - P.action_low[3], u.d, P.action_low[3])};
      if (B.actions_out != nullptr && active) reinterpret_cast<float4*>(B.actions_out)[toff + li] = a;
    } else {
      const float* ap = B.actions + (size_t)AD * (toff + li);
      a = float4{ap[0], ap[1], ap[2], ap[3]};
      if (AD == 6) { a4 = ap[4]; a5 = ap[5]; }
    }
    past_a4 = cur_a4;
    cur_a4 = (df & DF_ALIVE) ? a : float4{0.f, 0.f, 0.f, 0.f};  // culled agents: zero commands (:293-297)
    if (AD == 6) a45 = float4{(df & DF_ALIVE) ? a4 : 0.f, (df & DF_ALIVE) ? a5 : 0.f, a45.x, a45.y};
    // :300-301 remaps the LAST action entry; mode 0 (the Aviary's mode whatever assisted_flight says, :229) reads entries 0..3
    sp[0] = cur_a4.x; sp[1] = cur_a4.y; sp[2] = cur_a4.z; sp[3] = AD == 4 ? fmaf(cur_a4.w, 0.5f, 0.5f) : cur_a4.w;
    nz.begin_event(rng_ctr, 0u, B.xi);
    for (int s = 0; s < P.env_step_ratio; ++s) {
      world_aviary_step(s * P.ticks_per_control);
      update_states(s == P.env_step_ratio - 1);
    }
    if (df & DF_ALIVE) {  // pop (:316-330; dogfight :754-771)
      out_reward = acc; acc = 0.0f;
      out_term = (df & DF_ACC_TERM) != 0; out_trunc = (df & DF_ACC_TRUNC) != 0;
      if (out_term || out_trunc) df &= ~DF_ALIVE;
    }
    step_count += 1; rng_ctr += 1;
    if (V.nonfinite() || !(__builtin_fabsf(health + acc) < INFINITY)) flags |= PF_F_NONFINITE;
  }

  // ---------------------------------------------------------------- outputs: obs tile first, state after
  lds_sync_wave();
  {
    const bool wave_all = __all(active || !valid);
    if (wave_all) {
      const int rows = min(LPW, n - wave_base);
      stream_tile(tile, B.obs + (toff + (size_t)wave_base) * D, rows * D, tid, LPW);
    } else if (active) {
      float* g = B.obs + (toff + (size_t)lane) * D;
      const float* row = tile + tid * D;
      for (int k = 0; k < D; ++k) g[k] = row[k];
    }
  }
  if (active && op == 0) {
    B.reward[toff + li] = out_reward;
    B.terminated[toff + li] = out_term ? 1 : 0;
    B.truncated[toff + li] = out_trunc ? 1 : 0;
  }
  if (ROLLOUT) lds_sync_wave();  // (the tile is the next step's solver / exchange scratch)
  }  // for it
  if (active) {
    flags = (flags & ~(PF_F_TERMINATED | PF_F_TRUNCATED | PF_F_CONTACT | PF_F_INFO_COLLISION | PF_F_INFO_OOB)) | (out_term ? PF_F_TERMINATED : 0) |
            (out_trunc ? PF_F_TRUNCATED : 0) | (V.b.contact_now ? PF_F_CONTACT : 0) | ((df & DF_INFO_COLLISION) ? PF_F_INFO_COLLISION : 0) |
            ((df & DF_INFO_OOB) ? PF_F_INFO_OOB : 0);
    V.store(Sout, N, li, 0, 0.0f, int4{step_count, flags, (int)rng_ctr, 0});
    Sout[6 * N + li] = float4{health, acc, __int_as_float(received_hits), __int_as_float(df)};
    Sout[7 * N + li] = cur_a4;
    Sout[8 * N + li] = past_a4;
    float rd[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ra[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < A; ++j) { rd[j] = cur_d[j]; ra[j] = cur_ang[j]; }
    Sout[9 * N + li] = float4{rd[0], rd[1], rd[2], rd[3]};
    Sout[10 * N + li] = float4{rd[4], rd[5], rd[6], rd[7]};
    Sout[11 * N + li] = float4{ra[0], ra[1], ra[2], ra[3]};
    Sout[12 * N + li] = float4{ra[4], ra[5], ra[6], ra[7]};
    Sout[13 * N + li] = sp_a;
    Sout[14 * N + li] = sp_b;
    Sout[15 * N + li] = a45;
  }
}

}  // namespace pf
