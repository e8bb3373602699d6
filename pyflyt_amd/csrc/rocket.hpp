// rocket.hpp -- the Rocket "vehicle program" (drones/rocket.py:222-329) for the Aviary-level kernels:
// four grid fins (LiftingSurface), a booster with first-order throttle, fuel burn and a 2-axis thrust
// gimbal (abstractions/boosters.py:160-263, gimbals.py:151-217), per-axis body drag
// (boring_bodies.py:113-127), on a composite rigid body whose fuel-tank mass and inertia change every
// tick (boosters.py:193-198 -> changeDynamics), so mass, centre of mass and inertia are rebuilt per tick.
// No env uses it on the GPU (Rocket-Landing needs a resting contact): core/aviary.py surface only.
#pragma once
#include "uav_vehicles.hpp"

namespace pf {

struct Rocket {
  // g0 p, fuel_ratio | g1 q | g2 v, w.x | g3 w.yz, fin0, fin1 | g4 fin2, fin3, throttle, ignition
  // g5 gimbal0, gimbal1, -, - | g6 ints
  static constexpr int GROUPS = 7, G_INT = 6, G_TGT = 7, AUX = 9, SP = 7;
  static constexpr int TABLE_FLOATS = 4;  // no LDS constant table
  static PF_DEV void fill_table(float*, const pf_params*, int) {}
  PF_DEV void bind(const float*) {}
  // wind is sampled at the body link (the fuel tank, boring_bodies.py:93-96) and at the four links the
  // fin objects are bound to (lifting_surfaces.py:88-93)
  static constexpr int WIND_LINKS = 5;
  Body b;
  float act[4];
  float thr, fuel, ign;
  float gim[2];
  float cmd[8];

  PF_DEV void load(const float4* S, size_t n, size_t i, int mode, float& new_dist, int4& ints) {
    (void)mode;
    float4 g0 = S[0 * n + i], g1 = S[1 * n + i], g2 = S[2 * n + i], g3 = S[3 * n + i], g4 = S[4 * n + i], g5 = S[5 * n + i];
    float4 gi = S[6 * n + i];
    b.p = v3{g0.x, g0.y, g0.z}; fuel = g0.w; new_dist = 0.0f;
    b.q = quat{g1.x, g1.y, g1.z, g1.w};
    b.v = v3{g2.x, g2.y, g2.z};
    b.w = v3{g2.w, g3.x, g3.y};
    act[0] = g3.z; act[1] = g3.w; act[2] = g4.x; act[3] = g4.y; thr = g4.z; ign = g4.w;
    gim[0] = g5.x; gim[1] = g5.y;
    ints = int4{__float_as_int(gi.x), __float_as_int(gi.y), __float_as_int(gi.z), __float_as_int(gi.w)};
    b.contact_now = (ints.y & PF_F_CONTACT) != 0;
    b.contact_step = false;
    b.derive();
    b.rpy = v3{0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int k = 0; k < 8; ++k) cmd[k] = 0.0f;
  }
  PF_DEV void store(float4* S, size_t n, size_t i, int mode, float new_dist, int4 ints) const {
    (void)mode; (void)new_dist;
    S[0 * n + i] = float4{b.p.x, b.p.y, b.p.z, fuel};
    S[1 * n + i] = float4{b.q.x, b.q.y, b.q.z, b.q.w};
    S[2 * n + i] = float4{b.v.x, b.v.y, b.v.z, b.w.x};
    S[3 * n + i] = float4{b.w.y, b.w.z, act[0], act[1]};
    S[4 * n + i] = float4{act[2], act[3], thr, ign};
    S[5 * n + i] = float4{gim[0], gim[1], 0.0f, 0.0f};
    S[6 * n + i] = float4{__int_as_float(ints.x), __int_as_float(ints.y), __int_as_float(ints.z), __int_as_float(ints.w)};
  }
  PF_DEV void set_mode(int, float*) {}  // base_drone.py:243-259: records the mode, setpoint untouched
  PF_DEV void reset(const pf_params& P, const float* pose, float sp[8], const float* vel = nullptr) {  // rocket.py:222-236
    b.spawn(P, pose, vel);
#pragma unroll
    for (int k = 0; k < 4; ++k) act[k] = 0.0f;
    thr = 0.0f; ign = 0.0f; gim[0] = gim[1] = 0.0f;
    fuel = P.rocket.starting_fuel_ratio;
#pragma unroll
    for (int k = 0; k < 8; ++k) { sp[k] = 0.0f; cmd[k] = 0.0f; }
  }
  template <int MODE_T>
  PF_DEV void control(const pf_params& P, const float sp[8], float = 0.0f, int = 0) {  // rocket.py:249-257
    const pf_rocket& K = P.rocket;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      cmd[i] = clampf(K.finlet_map[i][0] * sp[0] + K.finlet_map[i][1] * sp[1] + K.finlet_map[i][2] * sp[2], -1.0f, 1.0f);
#pragma unroll
    for (int i = 0; i < 4; ++i) cmd[4 + i] = sp[3 + i];
  }
  PF_DEV v3 link_pos(const pf_params& P, int k) const {
    if (k == 0) return b.p + mul(b.R, v3{P.rocket.tank_r[0], P.rocket.tank_r[1], P.rocket.tank_r[2]});
    return b.p + mul(b.R, v3{P.surf[k - 1].r[0], P.surf[k - 1].r[1], P.surf[k - 1].r[2]});
  }
  // update_physics (rocket.py:268-290) + stepSimulation + update_state for one tick
  PF_DEV void tick(const pf_params& P, float xi, const float* wind = nullptr) {
    const pf_rocket& K = P.rocket;
    v3 F{0.0f, 0.0f, 0.0f}, tau{0.0f, 0.0f, 0.0f};
    {  // body drag at the fuel tank link (rocket.py:88-112,271)
      const v3 rt{K.tank_r[0], K.tank_r[1], K.tank_r[2]};
      v3 vd = b.vb + cross(b.wb, rt);
      if (wind) vd = vd - mulT(b.R, v3{wind[0], wind[1], wind[2]});
      v3 f{-P.drag_const[0] * sq_signed(vd.x), -P.drag_const[1] * sq_signed(vd.y), -P.drag_const[2] * sq_signed(vd.z)};
      F = F + f;
      tau = tau + cross(rt, f);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {  // grid fins (rocket.py:274); unrolled: static indices into the parameter block
      const pf_surface& S = P.surf[i];
      const float ai = fmaf(S.dt_over_tau, cmd[i] - act[i], act[i]);
      act[i] = ai;
      v3 r{S.r[0], S.r[1], S.r[2]};
      v3 vloc = b.vb + cross(b.wb, r);
      if (wind) vloc = vloc - mulT(b.R, v3{wind[3 * (i + 1) + 0], wind[3 * (i + 1) + 1], wind[3 * (i + 1) + 2]});
      v3 f, t;
      lifting_surface(S, vloc, ai, f, t);
      F = F + f;
      tau = tau + cross(r, f) + t;
    }
    // gimbal (gimbals.py:166-217): lag, then R_x(a1) R_y(a2) applied to the +z thrust unit
    gim[0] = fmaf(K.gimbal_dt_over_tau, cmd[6] - gim[0], gim[0]);
    gim[1] = fmaf(K.gimbal_dt_over_tau, cmd[7] - gim[1], gim[1]);
    float s1, c1, s2, c2;
    sincosf(gim[0] * K.gimbal_range_rad, &s1, &c1);
    sincosf(gim[1] * K.gimbal_range_rad, &s2, &c2);
    const v3 dir{s2, -s1 * c2, c1 * c2};
    // booster (boosters.py:213-263)
    ign = ((K.reignitable == 0 && ign != 0.0f) || cmd[4] > 0.5f) ? 1.0f : 0.0f;
    const float target = ign * fmaf(cmd[5], 1.0f - K.thrust_min_ratio, K.thrust_min_ratio);
    float t = fmaf(K.booster_dt_over_tau, target - thr, thr);
    t = fmaf(xi * t, K.booster_noise, t);
    t = fuel > 0.0f ? t : 0.0f;
    thr = t;
    fuel = clampf(fuel - t * K.fuel_rate_ratio * P.dt, 0.0f, 1.0f);
    {
      const v3 f = (t * K.max_thrust) * dir;
      const v3 rb{K.booster_r[0], K.booster_r[1], K.booster_r[2]};
      F = F + f;
      tau = tau + cross(rb, f);
    }
    integrate(P, F, tau);
  }
  PF_DEV void tick_unarmed(const pf_params& P) { integrate(P, v3{0.f, 0.f, 0.f}, v3{0.f, 0.f, 0.f}); }
  // stepSimulation on the composite body with the current fuel mass / inertia (fuel tank at tank_r, diagonal inertia)
  PF_DEV void integrate(const pf_params& P, v3 F, v3 tau) {
    const pf_rocket& K = P.rocket;
    const float mf = fuel * K.total_fuel;
    const float M = K.dry_mass + mf, iM = 1.0f / M;
    const v3 c{(K.dry_mr[0] + mf * K.tank_r[0]) * iM, (K.dry_mr[1] + mf * K.tank_r[1]) * iM, (K.dry_mr[2] + mf * K.tank_r[2]) * iM};
    const float rr = K.tank_r[0] * K.tank_r[0] + K.tank_r[1] * K.tank_r[1] + K.tank_r[2] * K.tank_r[2];
    const float cc = dot(c, c);
    float Ipa[6] = {K.dry_S[0] + mf * (rr - K.tank_r[0] * K.tank_r[0]) - M * (cc - c.x * c.x),
                    K.dry_S[1] - mf * K.tank_r[0] * K.tank_r[1] + M * c.x * c.y,
                    K.dry_S[2] - mf * K.tank_r[0] * K.tank_r[2] + M * c.x * c.z,
                    K.dry_S[3] + mf * (rr - K.tank_r[1] * K.tank_r[1]) - M * (cc - c.y * c.y),
                    K.dry_S[4] - mf * K.tank_r[1] * K.tank_r[2] + M * c.y * c.z,
                    K.dry_S[5] + mf * (rr - K.tank_r[2] * K.tank_r[2]) - M * (cc - c.z * c.z)};
    const float Io[3] = {fmaf(fuel, K.fuel_inertia[0], K.dry_I[0]), fmaf(fuel, K.fuel_inertia[1], K.dry_I[1]),
                         fmaf(fuel, K.fuel_inertia[2], K.dry_I[2])};
    // total inertia and its inverse (symmetric 3x3 by cofactors)
    const float a = Ipa[0] + Io[0], bb = Ipa[1], cx = Ipa[2], d = Ipa[3] + Io[1], e = Ipa[4], f = Ipa[5] + Io[2];
    const float C00 = d * f - e * e, C01 = cx * e - bb * f, C02 = bb * e - cx * d;
    const float C11 = a * f - cx * cx, C12 = bb * cx - a * e, C22 = a * d - bb * bb;
    const float idet = 1.0f / (a * C00 + bb * C01 + cx * C02);
    const float Iinv[6] = {C00 * idet, C01 * idet, C02 * idet, C11 * idet, C12 * idet, C22 * idet};
    float H[6] = {Ipa[0], Ipa[1], Ipa[2], Ipa[3], Ipa[4], Ipa[5]};  // gyroscopic inertia: own part gated by the Bullet flag
    if (P.use_gyro_term) { H[0] += Io[0]; H[3] += Io[1]; H[5] += Io[2]; }
    b.tick_var(P, F, tau, iM, c, H, Iinv);
  }
  template <int MODE_T>
  PF_DEV void aviary_step(const pf_params& P, const float sp[8], Noise& nz, int flat_base) {
    b.contact_step = false;
    control<MODE_T>(P, sp);
    for (int t = 0; t < P.ticks_per_control; ++t) tick(P, nz.get(flat_base + t));
    b.rpy = euler_from_quat_fast(b.q);
  }
  PF_DEV void aux(float* o) const {  // rocket.py:320-326: fins, (ignition, fuel, throttle), gimbal
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] = act[k];
    o[4] = ign; o[5] = fuel; o[6] = thr; o[7] = gim[0]; o[8] = gim[1];
  }
  PF_DEV float4 get_cmd() const { return float4{0.f, 0.f, 0.f, 0.f}; }  // stateless mixing
  PF_DEV void set_cmd(float4) {}
};

}  // namespace pf
