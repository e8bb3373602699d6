// quadx_control_d.hpp -- the cascaded flight modes' controller (quadx.py:401-493, modes 1 .. 7) in DOUBLE precision.
//
// Why (round 5): the outer loops differentiate what they read -- lin_vel k_d / T = 60, z_vel k_d / T = 6 per control tick -- and pass
// the result down three more PIDs to the motors, so a relative rounding of 6e-8 in a body-frame velocity or an Euler angle is worth
// 1e-5 .. 1e-4 in the angular velocity one env step later, and 1e-3 over an episode (tests/tools/fp32_rounding_sites.py: with the
// controllers' inputs ALONE rounded to float32 the fp64 oracle replays the mode-7 fixture 7.0e-4 away from itself, with the PIDs'
// internals alone 1.5e-4; the rigid-body state, which stays float32, accounts for 2.2e-4). The state derivation that feeds the PIDs
// (rotation matrix, R^T v, R^T w, Euler angles: quadx.py:512-535) and the PIDs themselves therefore run in fp64 here, from the
// float32 state; the PID memories are stored as float32 (the state groups' format). FP64 vector instructions issue at the float32
// rate on gfx950 and these modes are on no benchmark's critical path: the mode-0 instantiations do not contain this code.
// Used by the generic vehicle (uav_vehicles.hpp: QuadX::control) and by the specialised kernel's MODES instantiations
// (quadx_fast.hpp), so the two agree with each other bit for bit in these modes.
#pragma once
#include "uav_device.hpp"

namespace pf {

struct QuadCtlIn {  // update_state's outputs (quadx.py:512-535), fp64 from the float32 state
  double wb[3], vb[3], rpy[3], p[3];
  double cyaw, syaw;  // cos / sin of rpy[2] (modes 6, 7 rotate the setpoint by the yaw: quadx.py:448-451,460-463)
};
// Double-precision division, square root and atan2 at a fraction of the library routines' cost. The library forms are IEEE-exact with
// every special case (division ~30 instructions, sqrt ~30, atan2 ~300, asin ~200, sin / cos ~270 each -- 1 100 of a control update's
// 1 600 in round 6's first build); what the cascade needs is "far inside float32's rounding", not the last bit: these are good to a few
// units in 1e-16 on the ranges the controller feeds them (no infinities, no denormals).
PF_DEV double rcp_d(const double x) {  // v_rcp_f64 (~2^-23) + two Newton steps
  double r = __builtin_amdgcn_rcp(x);
  r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
  return __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
}
PF_DEV double sqrt_pos_d(const double x) {  // x > 0: v_rsq_f64 + the coupled Newton iteration for (sqrt x, 1 / (2 sqrt x))
  const double r = __builtin_amdgcn_rsq(x);
  double g = x * r, h = 0.5 * r;
  double e = __builtin_fma(-h, g, 0.5);
  g = __builtin_fma(g, e, g); h = __builtin_fma(h, e, h);
  e = __builtin_fma(-h, g, 0.5);
  g = __builtin_fma(g, e, g); h = __builtin_fma(h, e, h);
  return __builtin_fma(__builtin_fma(-g, g, x), h, g);
}
// atan2(y, x), with cos and sin of the result for free. The octant angle phi = atan(min / max) in [0, pi/4] is first estimated in
// float32 (uav_device.hpp's polynomial, 1e-7), then corrected by what the estimate misses: with (c, s) = (cos, sin)(phi0) from the
// fdlibm kernels on [-pi/4, pi/4] (no argument reduction), tan(phi - phi0) = (b c - a s) / (a c + b s) =: d exactly, and
// phi = phi0 + d - d^3/3 + ... with |d| < 1e-6: the cubic term is below 1e-18. Octant fix-ups as in fast_atan2. x = y = 0 -> 0.
PF_DEV double atan2_d(const double y, const double x, double* cs = nullptr, double* sn = nullptr) {
  const double ax = __builtin_fabs(x), ay = __builtin_fabs(y);
  const double a = ax > ay ? ax : ay, b = ax > ay ? ay : ax;
  const float af = (float)a, bf = (float)b;
  const float t = (af == 0.0f) ? 0.0f : bf * frcp(af);
  const float u = t * t;
  float pl = fmaf(u, 0.0029035410843789577f, -0.016282962635159492f);
  pl = fmaf(u, pl, 0.04303929582238197f);
  pl = fmaf(u, pl, -0.07533670216798782f);
  pl = fmaf(u, pl, 0.10654674470424652f);
  pl = fmaf(u, pl, -0.14207133650779724f);
  pl = fmaf(u, pl, 0.19993053376674652f);
  pl = fmaf(u, pl, -0.3333309292793274f);
  pl = fmaf(u, pl, 1.0f);
  const double p0 = (double)(pl * t), z = p0 * p0;
  // __kernel_sin / __kernel_cos (fdlibm), |p0| <= pi/4
  double ps = __builtin_fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
  double pc = __builtin_fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
  ps = __builtin_fma(z, ps, 2.75573137070700676789e-06);
  pc = __builtin_fma(z, pc, -2.75573143513906633035e-07);
  ps = __builtin_fma(z, ps, -1.98412698298579493134e-04);
  pc = __builtin_fma(z, pc, 2.48015872894767294178e-05);
  ps = __builtin_fma(z, ps, 8.33333333332248946124e-03);
  pc = __builtin_fma(z, pc, -1.38888888888741095749e-03);
  ps = __builtin_fma(z, ps, -1.66666666666666324348e-01);
  pc = __builtin_fma(z, pc, 4.16666666666666019037e-02);
  const double s = __builtin_fma(p0 * z, ps, p0);
  const double c = __builtin_fma(z * z, pc, __builtin_fma(z, -0.5, 1.0));
  const double den = __builtin_fma(a, c, b * s), num = __builtin_fma(b, c, -(a * s));
  const bool zero = !(den > 0.0);
  const double rden = rcp_d(zero ? 1.0 : den);
  double phi = p0 + num * rden;
  if (cs != nullptr) {  // (compile-time after inlining) the unit vector of (x, y): x / hypot = x / (den (1 + d^2/2 ...)), d^2 < 1e-12
    *cs = zero ? 1.0 : x * rden; *sn = zero ? 0.0 : y * rden;
  }
  phi = ay > ax ? 0.5 * 3.14159265358979323846 - phi : phi;
  phi = x < 0.0 ? 3.14159265358979323846 - phi : phi;
  return zero ? 0.0 : __builtin_copysign(phi, y);
}
// getEulerFromQuaternion (ZYX, gimbal-lock branch at |sarg| >= 0.99999) with cos / sin of the yaw; rd = 1 / |q|^2
PF_DEV void quad_ctl_euler_d(const double q4[4], const double rd, QuadCtlIn& o) {
  const double x = q4[0], y = q4[1], z = q4[2], w = q4[3];
  const double sqx = x * x, sqy = y * y, sqz = z * z, squ = w * w;
  const double sarg = -2.0 * (x * z - w * y) * rd;
  if (sarg <= -0.99999 || sarg >= 0.99999) {  // gimbal lock, rare: the library's functions
    o.rpy[0] = 0.0; o.rpy[1] = sarg < 0.0 ? -0.5 * 3.14159265358979323846 : 0.5 * 3.14159265358979323846;
    o.rpy[2] = sarg < 0.0 ? 2.0 * atan2(x, -y) : 2.0 * atan2(-x, y);
    o.cyaw = cos(o.rpy[2]); o.syaw = sin(o.rpy[2]);
  } else {
    o.rpy[0] = atan2_d(2.0 * (y * z + w * x), squ - sqx - sqy + sqz);
    o.rpy[1] = atan2_d(sarg, sqrt_pos_d((1.0 - sarg) * (1.0 + sarg)));  // asin(sarg), |sarg| < 0.99999
    o.rpy[2] = atan2_d(2.0 * (x * y + w * z), squ + sqx - sqy - sqz, &o.cyaw, &o.syaw);
  }
}
// (the double-precision core: the specialised kernel's cascaded-mode instantiations carry the rigid-body state itself in fp64 since
//  round 6 and call this directly -- quadx_fast.hpp: QuadStateD)
PF_DEV QuadCtlIn quad_ctl_inputs_d(const double q4[4], const double v3d[3], const double w3d[3], const double p3d[3]) {
  const double x = q4[0], y = q4[1], z = q4[2], w = q4[3];
  // btMatrix3x3::setRotation
  const double d = x * x + y * y + z * z + w * w, rd = rcp_d(d), s = 2.0 * rd;
  const double xs = x * s, ys = y * s, zs = z * s;
  const double wx = w * xs, wy = w * ys, wz = w * zs, xx = x * xs, xy = x * ys, xz = x * zs, yy = y * ys, yz = y * zs, zz = z * zs;
  const double R00 = 1.0 - (yy + zz), R01 = xy - wz, R02 = xz + wy, R10 = xy + wz, R11 = 1.0 - (xx + zz), R12 = yz - wx,
               R20 = xz - wy, R21 = yz + wx, R22 = 1.0 - (xx + yy);
  QuadCtlIn o;
  const double v0 = v3d[0], v1 = v3d[1], v2 = v3d[2], w0 = w3d[0], w1 = w3d[1], w2 = w3d[2];
  o.vb[0] = R00 * v0 + R10 * v1 + R20 * v2; o.vb[1] = R01 * v0 + R11 * v1 + R21 * v2; o.vb[2] = R02 * v0 + R12 * v1 + R22 * v2;
  o.wb[0] = R00 * w0 + R10 * w1 + R20 * w2; o.wb[1] = R01 * w0 + R11 * w1 + R21 * w2; o.wb[2] = R02 * w0 + R12 * w1 + R22 * w2;
  quad_ctl_euler_d(q4, rd, o);
  o.p[0] = p3d[0]; o.p[1] = p3d[1]; o.p[2] = p3d[2];
  return o;
}
PF_DEV QuadCtlIn quad_ctl_inputs(const quat qf, const v3 vf, const v3 wf, const v3 pf_) {
  const double q4[4] = {qf.x, qf.y, qf.z, qf.w}, v3d[3] = {vf.x, vf.y, vf.z}, w3d[3] = {wf.x, wf.y, wf.z}, p3d[3] = {pf_.x, pf_.y, pf_.z};
  return quad_ctl_inputs_d(q4, v3d, w3d, p3d);
}
PF_DEV double clampd(double x, double lo, double hi) { return x < lo ? lo : (x > hi ? hi : x); }
// abstractions/pid.py:70-94 for one component. MT: the memories' type -- float (the state groups' words: the generic vehicle, shared
// worlds) or double (the specialised kernel's cascaded-mode instantiations since round 6: a stored error rounded to float32 comes back
// through the derivative term, k_d / T = 60, in the very next control update -- with the PIDs' internals ALONE in float32 the fp64
// oracle replays the mode-7 fixture 1.5e-4 away from itself, with everything else the device still rounds 6.6e-6)
template <class MT>
PF_DEV double pid1d(const float kp, const float ki, const float kd, const float lim, const double T, const double iT, MT& I, MT& E, const double st, const double sp) {
  const double e = sp - st;
  const double l = lim;
  const double In = clampd((double)I + (double)ki * e * T, -l, l);
  const double der = (double)kd * (e - (double)E) * iT;  // (iT = 1 / T: the reference divides, the same to 1e-16)
  I = (MT)In;
  E = (MT)e;
  return clampd((double)kp * e + In + der, -l, l);
}
// The memories of one vehicle: pointers into the owner's members (everything is force-inlined: they stay in registers)
template <class MT>
struct QuadMemT {
  MT *I0, *E0;  // ang_vel  [3]
  MT *I1, *E1;  // ang_pos  [3]
  MT *I2, *E2;  // lin_vel  [2]
  MT *I3, *E3;  // lin_pos  [2]
  MT *zI, *zE;  // z_vel, z_pos
};
typedef QuadMemT<float> QuadMemD;
// update_control for mode 1 .. 7: setpoint -> the four motor commands. PP: (pointer to) the parameter block (a plain reference's
// address or the scalar-cache pointer of the specialised kernels). T: the control period.
template <class PP, class MT, class OT = float>
PF_DEV void quad_cascade_d(const PP P, const int mode, const double T, const QuadCtlIn& in, const QuadMemT<MT> M, const float sp[4], OT pwm[4]) {
  double a[3] = {sp[0], sp[1], sp[2]};
  double z = sp[3];
  const double iT = rcp_d(T);
  // (always_inline: left out of line -- the diagnostic build's generic kernels did that -- the captures become flat pointers to the
  //  caller's stack, and ROCm 7.2's instruction selection aborts on the private-aperture test that goes with them)
  auto pidn = [&](const int k, MT* I, MT* E, const double* st, const int n) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
      if (i < n) a[i] = pid1d(P->pid[k].kp[i], P->pid[k].ki[i], P->pid[k].kd[i], P->pid[k].lim[i], T, iT, I[i], E[i], st[i], a[i]);
  };
  if (mode == 2) {
    pidn(0, M.I0, M.E0, in.wb, 3);
  } else if (mode == 1 || mode == 3) {
    pidn(1, M.I1, M.E1, in.rpy, 3);
    pidn(0, M.I0, M.E0, in.wb, 3);
  } else {
    if (mode == 7) pidn(3, M.I3, M.E3, in.p, 2);
    if (mode == 6 || mode == 7) {  // quadx.py:448-451,460-463
      const double c = in.cyaw, s = in.syaw;
      const double a0 = c * a[0] + s * a[1], a1 = -s * a[0] + c * a[1];
      a[0] = a0; a[1] = a1;
    }
    pidn(2, M.I2, M.E2, in.vb, 2);
    { const double t0 = -a[1], t1 = a[0]; a[0] = t0; a[1] = t1; }
    pidn(1, M.I1, M.E1, in.rpy, mode == 7 ? 3 : 2);
    pidn(0, M.I0, M.E0, in.wb, 3);
  }
  if (!(mode == 1 || mode == 5 || mode == 6))
    z = pid1d(P->zpid[1].kp[0], P->zpid[1].ki[0], P->zpid[1].kd[0], P->zpid[1].lim[0], T, iT, M.zI[1], M.zE[1], in.p[2], z);
  z = pid1d(P->zpid[0].kp[0], P->zpid[0].ki[0], P->zpid[0].kd[0], P->zpid[0].lim[0], T, iT, M.zI[0], M.zE[0], in.vb[2], z);
  z = clampd(z, 0.0, 1.0);
  // mixing + saturation handling (quadx.py:482-493)
  const double cmd[4] = {a[0], a[1], a[2], z};
  double pw[4];
  double hi = -INFINITY, lo = INFINITY;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const double s = (double)P->motor_map[i][0] * cmd[0] + (double)P->motor_map[i][1] * cmd[1] + (double)P->motor_map[i][2] * cmd[2] + (double)P->motor_map[i][3] * cmd[3];
    pw[i] = s;
    hi = s > hi ? s : hi;
    lo = s < lo ? s : lo;
  }
  if (hi != lo) {
    const double pmax = hi < 1.0 ? hi : 1.0, pmin = lo > 0.05 ? lo : 0.05;
    const double ka = (pmin - lo) * rcp_d(pmax - lo), ks = (hi - pmax) * rcp_d(hi - pmin);
#pragma unroll
    for (int i = 0; i < 4; ++i) pw[i] += ka * (pmax - pw[i]) - ks * (pw[i] - pmin);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) pwm[i] = (OT)clampd(pw[i], 0.05, 1.0);
}

}  // namespace pf
