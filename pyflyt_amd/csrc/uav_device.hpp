// uav_device.hpp -- device-side math for the batched UAV step (gfx950 / CDNA4, fp32).
// One wavefront lane owns one drone; everything here is per-lane register arithmetic.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define PF_DEV __device__ __forceinline__

namespace pf {

constexpr float kPi = 3.14159265358979323846f;

struct v3 {
  float x, y, z;
};
PF_DEV v3 mk3(float x, float y, float z) { return v3{x, y, z}; }
PF_DEV v3 operator+(v3 a, v3 b) { return v3{a.x + b.x, a.y + b.y, a.z + b.z}; }
PF_DEV v3 operator-(v3 a, v3 b) { return v3{a.x - b.x, a.y - b.y, a.z - b.z}; }
PF_DEV v3 operator*(float s, v3 a) { return v3{s * a.x, s * a.y, s * a.z}; }
PF_DEV float dot(v3 a, v3 b) { return fmaf(a.x, b.x, fmaf(a.y, b.y, a.z * b.z)); }
PF_DEV v3 cross(v3 a, v3 b) {
  return v3{fmaf(a.y, b.z, -(a.z * b.y)), fmaf(a.z, b.x, -(a.x * b.z)), fmaf(a.x, b.y, -(a.y * b.x))};
}
PF_DEV float clampf(float x, float lo, float hi) { return __builtin_fminf(__builtin_fmaxf(x, lo), hi); }
// x*|x| == sign(x)*x^2 (np.sign(x) * x**2 in the reference)
PF_DEV float sq_signed(float x) { return x * __builtin_fabsf(x); }

struct quat {
  float x, y, z, w;
};
// body->world rotation, row major (btMatrix3x3::setRotation; quadx.py:521 uses its transpose)
struct m3 {
  float m00, m01, m02, m10, m11, m12, m20, m21, m22;
};
PF_DEV m3 rot_from_quat(quat q) {
  float d = fmaf(q.x, q.x, fmaf(q.y, q.y, fmaf(q.z, q.z, q.w * q.w)));
  float s = 2.0f / d;
  float xs = q.x * s, ys = q.y * s, zs = q.z * s;
  float wx = q.w * xs, wy = q.w * ys, wz = q.w * zs;
  float xx = q.x * xs, xy = q.x * ys, xz = q.x * zs;
  float yy = q.y * ys, yz = q.y * zs, zz = q.z * zs;
  return m3{1.0f - (yy + zz), xy - wz, xz + wy, xy + wz, 1.0f - (xx + zz), yz - wx, xz - wy, yz + wx, 1.0f - (xx + yy)};
}
PF_DEV v3 mul(const m3& R, v3 a) {  // R a : body -> world
  return v3{fmaf(R.m00, a.x, fmaf(R.m01, a.y, R.m02 * a.z)), fmaf(R.m10, a.x, fmaf(R.m11, a.y, R.m12 * a.z)),
            fmaf(R.m20, a.x, fmaf(R.m21, a.y, R.m22 * a.z))};
}
PF_DEV v3 mulT(const m3& R, v3 a) {  // R^T a : world -> body
  return v3{fmaf(R.m00, a.x, fmaf(R.m10, a.y, R.m20 * a.z)), fmaf(R.m01, a.x, fmaf(R.m11, a.y, R.m21 * a.z)),
            fmaf(R.m02, a.x, fmaf(R.m12, a.y, R.m22 * a.z))};
}
// symmetric 3x3 stored xx xy xz yy yz zz
PF_DEV v3 symmul(const float S[6], v3 a) {
  return v3{fmaf(S[0], a.x, fmaf(S[1], a.y, S[2] * a.z)), fmaf(S[1], a.x, fmaf(S[3], a.y, S[4] * a.z)),
            fmaf(S[2], a.x, fmaf(S[4], a.y, S[5] * a.z))};
}

// pybullet getEulerFromQuaternion (ZYX, gimbal-lock branch at |sarg| >= 0.99999) -- quadx.py:526
PF_DEV v3 euler_from_quat(quat q) {
  float sqx = q.x * q.x, sqy = q.y * q.y, sqz = q.z * q.z, squ = q.w * q.w;
  float sarg = -2.0f * (q.x * q.z - q.w * q.y) / (sqx + sqy + sqz + squ);
  v3 e;
  if (sarg <= -0.99999f) {
    e = v3{0.0f, -0.5f * kPi, 2.0f * atan2f(q.x, -q.y)};
  } else if (sarg >= 0.99999f) {
    e = v3{0.0f, 0.5f * kPi, 2.0f * atan2f(-q.x, q.y)};
  } else {
    e.x = atan2f(2.0f * (q.y * q.z + q.w * q.x), squ - sqx - sqy + sqz);
    e.y = asinf(sarg);
    e.z = atan2f(2.0f * (q.x * q.y + q.w * q.z), squ + sqx - sqy - sqz);
  }
  return e;
}
// pybullet getQuaternionFromEuler -- quadx_base_env.py:243
PF_DEV quat quat_from_euler(v3 e) {
  float sr, cr, sp, cp, sy, cy;
  sincosf(0.5f * e.x, &sr, &cr);
  sincosf(0.5f * e.y, &sp, &cp);
  sincosf(0.5f * e.z, &sy, &cy);
  quat q{sr * cp * cy - cr * sp * sy, cr * sp * cy + sr * cp * sy, cr * cp * sy - sr * sp * cy, cr * cp * cy + sr * sp * sy};
  float inv = rsqrtf(fmaf(q.x, q.x, fmaf(q.y, q.y, fmaf(q.z, q.z, q.w * q.w))));
  return quat{q.x * inv, q.y * inv, q.z * inv, q.w * inv};
}

// ---------------------------------------------------------------- lean math for the hot kernels
// Packed fp32 (gfx940+: v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32): two IEEE single operations per issue slot, each
// element rounded exactly as the scalar instruction would. A packed result read by the very next VALU instruction costs
// a wait state, so callers interleave independent work (the build runs with the machine scheduler off: statement order
// is the schedule). A splat operand (sp2) is an op_sel on any VGPR / SGPR, not an instruction.
typedef float f2 __attribute__((ext_vector_type(2)));
PF_DEV f2 fma2(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
PF_DEV f2 sp2(float x) { return f2{x, x}; }
PF_DEV float frcp(float x) { return __builtin_amdgcn_rcpf(x); }   // v_rcp_f32, 1 ulp
PF_DEV float frsq(float x) { return __builtin_amdgcn_rsqf(x); }   // v_rsq_f32, 1 ulp
PF_DEV float fsqrt(float x) { return __builtin_amdgcn_sqrtf(x); } // v_sqrt_f32, 1 ulp
PF_DEV float med3(float x, float lo, float hi) { return __builtin_amdgcn_fmed3f(x, lo, hi); }
// atan2 with a degree-8 minimax polynomial in t^2 on [0,1] (max abs error 1.2e-7 rad in fp32):
// a handful of FMAs instead of the library routine. atan2(0,0) = 0.
PF_DEV float fast_atan2(float y, float x) {
  float ax = __builtin_fabsf(x), ay = __builtin_fabsf(y);
  float mx = __builtin_fmaxf(ax, ay), mn = __builtin_fminf(ax, ay);
  float t = mn * frcp(mx);
  t = (mx == 0.0f) ? 0.0f : t;
  float s = t * t;
  float p = fmaf(s, 0.0029035410843789577f, -0.016282962635159492f);
  p = fmaf(s, p, 0.04303929582238197f);
  p = fmaf(s, p, -0.07533670216798782f);
  p = fmaf(s, p, 0.10654674470424652f);
  p = fmaf(s, p, -0.14207133650779724f);
  p = fmaf(s, p, 0.19993053376674652f);
  p = fmaf(s, p, -0.3333309292793274f);
  p = fmaf(s, p, 1.0f);
  float r = p * t;
  r = (ay > ax) ? (0.5f * kPi - r) : r;
  r = (x < 0.0f) ? (kPi - r) : r;
  return __builtin_copysignf(r, y);
}
PF_DEV float fast_asin(float x) {  // asin(x) = atan2(x, sqrt((1-x)(1+x)))
  return fast_atan2(x, fsqrt(__builtin_fmaxf((1.0f - x) * (1.0f + x), 0.0f)));
}
// getQuaternionFromEuler's product (x, y, z, w) from the half-angle cosines / sines, shared factors hoisted
PF_DEV quat quat_from_half_angles(float cr, float sr, float cp, float sp, float cy, float sy) {
  const float a = sr * cp, b = cr * sp, c = cr * cp, d = sr * sp;
  return quat{fmaf(a, cy, -(b * sy)), fmaf(b, cy, a * sy), fmaf(c, sy, -(d * cy)), fmaf(c, cy, d * sy)};
}
// cos(t/2), sin(t/2) of an angle t in (-pi, pi] given (cos t, sin t): no trig, no cancellation
PF_DEV void half_angle(float c, float s, float& ch, float& sh) {
  // a = sqrt((1+|c|)/2) is cos(t/2) when c >= 0 and |sin(t/2)| otherwise; the partner follows from
  // sin t = 2 sin(t/2) cos(t/2). Branch-free (selects), so it stays in registers.
  const float a = fsqrt(0.5f * (1.0f + __builtin_fabsf(c)));
  const float b = 0.5f * s * frcp(a);
  const bool pos = c >= 0.0f;
  ch = pos ? a : __builtin_fabsf(b);
  sh = pos ? b : __builtin_copysignf(a, s);
}

// sin/cos of a small angle (|x| <= 1: error < 2e-8) by Taylor polynomials, no range reduction
PF_DEV void sincos_small(float x, float& sn, float& cs) {
  const float t = x * x;
  sn = x * fmaf(t, fmaf(t, fmaf(t, fmaf(t, fmaf(t, -2.5052108e-8f, 2.7557319e-6f), -1.9841270e-4f), 8.3333333e-3f), -1.6666667e-1f), 1.0f);
  cs = fmaf(t, fmaf(t, fmaf(t, fmaf(t, fmaf(t, fmaf(t, 2.0876757e-9f, -2.7557319e-7f), 2.4801587e-5f), -1.3888889e-3f), 4.1666667e-2f), -0.5f), 1.0f);
}
// sin/cos of 2*pi*u for u in [0, 1] (an angle given in turns): quadrant reduction is exact in fp32
// (u - k/4 with k = rint(4u)), the residual |r| <= pi/4 goes through the Taylor pair. ~30 instructions
// against ~300 for two library sincosf calls; abs error < 1e-7.
PF_DEV void sincos_turns(float u, float& sn, float& cs) {
  const float k = __builtin_rintf(4.0f * u);
  const float r = (2.0f * kPi) * fmaf(k, -0.25f, u);
  float s, c;
  sincos_small(r, s, c);
  const int q = (int)k & 3;
  const bool odd = (q & 1) != 0;
  const float a = odd ? c : s, b = odd ? s : c;  // q=0: (s, c)  1: (c, -s)  2: (-s, -c)  3: (-c, s)
  sn = (q & 2) ? -a : a;
  cs = (q == 1 || q == 2) ? -b : b;
}
// getEulerFromQuaternion without libm (polynomial atan2/asin, 1.2e-7 rad); same branches
PF_DEV v3 euler_from_quat_fast(quat q) {
  float sqx = q.x * q.x, sqy = q.y * q.y, sqz = q.z * q.z, squ = q.w * q.w;
  float sarg = -2.0f * (q.x * q.z - q.w * q.y) * frcp(sqx + sqy + sqz + squ);
  if (sarg <= -0.99999f) return v3{0.0f, -0.5f * kPi, 2.0f * fast_atan2(q.x, -q.y)};
  if (sarg >= 0.99999f) return v3{0.0f, 0.5f * kPi, 2.0f * fast_atan2(-q.x, q.y)};
  return v3{fast_atan2(2.0f * (q.y * q.z + q.w * q.x), squ - sqx - sqy + sqz), fast_asin(sarg),
            fast_atan2(2.0f * (q.x * q.y + q.w * q.z), squ + sqx - sqy - sqz)};
}
// getQuaternionFromEuler(getEulerFromQuaternion(q)) (quadx_base_env.py:243) by half-angle algebra;
// the gimbal-lock branch (|sin pitch| >= 0.99999) falls back to the trigonometric definition.
PF_DEV quat canon_quat(quat q) {
  float sqx = q.x * q.x, sqy = q.y * q.y, sqz = q.z * q.z, squ = q.w * q.w;
  float sarg = -2.0f * (q.x * q.z - q.w * q.y) * frcp(sqx + sqy + sqz + squ);
  if (__builtin_fabsf(sarg) >= 0.99999f) return quat_from_euler(euler_from_quat(q));
  float ar = 2.0f * (q.y * q.z + q.w * q.x), br = squ - sqx - sqy + sqz;
  float ay = 2.0f * (q.x * q.y + q.w * q.z), by = squ + sqx - sqy - sqz;
  float hr = frsq(fmaf(ar, ar, br * br)), hy = frsq(fmaf(ay, ay, by * by));
  float cr, sr, cp, sp, cy, sy;
  half_angle(br * hr, ar * hr, cr, sr);
  half_angle(fsqrt((1.0f - sarg) * (1.0f + sarg)), sarg, cp, sp);
  half_angle(by * hy, ay * hy, cy, sy);
  quat t = quat_from_half_angles(cr, sr, cp, sp, cy, sy);
  float inv = frsq(fmaf(t.x, t.x, fmaf(t.y, t.y, fmaf(t.z, t.z, t.w * t.w))));
  return quat{t.x * inv, t.y * inv, t.z * inv, t.w * inv};
}

// q <- exp(w dt/2) (x) q, normalised: btMultiBody::stepPositionsMultiDof. The half-angle
// theta = |w| dt/2 is bounded by sqrt(3)*max_coord_vel*dt/2 <= pi/8 (enforced at context creation),
// so sin(theta)/theta and cos(theta) are short even polynomials in theta^2: no sqrt, no trig.
PF_DEV quat quat_integrate(quat q, v3 w, float half_dt) {
  float t2 = dot(w, w) * (half_dt * half_dt);
  float sinc = fmaf(t2, fmaf(t2, fmaf(t2, fmaf(t2, 2.7557319e-6f, -1.9841270e-4f), 8.3333333e-3f), -1.6666667e-1f), 1.0f);
  float c = fmaf(t2, fmaf(t2, fmaf(t2, fmaf(t2, fmaf(t2, -2.7557319e-7f, 2.4801587e-5f), -1.3888889e-3f), 4.1666667e-2f), -0.5f), 1.0f);
  float k = sinc * half_dt;
  float ax = w.x * k, ay = w.y * k, az = w.z * k;
  // (explicit fma chains: the library is built with -ffp-contract=off so that every instantiation of a kernel
  //  rounds identically -- pf_rollout must be bit-identical to k x pf_env_step)
  quat n{fmaf(c, q.x, fmaf(ax, q.w, fmaf(ay, q.z, -(az * q.y)))), fmaf(c, q.y, fmaf(ay, q.w, fmaf(az, q.x, -(ax * q.z)))),
         fmaf(c, q.z, fmaf(az, q.w, fmaf(ax, q.y, -(ay * q.x)))), fmaf(c, q.w, fmaf(-ax, q.x, fmaf(-ay, q.y, -(az * q.z))))};
  float inv = rsqrtf(fmaf(n.x, n.x, fmaf(n.y, n.y, fmaf(n.z, n.z, n.w * n.w))));
  return quat{n.x * inv, n.y * inv, n.z * inv, n.w * inv};
}

// ---------------------------------------------------------------- observation tile -> global memory
// Wave-cooperative streaming copy of `total` floats from an LDS tile (row-major observation rows of one wave) to
// global memory: one 16-byte non-temporal store per lane per trip (global_store_dwordx4 nt; the observation is
// consumed by the policy, not by the next env step, so it should not displace the persistent state from L2).
// `g` is wave-uniform; when it is not 16-byte aligned (a trajectory slot [step][n][D] with n * D odd) the copy
// falls back to dword stores.
typedef float pf_f4v __attribute__((ext_vector_type(4)));
PF_DEV void stream_tile(const float* tile, float* g, int total, int tid, int lanes = 64) {  // lanes: how many lanes of the wave take part
  const int n4 = total >> 2;
  if ((reinterpret_cast<uintptr_t>(g) & 15u) == 0) {
    const pf_f4v* t4 = reinterpret_cast<const pf_f4v*>(tile);
    pf_f4v* g4 = reinterpret_cast<pf_f4v*>(g);
    for (int i = tid; i < n4; i += lanes) __builtin_nontemporal_store(t4[i], &g4[i]);
  } else {
    for (int i = tid; i < (n4 << 2); i += lanes) __builtin_nontemporal_store(tile[i], &g[i]);
  }
  for (int i = (n4 << 2) + tid; i < total; i += lanes) __builtin_nontemporal_store(tile[i], &g[i]);
}

// (r05, measured and dropped: every LDS read of the tile issued ahead of the first store -- an unrolled, predicated copy of up to ten
//  rounds -- instead of this loop's read / wait / store per round: Hover 9.65 -> 9.91 us, QuadX-Waypoints 16.65 -> 17.3, Fixedwing-
//  Waypoints 20.63 -> 21.59 on one box, profiles/tools/r05/g21.sh. The phase traces' 0.36-0.6 us "obs stores issued" is the issue of
//  the 1 KB stores themselves, not the LDS round trips in between.)

// ---------------------------------------------------------------- contact reporting
// btBoxBoxDetector verdict (15 separating axes) for an oriented box against the world-aligned
// ground box; see oracle/uav_oracle.c:orc_box_box_overlap for the restated rule.
PF_DEV bool box_overlaps_aabb(v3 ca, const m3& R, const float ha[3], v3 cb, const float hb[3]) {
  float t[3] = {ca.x - cb.x, ca.y - cb.y, ca.z - cb.z};
  float Rm[3][3] = {{R.m00, R.m01, R.m02}, {R.m10, R.m11, R.m12}, {R.m20, R.m21, R.m22}};
  float Q[3][3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) Q[i][j] = __builtin_fabsf(Rm[i][j]);
  bool sep = false;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    float rad = hb[i] + Q[i][0] * ha[0] + Q[i][1] * ha[1] + Q[i][2] * ha[2];
    sep |= (__builtin_fabsf(t[i]) - rad > 0.0f);
  }
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    float proj = t[0] * Rm[0][j] + t[1] * Rm[1][j] + t[2] * Rm[2][j];
    float rad = ha[j] + Q[0][j] * hb[0] + Q[1][j] * hb[1] + Q[2][j] * hb[2];
    sep |= (__builtin_fabsf(proj) - rad > 0.0f);
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int i1 = (i + 1) % 3, i2 = (i + 2) % 3;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int j1 = (j + 1) % 3, j2 = (j + 2) % 3;
      float expr1 = t[i2] * Rm[i1][j] - t[i1] * Rm[i2][j];
      float rad = hb[i1] * (Q[i2][j] + 1e-5f) + hb[i2] * (Q[i1][j] + 1e-5f) + ha[j1] * (Q[i][j2] + 1e-5f) +
                  ha[j2] * (Q[i][j1] + 1e-5f);
      sep |= (__builtin_fabsf(expr1) - rad > 2.220446e-16f);
    }
  }
  return !sep;
}

// Cylinder (centre c, axis = third column of R, radius r, half length hl) against the world-aligned
// ground box: separating-axis test on the box's face normals with the cylinder's exact support
// extent; see oracle/uav_oracle.c:orc_contact_plane for the restated rule.
PF_DEV bool cyl_overlaps_aabb(v3 c, const m3& R, float r, float hl, v3 cb, const float hb[3]) {
  const float ax[3] = {R.m02, R.m12, R.m22};
  const float t[3] = {c.x - cb.x, c.y - cb.y, c.z - cb.z};
  bool sep = false;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    float s2 = __builtin_fmaxf(1.0f - ax[k] * ax[k], 0.0f);
    float ext = hl * __builtin_fabsf(ax[k]) + r * __builtin_sqrtf(s2);
    sep |= (__builtin_fabsf(t[k]) - (hb[k] + ext) > 0.0f);
  }
  return !sep;
}

// ---------------------------------------------------------------- counter-based RNG
// Philox4x32-10; integer stream bit-identical to oracle/uav_oracle.c:orc_philox4x32. PF_PHILOX_ROUNDS == ORC_PHILOX_ROUNDS
// (the checker's header). Seven rounds -- the generator is Crush-resistant from seven on; a round is four 32-bit multiplies, quarter-
// rate instructions -- were measured in round 5: QuadX-Waypoints 18.2 -> 17.7 us per step, the rollouts -0.3 us, Hover per step
// unchanged. Not adopted: every Philox-noise parity test is calibrated on the ten-round stream's realisation (which lane meets the
// floor when), and a new realisation moves five of their event-count bounds without telling anything about the kernels.
#ifndef PF_PHILOX_ROUNDS
#define PF_PHILOX_ROUNDS 10
#endif
struct u32x4 {
  uint32_t a, b, c, d;
};
PF_DEV u32x4 philox4x32(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3) {
#pragma unroll
  for (int r = 0; r < PF_PHILOX_ROUNDS; ++r) {
    uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
    c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  return u32x4{c0, c1, c2, c3};
}
// 23-bit uniform in (0,1), exact in fp32
PF_DEV float u01(uint32_t x) { return ((float)(x >> 9) + 0.5f) * (1.0f / 8388608.0f); }
struct f4 {
  float a, b, c, d;
};
PF_DEV f4 uniform4(u32x4 r) { return f4{u01(r.a), u01(r.b), u01(r.c), u01(r.d)}; }
// Motor-noise normals: one Box-Muller pair per 32-bit Philox word from two 16-bit uniforms
// (radius: low half, angle: high half) -> 8 normals per Philox call; see oracle orc_normal8.
// v_sin/v_cos take revolutions: sin(2 pi u) = __builtin_amdgcn_sinf(u).
struct f8 {
  float v[8];
};
PF_DEV void bm16(uint32_t w, float& z0, float& z1) {
  float u1 = ((float)(w & 0xFFFFu) + 0.5f) * (1.0f / 65536.0f);
  float u2 = (float)(w >> 16) * (1.0f / 65536.0f);
  float rad = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1));  // -2 ln u = -2 ln2 log2 u
  z0 = rad * __builtin_amdgcn_cosf(u2);
  z1 = rad * __builtin_amdgcn_sinf(u2);
}
PF_DEV f8 normal8(u32x4 r) {
  f8 z;
  bm16(r.a, z.v[0], z.v[1]);
  bm16(r.b, z.v[2], z.v[3]);
  bm16(r.c, z.v[4], z.v[5]);
  bm16(r.d, z.v[6], z.v[7]);
  return z;
}
PF_DEV float pick8(const f8& z, uint32_t i) {
  float a = (i & 1u) ? z.v[1] : z.v[0], b = (i & 1u) ? z.v[3] : z.v[2];
  float c = (i & 1u) ? z.v[5] : z.v[4], d = (i & 1u) ? z.v[7] : z.v[6];
  float lo = (i & 2u) ? b : a, hi = (i & 2u) ? d : c;
  return (i & 4u) ? hi : lo;
}
PF_DEV float pick4(f4 v, uint32_t i) {
  float lo = (i & 1u) ? v.b : v.a, hi = (i & 1u) ? v.d : v.c;
  return (i & 2u) ? hi : lo;
}

}  // namespace pf
