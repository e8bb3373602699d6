/*
 * uav_oracle.c -- TEST INFRASTRUCTURE ONLY. fp64 scalar restatement of the PyFlyt Aviary.step()
 * hot path (one drone at a time) + the slice of Bullet it drives. See uav_oracle.h for the
 * parity status. Citations are file:line under /root/reference/PyFlyt/ unless noted.
 *
 * Nothing in pyflyt_amd/ (the product) may include, link or call this file.
 */
#include "uav_oracle.h"

#include <math.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define PI 3.14159265358979323846

/* ------------------------------------------------------------------ small helpers */
static inline double sgn(double x) { return (double)((x > 0.0) - (x < 0.0)); } /* np.sign */
static inline double clipd(double x, double lo, double hi) { return x < lo ? lo : (x > hi ? hi : x); }
static inline void cross3(const double a[3], const double b[3], double o[3]) {
  double x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
  o[0] = x; o[1] = y; o[2] = z;
}
static inline double dot3(const double a[3], const double b[3]) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static inline void matvec(const double M[3][3], const double x[3], double o[3]) {
  double a = M[0][0] * x[0] + M[0][1] * x[1] + M[0][2] * x[2];
  double b = M[1][0] * x[0] + M[1][1] * x[1] + M[1][2] * x[2];
  double c = M[2][0] * x[0] + M[2][1] * x[1] + M[2][2] * x[2];
  o[0] = a; o[1] = b; o[2] = c;
}
static inline void matTvec(const double M[3][3], const double x[3], double o[3]) {
  double a = M[0][0] * x[0] + M[1][0] * x[1] + M[2][0] * x[2];
  double b = M[0][1] * x[0] + M[1][1] * x[1] + M[2][1] * x[2];
  double c = M[0][2] * x[0] + M[1][2] * x[1] + M[2][2] * x[2];
  o[0] = a; o[1] = b; o[2] = c;
}
static void inv3(const double A[3][3], double B[3][3]) {
  double c00 = A[1][1] * A[2][2] - A[1][2] * A[2][1];
  double c01 = A[1][2] * A[2][0] - A[1][0] * A[2][2];
  double c02 = A[1][0] * A[2][1] - A[1][1] * A[2][0];
  double det = A[0][0] * c00 + A[0][1] * c01 + A[0][2] * c02;
  double id = 1.0 / det;
  B[0][0] = c00 * id;
  B[0][1] = (A[0][2] * A[2][1] - A[0][1] * A[2][2]) * id;
  B[0][2] = (A[0][1] * A[1][2] - A[0][2] * A[1][1]) * id;
  B[1][0] = c01 * id;
  B[1][1] = (A[0][0] * A[2][2] - A[0][2] * A[2][0]) * id;
  B[1][2] = (A[0][2] * A[1][0] - A[0][0] * A[1][2]) * id;
  B[2][0] = c02 * id;
  B[2][1] = (A[0][1] * A[2][0] - A[0][0] * A[2][1]) * id;
  B[2][2] = (A[0][0] * A[1][1] - A[0][1] * A[1][0]) * id;
}

/* ------------------------------------------------------------------ RNG: Philox4x32-10
 * Counter-based generator (Salmon et al., SC'11), the integer stream is bit-identical to the
 * device implementation (pyflyt_amd/csrc/uav_device.hpp: philox4x32). The reference threads a numpy PCG64
 * Generator through its components (motors.py:134-138, waypoint_handler.py:72-83); that stream
 * cannot be matched on a GPU, so parity runs either inject the normals or share this Philox
 * stream (RNG row). orc_philox4x32_r takes the round count: tests/test_oracle_kat.py pins the round function
 * and the key schedule to the published known-answer vectors. */
void orc_philox4x32_r(uint64_t key, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, int rounds, uint32_t out[4]) {
  uint32_t k0 = (uint32_t)key, k1 = (uint32_t)(key >> 32);
  for (int r = 0; r < rounds; ++r) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
void orc_philox4x32(uint64_t key, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t out[4]) {
  orc_philox4x32_r(key, c0, c1, c2, c3, ORC_PHILOX_ROUNDS, out);
}
/* 23-bit uniform in (0,1): (k + 0.5) * 2^-23, k < 2^23 -- exactly representable in fp32 and fp64 */
static inline double u24(uint32_t x) { return ((double)(x >> 9) + 0.5) * (1.0 / 8388608.0); }
void orc_uniform4(uint64_t key, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, double u[4]) {
  uint32_t r[4];
  orc_philox4x32(key, c0, c1, c2, c3, r);
  for (int i = 0; i < 4; ++i) u[i] = u24(r[i]);
}
/* Motor-noise normals: each 32-bit Philox word yields one Box-Muller pair from two 16-bit
 * uniforms (radius from the low half, angle from the high half), i.e. 8 normals per Philox call.
 * 16-bit resolution (|z| <= 4.85, bulk granularity ~1e-4) is far below what a 2 % multiplicative
 * motor noise can resolve and halves the RNG cost on the device. Same definition in
 * pyflyt_amd/csrc/uav_device.hpp:normal8. */
void orc_normal8(uint64_t key, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, double z[8]) {
  uint32_t r[4];
  orc_philox4x32(key, c0, c1, c2, c3, r);
  for (int i = 0; i < 4; ++i) {
    double u1 = ((double)(r[i] & 0xFFFFu) + 0.5) * (1.0 / 65536.0);
    double u2 = (double)(r[i] >> 16) * (1.0 / 65536.0);
    double rad = sqrt(-2.0 * log(u1));
    z[2 * i] = rad * cos(2.0 * PI * u2);
    z[2 * i + 1] = rad * sin(2.0 * PI * u2);
  }
}
void orc_normal4(uint64_t key, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, double z[4]) {
  double z8[8];
  orc_normal8(key, c0, c1, c2, c3, z8);
  for (int i = 0; i < 4; ++i) z[i] = z8[i];
}
static double lane_normal(const orc_params* P, const orc_lane* L, uint32_t flat_idx, uint32_t stream) {
  double z[8];
  orc_normal8(P->seed, (uint32_t)L->lane_id, L->rng_ctr, flat_idx >> 3, stream, z);
  return z[flat_idx & 7];
}
static double lane_uniform(const orc_params* P, const orc_lane* L, uint32_t flat_idx, uint32_t stream) {
  double u[4];
  orc_uniform4(P->seed, (uint32_t)L->lane_id, L->rng_ctr, flat_idx >> 2, stream, u);
  return u[flat_idx & 3];
}

/* ------------------------------------------------------------------ Bullet helpers
 * [BULLET-FROM-MEMORY] pybullet.c getQuaternionFromEuler / getEulerFromQuaternion,
 * btMatrix3x3::setRotation. Quaternion order (x,y,z,w). Call sites: base_drone.py:115,
 * quadx.py:521,526, quadx_base_env.py:243, waypoint_handler.py:135. */
void orc_quat_from_euler(const double rpy[3], double q[4]) {
  double phi = rpy[0] / 2.0, the = rpy[1] / 2.0, psi = rpy[2] / 2.0;
  q[0] = sin(phi) * cos(the) * cos(psi) - cos(phi) * sin(the) * sin(psi);
  q[1] = cos(phi) * sin(the) * cos(psi) + sin(phi) * cos(the) * sin(psi);
  q[2] = cos(phi) * cos(the) * sin(psi) - sin(phi) * sin(the) * cos(psi);
  q[3] = cos(phi) * cos(the) * cos(psi) + sin(phi) * sin(the) * sin(psi);
  double len = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  q[0] /= len; q[1] /= len; q[2] /= len; q[3] /= len;
}
void orc_euler_from_quat(const double q[4], double rpy[3]) {
  double sqx = q[0] * q[0], sqy = q[1] * q[1], sqz = q[2] * q[2], squ = q[3] * q[3];
  double sarg = -2.0 * (q[0] * q[2] - q[3] * q[1]) / (sqx + sqy + sqz + squ);
  if (sarg <= -0.99999) {
    rpy[0] = 0.0; rpy[1] = -0.5 * PI; rpy[2] = 2.0 * atan2(q[0], -q[1]);
  } else if (sarg >= 0.99999) {
    rpy[0] = 0.0; rpy[1] = 0.5 * PI; rpy[2] = 2.0 * atan2(-q[0], q[1]);
  } else {
    rpy[0] = atan2(2.0 * (q[1] * q[2] + q[3] * q[0]), squ - sqx - sqy + sqz);
    rpy[1] = asin(sarg);
    rpy[2] = atan2(2.0 * (q[0] * q[1] + q[3] * q[2]), squ + sqx - sqy - sqz);
  }
}
void orc_matrix_from_quat(const double q[4], double R[3][3]) {
  double d = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
  double s = 2.0 / d;
  double xs = q[0] * s, ys = q[1] * s, zs = q[2] * s;
  double wx = q[3] * xs, wy = q[3] * ys, wz = q[3] * zs;
  double xx = q[0] * xs, xy = q[0] * ys, xz = q[0] * zs;
  double yy = q[1] * ys, yz = q[1] * zs, zz = q[2] * zs;
  R[0][0] = 1.0 - (yy + zz); R[0][1] = xy - wz; R[0][2] = xz + wy;
  R[1][0] = xy + wz; R[1][1] = 1.0 - (xx + zz); R[1][2] = yz - wx;
  R[2][0] = xz - wy; R[2][1] = yz + wx; R[2][2] = 1.0 - (xx + yy);
}

/* Contact reporting -- [BULLET-FROM-MEMORY] btBoxBoxDetector (ODE dBoxBox2): two boxes produce
 * manifold points iff none of the 15 separating-axis tests separates them (face axes: s > 0,
 * edge axes: s > SIMD_EPSILON with |R| fudged by 1e-5). Box a is oriented (centre ca, axes = the
 * columns of Ra, half extents ha); box b is world-axis-aligned (cb, hb). aviary.py:523-525. */
int orc_box_box_overlap(const double ca[3], const double Ra[3][3], const double ha[3],
                        const double cb[3], const double hb[3]) {
  double t[3] = {ca[0] - cb[0], ca[1] - cb[1], ca[2] - cb[2]};
  double Q[3][3];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) Q[i][j] = fabs(Ra[i][j]);
  for (int i = 0; i < 3; ++i) { /* faces of the axis-aligned box */
    double rad = hb[i] + Q[i][0] * ha[0] + Q[i][1] * ha[1] + Q[i][2] * ha[2];
    if (fabs(t[i]) - rad > 0.0) return 0;
  }
  for (int j = 0; j < 3; ++j) { /* faces of the oriented box */
    double proj = t[0] * Ra[0][j] + t[1] * Ra[1][j] + t[2] * Ra[2][j];
    double rad = ha[j] + Q[0][j] * hb[0] + Q[1][j] * hb[1] + Q[2][j] * hb[2];
    if (fabs(proj) - rad > 0.0) return 0;
  }
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) Q[i][j] += 1e-5;
  for (int i = 0; i < 3; ++i) { /* edge x edge: e_i x a_j */
    int i1 = (i + 1) % 3, i2 = (i + 2) % 3;
    for (int j = 0; j < 3; ++j) {
      int j1 = (j + 1) % 3, j2 = (j + 2) % 3;
      double expr1 = t[i2] * Ra[i1][j] - t[i1] * Ra[i2][j];
      double rad = hb[i1] * Q[i2][j] + hb[i2] * Q[i1][j] + ha[j1] * Q[i][j2] + ha[j2] * Q[i][j1];
      if (fabs(expr1) - rad > 2.220446049250313e-16) return 0;
    }
  }
  return 1;
}
static int contact_plane_reach(const orc_params* P, const double p[3], const double q[4], double rd);
int orc_contact_plane(const orc_params* P, const double p[3], const double q[4]) { /* a fresh pair (nothing persisting) */
  return contact_plane_reach(P, p, q, P->world.contact_report_distance);
}
static int contact_plane_reach(const orc_params* P, const double p[3], const double q[4], const double rd) {
  /* ground = plane.urdf collision box (30,30,10)*world_scale centred at z=-5*world_scale,
   * aviary.py:240-242. Cheap exact early-out on the bounding sphere.
   * rd: a contact is reported from this gap on -- the 15-axis verdict against the slab enlarged by it */
  if (p[2] - P->bound_radius > rd) return 0;
  double R[3][3];
  orc_matrix_from_quat(q, R);
  double cb[3] = {0.0, 0.0, -P->world.plane_half_z};
  double hb[3] = {P->world.plane_half_xy + rd, P->world.plane_half_xy + rd, P->world.plane_half_z + rd};
  for (int k = 0; k < P->n_boxes; ++k) {
    double off[3], ca[3];
    matvec(R, P->boxes[k].c, off);
    ca[0] = p[0] + off[0]; ca[1] = p[1] + off[1]; ca[2] = p[2] + off[2];
    if (P->boxes[k].kind == 1) {
      /* Cylinder along the link z axis against the ground slab: separating-axis test on the slab's
       * three face normals with the cylinder's exact support extent h|a_k| + r sqrt(1 - a_k^2); exact
       * except within one radius of the slab's rim. [BULLET-FROM-MEMORY]: Bullet runs GJK/EPA on this
       * convex pair; as for the boxes, the verdict restated is penetration >= 0. */
      const double r = P->boxes[k].h[0], hl = P->boxes[k].h[2];
      int sep = 0;
      for (int a = 0; a < 3; ++a) {
        double ax = R[a][2];
        double s2 = 1.0 - ax * ax;
        double ext = hl * fabs(ax) + r * sqrt(s2 > 0.0 ? s2 : 0.0);
        if (fabs(ca[a] - cb[a]) - (hb[a] + ext) > 0.0) sep = 1;
      }
      if (!sep) return 1;
    } else if (P->boxes[k].yaw != 0.0) { /* a box on a yaw-rotated link: axes = R * Rz(yaw) */
      const double cy = cos(P->boxes[k].yaw), sy = sin(P->boxes[k].yaw);
      double Rr[3][3];
      for (int a = 0; a < 3; ++a) {
        Rr[a][0] = R[a][0] * cy + R[a][1] * sy;
        Rr[a][1] = -R[a][0] * sy + R[a][1] * cy;
        Rr[a][2] = R[a][2];
      }
      if (orc_box_box_overlap(ca, Rr, P->boxes[k].h, cb, hb)) return 1;
    } else if (orc_box_box_overlap(ca, R, P->boxes[k].h, cb, hb)) {
      return 1;
    }
  }
  return 0;
}

/* One free-body tick of Bullet's multibody world = the arithmetic behind `stepSimulation`
 * (aviary.py:516) for a base with only fixed, possibly massive, child links.
 * [BULLET-FROM-MEMORY] btMultiBody::computeAccelerationsArticulatedBodyAlgorithmMultiDof ->
 * applyDeltaVeeMultiDof (per-coordinate clamp) -> stepPositionsMultiDof (exp-map quaternion).
 * F_b: net external force in the base frame (gravity excluded); tau_b: net external torque about
 * the BASE ORIGIN in the base frame. State (p, v) is that of the base origin, as Bullet keeps it. */
typedef struct { /* the composite body the tick integrates; constant for QuadX / Fixedwing, rebuilt per tick for the Rocket */
  double mass, com[3], I_own[3][3], I_pa[3][3], I_inv[3][3];
  int persisted; /* the body had contact points after the previous tick: they persist up to the contact breaking distance */
} orc_body;
/* how far above a face a vertex may be and still be a contact point (a constraint row) this tick */
static double contact_reach(const orc_world* W, int persisted) { return persisted ? W->contact_break_distance : W->contact_margin; }
/* ... and from what gap on the pair is REPORTED (getContactPoints) */
static double report_reach(const orc_world* W, int persisted) { return persisted ? W->contact_break_distance : W->contact_report_distance; }
static void rigid_tick_body(const orc_params* PP, const orc_body* B, double p[3], double q[4], double v[3], double w[3],
                            const double F_b[3], const double tau_b[3]);
void orc_rigid_tick(const orc_params* P, double p[3], double q[4], double v[3], double w[3],
                    const double F_b[3], const double tau_b[3]) {
  orc_body B;
  B.mass = P->mass;
  memcpy(B.com, P->com, sizeof(B.com));
  memcpy(B.I_own, P->I_own, sizeof(B.I_own));
  memcpy(B.I_pa, P->I_pa, sizeof(B.I_pa));
  memcpy(B.I_inv, P->I_inv, sizeof(B.I_inv));
  B.persisted = 0;
  rigid_tick_body(P, &B, p, q, v, w, F_b, tau_b);
}

/* ---- contact response against the ground slab (see orc_world in the header for the model) ---- */
static int contact_points_reach(const orc_params* P, const double p[3], const double q[4], double pts[][3], double depth[], double margin);
int orc_contact_points(const orc_params* P, const double p[3], const double q[4], double pts[][3], double depth[]) {
  return contact_points_reach(P, p, q, pts, depth, P->world.contact_margin);
}
static int contact_points_reach(const orc_params* P, const double p[3], const double q[4], double pts[][3], double depth[], const double margin) {
  int n = 0;
  if (p[2] - P->bound_radius > margin) return 0;
  double R[3][3];
  orc_matrix_from_quat(q, R);
  const double hxy = P->world.plane_half_xy, hz2 = 2.0 * P->world.plane_half_z;
  for (int k = 0; k < P->n_boxes; ++k) {
    const orc_box* b = &P->boxes[k];
    const double cy = cos(b->yaw), sy = sin(b->yaw);
    const int nv = b->kind == 1 ? 16 : 8;
    /* manifold reduction (boxes, contact_manifold_points < 8): only the four vertices of the face that looks down the most --
     * the incident face of a box-box face contact against the slab's top face [BULLET-FROM-MEMORY: dBoxBox2 clips the incident
     * face and keeps at most four points]. Link axes in the world frame: the columns of R Rz(yaw); their z components: */
    int face_axis = -1, face_up = 0;
    if (b->kind == 0 && P->world.contact_manifold_points < 8) {
      const double zr[3] = {R[2][0] * cy + R[2][1] * sy, -R[2][0] * sy + R[2][1] * cy, R[2][2]};
      face_axis = 0;
      for (int a = 1; a < 3; ++a) if (fabs(zr[a]) > fabs(zr[face_axis])) face_axis = a; /* the first axis on a tie */
      face_up = zr[face_axis] < 0.0; /* the face on the + side of that axis looks down when the axis points down */
    }
    for (int i = 0; i < nv; ++i) {
      if (face_axis >= 0 && (((i >> face_axis) & 1) != face_up)) continue;
      double l[3];
      if (b->kind == 1) { /* end disc e = -1, +1; rim point j at j * 45 degrees from the link x axis */
        const int e = i >> 3, j = i & 7;
        const double c45[8] = {1.0, 0.70710678118654752, 0.0, -0.70710678118654752, -1.0, -0.70710678118654752, 0.0, 0.70710678118654752};
        l[0] = b->h[0] * c45[j]; l[1] = b->h[0] * c45[(j + 6) & 7]; l[2] = (e ? 1.0 : -1.0) * b->h[2];
      } else {
        l[0] = (i & 1) ? b->h[0] : -b->h[0]; l[1] = (i & 2) ? b->h[1] : -b->h[1]; l[2] = (i & 4) ? b->h[2] : -b->h[2];
      }
      /* link frame (yawed about the base z axis) -> base frame -> world */
      double bl[3] = {b->c[0] + cy * l[0] - sy * l[1], b->c[1] + sy * l[0] + cy * l[1], b->c[2] + l[2]};
      double wpt[3];
      matvec(R, bl, wpt);
      wpt[0] += p[0]; wpt[1] += p[1]; wpt[2] += p[2];
      if (n < ORC_MAX_CONTACTS && wpt[2] <= margin && wpt[2] >= -hz2 && fabs(wpt[0]) <= hxy && fabs(wpt[1]) <= hxy) {
        memcpy(pts[n], wpt, sizeof(wpt));
        depth[n] = -wpt[2];
        ++n;
      }
    }
  }
  return n;
}
/* projected Gauss-Seidel on the base twist (v at the base origin, w; world frame). Returns the deepest penetration. */
/* diagnostics (tests/tools): [contacts 0..4+][0] = solves, [1] = sweeps, [2] = solves that ran into contact_iters */
static long long g_solve_stats[5][3];
static int g_solve_stats_on = 0; /* off unless a tool asks: the counters are shared between the OpenMP threads of the batch entry points */
void orc_debug_solve_stats(long long* out, int clear) {
  g_solve_stats_on = 1;
  memcpy(out, g_solve_stats, sizeof(g_solve_stats));
  if (clear) memset(g_solve_stats, 0, sizeof(g_solve_stats));
}
static double contact_solve(const orc_params* PP, const orc_body* B, const double p[3], const double q[4], double v[3], double w[3]) {
  const orc_world* W = &PP->world;
  double pts[ORC_MAX_CONTACTS][3], depth[ORC_MAX_CONTACTS];
  const int n = contact_points_reach(PP, p, q, pts, depth, contact_reach(W, B->persisted));
  if (n == 0) return 0.0;
  double R[3][3], Iw[3][3], tmp[3][3];
  orc_matrix_from_quat(q, R);
  for (int i = 0; i < 3; ++i) /* R I^-1 R^T */
    for (int j = 0; j < 3; ++j) { tmp[i][j] = 0; for (int k = 0; k < 3; ++k) tmp[i][j] += R[i][k] * B->I_inv[k][j]; }
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) { Iw[i][j] = 0; for (int k = 0; k < 3; ++k) Iw[i][j] += tmp[i][k] * R[j][k]; }
  double cw[3], vc[3], t[3];
  matvec(R, B->com, cw);
  cross3(w, cw, t);
  for (int i = 0; i < 3; ++i) vc[i] = v[i] + t[i]; /* COM velocity */
  const double im = 1.0 / B->mass;
  double lam[ORC_MAX_CONTACTS][3], vn0[ORC_MAX_CONTACTS], arm[ORC_MAX_CONTACTS][3], dmax = 0.0;
  for (int c = 0; c < n; ++c) {
    for (int i = 0; i < 3; ++i) { arm[c][i] = pts[c][i] - (p[i] + cw[i]); lam[c][i] = 0.0; }
    cross3(w, arm[c], t);
    vn0[c] = vc[2] + t[2];
    if (depth[c] > dmax) dmax = depth[c];
  }
  static const double dir[3][3] = {{0, 0, 1}, {1, 0, 0}, {0, 1, 0}}; /* normal, friction x, friction y */
  int sweeps_run = 0;
  for (int it = 0; it < W->contact_iters; ++it) {
    double res2 = 0.0; /* the sweep's largest squared row-velocity change: btMultiBodyConstraintSolver's least-squares residual */
    for (int c = 0; c < n; ++c) {
      for (int d = 0; d < 3; ++d) {
        double rxd[3], ang[3], axr[3], u[3];
        cross3(arm[c], dir[d], rxd);
        matvec(Iw, rxd, ang);
        cross3(ang, arm[c], axr);
        const double k = im + dot3(dir[d], axr);
        cross3(w, arm[c], t);
        for (int i = 0; i < 3; ++i) u[i] = vc[i] + t[i];
        double target = 0.0;
        if (d == 0) target = depth[c] < W->contact_slop ? (depth[c] - W->contact_slop) / W->dt /* may close the gap down to the slop, no more */
                                                        : (vn0[c] < 0.0 ? -W->contact_restitution * vn0[c] : 0.0);
        double dl = (target - dot3(u, dir[d])) / k, nl;
        if (d == 0) {
          nl = lam[c][0] + dl;
          if (nl < 0.0) nl = 0.0;
        } else {
          const double lim = W->contact_friction * lam[c][0];
          nl = lam[c][d] + dl;
          if (nl > lim) nl = lim;
          if (nl < -lim) nl = -lim;
        }
        dl = nl - lam[c][d];
        lam[c][d] = nl;
        for (int i = 0; i < 3; ++i) { vc[i] += im * dl * dir[d][i]; w[i] += dl * ang[i]; }
        if ((dl * k) * (dl * k) > res2) res2 = (dl * k) * (dl * k);
      }
    }
    sweeps_run = it + 1;
    if (res2 <= W->contact_residual_threshold) break;
  }
  if (g_solve_stats_on) {
    const int b = n > 4 ? 4 : n;
#pragma omp atomic
    g_solve_stats[b][0] += 1;
#pragma omp atomic
    g_solve_stats[b][1] += sweeps_run;
#pragma omp atomic
    g_solve_stats[b][2] += (sweeps_run >= W->contact_iters);
  }
  cross3(w, cw, t);
  for (int i = 0; i < 3; ++i) v[i] = vc[i] - t[i];
  return dmax;
}

/* first half: forces -> new velocities (semi-implicit Euler) */
static void rigid_tick_vel(const orc_params* PP, const orc_body* P, const double q[4], double v[3], double w[3],
                           const double F_b[3], const double tau_b[3]) {
  const orc_world* Wd = &PP->world;
  const double dt = Wd->dt;
  double R[3][3];
  orc_matrix_from_quat(q, R);
  double w_b[3];
  matTvec(R, w, w_b);
  /* torque about the composite COM */
  double cxF[3], tau_c[3];
  cross3(P->com, F_b, cxF);
  for (int i = 0; i < 3; ++i) tau_c[i] = tau_b[i] - cxF[i];
  /* gyroscopic bias: w x (I w); the links' own-inertia part is gated by m_useGyroTerm, the
   * point-mass (m w x v) part is unconditional in Bullet's ABA */
  double Iw[3], g1[3], g2[3] = {0, 0, 0};
  matvec(P->I_pa, w_b, Iw);
  cross3(w_b, Iw, g1);
  if (Wd->use_gyro_term) {
    matvec(P->I_own, w_b, Iw);
    cross3(w_b, Iw, g2);
  }
  double rhs[3], wdot_b[3], wdot[3];
  for (int i = 0; i < 3; ++i) rhs[i] = tau_c[i] - g1[i] - g2[i];
  matvec(P->I_inv, rhs, wdot_b);
  matvec(R, wdot_b, wdot);
  /* linear acceleration of the COM, then of the base origin */
  double F_w[3], a[3];
  matvec(R, F_b, F_w);
  a[0] = F_w[0] / P->mass; a[1] = F_w[1] / P->mass; a[2] = F_w[2] / P->mass + Wd->gravity_z;
  double c_w[3], t1[3], t2[3], t3[3];
  matvec(R, P->com, c_w);
  cross3(wdot, c_w, t1);
  cross3(w, c_w, t2);
  cross3(w, t2, t3);
  for (int i = 0; i < 3; ++i) a[i] = a[i] - t1[i] - t3[i];
  /* v += a dt with the per-coordinate clamp (order: omega then velocity) */
  const double vmax = Wd->max_coord_vel;
  for (int i = 0; i < 3; ++i) w[i] = clipd(w[i] + wdot[i] * dt, -vmax, vmax);
  for (int i = 0; i < 3; ++i) v[i] = clipd(v[i] + a[i] * dt, -vmax, vmax);
}
/* second half: ground contact solve on the new velocities, position / orientation update, penetration recovery.
 * shift: an additional translation after the position update (the pair stage's penetration recovery), or null */
static void rigid_tick_pos(const orc_params* PP, const orc_body* P, double p[3], double q[4], double v[3], double w[3], const double* shift) {
  const orc_world* Wd = &PP->world;
  const double dt = Wd->dt;
  /* constraint solve: contacts found at the pre-integration pose act on the new velocities */
  const double deepest = Wd->contact_response ? contact_solve(PP, P, p, q, v, w) : 0.0;
  /* x += v dt (semi-implicit Euler: new velocity) */
  for (int i = 0; i < 3; ++i) p[i] += dt * v[i];
  if (deepest > Wd->contact_slop) p[2] += Wd->contact_erp * (deepest - Wd->contact_slop); /* penetration recovery, position level */
  if (shift) for (int i = 0; i < 3; ++i) p[i] += shift[i];
  /* q <- exp(w dt / 2) * q with world-frame w, then normalise */
  double fAngle = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
  if (fAngle * dt > 0.25 * PI) fAngle = 0.25 * PI / dt; /* ANGULAR_MOTION_THRESHOLD */
  double ax[3], k;
  if (fAngle < 0.001)
    k = 0.5 * dt - (dt * dt * dt) * 0.020833333333 * fAngle * fAngle;
  else
    k = sin(0.5 * fAngle * dt) / fAngle;
  ax[0] = w[0] * k; ax[1] = w[1] * k; ax[2] = w[2] * k;
  double cw = cos(fAngle * dt * 0.5);
  double nq[4];
  nq[0] = cw * q[0] + ax[0] * q[3] + ax[1] * q[2] - ax[2] * q[1];
  nq[1] = cw * q[1] + ax[1] * q[3] + ax[2] * q[0] - ax[0] * q[2];
  nq[2] = cw * q[2] + ax[2] * q[3] + ax[0] * q[1] - ax[1] * q[0];
  nq[3] = cw * q[3] - ax[0] * q[0] - ax[1] * q[1] - ax[2] * q[2];
  double inv = 1.0 / sqrt(nq[0] * nq[0] + nq[1] * nq[1] + nq[2] * nq[2] + nq[3] * nq[3]);
  for (int i = 0; i < 4; ++i) q[i] = nq[i] * inv;
}
static void rigid_tick_body(const orc_params* PP, const orc_body* P, double p[3], double q[4], double v[3], double w[3],
                            const double F_b[3], const double tau_b[3]) {
  rigid_tick_vel(PP, P, q, v, w, F_b, tau_b);
  rigid_tick_pos(PP, P, p, q, v, w, 0);
}

/* ------------------------------------------------------------------ PID  (abstractions/pid.py:70-94) */
void orc_pid_step(const double* kp, const double* ki, const double* kd, const double* lim,
                  double period, int n, double* I, double* E, const double* state,
                  const double* setpoint, double* out) {
  double tmp[3];
  for (int i = 0; i < n; ++i) {
    double error = setpoint[i] - state[i];
    double proportional = kp[i] * error;
    I[i] = clipd(I[i] + ki[i] * error * period, -lim[i], lim[i]);
    double derivative = kd[i] * (error - E[i]) / period;
    E[i] = error;
    tmp[i] = clipd(proportional + I[i] + derivative, -lim[i], lim[i]);
  }
  for (int i = 0; i < n; ++i) out[i] = tmp[i];
}
static void pidn(const orc_params* P, orc_lane* L, int k, int n, const double* state, double* io) {
  const orc_pid_gains* g = &P->pid[k];
  orc_pid_step(g->kp, g->ki, g->kd, g->lim, P->control_period, n, L->pid_I[k], L->pid_E[k], state, io, io);
}
static double zpid(const orc_params* P, orc_lane* L, int k, double state, double sp) {
  const orc_pid_gains* g = &P->zpid[k];
  double out;
  orc_pid_step(g->kp, g->ki, g->kd, g->lim, P->control_period, 1, &L->zpid_I[k], &L->zpid_E[k], &state, &sp, &out);
  return out;
}

/* ------------------------------------------------------------------ QuadX control
 * motor mixing + saturation: drones/quadx.py:482-493 */
void orc_quadx_mix(const orc_params* P, const double cmd[4], double pwm[4]) {
  for (int i = 0; i < 4; ++i) {
    double s = 0.0;
    for (int j = 0; j < 4; ++j) s += P->motor_map[i][j] * cmd[j];
    pwm[i] = s;
  }
  double high = pwm[0], low = pwm[0];
  for (int i = 1; i < 4; ++i) {
    if (pwm[i] > high) high = pwm[i];
    if (pwm[i] < low) low = pwm[i];
  }
  if (high != low) {
    double pwm_max = high < 1.0 ? high : 1.0, pwm_min = low > 0.05 ? low : 0.05;
    for (int i = 0; i < 4; ++i) {
      double add = (pwm_min - low) / (pwm_max - low) * (pwm_max - pwm[i]);
      double sub = (high - pwm_max) / (high - pwm_min) * (pwm[i] - pwm_min);
      pwm[i] += add - sub;
    }
  }
  for (int i = 0; i < 4; ++i) pwm[i] = clipd(pwm[i], 0.05, 1.0);
}
/* drones/quadx.py:401-493 (the physics_step % ratio gate lives in orc_aviary_step) */
void orc_quadx_control(const orc_params* P, orc_lane* L) {
  double a[3] = {L->setpoint[0], L->setpoint[1], L->setpoint[2]};
  double z = L->setpoint[3];
  const int mode = L->mode;
  if (mode == -1) { /* quadx.py:432-434 */
    for (int i = 0; i < 3; ++i) L->pwm[i] = a[i];
    L->pwm[3] = z;
    return;
  }
  const double* w_b = L->w_b;
  const double* rpy = L->rpy;
  const double* v_b = L->v_b;
  const double* pos = L->p;
  if (mode == 0 || mode == 2) {
    pidn(P, L, 0, 3, w_b, a);
  } else if (mode == 1 || mode == 3) {
    pidn(P, L, 1, 3, rpy, a);
    pidn(P, L, 0, 3, w_b, a);
  } else if (mode == 4 || mode == 5 || mode == 6) {
    if (mode == 6) { /* quadx.py:448-451 */
      double c = cos(rpy[2]), s = sin(rpy[2]);
      double a0 = c * a[0] + s * a[1], a1 = -s * a[0] + c * a[1];
      a[0] = a0; a[1] = a1;
    }
    pidn(P, L, 2, 2, v_b, a);
    { double t0 = -a[1], t1 = a[0]; a[0] = t0; a[1] = t1; }
    pidn(P, L, 1, 2, rpy, a);
    pidn(P, L, 0, 3, w_b, a);
  } else if (mode == 7) {
    pidn(P, L, 3, 2, pos, a);
    double c = cos(rpy[2]), s = sin(rpy[2]);
    double a0 = c * a[0] + s * a[1], a1 = -s * a[0] + c * a[1];
    a[0] = a0; a[1] = a1;
    pidn(P, L, 2, 2, v_b, a);
    { double t0 = -a[1], t1 = a[0]; a[0] = t0; a[1] = t1; }
    pidn(P, L, 1, 3, rpy, a);
    pidn(P, L, 0, 3, w_b, a);
  }
  /* height controllers, quadx.py:471-479 */
  if (mode == 0) {
    z = clipd(z, 0.0, 1.0);
  } else if (mode == 1 || mode == 5 || mode == 6) {
    z = zpid(P, L, 0, v_b[2], z);
    z = clipd(z, 0.0, 1.0);
  } else {
    z = zpid(P, L, 1, pos[2], z);
    z = zpid(P, L, 0, v_b[2], z);
    z = clipd(z, 0.0, 1.0);
  }
  double cmd[4] = {a[0], a[1], a[2], z};
  orc_quadx_mix(P, cmd, L->pwm);
}

/* ------------------------------------------------------------------ Motors (abstractions/motors.py:110-195) */
void orc_motors_update(const orc_params* P, double* throttle, const double* pwm, double xi,
                       double thrust[4][3], double torque[4][3]) {
  for (int i = 0; i < P->n_motors; ++i) {
    throttle[i] += (P->world.dt / P->motor_tau[i]) * (pwm[i] - throttle[i]); /* :131 */
    throttle[i] += xi * throttle[i] * P->noise_ratio[i];                     /* :134-138 */
    double rpm = throttle[i] * P->max_rpm[i];
    for (int k = 0; k < 3; ++k) {
      double rpm_const = (rpm * rpm) * sgn(rpm) * P->thrust_unit[i][k]; /* :189 */
      thrust[i][k] = rpm_const * P->thrust_coef[i];
      torque[i][k] = rpm_const * P->torque_coef[i];
    }
  }
}
/* abstractions/boring_bodies.py:113-119 */
void orc_body_drag(const orc_params* P, const double v_b[3], double F[3]) {
  for (int k = 0; k < 3; ++k) F[k] = -sgn(v_b[k]) * P->drag_const[k] * (v_b[k] * v_b[k]);
}

/* ------------------------------------------------------------------ Lifting surface
 * abstractions/lifting_surfaces.py:349-448 (aero data) */
void orc_surface_aero(const orc_surface* S, double alpha, double actuation, double out[3]) {
  double deflection_radians = (actuation * S->deflection_limit) * (PI / 180.0); /* np.deg2rad */
  double delta_Cl = S->Cl_alpha_3D * S->aero_tau * S->eta * deflection_radians;
  double delta_Cl_max = S->flap_to_chord * delta_Cl;
  double Cl_max_P = S->Cl_alpha_3D * (S->alpha_stall_P_base - S->alpha_0_base) + delta_Cl_max;
  double Cl_max_N = S->Cl_alpha_3D * (S->alpha_stall_N_base - S->alpha_0_base) + delta_Cl_max;
  double alpha_0 = S->alpha_0_base - (delta_Cl / S->Cl_alpha_3D);
  double alpha_stall_P = alpha_0 + (Cl_max_P / S->Cl_alpha_3D);
  double alpha_stall_N = alpha_0 + (Cl_max_N / S->Cl_alpha_3D);
  double Cl, Cd, CM;
  if (alpha_stall_N < alpha && alpha < alpha_stall_P) { /* :397-406 */
    Cl = S->Cl_alpha_3D * (alpha - alpha_0);
    double alpha_i = Cl / (PI * S->aspect);
    double alpha_eff = alpha - alpha_0 - alpha_i;
    double CT = S->Cd_0 * cos(alpha_eff);
    double CN = (Cl + (CT * sin(alpha_eff))) / cos(alpha_eff);
    Cd = (CN * sin(alpha_eff)) + (CT * cos(alpha_eff));
    CM = -CN * (0.25 - (0.175 * (1.0 - ((2.0 * alpha_eff) / PI))));
    out[0] = Cl; out[1] = Cd; out[2] = CM;
    return;
  }
  double alpha_i;
  if (alpha > 0.0) { /* :409-416, np.interp over two points (clamped at the ends) */
    double Cl_stall = S->Cl_alpha_3D * (alpha_stall_P - alpha_0);
    double alpha_i_at_stall = Cl_stall / (PI * S->aspect);
    double x0 = alpha_stall_P, x1 = PI / 2.0;
    if (alpha <= x0) alpha_i = alpha_i_at_stall;
    else if (alpha >= x1) alpha_i = 0.0;
    else alpha_i = alpha_i_at_stall + (0.0 - alpha_i_at_stall) / (x1 - x0) * (alpha - x0);
  } else { /* :417-425 */
    double Cl_stall = S->Cl_alpha_3D * (alpha_stall_N - alpha_0);
    double alpha_i_at_stall = Cl_stall / (PI * S->aspect);
    double x0 = -PI / 2.0, x1 = alpha_stall_N;
    if (alpha <= x0) alpha_i = 0.0;
    else if (alpha >= x1) alpha_i = alpha_i_at_stall;
    else alpha_i = 0.0 + (alpha_i_at_stall - 0.0) / (x1 - x0) * (alpha - x0);
  }
  double alpha_eff = alpha - alpha_0 - alpha_i;
  double Cd_90 = ((-4.26e-2) * (deflection_radians * deflection_radians)) + ((2.1e-1) * deflection_radians) + 1.98;
  double CN = Cd_90 * sin(alpha_eff) *
              (1.0 / (0.56 + 0.44 * fabs(sin(alpha_eff))) - 0.41 * (1.0 - exp(-17.0 / S->aspect)));
  double CT = 0.5 * S->Cd_0 * cos(alpha_eff);
  Cl = (CN * cos(alpha_eff)) - (CT * sin(alpha_eff));
  Cd = (CN * sin(alpha_eff)) + (CT * cos(alpha_eff));
  CM = -CN * (0.25 - (0.175 * (1.0 - ((2.0 * fabs(alpha_eff)) / PI))));
  out[0] = Cl; out[1] = Cd; out[2] = CM;
}
/* lifting_surfaces.py:326-347 (alpha, V) and :450-498 (force, torque); actuation already updated */
void orc_surface_force(const orc_surface* S, const double v_local[3], double actuation,
                       double F[3], double T[3]) {
  double freestream_speed = sqrt(dot3(v_local, v_local));
  double lifting_airspeed = dot3(v_local, S->lift_unit);
  double forward_airspeed = dot3(v_local, S->drag_unit);
  double alpha = atan2(-lifting_airspeed, forward_airspeed);
  double c[3];
  orc_surface_aero(S, alpha, actuation, c);
  double Q = S->half_rho * (freestream_speed * freestream_speed);
  double Q_area = Q * S->area;
  double lift = c[0] * Q_area, drag = c[1] * Q_area;
  double force_normal = (lift * cos(alpha)) + (drag * sin(alpha));
  double force_parallel = (lift * sin(alpha)) - (drag * cos(alpha));
  for (int k = 0; k < 3; ++k) {
    F[k] = S->lift_unit[k] * force_normal + S->drag_unit[k] * force_parallel;
    T[k] = Q_area * c[2] * S->chord * S->torque_unit[k];
  }
}

/* ------------------------------------------------------------------ parameter sets */
static void world_defaults(orc_world* W) {
  W->dt = 1.0 / 240.0;       /* aviary.py:79 */
  W->gravity_z = -9.81;      /* aviary.py:226 */
  W->use_gyro_term = 1;      /* [BULLET-FROM-MEMORY] */
  W->max_coord_vel = 100.0;  /* [BULLET-FROM-MEMORY] */
  W->contact_response = 1; W->contact_restitution = 0.0; W->contact_friction = 0.5; W->contact_erp = 0.2; W->contact_iters = 50; W->contact_margin = 0.0; W->contact_slop = 1e-5; W->pair_response = 1; /* [BULLET-FROM-MEMORY] defaults: see orc_world */
  W->contact_report_distance = 0.0; W->contact_residual_threshold = 1e-7; W->contact_manifold_points = 4; W->contact_break_distance = 0.02;
  W->plane_half_xy = 15.0;   /* [BULLET-FROM-MEMORY] pybullet_data plane.urdf */
  W->plane_half_z = 5.0;
  W->ticks_per_control = 2;  /* 240/120, quadx.py:27-28 */
}
static void set_pid(orc_pid_gains* g, int n, const double* kp, const double* ki, const double* kd, const double* lim) {
  memset(g, 0, sizeof(*g));
  for (int i = 0; i < n; ++i) { g->kp[i] = kp[i]; g->ki[i] = ki[i]; g->kd[i] = kd[i]; g->lim[i] = lim[i]; }
}
void orc_finalize(orc_params* P) {
  double I[3][3];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) I[i][j] = P->I_own[i][j] + P->I_pa[i][j];
  inv3(I, P->I_inv);
  double r = 0.0;
  for (int k = 0; k < P->n_boxes; ++k) {
    double m = 0.0;
    for (int s = 0; s < 8; ++s) {
      double x = P->boxes[k].c[0] + ((s & 1) ? 1 : -1) * P->boxes[k].h[0];
      double y = P->boxes[k].c[1] + ((s & 2) ? 1 : -1) * P->boxes[k].h[1];
      double z = P->boxes[k].c[2] + ((s & 4) ? 1 : -1) * P->boxes[k].h[2];
      double d = sqrt(x * x + y * y + z * z);
      if (d > m) m = d;
    }
    if (P->boxes[k].yaw != 0.0) { /* rotated about z: bound by |c| + |h|, whatever the yaw */
      const double* c = P->boxes[k].c; const double* h = P->boxes[k].h;
      m = sqrt(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]) + sqrt(h[0] * h[0] + h[1] * h[1] + h[2] * h[2]);
    }
    if (m > r) r = m;
  }
  P->bound_radius = r;
  for (int i = 0; i < P->n_surf; ++i) { /* lifting_surfaces.py:228-239 */
    orc_surface* S = &P->surf[i];
    S->half_rho = 0.5 * 1.225;
    S->area = S->chord * S->span;
    S->aspect = S->span / S->chord;
    S->Cl_alpha_3D = S->Cl_alpha_2D * (S->aspect / (S->aspect + ((2.0 * (S->aspect + 4.0)) / (S->aspect + 2.0))));
    S->theta_f = acos(2.0 * S->flap_to_chord - 1.0);
    S->aero_tau = 1 - ((S->theta_f - sin(S->theta_f)) / PI);
    cross3(S->lift_unit, S->drag_unit, S->torque_unit);
  }
}

void orc_params_quadx(orc_params* P) {
  memset(P, 0, sizeof(*P));
  P->vehicle = ORC_QUADX;
  world_defaults(&P->world);
  /* cf2x.urdf:13-14 */
  P->mass = 0.027;
  P->I_own[0][0] = 1.4e-5; P->I_own[1][1] = 1.4e-5; P->I_own[2][2] = 2.17e-5;
  /* cf2x.urdf:30-36 collision box 0.09 x 0.09 x 0.02 on the base */
  P->n_boxes = 1;
  P->boxes[0].h[0] = 0.045; P->boxes[0].h[1] = 0.045; P->boxes[0].h[2] = 0.01;
  /* cf2x.yaml:1-6, quadx.py:93-128 */
  const double total_thrust = 2.0, thrust_coef = 3.16e-10, torque_coef = 7.94e-12, noise = 0.02, tau = 0.01;
  P->n_motors = 4;
  const double rx[4] = {0.028, -0.028, 0.028, -0.028}; /* cf2x.urdf:42,54,66,78 */
  const double ry[4] = {-0.028, 0.028, 0.028, -0.028};
  const double tq[4] = {-1, -1, +1, +1};               /* quadx.py:94-101 */
  for (int i = 0; i < 4; ++i) {
    P->motor_r[i][0] = rx[i]; P->motor_r[i][1] = ry[i]; P->motor_r[i][2] = 0.0;
    P->thrust_unit[i][2] = 1.0;
    P->thrust_coef[i] = thrust_coef;
    P->torque_coef[i] = tq[i] * torque_coef;
    P->max_rpm[i] = 1.0 * sqrt(total_thrust / (4 * thrust_coef)); /* quadx.py:111-113 */
    P->motor_tau[i] = tau;
    P->noise_ratio[i] = 1.0 * noise;
  }
  const double M[4][4] = {{-1, -1, -1, +1}, {+1, +1, -1, +1}, {+1, -1, +1, +1}, {-1, +1, +1, +1}}; /* quadx.py:130-137 */
  memcpy(P->motor_map, M, sizeof(M));
  for (int k = 0; k < 3; ++k) P->drag_const[k] = 0.5 * 1.225 * 3.0 * 4.0e-4; /* boring_bodies.py:63, cf2x.yaml:9-10 */
  P->drag_coef_pqr = 1.0e-4;                                                  /* cf2x.yaml:11 */
  { /* cf2x.yaml:13-54 */
    const double kp0[3] = {4.0e-2, 4.0e-2, 8.0e-2}, ki0[3] = {5.0e-7, 5.0e-7, 2.7e-4}, kd0[3] = {1.0e-4, 1.0e-4, 0.0}, l0[3] = {1, 1, 1};
    const double kp1[3] = {2, 2, 2}, z3[3] = {0, 0, 0}, l1[3] = {3, 3, 3};
    const double kp2[2] = {0.8, 0.8}, ki2[2] = {0.3, 0.3}, kd2[2] = {0.5, 0.5}, l2[2] = {0.4, 0.4};
    const double kp3[2] = {1, 1}, l3[2] = {2, 2};
    set_pid(&P->pid[0], 3, kp0, ki0, kd0, l0);
    set_pid(&P->pid[1], 3, kp1, z3, z3, l1);
    set_pid(&P->pid[2], 2, kp2, ki2, kd2, l2);
    set_pid(&P->pid[3], 2, kp3, z3, z3, l3);
    const double zvkp = 2.0, zvki = 0.5, zvkd = 0.05, zvl = 1.0;
    const double zpkp = 1.0, zero = 0.0, zpl = 1.0;
    set_pid(&P->zpid[0], 1, &zvkp, &zvki, &zvkd, &zvl);
    set_pid(&P->zpid[1], 1, &zpkp, &zero, &zero, &zpl);
  }
  P->control_period = 1.0 / 120.0; /* quadx.py:27 */
  P->start_pos[2] = 1.0;           /* quadx_base_env.py:23 */
  P->settle_steps = 10;
  P->angle_repr = 1;
  orc_finalize(P);
}

/* QuadX with drone_model="primitive_drone" (quadx.py:29; used by examples/core/08_mixed_drones.py and
 * the pole / ball-in-cup envs): models/vehicles/primitive_drone/primitive_drone.{urdf,yaml}. */
void orc_params_primitive_drone(orc_params* P) {
  orc_params_quadx(P);
  /* primitive_drone.urdf:27-30 */
  P->mass = 1.0;
  P->I_own[0][0] = 0.01; P->I_own[1][1] = 0.01; P->I_own[2][2] = 0.016;
  /* collision: base box 0.2 x 0.1 x 0.05 (:21-26) + four prop discs, cylinders r 0.12, length 0.01
   * (:42-47,69-74,96-101,123-128) at the prop joints (:56,83,110,138) */
  const double rx[4] = {0.16, -0.16, 0.16, -0.16}, ry[4] = {-0.16, 0.16, 0.16, -0.16};
  P->n_boxes = 5;
  memset(P->boxes, 0, sizeof(P->boxes));
  P->boxes[0].h[0] = 0.1; P->boxes[0].h[1] = 0.05; P->boxes[0].h[2] = 0.025;
  for (int i = 0; i < 4; ++i) {
    P->boxes[1 + i].kind = 1;
    P->boxes[1 + i].c[0] = rx[i]; P->boxes[1 + i].c[1] = ry[i];
    P->boxes[1 + i].h[0] = 0.12; P->boxes[1 + i].h[1] = 0.12; P->boxes[1 + i].h[2] = 0.005;
  }
  /* primitive_drone.yaml:1-6 */
  const double total_thrust = 40.0, thrust_coef = 3.0e-7, torque_coef = 3.0e-7, noise = 0.003, tau = 0.01;
  const double tq[4] = {-1, -1, +1, +1};
  for (int i = 0; i < 4; ++i) {
    P->motor_r[i][0] = rx[i]; P->motor_r[i][1] = ry[i];
    P->thrust_coef[i] = thrust_coef;
    P->torque_coef[i] = tq[i] * torque_coef;
    P->max_rpm[i] = sqrt(total_thrust / (4 * thrust_coef));
    P->motor_tau[i] = tau;
    P->noise_ratio[i] = noise;
  }
  for (int k = 0; k < 3; ++k) P->drag_const[k] = 0.5 * 1.225 * 2.0 * 0.08; /* primitive_drone.yaml:8-11 */
  P->drag_coef_pqr = 1.0e-4;
  { /* primitive_drone.yaml:13-54 */
    const double kp0[3] = {1.5e-2, 1.5e-2, 5.0e-3}, ki0[3] = {1.0e-5, 1.0e-5, 2.0e-6}, kd0[3] = {1.2e-5, 1.2e-5, 1.2e-6}, l0[3] = {1, 1, 1};
    const double kp1[3] = {2, 2, 2}, z3[3] = {0, 0, 0}, l1[3] = {6, 6, 6};
    const double kp2[2] = {0.3, 0.3}, ki2[2] = {0.03, 0.03}, kd2[2] = {0.3, 0.3}, l2[2] = {1.0, 1.0};
    const double kp3[2] = {1, 1}, l3[2] = {5, 5};
    set_pid(&P->pid[0], 3, kp0, ki0, kd0, l0);
    set_pid(&P->pid[1], 3, kp1, z3, z3, l1);
    set_pid(&P->pid[2], 2, kp2, ki2, kd2, l2);
    set_pid(&P->pid[3], 2, kp3, z3, z3, l3);
    const double zvkp = 3.0, zvki = 0.8, zvkd = 0.2, zvl = 1.0;
    const double zpkp = 1.0, zero = 0.0, zpl = 3.0;
    set_pid(&P->zpid[0], 1, &zvkp, &zvki, &zvkd, &zvl);
    set_pid(&P->zpid[1], 1, &zpkp, &zero, &zero, &zpl);
  }
  orc_finalize(P);
}

void orc_params_fixedwing(orc_params* P) {
  memset(P, 0, sizeof(*P));
  P->vehicle = ORC_FIXEDWING;
  world_defaults(&P->world);
  /* fixedwing.urdf: point masses at link origins, all link inertias zero */
  const double m[7] = {0.3, 0.1, 0.05, 0.2, 0.2, 0.5, 1.0}; /* base, h-tail, v-tail, L-ail, R-ail, main, fuselage */
  const double r[7][3] = {{0, 0, 0}, {-1.1, 0, 0}, {-1.1, 0, 0.15}, {-0.5, 0.95, 0}, {-0.5, -0.95, 0}, {-0.5, 0, 0}, {-0.45, 0, 0}};
  double M = 0, c[3] = {0, 0, 0};
  for (int i = 0; i < 7; ++i) { M += m[i]; for (int k = 0; k < 3; ++k) c[k] += m[i] * r[i][k]; }
  for (int k = 0; k < 3; ++k) c[k] /= M;
  P->mass = M;
  memcpy(P->com, c, sizeof(c));
  for (int i = 0; i < 7; ++i) {
    double d[3] = {r[i][0] - c[0], r[i][1] - c[1], r[i][2] - c[2]};
    double d2 = dot3(d, d);
    for (int a = 0; a < 3; ++a)
      for (int b = 0; b < 3; ++b) P->I_pa[a][b] += m[i] * ((a == b ? d2 : 0.0) - d[a] * d[b]);
  }
  /* collision boxes: fixedwing.urdf:44-49,70-75,96-101,122-127,148-153,174-179 */
  const double bc[6][3] = {{-1.1, 0, 0}, {-1.1, 0, 0.15}, {-0.5, 0.95, 0}, {-0.5, -0.95, 0}, {-0.5, 0, 0}, {-0.45, 0, 0}};
  const double bs[6][3] = {{0.3, 0.6, 0.05}, {0.3, 0.05, 0.3}, {0.31, 0.3, 0.06}, {0.31, 0.3, 0.06}, {0.3, 1.8, 0.05}, {1.4, 0.2, 0.2}};
  P->n_boxes = 6;
  for (int i = 0; i < 6; ++i)
    for (int k = 0; k < 3; ++k) { P->boxes[i].c[k] = bc[i][k]; P->boxes[i].h[k] = 0.5 * bs[i][k]; }
  /* motor: fixedwing.yaml:1-6, fixedwing.py:147-168 */
  P->n_motors = 1;
  P->thrust_unit[0][0] = 1.0;
  P->thrust_coef[0] = 3.16e-10;
  P->torque_coef[0] = 7.94e-12;
  P->max_rpm[0] = 1.0 * sqrt(18.0 / 3.16e-10);
  P->motor_tau[0] = 0.01;
  P->noise_ratio[0] = 0.02;
  /* surfaces: fixedwing.py:80-138 order, fixedwing.yaml:8-71 */
  P->n_surf = 5;
  const double sr[5][3] = {{-0.5, 0.95, 0}, {-0.5, -0.95, 0}, {-1.1, 0, 0}, {-1.1, 0, 0.15}, {-0.5, 0, 0}};
  const double chord[5] = {0.3, 0.3, 0.2, 0.2, 0.3}, span[5] = {0.3, 0.3, 0.625, 0.312, 1.6};
  const double a0[5] = {-2, -2, 0, 0, -2}, asp[5] = {14, 14, 9, 9, 14}, asn[5] = {-9, -9, -9, -9, -9};
  const double lim[5] = {30, 30, 20, 20, 0};
  for (int i = 0; i < 5; ++i) {
    orc_surface* S = &P->surf[i];
    memcpy(S->r, sr[i], sizeof(S->r));
    S->lift_unit[2] = 1.0;
    if (i == 3) { S->lift_unit[2] = 0.0; S->lift_unit[1] = 1.0; } /* v-tail, fixedwing.py:121 */
    S->drag_unit[0] = 1.0;
    S->Cl_alpha_2D = 6.283; S->chord = chord[i]; S->span = span[i]; S->flap_to_chord = 0.3; S->eta = 0.65;
    S->alpha_0_base = a0[i] * (PI / 180.0);
    S->alpha_stall_P_base = asp[i] * (PI / 180.0);
    S->alpha_stall_N_base = asn[i] * (PI / 180.0);
    S->Cd_0 = 0.01; S->deflection_limit = lim[i]; S->tau = 0.05;
  }
  const int ids[6] = {0, 0, 1, 2, 1, 3};           /* fixedwing.py:143 */
  const double sg[6] = {1, -1, 1, -1, -1, 1};      /* fixedwing.py:144 */
  memcpy(P->assist_ids, ids, sizeof(ids));
  memcpy(P->assist_signs, sg, sizeof(sg));
  P->control_period = 1.0 / 120.0;
  P->start_pos[2] = 10.0;   /* fixedwing_waypoints_env.py:63 */
  P->start_vel[0] = 20.0;   /* fixedwing.py:35 */
  P->settle_steps = 10;
  P->angle_repr = 1;
  orc_finalize(P);
}
/* Fixedwing with drone_model="acrowing" (ma_fixedwing_base_env.py:193-195): models/vehicles/acrowing/acrowing.{urdf,yaml} */
void orc_params_acrowing(orc_params* P) {
  orc_params_fixedwing(P);
  /* acrowing.urdf: base :21, h-tail :44,61, v-tail :70,87, ailerons :96,113 :122,139, main wing :148,165, fuselage :174,191 */
  const double m[7] = {0.3, 0.1, 0.05, 0.2, 0.2, 0.5, 1.0};
  const double r[7][3] = {{0, 0, 0}, {-1.1, 0, 0}, {-1.1, 0, 0.25}, {-0.35, 0.95, 0}, {-0.35, -0.95, 0}, {-0.35, 0, 0}, {-0.45, 0, 0}};
  double M = 0, c[3] = {0, 0, 0};
  for (int i = 0; i < 7; ++i) { M += m[i]; for (int k = 0; k < 3; ++k) c[k] += m[i] * r[i][k]; }
  for (int k = 0; k < 3; ++k) c[k] /= M;
  P->mass = M;
  memcpy(P->com, c, sizeof(c));
  memset(P->I_pa, 0, sizeof(P->I_pa));
  for (int i = 0; i < 7; ++i) {
    double d[3] = {r[i][0] - c[0], r[i][1] - c[1], r[i][2] - c[2]};
    double d2 = dot3(d, d);
    for (int a = 0; a < 3; ++a)
      for (int b = 0; b < 3; ++b) P->I_pa[a][b] += m[i] * ((a == b ? d2 : 0.0) - d[a] * d[b]);
  }
  /* collision boxes :50,76,102,128,154,180 */
  const double bs[6][3] = {{0.3, 0.8, 0.05}, {0.4, 0.05, 0.5}, {0.31, 0.3, 0.06}, {0.31, 0.3, 0.06}, {0.3, 1.8, 0.05}, {1.4, 0.2, 0.2}};
  for (int i = 0; i < 6; ++i)
    for (int k = 0; k < 3; ++k) { P->boxes[i].c[k] = r[1 + i][k]; P->boxes[i].h[k] = 0.5 * bs[i][k]; }
  P->max_rpm[0] = sqrt(30.0 / 3.16e-10); /* acrowing.yaml:2 */
  /* acrowing.yaml:8-71, surface order of fixedwing.py:80-138 */
  const int link_of[5] = {3, 4, 1, 2, 5};
  const double chord[5] = {0.3, 0.3, 0.3, 0.4, 0.3}, span[5] = {0.3, 0.3, 0.8, 0.4, 1.6}, f2c[5] = {0.3, 0.3, 0.5, 0.4, 0.1};
  const double a0[5] = {0, 0, 0, 0, -2}, asp[5] = {16, 16, 11, 11, 16}, asn[5] = {-12, -12, -11, -11, -10};
  const double lim[5] = {30, 30, 20, 35, 15};
  for (int i = 0; i < 5; ++i) {
    orc_surface* S = &P->surf[i];
    memcpy(S->r, r[link_of[i]], sizeof(S->r));
    S->chord = chord[i]; S->span = span[i]; S->flap_to_chord = f2c[i];
    S->alpha_0_base = a0[i] * (PI / 180.0);
    S->alpha_stall_P_base = asp[i] * (PI / 180.0);
    S->alpha_stall_N_base = asn[i] * (PI / 180.0);
    S->deflection_limit = lim[i];
  }
  orc_finalize(P);
}
void orc_task_hover(orc_params* P) { /* quadx_hover_env.py:32-37 */
  P->task = ORC_TASK_HOVER;
  P->flight_mode = 0; P->dome = 3.0; P->max_steps = 400; P->env_step_ratio = 3;
  P->start_pos[0] = 0; P->start_pos[1] = 0; P->start_pos[2] = 1.0;
  P->sparse_reward = 0; P->angle_repr = 1; P->num_targets = 0; P->collide_any = 0; P->throttle_remap = 0;
}
void orc_task_quadx_waypoints(orc_params* P) { /* quadx_waypoints_env.py:38-47,87 */
  P->task = ORC_TASK_WAYPOINTS;
  P->flight_mode = 0; P->dome = 5.0; P->max_steps = 300; P->env_step_ratio = 4;
  P->start_pos[0] = 0; P->start_pos[1] = 0; P->start_pos[2] = 1.0;
  P->sparse_reward = 0; P->angle_repr = 1; P->num_targets = 4; P->goal_reach_distance = 0.2;
  P->min_height = 0.1; P->collide_any = 0; P->throttle_remap = 0;
  P->wp_dist_reward = 0.1; P->wp_yaw_penalty = 0.01;
  P->use_yaw_targets = 0; P->goal_reach_angle = 0.1; /* quadx_waypoints_env.py:40,42 */
}
void orc_task_fixedwing_waypoints(orc_params* P) { /* fixedwing_waypoints_env.py:36-45,63,81 */
  P->task = ORC_TASK_WAYPOINTS;
  P->flight_mode = 0; P->dome = 100.0; P->max_steps = 3600; P->env_step_ratio = 4;
  P->start_pos[0] = 0; P->start_pos[1] = 0; P->start_pos[2] = 10.0;
  P->sparse_reward = 0; P->angle_repr = 1; P->num_targets = 4; P->goal_reach_distance = 2.0;
  P->min_height = 0.5; P->collide_any = 1; P->throttle_remap = 1;
  P->wp_dist_reward = 1.0; P->wp_yaw_penalty = 0.0;
  P->use_yaw_targets = 0; P->goal_reach_angle = INFINITY; /* fixedwing_waypoints_env.py:77-79 */
}

void orc_task_ma_hover(orc_params* P) { /* pz_envs/quadx_envs/ma_quadx_hover_env.py:36-52 */
  P->task = ORC_TASK_MA_HOVER;
  P->flight_mode = 0; P->dome = 10.0; P->max_steps = 1200; P->env_step_ratio = 3;
  P->sparse_reward = 0; P->angle_repr = 1; P->num_targets = 0; P->collide_any = 0; P->throttle_remap = 0;
}

/* ------------------------------------------------------------------ lane level */
/* quadx.py:512-535 / fixedwing.py:266-291 */
void orc_update_state(const orc_params* P, orc_lane* L) {
  double R[3][3];
  orc_matrix_from_quat(L->q, R);
  matTvec(R, L->v, L->v_b);   /* rotation = R^T (quadx.py:521-523) */
  matTvec(R, L->w, L->w_b);
  orc_euler_from_quat(L->q, L->rpy);
  /* Aviary.elapsed_time is refreshed after update_state (aviary.py:528-529), so the wind is sampled
   * at the time of the previous tick's end: physics_steps / physics_hz with physics_steps not yet
   * incremented for this tick. */
  const double now = (double)L->physics_steps * P->world.dt;
  if (P->vehicle == ORC_QUADX || P->vehicle == ORC_ROCKET) { /* boring_bodies.py:78-111: one body at the base origin (QuadX centre-of-mass link; Rocket fuel tank link, rocket.py:97) */
    double lv[3] = {L->v[0], L->v[1], L->v[2]};
    if (P->wind_fn) {
      double wnd[3];
      P->wind_fn(now, L->p, 1, wnd);
      for (int k = 0; k < 3; ++k) lv[k] -= wnd[k];
    }
    matTvec(R, lv, L->drag_v_b);
  }
  double pos[ORC_MAX_SURF][3], wnd[ORC_MAX_SURF][3];
  if (P->n_surf > 0 && P->wind_fn) {
    for (int i = 0; i < P->n_surf; ++i) {
      double rw[3];
      matvec(R, P->surf[i].r, rw);
      for (int k = 0; k < 3; ++k) pos[i][k] = L->p[k] + rw[k];
    }
    P->wind_fn(now, &pos[0][0], P->n_surf, &wnd[0][0]);
  }
  for (int i = 0; i < P->n_surf; ++i) { /* lifting_surfaces.py:73-110 */
    double rw[3], wxr[3], lv[3];
    matvec(R, P->surf[i].r, rw);
    cross3(L->w, rw, wxr);
    for (int k = 0; k < 3; ++k) lv[k] = L->v[k] + wxr[k] - (P->wind_fn ? wnd[i][k] : 0.0);
    matTvec(R, lv, L->surf_v[i]);
  }
}
/* quadx.py:233-373 ; fixedwing.py:206-227 */
void orc_set_mode(const orc_params* P, orc_lane* L, int mode) {
  L->mode = mode;
  if (P->vehicle == ORC_ROCKET) return; /* base_drone.py:243-259: only records the mode */
  if (P->vehicle == ORC_FIXEDWING) {
    for (int i = 0; i < 6; ++i) L->setpoint[i] = 0.0;
    return;
  }
  if (mode == -1) return;
  for (int i = 0; i < 6; ++i) L->setpoint[i] = 0.0;
  if (mode == 0) {
    L->setpoint[3] = -1.0;
  } else if (mode == 1 || mode == 5 || mode == 6) {
  } else if (mode == 7) {
    L->setpoint[0] = L->p[0]; L->setpoint[1] = L->p[1]; L->setpoint[2] = L->rpy[2]; L->setpoint[3] = L->p[2];
  } else {
    L->setpoint[3] = L->p[2];
  }
  /* fresh PID objects (z_PIDs are NOT re-created, quadx.py:206) */
  memset(L->pid_I, 0, sizeof(L->pid_I));
  memset(L->pid_E, 0, sizeof(L->pid_E));
}
/* aviary.py:218-312 + quadx.py:222-231 / fixedwing.py:194-204 */
void orc_aviary_reset(const orc_params* P, orc_lane* L, uint64_t lane_id) {
  uint32_t ctr = L->rng_ctr, rkey = L->reset_key;
  double keep[8];
  memcpy(keep, L->action, sizeof(double) * 4);
  memcpy(keep + 4, L->past_action, sizeof(double) * 4);
  memset(L, 0, sizeof(*L));
  L->rng_ctr = ctr;
  L->reset_key = rkey;
  if (P->task == ORC_TASK_MA_HOVER) { /* current/past actions are created once in __init__ (ma_quadx_base_env.py:139-150) */
    memcpy(L->action, keep, sizeof(double) * 4);
    memcpy(L->past_action, keep + 4, sizeof(double) * 4);
  }
  L->lane_id = lane_id;
  for (int k = 0; k < 3; ++k) { L->p[k] = P->start_pos[k]; L->v[k] = P->start_vel[k]; }
  orc_quat_from_euler(P->start_rpy, L->q); /* base_drone.py:115 */
  orc_set_mode(P, L, 0);
  for (int i = 0; i < 8; ++i) L->setpoint[i] = 0.0; /* quadx.py:225, rocket.py:228-229 */
  if (P->vehicle == ORC_ROCKET) L->fuel_ratio = P->starting_fuel_ratio; /* rocket.py:236, boosters.py:119-130 */
  memset(L->zpid_I, 0, sizeof(L->zpid_I));
  memset(L->zpid_E, 0, sizeof(L->zpid_E));
  orc_update_state(P, L);
}

static double tick_noise(const orc_params* P, const orc_lane* L, const double* xi, int t, uint32_t flat, uint32_t stream) {
  if (P->noise_mode == ORC_NOISE_OFF) return 0.0;
  /* np_random.normal(*throttle.shape) == normal(loc=num_motors, scale=1): motors.py:135.
   * Injected samples are the raw draws xi ~ N(num_motors, 1). */
  if (P->noise_mode == ORC_NOISE_INJECT) return xi[t];
  return (double)P->n_motors + lane_normal(P, L, flat, stream);
}

/* The composite of the Rocket's links with the fuel tank's current mass and inertia (the reference
 * rewrites them every tick: boosters.py:193-198 -> changeDynamics). */
static void rocket_composite(const orc_params* P, double fuel_ratio, orc_body* B) {
  double M = 0.0, mc[3] = {0, 0, 0};
  memset(B, 0, sizeof(*B));
  for (int i = 0; i < P->n_links; ++i) {
    const double m = (i == P->fueltank_link) ? fuel_ratio * P->total_fuel : P->link_mass[i];
    M += m;
    for (int k = 0; k < 3; ++k) mc[k] += m * P->link_r[i][k];
  }
  B->mass = M;
  for (int k = 0; k < 3; ++k) B->com[k] = mc[k] / M;
  for (int i = 0; i < P->n_links; ++i) {
    const int tank = (i == P->fueltank_link);
    const double m = tank ? fuel_ratio * P->total_fuel : P->link_mass[i];
    double d[3];
    for (int k = 0; k < 3; ++k) d[k] = P->link_r[i][k] - B->com[k];
    const double dd = dot3(d, d);
    for (int a = 0; a < 3; ++a) {
      B->I_own[a][a] += tank ? fuel_ratio * P->fuel_inertia[a] : P->link_I[i][a];
      for (int b = 0; b < 3; ++b) B->I_pa[a][b] += m * ((a == b ? dd : 0.0) - d[a] * d[b]);
    }
  }
  double I[3][3];
  for (int a = 0; a < 3; ++a)
    for (int b = 0; b < 3; ++b) I[a][b] = B->I_own[a][b] + B->I_pa[a][b];
  inv3(I, B->I_inv);
}
/* rocket.py:268-290 update_physics: body drag, grid fins, gimbal, booster (+ fuel tank mass properties) */
static void rocket_physics(const orc_params* P, orc_lane* L, double noise, double F_b[3], double T_b[3], orc_body* body) {
  const double dt = P->world.dt;
  double Fd[3];
  orc_body_drag(P, L->drag_v_b, Fd); /* rocket.py:271, the fuel tank link at the base origin */
  for (int k = 0; k < 3; ++k) F_b[k] += Fd[k];
  for (int i = 0; i < P->n_surf; ++i) { /* rocket.py:274, lifting_surfaces.py:266-324 */
    const orc_surface* S = &P->surf[i];
    L->actuation[i] += (dt / S->tau) * (L->cmd[i] - L->actuation[i]);
    double F[3], T[3], rxf[3];
    orc_surface_force(S, L->surf_v[i], L->actuation[i], F, T);
    cross3(S->r, F, rxf);
    for (int k = 0; k < 3; ++k) { F_b[k] += F[k]; T_b[k] += rxf[k] + T[k]; }
  }
  /* gimbals.py:151-217: first-order lag, then R = R_x(a1) R_y(a2) by Rodrigues' formula */
  for (int k = 0; k < 2; ++k) L->gimbal[k] += (dt / P->gimbal_tau) * (L->cmd[6 + k] - L->gimbal[k]);
  const double a1 = L->gimbal[0] * P->gimbal_range_rad, a2 = L->gimbal[1] * P->gimbal_range_rad;
  const double s1 = sin(a1), c1 = 1.0 - 2.0 * sin(a1 / 2.0) * sin(a1 / 2.0);
  const double s2 = sin(a2), c2 = 1.0 - 2.0 * sin(a2 / 2.0) * sin(a2 / 2.0);
  const double dir[3] = {s2, -s1 * c2, c1 * c2}; /* R1 R2 (0,0,1) */
  /* boosters.py:213-263 */
  const double ratio_min = P->min_thrust / P->max_thrust;
  L->ignition = ((!P->reignitable) && L->ignition) || (L->cmd[4] > 0.5);
  const double target = L->ignition ? (L->cmd[5] * (1.0 - ratio_min) + ratio_min) : 0.0;
  double thr = L->throttle[0];
  thr += (dt / P->booster_tau) * (target - thr);
  thr += noise * thr * P->booster_noise;
  thr *= (L->fuel_ratio > 0.0) ? 1.0 : 0.0;
  L->throttle[0] = thr;
  L->fuel_ratio = clipd(L->fuel_ratio - thr * (P->max_fuel_rate / P->total_fuel) * dt, 0.0, 1.0);
  const double thrust = thr * P->max_thrust;
  double F[3] = {dir[0] * thrust, dir[1] * thrust, dir[2] * thrust}, rxf[3];
  cross3(P->link_r[P->booster_link], F, rxf);
  for (int k = 0; k < 3; ++k) { F_b[k] += F[k]; T_b[k] += rxf[k]; }
  rocket_composite(P, L->fuel_ratio, body);
}

/* models/vehicles/rocket/rocket.{urdf,yaml}; drones/rocket.py:84-219 */
void orc_params_rocket(orc_params* P) {
  memset(P, 0, sizeof(*P));
  P->vehicle = ORC_ROCKET;
  world_defaults(&P->world);
  /* rocket.urdf: base :37-38, fuel tank :58-59,69, booster :78-79,95, fins :104,121 :130,147 :156,173 :182,199,
   * legs :208,225 :234,251 :260,277, flame :285,296 (massless links still listed: they carry collision shapes) */
  const double lm[10] = {91.0, 410.9, 47.0, 0.05, 0.05, 0.05, 0.05, 0.0, 0.0, 0.0};
  const double lr[10][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, -2}, {0.35, 0, 2.051}, {-0.35, 0, 2.051}, {0, 0.35, 2.051},
                            {0, -0.35, 2.051}, {0.0, 0.35, -2.4}, {0.3031, -0.175, -2.4}, {-0.3031, -0.175, -2.4}};
  const double li[10][3] = {{372.6, 372.6, 1.55}, {1678, 1678, 7.01}, {192.43, 192.43, 0.81}};
  P->n_links = 10;
  for (int i = 0; i < 10; ++i) {
    P->link_mass[i] = lm[i];
    for (int k = 0; k < 3; ++k) { P->link_r[i][k] = lr[i][k]; P->link_I[i][k] = li[i][k]; }
  }
  P->fueltank_link = 1; P->booster_link = 2;
  /* collision shapes: base cylinder :43 (the fuel tank's :64 lies inside it), booster cylinder :84, fin boxes
   * :110,136,162,188, leg boxes :214,240,266 with the legs' yaw :251,277 */
  int k = 0;
  P->boxes[k].kind = 1; P->boxes[k].h[0] = P->boxes[k].h[1] = 0.185; P->boxes[k].h[2] = 4.77 / 2; ++k;
  P->boxes[k].kind = 1; P->boxes[k].c[2] = -2.0; P->boxes[k].h[0] = P->boxes[k].h[1] = 0.25; P->boxes[k].h[2] = 0.25; ++k;
  for (int f = 0; f < 4; ++f, ++k) {
    for (int a = 0; a < 3; ++a) P->boxes[k].c[a] = lr[3 + f][a];
    P->boxes[k].h[0] = f < 2 ? 0.15 : 0.015; P->boxes[k].h[1] = f < 2 ? 0.015 : 0.15; P->boxes[k].h[2] = 0.15;
  }
  const double yaw[3] = {0.0, 4.188, -4.188};
  for (int g = 0; g < 3; ++g, ++k) {
    for (int a = 0; a < 3; ++a) P->boxes[k].c[a] = lr[7 + g][a];
    P->boxes[k].h[0] = 0.025; P->boxes[k].h[1] = 0.25; P->boxes[k].h[2] = 0.025;
    P->boxes[k].yaw = yaw[g];
  }
  P->n_boxes = k;
  /* rocket.yaml:7-19 */
  P->total_fuel = 410.9; P->max_fuel_rate = 1.451;
  P->fuel_inertia[0] = 1678; P->fuel_inertia[1] = 1678; P->fuel_inertia[2] = 7.01;
  P->min_thrust = 2966.7; P->max_thrust = 7607.0; P->reignitable = 1;
  P->gimbal_range_rad = 5.0 * PI / 180.0; P->booster_tau = 0.01; P->gimbal_tau = 0.01; P->booster_noise = 0.01;
  P->n_motors = 1; /* np_random.normal(*throttle.shape) with one booster: loc = 1 (boosters.py:236-240) */
  /* rocket.yaml:21-32, rocket.py:114-142: "x fins" lift along y, "y fins" lift along x, all facing -z.
   * The reference binds the four LiftingSurface objects to link ids 0..3 (surface_id=finlet_id), which
   * in rocket.urdf's joint order are the fuel tank, the booster, fin_pos_x and fin_neg_x -- not the four
   * fin links 2..5. That is where their velocities are sampled and their forces applied; reproduced. */
  P->n_surf = 4;
  for (int i = 0; i < 4; ++i) {
    orc_surface* S = &P->surf[i];
    for (int a = 0; a < 3; ++a) S->r[a] = lr[1 + i][a];
    S->lift_unit[i < 2 ? 1 : 0] = 1.0;
    S->drag_unit[2] = -1.0;
    S->Cl_alpha_2D = 6.283; S->chord = 0.5; S->span = 0.5; S->flap_to_chord = 1.0; S->eta = 0.65;
    S->alpha_0_base = 0.0; S->alpha_stall_P_base = 20.0 * PI / 180.0; S->alpha_stall_N_base = -20.0 * PI / 180.0;
    S->Cd_0 = 0.01; S->deflection_limit = 45.0; S->tau = 0.05;
  }
  const double fm[4][3] = {{0, 1, 1}, {0, 1, -1}, {1, 0, -1}, {1, 0, 1}}; /* rocket.py:152-159 */
  memcpy(P->finlet_map, fm, sizeof(fm));
  /* rocket.yaml:34-40, boring_bodies.py:63 */
  P->drag_const[0] = 0.5 * 1.225 * 1.16 * 1.7649; P->drag_const[1] = P->drag_const[0];
  P->drag_const[2] = 0.5 * 1.225 * 2.0 * 0.1075;
  P->control_period = 1.0 / 120.0; /* rocket.py:36 */
  P->starting_fuel_ratio = 0.05;   /* rocket.py:47 */
  P->start_pos[2] = 1.0;
  P->angle_repr = 1;
  /* mass / com / inertia of the full-tank composite, for orc_finalize()'s bookkeeping only */
  orc_body B;
  rocket_composite(P, 1.0, &B);
  P->mass = B.mass;
  memcpy(P->com, B.com, sizeof(B.com));
  memcpy(P->I_own, B.I_own, sizeof(B.I_own));
  memcpy(P->I_pa, B.I_pa, sizeof(B.I_pa));
  orc_finalize(P);
}

/* aviary.py:480-531 */
static void aviary_tick_one(const orc_params* P, orc_lane* L, const double* xi, int t, uint32_t flat_base, uint32_t stream);
void orc_aviary_step(const orc_params* P, orc_lane* L, const double* xi, uint32_t flat_base, uint32_t stream) {
  L->contact_step = 0; /* :507 */
  for (int t = 0; t < P->world.ticks_per_control; ++t) aviary_tick_one(P, L, xi, t, flat_base, stream);
}
/* one physics tick of Aviary.step for one drone (aviary.py:510-525): control, forces, stepSimulation, update_state -- in two
 * halves, so that a shared world can run its drone-drone stage between the velocity update and the ground solve of all its
 * bodies (orc_world_aviary_step) */
static void aviary_tick_pre(const orc_params* P, orc_lane* L, const double* xi, int t, uint32_t flat_base, uint32_t stream, orc_body* body) {
  {
    /* update_control */
    if (L->physics_steps % P->world.ticks_per_control == 0) {
      if (P->vehicle == ORC_QUADX) {
        orc_quadx_control(P, L);
      } else if (P->vehicle == ORC_ROCKET) { /* rocket.py:249-257 */
        for (int i = 0; i < 4; ++i) {
          double c = P->finlet_map[i][0] * L->setpoint[0] + P->finlet_map[i][1] * L->setpoint[1] + P->finlet_map[i][2] * L->setpoint[2];
          L->cmd[i] = clipd(c, -1.0, 1.0);
        }
        for (int i = 0; i < 4; ++i) L->cmd[4 + i] = L->setpoint[3 + i];
      } else if (L->mode == -1) { /* fixedwing.py:241-243 */
        for (int i = 0; i < 6; ++i) L->cmd[i] = L->setpoint[i];
      } else { /* fixedwing.py:246-250 */
        for (int i = 0; i < 6; ++i) L->cmd[i] = L->setpoint[P->assist_ids[i]] * P->assist_signs[i];
      }
    }
    /* update_physics */
    double F_b[3] = {0, 0, 0}, T_b[3] = {0, 0, 0};
    double thrust[4][3], torque[4][3];
    orc_body rocket_body;
    double noise = tick_noise(P, L, xi, t, flat_base + (uint32_t)t, stream);
    if (P->vehicle == ORC_QUADX) {
      double Fd[3];
      orc_body_drag(P, L->drag_v_b, Fd); /* quadx.py:498, body link at the origin */
      for (int k = 0; k < 3; ++k) F_b[k] += Fd[k];
      orc_motors_update(P, L->throttle, L->pwm, noise, thrust, torque); /* quadx.py:499 */
      for (int i = 0; i < 4; ++i) {
        double rxf[3];
        cross3(P->motor_r[i], thrust[i], rxf);
        for (int k = 0; k < 3; ++k) { F_b[k] += thrust[i][k]; T_b[k] += rxf[k] + torque[i][k]; }
      }
      if (!(L->contact_now || L->world_contact)) { /* quadx.py:502-510: no contact point anywhere in the world */
        for (int k = 0; k < 3; ++k) T_b[k] += -sgn(L->w_b[k]) * P->drag_coef_pqr * (L->w_b[k] * L->w_b[k]);
      }
    } else if (P->vehicle == ORC_ROCKET) {
      rocket_physics(P, L, noise, F_b, T_b, &rocket_body);
    } else {
      for (int i = 0; i < P->n_surf; ++i) { /* fixedwing.py:263, lifting_surfaces.py:266-324 */
        const orc_surface* S = &P->surf[i];
        L->actuation[i] += (P->world.dt / S->tau) * (L->cmd[i] - L->actuation[i]);
        double F[3], T[3], rxf[3];
        orc_surface_force(S, L->surf_v[i], L->actuation[i], F, T);
        cross3(S->r, F, rxf);
        for (int k = 0; k < 3; ++k) { F_b[k] += F[k]; T_b[k] += rxf[k] + T[k]; }
      }
      orc_motors_update(P, L->throttle, &L->cmd[5], noise, thrust, torque); /* fixedwing.py:264 */
      double rxf[3];
      cross3(P->motor_r[0], thrust[0], rxf);
      for (int k = 0; k < 3; ++k) { F_b[k] += thrust[0][k]; T_b[k] += rxf[k] + torque[0][k]; }
    }
    /* stepSimulation: collision detection at the pre-integration pose, then the free-body tick's velocity half */
    const int persisted = L->contact_now; /* contact points left by the previous stepSimulation */
    L->contact_now = contact_plane_reach(P, L->p, L->q, report_reach(&P->world, persisted)) || L->peer_contact;
    if (P->vehicle == ORC_ROCKET) {
      *body = rocket_body;
      body->persisted = persisted;
    } else {
      body->persisted = persisted;
      body->mass = P->mass;
      memcpy(body->com, P->com, sizeof(body->com));
      memcpy(body->I_own, P->I_own, sizeof(body->I_own));
      memcpy(body->I_pa, P->I_pa, sizeof(body->I_pa));
      memcpy(body->I_inv, P->I_inv, sizeof(body->I_inv));
    }
    rigid_tick_vel(P, body, L->q, L->v, L->w, F_b, T_b);
  }
}
static void aviary_tick_post(const orc_params* P, orc_lane* L, const orc_body* body, const double* shift) {
  rigid_tick_pos(P, body, L->p, L->q, L->v, L->w, shift);
  orc_update_state(P, L);
  if (L->contact_now) L->contact_step = 1; /* :523-525 */
  L->physics_steps += 1;
}
static void aviary_tick_one(const orc_params* P, orc_lane* L, const double* xi, int t, uint32_t flat_base, uint32_t stream) {
  orc_body body;
  aviary_tick_pre(P, L, xi, t, flat_base, stream, &body);
  aviary_tick_post(P, L, &body, 0);
}

/* ---- shared world ---- */
/* box k of drone a against box l of drone b: the 15-axis verdict with b's frame as the axis-aligned one */
static int drones_overlap(const orc_params* Pa, const orc_lane* La, const orc_params* Pb, const orc_lane* Lb) {
  double d[3] = {La->p[0] - Lb->p[0], La->p[1] - Lb->p[1], La->p[2] - Lb->p[2]};
  /* (contact_now: still the previous tick's here -- a pair one of whose bodies holds contact points is reported up to the breaking distance) */
  const double rd = report_reach(&Pa->world, La->contact_now || Lb->contact_now);
  const double rr = Pa->bound_radius + Pb->bound_radius + 1.7320508075688772 * rd; /* (the enlarged box's corner) */
  if (dot3(d, d) > rr * rr) return 0; /* bounding spheres apart */
  double Ra[3][3], Rb[3][3];
  orc_matrix_from_quat(La->q, Ra);
  orc_matrix_from_quat(Lb->q, Rb);
  for (int k = 0; k < Pa->n_boxes; ++k) {
    for (int l = 0; l < Pb->n_boxes; ++l) {
      double oa[3], ob[3], ca[3], rel[3], Rrel[3][3];
      matvec(Ra, Pa->boxes[k].c, oa);
      matvec(Rb, Pb->boxes[l].c, ob);
      for (int i = 0; i < 3; ++i) ca[i] = (La->p[i] + oa[i]) - (Lb->p[i] + ob[i]);
      matTvec(Rb, ca, rel); /* a's box centre in b's box frame */
      for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) { Rrel[i][j] = 0.0; for (int m = 0; m < 3; ++m) Rrel[i][j] += Rb[m][i] * Ra[m][j]; }
      const double zero[3] = {0.0, 0.0, 0.0};
      const double hbr[3] = {Pb->boxes[l].h[0] + rd, Pb->boxes[l].h[1] + rd, Pb->boxes[l].h[2] + rd};
      if (orc_box_box_overlap(rel, Rrel, Pa->boxes[k].h, zero, hbr)) return 1;
    }
  }
  return 0;
}
/* ---- drone-drone contact response (the model: orc_world.pair_response in the header) ---- */
typedef struct {
  int a, b;
  double ra[3], rb[3], dir[3][3], depth, lam[3], un0;
} orc_pair_contact;
static void plane_space(const double n[3], double p[3], double q[3]) { /* btPlaneSpace1 */
  if (fabs(n[2]) > 0.7071067811865475244) {
    double a = n[1] * n[1] + n[2] * n[2], k = 1.0 / sqrt(a);
    p[0] = 0.0; p[1] = -n[2] * k; p[2] = n[1] * k;
    q[0] = a * k; q[1] = -n[0] * p[2]; q[2] = n[0] * p[1];
  } else {
    double a = n[0] * n[0] + n[1] * n[1], k = 1.0 / sqrt(a);
    p[0] = -n[1] * k; p[1] = n[0] * k; p[2] = 0.0;
    q[0] = -n[2] * p[1]; q[1] = n[2] * p[0]; q[2] = a * k;
  }
}
static int pair_contacts(const orc_params* const* Pl, orc_lane* const* Ll, const orc_body* B, int A, orc_pair_contact* pc) {
  int n = 0;
  for (int a = 0; a < A; ++a) {
    for (int b = 0; b < A; ++b) {
      if (b == a) continue;
      const orc_params *Pa = Pl[a], *Pb = Pl[b];
      const orc_lane *La = Ll[a], *Lb = Ll[b];
      const double margin = contact_reach(&Pa->world, B[a].persisted || B[b].persisted);
      double d[3] = {La->p[0] - Lb->p[0], La->p[1] - Lb->p[1], La->p[2] - Lb->p[2]};
      const double rr = Pa->bound_radius + Pb->bound_radius + 2.0 * margin; /* (pruning only: conservative) */
      if (dot3(d, d) > rr * rr) continue;
      double Ra[3][3], Rb[3][3], ca_w[3], cb_w[3];
      orc_matrix_from_quat(La->q, Ra);
      orc_matrix_from_quat(Lb->q, Rb);
      matvec(Ra, B[a].com, ca_w);
      matvec(Rb, B[b].com, cb_w);
      for (int ka = 0; ka < Pa->n_boxes; ++ka) {
        for (int kb = 0; kb < Pb->n_boxes; ++kb) {
          if (Pa->boxes[ka].kind != 0 || Pb->boxes[kb].kind != 0) continue;
          double ob[3], cbox[3];
          matvec(Rb, Pb->boxes[kb].c, ob);
          for (int i = 0; i < 3; ++i) cbox[i] = Lb->p[i] + ob[i];
          for (int vi = 0; vi < 8; ++vi) {
            const double* h = Pa->boxes[ka].h;
            double l[3] = {Pa->boxes[ka].c[0] + ((vi & 1) ? h[0] : -h[0]), Pa->boxes[ka].c[1] + ((vi & 2) ? h[1] : -h[1]),
                           Pa->boxes[ka].c[2] + ((vi & 4) ? h[2] : -h[2])};
            double off[3], x[3], rel[3], loc[3];
            matvec(Ra, l, off);
            for (int i = 0; i < 3; ++i) { x[i] = La->p[i] + off[i]; rel[i] = x[i] - cbox[i]; }
            matTvec(Rb, rel, loc);
            int ks = 0;
            double pen = Pb->boxes[kb].h[0] - fabs(loc[0]);
            for (int k = 1; k < 3; ++k) {
              double pk = Pb->boxes[kb].h[k] - fabs(loc[k]);
              if (pk < pen) { pen = pk; ks = k; }
            }
            if (pen < -margin || n >= ORC_MAX_PAIR_CONTACTS) continue;
            orc_pair_contact* c = &pc[n++];
            c->a = a; c->b = b; c->depth = pen;
            const double sg = loc[ks] < 0.0 ? -1.0 : 1.0;
            for (int i = 0; i < 3; ++i) {
              c->dir[0][i] = sg * Rb[i][ks];
              c->ra[i] = x[i] - (La->p[i] + ca_w[i]);
              c->rb[i] = x[i] - (Lb->p[i] + cb_w[i]);
            }
            plane_space(c->dir[0], c->dir[1], c->dir[2]);
            c->lam[0] = c->lam[1] = c->lam[2] = 0.0;
          }
        }
      }
    }
  }
  return n;
}
/* the pair stage: impulses between the bodies of a world on their post-force velocities; shift[i]: the position-level recovery of body i */
static void pair_stage(const orc_params* const* Pl, orc_lane* const* Ll, const orc_body* B, int A, double shift[][3]) {
  const orc_world* W = &Pl[0]->world;
  orc_pair_contact pc[ORC_MAX_PAIR_CONTACTS];
  const int n = pair_contacts(Pl, Ll, B, A, pc);
  if (n == 0) return;
  double Iw[ORC_MAX_WORLD][3][3], cw[ORC_MAX_WORLD][3], vc[ORC_MAX_WORLD][3], t[3];
  for (int i = 0; i < A; ++i) {
    double R[3][3], tmp[3][3];
    orc_matrix_from_quat(Ll[i]->q, R);
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) { tmp[r][c] = 0; for (int k = 0; k < 3; ++k) tmp[r][c] += R[r][k] * B[i].I_inv[k][c]; }
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) { Iw[i][r][c] = 0; for (int k = 0; k < 3; ++k) Iw[i][r][c] += tmp[r][k] * R[c][k]; }
    matvec(R, B[i].com, cw[i]);
    cross3(Ll[i]->w, cw[i], t);
    for (int k = 0; k < 3; ++k) vc[i][k] = Ll[i]->v[k] + t[k];
  }
  for (int c = 0; c < n; ++c) {
    double ta[3], tb[3];
    cross3(Ll[pc[c].a]->w, pc[c].ra, ta);
    cross3(Ll[pc[c].b]->w, pc[c].rb, tb);
    pc[c].un0 = 0.0;
    for (int k = 0; k < 3; ++k) pc[c].un0 += ((vc[pc[c].a][k] + ta[k]) - (vc[pc[c].b][k] + tb[k])) * pc[c].dir[0][k];
  }
  const double mu = W->contact_friction * W->contact_friction;
  for (int it = 0; it < W->contact_iters; ++it) {
    double res2 = 0.0;
    for (int c = 0; c < n; ++c) {
      const int a = pc[c].a, b = pc[c].b;
      const double ima = 1.0 / B[a].mass, imb = 1.0 / B[b].mass;
      for (int d = 0; d < 3; ++d) {
        const double* dir = pc[c].dir[d];
        double rxd[3], anga[3], angb[3], axr[3], bxr[3], ta[3], tb[3];
        cross3(pc[c].ra, dir, rxd); matvec(Iw[a], rxd, anga); cross3(anga, pc[c].ra, axr);
        cross3(pc[c].rb, dir, rxd); matvec(Iw[b], rxd, angb); cross3(angb, pc[c].rb, bxr);
        const double k = ima + imb + dot3(dir, axr) + dot3(dir, bxr);
        cross3(Ll[a]->w, pc[c].ra, ta);
        cross3(Ll[b]->w, pc[c].rb, tb);
        double u = 0.0;
        for (int i = 0; i < 3; ++i) u += ((vc[a][i] + ta[i]) - (vc[b][i] + tb[i])) * dir[i];
        double target = 0.0;
        if (d == 0) target = pc[c].depth < W->contact_slop ? (pc[c].depth - W->contact_slop) / W->dt : (pc[c].un0 < 0.0 ? -W->contact_restitution * pc[c].un0 : 0.0);
        double nl = pc[c].lam[d] + (target - u) / k;
        if (d == 0) {
          if (nl < 0.0) nl = 0.0;
        } else {
          const double lim = mu * pc[c].lam[0];
          if (nl > lim) nl = lim;
          if (nl < -lim) nl = -lim;
        }
        const double dl = nl - pc[c].lam[d];
        pc[c].lam[d] = nl;
        for (int i = 0; i < 3; ++i) {
          vc[a][i] += ima * dl * dir[i]; Ll[a]->w[i] += dl * anga[i];
          vc[b][i] -= imb * dl * dir[i]; Ll[b]->w[i] -= dl * angb[i];
        }
        if ((dl * k) * (dl * k) > res2) res2 = (dl * k) * (dl * k);
      }
    }
    if (res2 <= W->contact_residual_threshold) break;
  }
  for (int i = 0; i < A; ++i) {
    cross3(Ll[i]->w, cw[i], t);
    for (int k = 0; k < 3; ++k) Ll[i]->v[k] = vc[i][k] - t[k];
  }
  /* position-level recovery: each body follows its deepest pair contact */
  double best[ORC_MAX_WORLD];
  for (int i = 0; i < A; ++i) best[i] = 0.0;
  for (int c = 0; c < n; ++c) {
    const double e = pc[c].depth - W->contact_slop;
    if (e <= 0.0) continue;
    if (e > best[pc[c].a]) { best[pc[c].a] = e; for (int k = 0; k < 3; ++k) shift[pc[c].a][k] = 0.5 * W->contact_erp * e * pc[c].dir[0][k]; }
    if (e > best[pc[c].b]) { best[pc[c].b] = e; for (int k = 0; k < 3; ++k) shift[pc[c].b][k] = -0.5 * W->contact_erp * e * pc[c].dir[0][k]; }
  }
}
void orc_world_aviary_step(const orc_params* const* Pl, orc_lane* const* Ll, int A, const double* const* xi,
                           uint32_t flat_base, uint32_t stream) {
  for (int i = 0; i < A; ++i) Ll[i]->contact_step = 0;
  const int tpc = Pl[0]->world.ticks_per_control;
  for (int t = 0; t < tpc; ++t) {
    int world_prev = 0; /* contact points left by the previous stepSimulation, anywhere in the world */
    for (int i = 0; i < A; ++i) world_prev |= Ll[i]->contact_now;
    for (int i = 0; i < A; ++i) { Ll[i]->world_contact = world_prev; Ll[i]->peer_contact = 0; }
    for (int i = 0; i < A; ++i) /* this tick's collision detection between the drones, at the pre-integration poses */
      for (int j = i + 1; j < A; ++j)
        if (drones_overlap(Pl[i], Ll[i], Pl[j], Ll[j])) { Ll[i]->peer_contact = 1; Ll[j]->peer_contact = 1; }
    /* stepSimulation for the whole world: forces and new velocities of every body, the drone-drone stage, then every body's
     * ground solve and integration */
    orc_body B[ORC_MAX_WORLD];
    double shift[ORC_MAX_WORLD][3];
    for (int i = 0; i < A; ++i) { shift[i][0] = shift[i][1] = shift[i][2] = 0.0; aviary_tick_pre(Pl[i], Ll[i], xi ? xi[i] : 0, t, flat_base, stream, &B[i]); }
    if (Pl[0]->world.pair_response && Pl[0]->world.contact_response) pair_stage(Pl, Ll, B, A, shift);
    for (int i = 0; i < A; ++i) aviary_tick_post(Pl[i], Ll[i], &B[i], shift[i]);
  }
  for (int i = 0; i < A; ++i) Ll[i]->peer_contact = 0;
}

/* gym_envs/utils/waypoint_handler.py:53-89 ; injected draws: u[0:n]=theta, u[n:2n]=phi,
 * u[2n:3n]=dist (already scaled, in the reference's draw order :72-75) */
static void sample_targets(const orc_params* P, orc_lane* L, const double* u_inj) {
  int n = P->num_targets;
  L->n_targets_left = n;
  L->new_dist = INFINITY; L->old_dist = INFINITY;
  for (int i = 0; i < n; ++i) {
    double theta, phi, dist;
    if (u_inj == 0) {
      theta = (2.0 * PI) * lane_uniform(P, L, (uint32_t)i, 2);
      phi = (2.0 * PI) * lane_uniform(P, L, (uint32_t)(n + i), 2);
      dist = 1.0 + (P->dome * 0.9 - 1.0) * lane_uniform(P, L, (uint32_t)(2 * n + i), 2);
    } else {
      theta = u_inj[i]; phi = u_inj[n + i]; dist = u_inj[2 * n + i];
    }
    double x = dist * sin(phi) * cos(theta), y = dist * sin(phi) * sin(theta), z = fabs(dist * cos(phi));
    L->targets[i][0] = x; L->targets[i][1] = y; L->targets[i][2] = z > P->min_height ? z : P->min_height;
  }
  if (P->use_yaw_targets) { /* waypoint_handler.py:85-89: drawn after all the positions */
    for (int i = 0; i < n; ++i)
      L->yaw_targets[i] = u_inj ? u_inj[3 * n + i] : -PI + (2.0 * PI) * lane_uniform(P, L, (uint32_t)(3 * n + i), 2);
  }
  L->yaw_error_scalar = 0.0;
}
/* waypoint_handler.py:117-158: distance_to_targets; column 3 = yaw error when use_yaw_targets (:144-156) */
static void compute_deltas(const orc_params* P, orc_lane* L, double deltas[][4]) {
  double qe[4], R[3][3];
  orc_quat_from_euler(L->rpy, qe); /* quadx_base_env.py:243 */
  orc_matrix_from_quat(qe, R);
  for (int i = 0; i < L->n_targets_left; ++i) {
    double d[3] = {L->targets[i][0] - L->p[0], L->targets[i][1] - L->p[1], L->targets[i][2] - L->p[2]};
    matTvec(R, d, deltas[i]); /* row-vector @ R */
    deltas[i][3] = 0.0;
    if (P->use_yaw_targets) {
      double e = L->yaw_targets[i] - L->rpy[2];
      if (e > PI) e -= 2.0 * PI;   /* rollover yaw, :147-149 */
      if (e < -PI) e += 2.0 * PI;
      deltas[i][3] = e;
    }
  }
  if (P->use_yaw_targets && L->n_targets_left > 0) L->yaw_error_scalar = fabs(deltas[0][3]);
}
int orc_obs_dim(const orc_params* P) {
  int att = (P->angle_repr ? 13 : 12) + 4 + (P->vehicle == ORC_QUADX ? 4 : 6);
  if (P->task == ORC_TASK_MA_HOVER) return att + 3;
  return att + (P->task == ORC_TASK_WAYPOINTS ? (P->use_yaw_targets ? 4 : 3) * P->num_targets : 0);
}
/* quadx_hover_env.py:85-115 ; quadx_waypoints_env.py:125-175 ; flatten_waypoint_env.py:42-62.
 * compute_state(): builds the observation (cached in the lane, as the reference caches
 * self.state) and, for waypoint envs, updates old/new distance (waypoint_handler.py:141-142). */
static void env_compute_state(const orc_params* P, orc_lane* L) {
  double* obs = L->obs;
  int k = 0;
  for (int i = 0; i < 3; ++i) obs[k++] = L->w_b[i];
  if (P->angle_repr) {
    double qe[4];
    orc_quat_from_euler(L->rpy, qe);
    for (int i = 0; i < 4; ++i) obs[k++] = qe[i];
  } else {
    for (int i = 0; i < 3; ++i) obs[k++] = L->rpy[i];
  }
  for (int i = 0; i < 3; ++i) obs[k++] = L->v_b[i];
  for (int i = 0; i < 3; ++i) obs[k++] = L->p[i];
  if (P->task == ORC_TASK_MA_HOVER) { /* ma_quadx_hover_env.py:141-166: aux, past action, start_pos */
    for (int i = 0; i < 4; ++i) obs[k++] = L->throttle[i];
    for (int i = 0; i < 4; ++i) obs[k++] = L->past_action[i];
    for (int i = 0; i < 3; ++i) obs[k++] = P->start_pos[i];
    return;
  }
  for (int i = 0; i < 4; ++i) obs[k++] = L->action[i];
  if (P->vehicle == ORC_QUADX) {
    for (int i = 0; i < 4; ++i) obs[k++] = L->throttle[i];
  } else { /* fixedwing.py:289-291 */
    for (int i = 0; i < 5; ++i) obs[k++] = L->actuation[i];
    obs[k++] = L->throttle[0];
  }
  if (P->task == ORC_TASK_WAYPOINTS) {
    double deltas[ORC_MAX_TARGETS][4];
    compute_deltas(P, L, deltas);
    const int wcols = P->use_yaw_targets ? 4 : 3;
    for (int i = 0; i < P->num_targets; ++i)
      for (int c = 0; c < wcols; ++c) obs[k++] = (i < L->n_targets_left) ? deltas[i][c] : 0.0;
    L->old_dist = L->new_dist;
    L->new_dist = sqrt(dot3(deltas[0], deltas[0]));
  }
}
void orc_env_obs(const orc_params* P, const orc_lane* L, double* obs) {
  memcpy(obs, L->obs, sizeof(double) * (size_t)orc_obs_dim(P));
}
/* quadx_base_env.py:251-267 + hover :117-138 + waypoints :177-204 + fixedwing_waypoints :169-190 */
static void env_term_trunc_reward(const orc_params* P, orc_lane* L) {
  if (L->step_count > P->max_steps) L->truncated = 1;
  if (P->task == ORC_TASK_MA_HOVER) { /* ma_quadx_hover_env.py:168-205: additive penalties */
    if (L->contact_step) { L->reward -= 100.0; L->info_collision = 1; L->terminated = 1; }
    if (sqrt(dot3(L->p, L->p)) > P->dome) { L->reward -= 100.0; L->info_oob = 1; L->terminated = 1; }
    if (!P->sparse_reward) {
      double d[3] = {L->p[0] - P->start_pos[0], L->p[1] - P->start_pos[1], L->p[2] - P->start_pos[2]};
      double linear_distance = sqrt(dot3(d, d));
      double angular_distance = sqrt(L->rpy[0] * L->rpy[0] + L->rpy[1] * L->rpy[1]);
      L->reward -= linear_distance + angular_distance * 0.1;
      L->reward += 1.0;
    }
    return;
  }
  if (L->contact_step) { L->reward = -100.0; L->info_collision = 1; L->terminated = 1; }
  if (sqrt(dot3(L->p, L->p)) > P->dome) { L->reward = -100.0; L->info_oob = 1; L->terminated = 1; }
  if (P->task == ORC_TASK_HOVER) {
    if (!P->sparse_reward) {
      double d[3] = {L->p[0] - 0.0, L->p[1] - 0.0, L->p[2] - 1.0};
      double linear_distance = sqrt(dot3(d, d));
      double yaw_rate = fabs(L->w_b[2]);
      L->reward -= 0.01 * (yaw_rate * yaw_rate);
      double angular_distance = sqrt(L->rpy[0] * L->rpy[0] + L->rpy[1] * L->rpy[1]);
      L->reward -= linear_distance + angular_distance;
      L->reward += 1.0;
    }
  } else if (P->task == ORC_TASK_WAYPOINTS) {
    if (!P->sparse_reward) {
      double progress = (isinf(L->old_dist + L->new_dist)) ? 0.0 : L->old_dist - L->new_dist;
      double pr = 3.0 * progress;
      L->reward += pr > 0.0 ? pr : 0.0;
      L->reward += P->wp_dist_reward / L->new_dist;
      if (P->wp_yaw_penalty != 0.0) {
        double yaw_rate = fabs(L->w_b[2]);
        L->reward -= P->wp_yaw_penalty * (yaw_rate * yaw_rate);
      }
    }
    if (L->new_dist < P->goal_reach_distance &&
        (!P->use_yaw_targets || L->yaw_error_scalar < P->goal_reach_angle)) { /* target_reached, waypoint_handler.py:167-179 */
      L->reward = 100.0;
      for (int i = 1; i < L->n_targets_left; ++i) {
        memcpy(L->targets[i - 1], L->targets[i], sizeof(L->targets[0]));
        L->yaw_targets[i - 1] = L->yaw_targets[i];
      }
      L->n_targets_left -= 1;
      int all = (L->n_targets_left == 0);
      if (all) L->truncated = 1;
      L->info_complete = all;
      L->num_targets_reached = P->num_targets - L->n_targets_left;
    }
  }
}

/* quadx_base_env.py:149-212 (+ quadx_waypoints_env.py:120-123) */
/* What an env.reset() draws -- the settle phase's motor noise (stream 1) and the waypoints (stream 2) -- is keyed, for the QuadX
 * Hover / Waypoints tasks, by the event counter AT THE PREVIOUS RESET of the lane (0 before the first), not by the current one
 * (round 5; the fixedwing and PettingZoo tasks keep the current counter). "The counter at the previous reset" is the value that reset
 * LEFT BEHIND (its counter + 1, round 6): strictly increasing from reset to reset. The reference's own generator is a sequential PCG64
 * stream that no device can follow, so the keying is this restatement's to choose; this choice makes the next episode's initial
 * state a function of something known an episode ahead, which lets the device compute it for many lanes at once while the
 * episode runs instead of for the two or three lanes of a wavefront that restart in a given step (quadx_fast.hpp: spares). */
static int orc_reset_rekeyed(const orc_params* P) {
  return P->vehicle == ORC_QUADX && (P->task == ORC_TASK_HOVER || P->task == ORC_TASK_WAYPOINTS);
}
void orc_env_reset(const orc_params* P, orc_lane* L, uint64_t lane_id, const double* xi_reset, const double* u_targets) {
  orc_aviary_reset(P, L, lane_id);
  const uint32_t ctr_now = L->rng_ctr;
  const int rekey = orc_reset_rekeyed(P);
  if (rekey) L->rng_ctr = L->reset_key; /* (lane_normal / lane_uniform read the counter from the lane) */
  if (P->task == ORC_TASK_WAYPOINTS) sample_targets(P, L, u_targets);
  orc_set_mode(P, L, P->flight_mode);
  const int tpc = P->world.ticks_per_control;
  for (int s = 0; s < P->settle_steps; ++s)
    orc_aviary_step(P, L, xi_reset ? xi_reset + s * tpc : 0, (uint32_t)(s * tpc), 1);
  /* the NEXT reset's key: the counter as this reset leaves it (ctr_now + 1). Keys are strictly increasing -- a fresh lane's first reset
   * draws with key 0 and leaves key 1; round 5 stored ctr_now, which is 0 again at the first reset, so that every lane's first and
   * second episodes drew the same settle noise and the same waypoints. Streams 1 / 2 never meet the step stream 0. */
  if (rekey) { L->rng_ctr = ctr_now; L->reset_key = ctr_now + 1u; }
  env_compute_state(P, L);
  L->rng_ctr += 1;
}
/* quadx_base_env.py:269-301 ; fixedwing_base_env.py:244-278 */
void orc_env_step(const orc_params* P, orc_lane* L, const double action[4], const double* xi) {
  if (P->task == ORC_TASK_MA_HOVER) { /* ma_quadx_base_env.py:309-371, one agent */
    for (int i = 0; i < 4; ++i) { L->past_action[i] = L->action[i]; L->action[i] = action[i]; L->setpoint[i] = action[i]; }
    L->reward = 0.0; L->terminated = 0; L->truncated = 0;
    const int tpc2 = P->world.ticks_per_control;
    for (int s = 0; s < P->env_step_ratio; ++s) { /* no early exit */
      orc_aviary_step(P, L, xi ? xi + s * tpc2 : 0, (uint32_t)(s * tpc2), 0);
      env_term_trunc_reward(P, L);
      env_compute_state(P, L);
    }
    L->step_count += 1;
    L->rng_ctr += 1;
    return;
  }
  for (int i = 0; i < 4; ++i) { L->action[i] = action[i]; L->setpoint[i] = action[i]; }
  if (P->throttle_remap) L->setpoint[3] = (action[3] / 2.0) + 0.5;
  L->reward = -0.1;
  const int tpc = P->world.ticks_per_control;
  for (int s = 0; s < P->env_step_ratio; ++s) {
    if (L->terminated || L->truncated) break;
    orc_aviary_step(P, L, xi ? xi + s * tpc : 0, (uint32_t)(s * tpc), 0);
    env_compute_state(P, L);
    env_term_trunc_reward(P, L);
  }
  L->step_count += 1;
  L->rng_ctr += 1;
}

/* ---- PettingZoo env on a shared world (ma_quadx_base_env.py:183-371, ma_quadx_hover_env.py:99-205) ---- */
void orc_world_env_reset(const orc_params* const* Pl, orc_lane* const* Ll, int A, uint64_t lane_id0, const double* const* xi_reset) {
  for (int i = 0; i < A; ++i) {
    orc_aviary_reset(Pl[i], Ll[i], lane_id0 + (uint64_t)i);
    orc_set_mode(Pl[i], Ll[i], Pl[i]->flight_mode);
    Ll[i]->world_contact = 0; Ll[i]->peer_contact = 0;
  }
  const int tpc = Pl[0]->world.ticks_per_control;
  const double* xs[64];
  for (int s = 0; s < Pl[0]->settle_steps; ++s) {
    for (int i = 0; i < A; ++i) xs[i] = (xi_reset && xi_reset[i]) ? xi_reset[i] + s * tpc : 0;
    orc_world_aviary_step(Pl, Ll, A, xi_reset ? xs : 0, (uint32_t)(s * tpc), 1);
  }
  for (int i = 0; i < A; ++i) { env_compute_state(Pl[i], Ll[i]); Ll[i]->rng_ctr += 1; }
}
void orc_world_env_step(const orc_params* const* Pl, orc_lane* const* Ll, int A, const double* actions, const double* const* xi) {
  for (int i = 0; i < A; ++i) {
    orc_lane* L = Ll[i];
    for (int k = 0; k < 4; ++k) { L->past_action[k] = L->action[k]; L->action[k] = actions[4 * i + k]; L->setpoint[k] = actions[4 * i + k]; }
    L->reward = 0.0; L->terminated = 0; L->truncated = 0;
  }
  const int tpc = Pl[0]->world.ticks_per_control;
  const double* xs[64];
  for (int s = 0; s < Pl[0]->env_step_ratio; ++s) { /* no early exit (:342-361) */
    for (int i = 0; i < A; ++i) xs[i] = (xi && xi[i]) ? xi[i] + s * tpc : 0;
    orc_world_aviary_step(Pl, Ll, A, xi ? xs : 0, (uint32_t)(s * tpc), 0);
    for (int i = 0; i < A; ++i) { env_term_trunc_reward(Pl[i], Ll[i]); env_compute_state(Pl[i], Ll[i]); }
  }
  for (int i = 0; i < A; ++i) { Ll[i]->step_count += 1; Ll[i]->rng_ctr += 1; }
}

/* ---- MAFixedwingDogfightEnv (pz_envs/fixedwing_envs/ma_fixedwing_dogfight_env.py, ma_fixedwing_base_env.py) ----
 * A = 2 * team_size Acrowing aircraft in one world; agents [0, team_size) are one team, the rest the other
 * (:108-126 team_flag, friendly_fire_mask). Per-agent spawn pose / velocity come in through each agent's orc_params
 * (start_pos, start_rpy, start_vel): orc_dogfight_spawn() restates _get_start_pos_orn (:176-213) and reset()'s 20 m/s
 * forward velocity (:216-222) from a vector of uniforms. */
void orc_dogfight_spawn(int team_size, double min_radius, double max_radius, const double* u /* [1 + 6 team_size] */,
                        double* pos /* [A][3] */, double* rpy /* [A][3] */, double* vel /* [A][3] */) {
  const int A = 2 * team_size;
  const double phase = u[0] * 2.0 * PI; /* np_random.uniform(0, 2 pi) */
  for (int i = 0; i < A; ++i) {
    const double rad = PI / team_size * i + phase;                              /* :189-191 */
    const double radius = min_radius + (max_radius - min_radius) * u[1 + i];       /* :192-196 */
    const double height = min_radius + (max_radius - min_radius) * u[1 + A + i];   /* :197-201: spawn_min/max_RADIUS, sic */
    pos[3 * i + 0] = radius * cos(rad); pos[3 * i + 1] = radius * sin(rad); pos[3 * i + 2] = height;
    rpy[3 * i + 0] = 0.0; rpy[3 * i + 1] = 0.0;
    rpy[3 * i + 2] = rad + u[1 + 2 * A + i] * PI / 8.0;                          /* :210-212 */
    /* compute_rotation_forward(start_orn)[1] * 20 (ma_fixedwing_base_env.py:403-405 with roll = pitch = 0) */
    vel[3 * i + 0] = 20.0 * cos(rpy[3 * i + 2]); vel[3 * i + 1] = 20.0 * sin(rpy[3 * i + 2]); vel[3 * i + 2] = -0.0 * 20.0;
  }
}

static int df_team(const orc_dogfight* D, int i) { return i >= D->team_size; }

/* update_states() = _compute_observation() (:466-549) + _compute_term_trunc_rew_info() (:651-722) */
static void dogfight_update_states(const orc_params* const* Pl, orc_lane* const* Ll, orc_dogfight* D) {
  const int A = D->A;
  double att[ORC_DF_MAX][12], R[ORC_DF_MAX][9], fwd[ORC_DF_MAX][3], gv[ORC_DF_MAX][3];
  (void)Pl;
  memcpy(D->prev_dist, D->cur_dist, sizeof(D->cur_dist));
  memcpy(D->prev_ang, D->cur_ang, sizeof(D->cur_ang));
  for (int i = 0; i < A; ++i) {
    const orc_lane* L = Ll[i];
    for (int k = 0; k < 3; ++k) { att[i][k] = L->w_b[k]; att[i][3 + k] = L->rpy[k]; att[i][6 + k] = L->v_b[k]; att[i][9 + k] = L->p[k]; }
    /* compute_rotation_forward (ma_fixedwing_base_env.py:336-405): rz @ ry @ rx and the nose direction */
    const double cr = cos(L->rpy[0]), sr = sin(L->rpy[0]), cp = cos(L->rpy[1]), sp = sin(L->rpy[1]), cy = cos(L->rpy[2]), sy = sin(L->rpy[2]);
    double* r = R[i];
    r[0] = cy * cp; r[1] = cy * sp * sr - sy * cr; r[2] = cy * sp * cr + sy * sr;
    r[3] = sy * cp; r[4] = sy * sp * sr + cy * cr; r[5] = sy * sp * cr - cy * sr;
    r[6] = -sp;     r[7] = cp * sr;                r[8] = cp * cr;
    fwd[i][0] = cy * cp; fwd[i][1] = sy * cp; fwd[i][2] = -sp;
    for (int k = 0; k < 3; ++k) att[i][9 + k] -= fwd[i][k] * 0.35; /* :318: the state is the nose's, shift to the body centre */
    for (int k = 0; k < 3; ++k) gv[i][k] = r[3 * k] * att[i][6] + r[3 * k + 1] * att[i][7] + r[3 * k + 2] * att[i][8]; /* :365 */
  }
  int hits[ORC_DF_MAX][ORC_DF_MAX];
  for (int i = 0; i < A; ++i) {
    for (int j = 0; j < A; ++j) {
      double sep[3], dist = 0.0, dotf = 0.0;
      for (int k = 0; k < 3; ++k) { sep[k] = att[j][9 + k] - att[i][9 + k]; dist += sep[k] * sep[k]; dotf += sep[k] * fwd[i][k]; }
      dist = sqrt(dist);
      const double ang = acos(dotf / dist); /* NaN on the diagonal, as in the reference (:327-333) */
      D->cur_dist[i][j] = dist; D->cur_ang[i][j] = ang;
      D->in_range[i][j] = dist < D->lethal_distance;
      D->chasing[i][j] = fabs(ang) < PI / 2.0;
      const int ff = df_team(D, i) != df_team(D, j);
      hits[i][j] = (ang < D->lethal_angle) && D->in_range[i][j] && D->chasing[i][j] && ff; /* :339-344, :496 */
      D->cur_hit[i][j] = hits[i][j];
      /* the other aircraft as seen from this one (:350-378) */
      double* o = D->other_att[i][j];
      for (int k = 0; k < 3; ++k) { o[k] = att[j][k]; o[3 + k] = att[j][3 + k] - att[i][3 + k]; }
      for (int k = 0; k < 3; ++k) { /* gv[j] @ R[i] (row vector times matrix) minus the own body velocity; sep @ R[i] */
        o[6 + k] = gv[j][0] * R[i][k] + gv[j][1] * R[i][3 + k] + gv[j][2] * R[i][6 + k] - att[i][6 + k];
        o[9 + k] = sep[0] * R[i][k] + sep[1] * R[i][3 + k] + sep[2] * R[i][6 + k];
      }
    }
  }
  for (int j = 0; j < A; ++j) { /* :499-503; healths is a float32 array */
    int rec = 0;
    for (int i = 0; i < A; ++i) rec += hits[i][j];
    D->received_hits[j] += rec;
    D->health[j] = (double)(float)(D->health[j] - D->damage_per_hit * rec);
    if (D->health[j] < 0.0) D->health[j] = 0.0;
  }
  double dist_origin[ORC_DF_MAX];
  for (int i = 0; i < A; ++i) {
    const double sp2 = att[i][6] * att[i][6] + att[i][7] * att[i][7] + att[i][8] * att[i][8];
    D->inactive[i] = (D->health[i] <= 0.0) && (att[i][11] < 2.0) && (sqrt(sp2) < 0.1); /* :505-510 */
    dist_origin[i] = sqrt(att[i][9] * att[i][9] + att[i][10] * att[i][10] + att[i][11] * att[i][11]);
  }
  const int AD = D->action_dim == 6 ? 6 : 4;
  const int Dobs = 19 + AD + (A - 1) * 14;
  for (int i = 0; i < A; ++i) { /* :519-549, pop_obs_by_id :724-752 (flattened, zero padded) */
    double* o = D->obs[i];
    int k = 0;
    memset(o, 0, sizeof(double) * (size_t)Dobs);
    for (int c = 0; c < 12; ++c) o[k++] = att[i][c];
    for (int c = 0; c < 5; ++c) o[k++] = Ll[i]->actuation[c]; /* aviary.aux_state(i): fixedwing.py:289-291 */
    o[k++] = Ll[i]->throttle[0];
    o[k++] = D->health[i];
    for (int c = 0; c < AD; ++c) o[k++] = D->past_action[i][c];
    for (int j = 0; j < A; ++j) {
      if (j == i || D->inactive[j]) continue;
      for (int c = 0; c < 12; ++c) o[k++] = D->other_att[i][j][c];
      o[k++] = D->health[j];
      o[k++] = df_team(D, j) == df_team(D, i) ? 1.0 : 0.0;
    }
  }
  /* ---- _compute_engagement_rewards (:551-620), _compute_boundary_rewards (:622-649) */
  int team_hits[2] = {0, 0};
  for (int i = 0; i < A; ++i)
    for (int j = 0; j < A; ++j) team_hits[df_team(D, i)] += hits[i][j];
  for (int i = 0; i < A; ++i) {
    double e = 0.0;
    for (int j = 0; j < A; ++j) {
      if (j == i) continue; /* fill_diagonal(0) */
      const int ff = df_team(D, i) != df_team(D, j);
      if (!D->sparse_reward) {
        double dd = D->prev_dist[i][j] - D->cur_dist[i][j];
        if (dd < 0.0) dd = 0.0;
        e += 4.0 * dd * ((!D->in_range[i][j]) && D->chasing[i][j] && ff);
        double da = (D->prev_ang[i][j] - D->cur_ang[i][j]) * (D->in_range[i][j] && ff);
        if (da < 0.0) da *= D->aggressiveness;
        e += 30.0 * da;
        const double iaa_ij = (1.0 / (D->cur_ang[i][j] + 0.1)) * (ff && D->in_range[i][j] && D->chasing[i][j]);
        const double iaa_ji = (1.0 / (D->cur_ang[j][i] + 0.1)) * (ff && D->in_range[j][i] && D->chasing[j][i]);
        e += 3.0 * (iaa_ij - (1.0 - D->aggressiveness) * iaa_ji);
      }
      e += 20.0 * (hits[i][j] - (1.0 - D->aggressiveness) * hits[j][i]);
    }
    e += D->cooperativeness * team_hits[df_team(D, i)]; /* :609-617 */
    double b = 0.0;
    if (!D->sparse_reward) {
      b += tanh(0.1 * att[i][11] - 1.0);
      b -= tanh(0.0025 * dist_origin[i] - 1.0);
      for (int j = 0; j < A; ++j)
        if (j != i && D->cur_dist[i][j] < 5.0) b -= 10.0 * (5.0 - D->cur_dist[i][j]);
    }
    D->acc_reward[i] += e + b;
    if (D->step_count > D->max_steps) D->acc_trunc[i] = 1;
  }
  for (int i = 0; i < A; ++i) { /* :665-680 */
    if (D->health[i] <= 1e-3) { D->acc_term[i] = 1; D->info_bits[i] |= 1; }
    if (Ll[i]->contact_step) { D->acc_term[i] = 1; D->acc_reward[i] = -1000.0; D->health[i] = 0.0; D->info_bits[i] |= 2; }
    if (dist_origin[i] > D->dome) { D->acc_term[i] = 1; D->acc_reward[i] = -1000.0; D->health[i] = 0.0; D->info_bits[i] |= 4; }
  }
  /* :682-690 team_wins[team] = (healths[other team] <= 0) & any(healths[team] > 0): an ELEMENT-WISE assignment -- member k of
   * a team "wins" when member k of the other team is out and someone of its own team is still up */
  int any_up[2] = {0, 0};
  for (int i = 0; i < A; ++i) any_up[df_team(D, i)] |= D->health[i] > 0.0;
  for (int i = 0; i < A; ++i) {
    const int t = df_team(D, i), k = i - t * D->team_size, opp = (1 - t) * D->team_size + k;
    if (D->health[opp] <= 0.0 && any_up[t]) { D->acc_term[i] = 1; D->acc_reward[i] = 300.0; D->info_bits[i] |= 8; }
  }
}

int orc_sizeof_dogfight(void) { return (int)sizeof(orc_dogfight); }

void orc_dogfight_reset(const orc_params* const* Pl, orc_lane* const* Ll, orc_dogfight* D, uint64_t lane_id0, const double* const* xi_reset) {
  const int A = D->A;
  for (int i = 0; i < A; ++i) {
    const uint32_t ctr = Ll[i]->rng_ctr;
    orc_aviary_reset(Pl[i], Ll[i], lane_id0 + (uint64_t)i);
    Ll[i]->rng_ctr = ctr;
    orc_set_mode(Pl[i], Ll[i], Pl[i]->flight_mode); /* end_reset: set_mode(0) (ma_fixedwing_base_env.py:229) */
    Ll[i]->world_contact = 0; Ll[i]->peer_contact = 0;
  }
  D->step_count = 0;
  memset(D->cur_dist, 0, sizeof(D->cur_dist)); memset(D->cur_ang, 0, sizeof(D->cur_ang));
  memset(D->prev_dist, 0, sizeof(D->prev_dist)); memset(D->prev_ang, 0, sizeof(D->prev_ang));
  memset(D->cur_hit, 0, sizeof(D->cur_hit)); memset(D->in_range, 0, sizeof(D->in_range)); memset(D->chasing, 0, sizeof(D->chasing));
  for (int i = 0; i < A; ++i) {
    D->alive[i] = 1; D->health[i] = 1.0; D->received_hits[i] = 0; D->inactive[i] = 0;
    D->acc_reward[i] = 0.0; D->acc_term[i] = 0; D->acc_trunc[i] = 0; D->info_bits[i] = 0;
    D->reward[i] = 0.0; D->terminated[i] = 0; D->truncated[i] = 0;
    /* current_actions / past_actions are created in __init__ and survive resets (ma_fixedwing_base_env.py:131-146) */
  }
  const int tpc = Pl[0]->world.ticks_per_control;
  const double* xs[ORC_DF_MAX];
  for (int s = 0; s < Pl[0]->settle_steps; ++s) { /* :232-233 */
    for (int i = 0; i < A; ++i) xs[i] = (xi_reset && xi_reset[i]) ? xi_reset[i] + s * tpc : 0;
    orc_world_aviary_step(Pl, Ll, A, xi_reset ? xs : 0, (uint32_t)(s * tpc), 1);
  }
  dogfight_update_states(Pl, Ll, D); /* :234; whatever it accumulates is popped by the first step */
  for (int i = 0; i < A; ++i) Ll[i]->rng_ctr += 1;
}

void orc_dogfight_step(const orc_params* const* Pl, orc_lane* const* Ll, orc_dogfight* D, const double* actions, const double* const* xi) {
  const int A = D->A;
  const int AD = D->action_dim == 6 ? 6 : 4;
  for (int i = 0; i < A; ++i) { /* ma_fixedwing_base_env.py:289-302 */
    for (int k = 0; k < AD; ++k) {
      D->past_action[i][k] = D->action[i][k];
      D->action[i][k] = D->alive[i] ? actions[AD * i + k] : 0.0;
    }
    for (int k = 0; k < 4; ++k) Ll[i]->setpoint[k] = D->action[i][k]; /* mode 0 reads setpoint[0:4] */
    if (AD == 4) Ll[i]->setpoint[3] = D->action[i][3] / 2.0 + 0.5;      /* aviary_action[..., -1] = a / 2 + 0.5: entry 5 of a 6-wide action */
  }
  const int tpc = Pl[0]->world.ticks_per_control;
  const double* xs[ORC_DF_MAX];
  for (int s = 0; s < D->env_step_ratio; ++s) { /* :305-307 */
    for (int i = 0; i < A; ++i) xs[i] = (xi && xi[i]) ? xi[i] + s * tpc : 0;
    orc_world_aviary_step(Pl, Ll, A, xi ? xs : 0, (uint32_t)(s * tpc), 0);
    dogfight_update_states(Pl, Ll, D);
  }
  for (int i = 0; i < A; ++i) { /* :316-330, pop_term_trunc_rew_info_by_id (dogfight :754-771) */
    D->reward[i] = 0.0; D->terminated[i] = 0; D->truncated[i] = 0;
    if (D->alive[i]) {
      D->reward[i] = D->acc_reward[i]; D->acc_reward[i] = 0.0;
      D->terminated[i] = D->acc_term[i]; D->truncated[i] = D->acc_trunc[i];
      if (D->terminated[i] || D->truncated[i]) D->alive[i] = 0;
    }
    Ll[i]->rng_ctr += 1;
  }
  D->step_count += 1;
}

/* ------------------------------------------------------------------ batch level */
void orc_env_reset_batch(const orc_params* P, orc_lane* L, int n, uint64_t lane0, const uint8_t* mask,
                         const double* xi_reset, const double* u_targets) {
  const int nr = P->settle_steps * P->world.ticks_per_control, nu = (P->use_yaw_targets ? 4 : 3) * P->num_targets;
#pragma omp parallel for schedule(static)
  for (int i = 0; i < n; ++i) {
    if (mask && !mask[i]) continue;
    orc_env_reset(P, &L[i], lane0 + (uint64_t)i, xi_reset ? xi_reset + (size_t)i * nr : 0,
                  u_targets ? u_targets + (size_t)i * nu : 0);
  }
}
void orc_env_step_batch(const orc_params* P, orc_lane* L, int n, const float* actions, const double* xi,
                        const double* xi_reset, const double* u_targets, int autoreset, double* obs,
                        double* reward, uint8_t* term, uint8_t* trunc, double* final_obs) {
  const int D = orc_obs_dim(P);
  const int ns = P->env_step_ratio * P->world.ticks_per_control;
  const int nr = P->settle_steps * P->world.ticks_per_control, nu = (P->use_yaw_targets ? 4 : 3) * P->num_targets;
#pragma omp parallel for schedule(static)
  for (int i = 0; i < n; ++i) {
    orc_lane* l = &L[i];
    const double* xr = xi_reset ? xi_reset + (size_t)i * nr : 0;
    const double* ut = u_targets ? u_targets + (size_t)i * nu : 0;
    if (autoreset == 1 && (l->terminated || l->truncated)) { /* gymnasium NEXT_STEP */
      orc_env_reset(P, l, l->lane_id, xr, ut);
      orc_env_obs(P, l, obs + (size_t)i * D);
      reward[i] = 0.0; term[i] = 0; trunc[i] = 0;
      continue;
    }
    double a[4] = {actions[4 * i + 0], actions[4 * i + 1], actions[4 * i + 2], actions[4 * i + 3]};
    orc_env_step(P, l, a, xi ? xi + (size_t)i * ns : 0);
    reward[i] = l->reward; term[i] = (uint8_t)l->terminated; trunc[i] = (uint8_t)l->truncated;
    if (autoreset == 2 && (l->terminated || l->truncated)) { /* SAME_STEP */
      if (final_obs) orc_env_obs(P, l, final_obs + (size_t)i * D);
      orc_env_reset(P, l, l->lane_id, xr, ut);
    }
    orc_env_obs(P, l, obs + (size_t)i * D);
  }
}
int orc_sizeof_lane(void) { return (int)sizeof(orc_lane); }
int orc_sizeof_params(void) { return (int)sizeof(orc_params); }
int orc_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
