// shared_world.hpp -- what couples the drones of ONE Bullet world (the PettingZoo envs: pz_envs/quadx_envs/ma_quadx_base_env.py:206-241,
// pz_envs/fixedwing_envs/ma_fixedwing_base_env.py): the A lanes of a world are adjacent lanes of one wavefront and exchange
// pose and contact bit through LDS before every physics tick.
#pragma once
#include "../../include/pyflyt_amd.h"
#include "uav_device.hpp"
#include "uav_vehicles.hpp"

namespace pf {

// the exchanged pose slot's eighth word: contact bit + 2 x "a wreck at rest" (world_exchange)
PF_DEV bool slot_contact(const float x) { return (((int)x) & 1) != 0; }
PF_DEV bool slot_at_rest(const float x) { return ((int)x) >= 2; }


PF_DEV void lds_sync_wave() {  // one wave per workgroup: LDS traffic ordered, no s_barrier needed
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_s_waitcnt(0xc07f);
  __builtin_amdgcn_wave_barrier();
}

// A reset mask that selects some agents of a shared world selects the world: the agents of a world are reset together (their
// lanes exchange data inside the kernels, and the reference rebuilds the whole world: ma_quadx_base_env.py:206-241). Every lane
// of the wave must call this (it is a ballot).
PF_DEV bool widen_to_world(const bool selected, const int tid, const int A) {
  const unsigned long long m = __ballot(selected);
  const int wbase = (tid / A) * A;
  const unsigned long long wm = (A >= 64 ? ~0ull : ((1ull << A) - 1ull)) << wbase;
  return (m & wm) != 0ull;
}

// The drone-drone box tests of one body against its touching peers, out of line: they run only when bounding spheres touch, and
// inlined they put the tick loops of their callers over the register budget (the shared-world instantiation of the QuadX
// kernel spilled to scratch memory). This drone's boxes in the peer's box frames, 15 axes each (btBoxBoxDetector's verdict).
__device__ __noinline__ bool peers_overlap_dev(const pf_params* __restrict__ Pd, const float* wpose, const int wbase, const int wlocal, const int A,
                                               const float px, const float py, const float pz, const quat q, const float rr2) {
  const m3 Ra = rot_from_quat(q);
  bool peer = false;
  const bool mine = slot_contact(wpose[(wbase + wlocal) * 8 + 7]);  // this body held contact points after the previous tick
  const float rd_fresh = Pd->contact_report_distance, rd_kept = Pd->contact_break_distance;
  for (int j = 1; j < A; ++j) {
    int jj = wlocal + j;
    jj = jj >= A ? jj - A : jj;
    const float* o = wpose + (wbase + jj) * 8;
    const v3 d{px - o[0], py - o[1], pz - o[2]};
    if (dot(d, d) <= rr2) {
      // ONE test per pair, whichever of its two lanes evaluates it: the box of the drone with the LOWER index (a) in the frame of the
      // other's (b), b's box enlarged by the report distance (oracle: drones_overlap(i, j), i < j). With an enlargement the verdict
      // is not symmetric -- a against b + rd is another shape than b against a + rd -- and the two lanes of a pair that separates
      // through the breaking distance disagreed about the tick in which the report ends (tests/test_gpu_onestep.py).
      const bool me_first = wlocal < jj;
      const m3 Rp = rot_from_quat(quat{o[3], o[4], o[5], o[6]});
      m3 RA, RB;
      v3 dab;
      if (me_first) { RA = Ra; RB = Rp; dab = d; }
      else { RA = Rp; RB = Ra; dab = v3{-d.x, -d.y, -d.z}; }
      const m3 Rrel{RB.m00 * RA.m00 + RB.m10 * RA.m10 + RB.m20 * RA.m20, RB.m00 * RA.m01 + RB.m10 * RA.m11 + RB.m20 * RA.m21, RB.m00 * RA.m02 + RB.m10 * RA.m12 + RB.m20 * RA.m22,
                    RB.m01 * RA.m00 + RB.m11 * RA.m10 + RB.m21 * RA.m20, RB.m01 * RA.m01 + RB.m11 * RA.m11 + RB.m21 * RA.m21, RB.m01 * RA.m02 + RB.m11 * RA.m12 + RB.m21 * RA.m22,
                    RB.m02 * RA.m00 + RB.m12 * RA.m10 + RB.m22 * RA.m20, RB.m02 * RA.m01 + RB.m12 * RA.m11 + RB.m22 * RA.m21, RB.m02 * RA.m02 + RB.m12 * RA.m12 + RB.m22 * RA.m22};
      const int nb = Pd->n_boxes;
      for (int k = 0; k < nb; ++k) {
        for (int l = 0; l < nb; ++l) {
          const pf_box bk = Pd->boxes[k], bl = Pd->boxes[l];
          const v3 ca = dab + mul(RA, v3{bk.c[0], bk.c[1], bk.c[2]}) - mul(RB, v3{bl.c[0], bl.c[1], bl.c[2]});
          // (reported from the gap rd on -- up to the breaking distance when either drone holds contact points: b's box enlarged)
          const float rd = (mine || slot_contact(o[7])) ? rd_kept : rd_fresh;
          const float hb[3] = {bl.h[0] + rd, bl.h[1] + rd, bl.h[2] + rd};
          peer |= box_overlaps_aabb(mulT(RB, ca), Rrel, bk.h, v3{0.f, 0.f, 0.f}, hb);
        }
      }
    }
  }
  return peer;
}

// ---------------------------------------------------------------- contact response between the drones of a world
// The pair stage of stepSimulation's contact response (the model: include/pyflyt_amd.h at pf_params.contact_response; two fp64
// restatements of it exist on the test side and agree to 1e-8): impulses between the bodies of a world on their post-force
// velocities, before each body's ground solve.
//   wpose: 8 floats per lane (pose, published by world_exchange before this tick); wvel: kPairVelStride floats per lane -- in: the
//   body's new world-frame velocity v, w; out: v, w after the impulses and the position-level shift to add after the position
//   update. rec: LDS for the contact records (kPairRecFloats per touching world; as many worlds per round as fit).
// Drones touch rarely (a hit ends both episodes of the PettingZoo task) and then for a few ticks: ONE lane per touching world
// -- its first -- walks that world's contacts serially, every body's twist staying in LDS where both partners of a contact find
// it; the other lanes wait. Airframes: plain boxes (checked at context creation); a centre of mass off the base origin (the
// aeroplanes: pf_params.com) is handled -- arms from the centre of mass, the exchanged base-origin velocities converted to
// centre-of-mass velocities for the sweeps and back. Every lane of the wave that is inside the caller's tick must call this
// together; the lane that works for a world is the first of ITS lanes inside the call (the dogfight leaves wrecks at rest out of
// the tick: their exchange entries hold zero velocity, world_exchange).
constexpr int kPairVelStride = 12;   // v (3), w (3), shift (3), pad
constexpr int kPairMaxContacts = 16; // = ORC_MAX_PAIR_CONTACTS
constexpr int kPairRec = 44;         // floats per contact record
constexpr int kPairRecFloats = kPairMaxContacts * kPairRec;
__device__ __noinline__ void pair_stage_dev(const pf_params* __restrict__ Pd, const float* wpose, float* wvel, float* rec_all, const int rec_floats, const int tid,
                                            const int A, const bool world_touch) {
  const int wbase = (tid / A) * A;
  const unsigned long long inside = __ballot(1);
  const unsigned long long wmask = (A >= 64 ? ~0ull : ((1ull << A) - 1ull)) << wbase;
  bool todo = world_touch && (tid == __ffsll((long long)(inside & wmask)) - 1);  // the world's first lane inside the call does the work
  const int slots = rec_floats / kPairRecFloats;
  unsigned long long m = __ballot(todo);
  while (m != 0ull) {
    const int rank = __popcll(m & ((1ull << (tid & 63)) - 1ull));
    if (todo && rank < slots) {
      todo = false;
      float* rec = rec_all + rank * kPairRecFloats;
      const float margin0 = Pd->contact_margin, brk = Pd->contact_break_distance, slop = Pd->contact_slop, inv_dt = 1.0f / Pd->dt, rest = Pd->contact_restitution;
      const float res_bound = __builtin_sqrtf(Pd->contact_residual_threshold);
      const float mu = Pd->contact_friction * Pd->contact_friction, erp = Pd->contact_erp, im = Pd->inv_mass, brad = Pd->bound_radius;
      const float Ii[6] = {Pd->I_inv[0], Pd->I_inv[1], Pd->I_inv[2], Pd->I_inv[3], Pd->I_inv[4], Pd->I_inv[5]};
      const int nb = Pd->n_boxes, iters = Pd->contact_iters;
      const bool offc = Pd->has_com_offset != 0;
      const v3 comb = offc ? v3{Pd->com[0], Pd->com[1], Pd->com[2]} : v3{0.0f, 0.0f, 0.0f};
      if (offc) {  // base-origin velocities -> centre-of-mass velocities: v_c = v + w x (R com)
        for (int i = 0; i < A; ++i) {
          const float* pi = wpose + (wbase + i) * 8;
          float* o = wvel + (wbase + i) * kPairVelStride;
          const v3 cw = mul(rot_from_quat(quat{pi[3], pi[4], pi[5], pi[6]}), comb);
          const v3 vc = v3{o[0], o[1], o[2]} + cross(v3{o[3], o[4], o[5]}, cw);
          o[0] = vc.x; o[1] = vc.y; o[2] = vc.z;
        }
      }
      int n = 0;
      // ---- contacts: every box vertex of a within the margin of being inside a box of b
      for (int a = 0; a < A; ++a) {
        const float* pa = wpose + (wbase + a) * 8;
        const m3 Ra = rot_from_quat(quat{pa[3], pa[4], pa[5], pa[6]});
        for (int b = 0; b < A; ++b) {
          if (b == a) continue;
          const float* pb = wpose + (wbase + b) * 8;
          const v3 d{pa[0] - pb[0], pa[1] - pb[1], pa[2] - pb[2]};
          // (a pair one of whose bodies held contact points after the previous tick keeps its points up to the breaking distance)
          const float margin = (slot_contact(pa[7]) || slot_contact(pb[7])) ? brk : margin0;
          const float rr = 2.0f * brad + 2.0f * margin;
          if (dot(d, d) > rr * rr) continue;
          const m3 Rb = rot_from_quat(quat{pb[3], pb[4], pb[5], pb[6]});
          const v3 cwa = mul(Ra, comb), cwb = mul(Rb, comb);  // the centres of mass from the base origins
          // world-frame inverse inertias R I^-1 R^T of both bodies (symmetric xx xy xz yy yz zz)
          float Iwa[6], Iwb[6];
          {
            const v3 r0{Ra.m00, Ra.m01, Ra.m02}, r1{Ra.m10, Ra.m11, Ra.m12}, r2{Ra.m20, Ra.m21, Ra.m22};
            const v3 c0 = symmul(Ii, r0), c1 = symmul(Ii, r1), c2 = symmul(Ii, r2);
            Iwa[0] = dot(r0, c0); Iwa[1] = dot(r0, c1); Iwa[2] = dot(r0, c2); Iwa[3] = dot(r1, c1); Iwa[4] = dot(r1, c2); Iwa[5] = dot(r2, c2);
          }
          {
            const v3 r0{Rb.m00, Rb.m01, Rb.m02}, r1{Rb.m10, Rb.m11, Rb.m12}, r2{Rb.m20, Rb.m21, Rb.m22};
            const v3 c0 = symmul(Ii, r0), c1 = symmul(Ii, r1), c2 = symmul(Ii, r2);
            Iwb[0] = dot(r0, c0); Iwb[1] = dot(r0, c1); Iwb[2] = dot(r0, c2); Iwb[3] = dot(r1, c1); Iwb[4] = dot(r1, c2); Iwb[5] = dot(r2, c2);
          }
          for (int ka = 0; ka < nb; ++ka) {
            for (int kb = 0; kb < nb; ++kb) {
              const pf_box ba = Pd->boxes[ka], bb = Pd->boxes[kb];
              const v3 cb = v3{pb[0], pb[1], pb[2]} + mul(Rb, v3{bb.c[0], bb.c[1], bb.c[2]});
              for (int vi = 0; vi < 8; ++vi) {
                const v3 l{ba.c[0] + ((vi & 1) ? ba.h[0] : -ba.h[0]), ba.c[1] + ((vi & 2) ? ba.h[1] : -ba.h[1]), ba.c[2] + ((vi & 4) ? ba.h[2] : -ba.h[2])};
                const v3 ro = mul(Ra, l);  // from a's base origin
                const v3 x = v3{pa[0], pa[1], pa[2]} + ro;
                const v3 ra = ro - cwa;    // arm from a's centre of mass
                const v3 loc = mulT(Rb, x - cb);
                const float pen0 = bb.h[0] - __builtin_fabsf(loc.x), pen1 = bb.h[1] - __builtin_fabsf(loc.y), pen2 = bb.h[2] - __builtin_fabsf(loc.z);
                int ks = 0;
                float pen = pen0;
                if (pen1 < pen) { pen = pen1; ks = 1; }
                if (pen2 < pen) { pen = pen2; ks = 2; }
                if (pen < -margin || n >= kPairMaxContacts) continue;
                const float lk = ks == 0 ? loc.x : (ks == 1 ? loc.y : loc.z);
                const float sg = lk < 0.0f ? -1.0f : 1.0f;
                const v3 nrm = ks == 0 ? v3{sg * Rb.m00, sg * Rb.m10, sg * Rb.m20} : (ks == 1 ? v3{sg * Rb.m01, sg * Rb.m11, sg * Rb.m21} : v3{sg * Rb.m02, sg * Rb.m12, sg * Rb.m22});
                const v3 rb = x - (v3{pb[0], pb[1], pb[2]} + cwb);
                v3 t1, t2;  // btPlaneSpace1
                if (__builtin_fabsf(nrm.z) > 0.70710678f) {
                  const float aa = fmaf(nrm.y, nrm.y, nrm.z * nrm.z), k = 1.0f / __builtin_sqrtf(aa);
                  t1 = v3{0.0f, -nrm.z * k, nrm.y * k};
                  t2 = v3{aa * k, -nrm.x * t1.z, nrm.x * t1.y};
                } else {
                  const float aa = fmaf(nrm.x, nrm.x, nrm.y * nrm.y), k = 1.0f / __builtin_sqrtf(aa);
                  t1 = v3{-nrm.y * k, nrm.x * k, 0.0f};
                  t2 = v3{-nrm.z * t1.y, nrm.z * t1.x, aa * k};
                }
                float* r = rec + n * kPairRec;
                r[0] = __int_as_float(a); r[1] = __int_as_float(b); r[2] = pen;
                r[4] = ra.x; r[5] = ra.y; r[6] = ra.z; r[7] = rb.x; r[8] = rb.y; r[9] = rb.z;
                const v3 dirs[3] = {nrm, t1, t2};
#pragma unroll
                for (int dd = 0; dd < 3; ++dd) {
                  const v3 dir = dirs[dd];
                  const v3 ga = symmul(Iwa, cross(ra, dir)), gb = symmul(Iwb, cross(rb, dir));
                  const float k = 1.0f / (im + im + dot(dir, cross(ga, ra)) + dot(dir, cross(gb, rb)));
                  float* q = r + 10 + dd * 10;  // dir (3), ga (3), gb (3), k
                  q[0] = dir.x; q[1] = dir.y; q[2] = dir.z; q[3] = ga.x; q[4] = ga.y; q[5] = ga.z; q[6] = gb.x; q[7] = gb.y; q[8] = gb.z; q[9] = k;
                }
                r[40] = 0.0f; r[41] = 0.0f; r[42] = 0.0f;  // accumulated impulses
                ++n;
              }
            }
          }
        }
      }
      // shifts start at zero
      for (int i = 0; i < A; ++i) { float* o = wvel + (wbase + i) * kPairVelStride; o[6] = 0.0f; o[7] = 0.0f; o[8] = 0.0f; }
      if (n > 0) {
        // the normal velocities the sweeps start from (restitution target)
        for (int c = 0; c < n; ++c) {
          float* r = rec + c * kPairRec;
          const float* va = wvel + (wbase + __float_as_int(r[0])) * kPairVelStride;
          const float* vb = wvel + (wbase + __float_as_int(r[1])) * kPairVelStride;
          const v3 ra{r[4], r[5], r[6]}, rb{r[7], r[8], r[9]}, dir{r[10], r[11], r[12]};
          const v3 ua = v3{va[0], va[1], va[2]} + cross(v3{va[3], va[4], va[5]}, ra), ub = v3{vb[0], vb[1], vb[2]} + cross(v3{vb[3], vb[4], vb[5]}, rb);
          r[3] = dot(ua - ub, dir);
        }
        for (int it = 0; it < iters; ++it) {
          float res = 0.0f;  // the sweep's largest row-velocity change (the residual exit: pf_params.contact_residual_threshold)
          for (int c = 0; c < n; ++c) {
            float* r = rec + c * kPairRec;
            float* va = wvel + (wbase + __float_as_int(r[0])) * kPairVelStride;
            float* vb = wvel + (wbase + __float_as_int(r[1])) * kPairVelStride;
            const v3 ra{r[4], r[5], r[6]}, rb{r[7], r[8], r[9]};
            const float depth = r[2], un0 = r[3];
            v3 Va{va[0], va[1], va[2]}, Wa{va[3], va[4], va[5]}, Vb{vb[0], vb[1], vb[2]}, Wb{vb[3], vb[4], vb[5]};
            float l0 = r[40];
#pragma unroll
            for (int dd = 0; dd < 3; ++dd) {
              const float* q = r + 10 + dd * 10;
              const v3 dir{q[0], q[1], q[2]}, ga{q[3], q[4], q[5]}, gb{q[6], q[7], q[8]};
              const float u = dot((Va + cross(Wa, ra)) - (Vb + cross(Wb, rb)), dir);
              const float target = dd == 0 ? (depth < slop ? (depth - slop) * inv_dt : (un0 < 0.0f ? -rest * un0 : 0.0f)) : 0.0f;
              const float lam = r[40 + dd];
              float nl = fmaf(target - u, q[9], lam);
              if (dd == 0) { nl = __builtin_fmaxf(nl, 0.0f); l0 = nl; }
              else { const float lim = mu * l0; nl = __builtin_fminf(__builtin_fmaxf(nl, -lim), lim); }
              const float dl = nl - lam;
              res = __builtin_fmaxf(res, __builtin_fabsf(dl) * (1.0f / q[9]));
              r[40 + dd] = nl;
              Va = Va + (im * dl) * dir; Wa = Wa + dl * ga;
              Vb = Vb - (im * dl) * dir; Wb = Wb - dl * gb;
            }
            va[0] = Va.x; va[1] = Va.y; va[2] = Va.z; va[3] = Wa.x; va[4] = Wa.y; va[5] = Wa.z;
            vb[0] = Vb.x; vb[1] = Vb.y; vb[2] = Vb.z; vb[3] = Wb.x; vb[4] = Wb.y; vb[5] = Wb.z;
          }
          if (!(res > res_bound)) break;
        }
        // position-level recovery: each body follows its deepest pair contact (the first on a tie)
        // (one slot per agent of a world, in registers: pf_ctx_create admits at most 8 agents per shared world)
        float best[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int c = 0; c < n; ++c) {
          const float* r = rec + c * kPairRec;
          const float e = r[2] - slop;
          if (e <= 0.0f) continue;
          const int a = __float_as_int(r[0]), b = __float_as_int(r[1]);
          // (best[] indexed by selects: a dynamically indexed private array would be scratch memory)
          float ba_ = 0.0f, bb_ = 0.0f;
#pragma unroll
          for (int i = 0; i < 8; ++i) { ba_ = i == a ? best[i] : ba_; bb_ = i == b ? best[i] : bb_; }
          const float h = 0.5f * erp * e;
          if (e > ba_) {
            float* o = wvel + (wbase + a) * kPairVelStride;
            o[6] = h * r[10]; o[7] = h * r[11]; o[8] = h * r[12];
#pragma unroll
            for (int i = 0; i < 8; ++i) best[i] = i == a ? e : best[i];
          }
          if (e > bb_) {
            float* o = wvel + (wbase + b) * kPairVelStride;
            o[6] = -h * r[10]; o[7] = -h * r[11]; o[8] = -h * r[12];
#pragma unroll
            for (int i = 0; i < 8; ++i) best[i] = i == b ? e : best[i];
          }
        }
      }
      if (offc) {  // ... and back: v = v_c - w x (R com)
        for (int i = 0; i < A; ++i) {
          const float* pi = wpose + (wbase + i) * 8;
          float* o = wvel + (wbase + i) * kPairVelStride;
          const v3 cw = mul(rot_from_quat(quat{pi[3], pi[4], pi[5], pi[6]}), comb);
          const v3 vb = v3{o[0], o[1], o[2]} - cross(v3{o[3], o[4], o[5]}, cw);
          o[0] = vb.x; o[1] = vb.y; o[2] = vb.z;
        }
      }
    }
    m = __ballot(todo);
  }
  lds_sync_wave();
}

// Shared world, before a physics tick: the A lanes of a world exchange pose and contact bit, test their collision boxes
// against each other (behind a bounding-sphere test) and OR the world's contact bits into the gate of the rotational drag
// (quadx.py:509). wpose: 8 floats per lane of the wave. Pd: the device copy of the parameter block (the collision boxes are
// indexed dynamically).
// at_rest: this body is not integrated any more (a wreck at rest): it still publishes its pose and reads the world's contact bit,
// but runs no box tests of its own -- two wrecks that came down within a wingspan of each other would otherwise run 36 box
// pairs x 15 axes in every tick for the rest of the episode (one such pair in 16 384 worlds made every launch 5x longer).
template <class BODY>
PF_DEV void world_exchange(BODY& b, float* wpose, const int tid, const int A, const float bound_radius, const pf_params* __restrict__ Pd,
                           const bool at_rest = false, float* wvel = nullptr, const bool frozen = false) {
  const int wbase = (tid / A) * A, wlocal = tid - wbase;
  float* me = wpose + tid * 8;
  if (wvel != nullptr && at_rest) {  // (a wreck at rest sits this tick out: what the pair stage finds for it is a body standing still)
    float* o = wvel + tid * kPairVelStride;
#pragma unroll
    for (int k = 0; k < 9; ++k) o[k] = 0.0f;
  }
  me[0] = b.p.x; me[1] = b.p.y; me[2] = b.p.z; me[3] = b.q.x; me[4] = b.q.y; me[5] = b.q.z; me[6] = b.q.w;
  me[7] = (b.contact_now ? 1.0f : 0.0f) + (at_rest ? 2.0f : 0.0f);  // (slot_contact / slot_at_rest)
  lds_sync_wave();
  bool world = false, touch = false, near = false, near_awake = false;
  // (gates only: the farthest a report or a contact point between two drones can reach -- the exact tests decide)
  const float far = __builtin_fmaxf(__builtin_fmaxf(Pd->contact_margin, Pd->contact_break_distance), Pd->contact_report_distance);
  const float rr = 2.0f * bound_radius + 1.7320508f * far, rr2 = rr * rr;
  const float rp = 2.0f * bound_radius + 2.0f * far, rp2 = rp * rp;  // within reach of the contact response between drones
  for (int j = 1; j < A; ++j) {
    int jj = wlocal + j;
    jj = jj >= A ? jj - A : jj;
    const float* o = wpose + (wbase + jj) * 8;
    world |= slot_contact(o[7]);
    const v3 d{b.p.x - o[0], b.p.y - o[1], b.p.z - o[2]};
    const float d2 = dot(d, d);
    touch |= d2 <= rr2;  // bounding spheres touch
    near |= d2 <= rp2;
    near_awake |= d2 <= rp2 && !slot_at_rest(o[7]);
  }
  // A wreck at rest that a body still in motion comes within reach of is woken (round 5): it takes part in this tick -- box tests,
  // the pair stage as a body of its own mass, the integration -- as it does in the reference, which never stopped stepping it. (Left
  // asleep, the pair stage solved contacts against it as a free body whose impulse was then thrown away: the momentum vanished.)
  // Two wrecks lying next to each other stay asleep.
  // (frozen: stopped for good by the opt-in df_freeze_wrecks, which is not the reference's behaviour -- never woken, never integrated;
  //  the pair stage still finds it standing still and the aircraft that hits it bounces off)
  b.woken = at_rest && !frozen && near_awake;
  touch = touch && (!at_rest || b.woken);
  b.world_touch = widen_to_world(near, tid, A);
  bool peer = false;
  if (__any(touch)) {
    if (touch) peer = peers_overlap_dev(Pd, wpose, wbase, wlocal, A, b.p.x, b.p.y, b.p.z, b.q, rr2);
  }
  b.world_contact = world;
  b.peer_contact = peer;
  lds_sync_wave();
}

// Body::pair_stage (declared in uav_vehicles.hpp): the generic vehicles' side of the pair stage.
PF_DEV v3 Body::pair_stage(const pf_params* Pd) {
  if (wvel_ == nullptr || Pd == nullptr) return v3{0.0f, 0.0f, 0.0f};  // (wave-uniform)
  float* o = wvel_ + wtid * kPairVelStride;
  o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = w.x; o[4] = w.y; o[5] = w.z; o[6] = 0.0f; o[7] = 0.0f; o[8] = 0.0f;
  lds_sync_wave();
  v3 shift{0.0f, 0.0f, 0.0f};
  const bool ask = world_touch && Pd->contact_response != 0;
  if (__any(ask)) {
    pair_stage_dev(Pd, wpose_, wvel_, (float*)cws, ccap, wtid, wA, ask);
    v = v3{o[0], o[1], o[2]}; w = v3{o[3], o[4], o[5]};
    shift = v3{o[6], o[7], o[8]};
  }
  return shift;
}

}  // namespace pf
