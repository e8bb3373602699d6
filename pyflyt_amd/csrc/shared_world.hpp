// shared_world.hpp -- what couples the drones of ONE Bullet world (the PettingZoo envs: pz_envs/quadx_envs/ma_quadx_base_env.py:206-241,
// pz_envs/fixedwing_envs/ma_fixedwing_base_env.py): the A lanes of a world are adjacent lanes of one wavefront and exchange
// pose and contact bit through LDS before every physics tick.
#pragma once
#include "../../include/pyflyt_amd.h"
#include "uav_device.hpp"

namespace pf {

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
  for (int j = 1; j < A; ++j) {
    int jj = wlocal + j;
    jj = jj >= A ? jj - A : jj;
    const float* o = wpose + (wbase + jj) * 8;
    const v3 d{px - o[0], py - o[1], pz - o[2]};
    if (dot(d, d) <= rr2) {
      const m3 Rb = rot_from_quat(quat{o[3], o[4], o[5], o[6]});
      const m3 Rrel{Rb.m00 * Ra.m00 + Rb.m10 * Ra.m10 + Rb.m20 * Ra.m20, Rb.m00 * Ra.m01 + Rb.m10 * Ra.m11 + Rb.m20 * Ra.m21, Rb.m00 * Ra.m02 + Rb.m10 * Ra.m12 + Rb.m20 * Ra.m22,
                    Rb.m01 * Ra.m00 + Rb.m11 * Ra.m10 + Rb.m21 * Ra.m20, Rb.m01 * Ra.m01 + Rb.m11 * Ra.m11 + Rb.m21 * Ra.m21, Rb.m01 * Ra.m02 + Rb.m11 * Ra.m12 + Rb.m21 * Ra.m22,
                    Rb.m02 * Ra.m00 + Rb.m12 * Ra.m10 + Rb.m22 * Ra.m20, Rb.m02 * Ra.m01 + Rb.m12 * Ra.m11 + Rb.m22 * Ra.m21, Rb.m02 * Ra.m02 + Rb.m12 * Ra.m12 + Rb.m22 * Ra.m22};
      const int nb = Pd->n_boxes;
      for (int k = 0; k < nb; ++k) {
        for (int l = 0; l < nb; ++l) {
          const pf_box bk = Pd->boxes[k], bl = Pd->boxes[l];
          const v3 ca = d + mul(Ra, v3{bk.c[0], bk.c[1], bk.c[2]}) - mul(Rb, v3{bl.c[0], bl.c[1], bl.c[2]});
          peer |= box_overlaps_aabb(mulT(Rb, ca), Rrel, bk.h, v3{0.f, 0.f, 0.f}, bl.h);
        }
      }
    }
  }
  return peer;
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
                           const bool at_rest = false) {
  const int wbase = (tid / A) * A, wlocal = tid - wbase;
  float* me = wpose + tid * 8;
  me[0] = b.p.x; me[1] = b.p.y; me[2] = b.p.z; me[3] = b.q.x; me[4] = b.q.y; me[5] = b.q.z; me[6] = b.q.w;
  me[7] = b.contact_now ? 1.0f : 0.0f;
  lds_sync_wave();
  bool world = false, touch = false;
  const float rr = 2.0f * bound_radius, rr2 = rr * rr;
  for (int j = 1; j < A; ++j) {
    int jj = wlocal + j;
    jj = jj >= A ? jj - A : jj;
    const float* o = wpose + (wbase + jj) * 8;
    world |= o[7] != 0.0f;
    const v3 d{b.p.x - o[0], b.p.y - o[1], b.p.z - o[2]};
    touch |= dot(d, d) <= rr2;  // bounding spheres touch
  }
  touch = touch && !at_rest;
  bool peer = false;
  if (__any(touch)) {
    if (touch) peer = peers_overlap_dev(Pd, wpose, wbase, wlocal, A, b.p.x, b.p.y, b.p.z, b.q, rr2);
  }
  b.world_contact = world;
  b.peer_contact = peer;
  lds_sync_wave();
}

}  // namespace pf
