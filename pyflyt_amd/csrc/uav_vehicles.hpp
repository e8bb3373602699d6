// uav_vehicles.hpp -- per-lane "vehicle programs": the batched equivalent of the reference's
// DroneClass contract (core/abstractions/base_drone.py:139-241: reset / update_control /
// update_physics / update_state) fused with the Bullet free-body tick, one drone per lane, fp32.
#pragma once
#include "../../include/pyflyt_amd.h"
#include "uav_device.hpp"
#include "quadx_control_d.hpp"

namespace pf {

// ------------------------------------------------------------------------------------------
// Motor-noise source. INJECT reads raw draws xi ~ N(num_motors, 1) laid out [tick][n] (coalesced);
// PHILOX derives them from (seed, global lane, event counter, call, stream) -- same integer stream
// as the oracle. motors.py:134-138.
struct Noise {
  int mode;
  const float* xi;  // current phase's injected draws
  int n, lane;
  uint32_t k0, k1, c0, c1, stream;
  float nmot;
  int cached;
  f8 z;
  PF_DEV void begin_event(uint32_t ctr, uint32_t strm, const float* inj) {
    c1 = ctr; stream = strm; xi = inj; cached = -1;
  }
  PF_DEV float get(int flat) {
    if (mode == PF_NOISE_OFF) return 0.0f;
    if (mode == PF_NOISE_INJECT) return xi[(size_t)flat * n + lane];
    int call = flat >> 3;
    if (call != cached) {
      z = normal8(philox4x32(k0, k1, c0, c1, (uint32_t)call, stream));
      cached = call;
    }
    return nmot + pick8(z, (uint32_t)flat & 7u);
  }
  PF_DEV float uniform(int flat, uint32_t strm) const {  // uncached, for reset-time sampling
    f4 u = uniform4(philox4x32(k0, k1, c0, c1, (uint32_t)(flat >> 2), strm));
    return pick4(u, (uint32_t)flat & 3u);
  }
};

// abstractions/pid.py:70-94 for one component
PF_DEV float pid1(float kp, float ki, float kd, float lim, float T, float invT, float& I, float& E, float st, float sp) {
  float e = sp - st;
  I = clampf(fmaf(ki * e, T, I), -lim, lim);
  float d = kd * (e - E) * invT;
  E = e;
  return clampf(kp * e + I + d, -lim, lim);
}

// ------------------------------------------------------------------------------------------
// Contact response against the ground slab: what stepSimulation does after collision detection, as the
// named-parameter model documented at pf_params.contact_response (oracle/uav_oracle.c:contact_solve is the fp64
// restatement, oracle/fake_bullet.py:_solve_contacts the independent second one). Contact vertices within the contact
// margin of the slab's top face at the pre-integration pose (p, q); contact_iters projected Gauss-Seidel sweeps on the COM
// velocity / angular velocity (world frame), in collider / vertex order; returns the new base twist and the deepest
// penetration.
//
// How it is laid out for one wavefront lane per body (round 3; round 2's version spent ~60 instructions and eleven dependent
// LDS reads per contact per sweep, profiles/README.md):
//   * a lone wave issues one instruction per four clocks whatever it is, and the Gauss-Seidel recurrence is serial in the
//     contacts, so the time of a solve IS (contacts x sweeps x instructions per row triple): everything that does not depend
//     on the running twist is computed ONCE per contact into a 20-float record -- arm, the normal row's target velocity, and
//     for each of the three rows the angular response I_w^-1 (a x e_d) and the inverse effective mass -- so that a row is
//     2 (row velocity) + 2..3 (projected impulse) + 1 (delta) + 4 (twist update) instructions;
//   * the records live in LDS (dynamically indexed; as private arrays they would be scratch memory) as six float4 each, read
//     with six ds_read_b128 ONE CONTACT AHEAD of their use (two register sets, the loop unrolled by two), so the sweep never
//     waits on LDS; the only write per contact and sweep is its three accumulated impulses;
//   * impulses are kept in VELOCITY units (round 4): lambda' = lambda / (effective mass of the row), the record holds the
//     responses per unit of row velocity (g k, k / m) and the friction rows' cone factors mu k_z / k_x, mu k_z / k_y. The row's
//     velocity change -- what the solver's residual exit (pf_params.contact_residual_threshold) is a bound on -- is then the
//     impulse change itself, and the projected impulse is an add and a max;
//   * a lane whose own sweep met the residual bound is DONE: it reads the sentinel record from then on (its rows are exact
//     no-ops), so lanes of a wave that solve side by side each stop after the sweep the scalar restatements stop after;
//   * vertex generation per box: the eight heights come from seven partial sums of the box's half axes' z components, a box
//     whose lowest vertex clears the margin is skipped after four instructions, and only a vertex that passes the height test
//     gets its x / y offsets and its record;
//   * a contact whose accumulated impulse is zero and whose row velocity is already above its target (a speculative contact
//     that is not closing: most of the vertices a tumbling airframe has within the margin) is skipped after six instructions
//     when that holds for every lane in the sweep at that moment -- the full update would compute exact zeros;
//   * a sweep that moved nothing ends the solve (every further sweep would repeat it exactly).
struct ContactOut {
  v3 v, w;
  float deepest;
};
// Per-contact record, six float4: (arm.xyz, normal target velocity) (gz kz, kz / m) (gx kx, kx / m) (gy ky, ky / m)
// (ln', lx', ly', mu kz / kx) (mu kz / ky, -, -, -); g_d = I_w^-1 (arm x e_d), k_d the row's effective mass.
// Only a few lanes of a wave need the solver in the same tick, so the callers cut their LDS (the hot kernels: the observation
// tile, idle during the physics ticks) into regions sized for the airframe's own worst-case contact count and deal them out by
// prefix sum over what each lane actually needs (contact_solve_impl), coming back for another round if not everything fits.
constexpr int kContactWords = 24;
constexpr int kContactF4 = kContactWords / 4;
constexpr int kContactSlotFloats = (PF_MAX_CONTACTS + 1) * kContactWords;  // worst-case region (+ the sentinel record): 1176 floats
typedef __attribute__((address_space(3))) float* lds_fptr;
typedef __attribute__((address_space(3))) pf_f4v* lds_f4ptr;  // (the native vector type: HIP's float4 class has no LDS-qualified assignment)
// The parameter block as the solver reads it: a wave-uniform pointer into the constant address space, so that every field --
// and the collision boxes, indexed by a uniform loop counter -- comes through the scalar cache (s_load). Inside an out-of-line
// function the plain pointer argument lives in VGPRs, and its fields were a chain of flat loads at full memory latency with one
// lane active.
typedef const pf_params __attribute__((address_space(4)))* pf_params_kptr;
PF_DEV pf_params_kptr uniform_params(const pf_params* P) {
  const uintptr_t a = reinterpret_cast<uintptr_t>(P);
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a), hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
  return (pf_params_kptr)(((uintptr_t)hi << 32) | (uintptr_t)lo);
}
// Rarely executed out-of-line code (the contact solve, the exact floor tests): in an env launch a handful of single lanes call it,
// each on its own CU, and the launch lasts as long as its slowest wave. One wave fetching 10 KB of instructions that are in no
// cache pays for every line on its own -- measured: after anything that swept the L2s (a rollout, a policy network between two
// env steps), +3 us per launch averaged over the next twenty launches, single launches at 2x. The functions live in one section
// of the code object (the linker brackets it with __start_ / __stop_ symbols), and the first workgroups of every launch -- one
// lands on each XCD, each XCD has its own L2 -- read the section once: the calls then miss the instruction cache into a warm L2.
#define PF_RARE_TEXT __attribute__((section("pf_rare_text")))
extern "C" __device__ const char __start_pf_rare_text[];
extern "C" __device__ const char __stop_pf_rare_text[];
constexpr int kRareTextPrefetchBlocks = 16;  // (workgroups go round the 8 XCDs; twice that, whatever XCD a dispatch starts on)
PF_DEV void rare_text_prefetch(const int lane) {
  const char* const a = __start_pf_rare_text;
  const long n = (long)(__stop_pf_rare_text - a);
  for (long base = 0; base < n; base += 2 * 64 * 128) {  // (uniform; one round covers 16 KB: two 128-byte lines per lane)
    long o0 = base + lane * 128, o1 = o0 + 64 * 128;
    o0 = o0 < n - 4 ? o0 : n - 4;
    o1 = o1 < n - 4 ? o1 : n - 4;
    uint32_t r0, r1;
    // (the wait is inside the statement: the compiler does not track loads issued by inline assembly)
    asm volatile("global_load_dword %0, %2, off\n\tglobal_load_dword %1, %3, off\n\ts_waitcnt vmcnt(0)" : "=&v"(r0), "=&v"(r1) : "v"(a + o0), "v"(a + o1) : "memory");
  }
}
// (r05, measured and dropped: the same through ONE plain, branch-free load per wave -- 136 waves requesting one 128-byte line each,
//  consumed in front of the state stores -- made every launch slower than no prefetch at all: Hover 9.69 -> 9.89 us, Fixedwing-
//  Waypoints 20.7 -> 21.4. The Fixedwing-Waypoints kernel, whose waves all take the same time, does without the prefetch: the
//  sixteen prefetching waves, waiting for their rounds and with them for every state group in flight, were its slowest -- 21.1 ->
//  20.7 us. profiles/tools/r05/g19.sh, g20.sh)
// The per-body part of the solve that does not depend on where the contact vertices come from: records are appended with
// add(), then sweeps() runs the projected Gauss-Seidel iteration. W4: this lane's LDS region.
struct ContactSet {
  lds_f4ptr W4;
  int n;
  float deepest;
  v3 cw;                // R com: the arms are taken from the centre of mass
  float I0, I1, I2, I3, I4, I5;  // world-frame inverse inertia R I^-1 R^T (symmetric xx xy xz yy yz zz)
  float inv_mass, slop, inv_dt, rest, mu;
  v3 vc, w;             // the running twist: COM velocity, angular velocity
  float res;            // the running sweep's largest row-velocity change (magnitude)
#if defined(PF_PHASE_TRACE) && defined(PF_SOLVER_TRACE)  // (the general solver's own counters: a second switch since r05 -- with them in the generic env kernel ROCm 7.2 aborts)
  int sweeps_done = 0, rows_full = 0, rows_skipped = 0;
#endif
  PF_DEV void begin(lds_fptr ws, const m3& R, v3 com, float im, v3 v_, v3 w_, float i0, float i1, float i2, float i3, float i4, float i5,
                    float slop_, float inv_dt_, float rest_, float mu_) {
    W4 = (lds_f4ptr)ws; n = 0; deepest = 0.0f; mu = mu_; res = 0.0f;
    cw = mul(R, com);
    const float Ii[6] = {i0, i1, i2, i3, i4, i5};
    const v3 r0{R.m00, R.m01, R.m02}, r1{R.m10, R.m11, R.m12}, r2{R.m20, R.m21, R.m22};
    const v3 c0 = symmul(Ii, r0), c1 = symmul(Ii, r1), c2 = symmul(Ii, r2);
    I0 = dot(r0, c0); I1 = dot(r0, c1); I2 = dot(r0, c2); I3 = dot(r1, c1); I4 = dot(r1, c2); I5 = dot(r2, c2);
    inv_mass = im; slop = slop_; inv_dt = inv_dt_; rest = rest_;
    w = w_;
    vc = v_ + cross(w_, cw);
  }
  // A contact vertex at world offset `off` from the base origin, `depth` below the slab's top face (negative: above it, inside
  // the margin). The three constraint directions are the world axes (normal +z, friction +x, +y), written out component by
  // component: with the direction as a vector the compiler may not drop the multiplications by its zeros.
  PF_DEV void add(v3 off, float depth) {
    const v3 a = off - cw;
    // angular response I_w^-1 (a x e_d) of a unit impulse along e_d at arm a, and the row's inverse effective mass
    // 1/m + e_d . ((I_w^-1 (a x e_d)) x a)
    const v3 gz{fmaf(I0, a.y, -(I1 * a.x)), fmaf(I1, a.y, -(I3 * a.x)), fmaf(I2, a.y, -(I4 * a.x))};
    const v3 gx{fmaf(I1, a.z, -(I2 * a.y)), fmaf(I3, a.z, -(I4 * a.y)), fmaf(I4, a.z, -(I5 * a.y))};
    const v3 gy{fmaf(I2, a.x, -(I0 * a.z)), fmaf(I4, a.x, -(I1 * a.z)), fmaf(I5, a.x, -(I2 * a.z))};
    const float dz = inv_mass + fmaf(gz.x, a.y, -(gz.y * a.x));
    const float dx = inv_mass + fmaf(gx.y, a.z, -(gx.z * a.y));
    const float dy = inv_mass + fmaf(gy.z, a.x, -(gy.x * a.z));
    const float kz = frcp(dz), kx = frcp(dx), ky = frcp(dy);
    // normal row: may close the gap down to the slop, no more; otherwise towards restitution x approach speed
    const float vn0 = vc.z + fmaf(w.x, a.y, -(w.y * a.x));
    const float tgt = depth < slop ? (depth - slop) * inv_dt : (vn0 < 0.0f ? -rest * vn0 : 0.0f);
    const float mz = mu * kz;
    lds_f4ptr r = W4 + kContactF4 * n;
    r[0] = pf_f4v{a.x, a.y, a.z, tgt};
    r[1] = pf_f4v{gz.x * kz, gz.y * kz, gz.z * kz, inv_mass * kz};
    r[2] = pf_f4v{gx.x * kx, gx.y * kx, gx.z * kx, inv_mass * kx};
    r[3] = pf_f4v{gy.x * ky, gy.y * ky, gy.z * ky, inv_mass * ky};
    r[4] = pf_f4v{0.0f, 0.0f, 0.0f, mz * dx};
    r[5] = pf_f4v{mz * dy, 0.0f, 0.0f, 0.0f};
    deepest = __builtin_fmaxf(deepest, depth);
    ++n;
  }
  // one contact of one sweep; returns 0 when it was skipped as idle in every lane, 1 when it ran (wave-uniform)
  PF_DEV uint32_t row3(const pf_f4v r0, const pf_f4v r1, const pf_f4v r2, const pf_f4v r3, const pf_f4v r4, const float cy, lds_f4ptr lout) {
    const float un = fmaf(w.x, r0.y, fmaf(-w.y, r0.x, vc.z));
    // (a contact with no accumulated impulse whose normal velocity is not below its target stays at exactly zero: n0 = 0,
    //  friction clamped to +-0; skipped when every lane of the wave agrees -- which includes the lanes that are past their
    //  last contact, or done, and sit on the sentinel record)
    const bool idle = (r4.x == 0.0f) && (un >= r0.w);
#if defined(PF_PHASE_TRACE) && defined(PF_SOLVER_TRACE)  // (the general solver's own counters: a second switch since r05 -- with them in the generic env kernel ROCm 7.2 aborts)
    if (__builtin_amdgcn_ballot_w64(!idle) == 0ull) { rows_skipped += 1; return 0u; }
    rows_full += 1;
#else
    if (__builtin_amdgcn_ballot_w64(!idle) == 0ull) return 0u;
#endif
    // (velocity units: the unclamped new impulse is the old one plus the row's velocity error)
    const float n0 = __builtin_fmaxf((r0.w - un) + r4.x, 0.0f);
    const float d0 = n0 - r4.x;
    vc.z = fmaf(r1.w, d0, vc.z);
    w = v3{fmaf(d0, r1.x, w.x), fmaf(d0, r1.y, w.y), fmaf(d0, r1.z, w.z)};
    const float limx = r4.w * n0;  // friction rows: clamped to mu x the normal impulse (in their own velocity units)
    const float ux = fmaf(w.y, r0.z, fmaf(-w.z, r0.y, vc.x));
    const float n1 = med3(r4.y - ux, -limx, limx);
    const float d1 = n1 - r4.y;
    vc.x = fmaf(r2.w, d1, vc.x);
    w = v3{fmaf(d1, r2.x, w.x), fmaf(d1, r2.y, w.y), fmaf(d1, r2.z, w.z)};
    const float limy = cy * n0;
    const float uy = fmaf(w.z, r0.x, fmaf(-w.x, r0.z, vc.y));
    const float n2 = med3(r4.z - uy, -limy, limy);
    const float d2 = n2 - r4.z;
    vc.y = fmaf(r3.w, d2, vc.y);
    w = v3{fmaf(d2, r3.x, w.x), fmaf(d2, r3.y, w.y), fmaf(d2, r3.z, w.z)};
    res = __builtin_fmaxf(__builtin_fmaxf(res, __builtin_fabsf(d0)), __builtin_fmaxf(__builtin_fabsf(d1), __builtin_fabsf(d2)));
    // (r4.w is written back as it came: that keeps its register out of the allocator's hands until the row is done -- reused
    //  earlier, it made the row wait for the whole prefetch it belongs to)
    *lout = pf_f4v{n0, n1, n2, r4.w};
    return 1u;  // (the row ran)
  }
  // Wave-uniform control flow: the contact counter and the sweep counter are scalars, a lane that is past its last contact reads
  // its sentinel record (zero impulse, target -FLT_MAX, zero responses: the row update is an exact no-op), a lane whose sweep
  // met the residual bound is done and reads nothing but the sentinel from then on, and the solve ends when every lane is done.
  // res_bound: sqrt(pf_params.contact_residual_threshold) -- the bound on a row's velocity change; 0: only a sweep that moved
  // nothing ends a lane's solve (every further sweep would repeat it exactly). No exec-mask bookkeeping in the loop: on a lone
  // wave a scalar instruction costs an issue slot like any other.
  PF_DEV void sweeps(const int iters, const float res_bound) {
    W4[kContactF4 * n + 0] = pf_f4v{0.0f, 0.0f, 0.0f, -3.4028235e38f};
    W4[kContactF4 * n + 1] = pf_f4v{0.0f, 0.0f, 0.0f, 0.0f};
    W4[kContactF4 * n + 2] = pf_f4v{0.0f, 0.0f, 0.0f, 0.0f};
    W4[kContactF4 * n + 3] = pf_f4v{0.0f, 0.0f, 0.0f, 0.0f};
    W4[kContactF4 * n + 4] = pf_f4v{0.0f, 0.0f, 0.0f, 0.0f};
    W4[kContactF4 * n + 5] = pf_f4v{0.0f, 0.0f, 0.0f, 0.0f};
    if (!__any(n > 0)) return;
    typedef __attribute__((address_space(3))) char* lds_cptr;
    constexpr uint32_t kRecBytes = 4u * kContactWords;
    const uint32_t end_all = (uint32_t)n * kRecBytes;  // byte offset of the sentinel
    // the largest contact count among the lanes in here, as a scalar (bit by bit from the top: n <= PF_MAX_CONTACTS < 64): the
    // contact loop then runs on scalar compares -- on a lone wave every vector compare + ballot + branch it does not need is
    // some thirty clocks per contact row
    int nmax = 0;
#pragma unroll
    for (int b = 5; b >= 0; --b)
      if (__ballot(n >= (nmax | (1 << b))) != 0ull) nmax |= 1 << b;
    const int npair = (nmax + 1) >> 1;
    bool done = n == 0;
    for (int it = 0; it < iters; ++it) {
      // (a done lane: base on the sentinel, no further records)
      const lds_cptr base = (lds_cptr)W4 + (done ? end_all : 0u);
      const uint32_t end = done ? 0u : end_all;
      auto rec = [&](uint32_t off) { return (lds_f4ptr)(base + (off < end ? off : end)); };
      res = 0.0f;
      // two register sets, each loaded one contact ahead of its use; contacts in pairs (for an odd count the last row of every lane
      // is the sentinel's, skipped as idle; a scalar exit between the two rows of a pair cost more than it saved: 19.8 -> 21.0 us per
      // tick for landed quadrotors)
      // (the sixth quad of a record holds one live word, the y cone factor: read as ONE dword. Read as a quad, its three dead
      //  registers were handed out as temporaries of the row that runs while the prefetch is in flight -- a write-after-write on
      //  the pending load's destination, for which the compiler waits for the WHOLE prefetch: lgkmcnt(0) in front of every second
      //  row, the LDS latency the prefetch exists to hide)
      lds_f4ptr p0 = rec(0u);
      pf_f4v a0 = p0[0], a1 = p0[1], a2 = p0[2], a3 = p0[3], a4 = p0[4];
      float a5 = ((lds_fptr)p0)[20];
      uint32_t off = 0u;  // (wave-uniform)
      for (int c = 0; c < npair; ++c) {
        lds_f4ptr pa = rec(off), nb = rec(off + kRecBytes);
        const pf_f4v b0 = nb[0], b1 = nb[1], b2 = nb[2], b3 = nb[3], b4 = nb[4];
        const float b5 = ((lds_fptr)nb)[20];
        row3(a0, a1, a2, a3, a4, a5, pa + 4);
        lds_f4ptr na = rec(off + 2u * kRecBytes);
        a0 = na[0]; a1 = na[1]; a2 = na[2]; a3 = na[3]; a4 = na[4]; a5 = ((lds_fptr)na)[20];
        row3(b0, b1, b2, b3, b4, b5, nb + 4);
        off += 2u * kRecBytes;
      }
#if defined(PF_PHASE_TRACE) && defined(PF_SOLVER_TRACE)  // (the general solver's own counters: a second switch since r05 -- with them in the generic env kernel ROCm 7.2 aborts)
      sweeps_done = it + 1;
#endif
      done = done || !(res > res_bound);
      if (__ballot(!done) == 0ull) break;
    }
  }
  PF_DEV ContactOut finish(v3 v_in, v3 w_in) const {
    if (n == 0) return ContactOut{v_in, w_in, 0.0f};
    return ContactOut{vc - cross(w, cw), w, __builtin_fmaxf(deepest - slop, 0.0f)};
  }
};
#ifdef PF_PHASE_TRACE
// diagnostic build only (profiles/tools/solver_trace.py): calls, shader-clock cycles in setup / in the sweeps, contacts and
// active lanes per call, summed by the first active lane of every call
__device__ unsigned long long g_solver_trace[8];
// (the diagnostic counters are bumped through a pointer in the GLOBAL address space: no "may this be private memory?" expansion
//  of the 64-bit atomic)
PF_DEV void trace_add(unsigned long long* p, const unsigned long long v) {
  __hip_atomic_fetch_add((__attribute__((address_space(1))) unsigned long long*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
#endif
// The vertices of one collision box (centre offset cwk in the world frame, half extents bh, link yawed by (cy, sy) about the base
// z axis) that lie within the contact margin of the slab's top face: f(off, z) in vertex order.
template <class F>
PF_DEV void box_contact_vertices(const v3 p, const m3& R, const v3 cwk, const float cy, const float sy, const float bh0, const float bh1, const float bh2,
                                 const float hxy, const float hz2, const float margin, const bool all8, F&& f) {
  // the link axes in the world frame (the link frame is the base frame yawed about z): their z components choose the incident
  // face, the axes scaled by the half extents are the half axes
  const float zx = fmaf(R.m20, cy, R.m21 * sy), zy = fmaf(R.m21, cy, -(R.m20 * sy)), zz = R.m22;
  const v3 ex{bh0 * fmaf(R.m00, cy, R.m01 * sy), bh0 * fmaf(R.m10, cy, R.m11 * sy), bh0 * zx};
  const v3 ey{bh1 * fmaf(R.m01, cy, -(R.m00 * sy)), bh1 * fmaf(R.m11, cy, -(R.m10 * sy)), bh1 * zy};
  const v3 ez{bh2 * R.m02, bh2 * R.m12, bh2 * zz};
  const float zc = p.z + cwk.z;
  if (zc - (__builtin_fabsf(ex.z) + __builtin_fabsf(ey.z) + __builtin_fabsf(ez.z)) > margin) return;  // the whole box clears the margin
  // manifold reduction (pf_params.contact_manifold_points = 4): only the four vertices of the face that looks down the most --
  // the axis with the largest |z component| (the first on a tie), the face on its + side when the axis points down. As a mask
  // over the vertex index (x sign bit 0, y sign bit 1, z sign bit 2).
  uint32_t keep = 0xffu;
  if (!all8) {
    const float ax = __builtin_fabsf(zx), ay = __builtin_fabsf(zy), az = __builtin_fabsf(zz);
    const bool use_y = ay > ax, use_z = az > __builtin_fmaxf(ax, ay);
    const uint32_t lo = use_z ? 0x0fu : (use_y ? 0x33u : 0x55u);  // the vertices on the - side of the axis
    const float zsel = use_z ? zz : (use_y ? zy : zx);
    keep = zsel < 0.0f ? (lo ^ 0xffu) : lo;
  }
  const float za0 = zc - ex.z, za1 = zc + ex.z;
  const float zb[4] = {za0 - ey.z, za1 - ey.z, za0 + ey.z, za1 + ey.z};
#pragma unroll
  for (int i = 0; i < 8; ++i) {  // vertex i: x sign bit 0, y sign bit 1, z sign bit 2 (the oracle's order)
    const float z = (i & 4) ? zb[i & 3] + ez.z : zb[i & 3] - ez.z;
    if (((keep >> i) & 1u) != 0u && z <= margin && z >= -hz2) {
      const float sx = (i & 1) ? 1.0f : -1.0f, syv = (i & 2) ? 1.0f : -1.0f, sz = (i & 4) ? 1.0f : -1.0f;
      const v3 off{fmaf(sx, ex.x, fmaf(syv, ey.x, fmaf(sz, ez.x, cwk.x))), fmaf(sx, ex.y, fmaf(syv, ey.y, fmaf(sz, ez.y, cwk.y))), z - p.z};
      if (__builtin_fabsf(p.x + off.x) <= hxy && __builtin_fabsf(p.y + off.y) <= hxy) f(off, z);
    }
  }
}
// The contact vertices of the airframe at pose (p, R), in collider / vertex order: f(off, z) for every collider vertex within the
// contact margin of the slab's top face (off: world offset from the base origin, z: its height). Per box the eight heights come
// from seven partial sums of the half axes' z components; a box whose lowest vertex clears the margin costs four instructions.
// Used twice per solve with identical arithmetic: to count the contacts (so that the LDS records can be packed) and to fill them.
template <class F>
PF_DEV void for_each_contact_vertex(const pf_params_kptr P, const float hxy, const float hz2, const float margin, const bool all8, const int nb, const v3 p, const m3& R, F&& f) {
  for (int k = 0; k < nb; ++k) {
    const float bc0 = P->boxes[k].c[0], bc1 = P->boxes[k].c[1], bc2 = P->boxes[k].c[2];
    const float bh0 = P->boxes[k].h[0], bh1 = P->boxes[k].h[1], bh2 = P->boxes[k].h[2];
    const int kind = P->boxes[k].kind;
    const float yaw = P->boxes[k].yaw;
    float sy = 0.0f, cy = 1.0f;
    if (yaw != 0.0f) sincosf(yaw, &sy, &cy);  // (wave-uniform)
    const v3 cwk = mul(R, v3{bc0, bc1, bc2});
    if (kind == 1) {  // cylinder: 8 rim points per end disc, -z then +z; rim point j at j * 45 degrees from the link x axis
#pragma unroll 1
      for (int i = 0; i < 16; ++i) {
        const int j = i & 7;
        const float c45 = (j == 0) ? 1.0f : ((j == 4) ? -1.0f : ((j == 2 || j == 6) ? 0.0f : ((j == 1 || j == 7) ? 0.70710678f : -0.70710678f)));
        const int js = (j + 6) & 7;
        const float s45 = (js == 0) ? 1.0f : ((js == 4) ? -1.0f : ((js == 2 || js == 6) ? 0.0f : ((js == 1 || js == 7) ? 0.70710678f : -0.70710678f)));
        const float l0 = bh0 * c45, l1 = bh0 * s45, l2 = (i >> 3) ? bh2 : -bh2;
        const v3 off = cwk + mul(R, v3{cy * l0 - sy * l1, sy * l0 + cy * l1, l2});
        const v3 x = p + off;
        if (x.z <= margin && x.z >= -hz2 && __builtin_fabsf(x.x) <= hxy && __builtin_fabsf(x.y) <= hxy) f(off, x.z);
      }
      continue;
    }
    box_contact_vertices(p, R, cwk, cy, sy, bh0, bh1, bh2, hxy, hz2, margin, all8, f);
  }
}
// Where the solve reads the world from: the device parameter block through the scalar cache. The contact model's constants
// are all requested in ONE batch when the source is built (the statements below are the schedule: the build runs with the
// machine scheduler off) -- read where they are used they were six dependent scalar-load round trips, and in the hover task,
// where a solve is rare, each of them missed the scalar cache while the whole launch waited for that one wave.
struct ParamContactSrc {
  pf_params_kptr P;
  float slop_, inv_dt_, rest_, mu_, hxy_, hz2_, margin_, brk_, res_bound_;
  int iters_, nb_, worst_;
  bool all8_;
  PF_DEV explicit ParamContactSrc(pf_params_kptr p) : P(p) {
    worst_ = p->contact_max_points;
    const float dt = p->dt, hz = p->plane_half_z, thr = p->contact_residual_threshold;
    slop_ = p->contact_slop; rest_ = p->contact_restitution; mu_ = p->contact_friction; iters_ = p->contact_iters;
    hxy_ = p->plane_half_xy; margin_ = p->contact_margin; brk_ = p->contact_break_distance; nb_ = p->n_boxes;
    all8_ = p->contact_manifold_points >= 8;
    inv_dt_ = 1.0f / dt; hz2_ = 2.0f * hz;
    res_bound_ = __builtin_sqrtf(thr);  // (wave-uniform: one scalar-side conversion per call)
  }
  // reach: how far above the face a vertex may be and still be a contact point -- the margin, or the breaking distance for a
  // body that held contact points after the previous tick (per lane)
  template <class F> PF_DEV void for_each(const v3 p, const m3& R, const float reach, F&& f) const { for_each_contact_vertex(P, hxy_, hz2_, reach, all8_, nb_, p, R, f); }
  PF_DEV float reach(const bool persisted) const { return persisted ? brk_ : margin_; }
  PF_DEV float slop() const { return slop_; }
  PF_DEV float inv_dt() const { return inv_dt_; }
  PF_DEV float rest() const { return rest_; }
  PF_DEV float mu() const { return mu_; }
  PF_DEV float res_bound() const { return res_bound_; }
  PF_DEV int iters() const { return iters_; }
  PF_DEV int worst() const { return worst_; }  // the airframe's worst-case contact count (pf_params.contact_max_points)
};

PF_DEV int wave_inclusive_scan_asking(const bool need, const int sz) {
  unsigned long long m = __ballot(need);
  const int lane = (int)(threadIdx.x & 63u);
  // a homogeneous population (every asking lane has the same contact count: bodies at rest on the floor): rank x size
  // (m != 0: the caller loops while some lane asks)
  const int szf = __builtin_amdgcn_readlane(sz, __ffsll((long long)m) - 1);
  if (__ballot(need && sz != szf) == 0ull) {
    const int rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));  // asking lanes below this one
    return need ? (rank + 1) * szf : 0;
  }
  int run = 0, mine = 0;
  while (m != 0ull) {
    const int l = __ffsll((long long)m) - 1;  // (wave-uniform)
    run += __builtin_amdgcn_readlane(sz, l);
    mine = lane == l ? run : mine;
    m &= m - 1ull;
  }
  return mine;
}
// One call per tick for the whole wave (every lane that is active at the call site enters; `need`: this lane asks for a solve):
//   1. count this lane's contact vertices -- a lane without any is done (the gate of the call is a conservative bound);
//   2. pack the records of the lanes that ask into the LDS behind `ws` (cap_floats of it) by a wave prefix sum over their
//      (contacts + 1 sentinel) x 20 floats -- as many bodies per round as actually fit, not as many as would fit if each had its
//      airframe's worst case (a landed aeroplane touches with 4 of its 48 collider vertices: six rounds became one);
//   3. fill the records and run the sweeps; lanes that did not fit come back in another round.
template <class SRC>
PF_DEV ContactOut contact_solve_impl(const SRC src, lds_fptr ws, const int cap_floats, bool need, const bool persisted, v3 p, const m3 R, v3 v, v3 w, float inv_mass, v3 com,
                                     float i0, float i1, float i2, float i3, float i4, float i5) {
  const float reach = src.reach(persisted);
#if defined(PF_PHASE_TRACE) && defined(PF_SOLVER_TRACE)  // (the general solver's own counters: a second switch since r05 -- with them in the generic env kernel ROCm 7.2 aborts)
  const unsigned long long pf_t0 = __builtin_readcyclecounter();
  unsigned long long pf_sweep = 0, pf_fill = 0;
#endif
  ContactOut out{v, w, 0.0f};
  // Few lanes ask (the env tasks: a handful of single-lane solves per launch, and the launch waits for each of them): when every
  // asking lane fits with its airframe's worst-case region there is nothing to pack -- no count pass, rank x worst-case size.
  const int worst_sz = (src.worst() + 1) * kContactWords;
  const bool roomy = __popcll(__ballot(need)) * worst_sz <= cap_floats;  // (wave-uniform)
  // Many lanes ask and their worst case is large (landed aeroplanes: 4 of 48 collider vertices touch): instead of counting first,
  // assume kOptimistic contacts each -- when all asking lanes fit with that -- and let the fill itself find the lanes that have
  // more; they come back in the next round with their real count. (The count pass was a sixth of a landed aeroplane's solve.)
  constexpr int kOptimistic = 8;
  const bool optimistic = !roomy && src.worst() > kOptimistic && __popcll(__ballot(need)) * (kOptimistic + 1) * kContactWords <= cap_floats;
  int n = roomy ? PF_MAX_CONTACTS : kOptimistic;
  if (!roomy && !optimistic) {
    n = 0;
    if (need) src.for_each(p, R, reach, [&](v3, float) { n += 1; });
    n = n > PF_MAX_CONTACTS ? PF_MAX_CONTACTS : n;
    need = need && n > 0;
  }
  int sz = need ? (roomy ? worst_sz : (n + 1) * kContactWords) : 0;
#if defined(PF_PHASE_TRACE) && defined(PF_SOLVER_TRACE)  // (the general solver's own counters: a second switch since r05 -- with them in the generic env kernel ROCm 7.2 aborts)
  const unsigned long long pf_tc = __builtin_readcyclecounter();
#endif
  ContactSet S;
  S.n = 0;
  while (__any(need)) {
    const int incl = wave_inclusive_scan_asking(need, sz);
    // (the first lane that asks always fits: the callers' LDS holds at least one worst-case region)
    if (need && incl <= cap_floats) {
#if defined(PF_PHASE_TRACE) && defined(PF_SOLVER_TRACE)  // (the general solver's own counters: a second switch since r05 -- with them in the generic env kernel ROCm 7.2 aborts)
      const unsigned long long pf_a = __builtin_readcyclecounter();
#endif
      S.begin(ws + (incl - sz), R, com, inv_mass, v, w, i0, i1, i2, i3, i4, i5, src.slop(), src.inv_dt(), src.rest(), src.mu());
      int seen = 0;
      src.for_each(p, R, reach, [&](v3 off, float z) { if (S.n < n) S.add(off, -z); seen += 1; });
#if defined(PF_PHASE_TRACE) && defined(PF_SOLVER_TRACE)  // (the general solver's own counters: a second switch since r05 -- with them in the generic env kernel ROCm 7.2 aborts)
      const unsigned long long pf_b = __builtin_readcyclecounter();
      pf_fill += pf_b - pf_a;
#endif
      if (optimistic && seen > n && n < PF_MAX_CONTACTS) {  // more contacts than assumed: once more, with the count the fill just took
        n = seen > PF_MAX_CONTACTS ? PF_MAX_CONTACTS : seen;
        sz = (n + 1) * kContactWords;
      } else {
        S.sweeps(src.iters(), src.res_bound());
        out = S.finish(v, w);
        need = false;
      }
#if defined(PF_PHASE_TRACE) && defined(PF_SOLVER_TRACE)  // (the general solver's own counters: a second switch since r05 -- with them in the generic env kernel ROCm 7.2 aborts)
      pf_sweep += __builtin_readcyclecounter() - pf_b;
#endif
    }
  }
#if defined(PF_PHASE_TRACE) && defined(PF_SOLVER_TRACE)  // (the general solver's own counters: a second switch since r05 -- with them in the generic env kernel ROCm 7.2 aborts)
  {
    const unsigned long long pf_t2 = __builtin_readcyclecounter();
    const unsigned long long m = __ballot(1);
    const int first = __ffsll((long long)m) - 1;
    const int solved = __popcll(__ballot(S.n > 0));
    int nmax = S.n, sd = S.sweeps_done, rf = S.rows_full;
    unsigned long long sw = pf_sweep, fl = pf_fill;
    for (int o = 32; o > 0; o >>= 1) {
      nmax = max(nmax, __shfl_xor(nmax, o)); sd = max(sd, __shfl_xor(sd, o)); rf = max(rf, __shfl_xor(rf, o));
      const unsigned long long t = __shfl_xor(sw, o), u = __shfl_xor(fl, o);
      sw = sw > t ? sw : t; fl = fl > u ? fl : u;
    }
    if ((int)(threadIdx.x & 63u) == first) {
      trace_add(&g_solver_trace[0], 1ull);
      trace_add(&g_solver_trace[1], pf_tc - pf_t0);              // count pass
      trace_add(&g_solver_trace[7], (pf_t2 - pf_tc) - sw - fl);   // scan / loop bookkeeping
      trace_add(&g_solver_trace[6], fl);                          // records
      trace_add(&g_solver_trace[2], sw);
      trace_add(&g_solver_trace[3], (unsigned long long)nmax);
      trace_add(&g_solver_trace[4], (unsigned long long)solved);
      trace_add(&g_solver_trace[5], (unsigned long long)sd);
    }
    (void)rf;
  }
#endif
  return out;
}
// (PF_SOLVE_INLINE: A/B build switch -- the solver inlined at every call site instead of called)
#ifdef PF_SOLVE_INLINE
#define PF_SOLVE_ATTR __device__ __forceinline__
#else
#define PF_SOLVE_ATTR __device__ __noinline__ PF_RARE_TEXT
#endif
// (the capacity is wave-uniform among the lanes that ask: the first of them carries it)
constexpr int kPersistedBit = 1 << 29;
PF_DEV int need_cap_of(const bool need, const int cap_floats, const bool persisted) { return need ? (cap_floats | (persisted ? kPersistedBit : 0)) : -1; }
PF_DEV int __reduce_max_cap(int ask) {
  const unsigned long long m = __ballot(ask >= 0);
  return m != 0ull ? (__builtin_amdgcn_readlane(ask, __ffsll((long long)m) - 1) & ~kPersistedBit) : 0;
}
// Constant mass properties (QuadX, Fixedwing): read from the parameter block inside the call, so that the call passes few
// dwords -- all in registers.
// need_cap: the floats of LDS behind ws for a lane that asks, -1 for a lane that does not; bit kPersistedBit set for a body that
// held contact points after the previous tick (its contact reach is the breaking distance: pf_params.contact_break_distance).
// (Register allocation of this function is touchy: with the LDS address folded into the same dword to keep every argument in
//  registers, the allocator reached into 48 callee-saved VGPRs -- 96 scratch accesses per call; as it is, one argument travels
//  over the stack and five scratch accesses remain.)
PF_SOLVE_ATTR ContactOut contact_solve_dev(const pf_params* __restrict__ Pg, lds_fptr ws, int need_cap, v3 p, quat q, v3 v, v3 w) {
  const bool need = need_cap >= 0;
  const int cap_floats = __reduce_max_cap(need_cap);
  const pf_params_kptr P = uniform_params(Pg);
  // (every scalar the solve needs, requested up front in one batch: see ParamContactSrc)
  const ParamContactSrc src(P);
  const float im = P->inv_mass, i0 = P->I_inv[0], i1 = P->I_inv[1], i2 = P->I_inv[2], i3 = P->I_inv[3], i4 = P->I_inv[4], i5 = P->I_inv[5];
  const float c0 = P->com[0], c1 = P->com[1], c2 = P->com[2];
  const v3 com = P->has_com_offset ? v3{c0, c1, c2} : v3{0.f, 0.f, 0.f};
  return contact_solve_impl(src, ws, cap_floats, need, need && (need_cap & kPersistedBit) != 0, p, rot_from_quat(q), v, w, im, com, i0, i1, i2, i3, i4, i5);
}
// The same, inlined: what the generic kernels (Body::respond) use. Called out of line from the generic Fixedwing env kernel --
// 255 VGPRs + AGPR spill space, its 17th argument dword over the stack -- the ragged last wave of a launch lost its observation
// rows (ROCm 7.2 hipcc; with the call inlined: correct). The hot kernels, which the call suits, are covered lane by lane by the
// fixture replays on ragged batches.
PF_DEV ContactOut contact_solve_inl(const pf_params* __restrict__ Pg, lds_fptr ws, int need_cap, v3 p, quat q, v3 v, v3 w) {
  const bool need = need_cap >= 0;
  const int cap_floats = __reduce_max_cap(need_cap);
  const pf_params_kptr P = uniform_params(Pg);
  const ParamContactSrc src(P);
  const float im = P->inv_mass, i0 = P->I_inv[0], i1 = P->I_inv[1], i2 = P->I_inv[2], i3 = P->I_inv[3], i4 = P->I_inv[4], i5 = P->I_inv[5];
  const float c0 = P->com[0], c1 = P->com[1], c2 = P->com[2];
  const v3 com = P->has_com_offset ? v3{c0, c1, c2} : v3{0.f, 0.f, 0.f};
  return contact_solve_impl(src, ws, cap_floats, need, need && (need_cap & kPersistedBit) != 0, p, rot_from_quat(q), v, w, im, com, i0, i1, i2, i3, i4, i5);
}
// Mass properties that change per tick (Rocket): passed by value.
PF_DEV ContactOut contact_solve_var_dev(const pf_params* __restrict__ P, lds_fptr ws, int need_cap, v3 p, quat q, v3 v, v3 w, float inv_mass, v3 com,
                                               float i0, float i1, float i2, float i3, float i4, float i5) {
  const bool need = need_cap >= 0;
  const int cap_floats = __reduce_max_cap(need_cap);
  return contact_solve_impl(ParamContactSrc(uniform_params(P)), ws, cap_floats, need, need && (need_cap & kPersistedBit) != 0, p, rot_from_quat(q), v, w, inv_mass, com, i0, i1, i2, i3, i4, i5);
}

// Rigid body shared by both vehicles: the Bullet base state + what update_state derives from it.
struct Body {
  v3 p;
  quat q;
  v3 v, w;      // world frame, at the base origin
  m3 R;         // body -> world
  v3 wb, vb;    // quadx.py:522-523
  v3 rpy;       // quadx.py:526 (refreshed once per Aviary step)
  bool contact_now, contact_step;
  bool persisted = false;  // this tick: the body held contact points after the previous tick (they persist up to the breaking distance)
  // shared world (PF_TASK_MA_HOVER with agents_per_world > 1); both false for a drone that is alone in its world
  int ccap = kContactSlotFloats;  // floats of LDS behind `cws` for the solver's contact records (at least one worst-case region)
  PF_DEV void contact_regions(const pf_params&, int floats) { ccap = floats; }
  bool world_contact = false;  // a contact point anywhere in the world after the previous tick (quadx.py:509)
  bool peer_contact = false;   // this tick's drone-drone verdict for this body
  bool world_touch = false;    // some pair of this world is close enough for contact impulses between drones this tick
  bool woken = false;          // world_exchange: this body was a wreck at rest and a moving body has come within reach of it
  // shared world, the pair stage (shared_world.hpp: pair_stage_dev): the wave's pose / velocity exchange arrays, this lane, agents per world
  const float* wpose_ = nullptr;
  float* wvel_ = nullptr;
  int wtid = 0, wA = 1;

  PF_DEV void derive() {
    R = rot_from_quat(q);
    wb = mulT(R, w);
    vb = mulT(R, v);
  }
  // collision detection at the current pose against the ground box (aviary.py:240-242,523-525)
  // (can the body reach the ground slab at all? within one bounding radius of its top face, not below its bottom face, not
  //  beyond its rim: an aircraft that has flown off the 30 m slab and keeps falling is "under the floor" for seconds)
  PF_DEV bool slab_in_reach(const pf_params& P, float extra) const {
    const float r = P.bound_radius + extra;
    return (p.z - r <= 0.0f) && (p.z + r >= -2.0f * P.plane_half_z) && (__builtin_fabsf(p.x) - r <= P.plane_half_xy) && (__builtin_fabsf(p.y) - r <= P.plane_half_xy);
  }
  // rd: the pair is reported from this gap on -- the 15-axis verdict against the slab enlarged by it (pf_params.contact_report_distance,
  // or contact_break_distance for a body that held contact points after the previous tick)
  PF_DEV bool detect_contact(const pf_params& P, const float rd) const {
    if (!slab_in_reach(P, rd)) return false;
    const float hb[3] = {P.plane_half_xy + rd, P.plane_half_xy + rd, P.plane_half_z + rd};
    v3 cb{0.0f, 0.0f, -P.plane_half_z};
    bool hit = false;
#pragma unroll
    for (int k = 0; k < PF_MAX_BOXES; ++k) {
      if (k < P.n_boxes) {
        v3 c = p + mul(R, v3{P.boxes[k].c[0], P.boxes[k].c[1], P.boxes[k].c[2]});
        if (P.boxes[k].kind == 1) {
          hit |= cyl_overlaps_aabb(c, R, P.boxes[k].h[0], P.boxes[k].h[2], cb, hb);
        } else if (P.boxes[k].yaw != 0.0f) {  // a box on a yaw-rotated link: axes = R * Rz(yaw)
          float sy, cy;
          sincosf(P.boxes[k].yaw, &sy, &cy);
          m3 Rr{R.m00 * cy + R.m01 * sy, -R.m00 * sy + R.m01 * cy, R.m02, R.m10 * cy + R.m11 * sy, -R.m10 * sy + R.m11 * cy, R.m12,
                R.m20 * cy + R.m21 * sy, -R.m20 * sy + R.m21 * cy, R.m22};
          hit |= box_overlaps_aabb(c, Rr, P.boxes[k].h, cb, hb);
        } else {
          hit |= box_overlaps_aabb(c, R, P.boxes[k].h, cb, hb);
        }
      }
    }
    return hit;
  }
  // stepSimulation (aviary.py:516): collision detection at the pre-integration pose, then the
  // semi-implicit Euler free-body tick. F, tau: body frame; tau about the base origin.
  template <bool SHARED = false>
  PF_DEV void tick(const pf_params& P, v3 F, v3 tau) {
    persisted = contact_now;
    contact_now = detect_contact(P, persisted ? P.contact_break_distance : P.contact_report_distance) || peer_contact;
    v3 com{P.com[0], P.com[1], P.com[2]};
    if (P.has_com_offset) tau = tau - cross(com, F);
    v3 h = symmul(P.I_pa, wb);
    if (P.use_gyro_term) h = h + symmul(P.I_own, wb);
    v3 wdot_b = symmul(P.I_inv, tau - cross(wb, h));
    v3 wdot = mul(R, wdot_b);
    v3 a = P.inv_mass * mul(R, F);
    a.z += P.gravity_z;
    if (P.has_com_offset) {
      v3 cw = mul(R, com);
      a = a - cross(wdot, cw) - cross(w, cross(w, cw));
    }
    const float dt = P.dt, vm = P.max_coord_vel;
    w = v3{clampf(fmaf(wdot.x, dt, w.x), -vm, vm), clampf(fmaf(wdot.y, dt, w.y), -vm, vm), clampf(fmaf(wdot.z, dt, w.z), -vm, vm)};
    v = v3{clampf(fmaf(a.x, dt, v.x), -vm, vm), clampf(fmaf(a.y, dt, v.y), -vm, vm), clampf(fmaf(a.z, dt, v.z), -vm, vm)};
    v3 shift{0.0f, 0.0f, 0.0f};
    if (SHARED) shift = pair_stage(pdev);  // contact response between the drones of the world, before the ground's
    const float lift = respond(pdev);
    p = v3{fmaf(dt, v.x, p.x) + shift.x, fmaf(dt, v.y, p.y) + shift.y, fmaf(dt, v.z, p.z) + lift + shift.z};
    q = quat_integrate(q, w, 0.5f * dt);
    derive();
    contact_step |= contact_now;
  }
  // constraint solve of stepSimulation: contacts found at the pre-integration pose act on the new velocities; returns the
  // position-level penetration recovery (contact_erp x deepest penetration) to add to z after the position update
  const pf_params* pdev;  // device copy of the parameter block (the out-of-line contact solver reads the colliders from it)
  lds_fptr cws;           // the wave's LDS regions for the contact solver
  // Can any contact constraint act this tick? Every vertex lies within bound_radius of the base origin, so its height is
  // >= low = p.z - bound_radius and its normal velocity >= v.z - |w| bound_radius: if even that worst case ends the tick
  // above the allowed overlap (and nothing is deeper than it now), every constraint of the solve is slack -- all impulses
  // exactly zero, no recovery -- and the call is skipped without changing the result.
  PF_DEV bool contact_may_act(const pf_params* Pd) const {
    const float low = p.z - Pd->bound_radius;
    const float reach = persisted ? Pd->contact_break_distance : Pd->contact_margin;
    if (!slab_in_reach(*Pd, reach)) return false;
    if (low > reach) return false;  // no vertex can be within the contact reach
    const float vlow = v.z - __builtin_sqrtf(dot(w, w)) * Pd->bound_radius;
    return (low + Pd->contact_slop + Pd->dt * vlow < 0.0f) || (low < -Pd->contact_slop);
  }
  // The pair stage (shared worlds): publish this body's new velocity, let the world's first lane resolve the drone-drone
  // contacts (shared_world.hpp: pair_stage_dev), take back the velocity and the position-level shift. Wave-uniform call.
  PF_DEV v3 pair_stage(const pf_params* Pd);
  PF_DEV float respond(const pf_params* Pd) {
    if (Pd == nullptr) return 0.0f;  // (wave-uniform: kernels without ticks)
    float lift = 0.0f;
    const bool need = Pd->contact_response && contact_may_act(Pd);
    if (__any(need)) {
      const ContactOut o = contact_solve_inl(Pd, cws, need_cap_of(need, ccap, persisted), p, q, v, w);
      v = o.v; w = o.w;  // (unchanged for a lane that did not ask or has no contact vertex)
      lift = Pd->contact_erp * o.deepest;  // (deepest: already net of the slop)
    }
    return lift;
  }
  PF_DEV float respond_var(const pf_params* Pd, float inv_mass, v3 com, const float Iinv[6]) {
    if (Pd == nullptr) return 0.0f;
    float lift = 0.0f;
    const bool need = Pd->contact_response && contact_may_act(Pd);
    if (__any(need)) {
      const ContactOut o = contact_solve_var_dev(Pd, cws, need_cap_of(need, ccap, persisted), p, q, v, w, inv_mass, com, Iinv[0], Iinv[1], Iinv[2], Iinv[3], Iinv[4], Iinv[5]);
      v = o.v; w = o.w;
      lift = Pd->contact_erp * o.deepest;
    }
    return lift;
  }
  // The same tick for a body whose mass properties change over time (Rocket): inverse mass, centre of
  // mass, gyroscopic inertia H and inverse inertia (symmetric xx xy xz yy yz zz) are arguments.
  PF_DEV void tick_var(const pf_params& P, v3 F, v3 tau, float inv_mass, v3 com, const float H[6], const float Iinv[6]) {
    persisted = contact_now;
    contact_now = detect_contact(P, persisted ? P.contact_break_distance : P.contact_report_distance);
    tau = tau - cross(com, F);
    v3 h = symmul(H, wb);
    v3 wdot_b = symmul(Iinv, tau - cross(wb, h));
    v3 wdot = mul(R, wdot_b);
    v3 a = inv_mass * mul(R, F);
    a.z += P.gravity_z;
    v3 cw = mul(R, com);
    a = a - cross(wdot, cw) - cross(w, cross(w, cw));
    const float dt = P.dt, vm = P.max_coord_vel;
    w = v3{clampf(fmaf(wdot.x, dt, w.x), -vm, vm), clampf(fmaf(wdot.y, dt, w.y), -vm, vm), clampf(fmaf(wdot.z, dt, w.z), -vm, vm)};
    v = v3{clampf(fmaf(a.x, dt, v.x), -vm, vm), clampf(fmaf(a.y, dt, v.y), -vm, vm), clampf(fmaf(a.z, dt, v.z), -vm, vm)};
    const float lift = respond_var(pdev, inv_mass, com, Iinv);
    p = v3{fmaf(dt, v.x, p.x), fmaf(dt, v.y, p.y), fmaf(dt, v.z, p.z) + lift};
    q = quat_integrate(q, w, 0.5f * dt);
    derive();
    contact_step |= contact_now;
  }
  PF_DEV bool nonfinite() const {  // any NaN / Inf among the base state words poisons the sum
    const float chk = ((p.x + p.y) + (p.z + q.x)) + ((q.y + q.z) + (q.w + v.x)) + ((v.y + v.z) + (w.x + w.y)) + w.z;
    return !(__builtin_fabsf(chk) < INFINITY);
  }
  PF_DEV void spawn(const pf_params& P, const float* pose /* [7] or null */, const float* vel = nullptr /* [3] or null */) {
    if (pose) {
      p = v3{pose[0], pose[1], pose[2]};
      q = quat{pose[3], pose[4], pose[5], pose[6]};
    } else {
      p = v3{P.start_pos[0], P.start_pos[1], P.start_pos[2]};
      q = quat{P.start_quat[0], P.start_quat[1], P.start_quat[2], P.start_quat[3]};
    }
    v = vel ? v3{vel[0], vel[1], vel[2]} : v3{P.start_vel[0], P.start_vel[1], P.start_vel[2]};
    w = v3{0.0f, 0.0f, 0.0f};
    contact_now = false;
    contact_step = false;
    persisted = false;
    derive();
    rpy = euler_from_quat_fast(q);
  }
};

// ------------------------------------------------------------------------------------------
// QuadX: drones/quadx.py. MODE_T == kRuntimeMode selects the flight mode from pf_params at run
// time; any other value folds the cascade at compile time (mode 0 is the env hot path).
constexpr int kRuntimeMode = 100;
constexpr int kNoModeOverride = -100;  // control(): no per-drone flight mode given

struct QuadX {
  static constexpr int GROUPS = 16, G_INT = 6, G_TGT = 12, AUX = 4, SP = 4;  // g15: MA-hover past action
  static constexpr int TABLE_FLOATS = 4;  // no LDS constant table
  static PF_DEV void fill_table(float*, const pf_params*, int) {}
  PF_DEV void bind(const float*) {}
  Body b;
  float thr[4];
  float pwm[4];
  float I0[3], E0[3];                    // ang_vel PID
  float I1[3], E1[3];                    // ang_pos PID
  float I2[2], E2[2], I3[2], E3[2];      // lin_vel, lin_pos PIDs
  float zI[2], zE[2];                    // z_vel, z_pos PIDs

  static PF_DEV bool needs_cascade(int mode) { return mode > 0; }

  PF_DEV void load(const float4* S, size_t n, size_t i, int mode, float& new_dist, int4& ints) {
    float4 g0 = S[0 * n + i], g1 = S[1 * n + i], g2 = S[2 * n + i], g3 = S[3 * n + i], g4 = S[4 * n + i],
           g5 = S[5 * n + i];
    float4 gi = S[6 * n + i];
    b.p = v3{g0.x, g0.y, g0.z}; new_dist = g0.w;
    b.q = quat{g1.x, g1.y, g1.z, g1.w};
    b.v = v3{g2.x, g2.y, g2.z};
    b.w = v3{g2.w, g3.x, g3.y};
    thr[0] = g3.z; thr[1] = g3.w; thr[2] = g4.x; thr[3] = g4.y;
    I0[0] = g4.z; I0[1] = g4.w; I0[2] = g5.x;
    E0[0] = g5.y; E0[1] = g5.z; E0[2] = g5.w;
    ints = int4{__float_as_int(gi.x), __float_as_int(gi.y), __float_as_int(gi.z), __float_as_int(gi.w)};
    if (needs_cascade(mode)) {
      float4 g7 = S[7 * n + i], g8 = S[8 * n + i], g9 = S[9 * n + i], g10 = S[10 * n + i], g11 = S[11 * n + i];
      I1[0] = g7.x; I1[1] = g7.y; I1[2] = g7.z; E1[0] = g7.w;
      E1[1] = g8.x; E1[2] = g8.y; I2[0] = g8.z; I2[1] = g8.w;
      E2[0] = g9.x; E2[1] = g9.y; I3[0] = g9.z; I3[1] = g9.w;
      E3[0] = g10.x; E3[1] = g10.y; zI[0] = g10.z; zI[1] = g10.w;
      zE[0] = g11.x; zE[1] = g11.y;
    }
    relaunch(mode, ints.y);
  }
  // What a launch starts from besides the stored words (load's tail). env_kernel's roll_steps calls it between the env steps of a
  // launch that keeps the lane's state in registers, so that step k + 1 starts from what a relaunch would have re-derived.
  PF_DEV void relaunch(int mode, int flags) {
    if (!needs_cascade(mode)) {  // (groups 7-11 are not stored for the direct modes)
      zero_cascade();
      zI[0] = zI[1] = zE[0] = zE[1] = 0.0f;
    }
    b.contact_now = (flags & PF_F_CONTACT) != 0;
    b.contact_step = false;
    b.derive();
    if (needs_cascade(mode)) b.rpy = euler_from_quat_fast(b.q);
    else b.rpy = v3{0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int k = 0; k < 4; ++k) pwm[k] = 0.0f;
  }
  PF_DEV void store(float4* S, size_t n, size_t i, int mode, float new_dist, int4 ints) const {
    S[0 * n + i] = float4{b.p.x, b.p.y, b.p.z, new_dist};
    S[1 * n + i] = float4{b.q.x, b.q.y, b.q.z, b.q.w};
    S[2 * n + i] = float4{b.v.x, b.v.y, b.v.z, b.w.x};
    S[3 * n + i] = float4{b.w.y, b.w.z, thr[0], thr[1]};
    S[4 * n + i] = float4{thr[2], thr[3], I0[0], I0[1]};
    S[5 * n + i] = float4{I0[2], E0[0], E0[1], E0[2]};
    S[6 * n + i] = float4{__int_as_float(ints.x), __int_as_float(ints.y), __int_as_float(ints.z), __int_as_float(ints.w)};
    if (needs_cascade(mode)) {
      S[7 * n + i] = float4{I1[0], I1[1], I1[2], E1[0]};
      S[8 * n + i] = float4{E1[1], E1[2], I2[0], I2[1]};
      S[9 * n + i] = float4{E2[0], E2[1], I3[0], I3[1]};
      S[10 * n + i] = float4{E3[0], E3[1], zI[0], zI[1]};
      S[11 * n + i] = float4{zE[0], zE[1], 0.0f, 0.0f};
    }
  }
  PF_DEV void zero_cascade() {
#pragma unroll
    for (int k = 0; k < 3; ++k) I1[k] = E1[k] = 0.0f;
#pragma unroll
    for (int k = 0; k < 2; ++k) I2[k] = E2[k] = I3[k] = E3[k] = 0.0f;
  }
  // set_mode (quadx.py:233-373): fresh PID objects (NOT the z PIDs, quadx.py:206) + default setpoint
  PF_DEV void set_mode(int mode, float sp[6]) {
    if (mode == -1) return;
#pragma unroll
    for (int k = 0; k < 3; ++k) I0[k] = E0[k] = 0.0f;
    zero_cascade();
    sp[0] = sp[1] = sp[2] = sp[3] = 0.0f;
    if (mode == 0) sp[3] = -1.0f;
    else if (mode == 1 || mode == 5 || mode == 6) {}
    else if (mode == 7) { sp[0] = b.p.x; sp[1] = b.p.y; sp[2] = b.rpy.z; sp[3] = b.p.z; }
    else sp[3] = b.p.z;
  }
  // drone.reset() + update_state (quadx.py:222-231, aviary.py:310-311); setpoint -> zeros(4)
  PF_DEV void reset(const pf_params& P, const float* pose, float sp[6], const float* vel = nullptr) {
    b.spawn(P, pose, vel);
#pragma unroll
    for (int k = 0; k < 4; ++k) thr[k] = pwm[k] = 0.0f;
    set_mode(0, sp);
    sp[0] = sp[1] = sp[2] = sp[3] = 0.0f;
    zI[0] = zI[1] = zE[0] = zE[1] = 0.0f;
  }
  PF_DEV void pid3(const pf_pid& g, float T, float invT, float* I, float* E, v3 st, float a[3], int n) {
    const float s[3] = {st.x, st.y, st.z};
#pragma unroll
    for (int k = 0; k < 3; ++k)
      if (k < n) a[k] = pid1(g.kp[k], g.ki[k], g.kd[k], g.lim[k], T, invT, I[k], E[k], s[k], a[k]);
  }
  // update_control (quadx.py:401-493). period_over > 0: this drone's own control period (an Aviary
  // whose drones run different control_hz, tests/test_core.py:34-62); otherwise the batch-wide one.
  template <int MODE_T>
  PF_DEV void control(const pf_params& P, const float sp[6], float period_over = 0.0f, int mode_over = kNoModeOverride) {
    const int mode = mode_over != kNoModeOverride ? mode_over : ((MODE_T == kRuntimeMode) ? P.flight_mode : MODE_T);
    const float cT = period_over > 0.0f ? period_over : P.control_period;
    const float cIT = period_over > 0.0f ? 1.0f / period_over : P.inv_control_period;
    float a[3] = {sp[0], sp[1], sp[2]};
    float z = sp[3];
    if (mode == -1) {
      pwm[0] = a[0]; pwm[1] = a[1]; pwm[2] = a[2]; pwm[3] = z;
      return;
    }
    if (mode != 0) {  // the cascaded modes: state derivation and PIDs in fp64 (quadx_control_d.hpp: why)
      const QuadCtlIn in = quad_ctl_inputs(b.q, b.v, b.w, b.p);
      const float sp4[4] = {sp[0], sp[1], sp[2], sp[3]};
      quad_cascade_d(&P, mode, (double)cT, in, QuadMemD{I0, E0, I1, E1, I2, E2, I3, E3, zI, zE}, sp4, pwm);
      return;
    }
    if (mode == 0 || mode == 2) {
      pid3(P.pid[0], cT, cIT, I0, E0, b.wb, a, 3);
    } else if (mode == 1 || mode == 3) {
      pid3(P.pid[1], cT, cIT, I1, E1, b.rpy, a, 3);
      pid3(P.pid[0], cT, cIT, I0, E0, b.wb, a, 3);
    } else {
      if (mode == 7) pid3(P.pid[3], cT, cIT, I3, E3, b.p, a, 2);
      if (mode == 6 || mode == 7) {  // quadx.py:448-451,460-463
        float s, c;
        sincosf(b.rpy.z, &s, &c);
        float a0 = c * a[0] + s * a[1], a1 = -s * a[0] + c * a[1];
        a[0] = a0; a[1] = a1;
      }
      pid3(P.pid[2], cT, cIT, I2, E2, b.vb, a, 2);
      { float t0 = -a[1], t1 = a[0]; a[0] = t0; a[1] = t1; }
      pid3(P.pid[1], cT, cIT, I1, E1, b.rpy, a, mode == 7 ? 3 : 2);
      pid3(P.pid[0], cT, cIT, I0, E0, b.wb, a, 3);
    }
    if (mode == 0) {
      z = clampf(z, 0.0f, 1.0f);
    } else {
      if (!(mode == 1 || mode == 5 || mode == 6))
        z = pid1(P.zpid[1].kp[0], P.zpid[1].ki[0], P.zpid[1].kd[0], P.zpid[1].lim[0], cT, cIT, zI[1], zE[1], b.p.z, z);
      z = pid1(P.zpid[0].kp[0], P.zpid[0].ki[0], P.zpid[0].kd[0], P.zpid[0].lim[0], cT, cIT, zI[0], zE[0], b.vb.z, z);
      z = clampf(z, 0.0f, 1.0f);
    }
    // mixing + saturation handling (quadx.py:482-493)
    const float cmd[4] = {a[0], a[1], a[2], z};
    float hi = -INFINITY, lo = INFINITY;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float s = P.motor_map[i][0] * cmd[0] + P.motor_map[i][1] * cmd[1] + P.motor_map[i][2] * cmd[2] + P.motor_map[i][3] * cmd[3];
      pwm[i] = s;
      hi = __builtin_fmaxf(hi, s);
      lo = __builtin_fminf(lo, s);
    }
    if (hi != lo) {
      float pmax = __builtin_fminf(hi, 1.0f), pmin = __builtin_fmaxf(lo, 0.05f);
      float ka = (pmin - lo) / (pmax - lo), ks = (hi - pmax) / (hi - pmin);
#pragma unroll
      for (int i = 0; i < 4; ++i) pwm[i] += ka * (pmax - pwm[i]) - ks * (pwm[i] - pmin);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) pwm[i] = clampf(pwm[i], 0.05f, 1.0f);
  }
  // update_physics + stepSimulation + update_state for one tick (quadx.py:495-535).
  // wind: world-frame wind at the body link as sampled by the previous update_state
  // (boring_bodies.py:93-96), or null.
  static constexpr int WIND_LINKS = 1;
  PF_DEV v3 link_pos(const pf_params&, int) const { return b.p; }  // centre-of-mass link at the base origin
  template <bool SHARED = false>
  PF_DEV void tick(const pf_params& P, float xi, const float* wind = nullptr) {
    v3 vd = b.vb;
    if (wind) vd = vd - mulT(b.R, v3{wind[0], wind[1], wind[2]});
    v3 F{-P.drag_const[0] * sq_signed(vd.x), -P.drag_const[1] * sq_signed(vd.y), -P.drag_const[2] * sq_signed(vd.z)};
    v3 tau{0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int i = 0; i < 4; ++i) {  // motors.py:131-138,182-193
      float t = fmaf(P.motor_dt_over_tau[i], pwm[i] - thr[i], thr[i]);
      t = fmaf(xi * t, P.motor_noise[i], t);
      thr[i] = t;
      float k = sq_signed(t);
      float f = k * P.motor_fmax[i];
      F.z += f;
      tau.x = fmaf(P.motor_r[i][1], f, tau.x);
      tau.y = fmaf(-P.motor_r[i][0], f, tau.y);
      tau.z = fmaf(k, P.motor_tmax[i], tau.z);
    }
    if (!(b.contact_now || b.world_contact)) {  // quadx.py:502-510: no contact point anywhere in the world
      tau.x = fmaf(-P.drag_coef_pqr, sq_signed(b.wb.x), tau.x);
      tau.y = fmaf(-P.drag_coef_pqr, sq_signed(b.wb.y), tau.y);
      tau.z = fmaf(-P.drag_coef_pqr, sq_signed(b.wb.z), tau.z);
    }
    b.template tick<SHARED>(P, F, tau);
  }
  PF_DEV void tick_unarmed(const pf_params& P) { b.tick(P, v3{0.f, 0.f, 0.f}, v3{0.f, 0.f, 0.f}); }  // aviary.py:510-521
  // one Aviary.step (aviary.py:480-531): control on the first tick, pwm held afterwards
  template <int MODE_T>
  PF_DEV void aviary_step(const pf_params& P, const float sp[6], Noise& nz, int flat_base) {
    b.contact_step = false;
    control<MODE_T>(P, sp);
    for (int t = 0; t < P.ticks_per_control; ++t) tick(P, nz.get(flat_base + t));
    b.rpy = euler_from_quat_fast(b.q);
  }
  PF_DEV void aux(float* o) const {
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] = thr[k];
  }
  PF_DEV bool nonfinite() const { return b.nonfinite() || !(__builtin_fabsf((thr[0] + thr[1]) + (thr[2] + thr[3])) < INFINITY); }
  // the motor commands held between the ticks of one Aviary step (pf_aviary_tick)
  PF_DEV float4 get_cmd() const { return float4{pwm[0], pwm[1], pwm[2], pwm[3]}; }
  PF_DEV void set_cmd(float4 c) { pwm[0] = c.x; pwm[1] = c.y; pwm[2] = c.z; pwm[3] = c.w; }
};

// lifting_surfaces.py:266-498 for one surface; returns force & torque in the (axis-aligned) link
// frame. libm-free: (cos a, sin a) come from the velocity components, the effective angle of attack
// a_eff = a - (a_0 + a_i) by the angle-difference identities with a small-angle polynomial for the
// second operand (|a_0 + a_i| < 0.8 rad for any surface the model admits), a itself (needed for the
// regime tests and CM) from the polynomial atan2.
PF_DEV void lifting_surface(const pf_surface& S, v3 vloc, float a, v3& F, v3& T) {
  v3 lift{S.lift[0], S.lift[1], S.lift[2]}, drag{S.drag[0], S.drag[1], S.drag[2]};
  float V2 = dot(vloc, vloc);
  float la = dot(vloc, lift), fa = dot(vloc, drag);
  float h2 = fmaf(la, la, fa * fa);
  float ih = frsq(h2);
  const bool still = !(h2 > 0.0f);
  float ca = still ? 1.0f : fa * ih, sa = still ? 0.0f : -la * ih;
  float alpha = fast_atan2(-la, fa);  // :342-345
  // :386-394
  float defl = a * S.deflection_limit_rad;
  float dCl = S.Cl_alpha_3D * S.aero_tau_eta * defl;
  float dClmax = S.flap_to_chord * dCl;
  float ClmaxP = fmaf(S.Cl_alpha_3D, S.alpha_stall_P_base - S.alpha_0_base, dClmax);
  float ClmaxN = fmaf(S.Cl_alpha_3D, S.alpha_stall_N_base - S.alpha_0_base, dClmax);
  float a0 = S.alpha_0_base - dCl * S.inv_Cl_alpha_3D;
  float aP = fmaf(ClmaxP, S.inv_Cl_alpha_3D, a0), aN = fmaf(ClmaxN, S.inv_Cl_alpha_3D, a0);
  const bool linear = (aN < alpha) && (alpha < aP);
  // induced angle: linear regime :397-399, post-stall two-point np.interp :409-425
  float Cl_lin = S.Cl_alpha_3D * (alpha - a0);
  float ai;
  if (linear) {
    ai = Cl_lin * S.inv_pi_aspect;
  } else if (alpha > 0.0f) {
    float ai_stall = S.Cl_alpha_3D * (aP - a0) * S.inv_pi_aspect;
    float x0 = aP, x1 = 0.5f * kPi;
    ai = (alpha <= x0) ? ai_stall : (alpha >= x1 ? 0.0f : ai_stall - ai_stall * frcp(x1 - x0) * (alpha - x0));
  } else {
    float ai_stall = S.Cl_alpha_3D * (aN - a0) * S.inv_pi_aspect;
    float x0 = -0.5f * kPi, x1 = aN;
    ai = (alpha <= x0) ? 0.0f : (alpha >= x1 ? ai_stall : ai_stall * frcp(x1 - x0) * (alpha - x0));
  }
  const float x = a0 + ai;
  const float ae = alpha - x;
  float sx, cx;
  sincos_small(x, sx, cx);
  const float se = sa * cx - ca * sx, ce = fmaf(ca, cx, sa * sx);
  float Cl, Cd, CM;
  if (linear) {  // :397-406
    Cl = Cl_lin;
    float CT = S.Cd_0 * ce;
    float CN = (Cl + CT * se) * frcp(ce);
    Cd = CN * se + CT * ce;
    CM = -CN * (0.25f - 0.175f * (1.0f - 2.0f * ae * (1.0f / kPi)));
  } else {  // :427-448
    float Cd90 = fmaf(-4.26e-2f, defl * defl, fmaf(2.1e-1f, defl, 1.98f));
    float CN = Cd90 * se * (frcp(0.56f + 0.44f * __builtin_fabsf(se)) - S.exp_term);
    float CT = 0.5f * S.Cd_0 * ce;
    Cl = CN * ce - CT * se;
    Cd = CN * se + CT * ce;
    CM = -CN * (0.25f - 0.175f * (1.0f - 2.0f * __builtin_fabsf(ae) * (1.0f / kPi)));
  }
  // :485-498
  float QA = S.half_rho_area * V2;
  float L = Cl * QA, D = Cd * QA;
  float fn = L * ca + D * sa, fp = L * sa - D * ca;
  F = v3{lift.x * fn + drag.x * fp, lift.y * fn + drag.y * fp, lift.z * fn + drag.z * fp};
  float tm = QA * CM * S.chord;
  T = v3{tm * S.torque[0], tm * S.torque[1], tm * S.torque[2]};
}

// ------------------------------------------------------------------------------------------
// Fixedwing: drones/fixedwing.py + abstractions/lifting_surfaces.py
struct Fixedwing {
  static constexpr int GROUPS = 9, G_INT = 5, G_TGT = 6, AUX = 6, SP = 6;
  // The five lifting surfaces carry 26 constants each -- far more than fit in SGPRs next to the
  // rest of the kernel (the first version spilled 1.9 KB/lane to scratch). They live in an LDS table
  // (32-float rows, one per surface) filled once per workgroup; tick() walks the surfaces in a real
  // loop and fetches one row (7 broadcast ds_read_b128) per iteration.
  static constexpr int TABLE_STRIDE = 32, TABLE_FLOATS = PF_MAX_SURF * TABLE_STRIDE;
  static PF_DEV void fill_table(float* tab, const pf_params* Pdev, int tid) {
    constexpr int kF = (int)(sizeof(pf_surface) / sizeof(float));
    static_assert(kF == 26, "pf_surface layout changed: update Fixedwing::row()");
    const float* src = reinterpret_cast<const float*>(Pdev->surf);
    for (int i = tid; i < PF_MAX_SURF * kF; i += 64) tab[(i / kF) * TABLE_STRIDE + (i % kF)] = src[i];
  }
  const float* sk;
  PF_DEV void bind(const float* tab) { sk = tab; }
  static PF_DEV pf_surface row(const float* r) {
    const float4* c = reinterpret_cast<const float4*>(r);
    const float4 c0 = c[0], c1 = c[1], c2 = c[2], c3 = c[3], c4 = c[4], c5 = c[5], c6 = c[6];
    pf_surface S;
    S.r[0] = c0.x; S.r[1] = c0.y; S.r[2] = c0.z; S.lift[0] = c0.w;
    S.lift[1] = c1.x; S.lift[2] = c1.y; S.drag[0] = c1.z; S.drag[1] = c1.w;
    S.drag[2] = c2.x; S.torque[0] = c2.y; S.torque[1] = c2.z; S.torque[2] = c2.w;
    S.Cl_alpha_3D = c3.x; S.inv_Cl_alpha_3D = c3.y; S.aero_tau_eta = c3.z; S.flap_to_chord = c3.w;
    S.inv_pi_aspect = c4.x; S.exp_term = c4.y; S.alpha_0_base = c4.z; S.alpha_stall_P_base = c4.w;
    S.alpha_stall_N_base = c5.x; S.Cd_0 = c5.y; S.deflection_limit_rad = c5.z; S.dt_over_tau = c5.w;
    S.half_rho_area = c6.x; S.chord = c6.y;
    return S;
  }
  Body b;
  float act[5];
  float thr;

  PF_DEV void load(const float4* S, size_t n, size_t i, int mode, float& new_dist, int4& ints) {
    (void)mode;
    float4 g0 = S[0 * n + i], g1 = S[1 * n + i], g2 = S[2 * n + i], g3 = S[3 * n + i], g4 = S[4 * n + i];
    float4 gi = S[5 * n + i];
    b.p = v3{g0.x, g0.y, g0.z}; new_dist = g0.w;
    b.q = quat{g1.x, g1.y, g1.z, g1.w};
    b.v = v3{g2.x, g2.y, g2.z};
    b.w = v3{g2.w, g3.x, g3.y};
    act[0] = g3.z; act[1] = g3.w; act[2] = g4.x; act[3] = g4.y; act[4] = g4.z; thr = g4.w;
    ints = int4{__float_as_int(gi.x), __float_as_int(gi.y), __float_as_int(gi.z), __float_as_int(gi.w)};
    relaunch(mode, ints.y);
  }
  PF_DEV void relaunch(int, int flags) {  // (see QuadX::relaunch)
    b.contact_now = (flags & PF_F_CONTACT) != 0;
    b.contact_step = false;
    b.derive();
    b.rpy = v3{0.0f, 0.0f, 0.0f};
  }
  PF_DEV void store(float4* S, size_t n, size_t i, int mode, float new_dist, int4 ints) const {
    (void)mode;
    S[0 * n + i] = float4{b.p.x, b.p.y, b.p.z, new_dist};
    S[1 * n + i] = float4{b.q.x, b.q.y, b.q.z, b.q.w};
    S[2 * n + i] = float4{b.v.x, b.v.y, b.v.z, b.w.x};
    S[3 * n + i] = float4{b.w.y, b.w.z, act[0], act[1]};
    S[4 * n + i] = float4{act[2], act[3], act[4], thr};
    S[5 * n + i] = float4{__int_as_float(ints.x), __int_as_float(ints.y), __int_as_float(ints.z), __int_as_float(ints.w)};
  }
  PF_DEV void set_mode(int mode, float sp[6]) {  // fixedwing.py:206-227
    (void)mode;
#pragma unroll
    for (int k = 0; k < 6; ++k) sp[k] = 0.0f;
  }
  PF_DEV void reset(const pf_params& P, const float* pose, float sp[6], const float* vel = nullptr) {  // fixedwing.py:194-204
    b.spawn(P, pose, vel);
#pragma unroll
    for (int k = 0; k < 5; ++k) act[k] = 0.0f;
    thr = 0.0f;
    set_mode(0, sp);
  }
  float cmd[6];
  template <int MODE_T>
  PF_DEV void control(const pf_params& P, const float sp[6], float = 0.0f, int = 0) {  // fixedwing.py:229-259 (stateless: no period)
    if (P.flight_mode == -1) {
#pragma unroll
      for (int k = 0; k < 6; ++k) cmd[k] = sp[k];
    } else {
      const float s4[4] = {sp[0], sp[1], sp[2], sp[3]};
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        int id = P.assist_ids[k];
        float val = id == 0 ? s4[0] : (id == 1 ? s4[1] : (id == 2 ? s4[2] : s4[3]));
        cmd[k] = val * P.assist_signs[k];
      }
    }
  }
  static constexpr int WIND_LINKS = PF_MAX_SURF;
  PF_DEV v3 link_pos(const pf_params& P, int k) const {  // surface link COMs (lifting_surfaces.py:83-93)
    return b.p + mul(b.R, v3{P.surf[k].r[0], P.surf[k].r[1], P.surf[k].r[2]});
  }
  template <bool SHARED = false>
  PF_DEV void tick(const pf_params& P, float xi, const float* wind = nullptr) {
    v3 F{0.0f, 0.0f, 0.0f}, tau{0.0f, 0.0f, 0.0f};
#pragma unroll 1
    for (int i = 0; i < PF_MAX_SURF; ++i) {
      const pf_surface S = row(sk + i * TABLE_STRIDE);
      // act[i] / cmd[i] by selects: a dynamically indexed register array would go to scratch
      float ai = i == 0 ? act[0] : (i == 1 ? act[1] : (i == 2 ? act[2] : (i == 3 ? act[3] : act[4])));
      const float ci = i == 0 ? cmd[0] : (i == 1 ? cmd[1] : (i == 2 ? cmd[2] : (i == 3 ? cmd[3] : cmd[4])));
      ai = fmaf(S.dt_over_tau, ci - ai, ai);  // lifting_surfaces.py:277
      act[0] = i == 0 ? ai : act[0]; act[1] = i == 1 ? ai : act[1]; act[2] = i == 2 ? ai : act[2];
      act[3] = i == 3 ? ai : act[3]; act[4] = i == 4 ? ai : act[4];
      v3 r{S.r[0], S.r[1], S.r[2]};
      v3 vloc = b.vb + cross(b.wb, r);  // lifting_surfaces.py:73-110
      if (wind) vloc = vloc - mulT(b.R, v3{wind[3 * i + 0], wind[3 * i + 1], wind[3 * i + 2]});
      v3 f, t;
      lifting_surface(S, vloc, ai, f, t);
      F = F + f;
      tau = tau + cross(r, f) + t;
    }
    {  // motor (fixedwing.py:147-168,264), thrust along +x at the base origin
      float t = fmaf(P.motor_dt_over_tau[0], cmd[5] - thr, thr);
      t = fmaf(xi * t, P.motor_noise[0], t);
      thr = t;
      float k = sq_signed(t);
      v3 u{P.thrust_unit[0][0], P.thrust_unit[0][1], P.thrust_unit[0][2]};
      v3 f = (k * P.motor_fmax[0]) * u;
      v3 r{P.motor_r[0][0], P.motor_r[0][1], P.motor_r[0][2]};
      F = F + f;
      tau = tau + cross(r, f) + (k * P.motor_tmax[0]) * u;
    }
    b.template tick<SHARED>(P, F, tau);
  }
  PF_DEV void tick_unarmed(const pf_params& P) { b.tick(P, v3{0.f, 0.f, 0.f}, v3{0.f, 0.f, 0.f}); }
  template <int MODE_T>
  PF_DEV void aviary_step(const pf_params& P, const float sp[6], Noise& nz, int flat_base) {
    b.contact_step = false;
    control<MODE_T>(P, sp);
    for (int t = 0; t < P.ticks_per_control; ++t) tick(P, nz.get(flat_base + t));
    b.rpy = euler_from_quat_fast(b.q);
  }
  PF_DEV void aux(float* o) const {
#pragma unroll
    for (int k = 0; k < 5; ++k) o[k] = act[k];
    o[5] = thr;
  }
  PF_DEV bool nonfinite() const { return b.nonfinite() || !(__builtin_fabsf(((act[0] + act[1]) + (act[2] + act[3])) + (act[4] + thr)) < INFINITY); }
  PF_DEV float4 get_cmd() const { return float4{0.f, 0.f, 0.f, 0.f}; }  // stateless mixing: nothing to carry
  PF_DEV void set_cmd(float4) {}
};

}  // namespace pf
