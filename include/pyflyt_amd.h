/*
 * pyflyt_amd.h -- C ABI of the MI355X-native batched UAV-physics step.
 *
 * Drop-in boundary for ONE hot path of jjshoots/PyFlyt (reference @ v0.30.0): `Aviary.step()` and
 * the per-drone `update_control / update_physics / update_state` loop plus the 6-DoF integrator
 * PyBullet supplies underneath it, for N independent drones at one wavefront lane per drone.
 * The reference is pure Python and has no FFI of its own; each entry point below names the
 * reference interface it replaces (file:line relative to /root/reference/PyFlyt/), and
 * INTEGRATION.md shows the ctypes binding a maintainer would add.
 *
 * Conventions
 *   - plain C, no C++/torch types; every pointer in pf_buffers is a DEVICE pointer into memory the
 *     caller owns (PyTorch-ROCm tensors: tensor.data_ptr()); the library only borrows them for the
 *     duration of a call and owns nothing but its context.
 *   - every call is asynchronous on the caller's HIP stream (`stream` = hipStream_t, e.g.
 *     torch.cuda.current_stream().cuda_stream); no internal threads, no hidden synchronisation.
 *   - every function returns 0 on success, a negative pf_status or a positive hipError_t
 *     otherwise; pf_last_error() gives the message. No exceptions cross the boundary.
 *   - one context per GPU; a context is not re-entrant.
 *   - there is NO CPU fallback: without a gfx950 device pf_ctx_create fails.
 */
#ifndef PYFLYT_AMD_H
#define PYFLYT_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PF_ABI_VERSION 8
#define PF_MAX_TARGETS 8
#define PF_MAX_BOXES 12
#define PF_MAX_SURF 5
#define PF_MAX_CONTACTS 48 /* contact vertices solved per body per tick (first in collider / vertex order) */

enum pf_status { PF_OK = 0, PF_ERR_ARG = -1, PF_ERR_UNSUPPORTED = -2, PF_ERR_NO_DEVICE = -3 };
enum pf_vehicle { PF_QUADX = 0, PF_FIXEDWING = 1, PF_ROCKET = 2 /* Aviary-level entry points only */ };
enum pf_task { PF_TASK_NONE = 0, PF_TASK_HOVER = 1, PF_TASK_WAYPOINTS = 2, PF_TASK_MA_HOVER = 3,
               PF_TASK_DOGFIGHT = 4 /* MAFixedwingDogfightEnv (pz_envs/fixedwing_envs/ma_fixedwing_dogfight_env.py), fixedwing only */ };
enum pf_noise { PF_NOISE_OFF = 0, PF_NOISE_INJECT = 1, PF_NOISE_PHILOX = 2 };
enum pf_autoreset { PF_AUTORESET_OFF = 0, PF_AUTORESET_NEXT_STEP = 1, PF_AUTORESET_SAME_STEP = 2 };

/* bits of the per-lane `flags` word (state group PF_G_INT, .y) */
enum pf_flag {
  PF_F_TERMINATED = 1, PF_F_TRUNCATED = 2, PF_F_CONTACT = 4, /* contact after the last tick */
  PF_F_INFO_COLLISION = 8, PF_F_INFO_OOB = 16, PF_F_INFO_COMPLETE = 32,
  /* a state word of the lane is NaN/Inf after an env step (the reference would carry the NaN on silently:
   * e.g. 0 * inf in the mixer's saturation rescale, quadx.py:490-491, when hi == pmin). Sticky until the
   * lane is reset; surfaced as infos["nonfinite"]; counted by bench.py. */
  PF_F_NONFINITE = 64
};

typedef struct pf_pid {
  float kp[3], ki[3], kd[3], lim[3];
} pf_pid;

typedef struct pf_box {
  float c[3], h[3]; /* centre in the base frame; box: half extents, cylinder: radius, radius, half length */
  int32_t kind;     /* 0 box, 1 cylinder along the link z axis (primitive_drone.urdf:42-47 prop discs) */
  float yaw;        /* rotation of the shape's link about the base z axis (rocket.urdf:251,277 legs) */
} pf_box;

/* one lifting surface: abstractions/lifting_surfaces.py:141-239 (constants precomputed on host) */
typedef struct pf_surface {
  float r[3];                       /* link COM offset in the base frame */
  float lift[3], drag[3], torque[3]; /* unit vectors */
  float Cl_alpha_3D, inv_Cl_alpha_3D, aero_tau_eta; /* Cl3D, 1/Cl3D, aero_tau*eta */
  float flap_to_chord, inv_pi_aspect, exp_term;     /* 1/(pi*AR), 0.41*(1-exp(-17/AR)) */
  float alpha_0_base, alpha_stall_P_base, alpha_stall_N_base; /* radians */
  float Cd_0, deflection_limit_rad, dt_over_tau;
  float half_rho_area, chord;
} pf_surface;

/* All constants of one batched simulation. Filled by the host from its own parameter tables
 * (pyflyt_amd/params.py; numbers from cf2x.yaml/.urdf and fixedwing.yaml/.urdf, cited there). */
/* Rocket (drones/rocket.py, abstractions/boosters.py, gimbals.py, models/vehicles/rocket/): the fuel
 * tank's mass and inertia change every tick (boosters.py:193-198), so the composite body is rebuilt per
 * tick from these aggregates over the other ("dry") links. */
typedef struct pf_rocket {
  float dry_mass;              /* sum m_i */
  float dry_mr[3];             /* sum m_i r_i */
  float dry_S[6];              /* sum m_i ((r_i.r_i) 1 - r_i r_i^T), symmetric xx xy xz yy yz zz */
  float dry_I[3];              /* sum of the links' own (diagonal) inertias */
  float tank_r[3];             /* fuel tank link COM in the base frame */
  float total_fuel, fuel_rate_ratio /* max_fuel_rate / total_fuel */, fuel_inertia[3];
  float thrust_min_ratio /* min_thrust / max_thrust */, max_thrust, booster_dt_over_tau, booster_noise;
  int32_t reignitable;
  float booster_r[3];          /* booster link COM */
  float gimbal_dt_over_tau, gimbal_range_rad;
  float finlet_map[4][3];      /* rocket.py:152-159 */
  float starting_fuel_ratio;   /* rocket.py:47 */
} pf_rocket;

typedef struct pf_params {
  int32_t vehicle;  /* pf_vehicle */
  int32_t task;     /* pf_task */
  int32_t flight_mode;        /* quadx: -1..7 (quadx.py:233-259); fixedwing: -1, 0 */
  int32_t noise_mode;         /* pf_noise */
  int32_t autoreset;          /* pf_autoreset */
  int32_t angle_repr;         /* 0 euler, 1 quaternion (quadx_base_env.py:62-69) */
  int32_t sparse_reward;
  int32_t num_targets;
  int32_t max_steps;          /* agent_hz * max_duration_seconds (quadx_base_env.py:121) */
  int32_t env_step_ratio;     /* 120 / agent_hz (quadx_base_env.py:122) */
  int32_t settle_steps;       /* 10 (quadx_base_env.py:209) */
  int32_t ticks_per_control;  /* physics_hz / control_hz (base_drone.py:102) */
  int32_t use_gyro_term;      /* [BULLET-FROM-MEMORY] btMultiBody::m_useGyroTerm */
  int32_t throttle_remap;     /* fixedwing_base_env.py:260 */
  int32_t n_motors, n_surf, n_boxes;
  int32_t has_com_offset;
  uint64_t seed;

  /* world / integrator: aviary.py:79,226 + Bullet defaults */
  float dt, gravity_z, max_coord_vel;
  float plane_half_xy, plane_half_z;
  /* Contact RESPONSE -- what stepSimulation (core/aviary.py:516) does after collision detection. [BULLET-FROM-MEMORY]
   * throughout: a named-parameter model (NOT btMultiBodyConstraintSolver digit for digit), every doubtful Bullet fact a field
   * with the best-known default (pyflyt_amd/params.py: WORLD; the argument for each default: DESIGN.md section 3;
   * tests/golden/capture_pybullet.py prints getPhysicsEngineParameters() so that one run where PyBullet exists settles them).
   *  - Contact points against the ground slab, at the pre-integration pose: per collider BOX the four vertices of the face
   *    that looks down the most (contact_manifold_points = 4: dBoxBox2 clips the incident face and is called with maxc = 4,
   *    btPersistentManifold holds four points; 8 = every vertex), per cylinder 8 rim points on either end disc; a vertex is a
   *    point when it is at most `reach` above the slab's top face (and over the slab), reach = contact_margin for a body
   *    without contact points after the previous tick (0: dBoxBox2 returns nothing while an axis separates the boxes),
   *    contact_break_distance for one that had some (btPersistentManifold keeps a point until the gap exceeds the contact
   *    breaking threshold, 0.02 m -- what lets a body REST instead of rattling on the two corners it has not just lifted).
   *  - Rows: normal impulse >= 0 towards contact_restitution x approach speed, or towards -(gap + slop) / dt for a point
   *    still above face - slop ("do not close more than the gap this tick"); two world-axis friction rows clamped to
   *    contact_friction x normal impulse. Projected Gauss-Seidel at the velocity level in collider / vertex order: at most
   *    contact_iters sweeps (50 = PyBullet's numSolverIterations), ended early once the largest squared change of a row's
   *    velocity within a sweep is <= contact_residual_threshold (1e-7 = PyBullet's solverResidualThreshold, i.e. 3.2e-4 m/s;
   *    0: only an idle sweep ends it). Every solve starts from zero impulses: no warm start, as in Bullet's multibody
   *    solver, where it is switched off (setupMultiBodyContactConstraint: `if (0)`) [BULLET-FROM-MEMORY].
   *  - After the position update a translation of contact_erp x (deepest penetration - contact_slop) along +z.
   *  - contact_response = 0: detection only (bodies fall through the floor).
   * Contact REPORT (getContactPoints, core/aviary.py:523-525): the 15-axis box-box verdict against the other box ENLARGED by
   * contact_report_distance (0: reported from touching on) -- by contact_break_distance for a body that held contact points
   * after the previous tick.
   * Shared worlds (agents_per_world > 1, the QuadX PettingZoo task): the same model BETWEEN the drones, one stage earlier in
   * the tick -- velocities after the forces -> pair stage -> ground solve per body -> integration. Pair contacts at the
   * pre-integration poses: for every ordered pair (a, b), a != b, in agent order, every box of a against every box of b, the
   * 8 vertices of a's box in vertex order: a vertex within `reach` of being inside b's box is a contact (reach as above, the
   * breaking distance when either body held contact points), its normal the face of b with the least penetration (first axis
   * on a tie) pointing out of b, its depth that penetration; at most 16 per world and tick. Rows: the normal and Bullet's
   * btPlaneSpace1 tangents on the relative point velocity; the ground rows' targets, sweeps and residual exit, friction clamp
   * contact_friction^2 x normal impulse; after the position update each body moves half of contact_erp x (its deepest pair
   * penetration - slop) along that normal (a: +, b: -).
   * (pz_envs/quadx_envs/ma_quadx_base_env.py:365-369: a culled drone that lands on a live one.) */
  int32_t contact_response, contact_iters;
  float contact_restitution, contact_friction, contact_erp;
  float contact_margin;              /* fresh contact points: how far above the face (0) */
  float contact_slop;                /* allowed overlap (1e-5 = PyBullet's m_linearSlop): what a resting body sinks in by */
  float contact_report_distance;     /* fresh pairs are reported from this gap on (0) */
  float contact_break_distance;      /* ... persisting ones up to this gap (0.02), and keep their points up to it */
  float contact_residual_threshold;  /* squared row-velocity change that ends the sweeps (1e-7) */
  int32_t contact_manifold_points;   /* 4: the incident face of a box; 8: every vertex */
  /* composite body */
  float inv_mass;
  float com[3];
  float I_own[6], I_pa[6], I_inv[6]; /* symmetric: xx xy xz yy yz zz */
  float bound_radius;
  pf_box boxes[PF_MAX_BOXES];
  /* motors: motors.py:110-195 */
  float motor_r[4][3];
  float thrust_unit[4][3];
  float motor_dt_over_tau[4], motor_fmax[4] /* Ct*max_rpm^2 */, motor_tmax[4] /* Cq*max_rpm^2, signed */;
  float motor_noise[4];
  /* quadx */
  float motor_map[4][4];      /* quadx.py:130-137 */
  float drag_const[3];        /* boring_bodies.py:63 */
  float drag_coef_pqr;        /* cf2x.yaml:11 */
  pf_pid pid[4];              /* ang_vel, ang_pos, lin_vel, lin_pos (cf2x.yaml:13-41) */
  pf_pid zpid[2];             /* z_vel, z_pos (cf2x.yaml:43-54) */
  float control_period, inv_control_period;
  /* fixedwing */
  pf_surface surf[PF_MAX_SURF];
  int32_t assist_ids[6];      /* fixedwing.py:143 */
  float assist_signs[6];      /* fixedwing.py:144 */
  /* env */
  float start_pos[3], start_quat[4], start_vel[3];
  float dome, goal_reach_distance, min_height;
  float wp_dist_reward, wp_yaw_penalty;
  /* QuadX-Waypoints yaw targets (quadx_waypoints_env.py:40-42, waypoint_handler.py:85-89,144-156,167-179): one more
   * uniform per target at reset (drawn after all the positions), target deltas 4 wide (body-frame delta + wrapped
   * yaw error), a target counts as reached only when the yaw error is under goal_reach_angle as well */
  int32_t use_yaw_targets;
  float goal_reach_angle;
  float action_low[4], action_high[4]; /* action space box (quadx_base_env.py:80-102) */
  /* PF_TASK_MA_HOVER: agents per SHARED world (pz_envs put every agent's drone in one Bullet world,
   * ma_quadx_base_env.py:206-241). 0 / 1 = every lane alone in its world. A > 1: lanes [w A, (w+1) A) are one world -- a hit
   * between two of its drones enters both contact arrays and ends both episodes (ma_quadx_hover_env.py:181), and a contact
   * point anywhere in the world switches off every drone's rotational drag (quadx.py:509); the drones push each other (the
   * pair stage above; box colliders). A must divide 64 and the lane count, and be at most 8. */
  int32_t agents_per_world;
  /* PF_TASK_DOGFIGHT (ma_fixedwing_dogfight_env.py:42-60): two teams of df_team_size aircraft in one shared world,
   * agents_per_world = 2 df_team_size <= 8 adjacent lanes, lanes [0, team) of a world one team, the rest the other.
   * df_sample_spawn: 1 = every reset draws the world's spawn circle (_get_start_pos_orn :176-213) from the counter RNG keyed
   * by the world's first lane; 0 = the spawn pose is read from the state's spawn groups (pos, rpy). The
   * spawn velocity is 20 m/s along the nose (:216-222). Observation = [attitude 12, surfaces + throttle 6, health, past
   * action 4] + 14 per other ACTIVE aircraft in index order (its attitude in the own body frame 12, its health, same-team
   * flag), zero padded to 23 + 14 (A - 1) (:519-549, :724-752). pf_env_step pops reward / terminated / truncated for the
   * agents still in the episode (PettingZoo parallel API: finished agents are culled, their aircraft fly on with zero
   * commands, ma_fixedwing_base_env.py:289-330); pf_params.autoreset must be PF_AUTORESET_OFF. */
  /* Filled in by pf_ctx_create (callers leave it 0): the most contact points the contact solve can see for this airframe --
   * its collider vertices (contact_manifold_points per box, 16 per cylinder), at most PF_MAX_CONTACTS. Sizes the solver's LDS regions, i.e. how many
   * lanes of a wave can be solved side by side. */
  int32_t contact_max_points;
  /* df_freeze_wrecks (default 0 = the reference's behaviour: a crashed aircraft keeps tumbling in the physics until it comes
   * to rest): 1 = an aircraft stops where it hits the ground -- velocities zeroed, not integrated any further, `inactive` from
   * the next update on. Nothing an agent is rewarded for depends on a wreck's tumbling; it spares the contact solve, which
   * otherwise dominates the step time of every wave that has an aircraft on the ground. */
  /* df_action_dim: 4 (assisted_flight=True, the default; 0 means 4) or 6 (assisted_flight=False). pf_buffers.actions is then
   * [n][df_action_dim]. With 6 the reference still leaves the Aviary in flight mode 0 (ma_fixedwing_base_env.py:229), which reads
   * setpoint[0:4] (fixedwing.py:246-250): entries 4 and 5 only appear in the observation's past action, and the thrust remap
   * of :300-301 lands on entry 5 -- the thrust command is action[3] as given. Reproduced as it is. */
  int32_t df_team_size, df_sample_spawn, df_freeze_wrecks, df_action_dim;
  float df_spawn_min_radius, df_spawn_max_radius;
  float df_damage_per_hit, df_lethal_distance, df_lethal_angle, df_aggressiveness, df_cooperativeness;
  pf_rocket rocket;
} pf_params;

/* Device buffers of one call. state layout: float4 groups, [n_groups][n_lanes][4]. The state belongs to the
 * library between calls: besides the lane's physical state it holds what the kernels prepare ahead -- for the QuadX Hover /
 * Waypoints tasks groups 7-11 carry the lane's "spare" (the random part of its NEXT episode: settled spawn state, targets) and the key
 * its next reset draws from. A caller that writes a state by hand clears bit 31 of group 7's fourth word (group 11's third in the
 * cascaded flight modes) and leaves the key in bits 0-30: the kernels then generate at the reset. A spare is only valid for the
 * context that made it (seed, lane offset, spawn pose, settle length, dome, number of targets): pf_env_reset with a NULL mask -- every
 * lane -- therefore ignores the spares it finds and prepares fresh ones, which is what a caller does first with a state buffer that
 * another context has used; a MASKED reset trusts them. The key is the event counter as the lane's previous reset left it: strictly
 * increasing from reset to reset (0 before the first).
 * QuadX env contexts in a cascaded flight mode (flight_mode != 0) on the specialised kernel without a shared world have 27 groups
 * (pf_state_groups): groups 16-26 hold the float32 REMAINDERS of values the kernel carries in fp64 -- 16-19 position, quaternion,
 * velocities, motor states; 20-21 the rate PID's memories; 22-26 the cascade's memories in the packing of groups 7-11 -- next to
 * their float32 roundings in groups 0-5 / 7-11. A hand-written state leaves them zero (the value is then its float32 word). */
typedef struct pf_buffers {
  float* state;            /* [pf_state_groups()][n][4] fp32/int32, persistent */
  const float* actions;    /* [n][4]   gym action (quadx_base_env.py:269); PF_TASK_DOGFIGHT with df_action_dim 6: [n][6] */
  float* obs;              /* [n][pf_obs_dim()] row-major */
  float* final_obs;        /* [n][pf_obs_dim()] or NULL; written for finished lanes under SAME_STEP */
  float* reward;           /* [n] */
  uint8_t* terminated;     /* [n] */
  uint8_t* truncated;      /* [n] */
  const float* xi;         /* PF_NOISE_INJECT: [env_step_ratio*ticks_per_control][n] raw motor-noise draws */
  const float* xi_reset;   /* PF_NOISE_INJECT: [settle_steps*ticks_per_control][n] */
  const float* u_targets;  /* PF_NOISE_INJECT, waypoint tasks: [3*num_targets][n] theta|phi|dist draws (+ [num_targets][n] yaw with use_yaw_targets) */
  /* Aviary-level calls only */
  const float* setpoints;  /* [n][4] (quadx, fixedwing mode 0), [n][6] (fixedwing mode -1) or [n][7] (rocket) */
  float* out_state;        /* [n][12]: ang_vel, ang_pos, lin_vel, lin_pos rows of Aviary.state(i) */
  float* out_aux;          /* [n][4] quadx throttle | [n][6] fixedwing surfaces + throttle | [n][9] rocket fins, ignition, fuel, throttle, gimbal */
  uint8_t* out_contact;    /* [n] contact_array[planeId] after the step, or NULL */
  const float* start_pose; /* pf_aviary_reset: [n][7] per-lane spawn (pos xyz, quat xyzw) or NULL = pf_params */
  /* wind field (pf_aviary_tick / pf_aviary_reset; ABI 2). K = pf_wind_links(): the links the reference
   * samples its wind field at -- QuadX 1 (body link, boring_bodies.py:93-96), Fixedwing 5 (surface
   * links, lifting_surfaces.py:88-93) */
  const float* wind;       /* [n][K][3] world-frame wind velocity at those links as of the previous update_state, or NULL */
  float* out_link_pos;     /* [n][K][3] world positions of those links after the call, or NULL */
  /* per-drone control rate (pf_aviary_step / pf_aviary_tick): physics ticks between controller updates of
   * each drone, a divisor of ticks_per_control (= physics_hz / the SLOWEST drone's control_hz,
   * aviary.py:288-289; drones given different `control_hz`, tests/test_core.py:34-62). NULL = uniform. */
  const int32_t* ctrl_ratio; /* [n] */
  /* per-drone flight modes (QuadX; Aviary.set_mode with a list, core/aviary.py:440-458): read by
   * pf_aviary_set_mode / _step / _tick instead of the context's mode. NULL = one mode for all. */
  const int32_t* modes;      /* [n] */
  /* per-drone spawn velocity for pf_aviary_reset (drone_options[i]["starting_velocity"], fixedwing.py:35,
   * ma_fixedwing_dogfight_env.py:218-222): world-frame linear velocity [n][3], NULL = pf_params.start_vel */
  const float* start_vel;
  /* Aviary.set_armed (core/aviary.py:423-438,510-521): a disarmed drone gets no controller update, no motor /
   * aerodynamic / drag forces and no state read-back -- PyBullet still integrates it under gravity, its
   * out_state / out_aux rows keep their last values. [n] bytes, NULL = all armed. */
  const uint8_t* armed;
  /* ABI 3 */
  /* SAME_STEP auto-reset: [n][2] int32 (flags word, targets left) of the finished episode as they were BEFORE the
   * lane was re-initialised -- gymnasium's `final_info` next to `final_obs`; NULL = not reported */
  int32_t* final_info;
  /* pf_rollout only: the sampled / consumed action of every step, [k_steps][n][4], or NULL = not stored */
  float* actions_out;
  /* pf_body_tick only: [n][6] body-frame force (3) and torque (3) applied at the base link for every tick */
  const float* wrench;
} pf_buffers;

typedef struct pf_ctx pf_ctx;

int pf_abi_version(void);
/* struct sizes as compiled, so that a foreign-language binding can verify its mirror of the structs */
size_t pf_sizeof_params(void);
size_t pf_sizeof_buffers(void);
/* message of the last failing call (per context, or global when ctx is NULL) */
const char* pf_last_error(const pf_ctx* ctx);

/* Replaces constructing `Aviary(...)` + the drone objects (core/aviary.py:69-216,
 * core/drones/quadx.py:22-220, fixedwing.py:18-192): binds the parameter block to a device.
 * lane_offset = global index of lane 0 (multi-GPU sharding; keys the counter-based RNG).
 * The context owns one device allocation: a copy of the parameter block (read by the rarely-taken floor paths). */
int pf_ctx_create(const pf_params* params, int n_lanes, int device, uint64_t lane_offset, pf_ctx** out);
void pf_ctx_destroy(pf_ctx* ctx);
int pf_state_groups(const pf_ctx* ctx); /* float4 groups per lane in pf_buffers.state */
int pf_obs_dim(const pf_ctx* ctx);
int pf_n_lanes(const pf_ctx* ctx);
/* which env kernel pf_env_step / pf_env_reset launch for this context: 0 the generic env_kernel, 1 the
 * specialised QuadX mode-0 kernel (Hover / Waypoints / MA-Hover), 2 the specialised Fixedwing-Waypoints kernel */
int pf_ctx_is_specialised(const pf_ctx* ctx);

/* env.reset(): gym_envs/quadx_envs/quadx_base_env.py:149-212 (begin_reset + end_reset incl. the
 * 10 settle Aviary steps), quadx_hover_env.py:70-83, quadx_waypoints_env.py:112-125,
 * fixedwing_waypoints_env.py:101-114. mask (device, [n] bytes) selects lanes; NULL = all. Shared worlds (agents_per_world > 1):
 * the agents of a world are reset together -- a mask that selects some of them is widened to the whole world on the device. */
int pf_env_reset(pf_ctx* ctx, const pf_buffers* b, const uint8_t* mask, void* stream);
/* env.step(action): quadx_base_env.py:269-301 / fixedwing_base_env.py:244-278 with the task's
 * compute_state + compute_term_trunc_reward, env_step_ratio x Aviary.step() (core/aviary.py:480-531)
 * fused in one launch; auto-reset per pf_params.autoreset. */
int pf_env_step(pf_ctx* ctx, const pf_buffers* b, void* stream);

/* Aviary-level surface (core/aviary.py): reset :218-312, set_mode :440-458, step :480-531 with
 * set_all_setpoints :470-478 folded in (b->setpoints), state/aux_state :335-369 -> out_state/out_aux.
 * n_steps Aviary steps are fused in one launch (setpoints held, as the reference holds them). */
int pf_aviary_reset(pf_ctx* ctx, const pf_buffers* b, void* stream);
/* setpoints_out: [n][4] (or [n][6] for fixedwing mode -1), read-modify-written with the mode's default
 * setpoint (quadx.py:275-290) */
int pf_aviary_set_mode(pf_ctx* ctx, const pf_buffers* b, int mode, float* setpoints_out, void* stream);
int pf_aviary_step(pf_ctx* ctx, const pf_buffers* b, int n_steps, void* stream);

/* ONE physics tick of Aviary.step (aviary.py:510-531), for callers that must get between the ticks:
 * a wind field (aviary.py:266-285,324-333; base_wind_field.py) is sampled by the reference in every
 * update_state at the link positions, and feeds the next tick's drag / aerodynamic velocities.
 * tick_index = position of the tick inside the Aviary step (0 .. ticks_per_control-1): the controller
 * runs at tick 0 (quadx.py:409), the motor commands are carried to the later ticks in state group 12
 * (QuadX) or recomputed from the setpoint (Fixedwing, stateless mixing). b->wind (may be NULL) is
 * subtracted from the link velocities; b->out_link_pos receives where to sample the field next;
 * b->out_contact the contact verdict of this tick. PF_NOISE_INJECT: b->xi holds this tick's draws [n].
 * The host side of the protocol is pyflyt_amd/core/aviary.py (Aviary.step with a wind field). */
int pf_aviary_tick(pf_ctx* ctx, const pf_buffers* b, int tick_index, void* stream);
int pf_wind_links(const pf_ctx* ctx);

/* Synthetic uniform actions inside [action_low, action_high] for benchmark rollouts
 * (the role of env.action_space.sample(), tests/test_gym_envs.py:104), keyed by
 * (seed, global lane, step_index). */
int pf_sample_actions(pf_ctx* ctx, float* actions, uint32_t step_index, void* stream);

/* k_steps consecutive env.step() calls (quadx_base_env.py:269-301 incl. auto-reset) in ONE launch with the
 * per-lane state resident in registers between the steps: the synthetic random-action rollout of
 * tests/test_gym_envs.py:100-110 (`env.step(env.action_space.sample())` in a loop) without the per-step
 * round trip of the state through HBM. Nothing is skipped: every step writes its observation, reward and
 * flags. Buffers are trajectories: b->obs [k_steps][n][obs_dim], b->reward / terminated / truncated
 * [k_steps][n], b->final_obs / final_info (SAME_STEP) [k_steps][n][..]. Actions: b->actions == NULL samples
 * step s of lane i exactly as pf_sample_actions(step_index0 + s) would (same Philox keys) and, if
 * b->actions_out != NULL, stores it there; otherwise b->actions is a given open-loop sequence
 * [k_steps][n][4]. Results are bit-identical to k_steps x (pf_sample_actions + pf_env_step).
 * State-resident on every env kernel: the specialised ones (QuadX Hover / Waypoints / multi-agent Hover with level spawns in any
 * flight mode, Fixedwing-Waypoints), the dogfight on either aircraft model (ma_fixedwing_base_env.py:272-334 in a loop,
 * tests/test_pz_envs.py:71-93; sampled actions four-wide, or a given sequence of either width [k_steps][n][4 | 6]) and the generic
 * env kernel behind every other configuration. PF_NOISE_OFF / PHILOX; PF_ERR_UNSUPPORTED for PF_NOISE_INJECT, for contexts without
 * an env task, for an auto-reset mode of OFF on the specialised single-agent kernels, and for sampling six-wide dogfight actions. */
int pf_rollout(pf_ctx* ctx, const pf_buffers* b, int k_steps, uint32_t step_index0, void* stream);

/* The reference's LOWER boundary for one drone: applyExternalForce / applyExternalTorque on the base link in
 * LINK_FRAME followed by stepSimulation (core/drones/quadx.py:502-510, core/aviary.py:516), n_ticks times with
 * the wrench b->wrench held: the free-body tick alone (collision detection, gyroscopic term, +-max_coord_vel
 * clamp, exponential-map attitude update), no motors / drag / controller. Exists so that the integrator's
 * analytic known-answer tests (free fall, constant torque, torque-free spin, clamp) run against the device
 * code directly. Fills b->out_state / out_contact. */
int pf_body_tick(pf_ctx* ctx, const pf_buffers* b, int n_ticks, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PYFLYT_AMD_H */
