/*
 * uav_oracle.h -- TEST INFRASTRUCTURE ONLY (parity oracle, fp64, scalar, one drone at a time).
 *
 * CPU restatement of the PyFlyt `Aviary.step()` hot path (reference @ /root/reference, v0.30.0)
 * and of the slice of Bullet's free-multibody step that PyFlyt drives underneath it.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * The product path (pyflyt_amd/) never links, imports or calls it.
 *
 * PARITY STATUS
 *   - PyFlyt-side arithmetic (PID, mixer, motors, drag, lifting surfaces, env logic): pinned
 *     against golden vectors generated in the build container by importing the reference's own
 *     Python (tests/golden/gen_*.py).
 *   - Bullet-side arithmetic (integrator, contact reporting, quaternion/Euler helpers):
 *     "parity unpinned" -- pybullet (unpinned dependency, pyproject.toml:19) is absent from
 *     /root/reference and from the container; those functions restate Bullet3's published
 *     algorithm from memory and are marked [BULLET-FROM-MEMORY]; every doubtful constant is a
 *     named parameter in orc_world.
 */
#ifndef UAV_ORACLE_H
#define UAV_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MAX_TARGETS 8
#define ORC_MAX_LINKS 12
#define ORC_MAX_BOXES 12
#define ORC_MAX_SURF 5

enum { ORC_QUADX = 0, ORC_FIXEDWING = 1, ORC_ROCKET = 2 };
enum { ORC_TASK_NONE = 0, ORC_TASK_HOVER = 1, ORC_TASK_WAYPOINTS = 2, ORC_TASK_MA_HOVER = 3 };
enum { ORC_NOISE_OFF = 0, ORC_NOISE_INJECT = 1, ORC_NOISE_PHILOX = 2 };

/* World / integrator knobs -- aviary.py:225-242 + [BULLET-FROM-MEMORY] defaults */
typedef struct {
  double dt;                 /* 1/physics_hz, aviary.py:79,164 */
  double gravity_z;          /* -9.81, aviary.py:226 */
  int use_gyro_term;         /* btMultiBody::m_useGyroTerm, believed default true */
  double max_coord_vel;      /* btMultiBody::m_maxCoordinateVelocity = 100 */
  double plane_half_xy;      /* plane.urdf collision box 30x30x10 -> 15 * world_scale */
  double plane_half_z;       /* 5 * world_scale; box centre at z = -plane_half_z */
  int ticks_per_control;     /* physics_hz / control_hz = 2, base_drone.py:102 */
  /* Contact RESPONSE against the ground slab (aviary.py:516 stepSimulation, the part after collision detection).
   * [BULLET-FROM-MEMORY], a named-parameter model, NOT Bullet's btMultiBodyConstraintSolver digit for digit:
   * contact points = the collider vertices (box corners; 8 rim points on either end disc of a cylinder) found at
   * or below the slab's top face at the pre-integration pose; projected Gauss-Seidel over them at the velocity level
   * (normal impulse >= 0 towards restitution * approach speed, two world-axis friction directions clamped to
   * friction * normal impulse), contact_iters sweeps in collider / vertex order; then, after the position update, a
   * translation of contact_erp * (deepest penetration) along +z (Bullet's split-impulse style recovery: no energy
   * is injected). Bullet's defaults: restitution 0, lateral friction 0.5 (body) x 1.0 (plane.urdf), erp 0.2. */
  int contact_response;      /* 1: solve contacts (default); 0: detection only, bodies pass through the floor */
  double contact_restitution, contact_friction, contact_erp;
  int contact_iters;
  /* speculative margin: vertices up to this far ABOVE the face are in the contact set too, with the constraint
   * "do not close more than the gap in this tick" (normal velocity >= -gap / dt). It binds only when the vertex would
   * otherwise penetrate within the tick, and it keeps the vertices of a resting body in the active set instead of
   * letting them drop in and out of it (Bullet's contact breaking threshold plays this role: 0.02 m). */
  double contact_margin;
  /* allowed penetration: the constraints let a vertex sink contact_slop below the face and the recovery only acts on
   * what is deeper, so a body at rest overlaps the slab by exactly this much -- which keeps the contact REPORT
   * (penetration >= 0, orc_contact_plane) true and stable while it rests, instead of flickering at a zero gap */
  double contact_slop;
  /* Contact response BETWEEN the drones of a shared world (the PettingZoo envs put every agent's drone in one Bullet world,
   * ma_quadx_base_env.py:206-241; a culled drone that falls onto a live one pushes it, :365-369). [BULLET-FROM-MEMORY], the
   * same named-parameter model as the ground's, one stage earlier in the tick:
   *   velocities after the forces -> PAIR STAGE -> ground solve per body -> integration.
   * Pair contacts at the pre-integration poses: for every ordered pair (a, b), a != b, in body order, every box of a against
   * every box of b (plain boxes), the 8 vertices of a's box in vertex order: a vertex within contact_margin of being inside b's
   * box is a contact; normal = b's face with the least penetration (first axis on a tie), pointing out of b; depth = that
   * penetration. At most ORC_MAX_PAIR_CONTACTS per world and tick (the first in order). Rows: the normal and Bullet's
   * btPlaneSpace1 tangents; relative point velocity u = (v_a + w_a x r_a) - (v_b + w_b x r_b); projected Gauss-Seidel,
   * contact_iters sweeps in contact order, same targets as the ground rows (slop, speculative margin, restitution), friction
   * clamp contact_friction^2 x the normal impulse (Bullet multiplies the two bodies' friction coefficients). After the
   * position update each body is translated by half of contact_erp x (its deepest pair penetration - slop) along that
   * contact's normal (a: +, b: -). */
  int pair_response;         /* 1: drone-drone impulses in shared worlds (default); 0: detection only */
  /* ---- round 4: the doubtful Bullet facts of the contact model, each a named parameter with the best-known default
   * ([BULLET-FROM-MEMORY] throughout; tests/golden/capture_pybullet.py prints getPhysicsEngineParameters() so that one run
   * on a machine with PyBullet settles them) ----
   * contact_report_distance: getContactPoints (aviary.py:523-525) lists a body pair from this gap on -- the 15-axis verdict
   *   against the other box ENLARGED by it. Default 0: btBoxBoxDetector (dBoxBox2) returns no point as soon as one of its 15
   *   axes separates the boxes, so a box-box pair is first reported when it touches. (A point that exists stays in the
   *   persistent manifold until the gap exceeds the contact breaking threshold, 0.02 m: a RECEDING pair is reported a little
   *   longer. Not modelled: the env tasks end the episode at the first report.)
   * contact_margin (above): 0 by default for the same reason -- no constraint row exists before the boxes overlap.
   * contact_manifold_points: at most this many points per collider box and slab (4: btPersistentManifold holds four, dBoxBox2
   *   is called with maxc = 4): the vertices of the box face that looks down the most (the incident face). 8: every vertex.
   * contact_iters: 50 = PyBullet's numSolverIterations default (Bullet's own default is 10). Every solve starts from zero
   *   impulses (btMultiBodyConstraintSolver::setupMultiBodyContactConstraint has the warm start switched off [BULLET-FROM-MEMORY]).
   * contact_residual_threshold: the sweeps stop once the largest squared change of a row's velocity in a sweep is at or below
   *   it (btMultiBodyConstraintSolver::solveSingleIteration's leastSquaredResidual against m_leastSquaresResidualThreshold,
   *   which PyBullet's server sets to 1e-7, i.e. 3.2e-4 m/s). 0: only an exactly idle sweep ends the solve early.
   * contact_slop: 1e-5 = PyBullet's m_linearSlop (the allowed overlap; Bullet's own default is 0). */
  double contact_report_distance;
  double contact_residual_threshold;
  int contact_manifold_points;
  /* contact_break_distance: a body that held contact points after the previous tick keeps them while the gap stays under
   * btPersistentManifold's contact breaking threshold (0.02 m): for such a body the vertices of the incident face up to this
   * far above the face are contact points (rows "do not close more than the gap this tick") and the pair keeps being
   * reported up to this gap. It is what lets a body REST: with points that exist only while they overlap, the two corners a
   * resting box lifts by a micrometre leave the set and the box rattles. Restated per body (Bullet: per manifold point);
   * between two drones: when either holds contact points. */
  double contact_break_distance;
} orc_world;
#define ORC_MAX_PAIR_CONTACTS 16
#define ORC_MAX_WORLD 8 /* drones per shared world (the device: agents_per_world <= 8) */

typedef struct {
  double kp[3], ki[3], kd[3], lim[3];
} orc_pid_gains;

typedef struct {
  double c[3];  /* centre in base frame */
  double h[3];  /* box: half extents; cylinder (axis = link z): radius, radius, half length */
  int kind;     /* 0 box, 1 cylinder (primitive_drone.urdf:42-47: the prop discs) */
  double yaw;   /* rotation of the shape's link about the base z axis (rocket.urdf:251,277: the legs) */
} orc_box;

/* One lifting surface -- lifting_surfaces.py:141-239, fixedwing.yaml:8-71 */
typedef struct {
  double r[3];          /* link COM offset in base frame (fixedwing.urdf joint origins) */
  double lift_unit[3], drag_unit[3], torque_unit[3];
  double Cl_alpha_2D, chord, span, flap_to_chord, eta;
  double alpha_0_base, alpha_stall_P_base, alpha_stall_N_base; /* radians (deg2rad applied) */
  double Cd_0, deflection_limit /* degrees */, tau;
  /* precomputed (lifting_surfaces.py:228-239) */
  double half_rho, area, aspect, Cl_alpha_3D, theta_f, aero_tau;
} orc_surface;

typedef struct {
  int vehicle;               /* ORC_QUADX / ORC_FIXEDWING */
  orc_world world;

  /* composite rigid body (cf2x.urdf:13-14 ; fixedwing.urdf point masses) */
  double mass;
  double com[3];             /* composite COM in base frame */
  double I_own[3][3];        /* sum of the links' own inertia tensors (gyro-term flag applies) */
  double I_pa[3][3];         /* parallel-axis part about the composite COM */
  double I_inv[3][3];        /* (I_own + I_pa)^-1 */
  int n_boxes;
  orc_box boxes[ORC_MAX_BOXES];
  double bound_radius;       /* sphere around base origin containing all boxes */

  /* motors -- motors.py, quadx.py:93-128, fixedwing.py:147-168 */
  int n_motors;
  double motor_r[4][3];      /* link COM offsets (cf2x.urdf:42,54,66,78) */
  double thrust_unit[4][3];
  double thrust_coef[4], torque_coef[4], max_rpm[4], motor_tau[4], noise_ratio[4];

  /* quadx only */
  double motor_map[4][4];    /* quadx.py:130-137 */
  double drag_const[3];      /* 0.5*1.225*Cd*A, boring_bodies.py:63 */
  double drag_coef_pqr;      /* cf2x.yaml:11 */
  orc_pid_gains pid[4];      /* 0 ang_vel 1 ang_pos 2 lin_vel 3 lin_pos */
  orc_pid_gains zpid[2];     /* 0 z_vel 1 z_pos  (quadx.py:205: z_PIDs=[z_vel,z_pos]) */
  double control_period;     /* 1/control_hz */

  /* fixedwing only */
  int n_surf;
  orc_surface surf[ORC_MAX_SURF]; /* order: L-ail, R-ail, h-tail, v-tail, main (fixedwing.py:80-138) */
  int assist_ids[6];
  double assist_signs[6];

  /* env (task) constants */
  int task;
  int flight_mode;
  double start_pos[3], start_rpy[3], start_vel[3];
  double dome;
  int max_steps;
  int env_step_ratio;
  int settle_steps;          /* 10 Aviary steps, quadx_base_env.py:209 */
  int sparse_reward;
  int angle_repr;            /* 0 euler, 1 quaternion */
  int num_targets;
  double goal_reach_distance;
  double min_height;
  double throttle_remap;     /* 1 -> a[3]/2+0.5 (fixedwing_base_env.py:260) */
  int collide_any;           /* fixedwing: any contact ; quadx: contact with plane (same thing here) */
  double wp_dist_reward;     /* 0.1 quadx, 1.0 fixedwing */
  double wp_yaw_penalty;     /* 0.01 quadx, 0 fixedwing */
  int use_yaw_targets;       /* quadx_waypoints_env.py:40, waypoint_handler.py:85-89,144-156,167-179 */
  double goal_reach_angle;   /* quadx_waypoints_env.py:42 (0.1 rad) */

  /* noise */
  int noise_mode;
  uint64_t seed;

  /* rocket only (drones/rocket.py, abstractions/boosters.py, gimbals.py; models/vehicles/rocket/) */
  int n_links;                           /* links[0] = base, then the URDF's child links in joint order */
  double link_mass[ORC_MAX_LINKS];
  double link_r[ORC_MAX_LINKS][3];       /* link COM in the base frame */
  double link_I[ORC_MAX_LINKS][3];       /* diagonal own inertia (all massive links are axis-aligned) */
  int fueltank_link, booster_link;       /* indices into links[] */
  double total_fuel, max_fuel_rate, fuel_inertia[3], min_thrust, max_thrust, booster_tau, booster_noise;
  int reignitable;
  double gimbal_tau, gimbal_range_rad;
  double finlet_map[4][3];               /* rocket.py:152-159 */
  double starting_fuel_ratio;            /* rocket.py:47 */

  /* wind field (aviary.py:266-285,324-333; base_wind_field.py:10-69): called from update_state with
   * the Aviary's elapsed time and the world positions of the n links it is sampled at (QuadX: the
   * body link, boring_bodies.py:93-96; Fixedwing: the five surface links, lifting_surfaces.py:88-93);
   * writes n x 3 wind velocities. NULL = no wind. */
  void (*wind_fn)(double time, const double* pos /* n x 3 */, int n, double* out /* n x 3 */);
} orc_params;

typedef struct {
  /* Bullet base state */
  double p[3], q[4], v[3], w[3];
  /* derived by update_state (quadx.py:512-535) */
  double w_b[3], rpy[3], v_b[3];
  double surf_v[ORC_MAX_SURF][3];
  double drag_v_b[3]; /* BoringBodies.local_body_velocities: v_b with the wind subtracted (boring_bodies.py:92-111) */
  /* actuators */
  double throttle[4];
  double actuation[ORC_MAX_SURF];
  double pwm[4];
  double cmd[8];
  double setpoint[8];
  /* rocket: boosters.py:119-130 (ignition, fuel ratio; the throttle lives in throttle[0]), gimbals.py:118-122 */
  double fuel_ratio;
  int ignition;
  double gimbal[2];
  /* controllers */
  double pid_I[4][3], pid_E[4][3];
  double zpid_I[2], zpid_E[2];
  int mode;
  int physics_steps;
  int contact_now;    /* this body has contact points after the last stepSimulation (floor, or another drone of its world) */
  int contact_step;   /* contact_array[drone.Id].any() after the last Aviary.step */
  /* shared world (pz_envs: all agents of an env live in ONE Bullet world, ma_quadx_base_env.py:206-241); both stay 0
   * for a drone that is alone in its world */
  int world_contact;  /* len(getContactPoints()) > 0 for the WHOLE world after the last stepSimulation: quadx.py:509 gates every drone's rotational drag on it */
  int peer_contact;   /* this tick's drone-drone verdict for this body, set by orc_world_aviary_step before the tick */
  /* env */
  int step_count, terminated, truncated;
  int info_oob, info_collision, info_complete, num_targets_reached;
  double reward;
  double action[4];
  double past_action[4]; /* MA hover: self.past_actions (ma_quadx_base_env.py:326), survives resets */
  double targets[ORC_MAX_TARGETS][3];
  double yaw_targets[ORC_MAX_TARGETS]; /* waypoint_handler.py:85-89 */
  double yaw_error_scalar;             /* :156 */
  int n_targets_left;
  double new_dist, old_dist;
  /* observation as left by the last compute_state() (quadx_hover_env.py:85-115): the reference
   * returns this cached vector, e.g. the target deltas still include a waypoint reached in the
   * same step (quadx_waypoints_env.py:171-175 precedes :195-199) */
  double obs[48];
  /* rng */
  uint32_t rng_ctr;
  uint32_t reset_key; /* the event counter at this lane's previous env reset: what the NEXT reset's draws are keyed by (orc_env_reset) */
  uint64_t lane_id;
} orc_lane;

/* ---------- parameter sets (numbers copied from the reference's YAML/URDF, cited in .c) ---------- */
void orc_params_quadx(orc_params* P);
void orc_params_fixedwing(orc_params* P);
void orc_params_primitive_drone(orc_params* P);
void orc_params_rocket(orc_params* P);
void orc_params_acrowing(orc_params* P);
/* MAFixedwingDogfightEnv (pz_envs/fixedwing_envs/ma_fixedwing_dogfight_env.py): env-level state of ONE world of A = 2 team_size
 * aircraft; the aircraft themselves are A orc_lane (stepped together by orc_world_aviary_step). */
#define ORC_DF_MAX 8
typedef struct {
  /* constants (:42-60) */
  int A, team_size;
  double damage_per_hit, lethal_distance, lethal_angle, aggressiveness, cooperativeness;
  int sparse_reward;
  double dome;
  int max_steps, env_step_ratio;
  /* assisted_flight (:52, ma_fixedwing_base_env.py:69): 4 = roll, pitch, yaw, thrust commands; 6 = the "raw actuator" action space.
   * With 6 the reference still leaves the Aviary in flight mode 0 (ma_fixedwing_base_env.py:229), which reads setpoint[0:4]
   * (fixedwing.py:246-250): entries 4 and 5 are ignored, and the thrust remap of :300-301 lands on entry 5 -- the thrust
   * command is action[3] as given. Restated as it is. */
  int action_dim;
  /* state */
  int step_count;
  int alive[ORC_DF_MAX];          /* still in self.agents */
  double health[ORC_DF_MAX];      /* float32 in the reference */
  int received_hits[ORC_DF_MAX];
  int inactive[ORC_DF_MAX];       /* dead, on the ground and at rest: dropped from the others' observations (:505-510) */
  double cur_dist[ORC_DF_MAX][ORC_DF_MAX], cur_ang[ORC_DF_MAX][ORC_DF_MAX];
  double prev_dist[ORC_DF_MAX][ORC_DF_MAX], prev_ang[ORC_DF_MAX][ORC_DF_MAX];
  int cur_hit[ORC_DF_MAX][ORC_DF_MAX], in_range[ORC_DF_MAX][ORC_DF_MAX], chasing[ORC_DF_MAX][ORC_DF_MAX];
  double other_att[ORC_DF_MAX][ORC_DF_MAX][12];
  double acc_reward[ORC_DF_MAX];  /* accumulated_rewards / _terminations / _truncations / infos (:651-722) */
  int acc_term[ORC_DF_MAX], acc_trunc[ORC_DF_MAX];
  int info_bits[ORC_DF_MAX];      /* 1 dead, 2 collision, 4 out_of_bounds, 8 team_win (sticky over the episode) */
  double action[ORC_DF_MAX][6], past_action[ORC_DF_MAX][6];
  double obs[ORC_DF_MAX][25 + (ORC_DF_MAX - 1) * 14];
  /* popped by the last step for the agents that were alive */
  double reward[ORC_DF_MAX];
  int terminated[ORC_DF_MAX], truncated[ORC_DF_MAX];
} orc_dogfight;
int orc_sizeof_dogfight(void);
void orc_dogfight_spawn(int team_size, double min_radius, double max_radius, const double* u, double* pos, double* rpy, double* vel);
void orc_dogfight_reset(const orc_params* const* Pl, orc_lane* const* Ll, orc_dogfight* D, uint64_t lane_id0, const double* const* xi_reset);
void orc_dogfight_step(const orc_params* const* Pl, orc_lane* const* Ll, orc_dogfight* D, const double* actions, const double* const* xi);

void orc_task_hover(orc_params* P);
void orc_task_quadx_waypoints(orc_params* P);
void orc_task_fixedwing_waypoints(orc_params* P);
void orc_task_ma_hover(orc_params* P);
void orc_finalize(orc_params* P); /* recompute derived quantities after edits */

/* ---------- components (golden-checked one by one) ---------- */
void orc_pid_step(const double* kp, const double* ki, const double* kd, const double* lim,
                  double period, int n, double* I, double* E, const double* state,
                  const double* setpoint, double* out);
void orc_quadx_mix(const orc_params* P, const double cmd[4], double pwm[4]);
void orc_quadx_control(const orc_params* P, orc_lane* L);
void orc_motors_update(const orc_params* P, double* throttle, const double* pwm, double xi,
                       double thrust[4][3], double torque[4][3]);
void orc_body_drag(const orc_params* P, const double v_b[3], double F[3]);
void orc_surface_aero(const orc_surface* S, double alpha, double actuation, double out_ClCdCM[3]);
void orc_surface_force(const orc_surface* S, const double v_local[3], double actuation,
                       double F[3], double T[3]);
void orc_quat_from_euler(const double rpy[3], double q[4]);
void orc_euler_from_quat(const double q[4], double rpy[3]);
void orc_matrix_from_quat(const double q[4], double R[3][3]);
int orc_contact_plane(const orc_params* P, const double p[3], const double q[4]);
int orc_box_box_overlap(const double ca[3], const double Ra[3][3], const double ha[3],
                        const double cb[3], const double hb[3]);
void orc_rigid_tick(const orc_params* P, double p[3], double q[4], double v[3], double w[3],
                    const double F_b[3], const double tau_b[3]);
/* contact vertices of the body at pose (p, q) that lie at or below the slab's top face: world positions and
 * penetration depths (negative = gap of a speculative contact), in collider / vertex order; returns their number
 * (<= ORC_MAX_CONTACTS) */
#define ORC_MAX_CONTACTS 48 /* same cap as the device code (PF_MAX_CONTACTS): vertices past it are ignored */
int orc_contact_points(const orc_params* P, const double p[3], const double q[4], double pts[][3], double depth[]);

/* ---------- RNG (same integer stream as the device) ---------- */
#define ORC_PHILOX_ROUNDS 10 /* == PF_PHILOX_ROUNDS (pyflyt_amd/csrc/uav_device.hpp) */
void orc_philox4x32(uint64_t key, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t out[4]);
void orc_philox4x32_r(uint64_t key, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, int rounds, uint32_t out[4]);
void orc_normal8(uint64_t key, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, double z[8]);
void orc_normal4(uint64_t key, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, double z[4]);
void orc_uniform4(uint64_t key, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, double u[4]);

/* ---------- lane level ---------- */
void orc_update_state(const orc_params* P, orc_lane* L);
void orc_set_mode(const orc_params* P, orc_lane* L, int mode);
/* Aviary.reset + drone.reset + update_state (aviary.py:218-312) */
void orc_aviary_reset(const orc_params* P, orc_lane* L, uint64_t lane_id);
/* one Aviary.step (aviary.py:480-531); xi: ticks_per_control normals (ORC_NOISE_INJECT) or NULL */
void orc_aviary_step(const orc_params* P, orc_lane* L, const double* xi,
                     uint32_t rng_call_base, uint32_t rng_stream);
/* ---------- shared world: A drones in ONE world (PettingZoo envs) ----------
 * drone-drone hits enter contact_array[drone.Id] (ma_quadx_hover_env.py:181: any contact of the agent's body ends its
 * episode) and the rotational-drag gate of EVERY drone looks at the contact points of the whole world (quadx.py:509).
 * Detection only between drones (box colliders, 15-axis test in the other box's frame): no drone-drone impulses.
 * Pl / Ll: arrays of A pointers (each drone has its own parameter block: spawn pose); xi: A pointers or NULL. */
void orc_world_aviary_step(const orc_params* const* Pl, orc_lane* const* Ll, int A, const double* const* xi,
                           uint32_t rng_call_base, uint32_t rng_stream);
void orc_world_env_reset(const orc_params* const* Pl, orc_lane* const* Ll, int A, uint64_t lane_id0, const double* const* xi_reset);
void orc_world_env_step(const orc_params* const* Pl, orc_lane* const* Ll, int A, const double* actions /* [A][4] */,
                        const double* const* xi);
/* env.reset(): begin_reset + waypoint sampling + end_reset (quadx_base_env.py:149-212);
 * xi_reset: settle_steps*ticks_per_control normals, u_targets: 3*num_targets uniforms (inject mode;
 * 4*num_targets with use_yaw_targets: theta | phi | dist | yaw, the reference's draw order) */
void orc_env_reset(const orc_params* P, orc_lane* L, uint64_t lane_id, const double* xi_reset,
                   const double* u_targets);
/* env.step(action) (quadx_base_env.py:269-301); xi: env_step_ratio*ticks_per_control normals */
void orc_env_step(const orc_params* P, orc_lane* L, const double action[4], const double* xi);
int orc_obs_dim(const orc_params* P);
/* flattened obs: attitude (+ num_targets*3 zero-padded deltas for waypoint tasks) */
void orc_env_obs(const orc_params* P, const orc_lane* L, double* obs);

/* ---------- batch level (OpenMP over lanes) ---------- */
/* autoreset: 0 none, 1 next-step (gymnasium default), 2 same-step */
void orc_env_reset_batch(const orc_params* P, orc_lane* L, int n, uint64_t lane0,
                         const uint8_t* mask, const double* xi_reset, const double* u_targets);
void orc_env_step_batch(const orc_params* P, orc_lane* L, int n, const float* actions,
                        const double* xi, const double* xi_reset, const double* u_targets,
                        int autoreset, double* obs, double* reward, uint8_t* term, uint8_t* trunc,
                        double* final_obs);
int orc_sizeof_lane(void);
/* diagnostics (tests/tools/solve_stats.py): solves, sweeps and cap hits by contact count (0..4+); the first call switches the counters on */
void orc_debug_solve_stats(long long* out, int clear);
int orc_sizeof_params(void);
int orc_num_threads(void);

#ifdef __cplusplus
}
#endif
#endif
