"""Analytic known-answer cases for the restated Bullet free-body tick (SURVEY.md section 8(c)(ii), Appendix C
"integrator" row): closed forms that hold for the semi-implicit Euler step
    v += a dt ; w += I^-1 (tau - w x I w) dt ; clamp +-max_coord_vel ; p += v dt ; q <- exp(w dt / 2) q
independently of any restatement. Shared by tests/test_oracle_kat.py (fp64 oracle, CPU) and
tests/test_gpu_kat.py (HIP kernels, through the C ABI), so both sides are held to the SAME closed forms --
the only evidence for SURVEY rows 10-12 that does not go through one of our own restatements.

Constants are the cf2x numbers (cf2x.urdf:13-14, cf2x.yaml:1-6, aviary.py:79,226)."""
import numpy as np

DT = 1.0 / 240.0
G = 9.81
MASS = 0.027
I_DIAG = np.array([1.4e-5, 1.4e-5, 2.17e-5])
MOTOR_LAG = DT / 0.01          # dt / tau = 0.41667 (motors.py:131)
TOTAL_THRUST = 2.0             # cf2x.yaml:2
HOVER_THROTTLE_SQ = MASS * G / TOTAL_THRUST  # 0.132435
VMAX = 100.0                   # btMultiBody::m_maxCoordinateVelocity
CONTACT_SLOP = 1e-5            # the contact model's allowed overlap (PyBullet's m_linearSlop; pyflyt_amd/params.py: WORLD)


def free_fall_z(z0, n):
    """z_n = z0 - g dt^2 n (n + 1) / 2 ; vz_n = -g dt n."""
    n = np.asarray(n, dtype=np.float64)
    return z0 - G * DT * DT * n * (n + 1) / 2.0, -G * DT * n


def const_torque_principal(tau, axis, n):
    """Constant torque about a principal axis from rest: w_n = n dt tau / I (gyro term vanishes: w || I w),
    rotation angle theta_n = dt * sum_k w_k = dt^2 tau / I * n (n + 1) / 2."""
    w = n * DT * tau / I_DIAG[axis]
    theta = DT * DT * tau / I_DIAG[axis] * n * (n + 1) / 2.0
    return w, theta


def motor_lag(p, n):
    """throttle_n = p (1 - (1 - dt/tau)^n), noise off (motors.py:131)."""
    return p * (1.0 - (1.0 - MOTOR_LAG) ** np.asarray(n, dtype=np.float64))


def gimbal_yaw(roll, yaw, sign):
    """getEulerFromQuaternion(getQuaternionFromEuler(roll, +-pi/2, yaw)) in the |sarg| >= 0.99999 branch:
    roll_out = 0, pitch_out = +-pi/2, yaw_out = 2 atan2(-+x, +-y) = yaw -+ roll (wrapped to (-pi, pi])."""
    y = yaw - sign * roll
    return (y + np.pi) % (2 * np.pi) - np.pi


def dogfight_tail_chase_expectations(damage, aggressiveness, cooperativeness, env_step_ratio=4):
    """1 v 1 tail chase with the quarry permanently in the hunter's cone (sparse reward): per env step the hit count after it, the
    hunter's and the quarry's popped reward, and whether the episode ends. One hit per update: the reset's update (its reward is
    popped by the first step), then env_step_ratio per step; health = 1 - damage * hits in float32 as the reference keeps it; dead
    at health <= 1e-3; the update that kills overrides the hunter's accumulated reward with 300 (team win) -- the quarry's
    accumulated penalties stay."""
    import numpy as np

    out, hits, acc_h, acc_q = [], 1, 20.0 + cooperativeness, -20.0 * (1.0 - aggressiveness)  # the reset's update
    health = np.float32(1.0) - np.float32(damage * 1)
    done = False
    while not done:
        for _ in range(env_step_ratio):
            hits += 1
            health = np.float32(np.float64(health) - damage)
            health = max(health, np.float32(0.0))
            acc_h += 20.0 + cooperativeness
            acc_q += -20.0 * (1.0 - aggressiveness)
            if health <= 1e-3:
                done = True
                acc_h = 300.0
            # (the reference keeps accumulating through the rest of the step; a dead quarry is still "hit")
        out.append((hits, acc_h, acc_q, done))
        acc_h, acc_q = 0.0, 0.0
    return out
