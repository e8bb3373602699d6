"""fake_bullet.py -- TEST INFRASTRUCTURE ONLY (used by tests/golden/gen_goldens.py, run only in the
build container where /root/reference exists).

A numpy/float64 stand-in for the slice of the PyBullet API that PyFlyt's hot path calls
("lower boundary"): enough to let the *reference's own* `Aviary`, `QuadX`,
`Fixedwing` and gym env classes run unmodified, so that golden trajectories can be captured with
the reference's real PyFlyt-side arithmetic.

It is an independent second restatement of Bullet's free-multibody tick [BULLET-FROM-MEMORY]:
where oracle/uav_oracle.c integrates the composite body about its centre of mass, this file
follows Bullet's own structure -- per-link spatial bias forces and inertias accumulated at the
base origin and a 6x6 solve (btMultiBody::computeAccelerationsArticulatedBodyAlgorithmMultiDof
with zero-DoF children). Agreement of the two to ~1e-12 is checked in tests/test_oracle_golden.py.

This is NOT PyBullet and pins nothing about real Bullet; "parity unpinned" at that boundary.
"""
from __future__ import annotations

import math
import os
import xml.etree.ElementTree as ET

import numpy as np


def _vec(s, n=3):
    v = [float(x) for x in s.split()]
    assert len(v) == n
    return np.array(v)


def quat_from_euler(rpy):
    phi, the, psi = rpy[0] / 2.0, rpy[1] / 2.0, rpy[2] / 2.0
    q = np.array(
        [
            math.sin(phi) * math.cos(the) * math.cos(psi) - math.cos(phi) * math.sin(the) * math.sin(psi),
            math.cos(phi) * math.sin(the) * math.cos(psi) + math.sin(phi) * math.cos(the) * math.sin(psi),
            math.cos(phi) * math.cos(the) * math.sin(psi) - math.sin(phi) * math.sin(the) * math.cos(psi),
            math.cos(phi) * math.cos(the) * math.cos(psi) + math.sin(phi) * math.sin(the) * math.sin(psi),
        ]
    )
    return q / math.sqrt(float(q @ q))


def euler_from_quat(q):
    sqx, sqy, sqz, squ = q[0] * q[0], q[1] * q[1], q[2] * q[2], q[3] * q[3]
    sarg = -2.0 * (q[0] * q[2] - q[3] * q[1]) / (sqx + sqy + sqz + squ)
    if sarg <= -0.99999:
        return (0.0, -0.5 * math.pi, 2.0 * math.atan2(q[0], -q[1]))
    if sarg >= 0.99999:
        return (0.0, 0.5 * math.pi, 2.0 * math.atan2(-q[0], q[1]))
    return (
        math.atan2(2.0 * (q[1] * q[2] + q[3] * q[0]), squ - sqx - sqy + sqz),
        math.asin(sarg),
        math.atan2(2.0 * (q[0] * q[1] + q[3] * q[2]), squ + sqx - sqy - sqz),
    )


def matrix_from_quat(q):
    d = float(np.dot(q, q))
    s = 2.0 / d
    xs, ys, zs = q[0] * s, q[1] * s, q[2] * s
    wx, wy, wz = q[3] * xs, q[3] * ys, q[3] * zs
    xx, xy, xz = q[0] * xs, q[0] * ys, q[0] * zs
    yy, yz, zz = q[1] * ys, q[1] * zs, q[2] * zs
    return np.array(
        [
            [1.0 - (yy + zz), xy - wz, xz + wy],
            [xy + wz, 1.0 - (xx + zz), yz - wx],
            [xz - wy, yz + wx, 1.0 - (xx + yy)],
        ]
    )


def _skew(r):
    return np.array([[0, -r[2], r[1]], [r[2], 0, -r[0]], [-r[1], r[0], 0]], dtype=np.float64)


class _Link:
    def __init__(self, name):
        self.name = name
        self.mass = 0.0
        self.inertia = np.zeros((3, 3))
        self.inertial_origin = np.zeros(3)
        self.joint_origin = np.zeros(3)
        self.boxes = []  # (centre in link frame, half extents)
        self.cyls = []   # (centre in link frame, radius, half length), axis = link z
        self.rot = np.eye(3)  # link frame -> base frame (joint rpy; only massless links may be rotated)


class _Body:
    """A free (or fixed) base with rigidly attached child links; all frames axis-aligned."""

    def __init__(self, links, fixed, pos, quat, scale=1.0):
        self.fixed = fixed
        self.links = links  # links[0] = base, children in joint order (link index = i-1)
        self.p = np.array(pos, dtype=np.float64)
        self.q = np.array(quat, dtype=np.float64)
        self.v = np.zeros(3)
        self.w = np.zeros(3)
        self.scale = scale
        # COM offset of each link in the base frame
        self.r = [l.joint_origin + l.inertial_origin for l in links]
        self.force = [np.zeros(3) for _ in links]  # world-frame, at the link COM
        self.torque = [np.zeros(3) for _ in links]
        self.use_gyro_term = True
        self.max_coord_vel = 100.0

    def clear_forces(self):
        for f in self.force:
            f[:] = 0.0
        for t in self.torque:
            t[:] = 0.0

    def world_boxes(self):
        R = matrix_from_quat(self.q)
        out = []
        for l in self.links:
            for c, h in l.boxes:
                centre = self.p + R @ ((l.joint_origin + l.rot @ c) * self.scale)
                out.append((centre, R @ l.rot, h * self.scale))
        return out

    def bound_radius(self):
        """Largest distance of a collider vertex from the base origin (a pruning radius only)."""
        return max(float(np.linalg.norm(v)) for v in self.contact_vertices())

    def contact_vertices(self):
        """Body-frame positions (relative to the base origin) of the contact vertices, in URDF link order, each link's
        boxes (8 corners: x sign fastest) then cylinders (end disc -z then +z, 8 rim points at 45 degree steps from the
        link x axis) -- the order the contact sweep runs in."""
        out = []
        c45 = [math.cos(j * math.pi / 4.0) for j in range(8)]
        s45 = [math.sin(j * math.pi / 4.0) for j in range(8)]
        for l in self.links:
            for c, h in l.boxes:
                for i in range(8):
                    loc = np.array([h[0] if i & 1 else -h[0], h[1] if i & 2 else -h[1], h[2] if i & 4 else -h[2]])
                    out.append((l.joint_origin + l.rot @ (c + loc)) * self.scale)
            for c, rad, hl in l.cyls:
                for e in (-1.0, 1.0):
                    for j in range(8):
                        loc = np.array([rad * c45[j], rad * s45[j], e * hl])
                        out.append((l.joint_origin + l.rot @ (c + loc)) * self.scale)
        return out

    def collider_vertices(self):
        """The same vertices collider by collider: [(kind, link rotation, [body-frame vertex, ...]), ...], kind 0 box / 1 cylinder."""
        out = []
        c45 = [math.cos(j * math.pi / 4.0) for j in range(8)]
        s45 = [math.sin(j * math.pi / 4.0) for j in range(8)]
        for l in self.links:
            for c, h in l.boxes:
                vs = []
                for i in range(8):
                    loc = np.array([h[0] if i & 1 else -h[0], h[1] if i & 2 else -h[1], h[2] if i & 4 else -h[2]])
                    vs.append((l.joint_origin + l.rot @ (c + loc)) * self.scale)
                out.append((0, l.rot, vs))
            for c, rad, hl in l.cyls:
                vs = []
                for e in (-1.0, 1.0):
                    for j in range(8):
                        loc = np.array([rad * c45[j], rad * s45[j], e * hl])
                        vs.append((l.joint_origin + l.rot @ (c + loc)) * self.scale)
                out.append((1, l.rot, vs))
        return out

    def world_cyls(self):
        R = matrix_from_quat(self.q)
        out = []
        for l in self.links:
            for c, rad, hl in l.cyls:
                centre = self.p + R @ ((l.joint_origin + l.rot @ c) * self.scale)
                out.append((centre, (R @ l.rot)[:, 2].copy(), rad * self.scale, hl * self.scale))
        return out


def _cyl_aabb_overlap(c, axis, radius, half_len, cb, hb):
    """Cylinder (centre c, unit axis, radius, half length) against a world-axis-aligned box:
    separating-axis test on the box's three face normals with the cylinder's exact support extent
    h|a_k| + r sqrt(1 - a_k^2) along each. Exact whenever the closest feature of the box is a face --
    for the 30 m ground slab, everywhere except within one prop radius of its rim. [BULLET-FROM-MEMORY]:
    Bullet runs GJK/EPA on the convex pair; the verdict compared here is penetration >= 0, as for boxes."""
    for k in range(3):
        ext = half_len * abs(axis[k]) + radius * np.sqrt(max(0.0, 1.0 - axis[k] * axis[k]))
        if abs(c[k] - cb[k]) - (hb[k] + ext) > 0.0:
            return False
    return True


def _box_box_overlap(ca, Ra, ha, cb, hb):
    """btBoxBoxDetector / dBoxBox2 separating-axis verdict; box b is world-axis-aligned."""
    t = ca - cb
    Q = np.abs(Ra)
    for i in range(3):
        if abs(t[i]) - (hb[i] + Q[i] @ ha) > 0.0:
            return False
    for j in range(3):
        if abs(t @ Ra[:, j]) - (ha[j] + Q[:, j] @ hb) > 0.0:
            return False
    Q = Q + 1e-5
    for i in range(3):
        i1, i2 = (i + 1) % 3, (i + 2) % 3
        for j in range(3):
            j1, j2 = (j + 1) % 3, (j + 2) % 3
            expr1 = t[i2] * Ra[i1, j] - t[i1] * Ra[i2, j]
            rad = hb[i1] * Q[i2, j] + hb[i2] * Q[i1, j] + ha[j1] * Q[i, j2] + ha[j2] * Q[i, j1]
            if abs(expr1) - rad > 2.220446049250313e-16:
                return False
    return True


def parse_urdf(path):
    text = open(path).read()
    end = text.find("</robot>")  # rocket.urdf closes <robot> twice; Bullet's parser stops at the first
    root = ET.fromstring(text[: end + len("</robot>")] if end >= 0 else text)
    links = {}
    order = []
    for le in root.findall("link"):
        l = _Link(le.get("name"))
        ine = le.find("inertial")
        if ine is not None:
            o = ine.find("origin")
            if o is not None:
                assert np.allclose(_vec(o.get("rpy", "0 0 0")), 0.0)
                l.inertial_origin = _vec(o.get("xyz", "0 0 0"))
            l.mass = float(ine.find("mass").get("value"))
            it = ine.find("inertia")
            ixx, iyy, izz = (float(it.get(k)) for k in ("ixx", "iyy", "izz"))
            ixy, ixz, iyz = (float(it.get(k)) for k in ("ixy", "ixz", "iyz"))
            l.inertia = np.array([[ixx, ixy, ixz], [ixy, iyy, iyz], [ixz, iyz, izz]])
        for ce in le.findall("collision"):
            o = ce.find("origin")
            c = np.zeros(3)
            if o is not None:
                assert np.allclose(_vec(o.get("rpy", "0 0 0")), 0.0)
                c = _vec(o.get("xyz", "0 0 0"))
            box = ce.find("geometry").find("box")
            if box is not None:
                l.boxes.append((c, 0.5 * _vec(box.get("size"))))
            cyl = ce.find("geometry").find("cylinder")
            if cyl is not None:  # URDF cylinders stand along the link z axis
                l.cyls.append((c, float(cyl.get("radius")), 0.5 * float(cyl.get("length"))))
        links[l.name] = l
        order.append(l.name)
    children = set()
    joint_children = []
    for je in root.findall("joint"):
        assert je.get("type") == "fixed", "only fixed joints are on the hot path"
        parent = je.find("parent").get("link")
        child = je.find("child").get("link")
        o = je.find("origin")
        if o is not None:
            rpy = _vec(o.get("rpy", "0 0 0"))
            if not np.allclose(rpy, 0.0):
                # rotated child frames are supported for massless links only (rocket.urdf legs): the
                # rotation then matters for their collision shapes alone
                assert links[child].mass == 0.0 and not links[child].inertia.any()
                links[child].rot = matrix_from_quat(quat_from_euler(rpy))
            links[child].joint_origin = _vec(o.get("xyz", "0 0 0"))
        children.add(child)
        joint_children.append((parent, child))
    base = [n for n in order if n not in children]
    assert len(base) == 1
    for parent, _ in joint_children:
        assert parent == base[0], "only one level of fixed children is supported"
    return [links[base[0]]] + [links[c] for _, c in joint_children]


class BulletClient:
    """Duck-type of pybullet_utils.bullet_client.BulletClient for the hot-path method set."""

    DEFAULT_CONTACT_RESPONSE = True  # (tests/golden/gen_goldens.py switches it off while it records the env-level fixtures)
    DEFAULT_PAIR_RESPONSE = True     # (... and this one for the control recording of the mid-air collision)
    DIRECT = 2
    GUI = 1
    LINK_FRAME = 1
    WORLD_FRAME = 2
    URDF_USE_INERTIA_FROM_FILE = 2

    def __init__(self, connection_mode=None):
        self._bodies = {}
        self._next_id = 0
        self._gravity = np.zeros(3)
        self._dt = 1.0 / 240.0
        self._contacts = []
        self._search = ""
        self.use_gyro_term = True
        # contact response against fixed bodies' top faces (the ground slab): see _solve_contacts
        self.contact_response = BulletClient.DEFAULT_CONTACT_RESPONSE
        # (the defaults and what each is believed to restate: oracle/uav_oracle.h, orc_world)
        self.contact_restitution, self.contact_friction, self.contact_erp, self.contact_iters = 0.0, 0.5, 0.2, 50
        self.contact_margin, self.contact_slop = 0.0, 1e-5
        self.contact_report_distance = 0.0
        self.contact_residual_threshold = 1e-7
        self.contact_manifold_points = 4
        self.contact_break_distance = 0.02  # points a body held after the previous tick persist up to this gap
        self._persisted = set()  # ids of the free bodies that held contact points after the previous tick
        self.pair_response = BulletClient.DEFAULT_PAIR_RESPONSE  # impulses between free bodies (oracle/uav_oracle.h: orc_world.pair_response)

    # ------------------------------------------------------------ no-ops
    def setAdditionalSearchPath(self, path):
        self._search = path

    def addUserDebugText(self, **kwargs):
        return 0

    def resetDebugVisualizerCamera(self, **kwargs):
        pass

    def disconnect(self):
        pass

    def changeDynamics(self, body, link, **kwargs):
        # PyFlyt zeroes the artificial damping (base_drone.py:301-304; the tick below has none) and, for
        # fuel tanks, rewrites a link's mass and diagonal inertia every tick (boosters.py:193-198)
        assert set(kwargs) <= {"linearDamping", "angularDamping", "mass", "localInertiaDiagonal"}
        assert all(kwargs[k] == 0.0 for k in ("linearDamping", "angularDamping") if k in kwargs)
        if "mass" in kwargs or "localInertiaDiagonal" in kwargs:
            l = self._bodies[body].links[int(link) + 1]
            if "mass" in kwargs:
                l.mass = float(kwargs["mass"])
            if "localInertiaDiagonal" in kwargs:
                l.inertia = np.diag(np.asarray(kwargs["localInertiaDiagonal"], dtype=np.float64))

    # ------------------------------------------------------------ world
    def resetSimulation(self):
        self._bodies = {}
        self._next_id = 0
        self._contacts = []

    def setGravity(self, x, y, z):
        self._gravity = np.array([x, y, z], dtype=np.float64)

    def loadURDF(self, fileName, basePosition=None, baseOrientation=None, useFixedBase=False,
                 globalScaling=1.0, flags=0):
        pos = np.zeros(3) if basePosition is None else np.array(basePosition, dtype=np.float64)
        quat = np.array([0, 0, 0, 1.0]) if baseOrientation is None else np.array(baseOrientation, dtype=np.float64)
        if os.path.basename(fileName) == "plane.urdf" and not os.path.isabs(fileName):
            # pybullet_data/plane.urdf [BULLET-FROM-MEMORY]: collision box 30x30x10 at z=-5
            l = _Link("planeLink")
            l.boxes.append((np.array([0.0, 0.0, -5.0]), np.array([15.0, 15.0, 5.0])))
            body = _Body([l], True, pos, quat, globalScaling)
        else:
            body = _Body(parse_urdf(fileName), bool(useFixedBase), pos, quat, globalScaling)
            assert flags & self.URDF_USE_INERTIA_FROM_FILE
        body.use_gyro_term = self.use_gyro_term
        bid = self._next_id
        self._next_id += 1
        self._bodies[bid] = body
        return bid

    def getNumBodies(self):
        return len(self._bodies)

    def getBodyUniqueId(self, i):
        return sorted(self._bodies)[i]

    def getNumJoints(self, body):
        return len(self._bodies[body].links) - 1

    # ------------------------------------------------------------ state access
    def resetBasePositionAndOrientation(self, body, pos, orn):
        b = self._bodies[body]
        b.p = np.array(pos, dtype=np.float64)
        b.q = np.array(orn, dtype=np.float64)
        b.v = np.zeros(3)
        b.w = np.zeros(3)

    def resetBaseVelocity(self, body, linearVelocity=None, angularVelocity=None):
        b = self._bodies[body]
        if linearVelocity is not None:
            b.v = np.array(linearVelocity, dtype=np.float64)
        if angularVelocity is not None:
            b.w = np.array(angularVelocity, dtype=np.float64)

    def getBasePositionAndOrientation(self, body):
        b = self._bodies[body]
        return tuple(b.p), tuple(b.q)

    def getBaseVelocity(self, body):
        b = self._bodies[body]
        return tuple(b.v), tuple(b.w)

    @staticmethod
    def getMatrixFromQuaternion(q):
        return tuple(matrix_from_quat(np.asarray(q, dtype=np.float64)).reshape(-1))

    @staticmethod
    def getEulerFromQuaternion(q):
        return euler_from_quat(np.asarray(q, dtype=np.float64))

    @staticmethod
    def getQuaternionFromEuler(rpy):
        return tuple(quat_from_euler(np.asarray(rpy, dtype=np.float64)))

    def getLinkStates(self, body, ids, computeLinkVelocity=False):
        b = self._bodies[body]
        R = matrix_from_quat(b.q)
        out = []
        for i in ids:
            r = R @ b.r[int(i) + 1]
            pos = b.p + r
            vel = b.v + np.cross(b.w, r)
            out.append((tuple(pos), tuple(b.q), (0, 0, 0), (0, 0, 0, 1), tuple(pos), tuple(b.q), tuple(vel), tuple(b.w)))
        return tuple(out)

    # ------------------------------------------------------------ forces
    def applyExternalForce(self, body, link, force, pos, frame):
        b = self._bodies[body]
        assert frame == self.LINK_FRAME and np.allclose(pos, 0.0)
        R = matrix_from_quat(b.q)
        b.force[int(link) + 1] += R @ np.asarray(force, dtype=np.float64)

    def applyExternalTorque(self, body, link, torque, frame):
        b = self._bodies[body]
        assert frame == self.LINK_FRAME
        R = matrix_from_quat(b.q)
        b.torque[int(link) + 1] += R @ np.asarray(torque, dtype=np.float64)

    def getContactPoints(self, *args, **kwargs):
        return list(self._contacts)

    def _reach(self, persisted, fresh):
        """How far above a face a vertex may be and still count (as a constraint row: fresh = contact_margin; for the
        report: fresh = contact_report_distance) -- a body that held contact points after the previous tick keeps them up to
        the contact breaking distance."""
        return self.contact_break_distance if persisted else fresh

    def _solve_contacts(self, b, I6, R, persisted=False):
        """Contact response of free body `b` against the ground slab, the SAME named-parameter model as
        oracle/uav_oracle.c:contact_solve but formulated independently: impulses act on the base twist (body frame,
        [angular; linear] at the base ORIGIN) through the 6x6 spatial inertia the tick already assembled -- no centre of
        mass, no 3x3 inertia. Returns the deepest penetration."""
        slabs = [bx for f in self._bodies.values() if f.fixed for bx in f.world_boxes()]
        margin = self._reach(persisted, self.contact_margin)
        pts = []
        for kind, lrot, verts in b.collider_vertices():
            cand = list(range(len(verts)))
            if kind == 0 and self.contact_manifold_points < 8:
                # manifold reduction: the four vertices of the box face that looks down the most (the incident face of a
                # box-box face contact against the slab's top face); the first axis on a tie
                zrow = (R @ lrot)[2]
                a = int(np.argmax(np.abs(zrow)))
                up = zrow[a] < 0.0  # the face on the +a side looks down when the axis itself points down
                cand = [i for i in cand if bool(i & (1 << a)) == up]
            for i in cand:
                if len(pts) >= 48:  # the device code's PF_MAX_CONTACTS: vertices past it are ignored
                    break
                rb = verts[i]
                x = b.p + R @ rb
                for cb, Rb, hb in slabs:
                    if x[2] <= cb[2] + hb[2] + margin and x[2] >= cb[2] - hb[2] and abs(x[0] - cb[0]) <= hb[0] and abs(x[1] - cb[1]) <= hb[1]:
                        pts.append((rb, (cb[2] + hb[2]) - x[2]))
                        break
        if not pts:
            return 0.0
        I6inv = np.linalg.inv(I6)
        tw = np.concatenate([R.T @ b.w, R.T @ b.v])  # body-frame twist
        dirs = [R.T @ np.array(d) for d in ((0.0, 0.0, 1.0), (1.0, 0.0, 0.0), (0.0, 1.0, 0.0))]
        lam = np.zeros((len(pts), 3))
        jac = [[np.concatenate([np.cross(rb, d), d]) for d in dirs] for rb, _ in pts]
        vn0 = [float(j[0] @ tw) for j in jac]
        for _ in range(self.contact_iters):
            res2 = 0.0  # the sweep's largest squared row-velocity change (the solver's least-squares residual)
            for c in range(len(pts)):
                for d in range(3):
                    j = jac[c][d]
                    resp = I6inv @ j
                    k = float(j @ resp)
                    target = 0.0
                    if d == 0:
                        depth = pts[c][1]
                        target = ((depth - self.contact_slop) / self._dt if depth < self.contact_slop
                                  else (-self.contact_restitution * vn0[c] if vn0[c] < 0.0 else 0.0))
                    dl = (target - float(j @ tw)) / k
                    if d == 0:
                        new = max(lam[c, 0] + dl, 0.0)
                    else:
                        lim = self.contact_friction * lam[c, 0]
                        new = min(max(lam[c, d] + dl, -lim), lim)
                    dl = new - lam[c, d]
                    lam[c, d] = new
                    tw = tw + dl * resp
                    res2 = max(res2, (dl * k) ** 2)
            if res2 <= self.contact_residual_threshold:
                break
        b.w = R @ tw[:3]
        b.v = R @ tw[3:]
        return max(0.0, max(d for _, d in pts) - self.contact_slop)

    def _solve_pair_contacts(self, pre):
        """Contact response BETWEEN free bodies (the PettingZoo envs put every agent's drone in one world), the SAME
        named-parameter model as oracle/uav_oracle.c:pair_stage but formulated independently: impulses act on the two base
        twists (body frames, [angular; linear] at the base ORIGINS) through the 6x6 spatial inertias -- no centres of mass,
        no 3x3 inertias. Vertex-in-box contacts (margin included), the least-penetrated face of the other box as the normal,
        btPlaneSpace1 tangents, projected Gauss-Seidel in contact order, friction clamp friction^2 x normal impulse.
        Returns {body id: translation} -- half of erp x (deepest pair penetration - slop) along that contact's normal."""
        ids = sorted(pre)
        pts = []
        for ia in ids:
            A = self._bodies[ia]
            for ib in ids:
                if ib == ia:
                    continue
                B = self._bodies[ib]
                margin = self._reach(ia in self._persisted or ib in self._persisted, self.contact_margin)
                if float((A.p - B.p) @ (A.p - B.p)) > (A.bound_radius() + B.bound_radius() + 2.0 * margin) ** 2:
                    continue
                for ca, Ra, ha in A.world_boxes():
                    for cb, Rb, hb in B.world_boxes():
                        for i in range(8):
                            x = ca + Ra @ np.array([ha[0] if i & 1 else -ha[0], ha[1] if i & 2 else -ha[1], ha[2] if i & 4 else -ha[2]])
                            loc = Rb.T @ (x - cb)
                            pen = hb - np.abs(loc)
                            ks = int(np.argmin(pen))  # (the first axis on a tie)
                            if pen[ks] < -margin or len(pts) >= 16:
                                continue
                            n = (1.0 if loc[ks] >= 0.0 else -1.0) * Rb[:, ks]
                            pts.append((ia, ib, x, n, float(pen[ks])))
        if not pts:
            return {}
        tw = {}
        inv = {}
        for bid in ids:
            b = self._bodies[bid]
            I6, R = pre[bid]
            tw[bid] = np.concatenate([R.T @ b.w, R.T @ b.v])
            inv[bid] = np.linalg.inv(I6)

        def plane_space(n):
            if abs(n[2]) > 0.7071067811865475244:
                a = n[1] * n[1] + n[2] * n[2]
                k = 1.0 / math.sqrt(a)
                p = np.array([0.0, -n[2] * k, n[1] * k])
                q = np.array([a * k, -n[0] * p[2], n[0] * p[1]])
            else:
                a = n[0] * n[0] + n[1] * n[1]
                k = 1.0 / math.sqrt(a)
                p = np.array([-n[1] * k, n[0] * k, 0.0])
                q = np.array([-n[2] * p[1], n[2] * p[0], a * k])
            return p, q

        rows = []
        for ia, ib, x, n, depth in pts:
            A, B = self._bodies[ia], self._bodies[ib]
            Ra, Rb = pre[ia][1], pre[ib][1]
            t1, t2 = plane_space(n)
            ja, jb = [], []
            for d in (n, t1, t2):
                da, db = Ra.T @ d, Rb.T @ d
                ja.append(np.concatenate([np.cross(Ra.T @ (x - A.p), da), da]))
                jb.append(np.concatenate([np.cross(Rb.T @ (x - B.p), db), db]))
            rows.append((ia, ib, ja, jb, depth))
        vn0 = [float(ja[0] @ tw[ia] - jb[0] @ tw[ib]) for ia, ib, ja, jb, _ in rows]
        lam = np.zeros((len(rows), 3))
        mu = self.contact_friction * self.contact_friction
        for _ in range(self.contact_iters):
            res2 = 0.0
            for c, (ia, ib, ja, jb, depth) in enumerate(rows):
                for d in range(3):
                    ra, rb = inv[ia] @ ja[d], inv[ib] @ jb[d]
                    k = float(ja[d] @ ra + jb[d] @ rb)
                    target = 0.0
                    if d == 0:
                        target = ((depth - self.contact_slop) / self._dt if depth < self.contact_slop
                                  else (-self.contact_restitution * vn0[c] if vn0[c] < 0.0 else 0.0))
                    new = lam[c, d] + (target - float(ja[d] @ tw[ia] - jb[d] @ tw[ib])) / k
                    if d == 0:
                        new = max(new, 0.0)
                    else:
                        lim = mu * lam[c, 0]
                        new = min(max(new, -lim), lim)
                    dl = new - lam[c, d]
                    lam[c, d] = new
                    tw[ia] = tw[ia] + dl * ra
                    tw[ib] = tw[ib] - dl * rb
                    res2 = max(res2, (dl * k) ** 2)
            if res2 <= self.contact_residual_threshold:
                break
        for bid in ids:
            b = self._bodies[bid]
            R = pre[bid][1]
            b.w = R @ tw[bid][:3]
            b.v = R @ tw[bid][3:]
        shift, best = {}, {}
        for ia, ib, x, n, depth in pts:
            e = depth - self.contact_slop
            if e <= 0.0:
                continue
            if e > best.get(ia, 0.0):
                best[ia] = e
                shift[ia] = 0.5 * self.contact_erp * e * n
            if e > best.get(ib, 0.0):
                best[ib] = e
                shift[ib] = -0.5 * self.contact_erp * e * n
        return shift

    # ------------------------------------------------------------ the tick
    def stepSimulation(self):
        dt = self._dt
        # 1) collision detection at the pre-integration pose
        self._persisted = {b for c in self._contacts for b in (c[1], c[2]) if not self._bodies[b].fixed}
        self._contacts = []
        ids = sorted(self._bodies)
        for ia in ids:
            for ib in ids:
                if ib <= ia:
                    continue
                A, B = self._bodies[ia], self._bodies[ib]
                if A.fixed == B.fixed:
                    if A.fixed:
                        continue
                    # two free bodies (the PettingZoo envs put every agent's drone in one world): box colliders, the 15-axis
                    # verdict evaluated in the second box's frame. Detection only -- no impulses between drones.
                    hit = False
                    for ca, Ra, ha in A.world_boxes():
                        for cb, Rb, hb in B.world_boxes():
                            rd = self._reach(ia in self._persisted or ib in self._persisted, self.contact_report_distance)
                            if _box_box_overlap(Rb.T @ (ca - cb), Rb.T @ Ra, ha, np.zeros(3), hb + rd):
                                hit = True
                    if hit:
                        self._contacts.append((0, ia, ib, -1, -1))
                    continue
                fixed, free = (A, B) if A.fixed else (B, A)
                idf, idr = (ia, ib) if A.fixed else (ib, ia)
                hit = False
                for cb, Rb, hb in fixed.world_boxes():
                    assert np.allclose(Rb, np.eye(3))
                    hb = hb + self._reach(idr in self._persisted, self.contact_report_distance)  # a contact is reported from this gap on
                    for ca, Ra, ha in free.world_boxes():
                        if _box_box_overlap(ca, Ra, ha, cb, hb):
                            hit = True
                    for cc, ax, rad, hl in free.world_cyls():
                        if _cyl_aabb_overlap(cc, ax, rad, hl, cb, hb):
                            hit = True
                if hit:
                    self._contacts.append((0, idf, idr, -1, -1))
        # 2) dynamics: new velocities of every free body ...
        pre = {}
        for bid in ids:
            b = self._bodies[bid]
            if b.fixed:
                b.clear_forces()
                continue
            R = matrix_from_quat(b.q)
            w_l = R.T @ b.w
            v_l = R.T @ b.v
            I6 = np.zeros((6, 6))   # spatial inertia at the base origin, [angular; linear]
            bias = np.zeros(6)      # zero-acceleration force
            for l, r, f, t in zip(b.links, b.r, b.force, b.torque):
                m = l.mass
                rx = _skew(r)
                # link COM velocity in (axis-aligned) link frame
                v_i = v_l + np.cross(w_l, r)
                # spatial bias force at the link COM: (w x I w [gyro flag], m w x v)
                p_ang = np.cross(w_l, l.inertia @ w_l) if b.use_gyro_term else np.zeros(3)
                p_lin = m * np.cross(w_l, v_i)
                # external + gravity, into link frame
                f_l = R.T @ (f + m * self._gravity)
                t_l = R.T @ t
                z_ang = p_ang - t_l
                z_lin = p_lin - f_l
                # shift to the base origin: torque += r x force
                bias[:3] += z_ang + np.cross(r, z_lin)
                bias[3:] += z_lin
                I6[:3, :3] += l.inertia + m * (rx.T @ rx)
                I6[:3, 3:] += m * rx
                I6[3:, :3] += m * rx.T
                I6[3:, 3:] += m * np.eye(3)
            acc = -np.linalg.solve(I6, bias)
            wdot = R @ acc[:3]
            vdot = R @ (acc[3:] + np.cross(w_l, v_l))
            vm = b.max_coord_vel
            b.w = np.clip(b.w + wdot * dt, -vm, vm)
            b.v = np.clip(b.v + vdot * dt, -vm, vm)
            pre[bid] = (I6, R)
        # ... the contact response between the free bodies (impulses on those velocities) ...
        shift = self._solve_pair_contacts(pre) if (self.contact_response and self.pair_response) else {}
        # ... then every body's ground solve and integration
        for bid in ids:
            b = self._bodies[bid]
            if b.fixed:
                continue
            I6, R = pre[bid]
            deepest = self._solve_contacts(b, I6, R, bid in self._persisted) if self.contact_response else 0.0
            b.p = b.p + dt * b.v
            b.p[2] += self.contact_erp * deepest
            if bid in shift:
                b.p = b.p + shift[bid]
            # exponential-map quaternion update with world-frame omega
            fAngle = math.sqrt(float(b.w @ b.w))
            if fAngle * dt > 0.25 * math.pi:
                fAngle = 0.25 * math.pi / dt
            if fAngle < 0.001:
                axis = b.w * (0.5 * dt - (dt * dt * dt) * 0.020833333333 * fAngle * fAngle)
            else:
                axis = b.w * (math.sin(0.5 * fAngle * dt) / fAngle)
            cw = math.cos(fAngle * dt * 0.5)
            x, y, z, w = b.q
            ax, ay, az = axis
            nq = np.array(
                [
                    cw * x + ax * w + ay * z - az * y,
                    cw * y + ay * w + az * x - ax * z,
                    cw * z + az * w + ax * y - ay * x,
                    cw * w - ax * x - ay * y - az * z,
                ]
            )
            b.q = nq * (1.0 / math.sqrt(float(nq @ nq)))
            b.clear_forces()
