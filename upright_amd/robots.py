"""Explicit kinematic chains for the robots the reference's configs name.

The reference builds its Pinocchio model at run time from a xacro-generated URDF
(`upright_cmd/config/robots/thing.yaml:52-55`) whose main include,
`mobile_manipulation_central/urdf/xacro/thing_no_wheels.urdf.xacro`, is NOT vendored under
/root/reference.  What IS in the reference:

  * the root joint: a planar composite PX, PY, RZ (`upright_control/include/upright_control/util.h:28-32`),
    locked at `base_pose` for the fixed-base UR10 configs (`util.h:35-47`, `robots/ur10.yaml:50`);
  * the tool link `gripped_object`, attached to link `gripper` by the calibrated transform
    `gripped_object_transform` (`upright_assets/thing/xacro/end_effectors/gripped_object.urdf.xacro:6-10`,
    values in `upright_cmd/config/robots/calibration/tray_transforms_real.yaml`);
  * the calibration delta `base_to_arm_transform` (same YAML);
  * joint names/order (`upright_cmd/config/robots/thing.yaml:24-33`) and the home configuration
    (`thing.yaml:16`).

Everything else below is OUR documented model, so forward-kinematics parity with the authors' URDF is
UNPINNED (stated in DESIGN.md):

  * the arm is a UR10 with the public ROS-Industrial `ur_description` kinematic parameters
    (d1 .1273, a2 -.612, a3 -.5723, d4 .163941, d5 .1157, d6 .0922, shoulder/elbow offsets
    .220941/-.1719) and that package's joint-origin convention;
  * `gripper` = UR10 `tool0` rotated by +pi/12 about its z axis.  This is inferred, not read: with it
    the tray normal at the reference's home configuration (wrist_3 = 0.417 pi = pi/2 - pi/12) is
    vertical to within the calibration residual (z_tray . z_world = 0.9999), which is the
    physical situation the home pose exists for;
  * arm mount on the mobile base: nominal offset ARM_MOUNT_XYZ and yaw ARM_MOUNT_RPY, then the calibration delta.  The yaw is
    fixed by two statements of the reference's own configs (round 4; `tests/test_host.py::test_home_pose_and_the_arm_mount`):
    (i) at the home configuration (`thing.yaml:16`, base at (-1, 1, 0)) every one of the 20 collision pairs of
    `obstacles/simple.yaml:11-41` must clear `minimum_distance` -- the authors start their static-obstacle runs from home;
    (ii) `ral23/experiments/_point1.yaml:1-2` says its waypoint, EE + (-2, 1, 0), is "doable with either base or arm ... not
    outside of the arm's workspace", i.e. home EE and target both within reach of the shoulder.  Of the four right-angle
    mounts only yaw = -pi/2 (shoulder_pan = pi/2 then points the arm along the base's +x) satisfies either: it clears every pair
    by >= 0.30 m and puts both points within 1.40 m of the shoulder axis; yaw = 0 (rounds 1 - 3) left the tray 0.16 m INSIDE the
    margin of obstacle 3 at home and the _point1 target 3.1 m from the shoulder.

A chain is a list of joints; joint i's frame = parent frame * (R_i, p_i) * motion(axis_i, q_i), and a
final fixed tool transform (tool_R, tool_p).
"""
from dataclasses import dataclass, field

import numpy as np

PRISMATIC = 0
REVOLUTE = 1

# UR10 (ur_description, ur10.urdf.xacro)
_SH = 0.1273
_UA = 0.612
_FA = 0.5723
_SO = 0.220941
_EO = -0.1719
_W1 = 0.163941 - _EO - _SO
_W2 = 0.1157
_W3 = 0.0922

# nominal mount of the arm base on the mobile base (documented assumption, see module docstring)
ARM_MOUNT_XYZ = np.array([0.27, 0.01, 0.653])
ARM_MOUNT_RPY = np.array([0.0, 0.0, -np.pi / 2])
GRIPPER_YAW = np.pi / 12

# upright_cmd/config/robots/calibration/tray_transforms_real.yaml (== ..._2025-01-28_11-09-50.yaml)
CALIBRATION_REAL = {
    "base_to_arm_transform": {
        "rpy": [-0.0032362802885472775, -0.0005883892881684005, 0.0055597638711333275],
        "xyz": [0.00890486128628254, -0.009501900523900986, 0.001751916715875268],
    },
    "gripped_object_transform": {
        "rpy": [-2.8848612308502197, -1.5584771633148193, -0.24215367436408997],
        "xyz": [0.038776129484176636, 0.0020609176717698574, 0.3146381676197052],
    },
}


def rpy_to_rot(rpy):
    """URDF fixed-axis roll-pitch-yaw: R = Rz(y) Ry(p) Rx(r)."""
    r, p, y = rpy
    cr, sr, cp, sp, cy, sy = np.cos(r), np.sin(r), np.cos(p), np.sin(p), np.cos(y), np.sin(y)
    Rx = np.array([[1, 0, 0], [0, cr, -sr], [0, sr, cr]])
    Ry = np.array([[cp, 0, sp], [0, 1, 0], [-sp, 0, cp]])
    Rz = np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1]])
    return Rz @ Ry @ Rx


@dataclass
class Joint:
    kind: int
    axis: np.ndarray
    R: np.ndarray = field(default_factory=lambda: np.eye(3))
    p: np.ndarray = field(default_factory=lambda: np.zeros(3))
    name: str = ""


@dataclass
class Chain:
    joints: list
    tool_R: np.ndarray
    tool_p: np.ndarray
    name: str = ""
    base_pose: tuple = None   # (x, y, yaw) of the locked base of the fixed-base arm

    @property
    def nq(self):
        return len(self.joints)

    def forward(self, q):
        """EE pose (p, C) -- plain numpy, used for target construction (wrappers.py:31-43)."""
        R = np.eye(3)
        o = np.zeros(3)
        for j, qi in zip(self.joints, q):
            o = o + R @ j.p
            R = R @ j.R
            if j.kind == REVOLUTE:
                R = R @ _axis_rot(j.axis, qi)
            else:
                o = o + R @ j.axis * qi
        o = o + R @ self.tool_p
        R = R @ self.tool_R
        return o, R

    def forward_motion(self, q, v, a):
        """EE pose, spatial velocity and CLASSICAL acceleration in the world frame: (p, C, v, omega, a, alpha) -- what
        robot.py's `link_pose`, `link_velocity` and `link_classical_acceleration` return for the tool link.  Plain numpy
        outward recursion over the serial chain (the host-side twin of upr_kin.h's value walk)."""
        R = np.eye(3)
        o = np.zeros(3); vo = np.zeros(3); ao = np.zeros(3); w = np.zeros(3); al = np.zeros(3)

        def shift(r):   # the point at r (world coordinates, from o) of the body that carries (vo, ao, w, al)
            return o + r, vo + np.cross(w, r), ao + np.cross(al, r) + np.cross(w, np.cross(w, r))

        for j, qi, vi, ai in zip(self.joints, q, v, a):
            o, vo, ao = shift(R @ j.p)
            R = R @ j.R
            z = R @ j.axis
            if j.kind == REVOLUTE:
                al = al + z * ai + np.cross(w, z) * vi
                w = w + z * vi
                R = R @ _axis_rot(j.axis, qi)
            else:
                o = o + z * qi
                ao = ao + z * ai + 2.0 * np.cross(w, z) * vi
                vo = vo + z * vi
        o, vo, ao = shift(R @ self.tool_p)
        R = R @ self.tool_R
        return o, R, vo, w, ao, al


def _axis_rot(ax, th):
    ax = np.asarray(ax, dtype=float)
    K = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
    return np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * (K @ K)


def _compose(Ra, pa, Rb, pb):
    return Ra @ Rb, pa + Ra @ pb


def _ur10_joints(R0, p0):
    """Six revolute joints; (R0, p0) is the fixed transform in front of shoulder_pan."""
    hp = np.pi / 2
    specs = [
        ("ur10_arm_shoulder_pan_joint", (0, 0, _SH), (0, 0, 0), (0, 0, 1)),
        ("ur10_arm_shoulder_lift_joint", (0, _SO, 0), (0, hp, 0), (0, 1, 0)),
        ("ur10_arm_elbow_joint", (0, _EO, _UA), (0, 0, 0), (0, 1, 0)),
        ("ur10_arm_wrist_1_joint", (0, 0, _FA), (0, hp, 0), (0, 1, 0)),
        ("ur10_arm_wrist_2_joint", (0, _W1, 0), (0, 0, 0), (0, 0, 1)),
        ("ur10_arm_wrist_3_joint", (0, 0, _W2), (0, 0, 0), (0, 1, 0)),
    ]
    joints = []
    for i, (name, xyz, rpy, axis) in enumerate(specs):
        R, p = rpy_to_rot(rpy), np.array(xyz, dtype=float)
        if i == 0:
            R, p = _compose(R0, p0, R, p)
        joints.append(Joint(REVOLUTE, np.array(axis, dtype=float), R, p, name))
    return joints


def _tool(calibration):
    # wrist_3 -> tool0 (ur_description) -> gripper (yaw pi/12) -> gripped_object (calibrated)
    R, p = rpy_to_rot((-np.pi / 2, 0, 0)), np.array([0, _W3, 0.0])
    R, p = _compose(R, p, rpy_to_rot((0, 0, GRIPPER_YAW)), np.zeros(3))
    g = calibration["gripped_object_transform"]
    R, p = _compose(R, p, rpy_to_rot(g["rpy"]), np.array(g["xyz"], dtype=float))
    return R, p


def _arm_mount(calibration, mount_yaw=None):
    rpy = ARM_MOUNT_RPY if mount_yaw is None else np.array([ARM_MOUNT_RPY[0], ARM_MOUNT_RPY[1], float(mount_yaw)])
    R, p = rpy_to_rot(rpy), ARM_MOUNT_XYZ.copy()
    b = calibration["base_to_arm_transform"]
    return _compose(R, p, rpy_to_rot(b["rpy"]), np.array(b["xyz"], dtype=float))


def thing(calibration=None, mount_yaw=None):
    """Omnidirectional base (PX, PY, RZ) + UR10: nq = 9 (`robots/thing.yaml:43-47`).  mount_yaw: another yaw of the arm mount than
    ARM_MOUNT_RPY's (comparisons with rounds 1 - 3, which had it at 0)."""
    calibration = calibration or CALIBRATION_REAL
    base = [
        Joint(PRISMATIC, np.array([1.0, 0, 0]), name="x_to_world_joint"),
        Joint(PRISMATIC, np.array([0, 1.0, 0]), name="y_to_x_joint"),
        Joint(REVOLUTE, np.array([0, 0, 1.0]), name="base_to_y_joint"),
    ]
    R0, p0 = _arm_mount(calibration, mount_yaw)
    tR, tp = _tool(calibration)
    return Chain(base + _ur10_joints(R0, p0), tR, tp, "thing")


def ur10(base_pose=(-1.0, 1.0, 0.0), calibration=None):
    """Fixed base: root joint locked at base_pose = (x, y, yaw) (`util.h:35-47`): nq = 6."""
    calibration = calibration or CALIBRATION_REAL
    Rb = rpy_to_rot((0, 0, base_pose[2]))
    pb = np.array([base_pose[0], base_pose[1], 0.0])
    R0, p0 = _compose(Rb, pb, *_arm_mount(calibration))
    tR, tp = _tool(calibration)
    return Chain(_ur10_joints(R0, p0), tR, tp, "ur10", tuple(float(v) for v in base_pose))


def from_config(robot_config):
    """Pick the chain for a controller `robot` config dict (`wrappers.py:143-147,246-260`)."""
    base_type = str(robot_config.get("base_type", "omnidirectional")).lower()
    nq = int(robot_config["dims"]["q"])
    if base_type == "fixed":
        chain = ur10(tuple(robot_config.get("base_pose", (0.0, 0.0, 0.0))))
    elif base_type == "omnidirectional":
        chain = thing()
    else:
        # base_type.h:24: nonholonomic/floating are declared but never used by a shipped config
        raise ValueError(f"unsupported base type: {base_type}")
    if chain.nq != nq:
        raise ValueError(f"robot dims.q = {nq} does not match the {chain.name} chain ({chain.nq} joints)")
    return chain


# ---- collision model ------------------------------------------------------------------------------------------------
# Collision spheres of the Thing (upright_assets/thing/xacro/collision_links.urdf.xacro:32-181): name ->
# (parent link, offset of the sphere centre in the parent link frame, radius).  Only translations matter for spheres.
COLLISION_SPHERES = {
    "balanced_object_collision_link": ("gripped_object", (0.0, 0.0, 0.07), 0.25),
    "shoulder_collision_link": ("ur10_arm_upper_arm_link", (0.0, 0.0, 0.0), 0.15),
    "wrist1_collision_link": ("ur10_arm_wrist_1_link", (0.0, 0.0, -0.05), 0.15),
    "wrist3_collision_link": ("ur10_arm_wrist_3_link", (0.0, 0.0, 0.0), 0.15),
    "base_collision_link": ("base_link", (0.0, 0.0, 0.0), 0.5),
    "forearm_collision_sphere_link1": ("ur10_arm_forearm_link", (-0.2, 0.0, 0.06), 0.15),
    "forearm_collision_sphere_link2": ("ur10_arm_forearm_link", (-0.4, 0.0, 0.06), 0.15),
}
# static obstacles of upright_assets/thing/xacro/obstacles/simple.urdf.xacro:41-100 (world frame)
SIMPLE_OBSTACLES = {
    f"sphere{i + 1}_{lvl}_link": ("world", (x, y, z), 0.25)
    for i, (x, y) in enumerate(((0.0, 0.25), (1.5, 1.0), (-0.5, 2.0)))
    for lvl, z in (("bottom", 0.25), ("middle", 0.5), ("top", 0.75))
}
# arm link -> index of the arm joint whose motion carries it (child link of that joint)
_ARM_LINKS = {"ur10_arm_shoulder_link": 0, "ur10_arm_upper_arm_link": 1, "ur10_arm_forearm_link": 2,
              "ur10_arm_wrist_1_link": 3, "ur10_arm_wrist_2_link": 4, "ur10_arm_wrist_3_link": 5}


def link_frame(chain, link):
    """Chain frame index of a URDF link: -1 = world (or a link that does not move with the chain, e.g. the base of
    the fixed-base arm), i = link carried by joint i, nq = tool frame (`gripped_object`)."""
    nq = chain.nq
    base = nq - 6   # 3 planar base joints in front of the arm, or none
    if link == "world":
        return -1
    if link == "gripped_object":
        return nq
    if link == "base_link":
        return base - 1 if base > 0 else -1
    if link in _ARM_LINKS:
        return base + _ARM_LINKS[link]
    raise ValueError(f"unknown link '{link}'")


def collision_model(chain, pairs, spheres=None, dynamic=None):
    """Sphere table and pair index lists for the named collision pairs of `obstacles.collision_pairs`
    (obstacles/simple.yaml:11-41, obstacles/dynamic.yaml:19-37; pinocchio appends `_0` to geometry names).
    `dynamic`: {name: radius} of the dynamic obstacles IN THE ORDER of `obstacles.dynamic` (controller_interface.cpp:55-82: a
    sphere on a translating joint each; the i-th owns the i-th 9-block of obstacle state); "ground" is the half-space z >= 0 (controller_interface.cpp:93-101).  Returns a dict with the Problem
    fields sph_frame, sph_off, sph_r, pair_a, pair_b (pair_b = -1: ground)."""
    table = dict(COLLISION_SPHERES)
    table.update(SIMPLE_OBSTACLES)
    if spheres:
        table.update(spheres)
    dyn_index = {}
    for i, (n, r) in enumerate((dynamic or {}).items()):   # (in the order of obstacles.dynamic: obstacle i owns state entries 9 i .. 9 i + 8)
        table[n] = ("dynamic", (0.0, 0.0, 0.0), float(r)); dyn_index[n] = i
    strip = lambda n: n[:-2] if n.endswith("_0") else n
    names = []
    for a, b in pairs:
        if strip(a) == "ground":
            raise ValueError("'ground' must be the second object of a collision pair")
        for n in (a, b):
            n = strip(n)
            if n == "ground":
                continue
            if n not in table:
                raise ValueError(f"unknown collision object '{n}'")
            if n not in names:
                names.append(n)
    idx = {n: i for i, n in enumerate(names)}
    idx["ground"] = -1
    frames, offs = [], []
    for n in names:
        link, off, _ = table[n]
        off = np.asarray(off, dtype=np.float64)
        if link == "dynamic":
            f = -2 - dyn_index[n]   # rides on dynamic obstacle dyn_index[n] (include/upright_mi.h: sph_frame)
        else:
            f = link_frame(chain, link)
            if link == "base_link" and f == -1 and chain.base_pose is not None:
                x, y, yaw = chain.base_pose
                c, s = np.cos(yaw), np.sin(yaw)
                off = np.array([x + c * off[0] - s * off[1], y + s * off[0] + c * off[1], off[2]])
        frames.append(f); offs.append(off)
    return dict(
        sph_frame=np.array(frames, dtype=np.int32), sph_off=np.array(offs, dtype=np.float64).reshape(len(names), 3),
        sph_r=np.array([table[n][2] for n in names], dtype=np.float64),
        pair_a=np.array([idx[strip(a)] for a, _ in pairs], dtype=np.int32),
        pair_b=np.array([idx[strip(b)] for _, b in pairs], dtype=np.int32),
        sphere_names=names,
    )


def add_projectile_rows(P, links, distances, scale):
    """projectile_path_constraint (controller_interface.cpp:272-294): the checked link positions are the origins of
    the named collision links; they become (or reuse) entries of the problem's sphere table."""
    names = list(getattr(P, "sphere_names", []))
    frames, offs, rads = list(P.sph_frame), [np.asarray(o) for o in P.sph_off], list(P.sph_r)
    idxs = []
    for n in links:
        n = n[:-2] if n.endswith("_0") else n
        if n not in names:
            if n not in COLLISION_SPHERES:
                raise ValueError(f"unknown collision object '{n}'")
            link, off, r = COLLISION_SPHERES[n]
            names.append(n); frames.append(link_frame(P.chain, link)); offs.append(np.asarray(off, dtype=np.float64)); rads.append(r)
        idxs.append(names.index(n))
    P.sphere_names = names
    P.sph_frame = np.array(frames, dtype=np.int32); P.sph_off = np.array(offs, dtype=np.float64).reshape(len(names), 3); P.sph_r = np.array(rads, dtype=np.float64)
    P.proj_sph = np.array(idxs, dtype=np.int32); P.proj_dist = np.asarray(distances, dtype=np.float64).copy(); P.proj_scale = float(scale)
    return P


# obstacles/simple.yaml:11-41
SIMPLE_COLLISION_PAIRS = (
    [("wrist1_collision_link_0", f"sphere{i}_top_link_0") for i in (1, 2, 3)]
    + [(f"forearm_collision_sphere_link{j}_0", f"sphere{i}_top_link_0") for i in (1, 2, 3) for j in (1, 2)]
    + [("base_collision_link_0", f"sphere{i}_bottom_link_0") for i in (1, 2, 3)]
    + [("balanced_object_collision_link_0", f"sphere{i}_top_link_0") for i in (1, 2, 3)]
    + [("balanced_object_collision_link_0", f"sphere{i}_middle_link_0") for i in (1, 2, 3)]
    + [("wrist1_collision_link_0", "shoulder_collision_link_0"), ("wrist1_collision_link_0", "base_collision_link_0")]
)
