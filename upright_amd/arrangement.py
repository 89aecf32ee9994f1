"""Arrangement front-end: controller config -> balanced bodies and contact points.

What the reference's `parse_control_objects` (`upright_core/src/upright_core/parsing.py:351-410`) produces from
the `objects` / `arrangements` / `balancing.arrangement` entries of a controller config: every balanced object is
a convex solid (cuboid, cylinder approximated by its inscribed square prism turned 45 degrees, or wedge) stacked
on its parent's top face; every `contacts` entry becomes the vertices of the overlap polygon of the two touching
faces, each with a normal pointing into the first object, a tangent basis, and lever arms that are pulled
towards the face centre by `support_area_inset`.  All positions are in the end-effector (tray) frame.

Pinned against the reference's own parser through `tests/golden/arrangements.json` (expected) and
`tests/golden/arrangement_inputs.json` (the same YAML entries as data): `tests/test_host.py`.  This is host-side
set-up code (runs once per controller); the GPU path consumes its output through `Problem` / `upr_problem`.
"""
import numpy as np
from scipy.linalg import null_space

from .core_bindings import ContactPoint, RigidBody

CONTACT_TOL = 1e-7   # parsing.py:168
GEOM_TOL = 1e-8      # polyhedron.py:8


def tangent_basis(normal):
    """Rows span the plane orthogonal to `normal` (math.py:163-177: transpose of scipy's null-space basis, which
    fixes the -- otherwise arbitrary -- rotation of the basis the contact `span` is expressed in)."""
    return null_space(np.asarray(normal, dtype=np.float64)[None, :]).T


def quat_xyzs_to_rot(q):
    x, y, z, s = np.asarray(q, dtype=np.float64) / np.linalg.norm(q)
    return np.array([
        [1 - 2 * (y * y + z * z), 2 * (x * y - s * z), 2 * (x * z + s * y)],
        [2 * (x * y + s * z), 1 - 2 * (x * x + z * z), 2 * (y * z - s * x)],
        [2 * (x * z - s * y), 2 * (y * z + s * x), 1 - 2 * (x * x + y * y)],
    ])


def rot_z(t):
    c, s = np.cos(t), np.sin(t)
    return np.array([[c, -s, 0.0], [s, c, 0.0], [0.0, 0.0, 1.0]])


class Hull:
    """Convex solid: vertices, outward unit face normals, and the reference point `origin` objects are placed by."""

    def __init__(self, vertices, normals, origin=None):
        self.vertices = np.asarray(vertices, dtype=np.float64)
        self.normals = np.asarray(normals, dtype=np.float64)
        self.origin = np.zeros(3) if origin is None else np.asarray(origin, dtype=np.float64)

    @staticmethod
    def cuboid(half):
        x, y, z = half
        V = [[x, y, z], [x, y, -z], [x, -y, -z], [x, -y, z], [-x, y, z], [-x, y, -z], [-x, -y, -z], [-x, -y, z]]
        return Hull(V, np.vstack([np.eye(3), -np.eye(3)]))

    @staticmethod
    def wedge(half):
        """Right triangular prism, the slope facing +x (polyhedron.py:66-90)."""
        x, y, z = half
        V = np.array([[-x, -y, -z], [x, -y, -z], [-x, -y, z], [-x, y, -z], [x, y, -z], [-x, y, z]], dtype=np.float64)
        slope = np.cross(V[4] - V[1], V[2] - V[1])
        return Hull(V, np.vstack([-np.eye(3), [0.0, 1.0, 0.0], slope / np.linalg.norm(slope)]))

    def moved(self, translation=None, rotation=None):
        t = np.zeros(3) if translation is None else np.asarray(translation, dtype=np.float64)
        R = np.eye(3) if rotation is None else rotation
        return Hull(t + self.vertices @ R.T, self.normals @ R.T, R @ self.origin + t)

    def extent(self, axis):
        p = self.vertices @ (axis / np.linalg.norm(axis))
        return p.min(), p.max()

    def farthest(self, axis):
        return self.vertices[np.argmax(self.vertices @ (axis / np.linalg.norm(axis)))]

    def exit_distance(self, axis):
        """Distance from `origin` to the boundary along `axis` (the reference solves a linear programme for the
        same number, polyhedron.py:195-229; a ray leaves a convex solid through the nearest face it heads for)."""
        a = np.asarray(axis, dtype=np.float64) / np.linalg.norm(axis)
        best = np.inf
        for n in self.normals:
            rate = n @ a
            if rate > 1e-12:
                best = min(best, ((self.vertices @ n).max() - n @ self.origin) / rate)
        if not np.isfinite(best) or best < -GEOM_TOL:
            raise ValueError("reference point outside its solid")
        return best

    def section(self, point, normal, basis, tol):
        """Vertices lying in the plane (point, normal), in `basis` coordinates, counter-clockwise."""
        V = self.vertices[np.abs((self.vertices - point) @ normal) < tol]
        return wind((V - point) @ basis.T)


def wind(P):
    c = P.mean(axis=0)
    return P[np.argsort(np.arctan2(P[:, 1] - c[1], P[:, 0] - c[0]))]


def _clip_edge(a, b, p, n, tol):
    """Part of segment a-b on the inner side of the line through p with inward normal n (() if none)."""
    da, db = n @ (a - p), n @ (b - p)
    if da >= -tol and db >= -tol:
        return (a, b)
    if da <= tol and db <= tol:
        return ()
    if abs(da) < tol:
        x = a
    elif abs(db) < tol:
        x = b
    else:
        x = a + (n @ (p - a)) / (n @ (b - a)) * (b - a)
    return (a, x) if da > 0 else (x, b)


def overlap_polygon(subject, clip, tol=GEOM_TOL):
    """Intersection of two convex counter-clockwise polygons (edge-by-edge clipping of `subject` by the half
    planes of `clip`, polyhedron.py:336-400); the vertex order it yields is the order of the contact points, hence
    of the contact-force block of the input vector."""
    P = subject
    for i in range(len(clip)):
        p = clip[i]
        d = clip[(i + 1) % len(clip)] - p
        if np.linalg.norm(d) < tol:
            raise ValueError("clipping polygon has repeated vertices")
        d = d / np.linalg.norm(d)
        n = np.array([-d[1], d[0]])
        pieces = []
        for j in range(len(P)):
            pieces.extend(_clip_edge(P[j], P[(j + 1) % len(P)], p, n, tol))
        kept = []
        for v in pieces:
            if not any(np.linalg.norm(v - w) < tol for w in kept):
                kept.append(v)
        if not kept:
            return None
        P = np.array(kept)
    return P


def touching_contact(A, B, tol=CONTACT_TOL):
    """Contact points and normal (pointing into A) of two solids that touch without penetrating
    (separating-axis sweep over face normals and their cross products, polyhedron.py:435-514).  When several axes
    touch, the last one in that order decides, as in the reference."""
    cross = []
    for na in A.normals:
        for nb in B.normals:
            c = np.cross(na, nb)
            m = np.linalg.norm(c)
            if m > tol:
                cross.append(c / m)
    axes = np.vstack([A.normals, B.normals] + ([np.array(cross)] if cross else []))
    hit = None
    for axis in axes:
        (alo, ahi), (blo, bhi) = A.extent(axis), B.extent(axis)
        upper, lower = min(ahi, bhi), max(alo, blo)
        if abs(upper - lower) < tol:
            hit = (axis, A.farthest(axis), -1.0) if alo < blo else (axis, B.farthest(axis), 1.0)
        elif upper < lower:
            return None, None   # apart
    if hit is None:
        return None, None       # penetrating
    axis, point, sign = hit
    S = tangent_basis(axis)
    poly = overlap_polygon(A.section(point, axis, S, tol), B.section(point, axis, S, tol), tol)
    if poly is None:
        return None, None
    return point + poly @ S, sign * axis


# ---- objects ------------------------------------------------------------------------------------------------------
def _half_extents(conf):
    shape = conf["shape"].lower()
    if shape in ("cuboid", "wedge"):
        return 0.5 * np.asarray(conf["side_lengths"], dtype=np.float64)
    if shape == "cylinder":
        w = np.sqrt(2.0) * conf["radius"]
        return 0.5 * np.array([w, w, conf["height"]])
    raise ValueError(f"Unsupported shape type: {shape}")


def solid_of(conf, translation=None, rotation=None):
    R = np.eye(3) if rotation is None else rotation
    shape = conf["shape"].lower()
    half = _half_extents(conf)
    if shape == "wedge":
        return Hull.wedge(half).moved(translation, R)
    if shape == "cylinder":
        R = R @ rot_z(np.pi / 4)   # contacts of the inscribed prism line up with the x / y axes (parsing.py:243-247)
    return Hull.cuboid(half).moved(translation, R)


def uniform_inertia(mass, conf):
    shape = conf["shape"].lower()
    if shape == "cylinder":
        r, h = conf["radius"], conf["height"]
        return np.diag([mass * (3 * r * r + h * h) / 12] * 2 + [0.5 * mass * r * r])
    if shape == "cuboid":
        lx, ly, lz = conf["side_lengths"]
        return mass * np.diag([ly * ly + lz * lz, lx * lx + lz * lz, lx * lx + ly * ly]) / 12.0
    if shape == "wedge":
        hx, hy, hz = 0.5 * np.asarray(conf["side_lengths"], dtype=np.float64)
        return mass * np.array([   # math.py:127-146 (the reference diagonalises this matrix and rotates it back)
            [hy ** 2 / 3 + 2 * hz ** 2 / 9, 0, hx * hz / 9],
            [0, 2 * hx ** 2 / 9 + 2 * hz ** 2 / 9, 0],
            [hx * hz / 9, 0, 2 * hx ** 2 / 9 + hy ** 2 / 3]])
    raise ValueError(f"Unsupported shape type {shape}.")


def support_offset(d):
    x, y = d.get("x", 0), d.get("y", 0)
    if "r" in d and "θ" in d:
        from .config import parse_number

        th = parse_number(d["θ"])
        x, y = x + d["r"] * np.cos(th), y + d["r"] * np.sin(th)
    elif "r" in d or "θ" in d:
        raise ValueError("Radius and angle must *both* be specified in support offset.")
    return np.array([x, y], dtype=np.float64)


def _body_and_solid(conf, base, quat):
    """`base` lies on the parent's top face directly below the object's reference point (parsing.py:298-348)."""
    mass = conf["mass"]
    C = quat_xyzs_to_rot(quat)
    com_local = np.array(conf["com_offset"], dtype=np.float64)
    if conf["shape"].lower() == "wedge":   # reference point = centre of the enclosing box, not the centroid
        hx, _, hz = 0.5 * np.asarray(conf["side_lengths"], dtype=np.float64)
        com_local = com_local + np.array([-hx, 0.0, -hz]) / 3
    if "inertia" in conf:
        I = np.array(conf["inertia"], dtype=np.float64)
        if I.shape == (3,):
            I = np.diag(I)
        elif I.shape != (3, 3):
            raise ValueError(f"Object inertia matrix has wrong shape: {I.shape}")
    elif "inertia_diag" in conf:
        I = np.diag(np.asarray(conf["inertia_diag"], dtype=np.float64))
    else:
        I = uniform_inertia(mass, conf)
    drop = solid_of(conf, rotation=C).exit_distance(np.array([0.0, 0.0, -1.0]))
    ref = np.asarray(base, dtype=np.float64) + np.array([0.0, 0.0, drop])
    return RigidBody(mass, C @ I @ C.T, ref + C @ com_local), solid_of(conf, ref, C)


def _inset(r, centre, S, inset):
    """Pull the tangential part of r towards `centre` by `inset` (math.py:149-158, parsing.py:196-211)."""
    t = S @ (r - centre)
    d = np.linalg.norm(t)
    if d <= inset:
        raise ValueError(f"Inset of {inset} is too large for the support area.")
    return r + ((d - inset) * t / d - t) @ S


def parse_control_objects(ctrl_conf):
    """(bodies: dict name -> RigidBody of the balanced objects, contacts: list of ContactPoint), the outputs of
    `upright_core.parsing.parse_control_objects` for the same config dict."""
    arrangement = ctrl_conf["arrangements"][ctrl_conf["balancing"]["arrangement"]]
    types = ctrl_conf["objects"]
    for conf in types.values():   # older config format: shape: {type: ..., <dimensions>}
        if isinstance(conf.get("shape"), dict):
            shape = dict(conf["shape"])
            conf["shape"] = shape.pop("type")
            conf.update(shape)
    ee = types["ee"]
    solids = {"ee": solid_of(ee, np.asarray(ee["position"], dtype=np.float64))}
    fixtures = {"ee": True}
    bodies = {}
    up = np.array([0.0, 0.0, 1.0])
    for inst in arrangement["objects"]:
        name = inst["name"]
        if name in solids:
            raise ValueError(f"Multiple control objects named {name}.")
        parent = solids[inst["parent"]]
        base = parent.origin.copy()
        if "offset" in inst:
            base[:2] += support_offset(inst["offset"])
        base[2] += parent.exit_distance(up)
        body, solid = _body_and_solid(types[inst["type"]], base, np.asarray(inst.get("orientation", [0, 0, 0, 1]), dtype=np.float64))
        solids[name] = solid
        fixtures[name] = bool(inst.get("fixture", False))
        if not fixtures[name]:
            bodies[name] = body
    contacts = []
    for c in arrangement["contacts"]:
        first, second = c["first"], c["second"]
        mu = c["mu"] - c.get("mu_margin", 0)
        inset = c.get("support_area_inset", 0)
        points, normal = touching_contact(solids[first], solids[second], CONTACT_TOL)
        if points is None:
            raise ValueError(f"No contact points found between {first} and {second}.")
        S = tangent_basis(normal)
        for r in points:
            p = ContactPoint()
            p.object1_name, p.object2_name, p.mu = first, second, mu
            p.normal, p.span = normal, S
            # no inset with respect to fixtures (the tray or anything bolted to it): their dynamics do not matter
            p.r_co_o1 = r if fixtures[first] else _inset(r, solids[first].origin, S, inset)
            p.r_co_o2 = _inset(r, solids[second].origin, S, inset)
            contacts.append(p)
    return bodies, contacts
