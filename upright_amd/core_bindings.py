"""Drop-in for the pybind11 module `upright_core.bindings` (upright_core/src/pybindings.cpp:13-57).

Same class names, constructor signatures, attribute names and function signatures; the two functions run
the HIP kernels of libupright_mi.so through its C-ABI (upr_core_object_dynamics, upr_core_friction_rows)
and raise if no GPU / library is available.
"""
import ctypes as C

import numpy as np

from . import _capi


class RigidBody:
    """pybindings.cpp:16-22; rigid_body.h:28-64."""

    def __init__(self, mass, inertia, com):
        self.mass = float(mass)
        self.inertia = np.array(inertia, dtype=np.float64).reshape(3, 3)
        self.com = np.array(com, dtype=np.float64).reshape(3)

    @staticmethod
    def num_parameters():
        return 10

    def get_parameters(self):
        """rigid_body.h:47-51: [m, m*c, vech(I)]."""
        I = self.inertia
        return np.concatenate(([self.mass], self.mass * self.com, [I[0, 0], I[0, 1], I[0, 2], I[1, 1], I[1, 2], I[2, 2]]))

    @classmethod
    def from_parameters(cls, p, index=0):
        """rigid_body.h:36-43."""
        p = np.asarray(p, dtype=np.float64)[index:index + 10]
        v = p[4:]
        I = np.array([[v[0], v[1], v[2]], [v[1], v[3], v[4]], [v[2], v[4], v[5]]])
        return cls(p[0], I, p[1:4] / p[0])


class ContactPoint:
    """pybindings.cpp:24-32; contact.h:10-46."""

    def __init__(self):
        self.object1_name = ""
        self.object2_name = ""
        self.mu = 0.0
        self.r_co_o1 = np.zeros(3)
        self.r_co_o2 = np.zeros(3)
        self.normal = np.zeros(3)
        self.span = np.zeros((2, 3))


class Pose:
    def __init__(self):
        self.position = np.zeros(3)
        self.orientation = np.eye(3)

    @staticmethod
    def Zero():
        return Pose()


class Twist:
    def __init__(self):
        self.linear = np.zeros(3)
        self.angular = np.zeros(3)

    @staticmethod
    def Zero():
        return Twist()


class RigidBodyState:
    def __init__(self):
        self.pose = Pose()
        self.velocity = Twist()
        self.acceleration = Twist()

    @staticmethod
    def Zero():
        return RigidBodyState()


def contact_tables(bodies, contacts):
    """(sorted body names, body_params[nb][10], UprProblem with the contact table filled).
    Body order = std::map order (contact_constraints.h:180)."""
    names = sorted(bodies)
    if len(names) > _capi.MAXB or len(contacts) > _capi.MAXC:
        raise ValueError("too many bodies / contacts for libupright_mi")
    idx = {n: i for i, n in enumerate(names)}
    params = np.array([bodies[n].get_parameters() for n in names], dtype=np.float64).reshape(len(names), 10)
    P = _capi.UprProblem()
    P.nb, P.nc = len(names), len(contacts)
    for i, c in enumerate(contacts):
        if c.object2_name not in idx:
            # contact_constraints.h:141 bodies.at(object2_name) throws std::out_of_range -> IndexError
            raise IndexError(f"contact {i}: object2 '{c.object2_name}' is not a balanced body")
        P.contact_body1[i] = idx.get(c.object1_name, -1)
        P.contact_body2[i] = idx[c.object2_name]
        P.contact_mu[i] = float(c.mu)
        _capi._fill(P.contact_normal[i], c.normal)
        _capi._fill(P.contact_span[i], c.span)
        _capi._fill(P.contact_r1[i], c.r_co_o1)
        _capi._fill(P.contact_r2[i], c.r_co_o2)
    return names, params, P


def compute_object_dynamics_constraints(bodies, contacts, forces, state, gravity):
    """contact_constraints.h:162-194 (UNnormalised: the 1/sqrt(6 nb) factor is applied only by
    ObjectDynamicsConstraints, balancing_constraints.cpp:144-151).  forces: (3c,) or (c,) frictionless."""
    names, params, P = contact_tables(bodies, contacts)
    forces = np.ascontiguousarray(forces, dtype=np.float64).ravel()
    if forces.size == len(contacts):
        P.nf = 1  # contact_constraints.h:111
    elif forces.size == 3 * len(contacts):
        P.nf = 3
    else:
        raise ValueError("forces must have length c or 3c")
    _capi._fill(P.gravity, gravity)
    out = np.zeros(6 * len(names))
    Cm = np.ascontiguousarray(state.pose.orientation, dtype=np.float64).reshape(9)
    w = np.ascontiguousarray(state.velocity.angular, dtype=np.float64)
    al = np.ascontiguousarray(state.acceleration.angular, dtype=np.float64)
    a = np.ascontiguousarray(state.acceleration.linear, dtype=np.float64)
    _capi.check(_capi.lib().upr_core_object_dynamics(
        C.byref(P), _capi.ptr(params), 1, _capi.ptr(forces), _capi.ptr(Cm), _capi.ptr(w), _capi.ptr(al), _capi.ptr(a), _capi.ptr(out)))
    return out


def compute_contact_force_constraints_linearized(contacts, forces):
    """contact_constraints.h:50-77."""
    P = _capi.UprProblem()
    P.nc, P.nb, P.nf = len(contacts), 1, 3
    if P.nc > _capi.MAXC:
        raise ValueError("too many contacts for libupright_mi")
    for i, c in enumerate(contacts):
        P.contact_mu[i] = float(c.mu)
        _capi._fill(P.contact_normal[i], c.normal)
        _capi._fill(P.contact_span[i], c.span)
    forces = np.ascontiguousarray(forces, dtype=np.float64).ravel()
    if forces.size != 3 * len(contacts):
        raise ValueError("forces must have length 3c")
    out = np.zeros(5 * len(contacts))
    _capi.check(_capi.lib().upr_core_friction_rows(C.byref(P), 1, _capi.ptr(forces), _capi.ptr(out)))
    return out
