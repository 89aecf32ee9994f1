"""upright_amd -- MI355X-native batched MPC engine for the waiter's problem.

Accelerates ONE path of utiasDSL/upright: the multiple-shooting SQP solve behind
`upright_control.bindings.ControllerInterface`.  Hand-written HIP kernels in
csrc/ behind the C-ABI of include/upright_mi.h; this package is the host-side mirror of the
reference's Python surface.  No CPU fallback: libupright_mi.so must be built (see __graft_entry__.build).
"""
from . import problem, robots  # noqa: F401

__all__ = ["problem", "robots"]
