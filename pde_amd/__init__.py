"""pde_amd — MI355X-native engine for the PINN collocation hot path of Emily-Qu/PDE.

Public surface: the reference's own class API (`PhysicsInformedNN`,
`bcPhysicsInformedNN`) re-implemented over hand-written HIP kernels
(pde_amd/csrc, C-ABI in include/pinnjet.h).  No CPU fallback exists: without
libpinnjet.so and an AMD GPU construction raises.
"""
from . import equations  # noqa: F401
from .pinn import PhysicsInformedNN, Variant, bcPhysicsInformedNN, make_class  # noqa: F401

__all__ = ["PhysicsInformedNN", "bcPhysicsInformedNN", "Variant", "make_class", "equations"]
