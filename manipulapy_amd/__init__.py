"""manipulapy_amd — MI355X-native batched trajectory + rigid-body dynamics.

A from-scratch gfx950 implementation of ONE hot path of boelnasr/ManipulaPy v1.4.1 (quintic / cubic
joint trajectories, product-of-exponentials FK and space Jacobian, inverse dynamics over
N timesteps x B trajectories) behind the reference's own backend-dispatch and kernel-registry seams.
Host code is Python + ctypes over the C ABI in include/manipula_hip.h; see DESIGN.md.
"""
from .backend import (ArrayBackend, HipBackend, NumpyBackend, get_backend, get_registered, register, set_backend,
                      use_backend)
from .registry import (BackendNotSupportedError, KernelRegistration, KernelRegistry, check_hip_availability,
                       execute_registered_kernel, get_context, get_gpu_properties, get_registered_kernel)
from .kinematics import SerialManipulator
from .dynamics import ManipulatorDynamics
from .planning import OptimizedTrajectoryPlanning, TrajectoryPlanning
from .control import ManipulatorController
from .singularity import Singularity
from . import ik_helpers, utils
from .potential_field import PotentialField
from .robots import load_robot, robot_tables, robot_urdf
from .urdf import URDFToSerialManipulator
from . import trac_ik
from .trac_ik import TracIKSolver, trac_ik_solve

__version__ = "0.1.0"
__all__ = ["ArrayBackend", "HipBackend", "NumpyBackend", "get_backend", "get_registered", "register", "set_backend",
           "use_backend", "BackendNotSupportedError", "KernelRegistration", "KernelRegistry", "check_hip_availability",
           "execute_registered_kernel", "get_context", "get_gpu_properties", "get_registered_kernel",
           "SerialManipulator", "ManipulatorDynamics", "OptimizedTrajectoryPlanning", "TrajectoryPlanning", "ManipulatorController", "Singularity", "ik_helpers", "utils", "PotentialField",
           "load_robot", "robot_tables", "robot_urdf", "URDFToSerialManipulator", "trac_ik", "TracIKSolver", "trac_ik_solve"]
