"""Joint-space artificial potential field (reference potential_field/fields.py:35-160, `PotentialField`).

Attraction 1/2 k_a |q - q_goal|^2; repulsion 10 * sum over obstacles within the influence distance of
2 k_r (1/d - 1/d0)^2; gradient k_a (q - q_goal) + sum of -40 k_r (1/d - 1/d0) / d^3 (q - q_obs), with the reference's guards
(distances and the influence distance floored at 1e-10; an obstacle exactly at q pushes along the first axis with
magnitude k_r).  Host NumPy, vectorised over the obstacles; the Cartesian fused field on the GPU is a different function
(registry name `potential_field.fused`).

`CollisionChecker` is the mesh-less form of the reference's (potential_field/collision.py:35-230): that one builds a convex hull
per link from the URDF's collision / visual MESHES and tests the hulls pairwise; without mesh files (trimesh is absent here, and
the robot descriptions this package ships are kinematic + inertial skeletons) it has no hulls and never reports a collision -
which is exactly what this class reproduces: it parses the URDF (a missing or malformed file raises, as the reference's does),
keeps `convex_hulls = {}` and answers False.  Mesh loading stays out of scope.
"""
from __future__ import annotations

import numpy as np

__all__ = ["PotentialField", "CollisionChecker"]


class PotentialField:
    def __init__(self, attractive_gain: float = 1.0, repulsive_gain: float = 100.0, influence_distance: float = 0.5) -> None:
        self.attractive_gain = attractive_gain
        self.repulsive_gain = repulsive_gain
        self.influence_distance = influence_distance

    def compute_attractive_potential(self, q, q_goal):
        diff = (np.asarray(q) - np.asarray(q_goal)) * 1.0
        return diff.dtype.type(0.5) * diff.dtype.type(self.attractive_gain) * np.sum(diff ** 2)

    def _obstacles(self, q, obstacles):
        obs = [np.asarray(o) for o in obstacles]
        if not obs:
            return None, None
        diff = (np.asarray(q)[None, :] - np.stack(obs)) * 1.0
        return diff, np.linalg.norm(diff, axis=1)

    def compute_repulsive_potential(self, q, obstacles):
        diff, d = self._obstacles(q, obstacles)
        if diff is None:
            return 0
        t = d.dtype.type
        d0 = max(t(self.influence_distance), t(1e-10))
        contrib = t(2.0) * t(self.repulsive_gain) * (t(1.0) / np.maximum(d, t(1e-10)) - t(1.0) / d0) ** 2
        return t(10.0) * np.sum(np.where(d <= t(self.influence_distance), contrib, t(0.0)))

    def compute_gradient(self, q, q_goal, obstacles) -> np.ndarray:
        q = np.asarray(q)
        att = (q - np.asarray(q_goal)) * 1.0
        grad = att.dtype.type(self.attractive_gain) * att
        diff, d = self._obstacles(q, obstacles)
        if diff is None:
            return grad
        t = d.dtype.type
        eps, d0 = t(1e-10), max(t(self.influence_distance), t(1e-10))
        exact = d < eps
        dr = np.where(exact, t(1.0), np.maximum(d, eps))
        regular = (-t(40.0) * t(self.repulsive_gain) * (t(1.0) / dr - t(1.0) / d0) * (t(1.0) / dr ** 3))[:, None] * diff
        escape = np.zeros_like(diff)
        escape[:, 0] = t(self.repulsive_gain)
        contrib = np.where(exact[:, None], escape, regular)
        return grad + np.sum(np.where((d <= t(self.influence_distance))[:, None], contrib, t(0.0)), axis=0)


class CollisionChecker:
    def __init__(self, urdf_path: str, backend: str = "builtin", load_meshes: bool = True) -> None:
        import os

        from .urdf import UrdfError, _parse

        del load_meshes
        if backend != "builtin":
            raise NotImplementedError("only the built-in URDF reader exists in manipulapy_amd")
        if not os.path.isfile(str(urdf_path)):
            raise FileNotFoundError(f"URDF file not found: {os.path.abspath(str(urdf_path))}")
        try:
            self.links, self.joints = _parse(str(urdf_path))
        except UrdfError:
            raise
        self.urdf_path = str(urdf_path)
        self.convex_hulls: dict = {}   # no mesh geometry is ever loaded

    def check_collision(self, thetalist) -> bool:
        """True if two links' hulls intersect at `thetalist` - never, without hulls."""
        del thetalist
        return False
