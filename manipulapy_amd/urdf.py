"""URDF -> model tables: host-side reader for the constant tables the hot path consumes.

Reproduces what the reference derives from a URDF (ManipulaPy/urdf_processor.py:82-302 ->
urdf/core.py:440-489 kinematic structure, :670-769 `extract_screw_axes`; urdf/types.py:100-133 rpy ->
rotation, :202-239 spatial inertia; urdf/parser.py:327-420, :621-643, :738-810 element parsing), so that a
robot can be loaded from its URDF instead of from a captured fixture:

  * joints in XML order; kinematic chain = BFS from the root link(s), children visited in XML order;
  * actuated joints = non-fixed, non-mimic joints of that chain (revolute / continuous / prismatic;
    planar and floating are rejected, as in the reference);
  * home pose of every link by chaining joint origins (mimic joints sit at their offset);
  * per actuated joint: w = R_joint . axis (normalised), p = joint position; revolute S = [w; -w x p],
    prismatic S = [0; w];
  * Mlist_per_link[i] = T_child_link(0) . inertial.origin (the link frame itself without <inertial>);
  * G_i = blockdiag(I + m (|r|^2 1 - r r^T), m 1) with r = inertial xyz when |r| >= 1e-10, else
    blockdiag(I, m 1) — the inertia tensor is NOT rotated by the inertial rpy (reference quirk, kept:
    parity is to the reference, not to physics); links without <inertial> get G = eye(6);
  * joint limits = (lower, upper) of <limit> (missing attributes read as 0), (-pi, pi) without <limit>;
  * B = Ad(M^-1) S.

One deliberate difference: the reference's default end effector is element 0 of a Python *set* of leaf
links, i.e. it depends on PYTHONHASHSEED.  Here the default is deterministic — the leaf
link reached last by the BFS — and `tip_link=` selects any other link.  Only M / B depend on that choice;
the dynamics do not.  Meshes, materials, collision geometry, xacro and PyBullet limits are out of scope.
"""
from __future__ import annotations

import xml.etree.ElementTree as ET
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import numpy as np

__all__ = ["URDFToSerialManipulator", "extract_tables", "UrdfError"]


class UrdfError(ValueError):
    pass


def _floats(text: Optional[str], default: str, k: int = 3) -> np.ndarray:
    v = np.array([float(x) for x in (text if text is not None else default).split()], dtype=np.float64)
    if v.shape != (k,):
        raise UrdfError(f"expected {k} numbers, got {text!r}")
    return v


def _origin_matrix(elem: Optional[ET.Element]) -> np.ndarray:
    """<origin xyz rpy> -> 4x4, ZYX (yaw-pitch-roll) convention (reference urdf/types.py:100-133)."""
    T = np.eye(4)
    if elem is None:
        return T
    xyz = _floats(elem.get("xyz"), "0 0 0")
    r, p, y = _floats(elem.get("rpy"), "0 0 0")
    cr, sr, cp, sp, cy, sy = np.cos(r), np.sin(r), np.cos(p), np.sin(p), np.cos(y), np.sin(y)
    T[:3, :3] = [[cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr],
                 [sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr],
                 [-sp, cp * sr, cp * cr]]
    T[:3, 3] = xyz
    return T


@dataclass
class _Link:
    name: str
    has_inertial: bool = False
    com: np.ndarray = field(default_factory=lambda: np.eye(4))  # inertial origin in the link frame
    com_xyz: np.ndarray = field(default_factory=lambda: np.zeros(3))
    mass: float = 0.0
    inertia: np.ndarray = field(default_factory=lambda: np.zeros((3, 3)))


@dataclass
class _Joint:
    name: str
    kind: str
    parent: str
    child: str
    origin: np.ndarray
    axis: np.ndarray
    limit: Optional[Tuple[float, float]]
    mimic: Optional[Tuple[str, float, float]]  # (joint, multiplier, offset)

    def child_pose(self, q: float) -> np.ndarray:
        """Pose of the child link in the parent link for joint value q."""
        T = np.eye(4)
        if self.kind in ("revolute", "continuous"):
            x, y, z = self.axis
            K = np.array([[0, -z, y], [z, 0, -x], [-y, x, 0]])
            T[:3, :3] = np.eye(3) + np.sin(q) * K + (1 - np.cos(q)) * (K @ K)
        elif self.kind == "prismatic":
            T[:3, 3] = self.axis * q
        return self.origin @ T

    def child_poses(self, q: np.ndarray) -> np.ndarray:
        """child_pose for a vector of joint values: (N,) -> (N,4,4)."""
        q = np.asarray(q, dtype=np.float64)
        T = np.tile(np.eye(4), (q.shape[0], 1, 1))
        if self.kind in ("revolute", "continuous"):
            x, y, z = self.axis
            K = np.array([[0, -z, y], [z, 0, -x], [-y, x, 0]])
            T[:, :3, :3] = np.eye(3) + np.sin(q)[:, None, None] * K + (1 - np.cos(q))[:, None, None] * (K @ K)
        elif self.kind == "prismatic":
            T[:, :3, 3] = np.outer(q, self.axis)
        return self.origin @ T


def _parse(path: str) -> Tuple[Dict[str, _Link], List[_Joint]]:
    try:
        root = ET.parse(path).getroot()
    except ET.ParseError as exc:
        raise UrdfError(f"{path}: not well-formed XML ({exc})") from exc
    if root.tag != "robot":
        raise UrdfError(f"{path}: root element is <{root.tag}>, expected <robot>")
    links: Dict[str, _Link] = {}
    for le in root.findall("link"):
        name = le.get("name")
        if not name:
            raise UrdfError("link without a name")
        link = _Link(name)
        ie = le.find("inertial")
        if ie is not None:
            link.has_inertial = True
            oe = ie.find("origin")
            link.com = _origin_matrix(oe)
            link.com_xyz = link.com[:3, 3].copy()
            me = ie.find("mass")
            link.mass = float(me.get("value", 0)) if me is not None else 0.0
            ine = ie.find("inertia")
            if ine is not None:
                g = lambda k: float(ine.get(k, 0))  # noqa: E731
                link.inertia = np.array([[g("ixx"), g("ixy"), g("ixz")], [g("ixy"), g("iyy"), g("iyz")],
                                         [g("ixz"), g("iyz"), g("izz")]])
        links[name] = link
    joints: List[_Joint] = []
    for je in root.findall("joint"):
        name, kind = je.get("name"), je.get("type", "fixed")
        pe, ce = je.find("parent"), je.find("child")
        if pe is None or ce is None or not pe.get("link") or not ce.get("link"):
            raise UrdfError(f"Joint '{name}' missing parent or child link attribute")
        ae = je.find("axis")
        axis = _floats(ae.get("xyz") if ae is not None else None, "1 0 0")
        nrm = np.linalg.norm(axis)
        if nrm > 1e-10:
            axis = axis / nrm
        lim = je.find("limit")
        limit = (float(lim.get("lower", 0)), float(lim.get("upper", 0))) if lim is not None else None
        mm = je.find("mimic")
        mimic = (mm.get("joint"), float(mm.get("multiplier", 1)), float(mm.get("offset", 0))) if mm is not None else None
        joints.append(_Joint(name, kind, pe.get("link"), ce.get("link"), _origin_matrix(je.find("origin")), axis, limit, mimic))
    return links, joints


def _skew(v):
    return np.array([[0.0, -v[2], v[1]], [v[2], 0.0, -v[0]], [-v[1], v[0], 0.0]])


def _adjoint(T):
    A = np.zeros((6, 6))
    A[:3, :3] = T[:3, :3]
    A[3:, 3:] = T[:3, :3]
    A[3:, :3] = _skew(T[:3, 3]) @ T[:3, :3]
    return A


def extract_tables(path: str, tip_link: Optional[str] = None) -> Dict[str, object]:
    """S_list (6,n), B_list (6,n), M (4,4), G_list (n,6,6), Mlist_per_link (n,4,4), joint_limits (n,2),
    omega_list (3,n), r_list (3,n), joint_names, ee_name."""
    links, joints = _parse(path)
    children = {j.child for j in joints}
    parents = {j.parent for j in joints}
    roots = [name for name in links if name in parents and name not in children]
    if not roots:
        roots = [p for p in dict.fromkeys(j.parent for j in joints) if p not in children]
    if not roots:
        raise UrdfError("URDF has no root link (cyclic structure?)")
    by_parent: Dict[str, List[_Joint]] = {}
    for j in joints:
        by_parent.setdefault(j.parent, []).append(j)
    chain: List[_Joint] = []
    seen, queue = set(roots), list(roots)
    while queue:
        cur = queue.pop(0)
        for j in by_parent.get(cur, []):
            if j.child not in seen:
                chain.append(j)
                seen.add(j.child)
                queue.append(j.child)
    actuated = [j for j in chain if j.kind != "fixed" and j.mimic is None]
    if not actuated:
        raise UrdfError("No actuated joints found")
    # home pose of every link
    fk = {r: np.eye(4) for r in roots}
    for j in chain:
        q = j.mimic[2] if j.mimic is not None else 0.0  # mimic joints sit at multiplier * 0 + offset
        fk[j.child] = fk.get(j.parent, np.eye(4)) @ j.child_pose(q)
    leaves = [j.child for j in chain if j.child not in parents]
    ee = tip_link if tip_link is not None else (leaves[-1] if leaves else roots[0])
    if ee not in fk:
        raise UrdfError(f"tip_link '{ee}' not found among links")
    M = fk[ee].copy()
    n = len(actuated)
    S = np.zeros((6, n))
    om, rl = np.zeros((3, n)), np.zeros((3, n))
    G, Mcom = np.zeros((n, 6, 6)), np.zeros((n, 4, 4))
    limits = np.zeros((n, 2))
    for i, j in enumerate(actuated):
        if j.kind in ("planar", "floating"):
            raise UrdfError(f"Joint '{j.name}' is {j.kind}, which is not supported for SerialManipulator conversion.")
        if j.kind not in ("revolute", "continuous", "prismatic"):
            raise UrdfError(f"Joint '{j.name}': unknown type {j.kind!r}")
        Tj = fk.get(j.parent, np.eye(4)) @ j.origin
        w = Tj[:3, :3] @ j.axis
        w = w / np.linalg.norm(w)
        p = Tj[:3, 3]
        if j.kind == "prismatic":
            S[3:, i] = w
        else:
            S[:3, i] = w
            S[3:, i] = -np.cross(w, p)
        om[:, i], rl[:, i] = w, p
        link = links.get(j.child, _Link(j.child))
        Tl = fk[j.child]
        Mcom[i] = Tl @ link.com if link.has_inertial else Tl
        if link.has_inertial:
            r = link.com_xyz
            I = link.inertia if np.linalg.norm(r) < 1e-10 else link.inertia + link.mass * (r @ r * np.eye(3) - np.outer(r, r))
            G[i, :3, :3] = I
            G[i, 3:, 3:] = link.mass * np.eye(3)
        else:
            G[i] = np.eye(6)
        limits[i] = j.limit if j.limit is not None else (-np.pi, np.pi)
    B = _adjoint(np.linalg.inv(M)) @ S
    return {"M": M, "S_list": S, "B_list": B, "G_list": G, "Mlist_per_link": Mcom, "joint_limits": limits, "omega_list": om,
            "r_list": rl, "joint_names": [j.name for j in actuated], "ee_name": ee,
            "_tree": {"links": links, "joints": joints, "chain": chain, "roots": roots, "actuated": actuated}}


class URDFToSerialManipulator:
    """Drop-in for the reference's `URDFToSerialManipulator(urdf_name)` (urdf_processor.py:82-138): builds
    `.serial_manipulator` and `.dynamics` (this package's HIP-backed mirrors) plus `.robot_data`."""

    def __init__(self, urdf_name, use_pybullet_limits: bool = False, backend: str = "builtin", load_meshes: bool = False,
                 validate: bool = False, tip_link: Optional[str] = None) -> None:
        if use_pybullet_limits or backend != "builtin" or load_meshes:
            raise NotImplementedError("only the built-in, mesh-less URDF path exists in manipulapy_amd")
        del validate
        from .dynamics import ManipulatorDynamics
        from .kinematics import SerialManipulator

        self.urdf_name = str(urdf_name)
        t = extract_tables(self.urdf_name, tip_link)
        self.tables = t
        self.robot_data = {"M": t["M"], "omega_list": t["S_list"][:3, :], "Slist": t["S_list"], "Blist": t["B_list"],
                           "Glist": t["G_list"], "actuated_joints_num": t["S_list"].shape[1],
                           "joint_limits": [tuple(r) for r in t["joint_limits"]], "Mlist_per_link": t["Mlist_per_link"]}
        d = self.robot_data
        self.serial_manipulator = SerialManipulator(M_list=d["M"], omega_list=d["omega_list"], S_list=d["Slist"],
                                                    B_list=d["Blist"], G_list=d["Glist"], joint_limits=d["joint_limits"])
        self.dynamics = ManipulatorDynamics(M_list=d["M"], omega_list=d["omega_list"], r_list=t["r_list"], b_list=None,
                                            S_list=d["Slist"], B_list=d["Blist"], Glist=d["Glist"],
                                            Mlist_per_link=d["Mlist_per_link"])
        self.manipulator_dynamics = self.dynamics
        self._tree = t["_tree"]
        self._cfg: Dict[str, float] = {}   # the "current configuration" link_fk(None) / get_transform(cfg=None) use

    # ---- convenience surface of the reference's processor (urdf_processor.py:140-168, :363-617); the tree walks are the
    #      reference's urdf/core.py:498-667 (update_cfg / link_fk / link_fk_batch / get_transform) on this module's parse
    @staticmethod
    def transform_to_xyz(T) -> np.ndarray:
        return np.array(np.asarray(T)[0:3, 3])

    @staticmethod
    def w_p_to_slist(w, p, robot_dof: int) -> np.ndarray:
        """Rows w_i, p_i -> (6, robot_dof) screw axes [w; -w x p]."""
        w, p = np.asarray(w, dtype=np.float64), np.asarray(p, dtype=np.float64)
        return np.transpose([np.concatenate([w[i], np.cross(-w[i], p[i])]) for i in range(robot_dof)])

    def get_link(self, robot=None, link_name: Optional[str] = None):
        """The parsed link record by name, None if absent.  (The reference's is a static (robot, link_name) helper; both call
        shapes are accepted: get_link(name) and get_link(processor, name).)"""
        name = link_name if link_name is not None else robot
        return self._tree["links"].get(name)

    @property
    def num_dofs(self) -> int:
        return len(self._tree["actuated"])

    @property
    def joint_names(self) -> List[str]:
        return [j.name for j in self._tree["actuated"]]

    @property
    def link_names(self) -> List[str]:
        return list(self._tree["links"])

    @property
    def end_effector_name(self) -> str:
        return self.tables["ee_name"]

    @property
    def joint_limits_array(self) -> np.ndarray:
        return np.array(self.tables["joint_limits"], dtype=np.float64)

    def print_joint_info(self) -> Dict[str, object]:
        names = [j.name for j in self._tree["joints"]]
        return {"num_joints": len(names), "joint_names": names}

    def load_urdf(self, urdf_name: str) -> Dict[str, object]:
        import warnings

        warnings.warn("load_urdf() is deprecated. Use _extract_robot_data() instead.", DeprecationWarning, stacklevel=2)
        return self.robot_data

    def initialize_serial_manipulator(self):
        return self.serial_manipulator

    def initialize_manipulator_dynamics(self):
        return self.dynamics

    def get_serial_manipulator(self):
        """A fresh SerialManipulator from the tables (reference urdf/core.py:771-795)."""
        from .kinematics import SerialManipulator

        d = self.robot_data
        return SerialManipulator(M_list=d["M"], omega_list=d["omega_list"], S_list=d["Slist"], B_list=d["Blist"], G_list=d["Glist"],
                                 joint_limits=d["joint_limits"])

    def get_manipulator_dynamics(self):
        """The reference's URDF.to_manipulator_dynamics() (urdf/core.py:797-817) passes no Mlist_per_link: the LEGACY object."""
        from .dynamics import ManipulatorDynamics

        d = self.robot_data
        return ManipulatorDynamics(M_list=d["M"], omega_list=d["omega_list"], r_list=self.tables["r_list"], b_list=None,
                                   S_list=d["Slist"], B_list=d["Blist"], Glist=d["Glist"])

    def forward_kinematics(self, cfg, frame: str = "space") -> np.ndarray:
        return self.serial_manipulator.forward_kinematics(cfg, frame=frame)

    def jacobian(self, cfg, frame: str = "space") -> np.ndarray:
        return self.serial_manipulator.jacobian(cfg, frame=frame)

    def inverse_kinematics(self, T_desired, initial_guess=None, method: str = "robust", **kwargs):
        """(theta, success, iterations) by "robust" (default), "smart" or "iterative" inverse kinematics."""
        if initial_guess is None:
            initial_guess = np.zeros(self.num_dofs)
        if method == "robust":
            theta, ok, iters, _ = self.serial_manipulator.robust_inverse_kinematics(T_desired, **kwargs)
            return theta, ok, iters
        if method == "smart":
            return self.serial_manipulator.smart_inverse_kinematics(T_desired, **kwargs)
        return self.serial_manipulator.iterative_inverse_kinematics(T_desired, initial_guess, **kwargs)

    def _update_cfg(self, cfg) -> None:
        if cfg is None:
            return
        if isinstance(cfg, dict):
            self._cfg.update({k: float(v) for k, v in cfg.items()})
            return
        arr = np.asarray(cfg, dtype=np.float64).flatten()
        if len(arr) != self.num_dofs:
            raise ValueError(f"Configuration length {len(arr)} != num_actuated_joints {self.num_dofs}")
        for j, v in zip(self._tree["actuated"], arr):
            self._cfg[j.name] = float(v)

    def _joint_value(self, j: _Joint) -> float:
        if j.mimic is not None:
            return self._cfg.get(j.mimic[0], 0.0) * j.mimic[1] + j.mimic[2]
        return self._cfg.get(j.name, 0.0)

    def link_fk(self, cfg=None, use_names: bool = True) -> Dict[str, np.ndarray]:
        """Pose of EVERY link in the world frame at `cfg` (array in actuated-joint order, {joint name: value}, or None for the
        configuration of the last call, which persists like the reference's).  Mimic joints follow their master."""
        del use_names
        self._update_cfg(cfg)
        out = {r: np.eye(4) for r in self._tree["roots"]}
        for j in self._tree["chain"]:
            out[j.child] = out.get(j.parent, np.eye(4)) @ j.child_pose(self._joint_value(j))
        return out

    def batch_forward_kinematics(self, cfgs, link_name: Optional[str] = None):
        """All-link forward kinematics for N configurations: {link: (N,4,4)}, or the (N,4,4) of `link_name`.  For the end
        effector of a plain serial chain this is the batched kinematics launch (kinematics.fk_jacobian); other links - and
        trees whose screw model differs from the tree walk: actuated joints off the tip's path (branches; the product of
        exponentials moves the tip with them, the tree does not) or mimic joints (the screw model holds them still) - walk
        the tree vectorised, like the reference's link_fk_batch."""
        cfgs = np.asarray(cfgs, dtype=np.float64)
        if cfgs.ndim == 1:
            cfgs = cfgs.reshape(1, -1)
        if cfgs.shape[1] != self.num_dofs:
            raise ValueError(f"Configuration columns {cfgs.shape[1]} != num_actuated_joints {self.num_dofs}")
        known = set(self._tree["roots"]) | {j.child for j in self._tree["chain"]}
        if link_name is not None and link_name not in known:
            raise ValueError(f"Unknown link: {link_name}. Available: {sorted(known)}")
        if link_name is not None and link_name == self.end_effector_name and self._serial_to_tip():
            return self.serial_manipulator.forward_kinematics(cfgs)
        n = cfgs.shape[0]
        col = {j.name: i for i, j in enumerate(self._tree["actuated"])}
        out = {r: np.tile(np.eye(4), (n, 1, 1)) for r in self._tree["roots"]}
        for j in self._tree["chain"]:
            if j.mimic is not None:
                q = cfgs[:, col.get(j.mimic[0], 0)] * j.mimic[1] + j.mimic[2]
            elif j.name in col:
                q = cfgs[:, col[j.name]]
            else:
                q = np.zeros(n)
            parent = out.get(j.parent)
            out[j.child] = np.matmul(parent if parent is not None else np.tile(np.eye(4), (n, 1, 1)), j.child_poses(q))
        return out[link_name] if link_name is not None else out

    def _serial_to_tip(self) -> bool:
        """True when the screw model and the tree agree on the tip: every actuated joint lies on the root -> tip path and no
        joint of the tree is a mimic joint."""
        by_child = {j.child: j for j in self._tree["chain"]}
        path, cur = set(), self.end_effector_name
        while cur in by_child:
            path.add(by_child[cur].name)
            cur = by_child[cur].parent
        return all(j.name in path for j in self._tree["actuated"]) and not any(j.mimic for j in self._tree["chain"])

    def get_end_effector_transforms(self, cfgs) -> np.ndarray:
        return self.batch_forward_kinematics(cfgs, link_name=self.end_effector_name)

    def get_transform(self, frame_to: str, frame_from: str = "world", cfg=None) -> np.ndarray:
        """Pose of link `frame_to` expressed in `frame_from` ("world" or a link name)."""
        fk = self.link_fk(cfg)
        if frame_to not in fk:
            raise ValueError(f"Unknown frame: {frame_to}")
        if frame_from == "world":
            return fk[frame_to]
        if frame_from not in fk:
            raise ValueError(f"Unknown frame: {frame_from}")
        return np.linalg.inv(fk[frame_from]) @ fk[frame_to]

    def validate(self) -> Dict[str, object]:
        """{"valid": bool, "issues": [{"severity", "message"}]}: the structural checks of the reference's validator
        (urdf/validation.py:119-326: missing links, several parents, no / several roots, disconnected links, cycles, zero
        axes, missing or empty limits, empty mimic references) on this module's parse; ERRORs make it invalid."""
        links, joints = self._tree["links"], self._tree["joints"]
        issues: List[Dict[str, str]] = []
        add = lambda sev, msg: issues.append({"severity": sev, "message": msg})  # noqa: E731
        if not links:
            return {"valid": False, "issues": [{"severity": "ERROR", "message": "URDF has no links"}]}
        parent_of: Dict[str, str] = {}
        kids: Dict[str, List[str]] = {}
        for j in joints:
            if j.parent not in links:
                add("ERROR", f"Parent link '{j.parent}' not found")
            if j.child not in links:
                add("ERROR", f"Child link '{j.child}' not found")
            if j.child in parent_of:
                add("ERROR", f"Link '{j.child}' has multiple parents: '{parent_of[j.child]}' and '{j.parent}'")
            else:
                parent_of[j.child] = j.parent
            kids.setdefault(j.parent, []).append(j.child)
        roots = [name for name in links if name not in parent_of]
        if not roots:
            add("ERROR", "No root link found - possible cycle in kinematic tree")
        elif len(roots) > 1:
            add("WARNING", f"Multiple root links found: {roots}. Only the first will be used for kinematics.")
        if roots:
            reach, stack = set(), [roots[0]]
            while stack:
                cur = stack.pop()
                if cur not in reach:
                    reach.add(cur)
                    stack.extend(kids.get(cur, []))
            lost = set(links) - reach
            if lost:
                add("WARNING", f"Disconnected links found: {sorted(lost)}")
        for start in links:   # a link that reaches itself by walking up its parents
            seen, cur = [start], parent_of.get(start)
            while cur is not None and cur not in seen:
                seen.append(cur)
                cur = parent_of.get(cur)
            if cur is not None:
                cyc = seen[seen.index(cur):] + [cur]
                add("ERROR", f"Cycle detected in kinematic tree: {' -> '.join(cyc)}")
                break
        for j in joints:
            norm = float(np.linalg.norm(j.axis))
            if abs(norm - 1.0) > 1e-6:
                add("WARNING", f"Joint axis not normalized (norm={norm:.6f})")
            if j.kind in ("revolute", "prismatic"):
                if j.limit is None:
                    add("WARNING", "Joint has no limits defined")
                elif j.limit[0] >= j.limit[1]:
                    add("WARNING", f"Joint limits invalid: lower ({j.limit[0]}) >= upper ({j.limit[1]})")
            if j.mimic is not None and not j.mimic[0]:
                add("ERROR", "Mimic joint reference is empty")
        return {"valid": not any(i["severity"] == "ERROR" for i in issues), "issues": issues}

    def __repr__(self) -> str:
        return f"URDFToSerialManipulator(urdf='{self.urdf_name}', dofs={self.num_dofs}, backend='builtin')"
