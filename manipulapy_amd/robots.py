"""Robot model tables for the benchmark robots.

The reference builds these tables from URDFs (urdf_processor.py:82-138 -> urdf/core.py:670-769).
`manipulapy_amd.urdf` reads any URDF the same way; this module additionally ships, under manipulapy_amd/data/, the
tables of the four configuration robots as small .npz files captured from the reference (tests/golden/make_golden.py,
numbers only) — the pin the URDF reader is tested against, and what bench.py loads — next to the four URDF files
themselves (data/urdf/, robot description data).
"""
from __future__ import annotations

import os
from typing import Dict, Tuple

import numpy as np

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")
ROBOTS = ("ur5", "iiwa14", "panda", "xarm6")
# "panda7": the 7 arm joints of the Panda.  The reference parses the packaged Panda URDF as EIGHT actuated joints (the
# arm + one prismatic finger joint), which is what "panda" reproduces; BASELINE names a 7-DOF Panda, so the
# first-seven-joint truncation of the same tables is offered next to it.  Tables only: it has
# no URDF of its own.
DERIVED = {"panda7": ("panda", 7)}


def robot_tables(name: str) -> Dict[str, np.ndarray]:
    """S_list (6,n), M_ee (4,4), Glist (n,6,6), Mlist_per_link (n,4,4), joint_limits (n,2), B_list (6,n)."""
    if name in DERIVED:
        base, n = DERIVED[name]
        t = robot_tables(base)
        return {"S_list": np.ascontiguousarray(t["S_list"][:, :n]), "B_list": np.ascontiguousarray(t["B_list"][:, :n]),
                "M_ee": t["M_ee"], "Glist": np.ascontiguousarray(t["Glist"][:n]),
                "Mlist_per_link": np.ascontiguousarray(t["Mlist_per_link"][:n]),
                "joint_limits": np.ascontiguousarray(t["joint_limits"][:n])}
    if name not in ROBOTS:
        raise KeyError(f"unknown robot {name!r}; available: {', '.join(ROBOTS + tuple(DERIVED))}")
    z = np.load(os.path.join(_DATA, f"model_{name}.npz"))
    return {k: z[k] for k in ("S_list", "B_list", "M_ee", "Glist", "Mlist_per_link", "joint_limits")}


def robot_urdf(name: str) -> str:
    """Path of the URDF of one of the benchmark robots (the counterpart of the reference's
    ManipulaPy_data.get_robot_urdf for these four)."""
    if name not in ROBOTS:
        raise KeyError(f"unknown robot {name!r}; available: {', '.join(ROBOTS)}")
    return os.path.join(_DATA, "urdf", f"{name}.urdf")


def load_robot(name: str) -> Tuple["SerialManipulator", "ManipulatorDynamics", np.ndarray]:
    """(serial_manipulator, dynamics, joint_limits) built like URDFToSerialManipulator(...) would
    (reference urdf_processor.py:264-302)."""
    from .dynamics import ManipulatorDynamics
    from .kinematics import SerialManipulator

    t = robot_tables(name)
    sm = SerialManipulator(M_list=t["M_ee"], omega_list=t["S_list"][:3], S_list=t["S_list"], B_list=t["B_list"],
                           G_list=t["Glist"], joint_limits=[tuple(r) for r in t["joint_limits"]])
    dyn = ManipulatorDynamics(M_list=t["M_ee"], omega_list=t["S_list"][:3], r_list=None, b_list=None,
                              S_list=t["S_list"], B_list=t["B_list"], Glist=t["Glist"],
                              Mlist_per_link=t["Mlist_per_link"])
    return sm, dyn, t["joint_limits"]
