"""Host-side mirror of the reference's robot model loader.

``robot = robotproperty2(id)`` returns the fields of the reference struct that the CFS path reads
(reference: Lib/functions/robotproperty2.m:1-153): ``name, nlink, delta_t, DH, base, cap, A, B``
(+ ``T`` for the two-link arm).  Constants are the reference's literals (DH alphas are 1.5708 /
3.1416, not pi/2 / pi).  ``to_c_robot`` packs a robot into the C ABI struct ``cfs_robot``.
"""
from __future__ import annotations

from types import SimpleNamespace

import numpy as np

from . import _lib

_MODELS = {
    # id: (nlink, DH rows [theta d a alpha], capsule axis end points per link (p1, p2), radii, base offset in mm)
    "M200i": dict(  # robotproperty2.m:12-55
        DH=[(0, 0, 0.050, -1.5708), (-1.5708, 0, 0.440, 3.1416), (0, 0, 0.035, -1.5708),
            (0, -0.420, 0, 1.5708), (0, 0, 0, -1.5708), (0, -0.080, 0, 3.1416)],
        cap=[((0, 0, 0), (0, 0, 0)), ((-0.4, 0, 0), (0, 0, 0)), ((-0.03, 0, 0.05), (-0.03, 0, 0.05)),
             ((0, 0, 0), (0, 0.4, 0)), ((0, 0, -0.26), (0, 0, 0.01)), ((0.05, 0, 0.1107), (0.18, 0, 0.1107))],
        r=[0, 0.13, 0, 0.068, 0.01, 0.06], offset_mm=(3150, 8500, 330)),
    "M16iB": dict(  # robotproperty2.m:58-99
        DH=[(0.5, 0.65, 0.15, 1.5708), (1.5708, 0, 0.77, 0), (0, 0, 0.1, 1.5708),
            (0, 0.74, 0, -1.5708), (-np.pi / 2, 0, 0, 1.5708), (np.pi, 0.1, 0, 0)],
        cap=[((0, 0, -0.1), (0, 0, 0.1)), ((-0.75, 0, -0.15), (0, 0, -0.15)), ((-0.03, 0, 0.05), (-0.03, 0, 0.05)),
             ((0, 0, 0), (0, 0.55, 0)), ((0, 0, -0.05), (0, 0, 0.110)), ((-0.11, 0, 0.09), (-0.11, 0, 0.09))],
        r=[0.15, 0.13, 0.22, 0.11, 0.07, 0.11], offset_mm=(3250, 8500, 0)),
    "2L": dict(  # robotproperty2.m:102-130
        DH=[(0, 0, 0.3, 0), (0, 0, 0.2, 0), (0, 0, 0, 0)],
        cap=[((0, 0, 0), (0.3, 0, 0)), ((0, 0, 0), (0.2, 0, 0))],
        r=[0.05, 0.05], offset_mm=(0, 0, 0)),
}


def robotproperty2(rid: str) -> SimpleNamespace:
    if rid not in _MODELS:
        raise ValueError(f"unknown robot id {rid!r} (expected one of {sorted(_MODELS)})")
    m = _MODELS[rid]
    robot = SimpleNamespace(name=rid)
    robot.DH = np.array(m["DH"], dtype=np.float64)
    robot.nlink = robot.DH.shape[0]
    robot.delta_t = 0.5
    robot.cap = [SimpleNamespace(p=np.array([p1, p2], dtype=np.float64).T.copy(), r=r)
                 for (p1, p2), r in zip(m["cap"], m["r"])]
    robot.base = np.array(m["offset_mm"], dtype=np.float64) / 1000
    robot.T = np.zeros((3, 3))
    if rid == "2L":
        robot.T[0, 2] = 0.3  # robotproperty2.m:117-119
    n, dt = robot.nlink, robot.delta_t
    eye, zero = np.eye(n), np.zeros((n, n))
    robot.A = np.block([[eye, dt * eye], [zero, eye]])  # :136-137
    robot.B = np.vstack([0.5 * dt * dt * eye, dt * eye])  # :138-139
    return robot


def to_c_robot(robot) -> _lib.cfs_robot:
    rb = _lib.cfs_robot()
    rb.kind = _lib.ROBOT_KIND[robot.name]
    rb.nlink = int(robot.nlink)
    DH = np.asfortranarray(robot.DH, dtype=np.float64)
    flat = DH.reshape(-1, order="F")
    for k in range(flat.size):
        rb.DH[k] = float(flat[k])
    for r in range(3):
        rb.base[r] = float(np.asarray(robot.base).reshape(-1)[r])
    for i, cp in enumerate(robot.cap):
        p = np.asarray(cp.p if hasattr(cp, "p") else cp["p"], dtype=np.float64)
        for k in range(2):
            for r in range(3):
                rb.cap[i * 6 + k * 3 + r] = float(p[r, k])
    T = np.asarray(getattr(robot, "T", np.zeros((3, 3))), dtype=np.float64)
    for c in range(3):
        for r in range(3):
            rb.T[c * 3 + r] = float(T[r, c])
    rb.delta_t = float(robot.delta_t)
    return rb
