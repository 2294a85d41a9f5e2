"""Mesh obstacles: host-side mirror of the mesh entry points of the C ABI.

``Mesh`` owns a device mesh + hierarchy (``cfs_mesh_*``); ``point2surface_dis`` / ``dist_arm_surf`` keep the names
the reference calls (M200i/dist_arm_surf_200i.m:21, Lib/functions/dist_arm_surface.m:43).  The generators and the
binary-STL reader / writer below are plain numpy: they make the synthetic maps of the tests and of config 5
(the reference's own maps, map/*.STL, are binary STL files in millimetres and load through ``Mesh.from_stl``).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from .robotproperty2 import to_c_robot

_REC = np.dtype([("n", "<3f4"), ("v", "<9f4"), ("a", "<u2")])


# ---- binary STL --------------------------------------------------------------------------------------
def read_stl_binary(path):
    """(nt, 3, 3) float64 triangle soup of a binary STL (80-byte header, uint32 count, 50-byte records)."""
    with open(path, "rb") as f:
        head = f.read(84)
        if len(head) < 84:
            raise ValueError(f"{path}: shorter than an STL header")
        nt = int(np.frombuffer(head[80:84], "<u4")[0])
        rec = np.fromfile(f, dtype=_REC, count=nt)
    if rec.shape[0] != nt or nt < 1:
        raise ValueError(f"{path}: not a binary STL")
    return rec["v"].astype(np.float64).reshape(nt, 3, 3)


def write_stl_binary(path, tri, title="motionplanning_5d_m_amd"):
    tri = np.asarray(tri, np.float64).reshape(-1, 3, 3)
    rec = np.zeros(tri.shape[0], dtype=_REC)
    n = np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0])
    ln = np.linalg.norm(n, axis=1, keepdims=True)
    rec["n"] = (n / np.where(ln > 0, ln, 1.0)).astype(np.float32)
    rec["v"] = tri.reshape(-1, 9).astype(np.float32)
    with open(path, "wb") as f:
        f.write(title.encode()[:80].ljust(80, b" "))
        f.write(np.uint32(tri.shape[0]).tobytes())
        rec.tofile(f)


def map_from_stl(tri, vmin=None):
    """Lib/functions/MapFromSTL.m:6-10: every axis shifted to start at 0, y -= 100, then (x, y, z) <- (z, x, y).
    vmin: the column minima to shift by (default: those of `tri`; a crop of a map passes the minima of the whole map)."""
    v = np.asarray(tri, np.float64).reshape(-1, 3).copy()
    v -= v.min(axis=0) if vmin is None else np.asarray(vmin, np.float64)
    v[:, 1] -= 100.0
    return v[:, [2, 0, 1]].reshape(-1, 3, 3)


def load_map_fixture(path):
    """(nt, 3, 3) float64 triangles in metres from a fixture written by tests/golden/make_reference_map.py: raw float32 STL
    vertices of a crop of one of the reference's maps + the raw column minima of the whole file; MapFromSTL.m:6-10 and the
    mm -> m scale are applied here in float64, exactly as cfs_mesh_load_stl(path, scale, 1) does on the file itself."""
    d = np.load(path, allow_pickle=False)
    return map_from_stl(d["tri_raw"].astype(np.float64), vmin=d["vmin_raw"].astype(np.float64)) * float(d["scale"])


# ---- synthetic maps ----------------------------------------------------------------------------------
def _grid_quad(p0, du, dv, nu, nv):
    """nu x nv grid of quads on the parallelogram p0 + s*du + t*dv, two triangles each."""
    s, t = np.meshgrid(np.arange(nu + 1) / nu, np.arange(nv + 1) / nv, indexing="ij")
    P = p0 + s[..., None] * du + t[..., None] * dv
    a, b, c, d = P[:-1, :-1], P[1:, :-1], P[1:, 1:], P[:-1, 1:]
    return np.concatenate([np.stack([a, b, c], -2).reshape(-1, 3, 3), np.stack([a, c, d], -2).reshape(-1, 3, 3)])


def box_mesh(lo, hi, n=4):
    """Axis-aligned box surface, n x n quads per face (12 n^2 triangles)."""
    lo, hi = np.asarray(lo, float), np.asarray(hi, float)
    e = np.diag(hi - lo)
    faces = []
    for ax in range(3):
        u, v = e[(ax + 1) % 3], e[(ax + 2) % 3]
        faces.append(_grid_quad(lo, u, v, n, n))
        faces.append(_grid_quad(lo + e[ax], v, u, n, n))
    return np.concatenate(faces)


def cylinder_mesh(center, radius, z0, z1, nseg=32, nring=8):
    """Closed vertical cylinder around (center x, y): nseg x nring side quads + two fans."""
    cx, cy = center
    a = np.arange(nseg + 1) * (2 * np.pi / nseg)
    ring = np.stack([cx + radius * np.cos(a), cy + radius * np.sin(a)], 1)
    z = np.linspace(z0, z1, nring + 1)
    tri = []
    for i in range(nseg):
        for k in range(nring):
            p00, p10 = [*ring[i], z[k]], [*ring[i + 1], z[k]]
            p01, p11 = [*ring[i], z[k + 1]], [*ring[i + 1], z[k + 1]]
            tri += [[p00, p10, p11], [p00, p11, p01]]
        tri += [[[cx, cy, z0], [*ring[i + 1], z0], [*ring[i], z0]], [[cx, cy, z1], [*ring[i], z1], [*ring[i + 1], z1]]]
    return np.asarray(tri, float)


def icosphere(center, radius, subdiv=3):
    """Geodesic sphere: 20 * 4^subdiv triangles."""
    t = (1 + 5 ** 0.5) / 2
    v = np.array([[-1, t, 0], [1, t, 0], [-1, -t, 0], [1, -t, 0], [0, -1, t], [0, 1, t], [0, -1, -t], [0, 1, -t],
                  [t, 0, -1], [t, 0, 1], [-t, 0, -1], [-t, 0, 1]], float)
    f = np.array([[0, 11, 5], [0, 5, 1], [0, 1, 7], [0, 7, 10], [0, 10, 11], [1, 5, 9], [5, 11, 4], [11, 10, 2], [10, 7, 6],
                  [7, 1, 8], [3, 9, 4], [3, 4, 2], [3, 2, 6], [3, 6, 8], [3, 8, 9], [4, 9, 5], [2, 4, 11], [6, 2, 10],
                  [8, 6, 7], [9, 8, 1]])
    tri = v[f]
    for _ in range(subdiv):
        a, b, c = tri[:, 0], tri[:, 1], tri[:, 2]
        ab, bc, ca = (a + b) / 2, (b + c) / 2, (c + a) / 2
        tri = np.concatenate([np.stack([a, ab, ca], 1), np.stack([b, bc, ab], 1), np.stack([c, ca, bc], 1), np.stack([ab, bc, ca], 1)])
    tri = tri / np.linalg.norm(tri, axis=2, keepdims=True)
    return np.asarray(center, float) + radius * tri


def assembly_line(base, n_target=10000, seed=0):
    """Synthetic stand-in for the reference's assembly-line map (map/assembly line_Assem1.STL, 27 396 triangles, in
    mm, unusable on a box without /root/reference): a conveyor table, two posts, a gantry beam and a few round parts
    around the robot base, about n_target triangles in metres."""
    rng = np.random.default_rng(seed)
    bx, by, bz = base
    parts = []
    k = max(2, int(round((max(n_target - 1200, 84) / 84.0) ** 0.5)))
    parts.append(box_mesh([bx + 0.55, by - 0.9, bz - 0.33], [bx + 1.05, by + 0.9, bz + 0.12], n=2 * k))        # conveyor
    parts.append(box_mesh([bx + 0.50, by - 0.95, bz - 0.33], [bx + 0.58, by - 0.87, bz + 1.25], n=k))           # post
    parts.append(box_mesh([bx + 0.50, by + 0.87, bz - 0.33], [bx + 0.58, by + 0.95, bz + 1.25], n=k))           # post
    parts.append(box_mesh([bx + 0.50, by - 0.95, bz + 1.25], [bx + 0.58, by + 0.95, bz + 1.33], n=k))           # beam
    for _ in range(3):
        c = [bx + rng.uniform(0.65, 0.95), by + rng.uniform(-0.7, 0.7), bz + 0.12 + 0.08]
        parts.append(icosphere(c, 0.08, subdiv=2))
    parts.append(cylinder_mesh((bx + 0.8, by + 0.35), 0.06, bz + 0.12, bz + 0.42, nseg=24, nring=4))
    return np.concatenate(parts)


# ---- device mesh ---------------------------------------------------------------------------------------
class Mesh:
    """A triangle mesh resident on the GPU with its hierarchy (cfs_mesh_create / cfs_mesh_load_stl)."""

    def __init__(self, tri=None, vertices=None, faces=None, device=None):
        lib = _lib.lib()
        if device is not None:
            _lib.check(lib.cfs_set_device(int(device)))
        if tri is not None:
            tri = np.ascontiguousarray(tri, np.float64).reshape(-1, 3, 3)
            vertices, faces = tri.reshape(-1, 3), np.arange(3 * tri.shape[0], dtype=np.int32).reshape(-1, 3)
        vertices = np.ascontiguousarray(vertices, np.float64)
        faces = np.ascontiguousarray(faces, np.int32)
        h = C.c_void_p()
        _lib.check(lib.cfs_mesh_create(vertices.ctypes.data_as(C.c_void_p), vertices.shape[0],
                                       faces.ctypes.data_as(C.c_void_p), faces.shape[0], C.byref(h)))
        self._h, self._lib = h, lib

    @classmethod
    def from_stl(cls, path, scale=1.0, map_from_stl=False, device=None):
        lib = _lib.lib()
        if device is not None:
            _lib.check(lib.cfs_set_device(int(device)))
        self = cls.__new__(cls)
        h = C.c_void_p()
        _lib.check(lib.cfs_mesh_load_stl(str(path).encode(), float(scale), 1 if map_from_stl else 0, C.byref(h)))
        self._h, self._lib = h, lib
        return self

    def info(self):
        nt, nn, dp = C.c_int(0), C.c_int(0), C.c_int(0)
        bb = (C.c_double * 6)()
        _lib.check(self._lib.cfs_mesh_info(self._h, C.byref(nt), C.byref(nn), C.byref(dp), bb))
        return dict(ntri=nt.value, nnodes=nn.value, depth=dp.value, bbox=np.array(bb[:]))

    def point2surface_dis(self, segs):
        """[dis, points, tri] for (n, 6) link axes [p(:,1); p(:,2)]."""
        segs = np.ascontiguousarray(np.atleast_2d(segs), np.float64)
        n = segs.shape[0]
        dis, pts, tri = np.zeros(n), np.zeros((n, 6)), np.zeros(n, np.int32)
        p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
        _lib.check(self._lib.cfs_mesh_segment_distance(self._h, n, p(segs), p(dis), p(pts), p(tri)))
        return dis, pts, tri

    def close(self):
        if getattr(self, "_h", None):
            self._lib.cfs_mesh_destroy(self._h)
            self._h = None

    __del__ = close


def dist_arm_surf(robot, theta, mesh):
    """[d, linkid, points] = dist_arm_surf_200i(theta, base, mesh, robot) for (N, nj) configurations
    (M200i/dist_arm_surf_200i.m:1-29; the robot model decides the joint offset as in dist_arm)."""
    theta = np.ascontiguousarray(np.atleast_2d(theta), np.float64)
    N, nj = theta.shape
    d, lid, pts = np.zeros(N), np.zeros(N, np.int32), np.zeros((N, 6))
    rb = to_c_robot(robot)
    p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    _lib.check(_lib.lib().cfs_dist_arm_mesh(C.byref(rb), nj, N, p(theta), mesh._h, p(d), p(lid), p(pts)))
    return d, lid, pts
