"""Makes tests/golden/assembly_line_cell.npz: the part of the REFERENCE's own map that surrounds the robot, as a data fixture.

Run in the build container (the reference does not exist on the GPU box):  python tests/golden/make_reference_map.py

Source: /root/reference/map/assembly line_Assem1.STL (binary STL, 27 396 triangles, millimetres) -- the file
Lib/functions/MapFromSTL.m:3 names and map/environment.mat holds in transformed form.  Transformed the way MapFromSTL.m:6-10
does (every axis shifted to start at 0, y -= 100, (x, y, z) <- (z, x, y)) and scaled mm -> m it is the cell the M200i stands
in: the floor at z = 0 under robot.base = [3.150, 8.500, 0.330] (robotproperty2.m:54-55) and the assembly line 0.75 m in front
of it.  BASELINE.json's config 5 is "M200i assembly-line STL mesh (~10k triangles)": the fixture keeps the 13 258 triangles with
a vertex within 2.5 m of robot.base (the arm reaches 0.9 m; nothing farther can ever be the closest triangle of a link).

What is stored is DATA, bit for bit as the STL holds it: the raw float32 vertices of the kept triangles and the three raw
column minima of the whole file (MapFromSTL's shift is relative to them), so that mesh.load_map_fixture reproduces exactly what
cfs_mesh_load_stl(path, 1e-3, map_from_stl = 1) computes for these triangles from the file itself.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from motionplanning_5d_m_amd import mesh  # noqa: E402

SRC = "/root/reference/map/assembly line_Assem1.STL"
BASE = np.array([3.150, 8.500, 0.330])          # robotproperty2.m:54-55
RADIUS = 2.5

raw = mesh.read_stl_binary(SRC)                   # (nt, 3, 3) float64 of float32 data
vmin = raw.reshape(-1, 3).min(axis=0)
tri_m = mesh.map_from_stl(raw, vmin=vmin) * 1e-3
keep = np.linalg.norm(tri_m - BASE, axis=2).min(axis=1) < RADIUS
raw32 = raw[keep].astype(np.float32)
assert np.array_equal(raw32.astype(np.float64), raw[keep]) and np.array_equal(vmin.astype(np.float32).astype(np.float64), vmin)
out = os.path.join(ROOT, "tests", "golden", "assembly_line_cell.npz")
np.savez_compressed(out, tri_raw=raw32, vmin_raw=vmin.astype(np.float32), scale=np.float64(1e-3), radius=np.float64(RADIUS),
                    source=np.array("map/assembly line_Assem1.STL: %d of %d triangles with a vertex within %.1f m of robot.base after "
                                    "MapFromSTL.m:6-10 and mm -> m" % (int(keep.sum()), raw.shape[0], RADIUS)))
print(out, int(keep.sum()), "triangles,", os.path.getsize(out), "bytes")
