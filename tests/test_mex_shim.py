"""The MATLAB side of the boundary (matlab/cfs_mex.cpp + the replacement classdefs) cannot be built or run here (no MATLAB, no
mex.h).  What CAN be checked without MATLAB: the gateway is valid C++ against the C ABI header and a declarations-only
stand-in of the documented MEX API (tests/stubs/mex.h), it uses only entry points include/cfs_hip.h declares, and the
classdefs keep the reference's class / method / property names."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_mex_gateway_is_valid_cpp_against_the_abi_header():
    r = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "tests", "stubs"),
                        "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "matlab", "cfs_mex.cpp")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_mex_gateway_calls_only_declared_entry_points():
    src = open(os.path.join(ROOT, "matlab", "cfs_mex.cpp")).read()
    hdr = open(os.path.join(ROOT, "include", "cfs_hip.h")).read()
    used = set(re.findall(r"\b(cfs_[a-z_]+)\s*\(", src)) - {"cfs_mex"}          # cfs_mex(...) appears in the usage comments
    declared = set(re.findall(r"\b(cfs_[a-z_]+)\s*\(", hdr))
    assert used and used <= declared, used - declared
    for cmd in ("solve", "get_con", "chomp", "dist_arm", "rrt", "cost_b", "mesh_load_stl", "mesh_segment_distance", "mesh_destroy"):
        assert f'"{cmd}"' in src


def test_classdefs_keep_the_reference_interface():
    # Lib/CFS_FANUC.m:40,62,101  Lib/PSGCFS_FANUC.m:43,65,145  Lib/CHOMP_FANUC.m:34,54: constructor, optimizer, get_con; the
    # result properties the drivers read (main_FANUC.m:144-162, RRTstar_CFS.m:197-203)
    for name, methods in (("CFS_FANUC", ("optimizer", "get_con")), ("PSGCFS_FANUC", ("optimizer", "get_con")), ("CHOMP_FANUC", ("optimizer",))):
        txt = open(os.path.join(ROOT, "matlab", name + ".m")).read()
        assert re.search(r"classdef\s+" + name + r"\b", txt)
        assert re.search(r"function\s+self\s*=\s*" + name + r"\(", txt)
        for m in methods:
            assert re.search(r"function\s+self\s*=\s*" + m + r"\(self", txt), (name, m)
        for prop in ("obs", "sys_info", "ROBOT", "u", "x_", "eval", "iter_O", "total_iter"):
            assert re.search(r"\b" + prop + r"\b", txt), (name, prop)
        assert "cfs_mex(" in txt
    for name in ("CFS_FANUC", "PSGCFS_FANUC"):
        txt = open(os.path.join(ROOT, "matlab", name + ".m")).read()
        assert "Ainq" in txt and "binq" in txt
    # RRT_FANUC: constructor and find_route with the reference's signature and outputs (Lib/RRT_FANUC.m:48,63)
    txt = open(os.path.join(ROOT, "matlab", "RRT_FANUC.m")).read()
    assert re.search(r"function\s+self\s*=\s*RRT_FANUC\(val,\s*val2,\s*val3,\s*val4,\s*val5,\s*val6,\s*varargin\)", txt)
    assert re.search(r"function\s+self\s*=\s*find_route\(self\)", txt) and "cfs_mex('rrt'" in txt
    for prop in ("route", "all_nodes", "total_dis", "all_ee", "fail", "node_num", "MAX_ITER", "bi"):
        assert re.search(r"\b" + prop + r"\b", txt), prop
    # the geometry primitive under its reference name and signature (Lib/200i/dist_arm_3D_200i_2.m:1)
    txt = open(os.path.join(ROOT, "matlab", "dist_arm_3D_200i_2.m")).read()
    assert re.search(r"function\s+\[d,\s*linkid\]\s*=\s*dist_arm_3D_200i_2\(theta,\s*base,\s*obs,\s*robot\)", txt) and "cfs_mex('dist_arm'" in txt
