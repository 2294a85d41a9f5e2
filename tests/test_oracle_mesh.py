"""Mesh obstacles (row f3), CPU side: the brute-force oracle of the build's own point2surface_dis contract
(parity unpinned: the reference calls the function but does not contain it) is checked against an independent
dense-sampling estimate, analytic cases and its invariances; plus the binary-STL reader / MapFromSTL transform."""
import os

import numpy as np
import pytest

from motionplanning_5d_m_amd import mesh as M


def _closest_on_triangles(P, tri):
    """independent numpy statement: distance from points P (n,3) to triangles (m,3,3) by projecting on the plane and
    clamping in barycentric coordinates through the edge cases (brute force over the 7 Voronoi regions by sampling
    the three edges + interior projection)."""
    A, B, C = tri[:, 0], tri[:, 1], tri[:, 2]
    best = np.full(P.shape[0], np.inf)
    n = np.cross(B - A, C - A)
    for k in range(tri.shape[0]):
        nn = n[k] @ n[k]
        cand = []
        if nn > 0:
            t = ((P - A[k]) @ n[k]) / nn
            Q = P - t[:, None] * n[k]
            inside = ((np.cross(B[k] - A[k], Q - A[k]) @ n[k] >= 0) & (np.cross(C[k] - B[k], Q - B[k]) @ n[k] >= 0)
                      & (np.cross(A[k] - C[k], Q - C[k]) @ n[k] >= 0))
            cand.append(np.where(inside, np.abs(t) * np.sqrt(nn), np.inf))
        for a, b in ((A[k], B[k]), (B[k], C[k]), (C[k], A[k])):
            ab = b - a
            s = np.clip(((P - a) @ ab) / max(ab @ ab, 1e-300), 0, 1)
            cand.append(np.linalg.norm(P - (a + s[:, None] * ab), axis=1))
        best = np.minimum(best, np.min(cand, axis=0))
    return best


def test_segment_mesh_distance_against_dense_sampling(O):
    rng = np.random.default_rng(5)
    tri = np.concatenate([M.icosphere([0.2, -0.1, 0.3], 0.5, subdiv=1), M.box_mesh([1.0, 0.0, 0.0], [1.4, 0.6, 0.8], n=1)])
    O.mesh_register(1, tri)
    segs = rng.uniform(-1.2, 2.0, (60, 6))
    segs[:5, 3:] = segs[:5, :3]                                  # zero-length links (M200i caps 1 and 3 are points)
    dis, pts, tid = O.mesh_seg_distance(1, segs)
    s = np.linspace(0, 1, 2001)
    for i in range(segs.shape[0]):
        P = segs[i, :3] + s[:, None] * (segs[i, 3:] - segs[i, :3])
        ref = _closest_on_triangles(P, tri).min()
        step = np.linalg.norm(segs[i, 3:] - segs[i, :3]) / 2000
        assert dis[i] <= ref + 1e-12                              # the true minimum can only be below any sample
        assert dis[i] >= ref - step - 1e-12                       # and no further below than the sampling step
        if dis[i] > 0:                                            # the reported points realise the distance
            assert abs(np.linalg.norm(pts[i, :3] - pts[i, 3:]) - dis[i]) < 1e-12
            assert abs(_closest_on_triangles(pts[i:i + 1, :3], tri[tid[i]:tid[i] + 1])[0] - dis[i]) < 1e-9


def test_segment_mesh_analytic_and_ties(O):
    tri = M.box_mesh([0, 0, 0], [1, 1, 1], n=2)
    O.mesh_register(2, tri)
    segs = np.array([[2, .5, .5, 3, .5, .5], [.5, .5, .5, .6, .5, .5], [-1, .5, .5, 2, .5, .5], [2, 2, 2, 3, 3, 3],
                     [.5, .5, 1.5, .5, .5, 1.2]], float)
    dis, pts, _ = O.mesh_seg_distance(2, segs)
    np.testing.assert_allclose(dis, [1.0, 0.4, 0.0, 3 ** 0.5, 0.2], atol=1e-15)
    np.testing.assert_allclose(pts[2], [0, .5, .5, 0, .5, .5], atol=1e-15)   # a piercing link reports its FIRST crossing
    # the answer does not depend on the order of the triangles
    perm = np.random.default_rng(0).permutation(tri.shape[0])
    O.mesh_register(3, tri[perm])
    rs = np.random.default_rng(1).uniform(-0.5, 1.5, (200, 6))
    d0, p0, _ = O.mesh_seg_distance(2, rs)
    d1, p1, _ = O.mesh_seg_distance(3, rs)
    np.testing.assert_array_equal(d0, d1)
    np.testing.assert_allclose(p0[:, :3], p1[:, :3], rtol=0, atol=1e-14)   # (ties between neighbours resolve to 1 ulp)


def test_dist_arm_over_a_mesh_uses_the_surrogate_and_first_minimum(O):
    rb = O.robotproperty2("M200i")
    th = np.array([0.78, 0.03, 0.2, 0.14, -1.1])
    pos = O.arm_pos(rb, th)
    # a small sphere around the free end of link 5: only link 5 pierces it -> near-zero branch (dist_arm_surf_200i.m:22-24)
    tip = pos[4, 0]
    L5 = np.linalg.norm(pos[4, 1] - pos[4, 0])
    l = O.mesh_register(4, M.icosphere(tip, 0.05, subdiv=2))
    d, lid = O.dist_arm(rb, th, l)
    assert lid == 5 and abs(d + (L5 - 0.05)) < 2e-3              # = -|crossing point - pos{5}.p(:,2)|
    # far sphere: plain positive distance, gradient by the literal num_jac
    l = O.mesh_register(4, M.icosphere(tip + np.array([0.3, 0.0, 0.1]), 0.05, subdiv=2))
    d, lid = O.dist_arm(rb, th, l)
    assert lid == 5 and abs(d - (np.linalg.norm([0.3, 0.0, 0.1]) - 0.05)) < 2e-3
    g = O.num_jac_dist(rb, th, l)
    assert np.all(np.isfinite(g)) and np.linalg.norm(g) > 1e-3


def test_stl_round_trip_and_map_from_stl(tmp_path):
    tri = M.cylinder_mesh((0.3, -0.2), 0.1, 0.0, 0.5, nseg=12, nring=3)
    f = tmp_path / "c.stl"
    M.write_stl_binary(f, tri)
    back = M.read_stl_binary(f)
    assert back.shape == tri.shape
    np.testing.assert_allclose(back, tri.astype(np.float32).astype(np.float64), atol=0)
    assert f.stat().st_size == 84 + 50 * tri.shape[0]
    m = M.map_from_stl(tri)                                       # MapFromSTL.m:6-10
    v, w = tri.reshape(-1, 3), m.reshape(-1, 3)
    np.testing.assert_allclose(w[:, 0], v[:, 2] - v[:, 2].min(), atol=1e-15)
    np.testing.assert_allclose(w[:, 1], v[:, 0] - v[:, 0].min(), atol=1e-15)
    np.testing.assert_allclose(w[:, 2], v[:, 1] - v[:, 1].min() - 100.0, atol=1e-12)
    big = M.assembly_line([3.15, 8.5, 0.33], n_target=10000)
    assert 8000 < big.shape[0] < 12000


@pytest.mark.skipif(not os.path.isdir("/root/reference/map"), reason="the reference's STL maps exist only in the build container")
def test_reference_stl_maps_parse_and_measure(O):
    """Container-only (the reference cannot travel): the package's binary-STL reader takes the reference's own maps
    (map/*.STL, Lib/functions/MapFromSTL.m:1-11) -- 27 396 / 12 620 / 2 156 triangles, SURVEY.md section 2 -- and the
    CPU mesh oracle returns finite distances on them.  (stlread itself is an external toolbox function.)"""
    from motionplanning_5d_m_amd import mesh as M
    want = {"assembly line_Assem1.STL": 27396, "material_traveller.STL": 12620, "table.STL": 2156}
    rng = np.random.default_rng(3)
    for k, (name, nt) in enumerate(want.items()):
        tri = M.read_stl_binary(os.path.join("/root/reference/map", name))
        assert tri.shape == (nt, 3, 3) and np.isfinite(tri).all()
        tri_m = M.map_from_stl(tri) / 1000.0                          # MapFromSTL.m:6-10, mm -> m
        assert np.isfinite(tri_m).all() and tri_m.min() >= -0.1 - 1e-9       # MapFromSTL.m:8 shifts one axis by -100 mm
        lo, hi = tri_m.reshape(-1, 3).min(axis=0), tri_m.reshape(-1, 3).max(axis=0)
        O.mesh_register(13 + k, tri_m)
        segs = np.concatenate([rng.uniform(lo, hi, (16, 3)), rng.uniform(lo, hi, (16, 3))], axis=1)
        d, pts, t = O.mesh_seg_distance(13 + k, segs)
        assert np.isfinite(d).all() and (d >= 0).all() and (t >= 0).all() and (t < nt).all()
        on = tri_m[t]                                                  # the reported mesh point lies on the reported triangle's plane
        nrm = np.cross(on[:, 1] - on[:, 0], on[:, 2] - on[:, 0])
        good = np.linalg.norm(nrm, axis=1) > 1e-12
        off = np.abs(np.einsum("ij,ij->i", nrm[good], pts[good][:, 3:] - on[good][:, 0])) / np.linalg.norm(nrm[good], axis=1)
        assert off.max() < 1e-9


def test_reference_map_fixture_is_the_reference_file(O):
    """tests/golden/assembly_line_cell.npz holds DATA of /root/reference/map/assembly line_Assem1.STL: where the reference is
    present (the build container) the fixture loader must reproduce, bit for bit, the triangles the package's STL reader +
    MapFromSTL.m:6-10 + mm -> m give for the whole file, cropped to 2.5 m around robot.base."""
    import os
    from motionplanning_5d_m_amd import mesh
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tri = mesh.load_map_fixture(os.path.join(root, "tests", "golden", "assembly_line_cell.npz"))
    assert tri.shape == (13258, 3, 3)
    src = "/root/reference/map/assembly line_Assem1.STL"
    if not os.path.exists(src):
        pytest.skip("the reference is not present on this box")
    full = mesh.map_from_stl(mesh.read_stl_binary(src)) * 1e-3
    base = np.array([3.150, 8.500, 0.330])
    keep = np.linalg.norm(full - base, axis=2).min(axis=1) < 2.5
    assert full.shape[0] == 27396
    np.testing.assert_array_equal(full[keep], tri)
    # the reference's own transformed copy of the same map (map/environment.mat: envir.v in mm, MapFromSTL's output) agrees
    import scipy.io
    env = scipy.io.loadmat("/root/reference/map/environment.mat")["envir"]["v"][0, 0]
    assert env.shape == (27396 * 3, 3)
    np.testing.assert_allclose(env.reshape(-1, 3, 3) * 1e-3, full, rtol=0, atol=1e-9)
