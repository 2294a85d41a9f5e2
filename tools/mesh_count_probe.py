import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, time
import motionplanning_5d_m_amd as pkg
from motionplanning_5d_m_amd import workloads
s, bt, tri = workloads.config5(B=64)
mesh = pkg.Mesh(tri=tri)
th = bt.x_init.reshape(64, 50, 10)[:, :, :5].reshape(-1, 5)
d, lid, pos = pkg.dist_arm(s.robot, th, np.zeros((1, 6)), want_pos=True)
segs = pos.reshape(-1, 6)
t0 = time.time(); dis, pts, tid = mesh.point2surface_dis(segs); dt = time.time() - t0
nn, nt = pts[:, 4], pts[:, 5]
ln = np.linalg.norm(segs[:, 3:] - segs[:, :3], axis=1)
print("queries", segs.shape[0], "time %.1f ms" % (dt * 1e3), "nodes mean %.0f max %.0f; tris mean %.0f max %.0f" % (nn.mean(), nn.max(), nt.mean(), nt.max()))
for k in range(5):
    m = np.arange(segs.shape[0]) % 5 == k
    print("link", k + 1, "len %.3f" % ln[m].mean(), "dist mean %.3f" % dis[m].mean(), "nodes %.0f tris %.0f (max %.0f)" % (nn[m].mean(), nt[m].mean(), nt[m].max()))
