"""Makes tests/golden/reference_rollout_M16.npz from two DATA files of the reference (build container only).

    python tests/golden/make_reference_rollout.py

data/M16_ref_2.mat holds `uref` (120 x 1: 24 inputs of 5 joints) and `xref` (250 x 1), data/good_xori.mat holds `xuori` (250 x 1):
a trajectory the reference's own legacy CFS script produced and saved (M16iB/main_CFS.m:19-21 loads both; its lines 162-169 show
how `xuori` was stacked from `xR(:,i)`).  `xuori` is -- bit for bit -- the double-integrator rollout
`xR(:,i) = A*xR(:,i-1) + B*u(i-1)` (Lib/CFS_FANUC.m:90-94 with robot.A, robot.B of robotproperty, delta_t = 0.5) of `uref` from
`xuori(1:10)`; `xref` is the same trajectory with 75 angles shifted by 2*pi.  They are the only stored OUTPUTS of the reference's
MATLAB runs in the repository that belong to the hot path: a known-answer vector for the rollout half of row a6.
Loaded with scipy.io.loadmat (MATLAB v5 container: no code is executed)."""
import os

import numpy as np
import scipy.io

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
a = scipy.io.loadmat("/root/reference/data/M16_ref_2.mat")
b = scipy.io.loadmat("/root/reference/data/good_xori.mat")
out = os.path.join(ROOT, "tests", "golden", "reference_rollout_M16.npz")
np.savez(out, uref=a["uref"].ravel(), xref=a["xref"].ravel(), xuori=b["xuori"].ravel())
print(out, os.path.getsize(out), "bytes")
