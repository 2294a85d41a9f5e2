"""Makes tests/golden/reference_capsules_M16iB.npz from a DATA file of the reference (build container only).

    python tests/golden/make_reference_capsules.py

figure/M16iBCapsules.mat (loaded by Lib/functions/robotproperty2.m:96 for its drawing surfaces; figure/RobotCapsules.mat is the
same data, Lib/M16iB/robotproperty.m:38) also holds `RoCap`: the six M16iB capsules in WORLD coordinates -- the stored output of
the reference's forward kinematics (Lib/functions/CapPos.m:8-22) for the DH table of robotproperty2.m:67-72 at
theta = [0, 1.5708, 0, 0, -pi/2, pi] with base = [0;0;0], and the capsule radii.  A known-answer vector for row a1 (DH convention,
chaining of the link transforms, placement of the capsule end points).  Loaded with scipy.io.loadmat (MATLAB v5 container; the
graphics handles in the file are opaque records that are neither needed nor interpreted: no code is executed)."""
import os

import numpy as np
import scipy.io

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
m = scipy.io.loadmat("/root/reference/figure/M16iBCapsules.mat", squeeze_me=True, struct_as_record=False)
p = np.stack([np.asarray(c.p, float) for c in m["RoCap"]])           # (6, 3, 2): link, xyz, end point
r = np.array([float(c.r) for c in m["RoCap"]])
out = os.path.join(ROOT, "tests", "golden", "reference_capsules_M16iB.npz")
np.savez(out, p=p, r=r)
print(out, os.path.getsize(out), "bytes")
