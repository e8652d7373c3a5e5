#!/usr/bin/env python3
"""Where does a fused depthwise -> pointwise case differ from the two-stage oracle?  (test infrastructure: imports the oracle)
Usage: python tests/fused_debug.py N C HW STRIDE M [dw_act pw_act]   -> per-stage mismatch counts and their coordinates."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402
from oracle import plref  # noqa: E402

plref.build()
pkg = ge.import_package()
capi = pkg.capi
n, c, hw, st, m = [int(v) for v in sys.argv[1:6]]
dw_act = int(sys.argv[6]) if len(sys.argv) > 6 else 1
pw_act = int(sys.argv[7]) if len(sys.argv) > 7 else 1
rng = np.random.default_rng(300)
x = rng.integers(-127, 128, (n, c, hw, hw)).astype(np.int8)
w_dw = rng.integers(-127, 128, (c, 1, 3, 3)).astype(np.int8)
w_pw = rng.integers(-127, 128, (m, c, 1, 1)).astype(np.int8)
b_dw = rng.uniform(-1, 1, c).astype(np.float32)
ws_dw = ((1 + np.arange(c) % 7) / 127.0 / 4.0).astype(np.float32)
in_s, mid_s = 1 / 127.0, 9 / 127.0
pad = (1, 1, 1, 1)
sd = plref.shape(n, c, hw, hw, c, 3, 3, pad, (st, st), (1, 1), c)
oh, ow = plref.out_dims(sd)
s1, b1, a1 = plref.fold_scales(1, in_s, ws_dw, mid_s, b_dw, c, dw_act, 0.0)
d_ref, _ = plref.conv2d(sd, x, w_dw, b_dw, in_s, ws_dw, mid_s, dw_act, 0.0, True)
sp = plref.shape(n, c, oh, ow, m, 1, 1, (0, 0, 0, 0), (1, 1), (1, 1), 1)
_, acc_ref = plref.conv2d(sp, d_ref, w_pw, None, mid_s, np.ones(m, np.float32), 1.0, pw_act, 0.0, True)
d_dw = capi.conv_desc(n, c, hw, hw, c, 3, 3, pad, (st, st), (1, 1), c, dw_act, a1)
with capi.Context(0) as ctx:
    # an identity pointwise stage shows the depthwise stage's output itself (m == c only)
    acc = ctx.dwpw_fused(d_dw, x, w_dw, s1, b1, w_pw, None, None, pw_act, 0.0, capi.OUT_I32)
bad = np.argwhere(acc != acc_ref)
print("accumulators: %d of %d differ" % (len(bad), acc.size))
if len(bad):
    for ax, nm in enumerate(("image", "channel", "row", "col")):
        u, cnt = np.unique(bad[:, ax], return_counts=True)
        print("  by %s: %s" % (nm, dict(zip(u.tolist()[:24], cnt.tolist()[:24]))))
    print("  first:", bad[:8].tolist())
    # which depthwise outputs would explain it?  solve per pixel: diff = W_pw (d_got - d_ref)
    b0, _, r0, c0 = bad[0]
    diff = (acc[b0, :, r0, c0].astype(np.int64) - acc_ref[b0, :, r0, c0].astype(np.int64))
    sol, res, rk, _ = np.linalg.lstsq(w_pw[:, :, 0, 0].astype(np.float64), diff.astype(np.float64), rcond=None)
    nz = np.argwhere(np.abs(sol) > 0.5)[:, 0]
    print("  pixel (%d, %d, %d): depthwise channels off: %s  by %s" % (b0, r0, c0, nz[:16].tolist(), np.round(sol[nz[:16]]).tolist()))
    print("  reference depthwise there:", d_ref[b0, nz[:16], r0, c0].tolist())
