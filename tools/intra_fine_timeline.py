#!/usr/bin/env python3
"""developer helper: when the CTUs of an I picture start and finish (intradev build, VVR_INTRA_TRACE=1 -> gpurun_out/intra_*poc0.bin), to see how the CTU wavefront advances:
the finish times of the luma units laid out as the CTU grid, their differences along a row and down a column, and where the blocks spend their time.
Written for the round-5 attempt to resolve the wavefront block by block (k_intra<.., FINE>, VVR_INTRA_FINE=1)."""
import sys
import numpy as np
d = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out"
ctusX = int(sys.argv[2]) if len(sys.argv) > 2 else 30
UNIT = np.dtype([("ent", "<u4"), ("i0", "<u4"), ("i1", "<u4"), ("bbox", "<u4"), ("ndeps", "<u4"), ("deps", "<u4", 26), ("iA", "<u4")])
ITEM = np.dtype([("x", "<u2"), ("y", "<u2"), ("lw", "u1"), ("lh", "u1"), ("mode", "u1"), ("flags", "u1"), ("nTL", "u1"), ("nA", "u1"), ("nL", "u1"), ("comp", "u1"), ("tu", "<u4")])
units = np.fromfile(d + "/intra_units_poc0.bin", UNIT); items = np.fromfile(d + "/intra_items_poc0.bin", ITEM)
ut = np.fromfile(d + "/intra_trace_poc0.bin", "<u8").reshape(-1, 8).astype(np.int64)
bt = np.fromfile(d + "/intra_btrace_poc0.bin", "<u8").reshape(-1, 8).astype(np.int64)
t0 = ut[:, 0][ut[:, 0] > 0].min()
comp = (units["ent"] >> 24) & 3; ctu = units["ent"] & 0xffffff
print("units", len(units), "kernel span %.0f us (developer build with the timeline: slower than the product build)" % ((ut[:, :5].max() - t0) / 100))
L = np.where(comp == 0)[0]
rows = int(ctu.max()) // ctusX + 1
S = np.full((rows, ctusX), np.nan); E = np.full((rows, ctusX), np.nan)
for u in L:
    x, y = ctu[u] % ctusX, ctu[u] // ctusX
    S[y, x] = (ut[u, 0] - t0) / 100; E[y, x] = (ut[u, 3] - t0) / 100
np.set_printoptions(linewidth=250, precision=0, suppress=True)
print("luma units: START (us after the launch), CTU rows 0..5, columns 0..14"); print(S[:6, :15])
print("luma units: all blocks DONE (us)"); print(E[:6, :15])
print("done(x) - done(x-1) along CTU row 3:", np.diff(E[3, :15]))
print("done(y) - done(y-1) down CTU column 5:", np.diff(E[:8, 5]))
u = L[len(L) // 2]; b = bt[units["iA"][u]:units["i1"][u]]
ratio = (b[:, 3].max() - b[:, 0].min()) / max(1, (ut[u, 3] - ut[u, 2]))
sel = np.concatenate([np.arange(units["iA"][u], units["i1"][u]) for u in L])
B = bt[sel]; ok = B[:, 3] > 0
w = (B[ok, 1] - B[ok, 0]) / ratio / 100; f = (B[ok, 2] - B[ok, 1]) / ratio / 100; p = (B[ok, 3] - B[ok, 2]) / ratio / 100
print("luma blocks, us: wait for the blocks before it in the CTU: median %.2f mean %.2f | wait for cells + fetch + reference fill: median %.2f mean %.2f p90 %.2f max %.1f | predict: median %.2f mean %.2f" % (np.median(w), w.mean(), np.median(f), f.mean(), np.percentile(f, 90), f.max(), np.median(p), p.mean()))
it = items[sel][ok]
edge = ((it["x"] % 128 == 0) | (it["y"] % 128 == 0))
print("blocks on a CTU's top / left edge: %d, (cells + fetch + fill) mean %.1f us; blocks inside: %d, mean %.2f us" % (edge.sum(), f[edge].mean(), (~edge).sum(), f[~edge].mean()))
