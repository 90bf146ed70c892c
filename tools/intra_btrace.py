#!/usr/bin/env python3
"""developer helper: per-block timeline of k_intra (intradev build, VVR_INTRA_TRACE=1): where a block's time goes on the serial path"""
import sys
import numpy as np
poc = sys.argv[1] if len(sys.argv) > 1 else "0"
d = sys.argv[2] if len(sys.argv) > 2 else "gpurun_out"
UNIT = np.dtype([("ent", "<u4"), ("i0", "<u4"), ("i1", "<u4"), ("bbox", "<u4"), ("ndeps", "<u4"), ("deps", "<u4", 26), ("iA", "<u4")])
ITEM = np.dtype([("x", "<u2"), ("y", "<u2"), ("lw", "u1"), ("lh", "u1"), ("mode", "u1"), ("flags", "u1"), ("nTL", "u1"), ("nA", "u1"), ("nL", "u1"), ("comp", "u1"), ("tu", "<u4")])
units = np.fromfile("%s/intra_units_poc%s.bin" % (d, poc), UNIT)
items = np.fromfile("%s/intra_items_poc%s.bin" % (d, poc), ITEM)
bt = np.fromfile("%s/intra_btrace_poc%s.bin" % (d, poc), "<u8").reshape(-1, 8).astype(np.int64)
ut = np.fromfile("%s/intra_trace_poc%s.bin" % (d, poc), "<u8").reshape(-1, 8).astype(np.int64)
print("units", len(units), "items", len(items))
# shader clock per 100 MHz tick: block loop of the long units
ratios = []
for t, u in enumerate(units):
    if u["i1"] - u["iA"] >= 16 and ut[t, 3] > ut[t, 2]:
        b = bt[u["iA"]:u["i1"]]
        ratios.append((b[:, 3].max() - b[:, 0].min()) / (ut[t, 3] - ut[t, 2]))
r = float(np.median(ratios)) if ratios else 21.0
print("shader clocks per 10 ns tick: %.2f  (%.2f GHz)" % (r, r / 10))
us = lambda c: c / r / 100.0
for comp in (0, 1):
    sel = [t for t, u in enumerate(units) if ((u["ent"] >> 24) & 3) == comp and u["iA"] < u["i1"]]
    if not sel:
        continue
    ph = np.array([[ut[t, 1] - ut[t, 0], ut[t, 2] - ut[t, 1], ut[t, 3] - ut[t, 2], ut[t, 4] - ut[t, 3]] for t in sel]) / 100.0
    nb = np.array([units[t]["i1"] - units[t]["iA"] for t in sel])
    print("comp %d: %d units, %.1f blocks/unit; per unit us: wait-deps %.1f  stage %.1f  blocks %.1f  write-back %.1f" % (comp, len(sel), nb.mean(), *ph.mean(0)))
    rows = []
    for t in sel:
        u = units[t]
        for q in range(u["iA"], u["i1"]):
            prev = bt[q - 4, 3] if q - 4 >= u["iA"] else bt[q, 0]
            rows.append((int(items[q]["lw"]) + int(items[q]["lh"]), us(bt[q, 0] - prev), us(bt[q, 1] - bt[q, 0]), us(bt[q, 2] - bt[q, 1]), us(bt[q, 3] - bt[q, 2]), int(items[q]["mode"]), int(items[q]["flags"]),
                         us(bt[q, 4] - bt[q, 2]) if bt[q, 4] else -1, us(bt[q, 5] - bt[q, 4]) if bt[q, 4] else -1, us(bt[q, 6] - bt[q, 5]) if bt[q, 4] else -1, us(bt[q, 3] - bt[q, 6]) if bt[q, 4] else -1))
    a = np.array(rows)
    print("   blocks %d: prologue %.2f  wait %.2f  fill %.2f  predict %.2f   (critical = fill + predict %.2f us)" % (len(a), a[:, 1].mean(), a[:, 2].mean(), a[:, 3].mean(), a[:, 4].mean(), (a[:, 3] + a[:, 4]).mean()))
    m = a[:, 7] >= 0
    if m.any():
        print("   regular blocks %d: smoothing + projection %.2f  loop set-up %.2f  group loop %.2f  done %.2f" % (m.sum(), a[m, 7].mean(), a[m, 8].mean(), a[m, 9].mean(), a[m, 10].mean()))
        sm = m & (a[:, 0] <= 8)
        print("   regular blocks of <= 256 samples %d: smoothing + projection %.2f  loop set-up %.2f  group loop %.2f  done %.2f" % (sm.sum(), a[sm, 7].mean(), a[sm, 8].mean(), a[sm, 9].mean(), a[sm, 10].mean()))
    for l2 in sorted(set(a[:, 0])):
        s = a[a[:, 0] == l2]
        print("     log2 samples %2d: n %6d  prologue %.2f  wait %.2f  fill %.2f  predict %.2f" % (l2, len(s), s[:, 1].mean(), s[:, 2].mean(), s[:, 3].mean(), s[:, 4].mean()))
    for name, m in (("planar", a[:, 5] == 0), ("dc", a[:, 5] == 1), ("angular", (a[:, 5] > 1) & (a[:, 5] < 67)), ("cclm", (a[:, 5] >= 67) & (a[:, 5] < 70)), ("mip(luma flag 8)", (a[:, 6].astype(int) & 8) != 0)):
        if m.any():
            s = a[m]
            print("     %-18s n %6d  fill %.2f  predict %.2f" % (name, len(s), s[:, 3].mean(), s[:, 4].mean()))
