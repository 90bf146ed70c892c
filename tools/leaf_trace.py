#!/usr/bin/env python3
"""developer helper: the per-item timeline of k_intra_leaf (intradev build, VVR_INTRA_TRACE=1 -> gpurun_out/leaf_trace_poc<N>.bin + leaf_items_poc<N>.bin):
when the items get their tickets, how long they wait for their neighbours, what fill / prediction / store cost, and the chain that ends last"""
import sys
import numpy as np
poc = sys.argv[1] if len(sys.argv) > 1 else "16"
d = sys.argv[2] if len(sys.argv) > 2 else "gpurun_out"
ITEM = np.dtype([("x", "<u2"), ("y", "<u2"), ("lw", "u1"), ("lh", "u1"), ("mode", "u1"), ("flags", "u1"), ("nTL", "u1"), ("nA", "u1"), ("nL", "u1"), ("comp", "u1"), ("tu", "<u4")])
items = np.fromfile("%s/leaf_items_poc%s.bin" % (d, poc), ITEM)
t = np.fromfile("%s/leaf_trace_poc%s.bin" % (d, poc), "<u8").reshape(-1, 8).astype(np.int64)
n = len(items)
ran = t[:, 1] > 0
t0 = t[ran, 0].min()
us = lambda v: (v - t0) / 100.0
print("items %d (ran as first wavefront of their block: %d), kernel span %.1f us" % (n, ran.sum(), (t[ran].max() - t0) / 100.0))
kinds = np.where(items["mode"] == 253, "csfac", np.where(items["mode"] == 255, "resi_add", np.where((items["comp"] & 3) == 0, "luma", "chroma")))
print("entry of workgroups: first %.1f, median %.1f, last %.1f us; ticket + item record: median %.2f, p90 %.2f, max %.2f us" % (
    us(t[ran, 0]).min(), np.median(us(t[ran, 0])), us(t[ran, 0]).max(), np.median(t[ran, 1] - t[ran, 0]) / 100, np.percentile(t[ran, 1] - t[ran, 0], 90) / 100, (t[ran, 1] - t[ran, 0]).max() / 100))
for k in ("luma", "csfac", "resi_add", "chroma"):
    s = ran & (kinds == k)
    if not s.any():
        continue
    q = t[s]
    full = q[:, 2] > 0
    line = "%-9s n %5d  start %6.1f..%6.1f  end %6.1f..%6.1f (median %.1f)" % (k, s.sum(), us(q[:, 1]).min(), us(q[:, 1]).max(), us(q[:, 6:8].max(1)).min(), us(q[:, 6:8].max(1)).max(), np.median(us(q[:, 6:8].max(1))))
    if full.any():
        f = q[full]
        ph = [np.median(f[:, i + 1] - f[:, i]) / 100 for i in range(1, 7)]
        line += "  | median us: wait %.2f fill %.2f predict %.2f store %.2f drain %.2f cells %.2f" % tuple(ph)
        w = (f[:, 2] - f[:, 1]) / 100
        line += " | wait p90 %.1f max %.1f; items that waited > 1 us: %d" % (np.percentile(w, 90), w.max(), (w > 1).sum())
    print(line)
# the chain that ends last: walk back from the last item through "the item whose end is closest before my wait ended"
end = t[:, 6:8].max(1)
last = int(np.argmax(np.where(ran, end, 0)))
print("last item to finish: #%d (%s, mode %d, %dx%d at %d,%d): ticket at %.1f, waited until %.1f, done %.1f" % (last, kinds[last], items["mode"][last], 1 << items["lw"][last], 1 << items["lh"][last], items["x"][last], items["y"][last], us(t[last, 1]), us(t[last, 2]), us(end[last])))
cur, hops = last, 0
while hops < 40 and t[cur, 2] - t[cur, 1] > 50:
    cand = np.where(ran & (end <= t[cur, 2]) & (end > t[cur, 2] - 300))[0]
    if not len(cand):
        break
    same = [c for c in cand if abs(int(items["x"][c]) - int(items["x"][cur])) < 160 and abs(int(items["y"][c]) - int(items["y"][cur])) < 160]
    if not same:
        break
    prv = max(same, key=lambda c: end[c])
    print("   <- #%d (%s mode %d %dx%d at %d,%d) ticket %.1f waited until %.1f done %.1f  [hop: producer done -> consumer go %.2f us; consumer go -> done %.2f us]" % (
        prv, kinds[prv], items["mode"][prv], 1 << items["lw"][prv], 1 << items["lh"][prv], items["x"][prv], items["y"][prv], us(t[prv, 1]), us(t[prv, 2]), us(end[prv]), (t[cur, 2] - end[prv]) / 100, (end[cur] - t[cur, 2]) / 100))
    print("        stages of #%d: fill %.2f predict %.2f store %.2f drain %.2f cells %.2f us" % ((cur,) + tuple((t[cur, i + 1] - t[cur, i]) / 100 for i in range(2, 7))))
    cur = prv; hops += 1
print("chain length", hops)
# ---- occupancy of the device over time and how long a wavefront lives
alive_from, alive_to = t[ran, 0], np.maximum(t[ran, 7], t[ran, 6:8].max(1))
life = (alive_to - alive_from) / 100.0
print("wavefront life (entry -> last stamp): median %.1f p90 %.1f max %.1f us; sum %.0f wave-us = %.0f waves alive on average over the span" % (np.median(life), np.percentile(life, 90), life.max(), life.sum(), life.sum() / ((t[ran].max() - t0) / 100.0)))
grid = np.arange(0, (t[ran].max() - t0) / 100.0, 5.0)
print("waves alive at t =", " ".join("%d:%d" % (g, ((us(alive_from) <= g) & (us(alive_to) > g)).sum()) for g in grid))
print("entries per 5 us   ", " ".join("%d:%d" % (g, ((us(alive_from) >= g) & (us(alive_from) < g + 5)).sum()) for g in grid))
