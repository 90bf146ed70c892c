#!/usr/bin/env python3
"""developer helper: summarise gpurun_out/intra_trace_poc*.bin written by the library under VVR_INTRA_TRACE=1"""
import sys, glob, numpy as np
for f in sorted(glob.glob(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/intra_trace_poc*.bin")):
    a = np.fromfile(f, np.uint64).reshape(-1, 8).astype(np.int64)
    t0 = a[:, 0][a[:, 0] > 0].min()
    us = lambda v: (v - t0) / 100.0
    start, dep, load, blk, store, pub = [a[:, k] for k in range(6)]
    ok = start > 0
    end = np.maximum(np.maximum(store, pub), np.maximum(blk, load))
    nb = a[:, 6] >> 32
    bulk = (a[:, 7] & 0x100) != 0
    nd = a[:, 7] & 0xff
    print(f, "units", int(ok.sum()), "kernel span %.1f us" % us(end[ok].max()))
    print("  start: first %.1f last %.1f us; median unit life %.1f us, p90 %.1f, max %.1f" % (us(start[ok].min()), us(start[ok].max()),
          np.median((end - start)[ok]) / 100, np.percentile((end - start)[ok], 90) / 100, (end - start)[ok].max() / 100))
    m = ok & ~bulk & (store > 0)
    for name, x in (("wait deps", dep - start), ("load refs", load - dep), ("blocks", blk - load), ("write back", store - blk)):
        print("  %-10s median %.1f  mean %.1f  p90 %.1f  max %.1f us" % (name, np.median(x[m]) / 100, x[m].mean() / 100, np.percentile(x[m], 90) / 100, x[m].max() / 100))
    print("  per block (blocks phase / blocks): %.2f us; bulk units %d, mean life %.1f us" % ((blk - load)[m].sum() / 100 / max(1, nb[m].sum()), int(bulk.sum()), ((store - start)[ok & bulk]).mean() / 100 if bulk.any() else 0))
    # concurrency: units alive over time
    ev = np.concatenate([np.stack([start[ok], np.ones(ok.sum(), np.int64)], 1), np.stack([end[ok], -np.ones(ok.sum(), np.int64)], 1)])
    ev = ev[np.argsort(ev[:, 0], kind="stable")]
    alive = np.cumsum(ev[:, 1])
    dt = np.diff(ev[:, 0], append=ev[-1, 0])
    print("  mean units alive %.0f, peak %d" % ((alive * dt).sum() / max(1, dt.sum()), alive.max()))
    with_dep = m & (nd > 0)
    print("  units with producers: %d, their mean wait %.1f us" % (int(with_dep.sum()), (dep - start)[with_dep].mean() / 100 if with_dep.any() else 0))
    # phases of the block loop (sums per unit): reference fill, set-up, prediction, residual stash + barrier
    A = a[:, 5] & 0xffffffff; B = a[:, 5] >> 32; Cc = (a[:, 7] >> 16) & 0xffffff; D = (a[:, 7] >> 40) & 0xffffff
    mm = m & (nb > 0)
    tot = max(1, int(nb[mm].sum()))
    print("  block loop per block: item+fetch+ref fill %.2f  smoothing+set-up %.2f  predict %.2f  stash+barrier %.2f us" % (A[mm].sum() / 100 / tot, B[mm].sum() / 100 / tot, Cc[mm].sum() / 100 / tot, D[mm].sum() / 100 / tot))
