#!/usr/bin/env python3
"""developer helper: random pictures with scaled reference pictures (reference picture resampling) - sizes, ratios from 1/8 to 2 incl. the filter-set
thresholds, scaling windows with negative offsets, chroma sample locations, 8 / 10 bit, 4:0:0 / 4:2:0, large motion vectors.
  python tools/fuzz_rpr.py ref  [seed] [cases]     oracle against the reference's classes (CPU, needs oracle/_ref)
  python tools/fuzz_rpr.py gpu  [seed] [cases]     the back-end against the oracle (MI355X)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import refdrv
from vvdec_amd import abi
from test_oracle_vs_ref import rpr_case, ALL

R1 = 1 << 14


def cases(seed, n):
    rng = np.random.default_rng(seed)
    for it in range(n):
        l2 = int(rng.integers(5, 8))
        W, H = int(rng.integers(16, 60)) * 8, int(rng.integers(12, 40)) * 8
        specs = []
        for k in range(2):
            if rng.random() < 0.75:
                rx, ry = [int(R1 * float(rng.choice([0.125, 0.3, 0.5, 0.75, 1.0, 1.1, 1.25, 1.26, 1.5, 1.75, 1.76, 2.0]))) + int(rng.integers(-3, 4)) for _ in range(2)]
                rx, ry = min(max(rx, 1 << 11), 1 << 15), min(max(ry, 1 << 11), 1 << 15)
                specs.append(dict(ratio=(rx, ry), size=(int(rng.integers(8, 80)) * 8, int(rng.integers(8, 60)) * 8), win=(int(rng.integers(-8, 9)) * 2, int(rng.integers(-8, 9)) * 2)))
            else:
                specs.append(None)
        win = (int(rng.integers(-6, 7)) * 2, int(rng.integers(-6, 7)) * 2)
        colloc = (int(rng.integers(0, 2)), int(rng.integers(0, 2)))
        cf = 0 if rng.random() < 0.15 else 1
        bd = int(rng.choice([8, 10]))
        tools = ALL | (abi.TOOL_WP if rng.random() < 0.3 else 0) | (abi.TOOL_LMCS if rng.random() < 0.3 else 0)
        idx = int(rng.integers(1, 5))
        kw = dict(p_intra=0.1, p_affine=0.3, p_geo=0.1, p_ciip=0.1, p_sbtmvp=0.15, p_bcw=0.3, p_imv_hpel=0.2, mv_sigma=float(rng.choice([2.0, 8.0, 40.0])))
        if not cf:
            kw.pop("p_ciip")
        if any(specs):
            yield it, rpr_case(W, H, l2, idx, seed * 1000 + it, specs, win=win, colloc=colloc, tools=tools, bit_depth=bd, chroma_format=cf, **kw), l2


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "ref"
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 30
    bad = done = 0
    for it, (d, refs), l2 in cases(seed, n):
        want = refdrv.oracle_reconstruct(d, refs)
        if mode == "ref":
            got = refdrv.reconstruct(d, refs)["planes"]
        else:
            import vvdec_amd
            h = d.hdr
            MW = max([h.width] + [r[0].shape[1] for r in refs.values()]); MH = max([h.height] + [r[0].shape[0] for r in refs.values()])
            rec = vvdec_amd.Reconstructor(MW, MH, num_slots=max(list(refs) + [h.out_slot]) + 1, num_streams=1, log2_ctu=l2, bit_depth=h.bit_depth, chroma_format=h.chroma_format)
            for slot, planes in refs.items():
                full = [np.zeros(rec.plane_shape(c), np.uint16) for c in range(len(planes))]
                for c, p in enumerate(planes):
                    full[c][:p.shape[0], :p.shape[1]] = p
                rec.write_picture(slot, full)
            rec.wait(rec.decompress_picture(d))
            got = [g[:w.shape[0], :w.shape[1]] for g, w in zip(rec.read_picture(h.out_slot), want)]
            rec.close()
        done += 1
        if not all(np.array_equal(a, b) for a, b in zip(got, want)):
            bad += 1
            print("MISMATCH case", it, "of seed", seed, flush=True)
    print("%s: %d cases, %d mismatches" % (mode, done, bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
