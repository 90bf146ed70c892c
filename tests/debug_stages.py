"""Developer tool (GPU box): find the first stage / component / position where the HIP path leaves the oracle."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import vvdec_amd, refdrv
from vvdec_amd import abi, synth, stream

W, H, seed = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
kw = eval(sys.argv[4]) if len(sys.argv) > 4 else {}
tools = abi.TOOL_SAO_LUMA | abi.TOOL_SAO_CHROMA | abi.TOOL_ALF | abi.TOOL_CCALF | abi.TOOL_DEP_QUANT | abi.TOOL_MTS
plans, nslots = stream.ra_plan(5, gop=4)
seed_pic = synth.natural_picture(W, H, seed)
pl = plans[0]
d = synth.picture_for_plan(pl, W, H, seed=seed, tool_flags=tools, p_intra=0.0, **kw)
for name, flag, stop in (("reco", 4, abi.STOP_RECO), ("dbk", 8, abi.STOP_DEBLOCK), ("sao", 16, abi.STOP_SAO), ("", 0, abi.STOP_NONE)):
    rec = vvdec_amd.Reconstructor(W, H, num_slots=nslots, num_streams=1, log2_ctu=d.hdr.log2_ctu, stop_after=stop)
    rec.write_picture(0, seed_pic)
    rec.wait(rec.decompress_picture(d))
    got = rec.read_picture(pl.slot)
    rec.close()
    want = refdrv.oracle_reconstruct(d, {0: seed_pic}, flags=flag)
    ok = True
    for c in range(3):
        ys, xs = np.nonzero(got[c] != want[c])
        if len(ys):
            ok = False
            print("stage %-5s comp %d: %d diffs; x range %d..%d y range %d..%d; first %s got %s want %s" % (name or "alf", c, len(ys), xs.min(), xs.max(), ys.min(), ys.max(),
                  list(zip(xs[:5], ys[:5])), got[c][ys[:5], xs[:5]], want[c][ys[:5], xs[:5]]))
    print("stage %-5s %s" % (name or "alf", "OK" if ok else "MISMATCH"))
    if not ok:
        # which CUs are affected
        bad = 0
        for i, cu in enumerate(d.cu):
            x, y, w, h = int(cu['x']), int(cu['y']), int(cu['w']), int(cu['h'])
            df = [int((got[c][y >> (c > 0):(y + h) >> (c > 0), x >> (c > 0):(x + w) >> (c > 0)] != want[c][y >> (c > 0):(y + h) >> (c > 0), x >> (c > 0):(x + w) >> (c > 0)]).sum()) for c in range(3)]
            if sum(df):
                bad += 1
                if bad <= 5:
                    print("  CU", i, (x, y, w, h), df, "mc", cu['mc_mode'], "ref", cu['ref_idx'], "mv", cu['mv'][0][0], cu['mv'][1][0], "imv", cu['imv'], "flags", cu['flags'])
                    for t in range(cu['first_tu'], cu['first_tu'] + cu['num_tu']):
                        tu = d.tu[t]; print("     TU", (tu['x'], tu['y'], tu['w'], tu['h']), "cbf", tu['cbf'], "mts", tu['mts_idx'], "msx", tu['max_scan_x'], "msy", tu['max_scan_y'], "qp", tu['qp'])
                    pred = refdrv.oracle_reconstruct(d, {0: seed_pic}, flags=flag)  # same as want
                    for c in range(3):
                        if df[c]:
                            sl = (slice(y >> (c > 0), (y >> (c > 0)) + 4), slice(x >> (c > 0), (x >> (c > 0)) + 8))
                            print("     comp", c, "got-want rows 0..3:\n", got[c][sl].astype(int) - want[c][sl].astype(int))
        print("  bad CUs:", bad)
        break
