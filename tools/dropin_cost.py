#!/usr/bin/env python3
"""what the host side of the drop-in costs per 4K picture: LF_INIT (LoopFilter::calcFilterStrengthsCTU over the picture) and the flatten step
(integration/vvr_extract.h) as integration/DecLibReconDropIn.cpp runs them - one task of the decoder's thread pool per picture - measured on the
stand-in runtime (no GPU; the back-end's own host stage is measured by tools/host_path_probe.py).  MIDER is skipped: the harness builds pictures
whose motion is final.   Usage: python tools/dropin_cost.py [threads]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["VVDEC_AMD_TIMES"] = "1"
import refdrv
from vvdec_amd import abi, synth, stream
import bench
import test_host_glue as T
W, H = 3840, 2160
threads = int(sys.argv[1]) if len(sys.argv) > 1 else 2
plans, _ = stream.ra_plan(33, gop=32, seed_poc0_is_external=False)
for pl in (plans[0], plans[2], plans[5]):
    d = synth.picture_for_plan(pl, W, H, seed=1234, tool_flags=bench._tools(abi), **bench.MIX)
    refs = {slot: synth.natural_picture(W, H, 100 + poc) for lst in pl.ref_slots for (slot, poc) in lst}
    print("POC %d (%s):" % (pl.poc, "I" if pl.slice_type == 2 else "B"), flush=True)
    refdrv.run_dropin(d, refs, T.build_stub(), threads=threads)
