#!/usr/bin/env python3
"""developer helper (GPU box): k_intra time of one 4K I picture and one 4K B picture, alone on the device"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vvdec_amd
from vvdec_amd import abi, synth, stream
W, H = 3840, 2160
tools = abi.TOOL_SAO_LUMA | abi.TOOL_SAO_CHROMA | abi.TOOL_ALF | abi.TOOL_CCALF | abi.TOOL_DEP_QUANT | abi.TOOL_MTS | abi.TOOL_LFNST
plans, nslots = stream.ra_plan(17, gop=16, seed_poc0_is_external=False)
rec = vvdec_amd.Reconstructor(W, H, num_slots=nslots, num_streams=1)
out = []
for pl in plans[:2]:
    d = synth.picture_for_plan(pl, W, H, seed=1234, tool_flags=tools)
    h = rec.prepare(d)
    for i in range(3):
        rec.submit_prepared(h)
    rec.sync()
    rec.enable_stats(True)
    for i in range(5):
        rec.submit_prepared(h)
    rec.sync()
    st = {s["name"]: 1e3 * s["total_ms"] / max(1, s["launches"]) for s in rec.stats()}
    rec.enable_stats(False)
    out.append("POC %d k_intra %.1f us" % (pl.poc, st.get("k_intra", 0)))
print(os.environ.get("VVR_INTRA_DBG", "0"), " | ".join(out))
