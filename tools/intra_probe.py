#!/usr/bin/env python3
"""developer helper (GPU box): per-kernel time of one 4K I picture and of B pictures of the benchmark stream, alone on the device"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vvdec_amd
from vvdec_amd import abi, synth, stream
import bench
W, H = 3840, 2160
tools = (abi.TOOL_SAO_LUMA | abi.TOOL_SAO_CHROMA | abi.TOOL_ALF | abi.TOOL_CCALF | abi.TOOL_DEP_QUANT | abi.TOOL_MTS | abi.TOOL_LFNST |
         abi.TOOL_BDOF | abi.TOOL_DMVR | abi.TOOL_PROF | abi.TOOL_LMCS | abi.TOOL_LMCS_CSCALE | abi.TOOL_LFP_ON_DEVICE)
if os.environ.get("PROBE_NO_CS"):
    tools &= ~abi.TOOL_LMCS_CSCALE
plans, nslots = stream.ra_plan(17, gop=16, seed_poc0_is_external=False)
rec = vvdec_amd.Reconstructor(W, H, num_slots=nslots, num_streams=1)
for pl in plans[:int(os.environ.get("PROBE_PICTURES", "4"))]:
    mix = dict(bench.MIX)
    if os.environ.get("PROBE_SPLIT"):
        mix["p_split_scale"] = float(os.environ["PROBE_SPLIT"])
    d = synth.picture_for_plan(pl, W, H, seed=1234, tool_flags=tools, **mix)
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
    print("POC %2d  " % pl.poc + "  ".join("%s %.0f" % (k[2:], v) for k, v in st.items() if v > 0), " sum %.0f" % sum(st.values()), flush=True)
