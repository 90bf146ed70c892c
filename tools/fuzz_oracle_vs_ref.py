"""developer helper: randomised sweep of the generator's parameter space (tools, sizes, CTU sizes, sample formats, stage) comparing the
plain-C oracle with the reference decoder's own classes (oracle/_ref).  Usage: tools/fuzz_oracle_vs_ref.py <seed> <seconds> [far]   (far: motion vectors up to thousands of samples, reference wrap-around in 6 pictures of 10).
Round 1: 4 x 200 s = 33 800 pictures, no mismatch.  Round 4 (after the two wrap-around findings): 2 x 600 s `far` = 54 811 pictures and 600 s without = 26 173
pictures, no mismatch (2 pictures refused by the reference: an 8-bit LMCS model of the generator that breaks the LmcsPivot constraint, Reshape.cpp:363)."""
import sys, random, time
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import numpy as np, refdrv
from vvdec_amd import abi, synth, stream
BASE = (abi.TOOL_SAO_LUMA | abi.TOOL_SAO_CHROMA | abi.TOOL_ALF | abi.TOOL_CCALF | abi.TOOL_DEP_QUANT | abi.TOOL_MTS | abi.TOOL_LFNST | abi.TOOL_BDOF | abi.TOOL_DMVR | abi.TOOL_PROF)
OPT = [abi.TOOL_LMCS, abi.TOOL_LMCS | abi.TOOL_LMCS_CSCALE, abi.TOOL_JCCR_SIGN, abi.TOOL_CCLM_COLLOC, abi.TOOL_WP, abi.TOOL_SCALING_LIST, abi.TOOL_SCALING_LIST | abi.TOOL_SCALING_LIST_NO_LFNST,
       abi.TOOL_IMPLICIT_MTS, abi.TOOL_IBC, abi.TOOL_STILL_REF]
plans, _ = stream.ra_plan(5, gop=4, seed_poc0_is_external=False)
def sweep(seed, seconds=None, cases=None):
  rnd = random.Random(seed)
  t_end = time.time() + (seconds or 1e9)
  n = bad = 0
  while time.time() < t_end and (cases is None or n < cases):
      seed = rnd.randrange(1 << 30)
      W, H = rnd.choice([(128, 64), (200, 136), (256, 128), (264, 200), (320, 192), (384, 256)])
      l2 = rnd.choice([5, 6, 7]); idx = rnd.randrange(5)
      tools = BASE
      for o in OPT:
          if rnd.random() < 0.3: tools |= o
      if rnd.random() < 0.15: tools &= ~abi.TOOL_DEP_QUANT
      kw = dict(p_intra=rnd.choice([0.0, 0.1, 0.3, 0.6]), p_split_scale=rnd.choice([0.5, 1.0, 1.5, 2.0]), p_coded=rnd.choice([0.2, 0.5, 0.9]), p_coded_chroma=rnd.choice([0.1, 0.5]),
                p_mts=rnd.choice([0, 0.3]), p_ts=rnd.choice([0, 0.2]), p_lfnst=rnd.choice([0, 0.4]), p_jccr=rnd.choice([0, 0.4]), p_mrl=rnd.choice([0, 0.3]), p_bdpcm=rnd.choice([0, 0.2]),
                p_affine=rnd.choice([0, 0.3]), p_geo=rnd.choice([0, 0.2]), p_ciip=rnd.choice([0, 0.3]), p_sbtmvp=rnd.choice([0, 0.3]), p_bcw=rnd.choice([0, 0.3]), p_cclm=rnd.choice([0, 0.4]),
                p_mip=rnd.choice([0, 0.3]), p_sbt=rnd.choice([0, 0.3]), p_isp=rnd.choice([0, 0.3]), p_ibc=rnd.choice([0, 0.4]), mv_sigma=rnd.choice([1.0, 8.0, 40.0]),
                p_imv_hpel=rnd.choice([0, 0.3]), p_small_corner=rnd.choice([0.2, 0.8]))
      if rnd.random() < 0.3: kw["min_cu_log2"] = 2
      if rnd.random() < 0.3: kw["dual_tree"] = rnd.choice([1.0, 2.0, 3.0])
      if FAR:
          # vectors far outside the picture (what AMVR in a parsed stream carries; the generator's default window keeps them near it): every MV clamp is taken,
          # and with reference wrap-around the moves by a period (round 4: the DMVR start clip and the SbTMVP pieces only showed with these)
          kw["mv_sigma"] = rnd.choice([300.0, 1500.0, 6000.0]); kw["mv_window"] = rnd.choice([500, 2000, 7000])          # ((W + window) * 16 stays inside the 18-bit MV range)
          off = W - 8 * rnd.randrange(5)
          if rnd.random() < 0.6 and off >= (1 << l2) + 16 and not (tools & abi.TOOL_IBC): kw["wrap_offset"] = off
      bd = rnd.choice([8, 10, 10]); cf = rnd.choice([1, 1, 1, 0])
      if not cf: tools &= ~abi.TOOL_LMCS_CSCALE
      if (tools & abi.TOOL_LMCS_CSCALE) and not (tools & abi.TOOL_LMCS): tools |= abi.TOOL_LMCS
      pl = plans[idx]
      try:
          d = synth.picture_for_plan(pl, W, H, seed=seed, tool_flags=tools, log2_ctu=l2, bit_depth=bd, chroma_format=cf, **kw)
          refs = {}
          for lst in pl.ref_slots:
              for (slot, poc) in lst: refs.setdefault(slot, synth.natural_picture(W, H, seed + 100 + poc, bit_depth=bd))
          fl = rnd.choice([0, 0, refdrv.STOP_AFTER_RECO, refdrv.STOP_AFTER_DBK, refdrv.DERIVE_LFP])
          want = refdrv.reconstruct(d, refs, flags=fl)["planes"]
          got = refdrv.oracle_reconstruct(d, refs, flags=fl & ~refdrv.DERIVE_LFP)
          diff = [int((a != b).sum()) for a, b in zip(got, want)]
          n += 1
          if any(diff):
              bad += 1
              print("MISMATCH", dict(W=W, H=H, l2=l2, idx=idx, seed=seed, tools=hex(tools), bd=bd, cf=cf, fl=fl, kw=kw), diff, flush=True)
      except Exception as e:
          n += 1; bad += 1
          print("EXC", repr(e)[:300], dict(W=W, H=H, l2=l2, idx=idx, seed=seed, tools=hex(tools), bd=bd, cf=cf, kw=kw), flush=True)
  return n, bad


FAR = len(sys.argv) > 3 and sys.argv[3] == "far"
if __name__ == "__main__":
    n, bad = sweep(int(sys.argv[1]), seconds=float(sys.argv[2]))
    print("cases", n, "bad", bad)
