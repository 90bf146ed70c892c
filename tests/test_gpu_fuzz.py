"""GPU (-m gpu): the RANDOMISED leg of the parity suite.  Every other GPU case is a fixed seed; here the seeds change with every build of the kernels, so that
each run of the suite covers pictures and streams no run before it has seen:
  * generated pictures from the parameter space of tools/fuzz_oracle_vs_ref.py (tool switches, picture and CTU sizes, sample formats, split / residual densities,
    MV spreads incl. the `far` mode with vectors of thousands of samples and reference wrap-around), reconstructed through the C ABI on the HIP path and compared
    with the CPU oracle on the same reference pictures;
  * bitstreams written by tools/mini_vvenc.py with mutated tool mixes, decoded by the reference's application on the DROP-IN library with the HIP back-end behind
    vvdec::DecLibRecon: output MD5 == the reference decoder's (oracle/_ref/vvdecapp_ref).
The seed is the MD5 of the kernel sources (VVR_FUZZ_SEED overrides it: a failure prints the seed it ran with)."""
import hashlib
import os
import random
import sys
import time

import numpy as np
import pytest

import refdrv
from vvdec_amd import abi, synth, stream

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def _seed():
    if os.environ.get("VVR_FUZZ_SEED"):
        return int(os.environ["VVR_FUZZ_SEED"])
    h = hashlib.md5()
    for f in ("vvr_kernels.hip", "vvr_intra_leaf.inc", "vvr_prepare.cpp", "vvr_api.cpp", "vvr_lf_init.h"):
        h.update(open(os.path.join(ROOT, "vvdec_amd", "csrc", f), "rb").read())
    return int(h.hexdigest()[:8], 16)


BASE = (abi.TOOL_SAO_LUMA | abi.TOOL_SAO_CHROMA | abi.TOOL_ALF | abi.TOOL_CCALF | abi.TOOL_DEP_QUANT | abi.TOOL_MTS | abi.TOOL_LFNST | abi.TOOL_BDOF | abi.TOOL_DMVR | abi.TOOL_PROF)
OPT = [abi.TOOL_LMCS, abi.TOOL_LMCS | abi.TOOL_LMCS_CSCALE, abi.TOOL_JCCR_SIGN, abi.TOOL_CCLM_COLLOC, abi.TOOL_WP, abi.TOOL_SCALING_LIST, abi.TOOL_SCALING_LIST | abi.TOOL_SCALING_LIST_NO_LFNST,
       abi.TOOL_IMPLICIT_MTS, abi.TOOL_IBC, abi.TOOL_STILL_REF, abi.TOOL_LFP_ON_DEVICE, abi.TOOL_LFP_ON_DEVICE | abi.TOOL_AFFINE_MV_ON_DEVICE]


def _case(rnd, far):
    """one point of the generator's parameter space (tools/fuzz_oracle_vs_ref.py::sweep, plus the switches that move work onto the device)"""
    seed = rnd.randrange(1 << 30)
    W, H = rnd.choice([(128, 64), (200, 136), (256, 128), (264, 200), (320, 192), (384, 256)])
    l2 = rnd.choice([5, 6, 7]); idx = rnd.randrange(5)
    tools = BASE
    for o in OPT:
        if rnd.random() < 0.3:
            tools |= o
    if rnd.random() < 0.15:
        tools &= ~abi.TOOL_DEP_QUANT
    kw = dict(p_intra=rnd.choice([0.0, 0.1, 0.3, 0.6]), p_split_scale=rnd.choice([0.5, 1.0, 1.5, 2.0]), p_coded=rnd.choice([0.2, 0.5, 0.9]), p_coded_chroma=rnd.choice([0.1, 0.5]),
              p_mts=rnd.choice([0, 0.3]), p_ts=rnd.choice([0, 0.2]), p_lfnst=rnd.choice([0, 0.4]), p_jccr=rnd.choice([0, 0.4]), p_mrl=rnd.choice([0, 0.3]), p_bdpcm=rnd.choice([0, 0.2]),
              p_affine=rnd.choice([0, 0.3]), p_geo=rnd.choice([0, 0.2]), p_ciip=rnd.choice([0, 0.3]), p_sbtmvp=rnd.choice([0, 0.3]), p_bcw=rnd.choice([0, 0.3]), p_cclm=rnd.choice([0, 0.4]),
              p_mip=rnd.choice([0, 0.3]), p_sbt=rnd.choice([0, 0.3]), p_isp=rnd.choice([0, 0.3]), p_ibc=rnd.choice([0, 0.4]), mv_sigma=rnd.choice([1.0, 8.0, 40.0]),
              p_imv_hpel=rnd.choice([0, 0.3]), p_small_corner=rnd.choice([0.2, 0.8]))
    if rnd.random() < 0.3:
        kw["min_cu_log2"] = 2
    if rnd.random() < 0.3:
        kw["dual_tree"] = rnd.choice([1.0, 2.0, 3.0])
    if far:
        kw["mv_sigma"] = rnd.choice([300.0, 1500.0, 6000.0]); kw["mv_window"] = rnd.choice([500, 2000, 7000])
        off = W - 8 * rnd.randrange(5)
        if rnd.random() < 0.6 and off >= (1 << l2) + 16 and not (tools & abi.TOOL_IBC):
            kw["wrap_offset"] = off
    bd = rnd.choice([8, 10, 10]); cf = rnd.choice([1, 1, 1, 0])
    if not cf:
        tools &= ~abi.TOOL_LMCS_CSCALE
    if (tools & abi.TOOL_LMCS_CSCALE) and not (tools & abi.TOOL_LMCS):
        tools |= abi.TOOL_LMCS
    if (tools & abi.TOOL_AFFINE_MV_ON_DEVICE) and not (tools & abi.TOOL_LFP_ON_DEVICE):
        tools &= ~abi.TOOL_AFFINE_MV_ON_DEVICE
    return dict(W=W, H=H, l2=l2, idx=idx, seed=seed, tools=tools, bd=bd, cf=cf, kw=kw)


def test_gpu_fuzz_slice(built):
    """300 generated pictures (a third of them in the `far` mode), HIP path == CPU oracle, every plane bit-exact, the DMVR delta MVs too"""
    import vvdec_amd
    seed0 = _seed()
    rnd = random.Random(seed0)
    plans, nslots = stream.ra_plan(5, gop=4, seed_poc0_is_external=False)
    ctxs = {}
    n = refused = 0
    t0 = time.time()
    budget = float(os.environ.get("VVR_FUZZ_SECONDS", "75"))
    while n < int(os.environ.get("VVR_FUZZ_PICTURES", "300")) and time.time() - t0 < budget:
        c = _case(rnd, far=(n % 3 == 2))
        pl = plans[c["idx"]]
        key = (c["W"], c["H"], c["l2"], c["bd"], c["cf"])
        if key not in ctxs:
            ctxs[key] = vvdec_amd.Reconstructor(c["W"], c["H"], num_slots=nslots, num_streams=1, host_threads=0, log2_ctu=c["l2"], bit_depth=c["bd"], chroma_format=c["cf"])
        rec = ctxs[key]
        ncomp = 3 if c["cf"] else 1
        try:
            d = synth.picture_for_plan(pl, c["W"], c["H"], seed=c["seed"], tool_flags=c["tools"], log2_ctu=c["l2"], bit_depth=c["bd"], chroma_format=c["cf"], **c["kw"])
        except Exception:
            refused += 1            # (a parameter combination the generator does not make)
            continue
        refs = {}
        for lst in pl.ref_slots:
            for (slot, poc) in lst:
                refs.setdefault(slot, synth.natural_picture(c["W"], c["H"], c["seed"] + 100 + poc, bit_depth=c["bd"])[:ncomp])
        for slot, planes in refs.items():
            rec.write_picture(slot, planes)
        try:
            want = refdrv.oracle_reconstruct(d, refs)
        except Exception:
            refused += 1            # (the oracle refuses what the back-end refuses: e.g. an LMCS model outside its constraints)
            continue
        job = rec.decompress_picture(d)
        rec.wait(job)
        got = rec.read_picture(pl.slot)
        for k in range(ncomp):
            assert np.array_equal(got[k], want[k]), "fuzz seed %d, case %d %r: component %d differs in %d samples" % (seed0, n, c, k, int((got[k] != want[k]).sum()))
        nd = getattr(d, "num_dmvr", 0)
        if nd:
            assert np.array_equal(rec.read_dmvr(job, nd), refdrv.oracle_dmvr(nd)), "fuzz seed %d, case %d %r: DMVR delta MVs differ" % (seed0, n, c)
        n += 1
    for rec in ctxs.values():
        rec.close()
    assert n >= 40, "only %d pictures in the time budget (%d refused)" % (n, refused)
    print("fuzz seed %d: %d pictures bit-exact, %d parameter sets refused by generator / oracle, %.0f s" % (seed0, n, refused, time.time() - t0))


INTRA_SWITCHES = ["sao", "lmcs", "jccr", "dep_quant", "mrl", "isp", "mip", "cclm", "lfnst", "mts", "alf", "ccalf", "dqp", "ts", "bdpcm", "big_resi", "ibc"]
INTER_SWITCHES = ["tmvp", "sbtmvp", "bdof", "dmvr", "mmvd", "affine", "ciip", "gpm", "amvr", "bcw", "smvd", "sbt"]


def test_gpu_fuzz_streams(tmp_path):
    """bitstreams with mutated tool mixes through the real parser and the drop-in library on the HIP back-end: output MD5 == the reference decoder's"""
    import mini_vvenc as mv
    import dropin_decode as dd
    if not (os.path.exists(dd.APP_DROPIN) and os.path.exists(os.path.join(ROOT, "oracle", "_ref", "vvdecapp_ref"))):
        pytest.skip("oracle/_ref/vvdecapp_dropin or vvdecapp_ref not built")
    seed0 = _seed()
    rnd = random.Random(seed0 ^ 0x5a5a)
    tables, renorm = mv.load_context_tables()
    want_n, budget = int(os.environ.get("VVR_FUZZ_STREAMS", "40")), float(os.environ.get("VVR_FUZZ_SECONDS", "75"))
    t0 = time.time()
    streams = refused = 0
    seed = 70000 + seed0 % 100000
    bad = []
    fixtures = list(mv.FIXTURES)
    while streams < want_n and time.time() - t0 < budget:
        name, kw, n, _ = fixtures[rnd.randrange(len(fixtures))]
        kw = dict(kw)
        if rnd.random() < 0.7:
            flips = INTRA_SWITCHES + (INTER_SWITCHES if kw.get("inter") else [])
            for k in rnd.sample(flips, rnd.randrange(1, 5)):
                kw[k] = not kw.get(k, False)
            if kw.get("ccalf") and not kw.get("alf"):
                kw["alf"] = True
            if rnd.random() < 0.15:
                kw["max_tb64"] = not kw.get("max_tb64", True)      # (the largest transform 32 instead of 64 or back: CUs of 64 then come in several transform units - finding 13)
            if not kw.get("max_tb64", True):
                kw["ciip"] = False                                 # (a CIIP CU of several transform units is refused by the back-end: DESIGN.md section 8)
            kw["qp"] = rnd.choice([22, 27, 32, 37, 42]); kw["p_split"] = rnd.choice([0.3, 0.6, 0.8]); kw["p_cbf"] = rnd.choice([0.2, 0.5, 0.9]); kw["p_cbf_chroma"] = rnd.choice([0.1, 0.4, 0.8])
            if "mtt_depth" not in kw or rnd.random() < 0.3:
                kw["mtt_depth"] = rnd.choice([0, 1, 2])
        seed += 1
        try:
            data, _ = mv.write_stream(mv.Cfg(**kw), n, seed, tables, renorm)
        except Exception:
            continue
        bit = str(tmp_path / "s.bit")
        open(bit, "wb").write(data)
        try:
            md5, _, log = mv.reference_md5(bit)
            if "ERROR" in log:
                raise RuntimeError("broken picture")
        except Exception:
            refused += 1
            continue
        open(bit[:-4] + ".yuv.md5", "w").write(md5 + "\n")
        r = dd.decode_stream(bit, threads=4, with_reference=False)          # (MD5 over the output frames and the hash SEI of every picture, like test_decodes_conformance_bitstreams)
        ok, out = r["ok"], str(r.get("dropin")) + str(r.get("dropin_dph"))
        streams += 1
        if not ok:
            keep = os.path.join(ROOT, "gpurun_out", "fuzz_differ_%s_seed%d.bit" % (name, seed))
            os.makedirs(os.path.dirname(keep), exist_ok=True)
            open(keep, "wb").write(data)
            bad.append((name, seed, kw, out[-300:]))
    assert not bad, "fuzz seed %d: %r" % (seed0, bad)
    assert streams >= 6, "only %d streams in the time budget (%d refused by the reference decoder)" % (streams, refused)
    print("fuzz seed %d: %d streams bit-exact through the drop-in, %d refused by the reference decoder, %.0f s" % (seed0, streams, refused, time.time() - t0))
