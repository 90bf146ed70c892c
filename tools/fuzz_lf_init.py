"""developer helper: randomised sweep of the generator's parameter space comparing the edge parameters the back-end derives itself (VVR_TOOL_LFP_ON_DEVICE: the product's
host code on the stand-in runtime of tests/hoststub, whose launch_lf_init runs vvdec_amd/csrc/vvr_lf_init.h - the source of k_lf_init - on the CPU) with the tables
the REFERENCE derives for the same picture (LoopFilter::calcFilterStrengthsCTU through oracle/_ref, refdrv.extract with DERIVE_LFP).  Compared: what the deblocking
kernels read of an entry (tests/test_lf_init.py::effective_differences).  Usage: tools/fuzz_lf_init.py <seed> <seconds>"""
import sys, random, time, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import numpy as np, refdrv
from vvdec_amd import abi, synth, stream
import test_host_glue as T
import test_lf_init as LF
BASE = (abi.TOOL_SAO_LUMA | abi.TOOL_SAO_CHROMA | abi.TOOL_ALF | abi.TOOL_CCALF | abi.TOOL_DEP_QUANT | abi.TOOL_MTS | abi.TOOL_LFNST | abi.TOOL_BDOF | abi.TOOL_DMVR | abi.TOOL_PROF)
OPT = [abi.TOOL_LMCS, abi.TOOL_LMCS | abi.TOOL_LMCS_CSCALE, abi.TOOL_WP, abi.TOOL_IBC, abi.TOOL_LADF, abi.TOOL_NO_LF_ACROSS_SLICES, abi.TOOL_NO_LF_ACROSS_TILES, abi.TOOL_AFFINE_MV_ON_DEVICE]
plans, _ = stream.ra_plan(5, gop=4, seed_poc0_is_external=False)


def sweep(seed, seconds):
    stub = C.CDLL(T.build_stub())
    stub.vvr_last_error.restype = C.c_char_p; stub.vvr_last_error.argtypes = [C.c_void_p]
    stub.vvr_prepare.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]; stub.vvr_submit_prepared.argtypes = [C.c_void_p, C.c_void_p]
    stub.vvr_free_prepared.argtypes = [C.c_void_p, C.c_void_p]; stub.vvr_destroy.argtypes = [C.c_void_p]; stub.vvr_sync.argtypes = [C.c_void_p]
    rnd = random.Random(seed)
    t_end = time.time() + seconds
    n = bad = 0
    while time.time() < t_end:
        s = rnd.randrange(1 << 30)
        W, H = rnd.choice([(128, 64), (200, 136), (256, 128), (264, 200), (320, 192), (384, 256), (512, 384)])
        l2 = rnd.choice([5, 6, 7]); idx = rnd.randrange(5)
        tools = BASE
        for o in OPT:
            if rnd.random() < 0.3: tools |= o
        kw = dict(p_intra=rnd.choice([0.0, 0.1, 0.3, 0.6]), p_split_scale=rnd.choice([0.5, 1.0, 1.5, 2.0]), p_coded=rnd.choice([0.1, 0.4, 0.9]), p_coded_chroma=rnd.choice([0.1, 0.5]),
                  p_jccr=rnd.choice([0, 0.4]), p_bdpcm=rnd.choice([0, 0.3]), p_affine=rnd.choice([0, 0.3, 0.6]), p_geo=rnd.choice([0, 0.2]), p_ciip=rnd.choice([0, 0.3]),
                  p_sbtmvp=rnd.choice([0, 0.3]), p_sbt=rnd.choice([0, 0.3]), p_isp=rnd.choice([0, 0.4]), p_ibc=rnd.choice([0, 0.4]), mv_sigma=rnd.choice([1.0, 8.0, 40.0]), p_mip=rnd.choice([0, 0.3]))
        if rnd.random() < 0.3: kw["min_cu_log2"] = 2
        if rnd.random() < 0.3: kw["dual_tree"] = rnd.choice([1.0, 2.0, 3.0])
        if rnd.random() < 0.3: kw["num_slices"] = rnd.choice([2, 3, 4])
        if rnd.random() < 0.3: kw["tile_cols"], kw["tile_rows"] = rnd.choice([(2, 1), (2, 2), (3, 2)])
        if rnd.random() < 0.25: kw["virtual_boundaries"] = rnd.choice([1 | (1 << 2), 2 | (2 << 2) | 16, 2 | (1 << 2), 1 | (2 << 2) | 16])      # (the reference's LF_INIT refuses three in a direction, LoopFilter.cpp:522)
        if rnd.random() < 0.15 and "tile_cols" in kw: kw["subpics"] = rnd.choice([1 | (1 << 1) | (1 << 3), 1 | (2 << 1) | (2 << 3), 1 | (2 << 1) | (1 << 3)])
        bd = rnd.choice([8, 10, 10]); cf = rnd.choice([1, 1, 1, 0])
        if not cf: kw.pop("min_cu_log2", None)                                                     # (combinations the generator does not make)
        if ( "tile_cols" in kw or "num_slices" in kw or "dual_tree" in kw ) and kw.get( "p_ibc" ): kw["p_ibc"] = 0
        if not cf: tools &= ~abi.TOOL_LMCS_CSCALE
        if (tools & abi.TOOL_LMCS_CSCALE) and not (tools & abi.TOOL_LMCS): tools |= abi.TOOL_LMCS
        pl = plans[idx]
        case = dict(W=W, H=H, l2=l2, idx=idx, seed=s, tools=hex(tools), bd=bd, cf=cf, kw=kw)
        try:
            d = synth.picture_for_plan(pl, W, H, seed=s, tool_flags=tools, log2_ctu=l2, bit_depth=bd, chroma_format=cf, **kw)
            refs = {}
            for lst in pl.ref_slots:
                for (slot, poc) in lst: refs.setdefault(slot, synth.natural_picture(W, H, s + 100 + poc, bit_depth=bd))
            want = refdrv.extract(d, refs, flags=refdrv.DERIVE_LFP)["lfp"]
            got = LF.derive(stub, d)
            diff = LF.effective_differences(want, got, d.w4, d.h4, cf != 0)
            n += 1
            if diff:
                bad += 1
                print("MISMATCH", case, diff[:3], flush=True)
        except Exception as e:
            n += 1; bad += 1
            print("EXC", repr(e)[:300], case, flush=True)
    return n, bad


if __name__ == "__main__":
    n, bad = sweep(int(sys.argv[1]), float(sys.argv[2]))
    print("cases", n, "bad", bad)
