"""developer helper: the configurations of tools/mini_vvenc.py's fixture streams with OTHER seeds, each stream decoded twice - by the reference decoder (oracle/_ref/
vvdecapp_ref: the expected output MD5) and by the reference's application on the DROP-IN library with the CPU oracle behind vvdec::DecLibRecon (tests/oraclestub:
the C ABI served by oracle/libvvoracle.so).  A difference is a flaw in the flattening of the parser's objects (integration/vvr_extract.h), in the back-end's
derivation of the deblocking edge parameters (vvr_lf_init.h) or in the oracle's arithmetic - the three things a parsed picture passes on its way to the kernels that
are pinned to the oracle.  No GPU.  Usage: tools/fuzz_dropin_on_the_oracle.py <first seed> <seconds> [mutate]   (mutate: also other mixes of the coding tools than the fixtures')"""
import os, re, sys, time, tempfile, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import mini_vvenc as mv
import dropin_decode as dd
import test_dropin_library as TD


INTRA_SWITCHES = ["sao", "lmcs", "jccr", "dep_quant", "mrl", "isp", "mip", "cclm", "lfnst", "mts", "alf", "ccalf", "dqp", "ts", "bdpcm", "big_resi", "ibc"]
INTER_SWITCHES = ["tmvp", "sbtmvp", "bdof", "dmvr", "mmvd", "affine", "ciip", "gpm", "amvr", "bcw", "smvd", "sbt"]
MUTATE = len(sys.argv) > 3 and sys.argv[3] == "mutate"


def main():
    import random
    seed0, seconds = int(sys.argv[1]), float(sys.argv[2])
    rnd = random.Random(seed0)
    dd.BACKEND = TD._oracle_backend()
    stub = TD._stub_path()           # the PRODUCT's host code on the stand-in HIP runtime: its record checks and work-list builder see every picture too (round 5, finding 13)
    tables, renorm = mv.load_context_tables()
    t_end = time.time() + seconds
    streams = refused = differ = 0
    tmp = tempfile.mkdtemp(prefix="oraclefuzz")
    seed = seed0
    while time.time() < t_end:
        for name, kw, n, _ in mv.FIXTURES:
            if time.time() >= t_end:
                break
            kw = dict(kw)
            if MUTATE and rnd.random() < 0.7:
                # (other mixes of the tools than the fixtures': a few switches flipped, other QPs and split / residual densities; what the writer cannot write or
                # the reference decoder does not take is passed over)
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
            try:
                data, _ = mv.write_stream(mv.Cfg(**kw), n, seed, tables, renorm)
            except Exception:
                continue
            bit = os.path.join(tmp, "s.bit")
            open(bit, "wb").write(data)
            try:
                md5, _, log = mv.reference_md5(bit)
                if "ERROR" in log:      # (a stream with a picture the reference's parser gives up on - the decoder conceals it and goes on: not what is compared here)
                    raise RuntimeError("broken picture")
            except Exception:
                refused += 1            # (a seed whose stream the reference decoder does not take)
                continue
            r, _ = dd.run_app(dd.APP_DROPIN, ["-b", bit, "-t", "4", "-v", "3", "-md5", md5], preload=dd.BACKEND, timeout=60)
            streams += 1
            out = r.stdout + r.stderr
            r2, _ = dd.run_app(dd.APP_DROPIN, ["-b", bit, "-t", "2", "-v", "3"], preload=stub, timeout=60)
            out2 = r2.stdout + r2.stderr
            if r2.returncode != 0 or re.search(r"vvdec_amd|exception", out2):
                out += " || the product's host code: " + out2[-300:]
                r = r2
            if r.returncode != 0 or re.search(r"WARNING:|runtime error|MD5 mismatch|vvdec_amd:|the product's host code", out):
                differ += 1
                keep = os.path.join(tmp, "differ_%s_seed%d.bit" % (name, seed))
                os.replace(bit, keep)
                print("DIFFER", name, "seed", seed, "rc", r.returncode, keep, out[-300:].replace("\n", " | "), flush=True)
        seed += 1
    print("streams", streams, "refused by the reference decoder", refused, "differ", differ)


if __name__ == "__main__":
    main()
