"""developer helper: the configurations of tools/mini_vvenc.py's fixture streams with OTHER seeds, each stream decoded twice - by the reference decoder (oracle/_ref/
vvdecapp_ref: the expected output MD5) and by the reference's application on the DROP-IN library with the CPU oracle behind vvdec::DecLibRecon (tests/oraclestub:
the C ABI served by oracle/libvvoracle.so).  A difference is a flaw in the flattening of the parser's objects (integration/vvr_extract.h), in the back-end's
derivation of the deblocking edge parameters (vvr_lf_init.h) or in the oracle's arithmetic - the three things a parsed picture passes on its way to the kernels that
are pinned to the oracle.  No GPU.  Usage: tools/fuzz_dropin_on_the_oracle.py <first seed> <seconds>"""
import os, re, sys, time, tempfile, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import mini_vvenc as mv
import dropin_decode as dd
import test_dropin_library as TD


def main():
    seed0, seconds = int(sys.argv[1]), float(sys.argv[2])
    dd.BACKEND = TD._oracle_backend()
    tables, renorm = mv.load_context_tables()
    t_end = time.time() + seconds
    streams = refused = differ = 0
    tmp = tempfile.mkdtemp(prefix="oraclefuzz")
    seed = seed0
    while time.time() < t_end:
        for name, kw, n, _ in mv.FIXTURES:
            if time.time() >= t_end:
                break
            try:
                data, _ = mv.write_stream(mv.Cfg(**kw), n, seed, tables, renorm)
            except Exception:
                continue
            bit = os.path.join(tmp, "s.bit")
            open(bit, "wb").write(data)
            try:
                md5, _, _ = mv.reference_md5(bit)
            except Exception:
                refused += 1            # (a seed whose stream the reference decoder does not take)
                continue
            r, _ = dd.run_app(dd.APP_DROPIN, ["-b", bit, "-t", "4", "-v", "3", "-md5", md5], preload=dd.BACKEND)
            streams += 1
            out = r.stdout + r.stderr
            if r.returncode != 0 or re.search(r"WARNING:|runtime error|MD5 mismatch|vvdec_amd:", out):
                differ += 1
                keep = os.path.join(tmp, "differ_%s_seed%d.bit" % (name, seed))
                os.replace(bit, keep)
                print("DIFFER", name, "seed", seed, "rc", r.returncode, keep, out[-300:].replace("\n", " | "), flush=True)
        seed += 1
    print("streams", streams, "refused by the reference decoder", refused, "differ", differ)


if __name__ == "__main__":
    main()
