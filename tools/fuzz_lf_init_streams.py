"""developer helper: the configurations of tools/mini_vvenc.py's fixture streams with OTHER seeds - every stream is parsed by the reference's own parser inside the
drop-in decoder library, whose self-check (VVDEC_AMD_LF_INIT=2) compares the back-end's derivation of the deblocking edge parameters with the reference's LF_INIT
for every picture.  No GPU (the back-end is the stand-in runtime of tests/hoststub).  Usage: tools/fuzz_lf_init_streams.py <first seed> <seconds>"""
import os, re, sys, time, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import mini_vvenc as mv
import dropin_decode as dd
import test_host_glue as T


def main():
    seed0, seconds = int(sys.argv[1]), float(sys.argv[2])
    stub = T.build_stub()
    tables, renorm = mv.load_context_tables()
    os.environ["VVDEC_AMD_LF_INIT"] = "2"
    t_end = time.time() + seconds
    streams = pictures_checked = differ = undecodable = 0
    tmp = tempfile.mkdtemp(prefix="lfstreams")
    seed = seed0
    while time.time() < t_end:
        for name, kw, n, _ in mv.FIXTURES:
            if time.time() >= t_end:
                break
            if kw.get("deblock") is False:
                continue
            try:
                data, _ = mv.write_stream(mv.Cfg(**kw), n, seed, tables, renorm)
            except Exception as e:                      # (a seed the writer cannot finish - its own assertions)
                continue
            bit = os.path.join(tmp, "s.bit")
            open(bit, "wb").write(data)
            r, _ = dd.run_app(dd.APP_DROPIN, ["-b", bit, "-t", "4", "-v", "0"], preload=stub)
            out = r.stdout + r.stderr
            m = re.findall(r"edge parameters: (\d+) entries checked against the reference's LF_INIT, (\d+) differ", out)
            c, d = sum(int(a) for a, _ in m), sum(int(x) for _, x in m)
            streams += 1
            # (anything the back-end's host stage says about a stream the reference parsed: a refusal of something valid would show here)
            said = [l for l in out.splitlines() if "vvdec_amd" in l and "edge parameters" not in l]
            if said:
                keep = os.path.join(tmp, "said_%s_seed%d.bit" % (name, seed))
                open(keep, "wb").write(data)
                print("SAID", name, "seed", seed, "rc", r.returncode, said[:2], keep, flush=True)
            if not c:
                undecodable += 1
                continue
            pictures_checked += c
            if d:
                differ += d
                keep = os.path.join(tmp, "differ_%s_seed%d.bit" % (name, seed))
                os.replace(bit, keep)
                print("DIFFER", name, "seed", seed, d, "entries;", keep, [l for l in out.splitlines() if "differ:" in l][:2], flush=True)
        seed += 1
    print("streams", streams, "not decoded", undecodable, "entries checked", pictures_checked, "differ", differ)


if __name__ == "__main__":
    main()
