#!/usr/bin/env python3
"""Conformance harness of the DROP-IN decoder: every `<dir>/<name>/<name>.bit` (the layout of the reference's conformance download,
CMakeLists.txt:509-527) is decoded by the reference's OWN application (source/App/vvdecapp, built by `make -C oracle vvdecapp` from the sources where
they lie) linked with oracle/_ref/libvvdec.so - the reference decoder whose DecLibRecon is integration/DecLibReconDropIn.cpp on libvvdec_amd.so -
exactly the way the reference's ctest does it (CMakeLists.txt:559: `vvdecapp -b x.bit -md5 <x.yuv.md5>`): the MD5 over all output frames must equal
the stored one.  A second run per stream checks the decoded picture hash SEI of every picture (`-dph`, verifyPictureHash = 1: DecLib.cpp:504), and
`--with-reference` also runs the reference's own library (oracle/_ref/vvdecapp_ref, CPU) on the same stream and prints its frame rate beside the
drop-in's.

  python tools/dropin_decode.py [--dir ext/bitstreams] [--threads N] [--only substring] [--with-reference] [--json out.json]

Exit status 0 = every stream matched.  Needs a gfx950 device (the back-end has no CPU path); TEST INFRASTRUCTURE - nothing here is shipped."""
import argparse
import glob
import json
import os
import re
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref")
APP_DROPIN, APP_REF = os.path.join(REF, "vvdecapp_dropin"), os.path.join(REF, "vvdecapp_ref")
BACKEND = os.environ.get("VVDEC_AMD_LIB") or os.path.join(ROOT, "vvdec_amd", "libvvdec_amd.so")


def expected_md5(bit):
    """<name>.yuv.md5 next to the stream (read_bitstream_yuv_md5, CMakeLists.txt:536-542): its first 32 characters"""
    for cand in (re.sub(r"\.bit$", ".yuv.md5", bit), re.sub(r"\.bit$", "_yuv.md5", bit), re.sub(r"\.bit$", ".md5", bit)):
        if os.path.exists(cand):
            return open(cand).read(32).strip().lower()
    return None


def run_app(app, args, preload=None, timeout=600):
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = REF + os.pathsep + os.path.dirname(BACKEND) + os.pathsep + env.get("LD_LIBRARY_PATH", "")
    if preload:
        env["LD_PRELOAD"] = preload + (" " + env["LD_PRELOAD"] if env.get("LD_PRELOAD") else "")      # the drop-in's vvr_* calls bind to the back-end library
    t0 = time.perf_counter()
    r = subprocess.run([app] + args, capture_output=True, text=True, env=env, timeout=timeout)
    return r, time.perf_counter() - t0


def frames_and_fps(text):
    m = re.search(r"(\d+)\s+frames?\s+decoded", text) or re.search(r"Total Frames:\s*(\d+)", text)
    f = re.search(r"@\s*([0-9.]+)\s*fps", text)
    return (int(m.group(1)) if m else None), (float(f.group(1)) if f else None)


def decode_stream(bit, threads, with_reference):
    md5 = expected_md5(bit)
    res = {"stream": os.path.basename(bit), "expected_md5": md5}
    common = ["-b", bit, "-t", str(threads), "-v", "3"]
    # 1. the ctest command of the reference: MD5 over the output frames against the stored one (exit status 0 = match)
    r, dt = run_app(APP_DROPIN, common + (["-md5", md5] if md5 else []), preload=BACKEND)
    out = r.stdout + r.stderr
    res["dropin"] = {"rc": r.returncode, "seconds": round(dt, 2), "frames": frames_and_fps(out)[0], "fps": frames_and_fps(out)[1], "md5_checked": md5 is not None}
    # the ctest fails a run that prints a warning (FAIL_REGULAR_EXPRESSION "(WARNING:|runtime error)", CMakeLists.txt:574)
    res["dropin"]["warnings"] = bool(re.search(r"WARNING:|runtime error", out))
    if r.returncode != 0 or res["dropin"]["warnings"]:
        res["dropin"]["tail"] = out[-1500:]
    # 2. the decoded picture hash SEI of every picture (where the stream carries them)
    r2, dt2 = run_app(APP_DROPIN, common + ["-dph"], preload=BACKEND)
    out2 = r2.stdout + r2.stderr
    res["dropin_dph"] = {"rc": r2.returncode, "seconds": round(dt2, 2), "mismatch": bool(re.search(r"MD5 mismatch|CRC mismatch|Checksum mismatch|\(\*\*\*ERROR\*\*\*\)", out2))}
    if r2.returncode != 0 or res["dropin_dph"]["mismatch"]:
        res["dropin_dph"]["tail"] = out2[-1500:]
    if with_reference and os.path.exists(APP_REF):
        r3, dt3 = run_app(APP_REF, common + (["-md5", md5] if md5 else []))
        res["reference"] = {"rc": r3.returncode, "seconds": round(dt3, 2), "frames": frames_and_fps(r3.stdout + r3.stderr)[0], "fps": frames_and_fps(r3.stdout + r3.stderr)[1]}
    res["ok"] = res["dropin"]["rc"] == 0 and not res["dropin"]["warnings"] and res["dropin_dph"]["rc"] == 0 and not res["dropin_dph"]["mismatch"]
    return res


def find_streams(directory, only=None):
    s = sorted(glob.glob(os.path.join(directory, "*", "*.bit")) + glob.glob(os.path.join(directory, "*.bit")))
    return [b for b in s if not only or only in os.path.basename(b)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dir", default=os.environ.get("VVDEC_BITSTREAMS", os.path.join(ROOT, "ext", "bitstreams")))
    ap.add_argument("--threads", type=int, default=8, help="threads of the decoder's pool (vvdecapp -t)")
    ap.add_argument("--only", default=None)
    ap.add_argument("--with-reference", action="store_true")
    ap.add_argument("--json", default=None)
    a = ap.parse_args()
    if not os.path.exists(APP_DROPIN):
        sys.exit("oracle/_ref/vvdecapp_dropin is missing: run `make -C oracle vvdecapp` where /root/reference is present")
    streams = find_streams(a.dir, a.only)
    if not streams:
        print("no bitstreams under %s (expected <name>/<name>.bit + <name>.yuv.md5)" % a.dir)
        return 2
    results = []
    for b in streams:
        r = decode_stream(b, a.threads, a.with_reference)
        results.append(r)
        print("%-4s %-44s frames %s, drop-in %.1f s%s%s" % ("ok" if r["ok"] else "FAIL", r["stream"], r["dropin"]["frames"], r["dropin"]["seconds"],
              ", %.1f fps" % r["dropin"]["fps"] if r["dropin"]["fps"] else "", ", reference %.1f s" % r["reference"]["seconds"] if "reference" in r else ""), flush=True)
        if not r["ok"]:
            print("     " + (r["dropin"].get("tail") or r["dropin_dph"].get("tail") or "")[-600:].replace("\n", "\n     "))
    ok = sum(1 for r in results if r["ok"])
    print("%d of %d streams bit-exact (output MD5 + decoded picture hashes)" % (ok, len(results)))
    if a.json:
        json.dump({"streams": results, "ok": ok, "total": len(results)}, open(a.json, "w"), indent=1)
    return 0 if ok == len(results) else 1


if __name__ == "__main__":
    sys.exit(main())
