#!/usr/bin/env python3
"""developer helper / bench.py leg: the reference's application on the DROP-IN library (this back-end behind vvdec::DecLibRecon) decoding the parser-fed 4K stream
(tests/bitstreams/mini_4k_all_tools_ctu128_3840x2176: 3840x2176, 17 pictures, every tool) several times in one process (vvdecapp --loops): pictures per second of the
whole decoder - parsing, motion derivation, flattening, the GPU back-end, the planes back, output - and the host milliseconds per picture the drop-in's own
stages take (VVDEC_AMD_TIMES).  Prints one JSON object."""
import json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import dropin_decode as dd


def _stream(copies):
    """the committed stream `copies` times behind itself: every copy starts with an IDR picture and its own parameter sets - a valid stream of `copies` coded video
    sequences that the decoder pipelines across; 1: the committed file"""
    bit = os.path.join(ROOT, "tests", "bitstreams", "mini_4k_all_tools_ctu128_3840x2176", "mini_4k_all_tools_ctu128_3840x2176.bit")
    if copies <= 1 or not os.path.exists(bit):
        return bit
    import tempfile
    out = os.path.join(tempfile.gettempdir(), "mini_4k_x%d.bit" % copies)
    if not os.path.exists(out):
        data = open(bit, "rb").read()
        open(out, "wb").write(data * copies)
    return out


def rate(threads=16, loops=4, readback=True, copies=1):
    bit = _stream(copies)
    if not (os.path.exists(bit) and os.path.exists(dd.APP_DROPIN)):
        return None
    os.environ["VVDEC_AMD_TIMES"] = "1"
    if not readback:
        os.environ["VVDEC_AMD_NO_READBACK"] = "1"
    r, dt = dd.run_app(dd.APP_DROPIN, ["-b", bit, "-t", str(threads), "-v", "3", "-L", str(loops)], preload=dd.BACKEND, timeout=300)
    os.environ.pop("VVDEC_AMD_NO_READBACK", None)
    out = r.stdout + r.stderr
    fps = [float(x) for x in re.findall(r"frames decoded @ ([0-9.]+) fps", out)]
    host = re.findall(r"host ms per picture: MIDER ([0-9.]+), LF_INIT ([0-9.]+), flatten ([0-9.]+), submit\+device ([0-9.]+), planes back ([0-9.]+)", out)
    if r.returncode != 0 or not fps:
        return {"error": out[-400:]}
    res = {"pictures_per_s_per_loop": fps, "pictures_per_s": round(max(fps[1:] or fps), 1), "loops": len(fps), "pool_threads": threads, "stream": "3840x2176, %d x 17 pictures (I + one GOP of 16), every tool" % copies,
           "what": "vvdecapp on the drop-in libvvdec.so: the reference's parser and motion derivation on its thread pool, this back-end behind vvdec::DecLibRecon; best loop after the first (the first creates the context)"}
    if host:
        m = host[-1]
        res["host_ms_per_picture"] = {"mider": float(m[0]), "lf_init": float(m[1]), "flatten": float(m[2]), "submit_and_device": float(m[3]), "planes_back": float(m[4])}
    return res


def reference_rate(threads=16, loops=4, copies=1):
    """the reference decoder itself (oracle/_ref/vvdecapp_ref: its own DecLibRecon on the CPU) on the same stream, same thread count"""
    bit = _stream(copies)
    app = os.path.join(ROOT, "oracle", "_ref", "vvdecapp_ref")
    if not (os.path.exists(bit) and os.path.exists(app)):
        return None
    r, dt = dd.run_app(app, ["-b", bit, "-t", str(threads), "-v", "3", "-L", str(loops)], timeout=300)
    fps = [float(x) for x in re.findall(r"frames decoded @ ([0-9.]+) fps", r.stdout + r.stderr)]
    return {"pictures_per_s": round(max(fps[1:] or fps), 1), "pictures_per_s_per_loop": fps, "pool_threads": threads} if fps else None


if __name__ == "__main__":
    t = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    c = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    print(json.dumps({"with_read_back": rate(t, copies=c, loops=2 if c > 1 else 4), "without_read_back": rate(t, readback=False, copies=c, loops=2 if c > 1 else 4), "reference_decoder_on_the_cpu": reference_rate(t, copies=c, loops=2 if c > 1 else 4)}, indent=1))
