"""developer helper: md5 of everything vvr_prepare uploads (work lists, unit tables, description arrays) for a corpus of generated
pictures, through the stand-in HIP runtime of tests/hoststub.  Usage: tools/host_tables_hash.py out.json -- run before and after a
host-side restructuring and compare the two files."""
import sys, hashlib, ctypes as C, json
sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo")
import numpy as np
from vvdec_amd import abi, synth, stream
import test_host_glue as T
import subprocess, os
subprocess.check_call(["g++", "-std=c++17", "-O1", "-fPIC", "-shared", "-pthread", "-Wl,-Bsymbolic", "-I" + T.HIP_INC, "-D__HIP_PLATFORM_AMD__", "-w", T.SRC, "-o", T.LIB])
L = C.CDLL(T.LIB)
L.vvr_prepare.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]; L.vvr_free_prepared.argtypes = [C.c_void_p, C.c_void_p]; L.vvr_destroy.argtypes = [C.c_void_p]
L.vvr_last_error.restype = C.c_char_p; L.vvr_last_error.argtypes = [C.c_void_p]
L.vvt_table.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
out = {}
extra = [("wp_sl", 256, 128, 5, 4, 520, T.TOOLS | abi.TOOL_WP | abi.TOOL_SCALING_LIST, dict(p_intra=0.2, p_affine=0.2, p_sbtmvp=0.2, p_geo=0.1, p_sbt=0.2)),
         ("mono8", 256, 128, 3, 2, 521, T.TOOLS | abi.TOOL_LMCS, dict(bit_depth=8, chroma_format=0, p_intra=0.3, p_mip=0.2))]
for (name, W, H, frames, gop, seed, tools, kw) in T.STREAMS + extra:
    kw = dict(kw); l2 = kw.pop("log2_ctu", 7)
    geo = dict(bit_depth=kw.get("bit_depth", 10), chroma_format=kw.get("chroma_format", 1))
    plans, nslots = stream.ra_plan(frames, gop=gop, seed_poc0_is_external=False)
    ctx = T.Ctx(L, W, H, nslots, log2_ctu=l2, **geo)
    hs = []
    for pl in plans:
        d = synth.picture_for_plan(pl, W, H, seed=seed, tool_flags=tools, log2_ctu=l2, **kw)
        h = ctx.prepare(d)
        p, n = C.c_void_p(), C.c_size_t()
        parts = []
        for which in range(11):                     # logical tables (independent of how the upload is laid out)
            assert L.vvt_table(h, which, C.byref(p), C.byref(n)) == 0
            parts.append(hashlib.md5(C.string_at(p.value, n.value) if n.value else b"").hexdigest()[:12] + ":%d" % n.value)
        hs.append(" ".join(parts))
        L.vvr_free_prepared(ctx.ctx, h)
    ctx.close()
    out[name] = hs
json.dump(out, open(sys.argv[1], "w"), indent=1)
print("pictures hashed:", sum(len(v) for v in out.values()))
