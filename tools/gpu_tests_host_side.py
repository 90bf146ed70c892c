"""developer helper (no GPU needed; run it before spending GPU time): run every GPU parity test's host side (validate / prepare / submit bookkeeping) against the stand-in HIP runtime: samples are all zero, the
oracle is replaced by zeros, so only host-side rejections or crashes show up"""
import sys, ctypes as C, inspect, traceback
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import numpy as np
import vvdec_amd
import test_host_glue as T
stub = C.CDLL(T.LIB)
vvdec_amd._lib = None
orig_lib = vvdec_amd.lib
vvdec_amd._LIBPATH = T.LIB            # the python binding now loads the stub build of the same host code
L = vvdec_amd.lib()
import refdrv
def fake_oracle(desc, refs=None, flags=0):
    nc = 3 if desc.hdr.chroma_format else 1
    return [np.zeros(desc.plane_shape(c), np.uint16) for c in range(nc)]
refdrv.oracle_reconstruct = fake_oracle
refdrv.oracle_dmvr = lambda n: np.zeros((n, 2), np.int32)
import test_gpu_parity as G
import pytest
ok = bad = 0
for name, fn in sorted(vars(G).items()):
    if not name.startswith("test_") or not callable(fn): continue
    marks = [m for m in getattr(fn, "pytestmark", []) if m.name == "parametrize"]
    sig = [p for p in inspect.signature(fn).parameters if p not in ("built", "monkeypatch")]
    calls = [{}]
    for m in marks:
        names = [x.strip() for x in m.args[0].split(",")]
        calls = [dict(c, **dict(zip(names, v if isinstance(v, (tuple, list)) and len(names) > 1 else (v,)))) for c in calls for v in m.args[1]]
    if "monkeypatch" in inspect.signature(fn).parameters or name == "test_golden_fixtures_reference_outputs":
        continue
    for kw in calls:
        try:
            fn(True, **kw); ok += 1
        except AssertionError as e:
            msg = str(e)[:200]
            # sample comparisons against fixtures / known outputs cannot hold without kernels; anything else is a host-side problem
            print("ASSERT", name, kw if len(str(kw)) < 80 else "...", msg); bad += 1
        except Exception as e:
            print("EXC", name, repr(e)[:300]); bad += 1
print("ok", ok, "bad", bad)
