"""developer helper: time of the host stage of vvr_submit (validation, work lists, packing) on this machine, against the stand-in runtime of
tests/hoststub (no GPU involved): ms per 4K picture for 0 / 4 / 8 / 16 worker threads.  Usage: python tools/host_path_probe.py"""
import ctypes as C, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from vvdec_amd import abi, synth, stream
import bench
import test_host_glue as T
lib = "/tmp/vvr_hoststub_o3.so"
subprocess.check_call(["g++", "-std=c++17", "-O3", "-fPIC", "-shared", "-pthread", "-Wl,-Bsymbolic", "-I" + T.HIP_INC, "-D__HIP_PLATFORM_AMD__", "-w", T.SRC, "-o", lib])
L = C.CDLL(lib)
L.vvr_submit.argtypes = [C.c_void_p, C.c_void_p]; L.vvr_sync.argtypes = [C.c_void_p]; L.vvr_destroy.argtypes = [C.c_void_p]
W, H = 3840, 2160
plans, nslots = stream.ra_plan(17, gop=16, seed_poc0_is_external=False, pool=24)
descs = [synth.picture_for_plan(pl, W, H, seed=1234, tool_flags=bench._tools(abi), **bench.MIX) for pl in plans[1:9]]
pics = [d.c() for d in descs]
for nt in (0, 4, 8, 16):
    cfg = abi.Config(); cfg.abi_version = abi.VVR_ABI_VERSION; cfg.max_width = W; cfg.max_height = H; cfg.chroma_format = 1; cfg.bit_depth = 10; cfg.log2_ctu = 7
    cfg.num_slots = 24; cfg.num_streams = 8; cfg.host_threads = nt
    ctx = C.c_void_p(); assert L.vvr_create(C.byref(cfg), C.byref(ctx)) == 0
    best = 1e9
    for rep in range(4):
        t0 = time.perf_counter(); n = 0
        for it in range(6):
            for p in pics:
                assert L.vvr_submit(ctx, C.byref(p)) >= 0; n += 1
        L.vvr_sync(ctx)
        best = min(best, (time.perf_counter() - t0) / n)
    print("host threads %2d: %.2f ms per 4K B picture (%.0f pictures/s)" % (nt, best * 1e3, 1 / best), flush=True)
    L.vvr_destroy(ctx)
