"""CPU: the drop-in decoder library oracle/_ref/libvvdec.so - the reference's own objects with the member functions of vvdec::DecLibRecon taken from
integration/DecLibReconDropIn.cpp (oracle/Makefile, target dropin).  Checked here, without a GPU (the back-end is the stand-in runtime of
tests/hoststub, which computes no sample): the export list is exactly the vvdec_* API of include/vvdec/vvdec.h.in, the library opens / refuses
garbage / flushes / closes through that API, and a picture goes through the drop-in class the way DecLib::reconPicture drives it.  What a picture
looks like after it is checked on the GPU (tests/test_gpu_parity.py::test_dropin_declibrecon).  Decoding a bitstream end to end needs a bitstream:
none is available offline; tests/test_dropin_library.py::test_decodes_conformance_bitstreams runs when ext/bitstreams/ exists."""
import ctypes as C
import glob
import os
import re
import subprocess
import sys

import numpy as np
import pytest

import refdrv
from vvdec_amd import abi, synth, stream

pytestmark = pytest.mark.skipif(not refdrv.dropin_available(), reason="oracle/_ref/libvvdec.so not built (needs /root/reference at build time)")
HERE = os.path.dirname(os.path.abspath(__file__))
API_H = "/root/reference/include/vvdec/vvdec.h.in"


def _stub_path():
    import test_host_glue as T
    return T.build_stub()


def test_exports_are_the_vvdec_api():
    out = subprocess.check_output(["nm", "-D", "--defined-only", refdrv.DROPIN_LIB]).decode()
    exported = sorted(l.split()[-1] for l in out.splitlines() if l.strip())
    # besides the C API the reference's library exports one C++ function for its application (VVDEC_DECL vvdec::rescalePlane, vvdecimpl.h:289): so does the drop-in
    cxx = [n for n in exported if not n.startswith("vvdec_")]
    assert all("rescalePlane" in n for n in cxx) and len(cxx) == 1, cxx
    exported = [n for n in exported if n.startswith("vvdec_")]
    if os.path.exists(API_H):
        declared = sorted(set(re.findall(r"VVDEC_DECL[^;(]*?\b(vvdec_\w+)\s*\(", open(API_H).read())))
        assert exported == declared, (set(exported) ^ set(declared))
    assert len(exported) == 26
    # the reconstruction stage in it is the drop-in's: its back-end calls are what the library leaves undefined
    und = subprocess.check_output(["nm", "-D", "--undefined-only", refdrv.DROPIN_LIB]).decode()
    assert "vvr_submit" in und and "vvr_wait" in und and "vvr_create" in und


class Params(C.Structure):          # vvdecParams (vvdec.h.in:487-502)
    _fields_ = [("threads", C.c_int), ("parseDelay", C.c_int), ("logLevel", C.c_int), ("verifyPictureHash", C.c_bool), ("filmGrainSynthesis", C.c_bool),
                ("simd", C.c_int), ("opaque", C.c_void_p), ("errHandlingFlags", C.c_int), ("reserved", C.c_int32 * 4)]


class AccessUnit(C.Structure):      # vvdecAccessUnit (vvdec.h.in:300-313)
    _fields_ = [("payload", C.POINTER(C.c_ubyte)), ("payloadSize", C.c_int), ("payloadUsedSize", C.c_int), ("cts", C.c_uint64), ("dts", C.c_uint64),
                ("ctsValid", C.c_bool), ("dtsValid", C.c_bool), ("rap", C.c_bool)]


@pytest.mark.parametrize("threads", [0, 2])
def test_open_decode_garbage_flush_close(threads):
    C.CDLL(_stub_path(), mode=C.RTLD_GLOBAL)
    L = C.CDLL(refdrv.DROPIN_LIB)
    L.vvdec_get_version.restype = C.c_char_p
    assert re.match(rb"\d+\.\d+", L.vvdec_get_version())
    L.vvdec_params_alloc.restype = C.POINTER(Params)
    L.vvdec_decoder_open.restype = C.c_void_p
    L.vvdec_decoder_open.argtypes = [C.POINTER(Params)]
    L.vvdec_accessUnit_alloc.restype = C.POINTER(AccessUnit)
    L.vvdec_accessUnit_alloc_payload.argtypes = [C.POINTER(AccessUnit), C.c_int]
    L.vvdec_decode.argtypes = [C.c_void_p, C.POINTER(AccessUnit), C.POINTER(C.c_void_p)]
    L.vvdec_flush.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
    L.vvdec_decoder_close.argtypes = [C.c_void_p]
    L.vvdec_get_last_error.restype = C.c_char_p
    L.vvdec_get_last_error.argtypes = [C.c_void_p]
    p = L.vvdec_params_alloc()
    L.vvdec_params_default(p)
    p.contents.threads = threads
    p.contents.logLevel = 0
    dec = L.vvdec_decoder_open(p)
    assert dec, "vvdec_decoder_open failed"
    au = L.vvdec_accessUnit_alloc()
    L.vvdec_accessUnit_alloc_payload(au, 256)
    rng = np.random.default_rng(7)
    data = bytes([0, 0, 0, 1]) + bytes(rng.integers(0, 256, 200, dtype=np.uint8))     # a start code and noise: no parameter sets, nothing decodable
    C.memmove(au.contents.payload, data, len(data))
    au.contents.payloadUsedSize = len(data)
    frame = C.c_void_p()
    rc = L.vvdec_decode(dec, au, C.byref(frame))
    assert rc <= 0 and not frame.value, rc                   # VVDEC_TRY_AGAIN / VVDEC_ERR_*: never a picture
    rc = L.vvdec_flush(dec, C.byref(frame))
    assert rc in (-50, -40, -30, -20, -10, -8, -7, -5, -3, -2, 0) or rc < 0 or not frame.value      # VVDEC_EOF (-50) on a healthy decoder; an error code after the garbage
    assert not frame.value
    L.vvdec_accessUnit_free(au)
    assert L.vvdec_decoder_close(dec) == 0
    L.vvdec_params_free(p)


@pytest.mark.parametrize("threads", [0, 3])
def test_picture_through_the_dropin_class(threads):
    """vvdec::DecLibRecon (drop-in) driven like DecLib::reconPicture drives it, on the stand-in runtime: the barrier task runs on the decoder's pool
    (or on the calling thread), LF_INIT + flatten + submit + wait + planes back + TaskFinishMotionInfo complete, reconDone is released, the motion
    field of the picture is what the description was built from (no refinement on the stand-in)"""
    W, H = 256, 128
    plans, _ = stream.ra_plan(5, gop=4, seed_poc0_is_external=False)
    ALL = abi.TOOL_SAO_LUMA | abi.TOOL_SAO_CHROMA | abi.TOOL_ALF | abi.TOOL_CCALF | abi.TOOL_DEP_QUANT | abi.TOOL_MTS | abi.TOOL_LFNST | abi.TOOL_BDOF | abi.TOOL_DMVR | abi.TOOL_PROF
    for idx, tools, kw in ((0, ALL, dict(p_cclm=0.3, p_mip=0.2)), (2, ALL | abi.TOOL_STILL_REF | abi.TOOL_LMCS | abi.TOOL_LMCS_CSCALE, dict(p_intra=0.2, p_affine=0.2, p_sbtmvp=0.1, p_ciip=0.1))):
        pl = plans[idx]
        d = synth.picture_for_plan(pl, W, H, seed=741 + idx, tool_flags=tools, **kw)
        refs = {slot: synth.natural_picture(W, H, 750 + poc) for lst in pl.ref_slots for (slot, poc) in lst}
        planes, motion = refdrv.run_dropin(d, refs, _stub_path(), threads=threads)
        assert [p.shape for p in planes] == [d.plane_shape(c) for c in range(3)]
        if pl.slice_type != abi.SLICE_I:
            inter = d.motion["ref_idx"].max(axis=1) >= 0
            assert inter.any()
            assert np.array_equal(motion["ref_idx"][inter], d.motion["ref_idx"][inter])
            l0 = inter & (d.motion["ref_idx"][:, 0] >= 0)
            assert np.array_equal(motion["mv"][l0][:, 0], d.motion["mv"][l0][:, 0])


def _conformance_streams():
    sys.path.insert(0, os.path.join(HERE, "..", "tools"))
    import dropin_decode
    for d in (os.environ.get("VVDEC_BITSTREAMS"), os.path.join(HERE, "..", "ext", "bitstreams"), os.path.join(HERE, "bitstreams")):
        if d and dropin_decode.find_streams(d):
            return dropin_decode, d
    return dropin_decode, None


@pytest.mark.gpu
def test_decodes_conformance_bitstreams():
    """<dir>/<name>/<name>.bit + <name>.yuv.md5 (the layout of the reference's conformance download, CMakeLists.txt:509-571; also tests/bitstreams, where
    the streams written by tools/mini_vvenc.py live): decoded by the reference's own application on the drop-in library with the GPU back-end behind
    DecLibRecon, the way the reference's ctest does it - MD5 over the output frames == the stored one, and every decoded picture hash SEI checks"""
    dd, d = _conformance_streams()
    if d is None:
        pytest.skip("no bitstreams (ext/bitstreams/, tests/bitstreams/, $VVDEC_BITSTREAMS) in this environment")
    if not os.path.exists(dd.APP_DROPIN):
        pytest.skip("oracle/_ref/vvdecapp_dropin not built")
    bad = []
    for b in dd.find_streams(d):
        r = dd.decode_stream(b, threads=4, with_reference=False)
        if not r["ok"]:
            bad.append((r["stream"], r["dropin"].get("tail") or r["dropin_dph"].get("tail")))
    assert not bad, bad


@pytest.mark.gpu
def test_a_wrong_picture_hash_is_noticed(tmp_path):
    """the streams of tests/bitstreams carry a decoded-picture-hash SEI behind every picture (the reference decoder's own hashes, tools/mini_vvenc.py): with
    one byte of the last SEI changed the decode through the drop-in library reports the mismatch - the per-picture check is live, not vacuous"""
    sys.path.insert(0, os.path.join(HERE, "..", "tools"))
    import dropin_decode as dd
    src = os.path.join(HERE, "bitstreams", "mini_all_tools_ctu128_384x256", "mini_all_tools_ctu128_384x256.bit")
    if not os.path.exists(dd.APP_DROPIN) or not os.path.exists(src):
        pytest.skip("oracle/_ref/vvdecapp_dropin or the stream missing")
    d = bytearray(open(src, "rb").read())
    i = d.rfind(bytes([0, 0, 1, 0, (24 << 3) | 1]))          # the last suffix SEI NAL unit
    assert i > 0
    d[i + 12] ^= 0xFF                                        # a byte of the luma MD5
    bad = tmp_path / "bad.bit"
    bad.write_bytes(bytes(d))
    r = dd.decode_stream(str(bad), threads=4, with_reference=False)
    assert r["dropin"]["rc"] == 0 and (r["dropin_dph"]["mismatch"] or r["dropin_dph"]["rc"] != 0) and not r["ok"], r


def test_conformance_harness_on_the_stand_in_runtime(tmp_path):
    """tools/dropin_decode.py end to end without a GPU: the reference's application (oracle/_ref/vvdecapp_dropin) starts on the drop-in library with
    the stand-in back-end bound the way the tool binds the real one, refuses a stream that is noise without crashing, and the tool reports the
    stream as failed (an MD5 was expected)"""
    sys.path.insert(0, os.path.join(HERE, "..", "tools"))
    import dropin_decode as dd
    if not os.path.exists(dd.APP_DROPIN):
        pytest.skip("oracle/_ref/vvdecapp_dropin not built (needs /root/reference)")
    d = tmp_path / "noise"
    d.mkdir()
    rng = np.random.default_rng(3)
    (d / "noise.bit").write_bytes(bytes([0, 0, 0, 1]) + bytes(rng.integers(0, 256, 4000, dtype=np.uint8)))
    (d / "noise.yuv.md5").write_text("0123456789abcdef0123456789abcdef  noise.yuv\n")
    assert dd.find_streams(str(tmp_path)) == [str(d / "noise.bit")] and dd.expected_md5(str(d / "noise.bit")) == "0123456789abcdef0123456789abcdef"
    dd.BACKEND = _stub_path()
    r = dd.decode_stream(str(d / "noise.bit"), threads=2, with_reference=False)
    assert r["dropin"]["rc"] not in (-11, 139, 134) and not r["ok"], r            # no crash; nothing decodable, so the stored MD5 cannot match
    v, _ = dd.run_app(dd.APP_DROPIN, ["--version"], preload=_stub_path())
    assert v.returncode == 0 and re.search(r"\d+\.\d+", v.stdout + v.stderr)


def test_reference_unit_test_on_the_dropin_object_set():
    """the reference's own unit test (tests/vvdec_unit_test/vvdec_unit_test.cpp: scalar-vs-SIMD differential tests of its kernels), linked with
    the objects the drop-in library is made of - all of the reference's except DecoderLib/DecLibRecon.o, plus integration/DecLibReconDropIn.cpp
    (`make -C oracle unit_test`): passes"""
    exe = os.path.join(os.path.dirname(refdrv.DROPIN_LIB), "vvdec_unit_test_dropin")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/vvdec_unit_test_dropin not built (needs /root/reference)")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "success: all tests passed!" in r.stdout, (r.stdout[-800:], r.stderr[-400:])


def test_pictures_with_scaled_reference_pictures_through_the_dropin_class():
    """the drop-in on pictures whose reference pictures have another size (reference picture resampling), on the stand-in runtime: the context is
    sized by the SPS's maximum picture size, the reference pictures are uploaded at their own sizes (vvr_slot_picture_size), the extractor's
    vvr_rpr_params passes the product's validation and work-list builder, the picture comes back at its own size"""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_oracle_vs_ref import RPR_CASES, rpr_case, ALL
    for (W, H, l2, idx, seed, specs, win, colloc, kw) in RPR_CASES:
        kw = dict(kw)
        tools = ALL | kw.pop("tool_flags_extra", 0)
        d, refs = rpr_case(W, H, l2, idx, seed, specs, win=win, colloc=colloc, tools=tools, **kw)
        planes, motion = refdrv.run_dropin(d, refs, _stub_path(), threads=2)
        assert [p.shape for p in planes] == [d.plane_shape(c) for c in range(len(planes))]
        inter = d.motion["ref_idx"].max(axis=1) >= 0
        assert inter.any() and np.array_equal(motion["ref_idx"][inter], d.motion["ref_idx"][inter])


def _oracle_backend():
    """tests/oraclestub: the ten vvr_* entry points the drop-in calls, served by the CPU oracle (test infrastructure) -> path of the built library"""
    src = os.path.join(HERE, "oraclestub", "vvr_oracle_backend.cpp")
    lib = os.path.join(HERE, "oraclestub", "libvvr_oracle_backend.so")
    odir = os.path.join(HERE, "..", "oracle")
    refdrv.oracle_lib()
    deps = [src, os.path.join(HERE, "..", "include", "vvr.h"), os.path.join(HERE, "..", "vvdec_amd", "csrc", "vvr_lf_init.h"), os.path.join(odir, "libvvoracle.so")]
    if not os.path.exists(lib) or os.path.getmtime(lib) < max(os.path.getmtime(p) for p in deps):
        tmp = "%s.%d.tmp" % (lib, os.getpid())
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-pthread", "-w", src, "-L" + odir, "-lvvoracle", "-Wl,-rpath," + os.path.abspath(odir), "-o", tmp])
        os.replace(tmp, lib)
    return lib


def test_parser_fed_streams_end_to_end_on_the_cpu_oracle():
    """every stream of tests/bitstreams decoded by the reference's application on the DROP-IN library with the CPU ORACLE behind vvdec::DecLibRecon
    (tests/oraclestub binds the oracle to the C ABI; no GPU, no product kernel): output MD5 and the hash SEI of every picture equal the reference decoder's.
    What this pins without a GPU: the flattening of what the real parser leaves in CodingStructure (integration/vvr_extract.h, DecLibReconDropIn.cpp) and the
    back-end's derivation of the deblocking edge parameters (vvr_lf_init.h, the source of k_lf_init - the drop-in does not run the reference's LF_INIT) on parsed
    pictures; the HIP kernels are pinned to the same oracle by tests/test_gpu_parity.py, and the same streams go through them in test_decodes_conformance_bitstreams"""
    dd, d = _conformance_streams()
    if d is None or not os.path.exists(dd.APP_DROPIN):
        pytest.skip("no bitstreams or oracle/_ref/vvdecapp_dropin not built (needs /root/reference)")
    keep = dd.BACKEND
    dd.BACKEND = _oracle_backend()
    try:
        bad = []
        # (incl. the four streams with vectors beyond a wrap period that the sweep on this back-end found in round 4: tests/bitstreams/wraparound_*)
        for b in dd.find_streams(d):
            if "mini_4k_" in b and not os.environ.get("VVDEC_BIG_STREAMS"):
                continue            # (17 pictures of 3840x2176 take the oracle 47 s - bit-exact, MD5 and picture hashes, when the stream was committed; the GPU suite decodes it every time)
            r = dd.decode_stream(b, threads=4, with_reference=False)
            if not r["ok"]:
                bad.append((r["stream"], r["dropin"].get("tail") or r["dropin_dph"].get("tail")))
    finally:
        dd.BACKEND = keep
    assert not bad, bad


def test_few_slots_are_enough():
    """the back-end's DPB smaller than the decoder's pool of Picture objects (VVDEC_AMD_SLOTS): the least recently used slot that is neither needed by the
    picture (itself and every picture of its reference picture lists) nor still in flight is given up - bit-exact all the same; when even that does not
    leave a slot the picture is refused with a message that names the setting (a slot of a picture in flight is never taken).  On the CPU oracle behind the drop-in"""
    import re
    dd, d = _conformance_streams()
    if d is None or not os.path.exists(dd.APP_DROPIN):
        pytest.skip("no bitstreams or oracle/_ref/vvdecapp_dropin not built (needs /root/reference)")
    bit = [b for b in dd.find_streams(d) if "mini_inter_tools_ctu128_384x256" in b][0]
    keep, env_keep = dd.BACKEND, dict(os.environ)
    dd.BACKEND = _oracle_backend()

    def run(threads, **env):
        os.environ.update({k: str(v) for k, v in env.items()})
        r, _ = dd.run_app(dd.APP_DROPIN, ["-b", bit, "-t", str(threads), "-v", "3", "-md5", dd.expected_md5(bit)], preload=dd.BACKEND)
        out = r.stdout + r.stderr
        m = re.search(r"slots given up: (\d+), reference pictures uploaded from host memory: (\d+)", out)
        return r.returncode, int(m.group(1)) if m else 0, out
    try:
        for threads in (0, 1, 4):
            rc, given_up, out = run(threads, VVDEC_AMD_TIMES=1, VVDEC_AMD_SLOTS=5)
            assert rc == 0 and given_up > 0, out[-800:]
        rc, _, out = run(1, VVDEC_AMD_SLOTS=2)
        assert rc != 0 and "need more DPB slots" in out, out[-800:]
    finally:
        dd.BACKEND = keep
        os.environ.clear(); os.environ.update(env_keep)


def test_every_stream_passes_the_products_record_checks_on_the_stand_in_runtime():
    """every stream of tests/bitstreams decoded by the reference's application on the drop-in library with the PRODUCT's host code behind it (tests/hoststub: the
    stand-in HIP runtime, no kernel runs, the output is not looked at): the record checks and the work-list builder of vvr_prepare.cpp accept every picture the real
    parser produces.  The CPU-oracle run above cannot see this - the oracle back-end does not run those checks - and the GPU suite saw it late: the randomised GPU
    leg of round 5 found IBC CUs of 64x64 refused in a sequence whose largest transform is 32 (four transform units), which no CPU test would have caught"""
    dd, d = _conformance_streams()
    if d is None or not os.path.exists(dd.APP_DROPIN):
        pytest.skip("no bitstreams or oracle/_ref/vvdecapp_dropin not built (needs /root/reference)")
    bad = []
    for b in dd.find_streams(d):
        if "mini_4k_" in b and not os.environ.get("VVDEC_BIG_STREAMS"):
            continue
        r, _ = dd.run_app(dd.APP_DROPIN, ["-b", b, "-t", "2", "-v", "3"], preload=_stub_path())
        out = r.stdout + r.stderr
        if r.returncode != 0 or re.search(r"vvdec_amd|exception|ERROR", out):
            bad.append((os.path.basename(b), out[-400:]))
    assert not bad, bad
