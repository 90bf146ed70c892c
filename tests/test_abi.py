"""The C-ABI boundary without a GPU: the library loads, exports every symbol include/vvr.h declares, the ctypes mirror
matches the header as compiled by a C compiler, and the product refuses to run without a device (no CPU fallback)."""
import ctypes as C
import os
import re
import subprocess
import sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HDR = os.path.join(ROOT, "include", "vvr.h")

STRUCTS = [("vvr_pic_header", "PicHeader"), ("vvr_cu", "Cu"), ("vvr_tu", "Tu"), ("vvr_motion", "Motion"), ("vvr_lfp", "Lfp"),
           ("vvr_sao_ctu", "SaoCtu"), ("vvr_alf_ctu", "AlfCtu"), ("vvr_alf_params", "AlfParams"), ("vvr_lmcs_params", "LmcsParams"),
           ("vvr_picture", "Picture"), ("vvr_config", "Config"), ("vvr_kernel_stat", "KernelStat"),
           ("vvr_wp_params", "WpParams"), ("vvr_scaling_list", "ScalingList"), ("vvr_subpic", "Subpic"), ("vvr_slice_header", "SliceHeader"),
           ("vvr_rpr_ref", "RprRef"), ("vvr_rpr_params", "RprParams")]


def declared_symbols():
    src = open(HDR).read()
    return sorted(set(re.findall(r"VVR_API\s+[\w\s\*]+?\b(vvr_\w+)\s*\(", src)))


def test_header_declares_what_python_binds(built):
    import vvdec_amd
    assert declared_symbols() == sorted(vvdec_amd.EXPORTED_SYMBOLS)


def test_library_exports_every_declared_symbol(built):
    import vvdec_amd
    L = vvdec_amd.lib()
    for s in declared_symbols():
        assert hasattr(L, s), "libvvdec_amd.so does not export " + s
    # nothing but the ABI is visible (-fvisibility=hidden): no C++ symbols leak
    out = subprocess.check_output(["nm", "-D", "--defined-only", os.path.join(ROOT, "vvdec_amd", "libvvdec_amd.so")]).decode()
    exported = [l.split()[-1] for l in out.splitlines() if " T " in l]
    assert sorted(exported) == declared_symbols()
    assert b"gfx950" in L.vvr_version()


def test_struct_layout_matches_a_c_compiler(built, tmp_path):
    """sizeof/offsetof as gcc sees include/vvr.h == the ctypes mirror == what the library was compiled with"""
    import vvdec_amd
    from vvdec_amd import abi
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "vvr.h"', 'int main(void){']
    want = {}
    for cname, pyname in STRUCTS:
        cls = getattr(abi, pyname)
        lines.append('printf("%s %%zu\\n", sizeof(%s));' % (cname, cname))
        want[cname] = C.sizeof(cls)
        for f in cls._fields_:
            fname = f[0]
            lines.append('printf("%s.%s %%zu\\n", offsetof(%s, %s));' % (cname, fname, cname, fname))
            want["%s.%s" % (cname, fname)] = getattr(cls, fname).offset
    lines.append("return 0;}")
    src = tmp_path / "probe.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "probe"
    subprocess.check_call(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = dict((k, int(v)) for k, v in (l.split() for l in subprocess.check_output([str(exe)]).decode().splitlines()))
    assert got == want
    L = vvdec_amd.lib()
    for i, (cname, pyname) in enumerate(STRUCTS):
        assert L.vvr_abi_sizeof(i) == want[cname], cname
    assert L.vvr_abi_sizeof(len(STRUCTS)) == 0


def test_record_sizes_are_the_documented_ones():
    from vvdec_amd import abi
    assert C.sizeof(abi.Cu) == 112 and C.sizeof(abi.Tu) == 44
    assert C.sizeof(abi.Motion) == 20      # MotionInfo, CommonLib/MotionInfo.h:122
    assert C.sizeof(abi.Lfp) == 8          # LoopFilterParam (6 B, CommonLib/TypeDef.h:694) padded to 8


def test_no_cpu_fallback(built):
    """Without a gfx950 device the product must fail loudly."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    import vvdec_amd
    with pytest.raises(vvdec_amd.VvrError):
        vvdec_amd.Reconstructor(128, 128)
    # bad ABI version / null arguments are parameter errors, not crashes
    from vvdec_amd import abi
    L = vvdec_amd.lib()
    cfg = abi.Config()
    ctx = C.c_void_p()
    assert L.vvr_create(C.byref(cfg), C.byref(ctx)) == abi.VVR_ERR_PARAMETER
    assert L.vvr_create(None, C.byref(ctx)) == abi.VVR_ERR_PARAMETER


def test_product_does_not_link_the_oracle(built):
    """The shipped library must not depend on, or contain, the CPU oracle."""
    so = os.path.join(ROOT, "vvdec_amd", "libvvdec_amd.so")
    deps = subprocess.check_output(["ldd", so]).decode()
    assert "vvoracle" not in deps and "vvref" not in deps
    syms = subprocess.check_output(["nm", "-D", so]).decode()
    assert "vvo_" not in syms and "vvref_" not in syms
    for root, _, files in os.walk(os.path.join(ROOT, "vvdec_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                txt = open(os.path.join(root, f), errors="ignore").read()
                assert "libvvoracle" not in txt and "refdrv" not in txt and "vvo_reconstruct" not in txt, f


def test_tr_type_helper(built):
    """vvr_resolve_tr_type is a pure host function (TrQuant::getTrTypes, TrQuant.cpp:330)."""
    import vvdec_amd
    from vvdec_amd import abi
    L = vvdec_amd.lib()
    L.vvr_resolve_tr_type.restype = C.c_uint8
    h = abi.PicHeader(); cu = abi.Cu(); tu = abi.Tu()
    cu.pred_mode = abi.PRED_INTRA; cu.w = cu.h = 16
    tu.w, tu.h = 16, 16
    # sps_mts_enabled_flag off -> DCT2/DCT2, also for ISP and SBT blocks and whatever the other switches say (TrQuant.cpp:346)
    assert L.vvr_resolve_tr_type(C.byref(h), C.byref(cu), C.byref(tu), 0, 1, 1, 1) == 0
    cu.isp_mode = 1
    assert L.vvr_resolve_tr_type(C.byref(h), C.byref(cu), C.byref(tu), 0, 0, 0, 0) == 0
    cu.isp_mode = 0
    inter = abi.Cu(); inter.pred_mode = abi.PRED_INTER; inter.w = inter.h = 16; inter.sbt_info = 1 | (1 << 4)
    assert L.vvr_resolve_tr_type(C.byref(h), C.byref(inter), C.byref(tu), 0, 0, 0, 1) == 0
    h.tool_flags = abi.TOOL_MTS
    assert L.vvr_resolve_tr_type(C.byref(h), C.byref(inter), C.byref(tu), 0, 0, 0, 1) != 0
    # MTS on, but neither implicit nor explicit selection -> DCT2/DCT2
    assert L.vvr_resolve_tr_type(C.byref(h), C.byref(cu), C.byref(tu), 0, 0, 0, 0) == 0
    # implicit MTS on an intra luma block 4..16 -> DST7 both ways ((ver<<2)|hor with DST7 = 2)
    assert L.vvr_resolve_tr_type(C.byref(h), C.byref(cu), C.byref(tu), 0, 1, 0, 0) == ((2 << 2) | 2)
    # chroma never uses MTS
    assert L.vvr_resolve_tr_type(C.byref(h), C.byref(cu), C.byref(tu), 1, 1, 0, 0) == 0
    # explicit MTS, mts_idx = MTS_DCT8_DST7 -> horizontal DCT8 (1), vertical DST7 (2)
    tu.mts_idx[0] = abi.MTS_DCT8_DST7
    assert L.vvr_resolve_tr_type(C.byref(h), C.byref(cu), C.byref(tu), 0, 0, 1, 0) == ((2 << 2) | 1)


@pytest.mark.parametrize("implicit", [0, 1])
def test_resolver_matches_reference_checked_streams(built, implicit):
    """vvr_resolve_tr_type (the host helper an integrator calls per TU) against the transform types of generated pictures whose
    reconstruction is checked against the reference decoder (explicit MTS, SBT by position, ISP / implicit DST-7, LFNST and MIP exemptions)"""
    import vvdec_amd
    from vvdec_amd import abi, synth, stream
    L = vvdec_amd.lib()
    L.vvr_resolve_tr_type.restype = C.c_uint8
    tools = abi.TOOL_MTS | abi.TOOL_LFNST | abi.TOOL_DEP_QUANT | (abi.TOOL_IMPLICIT_MTS if implicit else 0)
    plans, _ = stream.ra_plan(5, gop=4, seed_poc0_is_external=False)
    checked = 0
    for idx, kw in ((0, dict(p_isp=0.4, p_mip=0.2, p_lfnst=0.3, p_mts=0.4, p_coded=0.8)), (2, dict(p_intra=0.3, p_sbt=0.4, p_isp=0.3, p_mts=0.4, p_coded=0.8))):
        d = synth.picture_for_plan(plans[idx], 256, 128, seed=900 + idx, tool_flags=tools, log2_ctu=6, **kw)
        hdr = d.hdr
        cus = (abi.Cu * len(d.cu)).from_buffer_copy(d.cu.tobytes())
        tus = (abi.Tu * len(d.tu)).from_buffer_copy(d.tu.tobytes())
        for t in tus:
            cu = cus[t.cu]
            for comp in range(3):
                coded = (t.cbf >> comp) & 1
                if not coded or t.mts_idx[comp] == abi.MTS_SKIP:
                    continue
                got = L.vvr_resolve_tr_type(C.byref(hdr), C.byref(cu), C.byref(t), comp, implicit, 0 if implicit else 1, 1)
                assert got == t.tr_type[comp], "CU %dx%d pred %d isp %d sbt %d lfnst %d comp %d: %d != %d" % (cu.w, cu.h, cu.pred_mode, cu.isp_mode, cu.sbt_info, cu.lfnst_idx, comp, got, t.tr_type[comp])
                checked += 1
    assert checked > 100
