"""developer helper: a picture dumped by the oracle-backed C ABI of tests/oraclestub (VVR_ORACLE_DUMP_DIR) through the reference's own classes (oracle/_ref harness,
rebuilt from the description) and through the oracle, in this process: which of the two differs from what the real decoder put out for the picture tells a flaw of the
flattening from one of the oracle's arithmetic.  Usage: tools/replay_oracle_dump.py <dump dir> <poc> [reference decoder's yuv, frame index]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import numpy as np
import refdrv
from vvdec_amd import abi
from vvdec_amd.desc import PictureDesc, CU_DT, TU_DT, MOTION_DT, LFP_DT, SAO_DT, ALF_DT


def load(d, poc):
    f = lambda n: os.path.join(d, "poc%d_%s.bin" % (poc, n))
    has = lambda n: os.path.exists(f(n))
    raw = lambda n: open(f(n), "rb").read()
    hdr = abi.PicHeader.from_buffer_copy(raw("hdr"))
    p = PictureDesc(hdr.width, hdr.height, hdr.bit_depth, hdr.log2_ctu, hdr.chroma_format)
    p.hdr = hdr
    p.cu = np.frombuffer(raw("cu"), CU_DT).copy(); p.tu = np.frombuffer(raw("tu"), TU_DT).copy()
    p.ctu_first_cu = np.frombuffer(raw("ctu_first_cu"), np.uint32).copy(); p.coef = np.frombuffer(raw("coef"), np.int16).copy() if has("coef") else np.zeros(1, np.int16)
    p.lfp = [np.frombuffer(raw("lfp0"), LFP_DT).copy(), np.frombuffer(raw("lfp1"), LFP_DT).copy()]
    if has("motion"): p.motion = np.frombuffer(raw("motion"), MOTION_DT).copy()
    if has("sao"): p.sao = np.frombuffer(raw("sao"), SAO_DT).copy()
    if has("alf"): p.alf = np.frombuffer(raw("alf"), ALF_DT).copy()
    if has("alf_sets"):
        r = raw("alf_sets"); n = C.sizeof(abi.AlfParams)
        p.alf_sets = [abi.AlfParams.from_buffer_copy(r[o:o + n]) for o in range(0, len(r), n)]; p.alf_params = p.alf_sets[0]
    if has("lmcs"): p.lmcs = abi.LmcsParams.from_buffer_copy(raw("lmcs"))
    if has("wp_sets"):
        r = raw("wp_sets"); n = C.sizeof(abi.WpParams)
        p.wp_sets = [abi.WpParams.from_buffer_copy(r[o:o + n]) for o in range(0, len(r), n)]; p.wp = p.wp_sets[0]
    if has("scaling"): p.scaling = abi.ScalingList.from_buffer_copy(raw("scaling"))
    p.hdr.tool_flags &= ~abi.TOOL_LFP_ON_DEVICE           # (the dump holds the tables the back-end derived)
    refs = {}
    ncomp = 3 if hdr.chroma_format else 1
    for l in range(2):
        for i in range(hdr.num_ref[l]):
            s = hdr.ref_slot[l][i]
            if s not in refs and has("ref_%d_0" % s):
                refs[s] = [np.frombuffer(raw("ref_%d_%d" % (s, c)), np.uint16).reshape(p.plane_shape(c)).copy() for c in range(ncomp)]
    out = [np.frombuffer(raw("out_%d" % c), np.uint16).reshape(p.plane_shape(c)).copy() for c in range(ncomp)]
    return p, refs, out


if __name__ == "__main__":
    d, poc = sys.argv[1], int(sys.argv[2])
    p, refs, dumped = load(d, poc)
    ora = refdrv.oracle_reconstruct(p, refs)
    ref = refdrv.reconstruct(p, refs, flags=refdrv.SIMD)["planes"]
    print("oracle here == oracle in the back-end:", all(np.array_equal(a, b) for a, b in zip(ora, dumped)))
    for c, (a, b) in enumerate(zip(ora, ref)):
        nd = np.argwhere(a != b)
        print("comp %d: oracle vs reference classes: %d samples differ%s" % (c, len(nd), "" if not len(nd) else ", first at (x %d, y %d)" % (nd[0][1], nd[0][0])))
    if len(sys.argv) > 4:
        yuv = np.fromfile(sys.argv[3], np.uint16 if p.hdr.bit_depth > 8 else np.uint8); k = int(sys.argv[4])
        W, H = p.hdr.width, p.hdr.height; fs = W * H * 3 // 2
        Y = yuv[k * fs:k * fs + W * H].reshape(H, W)
        for name, planes in (("oracle", ora), ("reference classes", ref)):
            nd = np.argwhere(planes[0] != Y)
            print("luma, %s vs the decoder's frame %d: %d samples differ%s" % (name, k, len(nd), "" if not len(nd) else ", first at (x %d, y %d)" % (nd[0][1], nd[0][0])))
