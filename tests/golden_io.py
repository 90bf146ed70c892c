"""Serialisation of one pre-parsed picture + the reference decoder's outputs for it (tests/golden/*.npz).

TEST INFRASTRUCTURE.  A fixture holds the complete input of DecLibRecon::decompressPicture as the C ABI sees it (header
bytes, CU/TU records, packed levels, motion field, edge parameters, SAO/ALF controls), the reference pictures it predicts
from and the planes the REAL reference classes (oracle/_ref, built from /root/reference by oracle/Makefile) produced after
each stage.  tests/golden/make_golden.py writes them; /root/reference is not needed to read them."""
import ctypes as C
import numpy as np
from vvdec_amd import abi
from vvdec_amd.desc import PictureDesc, CU_DT, TU_DT, MOTION_DT, LFP_DT, SAO_DT, ALF_DT

STAGES = ("reco", "dbk", "sao", "final")


def _bytes_of(struct):
    return np.frombuffer(bytes(struct), dtype=np.uint8).copy()


def save(path, desc, refs, outputs):
    """refs: {slot: [Y, Cb, Cr]}; outputs: {stage: [Y, Cb, Cr]} from the reference."""
    d = dict(hdr=_bytes_of(desc.hdr), cu=desc.cu.view(np.uint8), tu=desc.tu.view(np.uint8), ctu_first_cu=desc.ctu_first_cu,
             coef=np.asarray(desc.coef, np.int16), lfp0=desc.lfp[0].view(np.uint8), lfp1=desc.lfp[1].view(np.uint8))
    if desc.motion is not None:
        d["motion"] = desc.motion.view(np.uint8)
    if desc.sao is not None:
        d["sao"] = desc.sao.view(np.uint8)
    if desc.alf is not None:
        d["alf"] = desc.alf.view(np.uint8)
    if desc.alf_params is not None:
        d["alf_params"] = _bytes_of(desc.alf_params)
    if desc.lmcs is not None:
        d["lmcs"] = _bytes_of(desc.lmcs)
    if desc.wp is not None:
        d["wp"] = _bytes_of(desc.wp)
    if desc.scaling is not None:
        d["scaling"] = _bytes_of(desc.scaling)
    if desc.ctu_slice is not None:
        d["ctu_slice"] = np.asarray(desc.ctu_slice, np.uint16)
    if desc.ctu_tile is not None:
        d["ctu_tile"] = np.asarray(desc.ctu_tile, np.uint16)
    if desc.subpics is not None and len(desc.subpics):
        d["subpics"] = np.frombuffer(np.ascontiguousarray(desc.subpics).tobytes(), np.uint8)
    if desc.slices is not None and len(desc.slices):
        d["slices"] = np.frombuffer(np.ascontiguousarray(desc.slices, dtype=np.dtype(abi.SliceHeader)).tobytes(), np.uint8)
    if getattr(desc, "rpr", None) is not None:
        d["rpr"] = _bytes_of(desc.rpr)
    if desc.alf_sets:
        d["alf_sets"] = np.concatenate([_bytes_of(a) for a in desc.alf_sets])
    if desc.wp_sets:
        d["wp_sets"] = np.concatenate([_bytes_of(w) for w in desc.wp_sets])
    for slot, planes in refs.items():
        for c, p in enumerate(planes):
            d["ref_%d_%d" % (slot, c)] = np.asarray(p, np.uint16)
    for st, planes in outputs.items():
        for c, p in enumerate(planes):
            d["out_%s_%d" % (st, c)] = np.asarray(p, np.uint16)
    np.savez_compressed(path, **d)


def load(path):
    """-> (PictureDesc, refs {slot: planes}, outputs {stage: planes})"""
    z = np.load(path)
    raw = z["hdr"].tobytes()
    if len(raw) < C.sizeof(abi.PicHeader):           # fixture written with ABI version 1: the header has grown at its end (LADF parameters, zero = off)
        raw = raw + bytes(C.sizeof(abi.PicHeader) - len(raw))
    hdr = abi.PicHeader.from_buffer_copy(raw)
    hdr.abi_version = abi.VVR_ABI_VERSION
    d = PictureDesc(hdr.width, hdr.height, hdr.bit_depth, hdr.log2_ctu, hdr.chroma_format)
    d.hdr = hdr
    d.cu = z["cu"].view(CU_DT).copy()
    d.tu = z["tu"].view(TU_DT).copy()
    d.ctu_first_cu = z["ctu_first_cu"].astype(np.uint32)
    d.coef = z["coef"].astype(np.int16)
    d.lfp = [z["lfp0"].view(LFP_DT).copy(), z["lfp1"].view(LFP_DT).copy()]
    if "motion" in z:
        d.motion = z["motion"].view(MOTION_DT).copy()
    if "sao" in z:
        d.sao = z["sao"].view(SAO_DT).copy()
    if "alf" in z:
        d.alf = z["alf"].view(ALF_DT).copy()
    if "alf_params" in z:
        d.alf_params = abi.AlfParams.from_buffer_copy(z["alf_params"].tobytes())
    if "lmcs" in z:
        d.lmcs = abi.LmcsParams.from_buffer_copy(z["lmcs"].tobytes())
    if "wp" in z:
        d.wp = abi.WpParams.from_buffer_copy(z["wp"].tobytes())
    if "scaling" in z:
        d.scaling = abi.ScalingList.from_buffer_copy(z["scaling"].tobytes())
    if "ctu_slice" in z:
        d.ctu_slice = z["ctu_slice"].astype(np.uint16)
    if "ctu_tile" in z:
        d.ctu_tile = z["ctu_tile"].astype(np.uint16)
    if "subpics" in z:
        d.subpics = np.frombuffer(z["subpics"].tobytes(), np.dtype(abi.Subpic)).copy()
    if "slices" in z:
        d.slices = np.frombuffer(z["slices"].tobytes(), np.dtype(abi.SliceHeader)).copy()
    if "rpr" in z:
        d.rpr = abi.RprParams.from_buffer_copy(z["rpr"].tobytes())
    for key, cls in (("alf_sets", abi.AlfParams), ("wp_sets", abi.WpParams)):
        if key in z:
            raw = z[key].tobytes()
            setattr(d, key, [cls.from_buffer_copy(raw[o:o + C.sizeof(cls)]) for o in range(0, len(raw), C.sizeof(cls))])
    refs, outs = {}, {}
    for k in z.files:
        if k.startswith("ref_"):
            _, slot, c = k.split("_")
            refs.setdefault(int(slot), [None, None, None])[int(c)] = z[k]
        elif k.startswith("out_"):
            _, st, c = k.split("_")
            outs.setdefault(st, [None, None, None])[int(c)] = z[k]
    return d, refs, outs
