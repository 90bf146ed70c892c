"""vvdec_amd — MI355X-native VVC (H.266) reconstruction back-end.

Python is only plumbing here (ctypes binding of the C ABI in include/vvr.h, used by tests, bench.py and the multi-GPU
launcher).  The product is `libvvdec_amd.so`: hand-written HIP kernels for gfx950 + the C++ host scheduler in
vvdec_amd/csrc/.  There is no CPU fallback: if the library is missing or no gfx950 device is present, creating a
`Reconstructor` raises.

`Reconstructor` mirrors the reference's DecLibRecon (source/Lib/DecoderLib/DecLibRecon.h:143-200):
    create / destroy               -> Reconstructor(...) / close()
    decompressPicture(Picture*)    -> decompress_picture(desc)        (asynchronous, returns a job id)
    waitForPrevDecompressedPic()   -> wait(job)
"""
import ctypes as C
import os
import subprocess
import numpy as np
from . import abi
from .desc import PictureDesc

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBPATH = os.environ.get("VVDEC_AMD_LIB") or os.path.join(_HERE, "libvvdec_amd.so")      # (VVDEC_AMD_LIB: developer builds of the same library, e.g. `make watchdog`)
_lib = None


class VvrError(RuntimeError):
    pass


def build(force=False):
    """Compile the HIP library for gfx950 (hipcc cross-compiles without a GPU)."""
    src = os.path.join(_HERE, "csrc")
    if force and os.path.exists(_LIBPATH):
        os.remove(_LIBPATH)
    subprocess.check_call(["make", "-C", src], stdout=subprocess.DEVNULL)
    return _LIBPATH


def lib():
    """The loaded C-ABI library; raises if it has not been built (never falls back to a CPU path)."""
    global _lib
    if _lib is None:
        if not os.path.exists(_LIBPATH):
            raise VvrError("libvvdec_amd.so is missing: run `python -c 'import __graft_entry__ as g; g.build()'` (no CPU fallback exists)")
        L = C.CDLL(_LIBPATH)
        L.vvr_version.restype = C.c_char_p
        L.vvr_last_error.restype = C.c_char_p
        L.vvr_last_error.argtypes = [C.c_void_p]
        L.vvr_slot_bytes.restype = C.c_size_t
        L.vvr_abi_sizeof.restype = C.c_size_t
        L.vvr_plane_ptr.restype = C.c_void_p
        L.vvr_plane_ptr.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.vvr_job_stream.restype = C.c_void_p
        L.vvr_job_stream.argtypes = [C.c_void_p, C.c_int]
        for f in ("vvr_create", "vvr_submit", "vvr_wait", "vvr_sync", "vvr_read_plane", "vvr_write_plane", "vvr_prepare",
                  "vvr_submit_prepared", "vvr_enable_stats", "vvr_get_stats", "vvr_plane_layout"):
            getattr(L, f).restype = C.c_int
        L.vvr_destroy.argtypes = [C.c_void_p]
        L.vvr_wait.argtypes = [C.c_void_p, C.c_int]
        L.vvr_test.argtypes = [C.c_void_p, C.c_int]
        L.vvr_sync.argtypes = [C.c_void_p]
        L.vvr_submit.argtypes = [C.c_void_p, C.c_void_p]
        L.vvr_prepare.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.vvr_submit_prepared.argtypes = [C.c_void_p, C.c_void_p]
        L.vvr_free_prepared.argtypes = [C.c_void_p, C.c_void_p]
        L.vvr_read_plane.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_size_t]
        L.vvr_write_plane.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_size_t]
        L.vvr_picture_hash.restype = C.c_int
        L.vvr_picture_hash.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.vvr_read_output.restype = C.c_int
        L.vvr_read_output.argtypes = [C.c_void_p] + [C.c_int] * 7 + [C.c_void_p, C.c_size_t]
        L.vvr_read_dmvr.restype = C.c_int
        L.vvr_read_dmvr.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]
        L.vvr_read_col_motion.restype = C.c_int
        L.vvr_read_col_motion.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]
        L.vvr_enable_stats.argtypes = [C.c_void_p, C.c_int]
        L.vvr_get_stats.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.vvr_plane_layout.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.vvr_inputs_done.restype = C.c_int
        L.vvr_inputs_done.argtypes = [C.c_void_p, C.c_int]
        L.vvr_host_alloc.restype = C.c_void_p
        L.vvr_host_alloc.argtypes = [C.c_void_p, C.c_size_t]
        L.vvr_host_free.argtypes = [C.c_void_p, C.c_void_p]
        L.vvr_measure_copy_bandwidth.restype = C.c_double
        L.vvr_measure_copy_bandwidth.argtypes = [C.c_void_p, C.c_int]
        L.vvr_stream_wait_job.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        L.vvr_stream_wait_slot.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        L.vvr_slot_external_event.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        _lib = L
    return _lib


EXPORTED_SYMBOLS = ["vvr_version", "vvr_create", "vvr_destroy", "vvr_submit", "vvr_wait", "vvr_test", "vvr_sync", "vvr_slot_bytes", "vvr_plane_layout",
                    "vvr_plane_ptr", "vvr_read_plane", "vvr_read_output", "vvr_picture_hash", "vvr_write_plane", "vvr_read_dmvr", "vvr_read_col_motion", "vvr_prepare", "vvr_submit_prepared",
                    "vvr_free_prepared", "vvr_job_stream", "vvr_last_error", "vvr_enable_stats", "vvr_get_stats", "vvr_resolve_tr_type", "vvr_abi_sizeof",
                    "vvr_inputs_done", "vvr_measure_copy_bandwidth", "vvr_host_alloc", "vvr_host_free",
                    "vvr_stream_wait_job", "vvr_stream_wait_slot", "vvr_slot_external_event", "vvr_slot_picture_size", "vvr_read_picture"]


class Reconstructor:
    def __init__(self, width, height, bit_depth=10, log2_ctu=7, chroma_format=1, num_slots=8, num_streams=2, device=0, ext_planes=None, host_threads=0, stop_after=0, ring_entries=0):
        self.L = lib()
        cfg = abi.Config()
        cfg.abi_version = abi.VVR_ABI_VERSION
        cfg.device, cfg.max_width, cfg.max_height = device, width, height
        cfg.chroma_format, cfg.bit_depth, cfg.log2_ctu = chroma_format, bit_depth, log2_ctu
        cfg.num_slots, cfg.num_streams, cfg.host_threads, cfg.stop_after, cfg.ring_entries = num_slots, num_streams, host_threads, stop_after, ring_entries
        cfg.ext_planes = ext_planes
        self.cfg = cfg
        self.ctx = C.c_void_p()
        rc = self.L.vvr_create(C.byref(cfg), C.byref(self.ctx))
        if rc != abi.VVR_OK:
            raise VvrError("vvr_create failed with %d (%s)" % (rc, {abi.VVR_ERR_NO_DEVICE: "no gfx950 device; there is no CPU fallback"}.get(rc, "see include/vvr.h")))
        self.width, self.height, self.chroma_format = width, height, chroma_format
        self._keep = {}

    # -- lifetime
    def close(self):
        if self.ctx:
            self.L.vvr_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc < 0:
            raise VvrError("vvr error %d: %s" % (rc, self.L.vvr_last_error(self.ctx).decode()))
        return rc

    # -- DecLibRecon interface
    def decompress_picture(self, d: PictureDesc):
        p = d.c()
        job = self._check(self.L.vvr_submit(self.ctx, C.byref(p)))
        self._keep[job] = (d, p)
        return job

    def submit_c(self, p):
        """vvr_submit of a ctypes abi.Picture built beforehand (`desc.c()`); the caller keeps the description alive until wait()"""
        return self._check(self.L.vvr_submit(self.ctx, C.byref(p)))

    def test(self, job):
        """vvr_test: True when `job` is reconstructed (wait() returns at once), False when it is not yet; raises if it failed"""
        rc = self.L.vvr_test(self.ctx, job)
        if rc == abi.VVR_NOT_READY:
            return False
        self._check(rc)
        return True

    def wait(self, job):
        try:
            self._check(self.L.vvr_wait(self.ctx, job))
        finally:
            self._keep.pop(job, None)

    # -- external users of DPB slots, ordered on the device (the collective of vvdec_amd.parallel.PictureParallel)
    def stream_wait_job(self, job, stream_ptr, blocking=True):
        """the caller's stream waits for picture `job`; False: not handed to the device yet (blocking=False only)"""
        return self._check(self.L.vvr_stream_wait_job(self.ctx, job, stream_ptr, 1 if blocking else 0)) == abi.VVR_OK

    def stream_wait_slot(self, slot, stream_ptr, blocking=True):
        """the caller's stream waits for every picture submitted so far that reads or writes `slot`; False: some of them are still being prepared"""
        return self._check(self.L.vvr_stream_wait_slot(self.ctx, slot, stream_ptr, 1 if blocking else 0)) == abi.VVR_OK

    def slot_external_event(self, slot, event_ptr, writes):
        """pictures submitted from now on that use `slot` wait for the caller's event first"""
        self._check(self.L.vvr_slot_external_event(self.ctx, slot, event_ptr, 1 if writes else 0))

    def inputs_done(self, job):
        self._check(self.L.vvr_inputs_done(self.ctx, job))

    def host_array(self, n, dtype):
        """numpy array of n records in host memory the device reads directly (vvr_host_alloc): descriptions built in such arrays are
        uploaded without a staging copy.  The memory belongs to the context (freed by close())."""
        dt = np.dtype(dtype)
        nbytes = max(1, int(n)) * dt.itemsize
        ptr = self.L.vvr_host_alloc(self.ctx, nbytes)
        if not ptr:
            raise VvrError("vvr_host_alloc(%d) failed" % nbytes)
        return np.frombuffer((C.c_char * nbytes).from_address(ptr), dt, count=max(1, int(n)))[:int(n)]

    def copy_bandwidth(self, iters=20):
        """practical HBM ceiling: bytes/s (read + written) of the library's copy kernel over one DPB slot"""
        return self.L.vvr_measure_copy_bandwidth(self.ctx, iters)

    def sync(self):
        self._check(self.L.vvr_sync(self.ctx))
        self._keep.clear()

    def read_dmvr(self, job, n):
        """delta MVs (n x 2 int32, 1/16 sample) DMVR produced for job's picture, indexed cu.dmvr_off + sub-block"""
        a = np.zeros((max(1, n), 2), np.int32)
        self._check(self.L.vvr_read_dmvr(self.ctx, job, a.ctypes.data, n))
        return a[:n]

    def read_col_motion(self, job):
        """collocated motion of a picture submitted with TOOL_COL_MOTION: abi.Motion records, ((h4 + 1) // 2) * ((w4 + 1) // 2) of them in raster order"""
        n = self._check(self.L.vvr_read_col_motion(self.ctx, job, None, 0))
        a = np.zeros(n, np.dtype(abi.Motion))
        if n:
            self._check(self.L.vvr_read_col_motion(self.ctx, job, a.ctypes.data, n))
        return a

    # -- resident pictures (pre-parsed stream already in HBM)
    def prepare(self, d: PictureDesc):
        p = d.c()
        h = C.c_void_p()
        self._check(self.L.vvr_prepare(self.ctx, C.byref(p), C.byref(h)))
        return h

    def submit_prepared(self, handle):
        return self._check(self.L.vvr_submit_prepared(self.ctx, handle))

    def free_prepared(self, handle):
        self.L.vvr_free_prepared(self.ctx, handle)

    # -- planes
    def plane_shape(self, comp):
        return (self.height >> (1 if comp else 0), self.width >> (1 if comp else 0))

    def read_picture(self, slot):
        out = []
        for c in range(3 if self.chroma_format else 1):
            a = np.zeros(self.plane_shape(c), np.uint16)
            self._check(self.L.vvr_read_plane(self.ctx, slot, c, a.ctypes.data, a.shape[1]))
            out.append(a)
        return out

    def read_picture_into(self, slot, size, pad=0, threads=4):
        """vvr_read_picture: the finished picture in `slot` (luma size `size`) into arrays whose rows are `pad` samples longer than the picture's
        (a decoder's own buffers have margins); the caller has waited for the picture.  -> list of planes (views without the padding)"""
        ncomp = 3 if self.chroma_format else 1
        arrs = [np.full((size[1] >> (1 if c else 0), (size[0] >> (1 if c else 0)) + pad), 0xffff, np.uint16) for c in range(ncomp)]
        dst = (C.c_void_p * 3)(*[a.ctypes.data for a in arrs] + [None] * (3 - ncomp))
        strides = (C.c_size_t * 3)(*[a.shape[1] for a in arrs] + [0] * (3 - ncomp))
        self.L.vvr_read_picture.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
        self._check(self.L.vvr_read_picture(self.ctx, slot, dst, strides, threads))
        for c, a in enumerate(arrs):
            assert pad == 0 or (a[:, a.shape[1] - pad:] == 0xffff).all(), "vvr_read_picture wrote outside the picture"
        return [a[:, :a.shape[1] - pad] if pad else a for a in arrs]

    def picture_hash(self, slot, method=0):
        """decoded picture hash (0 MD5, 1 CRC, 2 checksum): list of per-component digests (bytes)"""
        buf = (C.c_uint8 * 48)()
        n = C.c_int()
        self._check(self.L.vvr_picture_hash(self.ctx, slot, method, buf, C.byref(n)))
        return [bytes(buf[k * n.value:(k + 1) * n.value]) for k in range(3 if self.chroma_format else 1)]

    def read_output(self, slot, window=None, bytes_per_sample=2):
        """the picture as the application gets it: conformance window (x, y, w, h in luma samples, even) applied, 8- or 16-bit samples"""
        x, y, w, h = window or (0, 0, self.width, self.height)
        out = []
        for c in range(3 if self.chroma_format else 1):
            s = 1 if c else 0
            a = np.zeros((h >> s, w >> s), np.uint8 if bytes_per_sample == 1 else np.uint16)
            self._check(self.L.vvr_read_output(self.ctx, slot, c, x >> s, y >> s, w >> s, h >> s, bytes_per_sample, a.ctypes.data, a.strides[0]))
            out.append(a)
        return out

    def write_picture(self, slot, planes):
        for c, pl in enumerate(planes):
            a = np.ascontiguousarray(pl, dtype=np.uint16)
            assert a.shape == self.plane_shape(c)
            self._check(self.L.vvr_write_plane(self.ctx, slot, c, a.ctypes.data, a.shape[1]))

    @staticmethod
    def new_dpb_tensor(width, height, num_slots, chroma_format=1, device="cuda"):
        """uint8 torch tensor that can hold the DPB of a context (pass its data_ptr() as ext_planes): slot s is the byte range
        [s * slot_bytes, (s + 1) * slot_bytes), which is what vvdec_amd.parallel.PictureParallel broadcasts between ranks"""
        import torch
        cfg = abi.Config()
        cfg.abi_version = abi.VVR_ABI_VERSION
        cfg.max_width, cfg.max_height, cfg.chroma_format = width, height, chroma_format
        nbytes = lib().vvr_slot_bytes(C.byref(cfg)) * num_slots
        return torch.zeros(nbytes, dtype=torch.uint8, device=device)

    def plane_ptr(self, slot, comp):
        return self.L.vvr_plane_ptr(self.ctx, slot, comp)

    def slot_bytes(self):
        return self.L.vvr_slot_bytes(C.byref(self.cfg))

    def plane_layout(self, comp):
        off, st, w, h = C.c_size_t(), C.c_size_t(), C.c_int(), C.c_int()
        self._check(self.L.vvr_plane_layout(self.ctx, comp, C.byref(off), C.byref(st), C.byref(w), C.byref(h)))
        return off.value, st.value, w.value, h.value

    # -- statistics (HIP events around every kernel launch, on the launch stream)
    def enable_stats(self, on=True):
        self._check(self.L.vvr_enable_stats(self.ctx, 1 if on else 0))

    def stats(self):
        arr = (abi.KernelStat * 24)()
        n = self._check(self.L.vvr_get_stats(self.ctx, arr, 24))
        return [dict(name=arr[i].name.decode(), launches=arr[i].launches, total_ms=arr[i].total_ms, algo_bytes=arr[i].algo_bytes) for i in range(n)]
