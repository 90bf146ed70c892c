"""GPU helper of tests/test_gpu_parity.py::test_reference_slots_replicated_between_two_back_ends_on_the_device (run as a script: argv[1] = library threads).

Two back-ends with a DPB each take the pictures of one stream in turn; every reconstructed picture is copied slot to slot into the other DPB on the
collective's stream (vvdec_amd.parallel.TorchDeviceRuntime: a torch stream, torch events) - ordered behind the job that writes it
(vvr_stream_wait_job) and behind the other back-end's pictures that still use what the slot held (vvr_stream_wait_slot), and ahead of the pictures
that read it afterwards (vvr_slot_external_event) - without a host wait for any picture.  Both DPBs must end up holding what one back-end alone
produces.  This is the device-side half of vvdec_amd.parallel.PictureParallel with the RCCL transfer replaced by a device copy."""
import os
import sys
import numpy as np
import torch                      # first: its HIP runtime is the one the process initialises (as in bench.py)

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vvdec_amd                  # noqa: E402
from vvdec_amd import abi, synth, stream          # noqa: E402
from vvdec_amd.parallel import TorchDeviceRuntime  # noqa: E402


def main(threads):
    W, H = 832, 480
    plans, nslots = stream.ra_plan(17, gop=8, seed_poc0_is_external=False)
    T = (abi.TOOL_SAO_LUMA | abi.TOOL_SAO_CHROMA | abi.TOOL_ALF | abi.TOOL_CCALF | abi.TOOL_DEP_QUANT | abi.TOOL_MTS | abi.TOOL_BDOF | abi.TOOL_DMVR | abi.TOOL_PROF |
         abi.TOOL_LMCS | abi.TOOL_LMCS_CSCALE)
    dpb = [vvdec_amd.Reconstructor.new_dpb_tensor(W, H, nslots), vvdec_amd.Reconstructor.new_dpb_tensor(W, H, nslots)]
    rec = [vvdec_amd.Reconstructor(W, H, num_slots=nslots, num_streams=2, host_threads=threads, ext_planes=d.data_ptr()) for d in dpb]
    sb = rec[0].slot_bytes()
    rt = TorchDeviceRuntime()
    descs = [synth.picture_for_plan(pl, W, H, seed=741, tool_flags=T, p_intra=0.2, p_affine=0.1) for pl in plans]
    for i, (pl, d) in enumerate(zip(plans, descs)):
        own, other = i & 1, 1 - (i & 1)
        job = rec[own].decompress_picture(d)
        assert rec[own].stream_wait_job(job, rt.stream_ptr(), True)              # (blocks until the picture is handed to the device, not until it is done)
        assert rec[other].stream_wait_slot(pl.slot, rt.stream_ptr(), True)
        with torch.cuda.stream(rt.stream):
            dpb[other][pl.slot * sb:(pl.slot + 1) * sb].copy_(dpb[own][pl.slot * sb:(pl.slot + 1) * sb], non_blocking=True)
        rec[own].slot_external_event(pl.slot, rt.event_ptr(), writes=False)      # the copy reads the owner's slot ...
        rec[other].slot_external_event(pl.slot, rt.event_ptr(), writes=True)     # ... and writes the other one's
    rt.finish()
    for r in rec:
        r.sync()
    one = vvdec_amd.Reconstructor(W, H, num_slots=nslots, num_streams=1)         # the same stream through one back-end, one picture at a time
    for d in descs:
        one.wait(one.decompress_picture(d))
    for slot in {pl.slot for pl in plans}:
        want = one.read_picture(slot)
        for k in (0, 1):
            got = rec[k].read_picture(slot)
            for c in range(3):
                assert np.array_equal(got[c], want[c]), "back-end %d, slot %d, component %d: %d samples differ" % (k, slot, c, int((got[c] != want[c]).sum()))
    assert torch.equal(dpb[0], dpb[1])
    for r in rec + [one]:
        r.close()
    print("both DPBs equal the single back-end (%d pictures, %d library threads)" % (len(plans), threads))


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
