"""Reference wrap-around with motion vectors several wrap periods out - what a parsed stream with AMVR carries and the generator's default window
never produced.  Two rules of the reference only show with such vectors (finding 9):

  * the DMVR bilinear stage (xinitMC) runs once per CU: the start MVs are clipped against the CU, not the 16x16 sub-block;
  * an SbTMVP CU is predicted in the pieces xSubPuMC joins, and wrapClipMv depends on the piece's position and width.

The CPU side (oracle == reference classes, host glue, drop-in on the CPU oracle == reference decoder) is in test_oracle_vs_ref.py, test_host_glue.py and
test_dropin_library.py.  This file is the GPU side (green on the device since the end of round 4: GPUTEST_r04.json)."""
import glob
import os
import sys

import pytest

from vvdec_amd import abi

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("W,H,off,kw", [
    (384, 256, 368, dict(log2_ctu=6, p_bi=0.9, mv_sigma=1500.0, mv_window=2000)),                                                    # plain / BDOF / DMVR
    (384, 256, 368, dict(log2_ctu=6, p_bi=0.8, p_affine=0.2, p_geo=0.2, p_ciip=0.1, mv_sigma=3000.0, mv_window=4000)),              # + affine, GPM, CIIP
    (384, 256, 368, dict(log2_ctu=6, p_sbtmvp=0.5, p_bi=0.5, mv_sigma=1500.0, mv_window=2000)),                                      # SbTMVP pieces
    (512, 256, 512, dict(log2_ctu=7, p_sbtmvp=0.3, p_affine=0.2, p_bi=0.7, mv_sigma=2500.0, mv_window=4000)),
])
def test_vectors_beyond_a_wrap_period(built, W, H, off, kw):
    from test_gpu_parity import _run_stream, TOOLS_A
    _run_stream(W, H, 5, 4, 331, TOOLS_A | abi.TOOL_LMCS, intra=True, p_intra=0.05, wrap_offset=off, **kw)


def test_parsed_streams_with_vectors_beyond_a_wrap_period():
    """tests/bitstreams/wraparound_*: the four streams of the random sweep (tools/fuzz_dropin_on_the_oracle.py) that showed the two rules - decoded by the
    reference's application on the drop-in library with the GPU back-end: output MD5 == the reference decoder's, every decoded picture hash checks"""
    sys.path.insert(0, os.path.join(HERE, "..", "tools"))
    import dropin_decode as dd
    if not os.path.exists(dd.APP_DROPIN):
        pytest.skip("oracle/_ref/vvdecapp_dropin not built")
    streams = sorted(glob.glob(os.path.join(HERE, "bitstreams", "wraparound_ctu64_384x256_seed*", "*.bit")))
    assert len(streams) == 4
    bad = []
    for b in streams:
        r = dd.decode_stream(b, threads=4, with_reference=False)
        if not r["ok"]:
            bad.append((r["stream"], r["dropin"].get("tail") or r["dropin_dph"].get("tail")))
    assert not bad, bad
