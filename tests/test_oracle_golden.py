"""CPU: the plain-C oracle (oracle/libvvoracle.so) against the committed golden fixtures.

The fixtures (tests/golden/*.npz, written by tests/golden/make_golden.py) hold what the REAL reference decoder classes
produced for each picture after every stage.  This pins the oracle to the reference wherever the tests run (the GPU box has
no /root/reference).  Bit-exact."""
import glob
import os
import numpy as np
import pytest

import golden_io
import refdrv

HERE = os.path.dirname(os.path.abspath(__file__))
FIXTURES = sorted(glob.glob(os.path.join(HERE, "golden", "*.npz")))
STAGE_FLAGS = {"reco": refdrv.STOP_AFTER_RECO, "dbk": refdrv.STOP_AFTER_DBK, "sao": refdrv.STOP_AFTER_SAO, "final": 0}


def test_fixtures_present():
    assert len(FIXTURES) >= 7


@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(p)[:-4] for p in FIXTURES])
def test_oracle_matches_reference_outputs(built, path):
    d, refs, outs = golden_io.load(path)
    assert set(outs) == set(STAGE_FLAGS)
    for st, fl in STAGE_FLAGS.items():
        got = refdrv.oracle_reconstruct(d, refs, flags=fl)
        for c in range(3):
            assert np.array_equal(got[c], outs[st][c]), "%s stage %s comp %d: %d samples differ from the reference" % (
                os.path.basename(path), st, c, int((got[c] != outs[st][c]).sum()))


def test_fixture_roundtrip(tmp_path):
    """save/load of a fixture is lossless (same bytes of every record array)"""
    d, refs, outs = golden_io.load(FIXTURES[0])
    p = str(tmp_path / "x.npz")
    golden_io.save(p, d, refs, outs)
    d2, refs2, outs2 = golden_io.load(p)
    assert bytes(d.hdr) == bytes(d2.hdr)
    same = lambda a, b: len(a) == len(b) and all(np.array_equal(a[n], b[n]) for n in a.dtype.names)
    assert same(d.cu, d2.cu) and same(d.tu, d2.tu) and np.array_equal(d.coef, d2.coef) and same(d.lfp[0], d2.lfp[0]) and same(d.motion, d2.motion)
    assert all(np.array_equal(a, b) for s in outs for a, b in zip(outs[s], outs2[s]))
