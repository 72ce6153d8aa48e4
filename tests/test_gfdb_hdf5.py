"""Kiwi's HDF5 Green's function database (gfdb_io_hdf.f90 layout) through kiwi_amd/gfdb_hdf5.py: the strips the writer
makes are checked against the oracle's trace_pack (itself pinned to the reference's), and a database written in the
reference's layout reads back into exactly the dense tables the oracle's packed in-memory database holds."""
import os

import numpy as np
import pytest

from kiwi_amd import synthetic
from tests.common import Scenario

gfdb_hdf5 = pytest.importorskip("kiwi_amd.gfdb_hdf5")
from tests.common import HAVE_HDF5
pytestmark = pytest.mark.skipif(not HAVE_HDF5, reason="HDF5 C library not installed (tests/test_product_cpu.py reports that as a failure on the build image)")


def test_pack_trace_matches_oracle_trace_pack():
    from oracle import ko
    rng = np.random.default_rng(5)
    for _ in range(200):
        n = int(rng.integers(1, 120))
        d = rng.standard_normal(n).astype(np.float32)
        for _ in range(int(rng.integers(0, 4))):
            a = int(rng.integers(0, n))
            d[a:a + int(rng.integers(1, 14))] = 0
        if rng.random() < 0.3:
            d[-int(rng.integers(1, 9)):] = 0
        if rng.random() < 0.1:
            d[:] = 0
        lo = int(rng.integers(-50, 50))
        mine = gfdb_hdf5.pack_trace(lo, d)
        theirs, tspan = ko.trace_pack_spans(lo, d)        # [(lo, hi), ...] of the oracle's strips, trace span
        assert [(o, o + len(s) - 1) for o, s in mine] == [tuple(x) for x in theirs]
        assert tspan == (mine[0][0], mine[-1][0] + len(mine[-1][1]) - 1)
        for o, s in mine:
            assert np.array_equal(s, d[o - lo:o - lo + len(s)])


@pytest.mark.parametrize("nchunks", [1, 3])
def test_write_read_roundtrip_equals_packed_database(tmp_path, nchunks):
    sc = Scenario(nx=7, nz=3, ng=10, L=200, variant="static")
    gf = dict(sc.gf)
    gf["nsamp"] = gf["nsamp"].copy()
    gf["nsamp"][2, 1, 4] = 0                               # one trace missing
    base = str(tmp_path / "db")
    gfdb_hdf5.write(base, gf, nchunks=nchunks)
    assert sorted(os.listdir(tmp_path)) == ["db.%d.chunk" % (i + 1) for i in range(nchunks)] + ["db.index"]
    ix = gfdb_hdf5.read_index(base)
    assert (ix.nx, ix.nz, ix.ng, ix.nchunks) == (7, 3, 10, nchunks) and ix.dt == np.float32(gf["dt"])
    back = gfdb_hdf5.read(base)
    # the oracle's in-memory database packs the same dense traces with trace_pack: same spans, same samples
    e = sc.oracle()
    first, nsamp, data = sc.odb.dense_tables()
    keep = np.ones(first.shape, bool)
    keep[2, 1, 4] = False
    assert np.array_equal(back["first"][keep], first[keep]) and np.array_equal(back["nsamp"][keep], nsamp[keep])
    assert back["nsamp"][2, 1, 4] == 0
    L = back["data"].shape[-1]
    assert L == nsamp[keep].max()
    assert np.array_equal(back["data"][keep], data[keep][:, :L])
    for k in ("dt", "dx", "dz", "firstx", "firstz"):
        assert back[k] == np.float32(gf[k])
    e.close()


def test_errors():
    with pytest.raises(gfdb_hdf5.GfdbError, match="failed to open file"):
        gfdb_hdf5.read("/nonexistent/db")


def test_corrupt_strip_table_is_an_error_not_an_overflow(tmp_path, monkeypatch):
    """The strip offsets (attributes pofs / ofs) come from the file: unsorted or overlapping strips must end in the
    reader's 'failed to read a trace' error, not in a copy outside the dense row."""
    gf = synthetic.make_gfdb(nx=3, nz=2, ng=10, L=64)
    good = gfdb_hdf5.pack_trace

    def overlapping(lo, d):                      # second strip BEFORE the first one
        return [(lo + 40, np.asarray(d[:10], np.float32)), (lo - 500, np.asarray(d[10:30], np.float32))]

    monkeypatch.setattr(gfdb_hdf5, "pack_trace", overlapping)
    gfdb_hdf5.write(str(tmp_path / "bad"), gf)
    monkeypatch.setattr(gfdb_hdf5, "pack_trace", good)
    with pytest.raises(gfdb_hdf5.GfdbError, match="failed to read a trace"):
        gfdb_hdf5.read(str(tmp_path / "bad"))
