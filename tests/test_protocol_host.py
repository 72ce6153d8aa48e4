"""The Fortran protocol host (kiwi_amd/fortran/minimizer_hip): wire format on CPU, and on the GPU
the same answers as the Python host over the same library (and the oracle)."""
import os

import numpy as np
import pytest

from kiwi_amd import protocol, synthetic
from kiwi_amd import lib as klib

from tests.common import HAVE_FLANG, HAVE_HDF5
pytestmark = pytest.mark.skipif(not HAVE_FLANG, reason="amdflang not installed")


@pytest.fixture(scope="module")
def host():
    klib.build()
    if HAVE_HDF5:
        from kiwi_amd import gfdb_hdf5
        gfdb_hdf5.build()
    return protocol.build_host()


def test_wire_format_and_errors_without_device(host, tmp_path):
    p = protocol.MinimizerProcess(host)
    try:
        # unknown command -> "<cmd>: nok >" + message (minimizer.f90:1810)
        with pytest.raises(protocol.SeismosizerReturnedError, match="unknown command: frobnicate"):
            p.do("frobnicate")
        # comments and whitespace are ignored (minimizer.f90:1815-1846); commands with no device need answer plainly
        assert p.do("   set_verbose    T   # trailing comment") == ""
        assert p.do("set_ignore_sigint T") == ""
        with pytest.raises(protocol.SeismosizerReturnedError, match="unknown interpolation method"):
            p.do("set_local_interpolation", "cubic")
        with pytest.raises(protocol.SeismosizerReturnedError, match="can't open file"):
            p.do("set_receivers", str(tmp_path / "nofile"))
        with pytest.raises(protocol.SeismosizerReturnedError, match="unknown source type"):
            p.do("set_source_params", "banana", 1, 2, 3)
        with pytest.raises(protocol.SeismosizerReturnedError, match="expected 14 source parameters"):
            p.do("set_source_params", "bilateral", 1, 2, 3)
        with pytest.raises(protocol.SeismosizerReturnedError, match="unknown norm"):
            p.do("set_misfit_method", "l3norm")
        with pytest.raises(protocol.SeismosizerReturnedError, match="expected 15 source parameters"):
            p.do("set_source_params", "eikonal", 1, 2, 3)
        with pytest.raises(protocol.SeismosizerReturnedError, match="expected 20 source parameters"):
            p.do("set_source_params", "mt_eikonal", 1, 2, 3)
        with pytest.raises(protocol.SeismosizerReturnedError, match="not divideable by 6"):
            p.do("set_source_constraints", 0, 0, 1500, 0, 0)
        with pytest.raises(protocol.SeismosizerReturnedError, match="usage: set_source_crust "):
            p.do("set_source_crust", 1, 2, 3)
        with pytest.raises(protocol.SeismosizerReturnedError, match="usage: set_source_crustal_thickness_limit"):
            p.do("set_source_crustal_thickness_limit")
        # the process is still alive and in sync after all those errors
        assert p.do("set_verbose F") == ""
    finally:
        p.close()


@pytest.mark.gpu
def test_protocol_host_matches_python_host_and_oracle(host, tmp_path):
    from tests.common import Scenario, oracle_misfits
    sc = Scenario(nrec=4)
    e = sc.oracle()
    sc.make_references(e)
    sc.apply_setup(e, True)
    gf = dict(sc.gf)
    first, nsamp, data = sc.odb.dense_tables()
    gf.update(first=first, nsamp=nsamp, data=data)
    base = str(tmp_path / "db")
    protocol.write_flat_gfdb(base, gf)
    protocol.write_receivers(str(tmp_path / "receivers.table"), sc.lat, sc.lon, sc.comps)
    dt = gf["dt"]
    for (ir, k), (lo, d) in sc.refs.items():
        # sample index lo sits at time (lo-1)*dt (receiver.f90:649,842-849)
        protocol.write_table(str(tmp_path / ("ref-%d-%s.table" % (ir, sc.comps[ir - 1][k - 1]))), (lo - 1) * dt, dt, d)
    p = protocol.MinimizerProcess(host)
    try:
        p.do("set_database", base)
        p.do("set_effective_dt", sc.effective_dt)
        p.do("set_local_interpolation", "bilinear")
        p.do("set_receivers", str(tmp_path / "receivers.table"))
        p.do("set_source_location", 40.0, 30.0, 0.0)
        p.do("set_ref_seismograms", str(tmp_path / "ref"), "table")
        p.do("set_misfit_method", "l2norm")
        for ir, (x, y) in sc.tapers.items():
            p.do("set_misfit_taper", ir, *[v for xy in zip(x, y) for v in xy])
        trials = synthetic.bilat_strike_sweep(3, step=2.0)
        m, n, g = oracle_misfits(e, 1, trials)
        for i, t in enumerate(trials):
            p.do("set_source_params", "bilateral", *["%.9g" % v for v in t])
            ans = np.array(p.do("get_misfits").split(), np.float64)
            assert len(ans) == 2 * m.shape[1]
            # the reference traces went through a text file (9 significant digits): allow for that
            assert np.allclose(ans[0::2], m[i], rtol=2e-5, atol=1e-6 * n[i].max())
            assert np.allclose(ans[1::2], n[i], rtol=1e-5)
            gl = float(p.do("get_global_misfit"))
            assert abs(gl - g[i]) <= 2e-5 * g[i]
        # batch extension
        pf = tmp_path / "params.txt"
        with open(pf, "w") as f:
            for t in trials:
                f.write(" ".join("%.9g" % v for v in t) + "\n")
        assert p.do("eval_sources", "bilateral", str(pf), str(tmp_path / "out.txt")) == "3"
        out = np.loadtxt(tmp_path / "out.txt")
        assert out.shape == (3, 1 + 2 * m.shape[1])
        assert np.allclose(out[:, 0], g, rtol=2e-5)
        # masked parameters and minimize_lm (minimizer.f90:694-810,1048-1083,1199-1224) against the Python face of the
        # same C-ABI call, fed with the reference traces as they read back from the text files
        from kiwi_amd import lm
        start = trials[0].copy()
        start[5] += 3.0
        start[3] += 400.0
        p.do("set_source_params", "bilateral", *["%.9g" % v for v in start])
        with pytest.raises(protocol.SeismosizerReturnedError, match="wrong number of elements in mask"):
            p.do("set_source_params_mask", "T", "F")
        p.do("set_source_params_mask", *["T" if i in (3, 5, 6) else "F" for i in range(14)])
        assert np.array_equal(np.array(p.do("get_source_subparams").split(), np.float32), start[[3, 5, 6]])
        with pytest.raises(protocol.SeismosizerReturnedError, match="wrong number of subparams"):
            p.do("set_source_subparams", 1.0)
        p.do("set_source_subparams", "%.9g" % (start[3] + 100), "%.9g" % start[5], "%.9g" % start[6])
        start[3] += 100
        assert np.array_equal(np.array(p.do("get_source_subparams").split(), np.float32), start[[3, 5, 6]])
        g_start = float(p.do("get_global_misfit"))
        pp = sc.product()
        for (ir, k) in sc.refs:
            t, v = protocol.read_table(str(tmp_path / ("ref-%d-%s.table" % (ir, sc.comps[ir - 1][k - 1]))))
            pp.set_ref_seismogram(ir, k, int(round(t[0] / dt)) + 1, v.astype(np.float32))
        for ir, (x, y) in sc.tapers.items():
            pp.set_misfit_taper(ir, x, y)
        pp.set_misfit_method("l2norm")
        for lim in (None, ([8000.0, float(trials[0][5]) + 1.0, 50.0], [14000.0, 120.0, 95.0])):
            p.do("set_source_subparams", *["%.9g" % v for v in start[[3, 5, 6]]])
            if lim:
                with pytest.raises(protocol.SeismosizerReturnedError, match="wrong number of subparam_mins"):
                    p.do("set_source_subparams_limits", 1, 2)
                p.do("set_source_subparams_limits", *(lim[0] + lim[1]))
            info, iters, mis = p.do("minimize_lm").split()
            res = lm.minimize_lm(pp, "bilateral", start, [i in (3, 5, 6) for i in range(14)], *(lim or (None, None)))
            # (the traces reach the two engines through different text parsers; at the noise floor of a perfect fit the
            # runs may differ in the last digits, not in their course)
            assert int(info) in (1, 2, 3, 4) and res.info in (1, 2, 3, 4) and abs(int(iters) - res.iterations) <= 8
            assert abs(float(mis) - res.misfit) < 1e-4 * g_start and float(mis) < (0.9 if lim else 0.5) * g_start
            assert np.allclose(np.array(p.do("get_source_subparams").split(), np.float32), res.params[[3, 5, 6]], rtol=1e-4)
            assert np.float32(p.do("get_global_misfit")) == np.float32(mis)       # the engine holds the last forward step
            if lim:
                assert res.best[5] >= lim[0][1] - 1e-3
        pp.close()
        p.do("set_source_params_mask", *["T"] * 14)
        p.do("set_source_params", "bilateral", *["%.9g" % v for v in trials[0]])
        # inspection commands
        p.do("output_distances", str(tmp_path / "dist.txt"))
        d = np.loadtxt(tmp_path / "dist.txt")
        assert d.shape == (4, 3)
        for ir in range(4):
            az, _, dist = e.receiver_geometry(ir + 1)
            r2d = float(np.float32(360.) / np.float32(2.) / np.float32(3.14159265358979))     # orthodrome.f90:340-347
            assert abs(d[ir, 1] - dist) < 1e-3 and abs(d[ir, 2] - r2d * az) < 1e-9 * 360
        p.do("output_seismograms", str(tmp_path / "syn"), "table", "synthetics", "plain")
        t, v = protocol.read_table(str(tmp_path / "syn-1-n.table"))
        e.set_source_params(1, trials[0])
        e.get_misfits()
        lo_o, so = e.synthetic(1, 1, 1)
        i0 = int(round(t[0] / dt)) + 1
        a, b = max(lo_o, i0), min(lo_o + len(so), i0 + len(v))
        assert b - a > 200
        assert np.max(np.abs(so[a - lo_o:b - lo_o] - v[a - i0:b - i0])) <= 1e-5 * np.max(np.abs(so))
        with pytest.raises(protocol.SeismosizerReturnedError):
            p.do("switch_receiver", 99, "off")
        # eikonal source over the wire: crust profiles + constraints + 15 parameters
        from oracle import ko
        G = np.load(os.path.join(os.path.dirname(__file__), "golden", "eikonal_vectors.npz"))
        with pytest.raises(protocol.SeismosizerReturnedError, match="crust"):
            p.do("set_source_params", "eikonal", *G["e0_params"])
        p.do("set_source_crust", *G["rupture_profile"], *G["origin_profile"])
        thick = float(p.do("get_source_crustal_thickness"))
        assert np.float32(thick) == np.float32(ko.crust_thickness(ko.crust_profile(*np.split(G["origin_profile"], [8, 16, 24]))))
        p.do("set_source_crustal_thickness_limit", 10000)
        assert float(p.do("get_source_crustal_thickness")) == 10000.0
        cons = [0, 0, 6500, 0, 0, -1, 0, 0, 13500, 0, 0, 1]
        p.do("set_source_constraints", *cons)
        ep = np.array([0.1, 200, -100, 10000, 5e18, 85, 75, -160, 0, 0, 3000, 400, 100, 0.8, 1.2], np.float32)
        p.do("set_source_params", "eikonal", *["%.9g" % v for v in ep])
        gl = float(p.do("get_global_misfit"))
        c, mo, ri, _ = ko.discretize_eikonal(4, ep, sc.effective_dt, ko.crust_profile(*np.split(G["rupture_profile"], [8, 16, 24])),
                                             np.array(cons, np.float32).reshape(2, 6)[:, :3], np.array(cons, np.float32).reshape(2, 6)[:, 3:])
        assert len(c) > 20 and ri == np.float32(1.2)
        e.set_centroids(c, mo, ri)
        assert abs(gl - e.get_misfits()[2]) <= 2e-5 * gl
        # a sweep with trial sources the discretiser rejects: listed in the answer (seismosizer.py:703-720: failings),
        # their rows zeros, the others evaluated; a single rejected source is an error as in the reference
        bad = ep.copy()
        bad[3] = 500.0                                          # above the 6.5 km constraint: "Empty rupture area"
        ep2 = ep.copy()
        ep2[5] += 7.0
        pf = tmp_path / "esweep.txt"
        with open(pf, "w") as f:
            for t in (ep, bad, ep2, bad):
                f.write(" ".join("%.9g" % v for v in t) + "\n")
        assert p.eval_sources("eikonal", str(pf), str(tmp_path / "eout.txt")) == (4, [1, 3])
        eo = np.loadtxt(tmp_path / "eout.txt")
        assert eo.shape[0] == 4 and np.all(eo[[1, 3]] == 0) and abs(eo[0, 0] - gl) <= 1e-6 * gl and eo[2, 0] > 0 and eo[2, 0] != eo[0, 0]
        with pytest.raises(protocol.SeismosizerReturnedError, match="Empty rupture area"):
            p.do("set_source_params", "eikonal", *["%.9g" % v for v in bad])
        with open(pf, "w") as f:
            f.write(" ".join("%.9g" % v for v in bad) + "\n" + " ".join("%.9g" % v for v in bad) + "\n")
        assert p.eval_sources("eikonal", str(pf), str(tmp_path / "eout.txt")) == (2, [0, 1])
        p.do("set_source_params", "eikonal", *["%.9g" % v for v in ep])
        assert float(p.do("get_global_misfit")) == gl
        # floating norms over the wire (minimizer.f90:388-445)
        p.do("set_source_params", "bilateral", *["%.9g" % v for v in trials[1]])
        plain = float(p.do("get_global_misfit"))
        p.do("set_misfit_method", "floating_l2norm")
        p.do("set_floating_shiftrange", 0, 0, 0)
        assert float(p.do("get_global_misfit")) == plain
        assert np.array_equal(np.array(p.do("get_floating_shifts").split(), float), np.zeros(4))
        p.do("set_floating_shiftrange", 0, -1.0, 1.0)
        p.do("set_floating_shiftrange", 2, -2.0, 0.5)
        sh = np.array(p.do("get_floating_shifts").split(), float)
        assert sh.shape == (4,) and np.all(np.abs(sh) <= 2.0) and np.all(np.abs(sh / dt - np.rint(sh / dt)) < 1e-6)
        assert float(p.do("get_global_misfit")) <= plain * (1 + 1e-6)
        with pytest.raises(protocol.SeismosizerReturnedError, match="usage: set_floating_shiftrange"):
            p.do("set_floating_shiftrange", 1)
        # set_misfit_filter applies to every receiver, set_misfit_filter_1 to one (minimizer.f90:875-968)
        p.do("set_misfit_method", "l2norm")
        p.do("set_floating_shiftrange", 0, 0, 0)
        p.do("set_source_params", "bilateral", *["%.9g" % v for v in trials[1]])
        unfiltered = float(p.do("get_global_misfit"))
        p.do("set_misfit_filter", 0.01, 0, 0.02, 1, 0.1, 1, 0.2, 0)
        filtered_all = float(p.do("get_global_misfit"))
        assert filtered_all != unfiltered
        p.do("set_misfit_filter")                               # no points: filter removed everywhere
        assert float(p.do("get_global_misfit")) == unfiltered
        p.do("set_misfit_filter_1", 2, 0.01, 0, 0.02, 1, 0.1, 1, 0.2, 0)
        one = float(p.do("get_global_misfit"))
        assert one != unfiltered and one != filtered_all
        with pytest.raises(protocol.SeismosizerReturnedError, match="receiver index out of range"):
            p.do("set_misfit_taper", 0, 0, 0, 1, 1)
        # the reference probes can be written too (receiver.f90:618-680)
        p.do("output_seismograms", str(tmp_path / "refout"), "table", "references", "plain")
        t, v = protocol.read_table(str(tmp_path / "refout-1-n.table"))
        lo, d = sc.refs[(1, 1)]
        assert len(v) == len(d) and np.allclose(v, d, rtol=2e-7, atol=0)       # eight significant digits in the table
        p.do("output_seismograms", str(tmp_path / "reftap"), "table", "references", "tapered")
        t2, v2 = protocol.read_table(str(tmp_path / "reftap-1-n.table"))
        assert abs(v2[0]) == 0.0 and np.max(np.abs(v2)) <= np.max(np.abs(d)) * (1 + 1e-6)
        with pytest.raises(protocol.SeismosizerReturnedError, match="unknown probe"):
            p.do("output_seismograms", str(tmp_path / "x"), "table", "nonsense", "plain")
        # shake-map diagnostics over the wire (minimizer.f90:1305-1372)
        p.do("set_misfit_filter")
        e.set_source_params(1, trials[1])
        for cmd, want in (("get_peak_amplitudes 1", e.peak_amplitudes(1)), ("get_peak_amplitudes 2", e.peak_amplitudes(2)),
                          ("get_arias_intensities", e.arias_intensities())):
            got = np.array(p.do(*cmd.split()).split(), np.float32)
            assert got.shape == want.shape and np.allclose(got, want, rtol=2e-4)
        with pytest.raises(protocol.SeismosizerReturnedError, match="differentiate argument must be 1"):
            p.do("get_peak_amplitudes", 3)
        from kiwi_amd import engine as kengine
        axes = np.array(p.do("get_principal_axes").split(), np.float32)
        assert np.array_equal(axes, np.concatenate(kengine.principal_axes("bilateral", trials[1])))
        p.do("output_seismogram_spectra", str(tmp_path / "spec"), "synthetics", "plain")
        f, v = protocol.read_table(str(tmp_path / "spec-2-d.table"))
        odf, want_spec = e.amp_spectrum(2, sc.comps[1].index("d") + 1, True, False)
        assert len(v) == len(want_spec) and abs(f[1] - odf) < 1e-6 * odf and np.allclose(v, want_spec, rtol=1e-4, atol=1e-5 * want_spec.max())
        p.do("output_cross_correlations", str(tmp_path / "cc"), -2 * dt, 2 * dt)
        t, v = protocol.read_table(str(tmp_path / "cc-3-e.table"))
        e.get_misfits()
        want_cc = e.cross_correlations(3, -2, 2)[sc.comps[2].index("e")]
        assert np.allclose(t, np.arange(-2, 3) * dt) and np.allclose(v, want_cc, rtol=1e-4, atol=1e-5 * np.max(np.abs(want_cc)))
        # output_source_model: the centroid table the engine holds (minimizer_engine.f90:947-977)
        p.do("output_source_model", str(tmp_path / "sm"))
        tab = np.loadtxt(str(tmp_path / "sm-dsm.table"), dtype=np.float32, ndmin=2)
        ocent = ko.discretize(1, np.asarray(trials[1], np.float32), sc.effective_dt)[0]
        assert tab.shape == ocent.shape and np.array_equal(tab, ocent)
        info = open(str(tmp_path / "sm-tdsm.info")).read().split()
        assert info[0] == "ncentroids" and int(info[1]) == len(ocent)
        assert int(p.do("get_cached_traces_memory")) > 0
        p.do("set_cached_traces_memory_limit", 1000000)
    finally:
        p.close()


@pytest.mark.gpu
def test_config1_mini_inp_command_sequence(host, tmp_path):
    """BASELINE.json configs[0]: the command sequence of the reference's benchmark/mini.inp (first block) through the
    protocol host, with the Izmit receiver table (tests/golden/izmit-receivers.table, the reference's own data file)
    and a synthetic database in place of the 20000-km GEMINI one; `mseed` output replaced by `table`."""
    from oracle import ko
    gf = synthetic.make_gfdb(nx=64, nz=7, ng=10, L=512, dx=25e3, dz=2000.0, firstx=300e3, firstz=4e3)
    base = str(tmp_path / "db")
    if HAVE_HDF5:              # the reference's own database format (db.index + db.N.chunk)
        from kiwi_amd import gfdb_hdf5
        gfdb_hdf5.write(base, gf, nchunks=4)
        assert not os.path.exists(base + ".kiwiflat")
    else:                      # (tests/test_product_cpu.py fails on an image where this branch is taken unexpectedly)
        protocol.write_flat_gfdb(base, gf)
    table = os.path.join(os.path.dirname(__file__), "golden", "izmit-receivers.table")
    rec = [l.split() for l in open(table) if l.strip()]
    lat, lon, comps = [float(r[0]) for r in rec], [float(r[1]) for r in rec], [r[2] for r in rec]
    assert len(rec) == 11 and set(comps) == {"ned"}
    src = ["0 0 0 10000 2e20  91 87 164  0  20000 10000 9000  3500 2", "0 0 0 10000 2e20  92 87 164  0  20000 10000 9000  3500 2"]
    p = protocol.MinimizerProcess(host)
    out = {}
    try:
        with pytest.raises(protocol.SeismosizerReturnedError, match="no database set"):
            p.do("get_database_format")
        p.do("set_database           ", base)
        assert p.do("get_database_format") == ("hdf5" if HAVE_HDF5 else "kiwiflat")      # the reader that actually ran
        p.do("set_effective_dt        0.5")
        p.do("set_local_interpolation bilinear")
        p.do("set_receivers          ", table)
        p.do("set_source_location     40.75 29.86 0")
        for rep in range(2):                                   # mini.inp alternates the two sources eight times
            for k, sp in enumerate(src):
                p.do("set_source_params       bilateral " + sp)
                stem = str(tmp_path / ("izmit-seismogram%d%d" % (rep, k)))
                p.do("output_seismograms     ", stem, " table synthetics plain")
                out[(rep, k)] = [protocol.read_table("%s-%d-%s.table" % (stem, ir + 1, c)) for ir in range(11) for c in "ned"]
    finally:
        p.close()
    # repeated evaluation of the same source gives the same file, the two sources differ
    for a, b in zip(out[(0, 0)], out[(1, 0)]):
        assert np.array_equal(a[1], b[1])
    assert any(not np.array_equal(a[1], b[1]) for a, b in zip(out[(0, 0)], out[(0, 1)]))
    # against the oracle
    nx, nz, ng, L = gf["data"].shape
    db = ko.Gfdb(nx, nz, ng, gf["dt"], gf["dx"], gf["dz"], gf["firstx"], gf["firstz"])
    for ix in range(nx):
        for iz in range(nz):
            for ig in range(ng):
                db.set_trace(ix + 1, iz + 1, ig + 1, int(gf["first"][ix, iz, ig]), gf["data"][ix, iz, ig])
    e = ko.Engine(db)
    e.set_receivers(np.array(lat), np.array(lon), np.zeros(11, np.float32), comps)
    e.set_source_location(40.75, 29.86, 0.0)
    e.set_effective_dt(0.5)
    e.set_interpolation(True)
    dt = gf["dt"]
    for k, sp in enumerate(src):
        params = np.array(sp.split(), np.float32)
        assert len(ko.discretize(1, params, 0.5)[0]) > 500          # an extended rupture: ~1000 centroids
        e.set_source_params(1, params)
        e.calculate_seismograms()
        e.scale_seismograms()
        for ir in range(11):
            for ic in range(3):
                t, v = out[(0, k)][3 * ir + ic]
                lo_o, so = e.synthetic(ir + 1, ic + 1, 1)
                i0 = int(round(t[0] / dt)) + 1
                a, b = max(lo_o, i0), min(lo_o + len(so), i0 + len(v))
                assert b - a > 300
                assert np.max(np.abs(so[a - lo_o:b - lo_o] - v[a - i0:b - i0])) <= 1e-5 * np.max(np.abs(so))
    e.close()
    db.close()
