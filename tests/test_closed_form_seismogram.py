"""make_seismogram against values written out by hand: the oracle's COMPOSITION -- weights, component
order, sign map, rotation to north / east (oracle/ko_engine.c restating seismogram.f90:131-289) -- is otherwise pinned only
through its primitives.  Setup that has a closed form:

* Green's functions = one impulse per trace, amplitude (ig + 1) x (1 + ix / 4 + iz / 16) -- the node a trace comes from shows
  in the amplitude -- at strip index first(ix) + 10;
* ONE centroid on a depth node, nearest-neighbour interpolation (or bilinear with the four nodes made equal), time = an
  integer number of samples;
* receivers due NORTH of the source (azimuth 0: f1 = mxx, f2 = mxz, f3 = mzz, f4 = mxy, f5 = myz, f6 = myy) and due EAST of it
  on the equator (azimuth pi / 2: f1 = myy, f2 = myz, f3 = mzz, f4 = -mxy, f5 = -mxz, f6 = mxx), each with every component
  letter: a/c = +-radial, r/l = +-transverse, d/u = +-vertical, n/s, e/w = rotation of (away, right) by back-azimuth + pi
  (north receiver: n = away, e = right; east receiver: n = -right, e = away);
* radial = f1 G1 + f2 G2 + f3 G3 [+ f6 G9], transverse = f4 G4 + f5 G5, vertical = f1 G6 + f2 G7 + f3 G8 [+ f6 G10], times
  the moment; ng = 8 and 10.
The rotating branch (seismogram.f90:160-203) is entered by a centroid 300 m east of the origin: its own azimuth
atan2(-300, distance) enters the weights, the same angle rotates (radial, transverse) into the receiver's frame; written
out in plane geometry (the reference works on the sphere: 2e-4 there, 2e-6 in the plain cases).
The device goes through the same cases (GPU test)."""
import numpy as np
import pytest

from oracle import ko

DT, DX, DZ, FIRSTX, FIRSTZ = 0.5, 4000.0, 2000.0, 100e3, 6e3
NX, NZ, L = 6, 3, 21
M6 = np.array([1.0, 2.0, -3.0, 0.5, 0.25, -0.75], np.float32)          # mxx myy mzz mxy mxz myz
MOMENT, KSHIFT, DEPTH = 2.0, 6, 8000.0                                  # time = 6 samples; depth = node iz = 1
COMPS = ["ard", "clu", "ne", "sw"]
R_EARTH = 6371000.0


def amplitude(ix, iz, ig0, equal_nodes):
    return (ig0 + 1) * (1.5625 if equal_nodes else (1.0 + 0.25 * ix + 0.0625 * iz))


def tables(ng, equal_nodes):
    data = np.zeros((NX, NZ, ng, L), np.float32)
    first = np.zeros((NX, NZ, ng), np.int32)
    for ix in range(NX):
        for iz in range(NZ):
            for ig in range(ng):
                data[ix, iz, ig, 10] = amplitude(ix, iz, ig, equal_nodes)
                first[ix, iz, ig] = 200 + (0 if equal_nodes else ix)
    return data, first, np.full((NX, NZ, ng), L, np.int32)


def receivers():
    """four receivers due north of (0 N, 30 E), four due east on the equator, about 108 km away (node ix = 2)"""
    ang = np.degrees(108e3 / R_EARTH)
    lat = [ang] * 4 + [0.0] * 4
    lon = [30.0] * 4 + [30.0 + ang] * 4
    return np.array(lat), np.array(lon), np.zeros(8, np.float32), COMPS + COMPS


def weights(a):
    mxx, myy, mzz, mxy, mxz, myz = [float(v) for v in M6]
    c, s, s2, c2 = np.cos(a), np.sin(a), np.sin(2 * a), np.cos(2 * a)
    return (mxx * c * c + myy * s * s + mxy * s2, mxz * c + myz * s, mzz, 0.5 * (myy - mxx) * s2 + mxy * c2,
            myz * c - mxz * s, mxx * s * s + myy * c * c - mxy * s2)


def expected(ng, east_receiver, rotated, equal_nodes):
    """{letter: amplitude} of the impulse every component consists of"""
    g = [amplitude(2, 1, ig, equal_nodes) for ig in range(ng)]
    if not rotated:
        f = (M6[0], M6[4], M6[2], M6[3], M6[5], M6[1]) if not east_receiver else (M6[1], M6[5], M6[2], -M6[3], -M6[4], M6[0])
        f = [float(v) for v in f]
        lam = 0.0
    else:
        # centroid 300 m east of the origin: direction to a receiver 108 km north / east of the ORIGIN, plane geometry
        d = 108e3
        a = np.arctan2(-300.0, d) if not east_receiver else np.arctan2(d - 300.0, 0.0)
        f = weights(a)
        lam = a - (0.0 if not east_receiver else np.pi / 2)              # change of the ray's direction = rotation into the receiver's frame
    rad = f[0] * g[0] + f[1] * g[1] + f[2] * g[2] + (f[5] * g[8] if ng == 10 else 0.0)
    tra = f[3] * g[3] + f[4] * g[4]
    ver = f[0] * g[5] + f[1] * g[6] + f[2] * g[7] + (f[5] * g[9] if ng == 10 else 0.0)
    away = rad * np.cos(lam) - tra * np.sin(lam)
    right = tra * np.cos(lam) + rad * np.sin(lam)
    north, east = (away, right) if not east_receiver else (-right, away)
    v = {"a": away, "c": -away, "r": right, "l": -right, "d": ver, "u": -ver, "n": north, "s": -north, "e": east, "w": -east}
    return {k: MOMENT * x for k, x in v.items()}


def centroid(rotated):
    return np.array([[0.0, 300.0 if rotated else 0.0, DEPTH, KSHIFT * DT] + list(M6)], np.float32)


def check(get, ng, rotated, equal_nodes):
    """get(irec1, icomp1) -> (first strip index, samples)"""
    tol = 2e-4 if rotated else 2e-6
    lat, lon, depth, comps = receivers()
    t_imp = 200 + (0 if equal_nodes else 2) + 10 + KSHIFT                 # node ix = 2 (108 km), shifted by six samples
    for ir in range(8):
        want = expected(ng, ir >= 4, rotated, equal_nodes)
        scale = max(abs(v) for v in want.values())
        for k, letter in enumerate(comps[ir]):
            lo, d = get(ir + 1, k + 1)
            full = np.zeros(400, np.float64)
            full[lo:lo + len(d)] = d
            assert abs(full[t_imp] - want[letter]) <= tol * scale, (ng, rotated, ir, letter, full[t_imp], want[letter])
            full[t_imp] = 0.0
            assert np.max(np.abs(full)) <= tol * scale, (ng, rotated, ir, letter, "energy off the impulse")


CASES = [(ng, rot, eq, bil) for ng in (8, 10) for rot in (False, True) for (eq, bil) in ((False, False), (True, True))]


@pytest.mark.parametrize("ng,rotated,equal_nodes,bilinear", CASES)
def test_oracle_make_seismogram_closed_form(ng, rotated, equal_nodes, bilinear):
    data, first, nsamp = tables(ng, equal_nodes)
    db = ko.Gfdb(NX, NZ, ng, DT, DX, DZ, FIRSTX, FIRSTZ)
    for ix in range(NX):
        for iz in range(NZ):
            for ig in range(ng):
                db.set_trace(ix + 1, iz + 1, ig + 1, int(first[ix, iz, ig]), data[ix, iz, ig])
    e = ko.Engine(db)
    lat, lon, depth, comps = receivers()
    e.set_receivers(lat, lon, depth, comps)
    e.set_source_location(0.0, 30.0, 0.0)
    e.set_effective_dt(DT)
    e.set_interpolation(bilinear)
    e.set_centroids(centroid(rotated), MOMENT, 0.0)
    e.calculate_seismograms()
    e.scale_seismograms()
    check(lambda ir, k: e.synthetic(ir, k, 1), ng, rotated, equal_nodes)
    e.close()
    db.close()


@pytest.mark.gpu
@pytest.mark.parametrize("ng,rotated,equal_nodes,bilinear", CASES)
def test_device_make_seismogram_closed_form(ng, rotated, equal_nodes, bilinear):
    from kiwi_amd import Engine
    data, first, nsamp = tables(ng, equal_nodes)
    p = Engine(0)
    p.set_database(DT, DX, DZ, FIRSTX, FIRSTZ, data, first, nsamp)
    lat, lon, depth, comps = receivers()
    p.set_receivers(lat, lon, depth, comps)
    p.set_source_location(0.0, 30.0, 0.0)
    p.set_effective_dt(DT)
    p.set_local_interpolation("bilinear" if bilinear else "nearest")
    p.set_sources([centroid(rotated)], [MOMENT], [0.0])
    p.set_keep_synthetics(1)
    p.eval()
    check(lambda ir, k: p.get_synthetics(0, ir, k, 1), ng, rotated, equal_nodes)
    p.close()
