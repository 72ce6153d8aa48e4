"""Deterministic synthetic inputs for tests and bench.py: a closed-form
Green's function database, a ring of receivers, reference traces' tapers and trial-source grids.
No file of the reference is needed; everything is numpy."""
import os

import numpy as np

EARTH_R = 6371000.0


def _slabs(fn, starts):
    """fn(start) over the slabs of a database, on a few threads when there are many (numpy releases the interpreter lock in
    the array operations; a 4 GB database takes a minute and a half on one thread)"""
    starts = list(starts)
    nthr = min(len(starts), 12, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1))
    if nthr <= 1:
        for a in starts:
            fn(a)
        return
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(nthr) as ex:
        list(ex.map(fn, starts))


def make_gfdb(nx=128, nz=6, ng=10, L=4096, dt=0.5, dx=4000.0, dz=2000.0, firstx=100e3, firstz=6e3,
              variant="probe", center=600.0, width=400.0, vel=6000.0):
    """Returns dict(dt,dx,dz,firstx,firstz, data[nx,nz,ng,L] f32, first[nx,nz,ng] i32, nsamp[...] i32).

    variant "probe": the survey's probe database -- damped sinusoids, last sample forced to 0.
    variant "static": adds a non-zero static end value to components 2 and 7 and carves interior
    zero gaps of 9 samples (> maxgap = 5, sparse_trace.f90:25) so that gap-compressed traces with
    several strips and repeated non-zero end points are exercised.
    (Evaluated in slabs of distances: the float64 temporaries of a multi-GB database would not fit the host.)"""
    iz = np.arange(nz)[None, :, None, None]
    ig = np.arange(1, ng + 1)[None, None, :, None]
    i = np.arange(L)[None, None, None, :].astype(np.float64)
    # everything that does not depend on the distance, once
    shape = (1e-20 * np.sin(0.02 * i * (1 + 0.05 * ig) + 0.37 * ig + 0.11 * (iz + 1))
             * np.exp(-((i - center - center / 15.0 * ig) / width) ** 2))
    stat_ramp = None
    if variant == "static":
        ramp = 0.5 * (1 + np.tanh((i - 500.0) / 60.0))
        stat = np.zeros((1, 1, ng, 1))
        stat[0, 0, 1, 0] = 3e-22
        stat[0, 0, 6, 0] = -2e-22
        stat_ramp = stat * ramp
    data = np.empty((nx, nz, ng, L), np.float32)
    slab = max(1, (1 << 24) // (nz * ng * L))

    def fill(a):
        b = min(nx, a + slab)
        x = (firstx + np.arange(a, b) * dx)[:, None, None, None]
        val = shape / (x / 1e5)
        if variant == "static":
            val = val + stat_ramp / (x / 1e5)
            val[..., 300:309] = 0.0
            val[..., 1000:1009] = 0.0
        data[a:b] = val.astype(np.float32)

    _slabs(fill, range(0, nx, slab))
    if variant != "static":
        data[..., -1] = 0.0
    first = np.rint((firstx + np.arange(nx) * dx) / vel / dt).astype(np.int32)
    first = np.broadcast_to(first[:, None, None], (nx, nz, ng)).copy()
    nsamp = np.full((nx, nz, ng), L, np.int32)
    return dict(dt=dt, dx=dx, dz=dz, firstx=firstx, firstz=firstz, data=data, first=first, nsamp=nsamp)


def pack_gfdb(gf):
    """What a GFDB reader hands the engine for these traces: the reference stores every trace gap-compressed
    (trace_pack, sparse_trace.f90:443-560), so a stored trace begins at its first non-zero sample and ends one sample
    after its last non-zero one when a zero follows (an all-zero trace keeps ONE zero sample at its first position, :490-510).  Returns a new dict with `first`,
    `nsamp` and left-aligned `data` of the packed traces; interior zero gaps stay as zeros of the dense row.
    (In slabs of distances, like make_gfdb.)"""
    data = np.ascontiguousarray(gf["data"], np.float32)
    nx, nz, ng, L = data.shape
    out = np.empty_like(data)
    nsamp = np.empty((nx, nz, ng), np.int32)
    first = np.empty((nx, nz, ng), np.int32)
    slab = max(1, (1 << 23) // (nz * ng * L))
    ar = np.arange(L)[None, None, None, :]

    def pack(a):
        b = min(nx, a + slab)
        d = data[a:b]
        nzm = d != 0
        anyv = nzm.any(-1)
        lo = np.where(anyv, nzm.argmax(-1), 0)
        hi = np.where(anyv, np.minimum(L - 1 - nzm[..., ::-1].argmax(-1) + 1, L - 1), 0)    # "add one of the zeros", :535,545
        ns = (hi - lo + 1).astype(np.int32)
        nsamp[a:b] = ns
        first[a:b] = (np.asarray(gf["first"][a:b], np.int64) + lo).astype(np.int32)
        idx = lo[..., None] + ar
        o = np.take_along_axis(d, np.minimum(idx, L - 1), -1)
        o[ar >= ns[..., None]] = 0.0
        out[a:b] = o

    _slabs(pack, range(0, nx, slab))
    lmax = int(nsamp.max())
    g = dict(gf)
    g.update(data=np.ascontiguousarray(out[..., :lmax]) if lmax < L else out, first=first, nsamp=nsamp)
    return g


def make_receivers(nrec=50, lat0=40.0, lon0=30.0, dmin=150e3, dspan=400e3, comps="ned"):
    """Ring of receivers around (lat0, lon0): azimuth 2 pi (i-1)/N + 0.1, distance dmin + dspan (i-1)/N."""
    i = np.arange(nrec)
    az = 2 * np.pi * i / nrec + 0.1
    d = dmin + dspan * i / nrec
    lat = lat0 + np.degrees(d * np.cos(az) / EARTH_R)
    lon = lon0 + np.degrees(d * np.sin(az) / (EARTH_R * np.cos(np.radians(lat0))))
    return lat, lon, np.zeros(nrec, np.float32), [comps] * nrec, d


TRUE_BILAT = [0., 0., 0., 10000., 1e20, 91., 87., 164., 0., 4000., 2000., 4000., 3000., 2.]


def bilat_strike_sweep(nsrc, step=0.1, base=None, first=0):
    """`bilateral` trial sources differing in strike by `step` degrees (kiwibench.py:136); `first`: index of the sweep's first
    trial (a rank's shard of a longer sweep holds exactly the parameters the whole sweep holds there)."""
    base = np.array(TRUE_BILAT if base is None else base, np.float32)
    p = np.tile(base, (nsrc, 1))
    p[:, 5] = base[5] + step * (1 + first + np.arange(nsrc))
    return p


def mt_from_sdr(strike, dip, rake, m0=7e18):
    """Moment tensor (mxx,myy,mzz,mxy,mxz,myz) of a double couple: R m_unrot R^T with
    R = euler(dip, strike, -rake) (euler.f90:40-43, source_bilat.f90:342,437-438)."""
    a, b, g = np.radians(dip), np.radians(strike), -np.radians(rake)
    ca, cb, cg, sa, sb, sg = np.cos(a), np.cos(b), np.cos(g), np.sin(a), np.sin(b), np.sin(g)
    R = np.array([[cb * cg - ca * sb * sg, -cb * sg - ca * sb * cg, sa * sb],
                  [sb * cg + ca * cb * sg, -sb * sg + ca * cb * cg, -sa * cb],
                  [sa * sg, sa * cg, ca]])
    mu = np.array([[0, 0, -1.], [0, 0, 0], [-1., 0, 0]])
    m = R @ mu @ R.T * m0
    return [m[0, 0], m[1, 1], m[2, 2], m[0, 1], m[0, 2], m[1, 2]]


def mt_sdr_grid(step=10, depth=10000.0, risetime=1.0, m0=7e18):
    """cfg2 trial set: moment_tensor sources over strike x dip x rake (first parameter slowest,
    source.py:119-164): 36 x 10 x 36 = 12 960 at step 10."""
    out = []
    for s in range(0, 360, step):
        for d in range(0, 91, step):
            for r in range(-180, 180, step):
                out.append([0., 0., 0., depth] + mt_from_sdr(s, d, r, m0) + [risetime])
    return np.array(out, np.float32)


def full_taper(first, n, dt, ramp=10.0):
    """4-point cosine taper covering a reference trace that occupies samples first..first+n-1
    (sample j sits at abscissa j*dt for the taper, piecewise_linear_function.f90:225)."""
    t0, t1 = first * dt, (first + n - 1) * dt
    return [t0, t0 + ramp, t1 - ramp, t1], [0., 1., 1., 0.]


# generic continental crust, own values (t_crust2x2_1d_profile layout: vp[8] vs[8] rho[8] thickness[7];
# layers: water, ice, soft sed., hard sed., upper, middle, lower crust, below the crust); m/s, kg/m3, m
SYNTH_CRUST = np.array([1500., 3810., 2500., 4000., 6000., 6400., 6900., 8100.,
                        0., 1940., 1200., 2300., 3500., 3700., 3900., 4600.,
                        1020., 920., 2100., 2400., 2750., 2850., 3000., 3350.,
                        0., 0., 1000., 1000., 10000., 10000., 10000.], np.float32)

# mt_eikonal source of cfg4: vertical fault, 15 km bounding circle clipped to the depth range of the
# synthetic GF database by the constraints below (source_mt_eikonal.f90:71-72 for the parameter order)
CFG4_MT_EIKONAL = [0., 0., 0., 11000., 1., 91., 90., 0., 0., 15000., 2000., 0., 0.9] + \
    mt_from_sdr(91., 87., 164., 1e20) + [1.0]
CFG4_CONSTRAINTS = (np.array([[0, 0, 6500.], [0, 0, 15500.]], np.float32),
                    np.array([[0, 0, -1.], [0, 0, 1.]], np.float32))
SPECTRAL_FILTER = ([0.01, 0.02, 0.1, 0.2], [0., 1., 1., 0.])       # Hz, SURVEY.md 8d


def mt_eikonal_location_grid(n_north=10, n_east=10, n_depth=5, base=None):
    """cfg4 trial set: grid over (north-shift, east-shift, depth), first parameter slowest."""
    base = np.array(CFG4_MT_EIKONAL if base is None else base, np.float32)
    out = []
    for a in range(n_north):
        for b in range(n_east):
            for c in range(n_depth):
                p = base.copy()
                p[1] += 400.0 * (a - n_north // 2)
                p[2] += 400.0 * (b - n_east // 2)
                p[3] += 250.0 * (c - n_depth // 2)
                out.append(p)
    return np.array(out, np.float32)


def mt_eikonal_rupture_grid(base=None):
    """cfg4-nukl trial set: grid over the parameters that shape the rupture itself -- (nucleation-shift-x, nucleation-shift-y,
    rel-rupture-velocity), what source_mt_eikonal.f90:102-124 is inverted for -- at a fixed location: every trial has its own
    rupture-front solve (another start cell or another speed grid), the exact solve cache cannot help.  25 x 9 x 6 = 1350."""
    base = np.array(CFG4_MT_EIKONAL if base is None else base, np.float32)
    out = []
    for a in range(25):
        for b in range(9):
            for c in range(6):
                p = base.copy()
                p[10] = 500.0 * (a - 12)
                p[11] = 500.0 * (b - 4)
                p[12] = 0.70 + 0.05 * c
                out.append(p)
    return np.array(out, np.float32)


def bilat_sweep_5d(nsrc, base=None):
    """cfg5 trial set: the first `nsrc` points of a 10 x 10 x 10 x 10 x 10 product grid over
    (strike, dip, slip-rake, depth, time), last parameter fastest (source.py:119-164)."""
    base = np.array(TRUE_BILAT if base is None else base, np.float32)
    idx = np.arange(nsrc)
    p = np.tile(base, (nsrc, 1))
    p[:, 0] = base[0] + 0.05 * (idx % 10 - 5)
    p[:, 3] = base[3] + 200.0 * ((idx // 10) % 10 - 5)
    p[:, 7] = base[7] + 1.0 * ((idx // 100) % 10 - 5)
    p[:, 6] = base[6] - 0.5 * ((idx // 1000) % 10)
    p[:, 5] = base[5] + 0.5 * ((idx // 10000) % 10 - 5)
    return p


def workload(name, nsrc=None, trial0=0):
    """The BASELINE.json configurations as data: dict(name, sourcetype, true, trials, nrec, nx, method,
    filter, crust, constraints, batch) -- `trials` starts at index `trial0` of the config's trial list."""
    if name == "cfg2":
        grid = mt_sdr_grid()
        n = len(grid) if nsrc is None else nsrc
        tr = grid[(trial0 + np.arange(n)) % len(grid)]
        return dict(name="cfg2-mt-grid", sourcetype="moment_tensor", true=grid[4000], trials=tr, nrec=50, nx=128,
                    method="l2norm", filter=None, crust=None, constraints=None)
    if name == "cfg3":
        base = [0., 0., 0., 10000., 1e20, 91., 87., 164., 0., 4800., 2000., 2000., 3000., 2.]   # 100 centroids
        n = 256 if nsrc is None else nsrc
        tr = bilat_strike_sweep(n, step=0.1, base=base, first=trial0)
        return dict(name="cfg3-bilat", sourcetype="bilateral", true=np.array(base, np.float32), trials=tr, nrec=50,
                    nx=128, method="l2norm", filter=None, crust=None, constraints=None)
    if name in ("cfg3-w256", "cfg3-w600"):
        # cfg3 under the short taper windows real inversions use (body-wave windows of tens to hundreds of samples,
        # python/tunguska/misfit.py style): 256 / 600 samples inside the 4096-sample traces, 500 samples behind the first arrival
        w = workload("cfg3", nsrc, trial0)
        w.update(name=name, window=int(name[6:]), window_offset=500)
        return w
    if name in ("cfg3-ng8", "cfg3-static"):
        # cfg3 over the two other kinds of database real Kiwi installations hold: a far-field database of
        # EIGHT components (gfdb.f90:57: no near-field terms) and a database with non-zero static end values and interior zero
        # gaps (make_gfdb variant "static": traces of several strips, rows that do not end in zero -- no compact descriptors,
        # the tail rule of sparse_trace.f90:696-703 live)
        w = workload("cfg3", nsrc, trial0)
        w.update(name=name, **({"ng": 8} if name == "cfg3-ng8" else {"variant": "static"}))
        return w
    if name == "cfg3-100pt":
        # the north star's "100 sub-faults" taken literally: 25 x 4 sub-fault points of an 18 km x 4.5 km rupture, two
        # source-time-function steps each (200 centroids)
        base = [0., 0., 0., 10000., 1e20, 91., 87., 164., 0., 12000., 6000., 4500., 3000., 0.5]
        n = 256 if nsrc is None else nsrc
        tr = bilat_strike_sweep(n, step=0.1, base=base, first=trial0)
        return dict(name="cfg3-100pt", sourcetype="bilateral", true=np.array(base, np.float32), trials=tr, nrec=50,
                    nx=128, method="l2norm", filter=None, crust=None, constraints=None)
    if name == "cfg3-scatter":
        # the cfg3 source on a LOCATION grid (4 km steps over +-32 km north and east, 8 depths, strike sweep on top), in
        # shuffled order: co-resident workgroups of neighbouring trial sources do not read the same Green's function rows
        base = [0., 0., 0., 10000., 1e20, 91., 87., 164., 0., 4800., 2000., 2000., 3000., 2.]
        n = 256 if nsrc is None else nsrc
        total = 1 << 15
        idx = np.random.default_rng(20261002).permutation(total)[(trial0 + np.arange(n)) % total]
        tr = np.tile(np.array(base, np.float32), (n, 1))
        tr[:, 1] = 4000.0 * (idx % 16 - 8)
        tr[:, 2] = 4000.0 * ((idx // 16) % 16 - 8)
        tr[:, 3] = 8000.0 + 500.0 * ((idx // 256) % 8)
        tr[:, 5] = 91.0 + 0.1 * (idx // 2048)
        return dict(name="cfg3-scatter", sourcetype="bilateral", true=np.array(base, np.float32), trials=tr, nrec=50,
                    nx=128, method="l2norm", filter=None, crust=None, constraints=None)
    if name == "cfg3-bigdb":
        # the HBM regime: the cfg3 source over a 1 GB Green's function database (512 x 12 nodes)
        # at unique locations 5 km apart over 2500 km of distance and ten depths, in shuffled order -- the rows a launch
        # touches are (nearly) the whole database, four times the 256 MiB Infinity Cache
        base = [0., 0., 0., 10000., 1e20, 91., 87., 164., 0., 4800., 2000., 2000., 3000., 2.]
        n = 256 if nsrc is None else nsrc
        total = 5120
        idx = np.random.default_rng(20261003).permutation(total)[(trial0 + np.arange(n)) % total]
        tr = np.tile(np.array(base, np.float32), (n, 1))
        tr[:, 1] = 5000.0 * (idx % 512) - 1280e3
        tr[:, 3] = 8000.0 + 2000.0 * (idx // 512)
        return dict(name="cfg3-bigdb", sourcetype="bilateral", true=np.array(base, np.float32), trials=tr, nrec=50,
                    nx=512, nz=12, method="l2norm", filter=None, crust=None, constraints=None)
    if name in ("cfg3-bigdb4", "cfg3-bigdb4-ordered"):
        # the HBM regime proper: a 4.2 GB database (2048 x 12 nodes) -- sixteen times the 256 MiB
        # Infinity Cache, so that at most a sixteenth of the memory-side traffic can be cache hits -- and trial locations 4 km
        # apart (one per distance node) over the whole distance range and ten depths; shuffled, or ("-ordered") in the order a
        # P3 location grid delivers them (first parameter slowest: neighbours in the list are neighbours in depth, then distance)
        base = [0., 0., 0., 10000., 1e20, 91., 87., 164., 0., 4800., 2000., 2000., 3000., 2.]
        n = 256 if nsrc is None else nsrc
        total = 19000
        k = (trial0 + np.arange(n)) % total
        idx = np.random.default_rng(20261004).permutation(total)[k] if name == "cfg3-bigdb4" else k
        tr = np.tile(np.array(base, np.float32), (n, 1))
        tr[:, 1] = 4000.0 * (idx // 10) - 3900e3
        tr[:, 3] = 8000.0 + 2000.0 * (idx % 10)
        return dict(name=name, sourcetype="bilateral", true=np.array(base, np.float32), trials=tr, nrec=50,
                    nx=2048, nz=12, method="l2norm", filter=None, crust=None, constraints=None)
    if name == "cfg4":
        grid = mt_eikonal_location_grid()
        n = 32 if nsrc is None else nsrc
        tr = grid[(trial0 + np.arange(n)) % len(grid)]
        return dict(name="cfg4-mt-eikonal", sourcetype="mt_eikonal", true=np.array(CFG4_MT_EIKONAL, np.float32),
                    trials=tr, nrec=200, nx=160, method="l2norm", filter=None, crust=SYNTH_CRUST,
                    constraints=CFG4_CONSTRAINTS)
    if name == "cfg4-nukl":
        grid = mt_eikonal_rupture_grid()
        n = 32 if nsrc is None else nsrc
        tr = grid[(trial0 + np.arange(n)) % len(grid)]
        return dict(name="cfg4-nukl-mt-eikonal", sourcetype="mt_eikonal", true=np.array(CFG4_MT_EIKONAL, np.float32),
                    trials=tr, nrec=200, nx=160, method="l2norm", filter=None, crust=SYNTH_CRUST,
                    constraints=CFG4_CONSTRAINTS)
    if name == "cfg5":
        n = 256 if nsrc is None else nsrc
        tr = bilat_sweep_5d(n + trial0)[trial0:]
        return dict(name="cfg5-spectral", sourcetype="bilateral", true=np.array(TRUE_BILAT, np.float32), trials=tr,
                    nrec=50, nx=128, method="ampspec_l2norm", filter=SPECTRAL_FILTER, crust=None, constraints=None)
    if name == "cfg5-td":
        # the cfg5 trial set under a TIME-domain norm on frequency-filtered traces (the comparator transforms forward, filters,
        # transforms back: comparator.f90:810-813,1233-1263) -- the other way the reference uses its misfit filter
        w = workload("cfg5", nsrc, trial0)
        w.update(name="cfg5-td-filtered", method="l2norm")
        return w
    raise ValueError("unknown workload " + name)
