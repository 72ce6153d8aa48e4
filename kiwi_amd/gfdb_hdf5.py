"""Kiwi's on-disk Green's function database (HDF5; gfdb_io_hdf.f90) <-> the dense arrays the engine takes.

    gf = gfdb_hdf5.read("/path/to/db")            # -> dict(dt, dx, dz, firstx, firstz, data, first, nsamp)
    engine.set_database(gf["dt"], gf["dx"], gf["dz"], gf["firstx"], gf["firstz"], gf["data"], gf["first"], gf["nsamp"])
    python -m kiwi_amd.gfdb_hdf5 /path/to/db out_base     # -> out_base.kiwiflat for kiwi_amd/fortran/minimizer_hip

The C side (kiwi_amd/gfdb/kiwi_gfdb_hdf5.c -> kiwi_amd/libkiwi_gfdb.so) needs the HDF5 C library and is built on
demand; it is separate from libkiwi_hip.so.  `write` produces a database in the same layout (strips as
`trace_pack` makes them, sparse_trace.f90:443-555) -- used to convert dense tables and by the tests."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libkiwi_gfdb.so")
MAXGAP = 5                      # sparse_trace.f90:25


class GfdbError(Exception):
    pass


class Index(C.Structure):
    _fields_ = [("dt", C.c_float), ("dx", C.c_float), ("dz", C.c_float), ("firstx", C.c_float), ("firstz", C.c_float),
                ("nchunks", C.c_int), ("nx", C.c_int), ("nxc", C.c_int), ("nz", C.c_int), ("ng", C.c_int)]


_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", os.path.join(HERE, "gfdb")])
    return LIB_PATH


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        _lib = C.CDLL(LIB_PATH)
        fp, ip, llp = C.POINTER(C.c_float), C.POINTER(C.c_int), C.POINTER(C.c_longlong)
        _lib.kiwi_gfdb_read_index.argtypes = [C.c_char_p, C.POINTER(Index), C.c_char_p, C.c_int]
        _lib.kiwi_gfdb_read_dense.argtypes = [C.c_char_p, C.c_int, fp, ip, ip, ip, C.c_char_p, C.c_int]
        _lib.kiwi_gfdb_write.argtypes = [C.c_char_p, C.POINTER(Index), fp, llp, ip, ip, llp, C.c_char_p, C.c_int]
    return _lib


def _ck(rc, buf):
    if rc != 0:
        raise GfdbError(buf.value.decode() or "gfdb: error")


def read_index(base):
    L = load()
    ix, buf = Index(), C.create_string_buffer(1024)
    _ck(L.kiwi_gfdb_read_index(base.encode(), C.byref(ix), buf, 1024), buf)
    return ix


def read(base):
    """The whole database as dense arrays: data[nx, nz, ng, L] (zero padded), first[nx, nz, ng] (sample index of the
    first stored sample), nsamp (span length; 0 = trace not stored)."""
    L = load()
    ix = read_index(base)
    shape = (ix.nx, ix.nz, ix.ng)
    first, nsamp = np.zeros(shape, np.int32), np.zeros(shape, np.int32)
    lmax, buf = C.c_int(), C.create_string_buffer(1024)
    ip = C.POINTER(C.c_int)
    _ck(L.kiwi_gfdb_read_dense(base.encode(), 0, None, first.ctypes.data_as(ip), nsamp.ctypes.data_as(ip), C.byref(lmax),
                               buf, 1024), buf)
    Lrow = max(lmax.value, 1)
    data = np.zeros(shape + (Lrow,), np.float32)
    _ck(L.kiwi_gfdb_read_dense(base.encode(), Lrow, data.ctypes.data_as(C.POINTER(C.c_float)), first.ctypes.data_as(ip),
                               nsamp.ctypes.data_as(ip), C.byref(lmax), buf, 1024), buf)
    return dict(dt=ix.dt, dx=ix.dx, dz=ix.dz, firstx=ix.firstx, firstz=ix.firstz, data=data, first=first, nsamp=nsamp,
                nchunks=ix.nchunks, nxc=ix.nxc)


def pack_trace(first, data):
    """trace_pack (sparse_trace.f90:443-555): strips of a dense trace whose first sample has index `first`.  A strip
    ends when more than MAXGAP consecutive zeros follow; it keeps ONE of the trailing zeros (so that the repeated end
    value is zero).  An all-zero trace becomes a single zero at its first sample.  Returns [(offset, samples), ...]."""
    data = np.asarray(data, np.float32)
    strips = []
    interest, gap, ibeg, iend = False, 0, 0, 0
    for i, v in enumerate(data):
        if v != 0.0:
            if not interest:
                interest, ibeg = True, i
            gap, iend = 0, i
        elif interest:
            gap += 1
            if gap > MAXGAP:
                strips.append((first + ibeg, data[ibeg:iend + 2].copy()))
                interest = False
    if interest:
        strips.append((first + ibeg, data[ibeg:iend + (2 if gap > 0 else 1)].copy()))
    if not strips:
        strips.append((first, np.zeros(1, np.float32)))
    return strips


def write(base, gf, nchunks=1):
    """Writes dict(dt, dx, dz, firstx, firstz, data[nx,nz,ng,L], first, nsamp) as <base>.index + <base>.<i>.chunk
    (gfdb.f90:186-200 for the distances per chunk).  nsamp == 0 leaves a trace out."""
    L = load()
    data = np.asarray(gf["data"], np.float32)
    nx, nz, ng, _ = data.shape
    nchunks = min(nchunks, nx)
    nxc = nx // nchunks + 1
    if nxc > nx:
        nxc = nx
    while nx - nxc * (nchunks - 1) <= 0:
        nxc -= 1
    ix = Index(gf["dt"], gf["dx"], gf["dz"], gf["firstx"], gf["firstz"], nchunks, nx, nxc, nz, ng)
    pk, pk_ofs, pofs, ofs, st_ofs = [], [0], [], [], [0]
    for i in range(nx):
        for j in range(nz):
            for k in range(ng):
                n = int(gf["nsamp"][i, j, k])
                if n > 0:
                    p = 1
                    for o, d in pack_trace(int(gf["first"][i, j, k]), data[i, j, k, :n]):
                        pofs.append(p)
                        ofs.append(o)
                        pk.append(d)
                        p += len(d)
                    pk_ofs.append(pk_ofs[-1] + p - 1)
                else:
                    pk_ofs.append(pk_ofs[-1])
                st_ofs.append(len(pofs))
    pk = np.ascontiguousarray(np.concatenate(pk) if pk else np.zeros(1), np.float32)
    pk_ofs, st_ofs = np.array(pk_ofs, np.int64), np.array(st_ofs, np.int64)
    pofs, ofs = np.array(pofs or [0], np.int32), np.array(ofs or [0], np.int32)
    buf = C.create_string_buffer(1024)
    ip, llp = C.POINTER(C.c_int), C.POINTER(C.c_longlong)
    _ck(L.kiwi_gfdb_write(base.encode(), C.byref(ix), pk.ctypes.data_as(C.POINTER(C.c_float)), pk_ofs.ctypes.data_as(llp),
                          pofs.ctypes.data_as(ip), ofs.ctypes.data_as(ip), st_ofs.ctypes.data_as(llp), buf, 1024), buf)


if __name__ == "__main__":
    if len(sys.argv) != 3:
        sys.exit("usage: python -m kiwi_amd.gfdb_hdf5 <gfdb base> <output base>   (writes <output base>.kiwiflat)")
    from .protocol import write_flat_gfdb
    write_flat_gfdb(sys.argv[2], read(sys.argv[1]))
