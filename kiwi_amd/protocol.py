"""Driver for the Fortran protocol host kiwi_amd/fortran/minimizer_hip (the stdin/stdout command
protocol of Kiwi's `minimizer`): the Python-3 counterpart of SeismosizerProcess._do
(python/tunguska/seismosizer.py:306-338), plus writers for the plain files the host reads."""
import os
import struct
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
HOST = os.path.join(HERE, "fortran", "minimizer_hip")


class SeismosizerReturnedError(Exception):
    """The host answered '<cmd>: nok' (seismosizer.py:96-100)."""


def build_host():
    subprocess.check_call(["make", "-s", "-C", os.path.join(HERE, "fortran"), "minimizer_hip"])
    return HOST


class MinimizerProcess:
    def __init__(self, exe=None, env=None):
        self.p = subprocess.Popen([exe or HOST], stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True, bufsize=1,
                                  env=env)

    def do(self, *words):
        """Send one command line; return the answer string ('' if none) or raise on nok."""
        line = " ".join(str(w) for w in words)
        self.p.stdin.write(line + "\n")
        self.p.stdin.flush()
        head = self.p.stdout.readline()
        if not head:
            raise RuntimeError("minimizer_hip died")
        head = head.rstrip("\n")
        cmd, _, status = head.partition(": ")
        sent = line.split('#')[0].split()
        if sent and cmd != sent[0]:
            raise RuntimeError("protocol out of sync: sent %r, got %r" % (words[0], head))
        answer = ""
        if status.endswith(">"):
            answer = self.p.stdout.readline().rstrip("\n")
        if status.startswith("nok"):
            raise SeismosizerReturnedError("%s: %s" % (cmd, answer))
        return answer

    def eval_sources(self, sourcetype, paramfile, outfile):
        """`eval_sources`: (number of sources, failings) -- failings 0-based like seismosizer.py:716-717."""
        words = self.do("eval_sources", sourcetype, paramfile, outfile).split()
        failings = [int(w) - 1 for w in words[2:]] if len(words) > 1 and words[1] == "failed" else []
        return int(words[0]), failings

    def close(self):
        if self.p.poll() is None:
            self.p.stdin.close()
            self.p.wait(timeout=60)


def write_flat_gfdb(base, gf):
    """<base>.kiwiflat as minimizer_hip's set_database reads it (header, first, nsamp, dense data)."""
    data = np.ascontiguousarray(gf["data"], np.float32)
    nx, nz, ng, L = data.shape
    with open(base + ".kiwiflat", "wb") as f:
        f.write(b"KIWIFLAT")
        f.write(struct.pack("<5i", 1, nx, nz, ng, L))
        f.write(struct.pack("<5f", gf["dt"], gf["dx"], gf["dz"], gf["firstx"], gf["firstz"]))
        f.write(np.ascontiguousarray(gf["first"], np.int32).tobytes())
        f.write(np.ascontiguousarray(gf["nsamp"], np.int32).tobytes())
        f.write(data.tobytes())


def write_receivers(path, lat, lon, comps, depth=None):
    with open(path, "w") as f:
        for i in range(len(lat)):
            if depth is None:
                f.write("%.12f %.12f %s\n" % (lat[i], lon[i], comps[i]))
            else:
                f.write("%.12f %.12f %.3f %s\n" % (lat[i], lon[i], depth[i], comps[i]))


def write_table(path, t0, dt, data):
    with open(path, "w") as f:
        for i, v in enumerate(data):
            f.write("%.9g %.9e\n" % (t0 + i * dt, v))


def read_table(path):
    a = np.loadtxt(path, ndmin=2)
    return a[:, 0], a[:, 1].astype(np.float32)
