"""Generates tests/golden/lm_vectors.npz: what the reference's own single-precision MINPACK (sminpack/lmdif.f, built
unmodified into oracle/_ref by `make -C oracle ref`) returns for the problems of tests/lm_problems.py, with
minimize_lm's settings and with lmdif1's.  Run in the dev container (needs /root/reference for the _ref build):

    python tests/golden/make_golden_lm.py
"""
import os
import sys
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import lm_problems as P  # noqa: E402
from oracle import ko  # noqa: E402

warnings.simplefilter("ignore")
R = ko.ref()
assert R is not None, "build oracle/_ref first"
out = {}
for sname, st in P.SETTINGS.items():
    for name, m, n, x0, f in P.PROBLEMS:
        x, fvec, info, nfev = P.run_reference(R, name, m, n, x0, f, st)
        key = sname + "/" + name
        out[key + "/x"], out[key + "/fvec"], out[key + "/info_nfev"] = x, fvec, np.array([info, nfev], np.int32)
np.savez_compressed(os.path.join(HERE, "lm_vectors.npz"), **out)
print("wrote", len(out) // 3, "cases")
