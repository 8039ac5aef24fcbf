"""Generates tests/golden/flat_golden.json from the reference's own data fixtures with the independent
numpy emulation of the strict f32 fold (oracle/np_ref.py).  The reference itself (Rust) cannot be built or
imported in the build container, so these vectors pin the C oracle against a second implementation, and
reproduce the golden candidates listed in SURVEY.md section 8c.  Run: python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import np_ref as R  # noqa: E402

base = np.fromfile(os.path.join(HERE, "gist_1000.bin"), dtype=np.float32).reshape(1000, 960)
test = np.fromfile(os.path.join(HERE, "gist_test.bin"), dtype=np.float32).reshape(1000, 960)


def hexes(a):
    return [float(x).hex() for x in a]


out = {"flat_l2": {}, "flat_cosine": {}, "cached_l2": {}}
for q in (0, 1, 2, 3, 500, 999):
    i, d = R.flat_knn(base, test[q], 10)
    out["flat_l2"][str(q)] = {"idx": i.tolist(), "dist": hexes(d)}
    i, d = R.flat_knn(base, test[q], 10, cosine=True)
    out["flat_cosine"][str(q)] = {"idx": i.tolist(), "dist": hexes(d)}
    d = R.l2_cached_rows(base[:16], test[q])
    out["cached_l2"][str(q)] = hexes(d)
b12 = np.ascontiguousarray(base[:, :12])
i, d = R.flat_knn(b12, b12[200], 4)
out["clip12_row200_k4"] = {"idx": i.tolist(), "dist": hexes(d)}
out["selfdot_first8"] = hexes(R.selfdot_rows(base[:8]))
json.dump(out, open(os.path.join(HERE, "flat_golden.json"), "w"), indent=1)
print("written", os.path.join(HERE, "flat_golden.json"))
