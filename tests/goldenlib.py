"""Loader for the committed golden fixtures (tests/golden/golden.npz + manifest.json)."""
import hashlib
import json
import os

import numpy as np

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


class Golden(dict):
    meta = None


def load():
    z = np.load(os.path.join(HERE, "golden.npz"))
    g = Golden({k: z[k] for k in z.files})
    g.meta = json.load(open(os.path.join(HERE, "manifest.json")))
    return g


def verify(g):
    """Every array must match the SHA-256 recorded when it was generated from the reference."""
    bad = []
    for k, e in g.meta["entries"].items():
        if hashlib.sha256(np.ascontiguousarray(g[k]).tobytes()).hexdigest() != e["sha256"]:
            bad.append(k)
    return bad
