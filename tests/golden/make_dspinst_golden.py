#!/usr/bin/env python3
"""Generate tests/golden/dspinst_prims.npz: inputs and outputs of the Teensy Audio library's DSP-instruction wrappers
(src/Audio/utility/dspinst.h, the header's own plain-C KINETISL bodies compiled by oracle/build_ref.sh into
oracle/_ref/libmsdr_ref.so): smulwb / smulwt (signed_multiply_32x16b/t), ssat-with-shift (signed_saturate_rshift, 16 bits,
the shifts the path uses: 0, 14, 15) and pack_16b_16b.  Runs ONLY in the build container; the fixture is numbers.
The oracle's primitives (mulw16, ssat16(v >> s), the history-word packing) are checked AGAINST these vectors by
tests/test_oracle_pinned.py, also where /root/reference is absent."""
import ctypes as C
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import orclib  # noqa: E402


def inputs():
    rng = np.random.default_rng(20261005)
    edge = np.array([0, 1, -1, 2, -2, 0x7FFF, 0x8000, 0xFFFF, 0x10000, -0x10000, 0x3FFF, 0x4000, -0x4000, 0x3FFFFFFF, 0x40000000,
                     -0x40000000, 0x7FFFFFFF, -0x80000000, 0x7FFF0000, -0x7FFF0000, 0x00008000, 0x80008000 - (1 << 32)], np.int64)
    a = np.concatenate([edge, rng.integers(-2 ** 31, 2 ** 31, 20000)]).astype(np.int32)
    b = np.concatenate([edge[::-1], rng.integers(-2 ** 31, 2 ** 31, 20000)]).astype(np.int32)
    # every edge against every edge as well
    ea, eb = np.meshgrid(edge.astype(np.int32), edge.astype(np.int32))
    return np.concatenate([a, ea.ravel()]), np.concatenate([b, eb.ravel()])


def main():
    L = C.CDLL(orclib.REF_SO)
    for n in ("dspinst_signed_multiply_32x16b", "dspinst_signed_multiply_32x16t"):
        getattr(L, n).restype = C.c_int32
        getattr(L, n).argtypes = [C.c_int32, C.c_uint32]
    L.dspinst_signed_saturate_rshift.restype = C.c_int32
    L.dspinst_signed_saturate_rshift.argtypes = [C.c_int32, C.c_int, C.c_int]
    L.dspinst_pack_16b_16b.restype = C.c_uint32
    L.dspinst_pack_16b_16b.argtypes = [C.c_int32, C.c_int32]
    a, b = inputs()
    out = {"a": a, "b": b}
    out["mulwb"] = np.array([L.dspinst_signed_multiply_32x16b(int(x), int(y) & 0xFFFFFFFF) for x, y in zip(a, b)], np.int32)
    out["mulwt"] = np.array([L.dspinst_signed_multiply_32x16t(int(x), int(y) & 0xFFFFFFFF) for x, y in zip(a, b)], np.int32)
    for sh in (0, 14, 15):
        out["ssat16_asr%d" % sh] = np.array([L.dspinst_signed_saturate_rshift(int(x), 16, sh) for x in a], np.int32)
    out["pack_bb"] = np.array([L.dspinst_pack_16b_16b(int(x), int(y)) for x, y in zip(a, b)], np.uint32)
    path = os.path.join(HERE, "dspinst_prims.npz")
    np.savez_compressed(path, **out)
    print(path, a.size, "cases, sha256", hashlib.sha256(open(path, "rb").read()).hexdigest())


if __name__ == "__main__":
    main()
