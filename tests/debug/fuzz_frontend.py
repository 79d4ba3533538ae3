"""Front end (DC block -> gain -> AGC, row f1), random channel counts / call lengths / signal shapes, bit-exact against the oracle
(run from the repository root on a GPU box):   gpurun -- python tests/debug/fuzz_frontend.py [seconds] [seed]
Whole 64-channel groups take the two-wave pipeline kernel, the rest the one-wave kernel; both must agree with the oracle."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from gpuhelp import msdr  # noqa: E402  (imports torch first)
import orclib  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
orc = orclib.Oracle()
ctx = msdr.Context(0)
B = 128
t_end = time.time() + budget
case = bad = 0
while time.time() < t_end:
    case += 1
    rng = np.random.default_rng([seed, case])
    ch = int(rng.choice([1, 5, 16, 48, 64, 65, 128, 192, 200, 256]))
    os.environ.pop("MSDR_FRONTEND_PIPE_CH", None)
    per_group = int(rng.choice([0, 16, 32, 64]))          # frontend_pipe4_kernel<CH>: the host's own choice, or forced (read at create time)
    if per_group:
        os.environ["MSDR_FRONTEND_PIPE_CH"] = str(per_group)
    nblk = int(rng.integers(1, 60))
    n = nblk * B
    kind = int(rng.integers(0, 4))
    t = np.arange(n)
    if kind == 0:
        x = rng.integers(0, 65536, (ch, n))
    elif kind == 1:
        x = 32768 + rng.integers(-200, 201, (ch, n))
    elif kind == 2:
        x = 32768 + (rng.uniform(100, 32000, (ch, 1)) * np.cos(2 * np.pi * rng.uniform(0.001, 0.4, (ch, 1)) * t)).astype(np.int64) + rng.integers(-30, 31, (ch, n))
    else:
        x = np.where((t // int(rng.integers(3, 400))) % 2, 65535, 0)[None, :].repeat(ch, 0)
    x = np.clip(x, 0, 65535).astype(np.uint16)
    fe = msdr.Frontend(ctx, ch)
    fe.prime(x[:, 0])
    got = np.empty((ch, n), np.int16)
    b0 = 0
    while b0 < nblk:
        m = int(min(nblk - b0, rng.choice([1, 1, 2, 3, 7, 20, nblk]))) * B
        seg = np.ascontiguousarray(x[:, b0 * B:b0 * B + m])
        dx, dy = ctx.to_device(seg), ctx.array((ch, m), np.int16)
        fe.update(dx, dy, m)
        got[:, b0 * B:b0 * B + m] = dy.download()
        b0 += m // B
    for c in sorted(set(int(v) for v in rng.choice(ch, min(ch, 6), replace=False)) | {0, ch - 1}):
        f = orc.frontend_new(first_conversion=int(x[c, 0]))
        want = orc.frontend_run(f, x[c])
        if not np.array_equal(got[c], want):
            bad += 1
            print("MISMATCH", dict(seed=seed, case=case, ch=ch, nblk=nblk, kind=kind, channel=c, first=int(np.argmax(got[c] != want))), flush=True)
            break
    fe.close()
print("fuzz_frontend done: %d cases, %d mismatches (seed %d)" % (case, bad, seed))
