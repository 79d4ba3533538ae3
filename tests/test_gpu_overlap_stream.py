"""msdr_dist.OverlappedGather with a context that runs on its OWN stream (msdr.Context(device) creates one): the gather of block k
must start after the demodulation of block k and block k + 2 must not overwrite a buffer the gather still reads.  One rank on the
one card of the test box (RCCL, world size 1: every ordering step of the path is taken; the data movement with peers is
tests/test_dist_gloo.py).  Runs in a child process so that the process group does not outlive the test."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import os, socket, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, os.path.join(%(root)r, "minimal-sdr_amd", "python"))
sys.path.insert(0, os.path.join(%(root)r, "tests"))
import msdr, msdr_dist, orclib
s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
ctx = msdr.Context(0)                                   # its own stream: NOT torch's current stream
assert ctx.stream() != 0 and ctx.stream() != torch.cuda.current_stream(dev).cuda_stream
ch, n, blocks = 512, 1 << 16, 6                         # 32 Msamples per block: the demodulation takes long enough to lose a race against
taps = msdr.calc_fir_coeffs(102, 2800)[:102]
chain = msdr.Chain(ctx, msdr.ARITH_Q15, ch, taps, taps, mode=msdr.MODE_AM)
g = torch.Generator(device=dev); g.manual_seed(5)
x = torch.randint(-12000, 12001, (ch, blocks * n), device=dev, generator=g, dtype=torch.int32).to(torch.int16)
torch.cuda.synchronize(dev)
og = msdr_dist.OverlappedGather(ch, ch, n, torch.int16, dev, root=None, compute_stream=ctx.stream())
got = {}
for k in range(blocks):
    buf = og.buffer(k)
    if k >= 2:
        got[k - 2] = og.result(k - 2).clone()          # on torch's stream, which buffer() made wait for that gather
    xs = x[:, k * n:(k + 1) * n].contiguous()
    torch.cuda.current_stream(dev).synchronize()       # (the slice copy ran on torch's stream; the library reads it on its own)
    chain.process(xs.data_ptr(), buf.data_ptr(), n)
    og.submit(k)
og.finish()
for k in (blocks - 2, blocks - 1):
    got[k] = og.result(k).clone()
torch.cuda.synchronize(dev)
orc = orclib.Oracle()
xh = x.cpu().numpy()
for c in (0, 255, ch - 1):
    want = orc.chain_q15(xh[c], msdr.MODE_AM, taps, taps)
    have = np.concatenate([got[k][c].cpu().numpy() for k in range(blocks)])
    assert np.array_equal(have, want), "channel %%d differs" %% c
chain.close(); ctx.close()
dist.destroy_process_group()
print("OK")
'''


def test_overlapped_gather_with_a_context_on_its_own_stream():
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT}], cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       text=True, timeout=600)
    assert r.returncode == 0 and "OK" in r.stdout, (r.stdout + r.stderr)[-3000:]
