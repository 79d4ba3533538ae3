"""Channel sharding across the GPUs of one node (one process per GPU, torch.distributed; backend "nccl" is
RCCL on ROCm, "gloo" in the CPU tests).

Receiver channels are independent: rank r owns a contiguous channel range, reads only its own
IF shard from its own HBM and keeps taps/state locally -- no exchange during compute.  The ONE collective on
the path is the gather of demodulated audio after the IIR stage:

  gather_audio        every rank receives all audio (all_gather_into_tensor)
  gather_audio_root   only `root` receives it (what a play-out host needs; a peer sends its shard over ONE xGMI link, the root
                      takes in world - 1 links at once: SURVEY.md 7.2(5))
  OverlappedGather    block k's audio is gathered while block k + 1 is demodulated (two audio buffers)

All of them move BYTES (a uint8 view), so int16 (Q15 flavour) and fp32 audio take the same path and an int16 gather moves half
the bytes of an fp32 one."""
import torch
import torch.distributed as dist


def channel_shard(total_channels, world_size, rank):
    """Contiguous range [start, start+count) of `rank`; the remainder goes to the first ranks."""
    base, rem = divmod(int(total_channels), int(world_size))
    count = base + (1 if rank < rem else 0)
    start = rank * base + min(rank, rem)
    return start, count


def _padded_bytes(local, total_channels, world):
    """(send buffer as bytes, padded to the largest shard; per-rank channel counts; largest count)"""
    local = local.contiguous().view(torch.uint8)      # bytes: every backend (RCCL, gloo) moves them, whatever the audio type
    counts = [channel_shard(total_channels, world, r)[1] for r in range(world)]
    cmax = max(counts)
    send = local
    if local.shape[0] != cmax:
        send = torch.zeros((cmax, local.shape[1]), dtype=local.dtype, device=local.device)
        send[:local.shape[0]] = local
    return send.contiguous(), counts, cmax


def _trim(recv, counts, cmax):
    if all(c == cmax for c in counts):
        return recv
    return torch.cat([recv[r * cmax:r * cmax + counts[r]] for r in range(len(counts))], dim=0)


def gather_audio(local, total_channels, group=None):
    """All-gather the per-rank audio [count_r, n] into [total_channels, n] (same on every rank).
    Uneven shards are padded to the largest shard for the collective and trimmed afterwards."""
    world = dist.get_world_size(group)
    dtype = local.dtype
    send, counts, cmax = _padded_bytes(local, total_channels, world)
    recv = torch.empty((world * cmax, send.shape[1]), dtype=send.dtype, device=send.device)
    dist.all_gather_into_tensor(recv, send, group=group)
    return _trim(recv, counts, cmax).view(dtype)


def gather_audio_root(local, total_channels, root=0, group=None):
    """Gather to ONE rank: returns [total_channels, n] on `root`, None elsewhere."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    dtype = local.dtype
    send, counts, cmax = _padded_bytes(local, total_channels, world)
    if rank == root:
        recv = torch.empty((world * cmax, send.shape[1]), dtype=send.dtype, device=send.device)
        dist.gather(send, list(recv.view(world, cmax, send.shape[1]).unbind(0)), dst=root, group=group)
        return _trim(recv, counts, cmax).view(dtype)
    dist.gather(send, None, dst=root, group=group)
    return None


class OverlappedGather:
    """gather(k) runs while compute(k + 1) does: the caller alternates between the two audio buffers this object hands out.

        og = OverlappedGather(total_channels, count, n, dtype, device, root=0 or None)
        for k in range(blocks):
            buf = og.buffer(k)            # waits until the gather that last used this buffer has finished
            ... demodulate block k into buf ...
            og.submit(k)                  # asynchronous: returns at once
        og.finish()                       # all gathers done
        og.result(k)                      # gathered audio of block k (k = last or last - 1), on root (all ranks if root is None)

    Stream ordering.  With RCCL the collective runs on the process group's own stream, ordered after the work already queued on
    torch's CURRENT stream at submit time, and work.wait() makes torch's current stream wait for it.  The demodulation, however,
    runs on the msdr context's stream (msdr.Context(device) creates its own unless it is handed one).  Pass that stream as
    `compute_stream` (the hipStream_t from msdr_ctx_stream(), as an int / ctypes pointer, or a torch.cuda.Stream): submit() then
    makes the current stream wait for the demodulation queued so far before the collective starts, and buffer() makes the compute
    stream wait for the gather that last read the buffer before block k + 2 overwrites it.  Without `compute_stream` the caller
    must run the context ON torch's current stream (bench.py does: msdr.Context(dev, stream.cuda_stream)); host tensors (gloo)
    need no ordering."""

    def __init__(self, total_channels, count, n, dtype, device, root=None, group=None, compute_stream=None):
        self.total, self.root, self.group = int(total_channels), root, group
        self.cstream = None
        if compute_stream is not None and torch.device(device).type == "cuda":
            if isinstance(compute_stream, torch.cuda.Stream):
                self.cstream = compute_stream
            else:
                h = getattr(compute_stream, "value", compute_stream)      # ctypes.c_void_p or int
                self.cstream = torch.cuda.ExternalStream(int(h), device=device) if h else None   # NULL = the legacy default stream: ordered with everything
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.dtype = dtype
        self.bufs = [torch.zeros((count, n), dtype=dtype, device=device) for _ in range(2)]
        self.counts = [channel_shard(self.total, self.world, r)[1] for r in range(self.world)]
        self.cmax = max(self.counts)
        nbytes = n * torch.empty((), dtype=dtype).element_size()
        self.send = [torch.zeros((self.cmax, nbytes), dtype=torch.uint8, device=device) if count != self.cmax else None for _ in range(2)]
        receives = root is None or self.rank == root
        self.recv = [torch.empty((self.world * self.cmax, nbytes), dtype=torch.uint8, device=device) if receives else None for _ in range(2)]
        self.work = [None, None]
        self.submitted = 0

    def buffer(self, k):
        w = self.work[k & 1]
        if w is not None:
            w.wait()                                  # torch's current stream now waits for the gather
            self.work[k & 1] = None
            if self.cstream is not None and self.cstream != torch.cuda.current_stream(self.cstream.device):
                self.cstream.wait_stream(torch.cuda.current_stream(self.cstream.device))      # ... and so does the demodulation of block k
        return self.bufs[k & 1]

    def submit(self, k):
        i = k & 1
        if self.cstream is not None and self.cstream != torch.cuda.current_stream(self.cstream.device):
            torch.cuda.current_stream(self.cstream.device).wait_stream(self.cstream)          # the demodulation of block k, queued on the context's stream
        send = self.bufs[i].view(torch.uint8)
        if self.send[i] is not None:                  # a shard smaller than the largest: padded copy
            self.send[i][:send.shape[0]] = send
            send = self.send[i]
        if self.root is None:
            self.work[i] = dist.all_gather_into_tensor(self.recv[i], send, group=self.group, async_op=True)
        elif self.rank == self.root:
            self.work[i] = dist.gather(send, list(self.recv[i].view(self.world, self.cmax, -1).unbind(0)), dst=self.root, group=self.group, async_op=True)
        else:
            self.work[i] = dist.gather(send, None, dst=self.root, group=self.group, async_op=True)
        self.submitted += 1

    def finish(self):
        for i in (0, 1):
            if self.work[i] is not None:
                self.work[i].wait()
                self.work[i] = None

    def result(self, k):
        r = self.recv[k & 1]
        return None if r is None else _trim(r, self.counts, self.cmax).view(self.dtype)


def max_over_ranks(value, device, group=None):
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())
