"""Channel sharding across the GPUs of one node (one process per GPU, torch.distributed; backend "nccl" is
RCCL on ROCm, "gloo" in the CPU tests).

Receiver channels are independent: rank r owns a contiguous channel range, reads only its own
IF shard from its own HBM and keeps taps/state locally -- no exchange during compute.  The ONE collective on
the path is the gather of demodulated audio after the IIR stage."""
import torch
import torch.distributed as dist


def channel_shard(total_channels, world_size, rank):
    """Contiguous range [start, start+count) of `rank`; the remainder goes to the first ranks."""
    base, rem = divmod(int(total_channels), int(world_size))
    count = base + (1 if rank < rem else 0)
    start = rank * base + min(rank, rem)
    return start, count


def gather_audio(local, total_channels, group=None):
    """All-gather the per-rank audio [count_r, n] into [total_channels, n] (same on every rank).
    Uneven shards are padded to the largest shard for the collective and trimmed afterwards."""
    world = dist.get_world_size(group)
    dtype = local.dtype
    local = local.contiguous().view(torch.uint8)      # bytes: every backend (RCCL, gloo) moves them, whatever the audio type
    n = local.shape[1]
    counts = [channel_shard(total_channels, world, r)[1] for r in range(world)]
    cmax = max(counts)
    send = local
    if local.shape[0] != cmax:
        send = torch.zeros((cmax, n), dtype=local.dtype, device=local.device)
        send[:local.shape[0]] = local
    recv = torch.empty((world * cmax, n), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(recv, send.contiguous(), group=group)
    if not all(c == cmax for c in counts):
        recv = torch.cat([recv[r * cmax:r * cmax + counts[r]] for r in range(world)], dim=0)
    return recv.view(dtype)


def max_over_ranks(value, device, group=None):
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())
