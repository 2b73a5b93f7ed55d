"""Multi-GPU sharding of the ORB path: frames are independent units, so the batch
is partitioned contiguously across ranks with NO data-path collective; the only exchange is one
all-gather of fixed-size per-frame result records at the end (RCCL on GPUs, gloo in CPU tests)
and the max-over-ranks of the timed region.  Local BA: graphs are replicas, no collective."""
import torch
import torch.distributed as dist


def frame_range(rank, world, total_frames):
    """Contiguous partition of [0,total) ; the first (total % world) ranks get one extra frame."""
    base, rem = divmod(total_frames, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def weak_first_frame(rank, per_rank_batch):
    """Weak scaling (bench.py): every rank owns `per_rank_batch` frames of its own."""
    return rank * per_rank_batch


def max_over_ranks(value, device="cpu"):
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def allgather_records(records):
    """records: int32 tensor [frames_local_padded, k] (e.g. count, mono_index per frame), same shape on
    every rank.  Returns [world, frames_local_padded, k] on every rank."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return records.unsqueeze(0)
    world = dist.get_world_size()
    flat = torch.empty((world * records.shape[0],) + tuple(records.shape[1:]), dtype=records.dtype, device=records.device)
    dist.all_gather_into_tensor(flat, records.contiguous())        # concatenation form: accepted by gloo and RCCL
    return flat.view((world,) + tuple(records.shape))


def make_ba_exchange(xbuf, stride, group=None):
    """Exchange callback for orbhip.BaBatch.solve_sharded: xbuf = 1-D float64 torch tensor [world * stride] on the GPU the batch
    lives on; slot r = xbuf[r*stride : (r+1)*stride].  One all-gather per call (backend "nccl" = RCCL over xGMI on GPUs; with a
    CPU backend such as gloo -- rehearsals on one card -- the payload takes a round trip through host memory)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    on_gpu = dist.get_backend(group) == "nccl"

    def exchange(stage, count):
        mine = xbuf[rank * stride: rank * stride + count]
        if on_gpu:
            send = mine.clone()
            outs = [xbuf[r * stride: r * stride + count] for r in range(world)]
            dist.all_gather(outs, send, group=group)
            torch.cuda.synchronize()
        else:
            send = mine.cpu()
            outs = [torch.empty_like(send) for _ in range(world)]
            dist.all_gather(outs, send, group=group)
            for r in range(world):
                if r != rank:
                    xbuf[r * stride: r * stride + count].copy_(outs[r])
            torch.cuda.synchronize()
    return exchange
