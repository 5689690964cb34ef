"""Frame-parallel multi-GPU plumbing: frames are independent given the calibration
tables, so frame t goes to rank t mod G, every rank holds a replica of the static tables, and the only
exchange is the final gather of the finished equirect slabs on the sink rank (RCCL send/recv over xGMI;
`gloo` on CPU in the tests).  No collective touches the per-frame compute path."""
import torch
import torch.distributed as dist


def frame_owner(t, world_size):
    """frame index -> rank (round robin, preserves display order per rank)."""
    return t % world_size


def local_frames(first, count, rank, world_size):
    """Global indices of the frames in [first, first+count) this rank stitches."""
    return [t for t in range(first, first + count) if frame_owner(t, world_size) == rank]


def gather_slabs(slab, rank, world_size, dst=0, async_op=False, out=None):
    """Gather one contiguous uint8 tensor per rank on `dst`.  Returns (work, list_on_dst_or_None).
    `slab` is the pano ROI rows of the canvas (contiguous), ~7 MB/frame at config 2."""
    if world_size == 1:
        return None, [slab]
    gl = None
    if rank == dst:
        gl = out if out is not None else [torch.empty_like(slab) for _ in range(world_size)]
    work = dist.gather(slab, gather_list=gl, dst=dst, async_op=async_op)
    return work, gl


def reorder(gathered_steps, world_size):
    """[step][rank][frames_per_rank, ...] -> frames in global display order t = (step*F + j)*G + rank ...
    With round-robin ownership, frame j of rank r in a step of F frames per rank is global index
    step*F*G + j*G + r."""
    order = []
    for s, per_rank in enumerate(gathered_steps):
        F = per_rank[0].shape[0]
        for j in range(F):
            for r in range(world_size):
                order.append((s * F * world_size + j * world_size + r, per_rank[r][j]))
    order.sort(key=lambda kv: kv[0])
    return [v for _, v in order]
