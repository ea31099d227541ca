"""Multi-GPU scaling of the localisation path: independent query images shard round-robin over the ranks
(the reference's `for trial, filename in enumerate(filenames)` loop, localize.py:143,357, is the natural shard point);
the cloud is replicated, every panorama lives only on its rank, and the ONLY collective is one all_gather of the
per-image result rows at the end (RCCL on GPUs, gloo in the CPU tests).  No data-path collective exists: candidates,
points and iterations of one image never leave their GPU.

One process per GPU (torch.distributed.run); works unchanged with world size 1 and without an initialised group.
"""
import torch

RESULT_WIDTH = 16      # t(3), R(9), loss, t_err, r_err, seconds


def world():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard(n_items, rank=None, world_size=None):
    """Indices of the items rank `rank` owns: item k -> rank k mod world (round-robin)."""
    if rank is None:
        rank, world_size = world()
    return list(range(rank, n_items, world_size))


def gather_rows(local_rows, n_items, rank=None, world_size=None, group=None):
    """All ranks' result rows in item order.

    local_rows: (len(shard(n_items)), RESULT_WIDTH) tensor, row j = item rank + j * world.  Returns (n_items, W) on the
    same device, identical on every rank.  One all_gather_into_tensor of a padded (ceil(n/world), W) block per rank —
    a few KB, latency-bound; xGMI bandwidth is irrelevant here."""
    if rank is None:
        rank, world_size = world()
    width = local_rows.shape[1]
    if world_size == 1:
        assert local_rows.shape[0] == n_items
        return local_rows
    import torch.distributed as dist
    per_rank = (n_items + world_size - 1) // world_size
    # RCCL gathers device tensors in place; gloo (CPU tests, or several ranks sharing one GPU: PCL_DIST_BACKEND=gloo) goes
    # through host memory
    home = local_rows.device
    via_host = dist.get_backend(group) == "gloo" and local_rows.is_cuda
    dev = torch.device("cpu") if via_host else home
    block = torch.full((per_rank, width), float("nan"), dtype=local_rows.dtype, device=dev)
    block[: local_rows.shape[0]] = local_rows.to(dev)
    out = torch.empty(world_size * per_rank, width, dtype=local_rows.dtype, device=dev)
    dist.all_gather_into_tensor(out, block, group=group)
    # out[r * per_rank + j] is item r + j * world  ->  item-major order
    ordered = out.reshape(world_size, per_rank, width).transpose(0, 1).reshape(per_rank * world_size, width)
    return ordered[:n_items].contiguous().to(home)


def localize_sharded(n_items, refine, device, group=None):
    """Run `refine(item_index) -> (RESULT_WIDTH,) tensor` on this rank's items and gather everything.

    `refine` is the per-image body (make_input + omniloc/omniloc_batch + error metrics); it must leave its result on
    `device`.  Returns the (n_items, RESULT_WIDTH) table on every rank."""
    rank, world_size = world()
    mine = shard(n_items, rank, world_size)
    rows = torch.empty(len(mine), RESULT_WIDTH, dtype=torch.float32, device=device)
    for j, k in enumerate(mine):
        rows[j] = refine(k).to(device=device, dtype=torch.float32).reshape(RESULT_WIDTH)
    return gather_rows(rows, n_items, rank, world_size, group)
