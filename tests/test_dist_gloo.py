"""The N > 1 path on CPU: two processes, gloo backend, world_size 2 — the sharding of query images and the single
result gather of piccolo_amd.dist (the GPU run uses the same code with RCCL)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from piccolo_amd import dist as pdist


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _fake_refine(k):
    """Stands in for make_input + omniloc_batch: a deterministic row per image index."""
    g = torch.Generator().manual_seed(1000 + k)
    return torch.cat([torch.tensor([float(k)]), torch.rand(pdist.RESULT_WIDTH - 1, generator=g)])


def _worker(rank, world, port, n_items, out_dir):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        assert pdist.world() == (rank, world)
        mine = pdist.shard(n_items)
        assert mine == list(range(rank, n_items, world))
        table = pdist.localize_sharded(n_items, _fake_refine, torch.device("cpu"))
        np.save(os.path.join(out_dir, "table_%d.npy" % rank), table.numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_items", [8, 7, 1])
def test_sharded_localisation_equals_single_process(tmp_path, n_items):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), n_items, str(tmp_path)), nprocs=world, join=True)
    want = torch.stack([_fake_refine(k) for k in range(n_items)]).numpy()
    for r in range(world):
        got = np.load(tmp_path / ("table_%d.npy" % r))
        assert got.shape == (n_items, pdist.RESULT_WIDTH)
        assert np.array_equal(got, want)            # every rank holds the full table, in image order


def test_single_process_path_needs_no_group():
    table = pdist.localize_sharded(5, _fake_refine, torch.device("cpu"))
    assert np.array_equal(table.numpy(), torch.stack([_fake_refine(k) for k in range(5)]).numpy())
    assert pdist.shard(10, 3, 4) == [3, 7] and pdist.shard(2, 3, 4) == []
