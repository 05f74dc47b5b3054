"""CPU, world_size 2, gloo: the N>1 path -- static image sharding with no data-path collective and
the single statistics all-reduce (Generate Dataset/statistic.py:15-21) -- equals the serial result."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from samrs_amd import driver
from oracle import sam_oracle as so

N_CLASSES, N_IMAGES = 18, 7


def fake_image_masks(i):
    """Deterministic stand-in for one image's masks / labels (the GPU is not available here)."""
    rng = np.random.default_rng(500 + i)
    n = int(rng.integers(1, 6))
    masks = rng.random((n, 32, 32)) > rng.uniform(0.3, 1.0, (n, 1, 1))   # some masks are empty
    if i == 3:
        masks[0] = False
    labels = rng.integers(0, N_CLASSES, n)
    return masks, labels


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pix = torch.zeros(N_CLASSES, dtype=torch.int64)
    ins = torch.zeros(N_CLASSES, dtype=torch.int64)
    mine = driver.shard(list(range(N_IMAGES)), rank, world)
    for i in mine:
        masks, labels = fake_image_masks(i)
        _, areas = so.paint_semantic(masks, labels, (32, 32))
        p, q = so.class_statistics(areas, labels, N_CLASSES)
        pix += torch.from_numpy(p)
        ins += torch.from_numpy(q)
    tp, ti = driver.reduce_statistics(pix, ins)
    out[rank] = (mine, tp.numpy().copy(), ti.numpy().copy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_statistics_equal_serial():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    pix = np.zeros(N_CLASSES, np.int64)
    ins = np.zeros(N_CLASSES, np.int64)
    for i in range(N_IMAGES):
        masks, labels = fake_image_masks(i)
        _, areas = so.paint_semantic(masks, labels, (32, 32))
        p, q = so.class_statistics(areas, labels, N_CLASSES)
        pix += p
        ins += q
    assert sorted(out[0][0] + out[1][0]) == list(range(N_IMAGES))       # shards partition the images
    for r in (0, 1):
        assert np.array_equal(out[r][1], pix) and np.array_equal(out[r][2], ins)


def test_single_process_reduce_is_identity():
    a = torch.arange(18, dtype=torch.int64)
    b = torch.ones(18, dtype=torch.int64)
    x, y = driver.reduce_statistics(a, b)
    assert torch.equal(x, a) and torch.equal(y, b)
