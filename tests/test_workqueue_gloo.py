"""CPU, world_size 2, gloo: the shared-counter work queue (driver.WorkQueue, SURVEY.md 8e "dynamic chunking") and the
variable-length all-gather of the mask-size list (Generate Dataset/statistic.py:34-53)."""
import os
import socket
import time

import numpy as np
import torch.distributed as dist
import torch.multiprocessing as mp

from samrs_amd import driver

N_ITEMS, CHUNK = 53, 4


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    wq = driver.WorkQueue(N_ITEMS, chunk=CHUNK, rank=rank, world=world, mode="dynamic")
    mine = []
    for s, e in wq:
        mine.append((s, e))
        time.sleep(0.02 if rank == 0 else 0.002)        # rank 0 is the "slow GPU": it must end up with fewer chunks
    sizes = [100 * rank + k for k in range(3 + 4 * rank)]   # ragged: 3 entries on rank 0, 7 on rank 1
    gathered = driver.gather_mask_sizes(sizes)
    # generate --resume: the ranks may SEE different todo lists (rank 1 lists the output directory later, after rank 0 wrote
    # two more images); everybody must work from rank 0's list (driver.agree_on_list)
    seen = ["a", "b", "c", "d"] if rank == 0 else ["c", "d"]
    agreed = driver.agree_on_list(seen if rank == 0 else [])
    out[rank] = (mine, gathered, agreed)
    dist.barrier()
    dist.destroy_process_group()


def test_dynamic_queue_covers_everything_once_and_balances():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    r0, r1 = out[0][0], out[1][0]
    covered = sorted(i for s, e in r0 + r1 for i in range(s, e))
    assert covered == list(range(N_ITEMS)), "every index exactly once across the ranks"
    assert all(e - s == CHUNK or e == N_ITEMS for s, e in r0 + r1)
    assert len(r1) > len(r0), f"the faster rank should have pulled more chunks ({len(r0)} vs {len(r1)})"
    expect = [0, 1, 2] + [100 + k for k in range(7)]
    assert out[0][1] == expect and out[1][1] == expect
    assert out[0][2] == out[1][2] == ["a", "b", "c", "d"]


def test_static_queue_is_the_strided_shard():
    for world in (1, 2, 3):
        seen = []
        for r in range(world):
            got = [i for s, e in driver.WorkQueue(10, chunk=1, rank=r, world=world) for i in range(s, e)]
            assert got == driver.shard(list(range(10)), r, world)
            seen += got
        assert sorted(seen) == list(range(10))
    # chunked static queue: rank r owns chunks r, r + world, ...
    assert list(driver.WorkQueue(20, chunk=8, rank=1, world=2)) == [(8, 16)]
    assert list(driver.WorkQueue(20, chunk=8, rank=0, world=2)) == [(0, 8), (16, 20)]
    # single-process dynamic queue degenerates to a local counter
    assert list(driver.WorkQueue(5, chunk=2, mode="dynamic")) == [(0, 2), (2, 4), (4, 5)]


def test_gather_mask_sizes_single_process():
    assert driver.gather_mask_sizes([5, 0, 7]) == [5, 0, 7]
    assert driver.agree_on_list(["x", "y"]) == ["x", "y"]                  # no process group: the caller's own list
