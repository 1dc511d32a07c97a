"""N>1 path on CPU: two gloo ranks shard images round-robin and all-gather ragged metric rows."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from multi_view_stereonet_amd import distributed as mdist


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, num_images, skip, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    r, w, _ = mdist.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    mine = [i for i in mdist.shard_indices(num_images, r, w) if i not in skip]      # ragged: some images skipped
    rows = torch.tensor([[float(i), float(i) ** 2, 1.0 / (i + 1)] for i in mine], dtype=torch.float32).reshape(-1, 3)
    idx = torch.tensor(mine, dtype=torch.int64)
    all_rows, all_idx = mdist.gather_metric_rows(rows, idx)
    ret[rank] = (all_rows.clone(), all_idx.clone(), mdist.average_rows(all_rows).clone())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("num_images,skip", [(7, ()), (5, (1, 3)), (1, ())])
def test_gather_metric_rows_gloo_world2(num_images, skip):
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, num_images, set(skip), ret), nprocs=world, join=True)
    kept = [i for i in range(num_images) if i not in skip]
    want = torch.tensor([[float(i), float(i) ** 2, 1.0 / (i + 1)] for i in kept]).reshape(-1, 3)
    for rank in range(world):
        rows, idx, avg = ret[rank]
        assert idx.tolist() == kept
        assert torch.allclose(rows, want)
        assert torch.allclose(avg, want.double().mean(0))


def test_single_process_is_identity():
    rows = torch.tensor([[3.0, 1.0], [1.0, 2.0]])
    idx = torch.tensor([5, 2])
    r, i = mdist.gather_metric_rows(rows, idx)
    assert i.tolist() == [2, 5] and r.tolist() == [[1.0, 2.0], [3.0, 1.0]]
    assert mdist.shard_indices(10, 1, 4) == [1, 5, 9]


def _single_rank_worker(rank, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    r, w, _ = mdist.init_from_env(backend="gloo", force=True)
    assert (r, w) == (0, 1) and dist.is_initialized() and dist.get_world_size() == 1
    calls = []
    real = dist.all_gather
    dist.all_gather = lambda out, x, *a, **k: (calls.append(tuple(x.shape)), real(out, x, *a, **k))[1]
    try:
        rows = torch.tensor([[3.0, 1.0], [1.0, 2.0], [0.5, 4.0]])
        all_rows, all_idx = mdist.gather_metric_rows(rows, torch.tensor([5, 2, 9]))
    finally:
        dist.all_gather = real
    ret["rows"], ret["idx"], ret["calls"] = all_rows.clone(), all_idx.clone(), calls
    dist.destroy_process_group()


def test_forced_single_rank_group_goes_through_the_collective():
    """A world-size-1 group (what `MVSN_BENCH_BACKEND=nccl python bench.py --gpus 1` creates to load RCCL on a one-GPU
    box) must not take the no-group shortcut: counts and the (rows + index) float64 payload go through all_gather."""
    ret = mp.Manager().dict()
    mp.spawn(_single_rank_worker, args=(_free_port(), ret), nprocs=1, join=True)
    assert ret["idx"].tolist() == [2, 5, 9]
    assert ret["rows"].tolist() == [[1.0, 2.0], [3.0, 1.0], [0.5, 4.0]]
    assert ret["calls"] == [(1,), (3, 3)]          # the count, then 3 rows x (2 columns + the index column)
