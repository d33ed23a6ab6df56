"""Prompt-parallel replicas over torch.distributed: world_size 2 on CPU with gloo."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from stabletriton_amd import parallel, synth
from stabletriton_amd.unet import TINY, UNet2DConditionModel


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    r, w, _ = parallel.init_from_env("gloo")
    torch.manual_seed(100 + rank)                       # replicas start different ...
    model = UNet2DConditionModel(TINY)
    if r == 0:
        synth.fill_module_(model, 0)                    # ... rank 0 owns the weights
    n = parallel.broadcast_module(model, src=0, bucket_bytes=1 << 20)     # small buckets -> several collectives
    ref = UNet2DConditionModel(TINY)
    synth.fill_module_(ref, 0)
    same = all(torch.equal(a, b) for a, b in zip(model.state_dict().values(), ref.state_dict().values()))
    mine = parallel.shard_prompts(7, r, w)
    slowest = parallel.max_over_ranks(1.0 + r, torch.device("cpu"))
    parallel.barrier()
    q.put((r, same, n, mine, slowest))
    dist.destroy_process_group()


def test_weight_broadcast_and_prompt_sharding_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert all(same for _, same, _, _, _ in res)
    assert res[0][2] == res[1][2] and res[0][2] > 1           # same number of bucketed broadcasts on both ranks
    assert res[0][3] == [0, 2, 4, 6] and res[1][3] == [1, 3, 5]  # every prompt owned exactly once
    assert res[0][4] == res[1][4] == 2.0                      # max over ranks


def test_single_process_is_a_no_op():
    m = UNet2DConditionModel(TINY)
    assert parallel.broadcast_module(m) == 0
    assert parallel.shard_prompts(3, 0, 1) == [0, 1, 2]
    assert parallel.max_over_ranks(3.5, torch.device("cpu")) == 3.5
