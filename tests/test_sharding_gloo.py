"""The N>1 path of bench.py on CPU: world size 2 over gloo.  Blocks shard across ranks with no
data-path collective; the only collectives are the max-over-ranks time and the sum of bytes."""
import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    import bench
    plan = bench.shard_plan(256, rank, world)
    # every rank brings its own blocks (weak scaling); ranks never own the same global block
    owned = torch.zeros(world * 256, dtype=torch.int64)
    owned[plan["first_block"]:plan["first_block"] + plan["n_blocks"]] = 1
    dist.all_reduce(owned)
    t_max, b_sum = bench.aggregate(dist, 1.0 + rank, 1000 * (rank + 1))
    out[rank] = (plan, int(owned.min()), int(owned.max()), t_max, b_sum)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_and_aggregation():
    world, port = 2, _free_port()
    with mp.Manager() as m:
        out = m.dict()
        mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
        res = dict(out)
    assert set(res) == {0, 1}
    for rank, (plan, omin, omax, t_max, b_sum) in res.items():
        assert plan["rank"] == rank and plan["world"] == world and plan["seed"] == rank
        assert omin == 1 and omax == 1                     # a partition: every block exactly once
        assert t_max == 2.0                                # max over ranks
        assert b_sum == 3000.0                             # sum over ranks
