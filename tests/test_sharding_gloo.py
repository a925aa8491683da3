"""The N>1 path of bench.py on CPU: world size 2 over gloo.  Blocks shard across ranks; the codec has no
collective; the movement of BASELINE configs[4] (scatter of the input from rank 0, all-gather of the compressed
sizes, exact-length gather of the packed payloads - bench.data_path, the very function the GPU job runs over RCCL) is
exercised with real bytes: every rank brings its own shard, the root collects the corpus and scatters the distinct shards,
every rank's blocks are compressed by the oracle (the CPU stand-in for the kernels here), gathered, put back in
(rank, block) order from the sizes table and decoded against the shard they came from."""
import ctypes
import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

NB, BS = 6, 20000          # blocks per rank, bytes per block (ragged compressed sizes)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _oracle():
    so = os.path.join(ROOT, "oracle", "liblz4oracle.so")
    if not os.path.exists(so):
        import subprocess
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "liblz4oracle.so"], check=True)
    return ctypes.CDLL(so)


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

    # ---- real bytes through the gather-to-root / scatter / all-gather / exact-length gather path
    orc = _oracle()
    dev = torch.device("cpu")
    # every rank generates its own shard (datagen -s<rank>, different compressibility: ragged sizes); the root collects the corpus
    host = bench.gen_data(NB * BS, 60 - 10 * rank, plan["seed"])
    data = torch.from_numpy(host.copy())
    stride = orc.lz4o_compress_bound(BS)
    comp = torch.zeros((NB, stride), dtype=torch.uint8)
    csizes = []
    for i in range(NB):
        blk = host[i * BS:(i + 1) * BS].tobytes()
        dst = ctypes.create_string_buffer(stride)
        c = orc.lz4o_compress_default(blk, dst, BS, stride)
        assert c > 0
        comp[i, :c] = torch.frombuffer(bytearray(dst.raw[:c]), dtype=torch.uint8)
        csizes.append(c)
    r = bench.data_path(dist, torch, dev, rank, world, data, comp, csizes)
    scattered_ok = bool(torch.equal(r["received"], data))      # the root sent every rank ITS shard, not a copy of its own
    ok_blocks = None
    if rank == 0:
        ok_blocks = 0
        assert len(r["shards"]) == world and not torch.equal(r["shards"][0], r["shards"][1])
        for k, blk in enumerate(r["blocks"]):               # every gathered block decodes to the block of the shard it came from
            cbytes = blk.numpy().tobytes()
            dst = ctypes.create_string_buffer(BS)
            n = orc.lz4o_decompress_safe(cbytes, dst, len(cbytes), BS)
            rr, i = divmod(k, NB)
            if n == BS and dst.raw == r["shards"][rr][i * BS:(i + 1) * BS].numpy().tobytes():
                ok_blocks += 1
    out[rank] = (plan, int(owned.min()), int(owned.max()), t_max, b_sum, scattered_ok, r["sizes"], ok_blocks, csizes)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_movement_and_aggregation():
    world, port = 2, _free_port()
    with mp.Manager() as m:
        out = m.dict()
        mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
        res = dict(out)
    assert set(res) == {0, 1}
    for rank, (plan, omin, omax, t_max, b_sum, scattered_ok, sizes, ok_blocks, csizes) in res.items():
        assert plan["rank"] == rank and plan["world"] == world and plan["seed"] == rank
        assert omin == 1 and omax == 1                     # a partition: every block exactly once
        assert t_max == 2.0                                # max over ranks
        assert b_sum == 3000.0                             # sum over ranks
        assert scattered_ok                                # every rank received its own shard from the root
        assert len(sizes) == world and all(len(s) == NB for s in sizes) and sizes[rank] == csizes   # every rank knows every size
        assert sizes[0] != sizes[1]                        # distinct shards: ragged payloads, exact-length transfers
    assert res[0][7] == world * NB                         # rank 0 put every payload back in order and decoded it


def test_bench_command_line_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it starts two ranks itself (bench.launch_ranks: RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_* per rank, rendezvous on 127.0.0.1) - here in its dry-run mode (no codec, no GPU: the launcher, the
    rendezvous, the shard plan and the scatter / all_gather / exact-length gather over gloo with real bytes), through the
    command line, not through an import."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run", "--blocks", "5", "--block-bytes", "30000"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout                           # rank 0 alone prints
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["dry_run"] is True and d["value"] is None
    assert d["data_path"]["ok"] is True and d["data_path"]["world_size"] == 2
    assert d["data_path"]["scatter_bytes"] == 5 * 30000 and d["data_path"]["gather_bytes"] == 5 * 30000
    assert d["bytes_all_ranks"] == 2 * 5 * 30000
    # one rank: no launcher, no process group
    r1 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-run", "--blocks", "2", "--block-bytes", "4096"],
                        capture_output=True, text=True, timeout=300, env=env)
    assert r1.returncode == 0 and json.loads([l for l in r1.stdout.splitlines() if l.startswith("{")][0])["n_gpus"] == 1
