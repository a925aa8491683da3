#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on BASELINE.json's config, one process per GPU.

    python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): per GPU, 1 GiB of independent 4 MiB blocks of
`datagen -P60 -s<rank>` data, resident in HBM.  A "step" is one pass of the hot path over that
batch: LZ4 block compression of every block (LZ4_compress_default semantics) followed by
decompression of every block (LZ4_decompress_safe semantics), through the C ABI of
include/lz4amd.h.  GB/s counts UNCOMPRESSED bytes per second (programs/bench.c:500-503,
564-568).  Blocks shard across ranks with no data-path collective (weak scaling).

Prints ONE JSON line on rank 0:
  value            whole-job round-trip throughput: ranks * 1 GiB * K / max-over-ranks time
  compress_GBps / decompress_GBps   the two halves, from HIP events around their kernels
  roofline         dominant kernel: algorithmic bytes (U + C per block) / HIP-event time
  kernels          the same for every kernel of the step
  cpu_baseline     the reference lib/lz4.c (oracle/_ref, kind "reference") or the oracle port,
                   timed on this box's host cores on a bounded sample of the same workload
  hc               (N=1) BASELINE configs[3] beside it: LZ4_compress_HC level 9 on 256 KiB blocks of the
                   same GiB, with its own roofline and the reference lz4hc.c on the host cores
"""
import argparse
import ctypes
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 achievable


def shard_plan(total_blocks_per_rank, rank, world):
    """Which blocks does `rank` own?  Independent blocks: contiguous ranges, weak scaling
    (every rank brings its own `total_blocks_per_rank`); the global block id only seeds datagen."""
    first = rank * total_blocks_per_rank
    return {"rank": rank, "world": world, "first_block": first, "n_blocks": total_blocks_per_rank,
            "seed": rank}


def aggregate(dist, local_seconds, local_bytes, device=None):
    """max-over-ranks time and sum-over-ranks bytes (the only collectives of the bench)."""
    import torch
    t = torch.tensor([local_seconds], dtype=torch.float64, device=device)
    b = torch.tensor([float(local_bytes)], dtype=torch.float64, device=device)
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(b, op=dist.ReduceOp.SUM)
    return float(t.item()), float(b.item())


def gen_data(nbytes, pct, seed):
    so = os.path.join(ROOT, "tools", "libdatagen.so")
    if not os.path.exists(so):
        subprocess.run(["gcc", "-O2", "-shared", "-fPIC", "-o", so, os.path.join(ROOT, "tools", "datagen.c")], check=True)
    L = ctypes.CDLL(so)
    L.lz4amd_datagen.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_double, ctypes.c_double, ctypes.c_uint32]
    import numpy as np
    buf = np.empty(nbytes, dtype=np.uint8)
    assert L.lz4amd_datagen(buf.ctypes.data, nbytes, pct / 100.0, 0.0, seed) == 0
    return buf


def cpu_baseline(n_blocks, block_bytes, pct, seed):
    """Reference (or oracle port) on the host cores, bounded sample, best of 3 (bench.c style)."""
    cores = os.cpu_count() or 1
    exe, kind = os.path.join(ROOT, "oracle", "_ref", "refbench"), "reference"
    if not os.path.exists(exe):
        exe, kind = os.path.join(ROOT, "oracle", "oraclebench"), "port"
        if not os.path.exists(exe):
            try:
                subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "oraclebench"], check=True,
                               stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            except Exception:
                return None
    sample_blocks = min(n_blocks, 64)                       # 256 MiB of the same stream
    out = {}
    try:
        for threads in (1, cores):
            r = subprocess.run([exe, str(threads), str(sample_blocks), str(block_bytes), str(pct), str(seed), "3"],
                               capture_output=True, text=True, check=True, timeout=600)
            out[threads] = json.loads(r.stdout)
    except Exception as e:                                   # never let the baseline kill the bench
        return {"error": str(e)}
    full = out[cores]
    return {"value": round(full["roundtrip_GBps"], 3), "unit": "GB/s", "cores": cores, "kind": kind,
            "sample": f"{sample_blocks} x {block_bytes} B blocks of datagen -P{pct} -s{seed} (first "
                      f"{sample_blocks * block_bytes >> 20} MiB of rank 0's shard), best of 3, static partition over {cores} threads",
            "compress_GBps": round(full["compress_GBps"], 3), "decompress_GBps": round(full["decompress_GBps"], 3),
            "single_thread": {"compress_GBps": round(out[1]["compress_GBps"], 3),
                              "decompress_GBps": round(out[1]["decompress_GBps"], 3),
                              "roundtrip_GBps": round(out[1]["roundtrip_GBps"], 3)},
            "ref_comp_bytes": full["comp_bytes"], "ref_src_bytes": full["src_bytes"]}


def cpu_baseline_hc(block_bytes, pct, seed, level):
    """Reference LZ4_compress_HC (oracle/_ref only: the oracle port has no HC) on the host cores,
    bounded sample of configs[3]'s table."""
    cores = os.cpu_count() or 1
    exe = os.path.join(ROOT, "oracle", "_ref", "refbench")
    if not os.path.exists(exe):
        return None
    sample_blocks = 2 * cores if cores >= 64 else 128        # ~0.1 s of one-thread work per thread
    try:
        one = json.loads(subprocess.run([exe, "1", "16", str(block_bytes), str(pct), str(seed), "1", str(level)],
                                        capture_output=True, text=True, check=True, timeout=300).stdout)
        full = json.loads(subprocess.run([exe, str(cores), str(sample_blocks), str(block_bytes), str(pct), str(seed), "3", str(level)],
                                         capture_output=True, text=True, check=True, timeout=600).stdout)
    except Exception as e:
        return {"error": str(e)}
    return {"value": round(full["compress_GBps"], 3), "unit": "GB/s", "cores": cores, "kind": "reference",
            "sample": f"{sample_blocks} x {block_bytes} B blocks of datagen -P{pct} -s{seed}, LZ4_compress_HC level {level}, best of 3, "
                      f"static partition over {cores} threads",
            "single_thread_GBps": round(one["compress_GBps"], 4),
            "ref_comp_bytes": full["comp_bytes"], "ref_src_bytes": full["src_bytes"], "sample_blocks": sample_blocks}


def bench_hc(ctx, lz4_amd, torch, data, out, stream, pct, seed, level=9, bs=256 << 10, with_cpu=True):
    """BASELINE configs[3]: LZ4_compress_HC level 9 on 256 KiB blocks of the same GiB, device resident.
    Not part of `value`; reported next to it with its own roofline and CPU baseline."""
    U = data.numel()
    nb = U // bs
    stride = (lz4_amd.compress_bound(bs) + 255) & ~255
    comp = torch.empty((nb, stride), dtype=torch.uint8, device=data.device)
    tab = lz4_amd.BlockTable([data.data_ptr() + i * bs for i in range(nb)], [bs] * nb,
                             [comp.data_ptr() + i * stride for i in range(nb)], [stride] * nb)
    plan = lz4_amd.Plan(ctx, lz4_amd.OP_COMPRESS_HC, tab, level=level)
    plan.launch(stream)
    cs = plan.results(stream)
    assert all(c > 0 for c in cs), "HC compression failed"
    dtab = lz4_amd.BlockTable([comp.data_ptr() + i * stride for i in range(nb)], cs,
                              [out.data_ptr() + i * bs for i in range(nb)], [bs] * nb)
    dplan = lz4_amd.Plan(ctx, lz4_amd.OP_DECOMPRESS, dtab)
    out.zero_()
    dplan.launch(stream)
    assert dplan.results(stream) == [bs] * nb and torch.equal(out, data), "HC round trip is not bit exact"
    ms = min(plan.launch_timed(stream)[0][0] for _ in range(3))
    dms = min(dplan.launch_timed(stream)[0][0] for _ in range(3))
    C = sum(cs)
    res = {"workload": "configs[3]: %d independent %d-byte blocks (%.2f GiB), datagen -P%d, LZ4_compress_HC level %d, device resident"
                       % (nb, bs, U / 2**30, pct, level),
           "compress_GBps": round(U / (ms * 1e-3) / 1e9, 2), "kernel_ms": round(ms, 3),
           "decompress_GBps": round(U / (dms * 1e-3) / 1e9, 2), "ratio": round(U / C, 4), "compressed_bytes": C,
           "roofline": {"kernel": "compress_hc", "bound": "hbm", "achieved": round((U + C) / (ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBPS,
                        "unit": "GB/s", "frac": round((U + C) / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 5), "traffic": None,
                        "algorithmic_bytes_per_launch": U + C, "avg_ms": round(ms, 3)}}
    try:                                                     # HBM bytes per launch from the committed PMC passes
        pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")))
        if nb == 4096 and bs == 256 << 10 and pct == 60 and level == 9:
            res["roofline"]["traffic"] = pmc["compress_hc"]["hbm_bytes_per_launch"]
    except Exception:
        pass
    if with_cpu:
        cb = cpu_baseline_hc(bs, pct, seed, level)
        res["cpu_baseline"] = cb
        if cb and "ref_comp_bytes" in cb:
            res["ratio_vs_reference"] = round(cb["ref_comp_bytes"] / sum(cs[:cb["sample_blocks"]]), 4)
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--blocks", type=int, default=256, help="blocks per GPU (256 x 4 MiB = 1 GiB)")
    ap.add_argument("--block-bytes", type=int, default=4 << 20)
    ap.add_argument("--pct", type=int, default=60, help="datagen -P compressibility")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-hc", action="store_true", help="skip the LZ4_compress_HC (configs[3]) side measurement")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import lz4_amd

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", local_rank if world > 1 else 0)

    ctx = lz4_amd.Context(dev.index)                     # raises loudly without the HIP library / GPU
    plan_s = shard_plan(args.blocks, rank, world)
    bs, nb = args.block_bytes, plan_s["n_blocks"]
    U = nb * bs

    host = gen_data(U, args.pct, plan_s["seed"])
    data = torch.from_numpy(host).to(dev)
    stream = torch.cuda.current_stream().cuda_stream

    # ---- block tables (built once, like the reference bench's blockParam_t table)
    stride = (lz4_amd.compress_bound(bs) + 255) & ~255
    comp = torch.empty((nb, stride), dtype=torch.uint8, device=dev)
    out = torch.empty(U, dtype=torch.uint8, device=dev)
    ctab = lz4_amd.BlockTable([data.data_ptr() + i * bs for i in range(nb)], [bs] * nb,
                              [comp.data_ptr() + i * stride for i in range(nb)], [stride] * nb)
    cplan = lz4_amd.Plan(ctx, lz4_amd.OP_COMPRESS, ctab)
    cplan.launch(stream)
    csizes = cplan.results(stream)
    assert all(c > 0 for c in csizes), "compression failed"
    C = sum(csizes)
    dtab = lz4_amd.BlockTable([comp.data_ptr() + i * stride for i in range(nb)], csizes,
                              [out.data_ptr() + i * bs for i in range(nb)], [bs] * nb)
    dplan = lz4_amd.Plan(ctx, lz4_amd.OP_DECOMPRESS, dtab)
    dplan.launch(stream)
    dres = dplan.results(stream)
    assert dres == [bs] * nb, "decompression failed"
    assert torch.equal(out, data), "round trip is not bit exact"

    def step():
        cplan.launch(stream)
        dplan.launch(stream)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    t_max, bytes_all = aggregate(dist if world > 1 else None, elapsed, U * args.steps, device=dev)

    # ---- per-kernel durations, HIP events on the launch stream, same K steps
    k_ms = {"compress": 0.0, "decompress": 0.0}
    c_total = d_total = 0.0
    for _ in range(args.steps):
        km, tot = cplan.launch_timed(stream)
        k_ms["compress"] += km[0]
        c_total += tot
        km, tot = dplan.launch_timed(stream)
        k_ms["decompress"] += km[0]
        d_total += tot
    for k in k_ms:
        k_ms[k] /= args.steps
    c_total /= args.steps
    d_total /= args.steps
    assert torch.equal(out, data), "round trip is not bit exact after the timed loop"

    # ---- not part of the step: the batched XXH32 kernel (frame block checksums) over the same blocks
    xplan = lz4_amd.Plan(ctx, lz4_amd.OP_XXH32, lz4_amd.BlockTable([data.data_ptr() + i * bs for i in range(nb)], [bs] * nb, [0] * nb, [0] * nb))
    xplan.launch(stream); xplan.results(stream)
    x_ms = sum(xplan.launch_timed(stream)[1] for _ in range(3)) / 3

    if rank == 0:
        alg = {"compress": U + C, "decompress": U + C}        # U read + C written / C read + U written
        kernels = []
        for name, ms in k_ms.items():
            gbps = alg[name] / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
            kernels.append({"kernel": name, "avg_ms": round(ms, 4), "algorithmic_bytes": alg[name],
                            "GBps": round(gbps, 1), "frac_of_hbm_peak": round(gbps / HBM_PEAK_GBPS, 4)})
        # HBM bytes per launch from the committed PMC passes (rocprofv3 cannot run inside this process)
        traffic = {}
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")))
            if nb == 256 and bs == 4 << 20 and args.pct == 60:
                traffic = {k: pmc[k]["hbm_bytes_per_launch"] for k in ("compress", "decompress")}
        except Exception:
            pass
        dom = max(kernels, key=lambda k: k["avg_ms"])
        dec = next(k for k in kernels if k["kernel"] == "decompress")
        result = {
            "metric": "GB/s compress + decompress, 4 MB independent blocks (uncompressed bytes through one compress+decompress pass per second)",
            "value": round(bytes_all / t_max / 1e9, 3), "unit": "GB/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(t_max / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic (datagen -P%d restated in tools/datagen.c, md5-pinned to the reference tool)" % args.pct,
            "config": {"workload": "configs[1]: %d independent %d-byte blocks per GPU (%.2f GiB), datagen -P%d -s<rank>, block compress + decompress, device resident"
                                   % (nb, bs, U / 2**30, args.pct),
                       "blocks_per_gpu": nb, "block_bytes": bs, "parallelism": "blocks sharded over %d GPU(s), no data-path collective" % world},
            "compress_GBps": round(U / (c_total * 1e-3) / 1e9, 2),
            "decompress_GBps": round(U / (d_total * 1e-3) / 1e9, 2),
            "ratio": round(U / C, 4), "compressed_bytes": C,
            "roofline": {"kernel": dom["kernel"], "bound": "hbm", "achieved": dom["GBps"], "peak": HBM_PEAK_GBPS,
                         "unit": "GB/s", "frac": dom["frac_of_hbm_peak"], "traffic": traffic.get(dom["kernel"]),
                         "algorithmic_bytes_per_launch": dom["algorithmic_bytes"], "avg_ms": dom["avg_ms"]},
            "roofline_decompress": {"bound": "hbm", "achieved": dec["GBps"], "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                    "frac": dec["frac_of_hbm_peak"], "traffic": traffic.get("decompress"),
                                    "algorithmic_bytes_per_launch": dec["algorithmic_bytes"], "avg_ms": dec["avg_ms"]},
            "kernels": kernels,
            "extras": {"xxh32_batch_GBps": round(U / (x_ms * 1e-3) / 1e9, 1), "xxh32_batch_ms": round(x_ms, 3),
                       "note": "XXH32 (seed 0) of every 4 MiB block, one wave per block; not in `value`"},
        }
        if world == 1 and not args.no_hc and U % (256 << 10) == 0:
            try:
                result["hc"] = bench_hc(ctx, lz4_amd, torch, data, out, stream, args.pct, plan_s["seed"],
                                        with_cpu=not args.no_cpu_baseline)
            except Exception as e:                           # the side measurement never kills the bench line
                result["hc"] = {"error": str(e)}
        if world == 1 and not args.no_cpu_baseline:
            cb = cpu_baseline(nb, bs, args.pct, plan_s["seed"])
            result["cpu_baseline"] = cb
            if cb and "ref_comp_bytes" in cb:
                ours = sum(csizes[:min(nb, 64)])
                result["ratio_vs_reference"] = round(cb["ref_comp_bytes"] / ours, 4)
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
