#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on BASELINE.json's config, one process per GPU.

    python bench.py --gpus N --steps K --warmup W

With N > 1 and no launcher around it (no WORLD_SIZE in the environment) this process starts the N ranks itself (one per
GPU, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, rendezvous on 127.0.0.1) and waits for them; under torchrun it is a rank.

Workload (BASELINE.json configs[1]): per GPU, 1 GiB of independent 4 MiB blocks of
`datagen -P60 -s<rank>` data, resident in HBM.  A "step" is one pass of the hot path over that
batch: LZ4 block compression of every block (LZ4_compress_default semantics) followed by
decompression of every block (LZ4_decompress_safe semantics), through the C ABI of
include/lz4amd.h.  GB/s counts UNCOMPRESSED bytes per second (programs/bench.c:500-503,
564-568).  Blocks shard across ranks; the codec itself has no collective (weak scaling).

Prints ONE JSON line on rank 0:
  value            whole-job round-trip throughput: ranks * 1 GiB * K / max-over-ranks time
  compress_GBps / decompress_GBps   the two halves, from HIP events around their kernels
  roofline         dominant kernel: algorithmic bytes (U + C per block) / HIP-event time; `peak` is the
                   8 TB/s spec, `frac_of_measured_copy` the same rate against this box's own stream-copy kernel
  kernels          the same for every kernel of the step
  cpu_baseline     the reference lib/lz4.c (oracle/_ref, kind "reference") or the oracle port, timed on this box's
                   host cores (1 thread, one per physical core, one per logical CPU; >= 4 blocks per thread,
                   loops of >= 1 s, best of 3) on a bounded sample of the same workload
  hc               (N=1) BASELINE configs[3]: LZ4_compress_HC level 9 on 256 KiB blocks of the same GiB (+ levels 12 and 2)
  shape_2048       (N=1) the per-GPU shape of configs[4]: 2048 x 4 MiB blocks (8 GiB), blocks queue 8 deep per CU
  frame            (N=1) configs[2]: the GiB as ONE frame, 4 MB linked blocks + content checksum, through the
                   host-pointer API LZ4F_compressFrame / LZ4F_decompress (PCIe inclusive)
  data_path        (N>1) configs[4]'s movement over RCCL: scatter of the input from rank 0, all-gather of the
                   compressed sizes, gather of the payloads; reported beside the kernel-only `value`
"""
import argparse
import ctypes
import hashlib
import json
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md); a float4 copy reaches ~6300


def shard_plan(total_blocks_per_rank, rank, world):
    """Which blocks does `rank` own?  Independent blocks: contiguous ranges, weak scaling
    (every rank brings its own `total_blocks_per_rank`); the global block id only seeds datagen."""
    first = rank * total_blocks_per_rank
    return {"rank": rank, "world": world, "first_block": first, "n_blocks": total_blocks_per_rank,
            "seed": rank}


def aggregate(dist, local_seconds, local_bytes, device=None):
    """max-over-ranks time and sum-over-ranks bytes."""
    import torch
    t = torch.tensor([local_seconds], dtype=torch.float64, device=device)
    b = torch.tensor([float(local_bytes)], dtype=torch.float64, device=device)
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(b, op=dist.ReduceOp.SUM)
    return float(t.item()), float(b.item())


def gen_data_seeds(nbytes, pct, seeds):
    """nbytes of datagen output as len(seeds) equal streams behind one another (stream k: `datagen -s<seeds[k]>`), generated on as
    many threads (the generator is one serial PRNG chain per stream: ~0.5 GB/s per core; ctypes releases the GIL)."""
    import numpy as np
    k = len(seeds)
    part = nbytes // k
    assert part * k == nbytes
    buf = np.empty(nbytes, dtype=np.uint8)
    L = _datagen_lib()
    errs = []

    def work(i):
        if L.lz4amd_datagen(buf.ctypes.data + i * part, part, pct / 100.0, 0.0, seeds[i]) != 0:
            errs.append(i)
    ts = [threading.Thread(target=work, args=(i,)) for i in range(k)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs
    return buf


def _datagen_lib():
    so = os.path.join(ROOT, "tools", "libdatagen.so")
    if not os.path.exists(so):
        subprocess.run(["gcc", "-O3", "-shared", "-fPIC", "-o", so, os.path.join(ROOT, "tools", "datagen.c")], check=True)
    L = ctypes.CDLL(so)
    L.lz4amd_datagen.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_double, ctypes.c_double, ctypes.c_uint32]
    return L


def gen_data(nbytes, pct, seed):
    so = os.path.join(ROOT, "tools", "libdatagen.so")
    if not os.path.exists(so):
        subprocess.run(["gcc", "-O3", "-shared", "-fPIC", "-o", so, os.path.join(ROOT, "tools", "datagen.c")], check=True)
    L = ctypes.CDLL(so)
    L.lz4amd_datagen.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_double, ctypes.c_double, ctypes.c_uint32]
    import numpy as np
    buf = np.empty(nbytes, dtype=np.uint8)
    assert L.lz4amd_datagen(buf.ctypes.data, nbytes, pct / 100.0, 0.0, seed) == 0
    return buf


KERNEL_FILES = {   # the kernel headers each kernel is built from (lz4amd_device.hip instantiates all of them)
    "compress": ["lz4_common.h", "platform_hip.h", "lz4_compress_kernel.h"],
    "decompress": ["lz4_common.h", "platform_hip.h", "lz4_decompress_kernel.h", "lz4_preparse_kernel.h"],
    "compress_hc": ["lz4_common.h", "platform_hip.h", "lz4_compress_kernel.h", "lz4_hc_kernel.h"],
    "xxh32": ["lz4_common.h", "platform_hip.h", "xxh32_kernel.h"],
}


def kernel_sources_sha(kernel=None):
    """Identity of the device code (of one kernel, or of all of it): the traffic file under profiles/ is only quoted for a kernel
    when it was measured on that kernel's sources.  Comments and blank lines do not count (they do not reach the device)."""
    import re
    h = hashlib.sha256()
    kdir = os.path.join(ROOT, "lz4_amd", "csrc", "kernels")
    names = sorted(os.listdir(kdir)) if kernel is None else KERNEL_FILES[kernel]
    for f in names + ["../lz4amd_device.hip"]:
        with open(os.path.join(kdir, f), "r", encoding="utf-8", errors="replace") as fh:
            text = fh.read()
        text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
        text = re.sub(r"//[^\n]*", "", text)
        lines = [" ".join(l.split()) for l in text.split("\n")]
        h.update("\n".join(l for l in lines if l).encode())
    return h.hexdigest()[:16]


def measured_traffic():
    """HBM bytes per launch from the committed PMC passes (rocprofv3 cannot run inside this process); a kernel's entry is left
    out when the file was measured on other sources of that kernel."""
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
        shas = pmc.get("kernel_sources_sha", {})
        return {k: v for k, v in pmc.items() if k in KERNEL_FILES and shas.get(k) == kernel_sources_sha(k)}
    except Exception:
        return {}


def host_cpus():
    """(physical cores, logical CPUs) of this box."""
    logical = os.cpu_count() or 1
    cores = set()
    try:
        phys = core = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                phys = line.split(":")[1].strip()
            elif line.startswith("core id"):
                core = line.split(":")[1].strip()
            elif not line.strip():
                if phys is not None and core is not None:
                    cores.add((phys, core))
                phys = core = None
    except Exception:
        pass
    return (len(cores) or logical), logical


def _mem_available():
    try:
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable"):
                return int(line.split()[1]) * 1024
    except Exception:
        pass
    return 8 << 30


def _refbench(exe, threads, block_bytes, pct, seed, level=0, min_s=1.0, reps=3, unique=64):
    per_block = 3 * block_bytes + block_bytes // 255 + 16
    per_thread = 4                                           # >= 4 blocks per thread
    while per_thread > 1 and threads * per_thread * per_block > _mem_available() // 3:
        per_thread -= 1
    nblocks = threads * per_thread
    r = subprocess.run([exe, str(threads), str(nblocks), str(block_bytes), str(pct), str(seed), str(reps), str(level),
                        str(min_s), str(min(unique, nblocks))], capture_output=True, text=True, check=True, timeout=900)
    return json.loads(r.stdout)


def cpu_baseline(block_bytes, pct, seed):
    """Reference (or oracle port) on the host cores: persistent threads, every timed pass bracketed by barriers,
    passes repeated for >= 1 s, fastest of 3 such loops (bench.c:466-493)."""
    phys, logical = host_cpus()
    exe, kind = os.path.join(ROOT, "oracle", "_ref", "refbench"), "reference"
    if not os.path.exists(exe):
        exe, kind = os.path.join(ROOT, "oracle", "oraclebench"), "port"
        if not os.path.exists(exe):
            try:
                subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "oraclebench"], check=True,
                               stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            except Exception:
                return None
    out = {}
    try:
        for threads in sorted({1, phys, logical}):
            out[threads] = _refbench(exe, threads, block_bytes, pct, seed)
    except Exception as e:                                   # never let the baseline kill the bench
        return {"error": str(e)}
    best_t = max(out, key=lambda t: out[t]["roundtrip_GBps"])
    full = out[best_t]
    per = {str(t): {"compress_GBps": round(o["compress_GBps"], 3), "decompress_GBps": round(o["decompress_GBps"], 3),
                    "roundtrip_GBps": round(o["roundtrip_GBps"], 3), "blocks": o["blocks"]} for t, o in out.items()}
    return {"value": round(full["roundtrip_GBps"], 3), "unit": "GB/s", "cores": best_t, "kind": kind,
            "physical_cores": phys, "logical_cpus": logical,
            "sample": f"{full['blocks']} x {block_bytes} B blocks (the first {full['unique_blocks']} blocks of datagen -P{pct} -s{seed}, repeated), "
                      f"{full['blocks'] // best_t} per thread, static partition, threads created once, passes repeated >= {full['min_loop_s']:.0f} s, best of 3",
            "compress_GBps": round(full["compress_GBps"], 3), "decompress_GBps": round(full["decompress_GBps"], 3),
            "by_threads": per,
            "ref_comp_bytes_per_unique": full["comp_bytes"] * full["unique_blocks"] // full["blocks"], "unique_blocks": full["unique_blocks"]}


def cpu_baseline_hc(block_bytes, pct, seed, level):
    """Reference LZ4_compress_HC (oracle/_ref only: the oracle port has no HC) on the host cores."""
    phys, logical = host_cpus()
    exe = os.path.join(ROOT, "oracle", "_ref", "refbench")
    if not os.path.exists(exe):
        return None
    try:
        one = _refbench(exe, 1, block_bytes, pct, seed, level=level, min_s=1.0, reps=1, unique=16)
        full = _refbench(exe, logical, block_bytes, pct, seed, level=level, min_s=1.0, reps=3, unique=4 * logical)
    except Exception as e:
        return {"error": str(e)}
    return {"value": round(full["compress_GBps"], 3), "unit": "GB/s", "cores": logical, "kind": "reference",
            "sample": f"{full['blocks']} x {block_bytes} B blocks of datagen -P{pct} -s{seed}, LZ4_compress_HC level {level}, "
                      f"{full['blocks'] // logical} per thread, passes repeated >= 1 s, best of 3",
            "single_thread_GBps": round(one["compress_GBps"], 4),
            "ref_comp_bytes": full["comp_bytes"], "ref_src_bytes": full["src_bytes"], "sample_blocks": full["blocks"]}


def reference_blocks(host, bs, n):
    """The first n blocks of `host` compressed by the reference's LZ4_compress_default (oracle/_ref, test infrastructure used
    here as a data source only) -> (uint8 array [n, stride], sizes); None when the reference is not built."""
    import numpy as np
    so = os.path.join(ROOT, "oracle", "_ref", "liblz4_ref.so")
    if not os.path.exists(so):
        return None
    R = ctypes.CDLL(so)
    R.LZ4_compress_default.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    stride = (bs + bs // 255 + 16 + 255) & ~255
    comp = np.zeros((n, stride), dtype=np.uint8)
    sizes = []
    for i in range(n):
        r = R.LZ4_compress_default(host.ctypes.data + i * bs, comp.ctypes.data + i * stride, bs, stride)
        assert r > 0
        sizes.append(int(r))
    return comp, sizes


def reference_hc_sizes(host, bs, n, level):
    """Per-block sizes of the first n blocks of `host` under the reference's LZ4_compress_HC (oracle/_ref, a checker); None when it is not built."""
    so = os.path.join(ROOT, "oracle", "_ref", "liblz4_ref.so")
    if not os.path.exists(so):
        return None
    R = ctypes.CDLL(so)
    R.LZ4_compress_HC.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    cap = bs + bs // 255 + 16
    dst = ctypes.create_string_buffer(cap)
    return [int(R.LZ4_compress_HC(host.ctypes.data + i * bs, dst, bs, cap, level)) for i in range(n)]


def stream_copy_gbps(ctx, lz4_amd, torch, nbytes, stream):
    """This box's own read+write stream rate: a plain 16-bytes-per-lane copy kernel of the library (2 * bytes / time)."""
    L = lz4_amd.lib()
    L.lz4amd_stream_copy_ms.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int,
                                        ctypes.c_void_p, ctypes.POINTER(ctypes.c_float)]
    a = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    b = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    a.fill_(7)
    ms = ctypes.c_float(0)
    rc = L.lz4amd_stream_copy_ms(ctx._h, b.data_ptr(), a.data_ptr(), nbytes, 4, stream, ctypes.byref(ms))
    if rc != 0 or ms.value <= 0:
        return None
    return 2.0 * nbytes / (ms.value * 1e-3) / 1e9


def roofline_obj(kernel, ms, alg_bytes, copy_gbps, traffic, table_bytes=0):
    """alg_bytes: SURVEY 8(d)'s algorithmic bytes (U + C per launch).  table_bytes: the entry-point tables' rows written / read on top of
    them - real traffic, but not in 8(d)'s figure: `frac` is on U + C alone, `frac_with_table_rows` beside it."""
    ach = alg_bytes / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
    r = {"kernel": kernel, "bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
         "frac": round(ach / HBM_PEAK_GBPS, 5),
         "measured_copy_GBps": round(copy_gbps, 1) if copy_gbps else None,
         "frac_of_measured_copy": round(ach / copy_gbps, 5) if copy_gbps else None,
         "traffic": traffic, "algorithmic_bytes_per_launch": alg_bytes, "avg_ms": round(ms, 4)}
    if table_bytes:
        r["table_row_bytes_per_launch"] = table_bytes
        r["frac_with_table_rows"] = round((alg_bytes + table_bytes) / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 5) if ms > 0 else 0.0
    # what holds the kernel below that roofline today: VALU issue (committed SQ counters of the same kernel sources)
    v = measured_traffic().get(kernel, {}).get("valu")
    if v:
        r["limited_by"] = {"what": "VALU issue: a SIMD takes one wave64 instruction per ~4.2 cycles in mixed code (measured: profiles/r05_valu_issue.txt - "
                                   "VOP3 / DPP / compare / min-max / multiply 4.2-4.6, runs of plain add / logic / shift 2.4, any mix ~4.2)", **v}
    return r


def bench_hc(ctx, lz4_amd, torch, data, out, stream, pct, seed, copy_gbps, level=9, bs=256 << 10, with_cpu=True):
    """BASELINE configs[3]: LZ4_compress_HC level 9 on 256 KiB blocks of the same GiB, device resident.
    Not part of `value`; reported next to it with its own roofline and CPU baseline."""
    U = data.numel()
    nb = U // bs
    stride = (lz4_amd.compress_bound(bs) + 255) & ~255
    comp = torch.empty((nb, stride), dtype=torch.uint8, device=data.device)
    tab = lz4_amd.BlockTable([data.data_ptr() + i * bs for i in range(nb)], [bs] * nb,
                             [comp.data_ptr() + i * stride for i in range(nb)], [stride] * nb)
    plan = lz4_amd.Plan(ctx, lz4_amd.OP_COMPRESS_HC, tab, level=level)
    hints = torch.zeros((nb, lz4_amd.hint_bytes(bs)), dtype=torch.uint8, device=data.device)      # the HC compressor writes the entry-point tables too
    plan.attach_hints(hints.data_ptr(), hints.stride(0))
    plan.launch(stream)
    cs = plan.results(stream)
    assert all(c > 0 for c in cs), "HC compression failed"
    dtab = lz4_amd.BlockTable([comp.data_ptr() + i * stride for i in range(nb)], cs,
                              [out.data_ptr() + i * bs for i in range(nb)], [bs] * nb)
    dplan = lz4_amd.Plan(ctx, lz4_amd.OP_DECOMPRESS, dtab)
    dplan.attach_hints(hints.data_ptr(), hints.stride(0))
    out.zero_()
    dplan.launch(stream)
    assert dplan.results(stream) == [bs] * nb and torch.equal(out, data), "HC round trip is not bit exact"
    fplan = lz4_amd.Plan(ctx, lz4_amd.OP_DECOMPRESS, dtab)      # the same blocks as plain LZ4 blocks
    out.zero_()
    fplan.launch(stream)
    assert fplan.results(stream) == [bs] * nb and torch.equal(out, data), "HC round trip without tables is not bit exact"
    ms = min(plan.launch_timed(stream)[0][0] for _ in range(3))
    dms = min(dplan.launch_timed(stream)[0][0] for _ in range(3))
    fms = min(fplan.launch_timed(stream)[0][0] for _ in range(3))
    C = sum(cs)
    tr = measured_traffic().get("compress_hc", {}).get("hbm_bytes_per_launch") if (nb == 4096 and pct == 60 and level == 9) else None
    res = {"workload": "configs[3]: %d independent %d-byte blocks (%.2f GiB), datagen -P%d, LZ4_compress_HC level %d, device resident"
                       % (nb, bs, U / 2**30, pct, level),
           "compress_GBps": round(U / (ms * 1e-3) / 1e9, 2), "kernel_ms": round(ms, 3),
           "decompress_GBps": round(U / (dms * 1e-3) / 1e9, 2), "decompress_GBps_without_tables": round(U / (fms * 1e-3) / 1e9, 2),
           "tables": {"blocks_decoded_from_their_table": dplan.hint_stats()[0], "tables_rejected": dplan.hint_stats()[1], "bytes_written": table_bytes_written(torch, hints)},
           "ratio": round(U / C, 4), "compressed_bytes": C,
           "roofline": roofline_obj("compress_hc", ms, U + C, copy_gbps, tr)}
    if with_cpu:
        cb = cpu_baseline_hc(bs, pct, seed, level)
        res["cpu_baseline"] = cb
        if cb and "ref_comp_bytes" in cb:
            n_s = cb["sample_blocks"]                        # the CPU sample is the table's first blocks
            if n_s <= len(cs):
                res["ratio_vs_reference"] = round(cb["ref_comp_bytes"] / sum(cs[:n_s]), 4)
    if level == 9:
        # highly repetitive data (copies of copies: datagen -P99): where this search is NOT within the reference's +-3 % (no look-back through later
        # positions' chains, lz4hc.c:906-960) - the gap is part of the line
        try:
            nrep = 1024
            rhost = gen_data(nrep * bs, 99, seed)
            rdata = torch.from_numpy(rhost).to(data.device)
            rtab = lz4_amd.BlockTable([rdata.data_ptr() + i * bs for i in range(nrep)], [bs] * nrep, [comp.data_ptr() + i * stride for i in range(nrep)], [stride] * nrep)
            rplan = lz4_amd.Plan(ctx, lz4_amd.OP_COMPRESS_HC, rtab, level=9)
            rplan.launch(stream)
            rcs = rplan.results(stream)
            rms = min(rplan.launch_timed(stream)[0][0] for _ in range(2))
            rep = {"workload": "%d x %d-byte blocks of datagen -P99, level 9" % (nrep, bs), "ratio": round(nrep * bs / sum(rcs), 4),
                   "compress_GBps": round(nrep * bs / (rms * 1e-3) / 1e9, 2)}
            if with_cpu:
                cbr = cpu_baseline_hc(bs, 99, seed, 9)
                if cbr and "ref_comp_bytes" in cbr and cbr["sample_blocks"] <= nrep:
                    rep["ratio_vs_reference"] = round(cbr["ref_comp_bytes"] / sum(rcs[:cbr["sample_blocks"]]), 4)
                    rep["reference_GBps"] = cbr["value"]
                # block by block (the aggregate hides single blocks: the start of a -P99 stream has few distinct bytes yet): the first 128 blocks
                rs = reference_hc_sizes(rhost, bs, 128, 9)
                if rs and all(r > 0 for r in rs):
                    per = [r / c for r, c in zip(rs, rcs)]                                  # reference bytes / our bytes, per block
                    rep["worst_block_ratio_vs_reference"] = round(min(per), 4)
                    rep["worst_block"] = int(per.index(min(per)))
                    rep["blocks_more_than_5pct_larger"] = int(sum(1 for x in per if x < 1 / 1.05))
                    rep["per_block_sample"] = "the stream's first 128 blocks, reference LZ4_compress_HC level 9 on the host"
            res["repetitive"] = rep
        except Exception as e:
            res["repetitive"] = {"error": str(e)}
        # levels 10-12 (the optimal parse over the same search), same blocks: one timed launch, bit-exact round trip
        try:
            oplan = lz4_amd.Plan(ctx, lz4_amd.OP_COMPRESS_HC, tab, level=12)
            oplan.launch(stream)
            ocs = oplan.results(stream)
            odtab = lz4_amd.BlockTable([comp.data_ptr() + i * stride for i in range(nb)], ocs,
                                       [out.data_ptr() + i * bs for i in range(nb)], [bs] * nb)
            odplan = lz4_amd.Plan(ctx, lz4_amd.OP_DECOMPRESS, odtab)
            out.zero_()
            odplan.launch(stream)
            ok = all(c > 0 for c in ocs) and odplan.results(stream) == [bs] * nb and torch.equal(out, data)
            oms = oplan.launch_timed(stream)[0][0]
            res["optimal_parse_level12"] = {"compress_GBps": round(U / (oms * 1e-3) / 1e9, 2), "kernel_ms": round(oms, 3),
                                            "ratio": round(U / sum(ocs), 4), "bit_exact": bool(ok)}
        except Exception as e:                               # a side measurement: never takes the line down
            res["optimal_parse_level12"] = {"error": str(e)}
        # levels 1-2 (LZ4MID, lz4hc.c:93-95: two tables, one candidate each), same blocks; the reference's own level 2 beside it
        try:
            mplan = lz4_amd.Plan(ctx, lz4_amd.OP_COMPRESS_HC, tab, level=2)
            mplan.launch(stream)
            mcs = mplan.results(stream)
            mdtab = lz4_amd.BlockTable([comp.data_ptr() + i * stride for i in range(nb)], mcs,
                                       [out.data_ptr() + i * bs for i in range(nb)], [bs] * nb)
            mdplan = lz4_amd.Plan(ctx, lz4_amd.OP_DECOMPRESS, mdtab)
            out.zero_()
            mdplan.launch(stream)
            ok = all(c > 0 for c in mcs) and mdplan.results(stream) == [bs] * nb and torch.equal(out, data)
            mms = min(mplan.launch_timed(stream)[0][0] for _ in range(2))
            mid = {"compress_GBps": round(U / (mms * 1e-3) / 1e9, 2), "kernel_ms": round(mms, 3), "ratio": round(U / sum(mcs), 4), "bit_exact": bool(ok),
                   "roofline_frac": round((U + sum(mcs)) / (mms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 5)}
            if with_cpu:
                cb2 = cpu_baseline_hc(bs, pct, seed, 2)
                if cb2 and "ref_comp_bytes" in cb2 and cb2["sample_blocks"] <= len(mcs):
                    mid["ratio_vs_reference"] = round(cb2["ref_comp_bytes"] / sum(mcs[:cb2["sample_blocks"]]), 4)
                    mid["cpu_baseline"] = {k: cb2[k] for k in ("value", "unit", "cores", "kind", "single_thread_GBps")}
            res["two_table_level2"] = mid
        except Exception as e:
            res["two_table_level2"] = {"error": str(e)}
    return res


def bench_shape_2048(ctx, lz4_amd, torch, data, stream, bs, copy_gbps, use_hints=True):
    """The per-GPU shape of configs[4]: 2048 x 4 MiB blocks (the GiB eight times over), blocks queue on the CUs."""
    reps = (2048 * bs) // data.numel()
    big = data.repeat(reps)
    nb = big.numel() // bs
    stride = (lz4_amd.compress_bound(bs) + 255) & ~255
    comp = torch.empty((nb, stride), dtype=torch.uint8, device=data.device)
    out = torch.empty(big.numel(), dtype=torch.uint8, device=data.device)
    ctab = lz4_amd.BlockTable([big.data_ptr() + i * bs for i in range(nb)], [bs] * nb,
                              [comp.data_ptr() + i * stride for i in range(nb)], [stride] * nb)
    cplan = lz4_amd.Plan(ctx, lz4_amd.OP_COMPRESS, ctab)
    hints = torch.zeros((nb, lz4_amd.hint_bytes(bs)), dtype=torch.uint8, device=data.device) if use_hints else None
    if hints is not None:
        cplan.attach_hints(hints.data_ptr(), hints.stride(0))
    cplan.launch(stream)
    cs = cplan.results(stream)
    dtab = lz4_amd.BlockTable([comp.data_ptr() + i * stride for i in range(nb)], cs,
                              [out.data_ptr() + i * bs for i in range(nb)], [bs] * nb)
    dplan = lz4_amd.Plan(ctx, lz4_amd.OP_DECOMPRESS, dtab)
    if hints is not None:
        dplan.attach_hints(hints.data_ptr(), hints.stride(0))
    dplan.launch(stream)
    assert dplan.results(stream) == [bs] * nb and torch.equal(out, big), "2048-block round trip is not bit exact"
    cms = min(cplan.launch_timed(stream)[0][0] for _ in range(2))
    dms = min(dplan.launch_timed(stream)[0][0] for _ in range(3))
    U, C = big.numel(), sum(cs)
    return {"workload": "configs[4] per-GPU shape: %d independent %d-byte blocks (%.0f GiB: the GiB of configs[1] x %d), device resident%s" % (nb, bs, U / 2**30, reps, ", entry-point tables as in the step" if use_hints else ""),
            "compress_GBps": round(U / (cms * 1e-3) / 1e9, 2), "decompress_GBps": round(U / (dms * 1e-3) / 1e9, 2),
            "roundtrip_GBps": round(U / ((cms + dms) * 1e-3) / 1e9, 2),
            "roofline_compress": roofline_obj("compress", cms, U + C, copy_gbps, None),
            "roofline_decompress": roofline_obj("decompress", dms, U + C, copy_gbps, None)}


def table_bytes_written(torch, hints):
    """Bytes of the entry-point tables the compressor actually wrote (header + rows + end row of every valid table), from the
    tables themselves (csrc/lz4amd_params.h: 32 bytes of header, word 4 = the number of rows, then rows + 1 entries of 8 bytes)."""
    if hints is None:
        return 0
    w = hints.view(torch.int32).view(hints.shape[0], -1)[:, :8].cpu()
    valid = w[:, 0] == 0x32485A4C
    return int((32 + (w[:, 4][valid].to(torch.int64) + 1) * 8).sum().item())


def bench_shape_small(ctx, lz4_amd, torch, data, stream, bs=64 << 10):
    """The small end of north_star's block range: the same GiB as 16384 independent 64 KiB blocks (the frame format's default block
    size): compress, decode with the tables, decode without."""
    U = data.numel()
    nb = U // bs
    stride = (lz4_amd.compress_bound(bs) + 255) & ~255
    comp = torch.empty((nb, stride), dtype=torch.uint8, device=data.device)
    out = torch.empty(U, dtype=torch.uint8, device=data.device)
    ctab = lz4_amd.BlockTable([data.data_ptr() + i * bs for i in range(nb)], [bs] * nb, [comp.data_ptr() + i * stride for i in range(nb)], [stride] * nb)
    cplan = lz4_amd.Plan(ctx, lz4_amd.OP_COMPRESS, ctab)
    hints = torch.zeros((nb, lz4_amd.hint_bytes(bs)), dtype=torch.uint8, device=data.device)
    cplan.attach_hints(hints.data_ptr(), hints.stride(0))
    cplan.launch(stream)
    cs = cplan.results(stream)
    dtab = lz4_amd.BlockTable([comp.data_ptr() + i * stride for i in range(nb)], cs, [out.data_ptr() + i * bs for i in range(nb)], [bs] * nb)
    dplan = lz4_amd.Plan(ctx, lz4_amd.OP_DECOMPRESS, dtab)
    dplan.attach_hints(hints.data_ptr(), hints.stride(0))
    dplan.launch(stream)
    ok = all(c > 0 for c in cs) and dplan.results(stream) == [bs] * nb and bool(torch.equal(out, data))
    fplan = lz4_amd.Plan(ctx, lz4_amd.OP_DECOMPRESS, dtab)
    out.zero_()
    fplan.launch(stream)
    ok = ok and fplan.results(stream) == [bs] * nb and bool(torch.equal(out, data))
    cms = min(cplan.launch_timed(stream)[0][0] for _ in range(3))
    dms = min(dplan.launch_timed(stream)[0][0] for _ in range(3))
    fms = min(fplan.launch_timed(stream)[0][0] for _ in range(3))
    C = sum(cs)
    return {"workload": "%d independent %d-byte blocks (%.2f GiB, the same data), device resident" % (nb, bs, U / 2**30), "bit_exact": ok, "ratio": round(U / C, 4),
            "compress_GBps": round(U / (cms * 1e-3) / 1e9, 2), "decompress_GBps_with_tables": round(U / (dms * 1e-3) / 1e9, 2),
            "decompress_GBps_without_tables": round(U / (fms * 1e-3) / 1e9, 2),
            "compress_frac_of_hbm_peak": round((U + C) / (cms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
            "decompress_frac_of_hbm_peak_with_tables": round((U + C) / (dms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
            "decompress_frac_of_hbm_peak_without_tables": round((U + C) / (fms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
            "note": "one 1024-thread workgroup per block whatever its size: a 64 KiB block's matches reach into the block itself only, its regions wait for the ones just before them"}


def bench_by_compressibility(ctx, lz4_amd, torch, stream, bs, use_hints, pcts=(0, 20, 90), nblk=256, nref=16):
    """The step's two kernels on other compressibilities (the ratio window is two-sided, and the rates depend on the
    data): nblk blocks of `datagen -P<pct> -s1`, compressed and decoded like the step's; ratio_vs_reference = reference bytes / our
    bytes on the first nref blocks (> 1: ours is smaller)."""
    out = {}
    for pct in tuple(pcts) + ("far",):
        if pct == "far":                                     # matches at the far end of the window only: noise of period 65 520 (round 5's advisor: the ratio there)
            import numpy as np
            nblk = min(nblk, 64)
            host = np.resize(np.random.default_rng(7).integers(0, 256, 65520, dtype=np.uint8), nblk * bs)
        else:
            host = gen_data(nblk * bs, pct, 1)
        data = torch.from_numpy(host).cuda()
        stride = (lz4_amd.compress_bound(bs) + 255) & ~255
        comp = torch.empty((nblk, stride), dtype=torch.uint8, device=data.device)
        dec = torch.empty(nblk * bs, dtype=torch.uint8, device=data.device)
        ctab = lz4_amd.BlockTable([data.data_ptr() + i * bs for i in range(nblk)], [bs] * nblk, [comp.data_ptr() + i * stride for i in range(nblk)], [stride] * nblk)
        cplan = lz4_amd.Plan(ctx, lz4_amd.OP_COMPRESS, ctab)
        hints = torch.zeros((nblk, lz4_amd.hint_bytes(bs)), dtype=torch.uint8, device=data.device) if use_hints else None
        if hints is not None:
            cplan.attach_hints(hints.data_ptr(), hints.stride(0))
        cplan.launch(stream)
        cs = cplan.results(stream)
        dtab = lz4_amd.BlockTable([comp.data_ptr() + i * stride for i in range(nblk)], cs, [dec.data_ptr() + i * bs for i in range(nblk)], [bs] * nblk)
        dplan = lz4_amd.Plan(ctx, lz4_amd.OP_DECOMPRESS, dtab)
        if hints is not None:
            dplan.attach_hints(hints.data_ptr(), hints.stride(0))
        dplan.launch(stream)
        ok = all(c > 0 for c in cs) and dplan.results(stream) == [bs] * nblk and bool(torch.equal(dec, data))
        cms = min(cplan.launch_timed(stream)[0][0] for _ in range(3))
        dms = min(dplan.launch_timed(stream)[0][0] for _ in range(3))
        U, C = nblk * bs, sum(cs)
        o = {"blocks": nblk, "ratio": round(U / C, 4), "bit_exact": ok,
             "compress_GBps": round(U / (cms * 1e-3) / 1e9, 1), "decompress_GBps": round(U / (dms * 1e-3) / 1e9, 1),
             "compress_frac_of_hbm_peak": round((U + C) / (cms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
             "decompress_frac_of_hbm_peak": round((U + C) / (dms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)}
        rb = reference_blocks(host, bs, min(nref, nblk))
        if rb is not None:
            o["ratio_vs_reference"] = round(sum(rb[1]) / sum(cs[:len(rb[1])]), 4)
        out["far_window_period_65520" if pct == "far" else "P%d" % pct] = o
    out["note"] = ("%d blocks per row, like the step's table; the step's own compressibility is in the top-level fields.  P0 (incompressible): the compress "
                   "rate of blocks without a single match depends on where their buffers lie - 2.4 ms or 12-17 ms per 64 blocks of 4 MiB, same bytes (DESIGN.md "
                   "section 6, open); from -P2 on it does not.  far_window_period_65520: noise of period 65520 - the reference finds the match once per block and lets it "
                   "run; the tile-parallel parse has to find it again in every 8 KB tile, through a hash table into which 65520 positions went since (DESIGN.md section 8)" % nblk)
    return out


def bench_frame_device(ctx, lz4_amd, torch, data, out, stream, bs, copy_gbps):
    """configs[2]'s device side on HBM-resident data: the kernels LZ4F_compressFrame / LZ4F_decompress launch for a frame of linked
    4 MiB blocks - ONE compress launch (every block sees the 64 KB of source before it), ONE gather launch that packs the blocks
    behind one another in frame layout (4 bytes of room for every block's size field), the decode of the linked blocks (side by side: kernels/chain_spec_kernel.h; block i's
    window is block i-1's output) - without the host's share (transfers, size fields, the serial XXH32 of the content)."""
    U = data.numel()
    nb = U // bs
    stride = (lz4_amd.compress_bound(bs) + 255) & ~255
    comp = torch.empty((nb, stride), dtype=torch.uint8, device=data.device)
    ctab = lz4_amd.BlockTable([data.data_ptr() + i * bs for i in range(nb)], [bs] * nb, [comp.data_ptr() + i * stride for i in range(nb)], [stride] * nb)
    cplan = lz4_amd.Plan.compress_with_history(ctx, ctab, [min(i * bs, 65536) for i in range(nb)])
    cplan.launch(stream)
    cs = cplan.results(stream)
    assert all(c > 0 for c in cs), "linked compression failed"
    C = sum(cs)
    packed = torch.empty(C + 4 * nb + 64, dtype=torch.uint8, device=data.device)
    offs, o = [], 7
    for c in cs:
        offs.append(o + 4); o += 4 + c
    gtab = lz4_amd.BlockTable([comp.data_ptr() + i * stride for i in range(nb)], cs, [packed.data_ptr() + off for off in offs], cs)
    gplan = lz4_amd.Plan(ctx, lz4_amd.OP_GATHER, gtab)
    gplan.launch(stream)
    assert gplan.results(stream) == cs, "gather failed"
    dplan = lz4_amd.Plan.chained(ctx, [packed.data_ptr() + off for off in offs], cs, out.data_ptr(), [bs] * nb)
    out.zero_()
    dplan.launch(stream)
    ok = dplan.results(stream) == [bs] * nb and bool(torch.equal(out, data))
    cms = min(cplan.launch_timed(stream)[0][0] for _ in range(3))
    gms = min(gplan.launch_timed(stream)[1] for _ in range(3))
    dms = min(dplan.launch_timed(stream)[1] for _ in range(3))
    cstats = dplan.chain_stats()
    dplan.close()
    # the chain of copy stages the side-by-side decode replaces (round 5; what a plan without memory for the slots falls back to)
    os.environ["LZ4AMD_CHAIN_SERIAL"] = "1"
    try:
        splan = lz4_amd.Plan.chained(ctx, [packed.data_ptr() + off for off in offs], cs, out.data_ptr(), [bs] * nb)
    finally:
        del os.environ["LZ4AMD_CHAIN_SERIAL"]
    out.zero_()
    sms = splan.launch_timed(stream)[1]
    ok = ok and splan.results(stream) == [bs] * nb and bool(torch.equal(out, data))
    splan.close()
    return {"workload": "configs[2], device side only: %d linked %d-byte blocks (%.2f GiB) resident in HBM: compress with 64 KB of history (one launch), gather into frame layout (one launch), "
                        "decode of the linked blocks side by side (kernels/chain_spec_kernel.h: the ordinary decoder over block 0 and two copies of every other block - a third where the two cannot tell; "
                        "the second from the entry-point table the first one's decode writes -, merge, patch); "
                        "no transfers, no host checksum" % (nb, bs, U / 2**30),
            "bit_exact": ok, "ratio": round(U / C, 4),
            "compress_GBps": round(U / (cms * 1e-3) / 1e9, 2), "gather_GBps": round(C / (gms * 1e-3) / 1e9, 2), "decompress_GBps": round(U / (dms * 1e-3) / 1e9, 3),
            "decompress_serial_chain_GBps": round(U / (sms * 1e-3) / 1e9, 3),
            "side_by_side": {"units": int(cstats["units"]), "units_decoded_three_times": int(cstats["units_decoded_three_times"]),
                             "fraction_walked_by_patch": round(cstats["bytes_walked_by_patch"] / max(1, cstats["bytes_of_units_1_on"]), 4),
                             "note": "lz4amd_plan_chain_stats: a unit is decoded against two made-up histories; a third decode only where a match in the unit's first 255 bytes "
                                     "reads one of the first 256 bytes of the 64 KB before it (gated inside the launch); the patch pass walks every unit up to its last "
                                     "byte that is a copy of a history byte"},
            "compress_plus_gather_GBps": round(U / ((cms + gms) * 1e-3) / 1e9, 2),
            "roofline_compress": roofline_obj("compress", cms, U + C, copy_gbps, None),
            "roofline_gather": {"kernel": "gather", "bound": "hbm", "achieved": round(2 * C / (gms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                "frac": round(2 * C / (gms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 5), "algorithmic_bytes_per_launch": 2 * C, "avg_ms": round(gms, 4)},
            "roofline_decompress": {"kernel": "decompress_runs + spec_gate + decompress + spec_merge + spec_patch (linked blocks side by side)", "bound": "hbm", "achieved": round((U + C) / (dms * 1e-3) / 1e9, 2),
                                    "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                    "frac": round((U + C) / (dms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 6), "algorithmic_bytes_per_launch": U + C, "avg_ms": round(dms, 3),
                                    "limited_by": "every block but the first is decoded twice (against two made-up histories: which bytes depend on the history, and on which "
                                                  "byte of it; a third time where those two cannot tell): the first copy without an entry-point table (a frame has no room for them), writing one, "
                                                  "the second from it in a second launch, then two bandwidth passes; round 5's chain of copy stages, one CU at a time: "
                                                  "decompress_serial_chain_GBps"}}


def bench_frame(lz4_amd, host):
    """configs[2]: the GiB as one frame (LZ4F_max4MB, blockLinked, content checksum) through the host-pointer API."""
    class FrameInfo(ctypes.Structure):
        _fields_ = [("blockSizeID", ctypes.c_int), ("blockMode", ctypes.c_int), ("contentChecksumFlag", ctypes.c_int),
                    ("frameType", ctypes.c_int), ("contentSize", ctypes.c_ulonglong), ("dictID", ctypes.c_uint),
                    ("blockChecksumFlag", ctypes.c_int)]

    class Prefs(ctypes.Structure):
        _fields_ = [("frameInfo", FrameInfo), ("compressionLevel", ctypes.c_int), ("autoFlush", ctypes.c_uint),
                    ("favorDecSpeed", ctypes.c_uint), ("reserved", ctypes.c_uint * 3)]
    L = lz4_amd.lib()
    st, vp = ctypes.c_size_t, ctypes.c_void_p
    L.LZ4F_compressFrameBound.restype = st
    L.LZ4F_compressFrameBound.argtypes = [st, ctypes.POINTER(Prefs)]
    L.LZ4F_compressFrame.restype = st
    L.LZ4F_compressFrame.argtypes = [vp, st, vp, st, ctypes.POINTER(Prefs)]
    L.LZ4F_isError.argtypes = [st]
    L.LZ4F_createDecompressionContext.restype = st
    L.LZ4F_createDecompressionContext.argtypes = [ctypes.POINTER(vp), ctypes.c_uint]
    L.LZ4F_freeDecompressionContext.argtypes = [vp]
    L.LZ4F_decompress.restype = st
    L.LZ4F_decompress.argtypes = [vp, vp, ctypes.POINTER(st), vp, ctypes.POINTER(st), vp]
    import numpy as np
    n = host.size

    def one_frame(bsid, mode, csum):
        """second of two calls (the first allocates the library's staging buffers and touches the pages of dst / out)"""
        p = Prefs()
        p.frameInfo.blockSizeID = bsid; p.frameInfo.blockMode = mode; p.frameInfo.contentChecksumFlag = csum
        cap = L.LZ4F_compressFrameBound(n, ctypes.byref(p))
        dst = np.empty(cap, dtype=np.uint8)
        out = np.empty(n, dtype=np.uint8)
        tc = td = None
        for _ in range(2):
            t0 = time.perf_counter()
            fsz = L.LZ4F_compressFrame(dst.ctypes.data, cap, host.ctypes.data, n, ctypes.byref(p))
            tc = time.perf_counter() - t0
            if L.LZ4F_isError(fsz):
                return {"error": "LZ4F_compressFrame failed"}
            d = vp()
            L.LZ4F_createDecompressionContext(ctypes.byref(d), 100)
            t0 = time.perf_counter()
            ipos = opos = 0
            while ipos < fsz:
                ss = st(fsz - ipos); ds = st(n - opos)
                r = L.LZ4F_decompress(d, out.ctypes.data + opos, ctypes.byref(ds), dst.ctypes.data + ipos, ctypes.byref(ss), None)
                if L.LZ4F_isError(r):
                    L.LZ4F_freeDecompressionContext(d)
                    return {"error": "LZ4F_decompress failed"}
                ipos += ss.value; opos += ds.value
                if r == 0:
                    break
            td = time.perf_counter() - t0
            L.LZ4F_freeDecompressionContext(d)
        ok = opos == n and bool((out == host).all())
        return {"header_hex": bytes(dst[:7]).hex(), "frame_bytes": int(fsz), "ratio": round(n / fsz, 4),
                "compress_GBps": round(n / tc / 1e9, 3), "decompress_GBps": round(n / td / 1e9, 3), "bit_exact": ok}

    def ref_frame(bsid, mode, csum, sample):
        """the reference's own LZ4F_compressFrame / LZ4F_decompress (oracle/_ref, one host thread) on the first `sample` bytes"""
        so = os.path.join(ROOT, "oracle", "_ref", "liblz4_ref.so")
        if not os.path.exists(so):
            return None
        R = ctypes.CDLL(so)
        R.LZ4F_compressFrameBound.restype = st; R.LZ4F_compressFrameBound.argtypes = [st, ctypes.POINTER(Prefs)]
        R.LZ4F_compressFrame.restype = st; R.LZ4F_compressFrame.argtypes = [vp, st, vp, st, ctypes.POINTER(Prefs)]
        R.LZ4F_isError.argtypes = [st]
        R.LZ4F_createDecompressionContext.restype = st; R.LZ4F_createDecompressionContext.argtypes = [ctypes.POINTER(vp), ctypes.c_uint]
        R.LZ4F_freeDecompressionContext.argtypes = [vp]
        R.LZ4F_decompress.restype = st; R.LZ4F_decompress.argtypes = [vp, vp, ctypes.POINTER(st), vp, ctypes.POINTER(st), vp]
        p = Prefs()
        p.frameInfo.blockSizeID = bsid; p.frameInfo.blockMode = mode; p.frameInfo.contentChecksumFlag = csum
        cap = R.LZ4F_compressFrameBound(sample, ctypes.byref(p))
        dst = np.empty(cap, dtype=np.uint8); out = np.empty(sample, dtype=np.uint8)
        dst[:] = 0; out[:] = 0                                        # touch the pages outside the timed region
        t0 = time.perf_counter()
        fsz = R.LZ4F_compressFrame(dst.ctypes.data, cap, host.ctypes.data, sample, ctypes.byref(p))
        tc = time.perf_counter() - t0
        if R.LZ4F_isError(fsz):
            return {"error": "reference LZ4F_compressFrame failed"}
        d = vp()
        R.LZ4F_createDecompressionContext(ctypes.byref(d), 100)
        t0 = time.perf_counter()
        ipos = opos = 0
        while ipos < fsz:
            ss = st(fsz - ipos); ds = st(sample - opos)
            rr = R.LZ4F_decompress(d, out.ctypes.data + opos, ctypes.byref(ds), dst.ctypes.data + ipos, ctypes.byref(ss), None)
            if R.LZ4F_isError(rr):
                break
            ipos += ss.value; opos += ds.value
            if rr == 0:
                break
        td = time.perf_counter() - t0
        R.LZ4F_freeDecompressionContext(d)
        return {"kind": "reference", "cores": 1, "unit": "GB/s", "compress_GBps": round(sample / tc / 1e9, 3), "decompress_GBps": round(sample / td / 1e9, 3),
                "frame_bytes": int(fsz), "bit_exact": bool(opos == sample and (out == host[:sample]).all()),
                "sample": "the first %d MiB of the same data through the reference's LZ4F_compressFrame / LZ4F_decompress (lib/lz4frame.c), same preferences, one host thread, one call each" % (sample >> 20)}

    r = {"workload": "configs[2]: %.2f GiB as one frame, LZ4F_max4MB, blockLinked, content checksum; host buffers (upload, kernels, download, XXH32 on the host)" % (n / 2**30)}
    r.update(one_frame(7, 0, 1))
    try:
        r["cpu_baseline"] = ref_frame(7, 0, 1, min(n, 512 << 20))
    except Exception as e:
        r["cpu_baseline"] = {"error": str(e)}
    nc = one_frame(7, 0, 0)                                  # the same linked frame without the content checksum: no serial XXH32 on the host
    r["linked_4M_no_checksum"] = {k: nc[k] for k in ("compress_GBps", "decompress_GBps", "bit_exact", "header_hex") if k in nc} if "error" not in nc else nc
    l64 = one_frame(4, 0, 0)                                 # the frame format's defaults: linked 64 KiB blocks, no checksums
    r["linked_64K_no_checksum"] = {k: l64[k] for k in ("compress_GBps", "decompress_GBps", "bit_exact", "header_hex", "ratio") if k in l64} if "error" not in l64 else l64
    d64 = one_frame(4, 1, 0)
    d64["workload"] = "the same GiB as one frame of independent 64 KiB blocks (the frame format's default block size), no checksums"
    try:
        d64["cpu_baseline"] = ref_frame(4, 1, 0, min(n, 512 << 20))
    except Exception as e:
        d64["cpu_baseline"] = {"error": str(e)}
    r["independent_64K"] = d64
    r["note"] = ("PCIe inclusive (host buffers in, host buffers out): a parity path, not the HBM-resident rate; second of two calls. "
                 "The decoder works in batches of up to 32 MiB of input / 1024 blocks: a batch is on the device (a helper thread) while the one before is handed "
                 "to the caller and the next is taken in, through page-locked buffers.  Linked blocks decode side by side, batch by batch (kernels/chain_spec_kernel.h; round 5's chain of copy "
                 "stages bound a linked frame at ~3.4 GB/s whatever else overlapped); what bounds a frame now is the host's share: the transfers, and for "
                 "frames with a content checksum the one serial XXH32 over the content on the host (~6.3 GB/s on this box)")
    return r


def _sync(torch, dev):
    if dev.type == "cuda":
        torch.cuda.synchronize()


def pack_python(torch, dev, comp, csizes):
    """The used part of every slot, one behind the other (the CPU test's packer; the GPU job packs with LZ4AMD_OP_GATHER)."""
    packed = torch.empty(sum(csizes), dtype=torch.uint8, device=dev)
    off = 0
    for i, c in enumerate(csizes):
        packed[off:off + c] = comp[i, :c]
        off += c
    return packed


def gather_corpus(dist, torch, rank, world, data):
    """Setup of data_path, untimed: the corpus lives on the root, so rank 0 first collects every rank's shard (each rank generated
    its own).  Returns the list of shards on rank 0, None elsewhere."""
    shards = [torch.empty_like(data) for _ in range(world)] if rank == 0 else None
    dist.gather(data, gather_list=shards, dst=0)
    return shards


def data_path(dist, torch, dev, rank, world, data, comp, csizes, pack=None, shards="gather"):
    """configs[4]'s movement with torch.distributed (backend nccl = RCCL over xGMI on the GPU box, gloo in
    the CPU test).  Setup, untimed: the corpus lives on the root, so rank 0 first collects every rank's shard (each
    rank generated its own, `datagen -s<rank>`).  Timed: (1) scatter of the DISTINCT shards from rank 0, (2) all_gather
    of the int32 compressed sizes, (3) the payloads, packed per rank (`pack`), to rank 0 with exact-length point-to-point
    transfers (no padding to the largest rank).  Returns seconds per phase, what this rank received and - on rank 0 -
    every shard and the payloads cut back into (rank, block) order with the sizes table."""
    U = data.numel()
    pack = pack or (lambda c, z: pack_python(torch, dev, c, z))
    if isinstance(shards, str):
        shards = gather_corpus(dist, torch, rank, world, data)         # setup: the root holds the whole corpus
    recv = torch.empty_like(data)
    _sync(torch, dev); dist.barrier()
    t0 = time.perf_counter()
    dist.scatter(recv, scatter_list=shards, src=0)
    _sync(torch, dev); dist.barrier()
    t_scatter = time.perf_counter() - t0
    # compressed sizes of every block of every rank
    mine = torch.tensor(list(csizes), dtype=torch.int32, device=dev)
    allsz = [torch.empty_like(mine) for _ in range(world)]
    _sync(torch, dev); dist.barrier()
    t0 = time.perf_counter()
    dist.all_gather(allsz, mine)
    _sync(torch, dev)
    t_sizes = time.perf_counter() - t0
    totals = [int(t.sum().item()) for t in allsz]
    # payloads: packed on the device, then exact lengths to the root
    _sync(torch, dev); dist.barrier()
    t0 = time.perf_counter()
    packed = pack(comp, list(csizes))
    _sync(torch, dev)
    t_pack = time.perf_counter() - t0
    dist.barrier()
    t0 = time.perf_counter()
    bufs = None
    if rank == 0:
        bufs = [packed] + [torch.empty(totals[r], dtype=torch.uint8, device=dev) for r in range(1, world)]
        ops = [dist.P2POp(dist.irecv, bufs[r], r) for r in range(1, world) if totals[r]]
    else:
        ops = [dist.P2POp(dist.isend, packed, 0)] if totals[rank] else []
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    _sync(torch, dev); dist.barrier()
    t_gather = time.perf_counter() - t0
    blocks = None
    if rank == 0:                                            # (rank, block) order from the sizes table
        blocks = []
        for r in range(world):
            o = 0
            for c in allsz[r].tolist():
                blocks.append(bufs[r][o:o + c])
                o += c
    return {"scatter_s": t_scatter, "sizes_s": t_sizes, "pack_s": t_pack, "gather_s": t_gather, "received": recv, "shards": shards,
            "scatter_bytes": U * (world - 1), "gather_bytes": sum(totals) - totals[0], "blocks": blocks, "sizes": [t.tolist() for t in allsz]}


def rccl_version(torch):
    """The RCCL (torch's "nccl") version this process is linked with, as a string; None when torch has none."""
    try:
        v = torch.cuda.nccl.version()
        return ".".join(str(x) for x in v) if isinstance(v, (tuple, list)) else str(v)
    except Exception:
        return None


def launch_ranks(n, argv):
    """`bench.py --gpus N` without a launcher: start the N ranks (this file again, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the
    environment), wait for them; rank 0 prints the JSON line on the stdout it inherits.  Returns the worst exit code."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env))
    rc = 0
    live = list(procs)
    while live:                                              # all ranks are polled together: the first one that dies ends the others at once
        time.sleep(0.05)
        for p in list(live):
            code = p.poll()
            if code is None:
                continue
            live.remove(p)
            if code and not rc:
                rc = code
                for q in live:                               # (do not leave them at a rendezvous until its timeout)
                    q.kill()
    return rc


def dry_run(args, world, rank):
    """`--dry-run`: the N-rank plumbing of this file WITHOUT the codec and without a GPU (the CPU container's check of the command
    line: launcher, rendezvous, shard plan, the scatter / all_gather / gather movement over gloo with every block "stored", the
    aggregation).  Prints a line with n_gpus and no value: it measures nothing."""
    import torch
    import torch.distributed as dist
    if world > 1:
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    nb, bs = args.blocks or 4, args.block_bytes
    plan_s = shard_plan(nb, rank, world)
    host = gen_data(nb * bs, args.pct, plan_s["seed"])
    data = torch.from_numpy(host)
    comp = data.view(nb, bs)                                  # every block "stored": the movement is what is exercised
    csizes = [bs] * nb
    t_max, bytes_all = aggregate(dist if world > 1 else None, 1.0, nb * bs)
    dp = None
    if world > 1:
        r = data_path(dist, torch, torch.device("cpu"), rank, world, data, comp, csizes)
        ok = bool(torch.equal(r["received"], data))
        if rank == 0:
            ok = ok and len(r["blocks"]) == world * nb and all(bool(torch.equal(r["blocks"][rr * nb + i], r["shards"][rr][i * bs:(i + 1) * bs]))
                                                               for rr in range(world) for i in range(nb))
        okt = torch.tensor([1.0 if ok else 0.0], dtype=torch.float64)
        dist.all_reduce(okt, op=dist.ReduceOp.MIN)
        dp = {"ok": bool(okt.item() == 1.0), "backend": dist.get_backend(), "world_size": dist.get_world_size(),
              "scatter_bytes": r["scatter_bytes"], "gather_bytes": r["gather_bytes"]}
    if rank == 0:
        print(json.dumps({"metric": "GB/s compress + decompress, 4 MB independent blocks", "value": None, "unit": "GB/s", "dry_run": True,
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "higher_is_better": True, "scaling": "weak",
                          "config": {"workload": "dry run: %d x %d-byte blocks per rank, no codec, no GPU" % (nb, bs), "blocks_per_gpu": nb, "block_bytes": bs},
                          "bytes_all_ranks": bytes_all, "data_path": dp,
                          "note": "plumbing check of `bench.py --gpus N` (launcher, rendezvous, sharding, collectives over gloo): nothing is measured"}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--blocks", type=int, default=None,
                    help="blocks per GPU; default 256 x 4 MiB = 1 GiB at N = 1 (configs[1]), 2048 x 4 MiB = 8 GiB per GPU at N > 1 (configs[4])")
    ap.add_argument("--block-bytes", type=int, default=4 << 20)
    ap.add_argument("--pct", type=int, default=60, help="datagen -P compressibility")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-hc", action="store_true", help="skip the LZ4_compress_HC (configs[3]) side measurement")
    ap.add_argument("--no-extras", action="store_true", help="skip the 2048-block shape and the configs[2] frame object")
    ap.add_argument("--no-data-path", action="store_true", help="N>1: skip the RCCL scatter / gather measurement")
    ap.add_argument("--shards", choices=("root", "local"), default="root",
                    help="N>1: root = configs[4]'s movement (rank 0 holds the corpus: scatter of the shards, gather of the payloads, all of it through "
                         "rank 0's xGMI links); local = every rank keeps the shard it generated, only the compressed sizes are all-gathered: the codec's "
                         "scaling without the funnel through rank 0")
    ap.add_argument("--no-foreign", action="store_true", help="skip the decode of the same blocks without tables and of reference-compressed blocks (profiling runs: the decompress kernel's average then is the step's)")
    ap.add_argument("--no-hints", action="store_true",
                    help="do not pass the compressor's entry-point tables to the decoder (include/lz4amd.h): every block is decoded the way a foreign block is")
    ap.add_argument("--dry-run", action="store_true",
                    help="plumbing check without codec and GPU: launcher, rendezvous, sharding and the collectives over gloo (prints n_gpus, measures nothing)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:     # no launcher around us: be the launcher
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    if args.dry_run or os.environ.get("LZ4AMD_BENCH_DRYRUN") == "1":
        return dry_run(args, int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")))

    import torch
    import torch.distributed as dist
    import lz4_amd

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        # developer aid: LZ4AMD_BENCH_LOOPBACK=1 runs the N > 1 code path on a box with ONE GPU (every rank on cuda:0,
        # gloo instead of RCCL) - a functional check of this file, not a measurement
        loopback = os.environ.get("LZ4AMD_BENCH_LOOPBACK") == "1"
        if loopback:
            local_rank = 0
        elif torch.cuda.device_count() < world:
            sys.exit("bench.py: %d ranks asked for, %d GPU(s) visible (rank %d)" % (world, torch.cuda.device_count(), rank))
        torch.cuda.set_device(local_rank)
        if loopback:
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group(backend="nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", local_rank if world > 1 else 0)

    if args.blocks is None:
        args.blocks = 256 if world == 1 else 2048
    ctx = lz4_amd.Context(dev.index)                     # raises loudly without the HIP library / GPU
    plan_s = shard_plan(args.blocks, rank, world)
    bs, nb = args.block_bytes, plan_s["n_blocks"]
    U = nb * bs

    # the synthetic shard: one datagen stream per GiB (seeds 1000 * rank + k), generated on one thread each - every block of
    # the shard is distinct data (a single 8 GiB stream is one serial PRNG chain: 15 s per rank)
    n_streams = U >> 30 if (U >= (2 << 30) and U % (1 << 30) == 0) else 1
    seeds = [plan_s["seed"]] if n_streams == 1 else [1000 * plan_s["seed"] + k for k in range(n_streams)]
    host = gen_data(U, args.pct, seeds[0]) if n_streams == 1 else gen_data_seeds(U, args.pct, seeds)
    gen_u = U
    data = torch.from_numpy(host).to(dev)
    stream = torch.cuda.current_stream().cuda_stream

    # ---- block tables (built once, like the reference bench's blockParam_t table)
    stride = (lz4_amd.compress_bound(bs) + 255) & ~255
    comp = torch.empty((nb, stride), dtype=torch.uint8, device=dev)
    out = torch.empty(U, dtype=torch.uint8, device=dev)
    ctab = lz4_amd.BlockTable([data.data_ptr() + i * bs for i in range(nb)], [bs] * nb,
                              [comp.data_ptr() + i * stride for i in range(nb)], [stride] * nb)
    cplan = lz4_amd.Plan(ctx, lz4_amd.OP_COMPRESS, ctab)
    # the optional out-of-band column of the block table: the compressor writes an entry-point table next to every block
    # (16 B per KB of source), the decoder of the same job parses from it.  Part of the step on both sides.
    hstride = lz4_amd.hint_bytes(bs)
    hints = None if args.no_hints else torch.zeros((nb, hstride), dtype=torch.uint8, device=dev)
    if hints is not None:
        cplan.attach_hints(hints.data_ptr(), hstride)
    cplan.launch(stream)
    csizes = cplan.results(stream)
    assert all(c > 0 for c in csizes), "compression failed"
    C = sum(csizes)
    dtab = lz4_amd.BlockTable([comp.data_ptr() + i * stride for i in range(nb)], csizes,
                              [out.data_ptr() + i * bs for i in range(nb)], [bs] * nb)
    dplan = lz4_amd.Plan(ctx, lz4_amd.OP_DECOMPRESS, dtab)
    if hints is not None:
        dplan.attach_hints(hints.data_ptr(), hstride)
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
    table_stats = dplan.hint_stats() if hints is not None else None
    # ---- not part of the step: the same blocks decoded WITHOUT their tables (what a block of foreign origin costs: the
    #      decoder first discovers the token chain), and blocks the reference compressor made (oracle/_ref, on the host)
    foreign = {}
    if rank == 0 and not args.no_foreign:
        fplan = lz4_amd.Plan(ctx, lz4_amd.OP_DECOMPRESS, dtab)
        out.zero_()
        fplan.launch(stream)
        assert fplan.results(stream) == [bs] * nb and torch.equal(out, data), "round trip without tables is not bit exact"
        fms = sum(fplan.launch_timed(stream)[0][0] for _ in range(args.steps)) / args.steps
        foreign["own_blocks_without_table"] = {"decompress_GBps": round(U / (fms * 1e-3) / 1e9, 2), "avg_ms": round(fms, 4),
                                               "frac_of_hbm_peak": round((U + C) / (fms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)}
        try:
            rb = reference_blocks(host, bs, min(nb, 64))
            if rb is not None:
                rcomp, rsz = rb
                nr = len(rsz)
                rdev = torch.from_numpy(rcomp).to(dev)
                rstride = rcomp.shape[1]
                reps = nb // nr                                  # the sample, repeated to the full table (same work per block)
                rtab = lz4_amd.BlockTable([rdev.data_ptr() + (i % nr) * rstride for i in range(nr * reps)], [rsz[i % nr] for i in range(nr * reps)],
                                          [out.data_ptr() + i * bs for i in range(nr * reps)], [bs] * (nr * reps))
                rplan = lz4_amd.Plan(ctx, lz4_amd.OP_DECOMPRESS, rtab)
                out.zero_()
                rplan.launch(stream)
                ok = rplan.results(stream) == [bs] * (nr * reps) and torch.equal(out[:nr * bs], data[:nr * bs])
                rms = sum(rplan.launch_timed(stream)[0][0] for _ in range(args.steps)) / args.steps
                Ur, Cr = nr * reps * bs, sum(rsz) * reps
                foreign["_ref_ratio"] = round(sum(rsz) / sum(csizes[:nr]), 4)      # reference bytes / our bytes on the same blocks
                foreign["reference_compressed_blocks"] = {"decompress_GBps": round(Ur / (rms * 1e-3) / 1e9, 2), "avg_ms": round(rms, 4), "bit_exact": bool(ok),
                                                          "frac_of_hbm_peak": round((Ur + Cr) / (rms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                                                          "sample": "%d blocks compressed by the reference's LZ4_compress_default on the host, %d times over" % (nr, reps)}
                # the same blocks again, this time letting the first decode write their entry-point tables (lz4amd_plan_make_hints)
                mh = torch.zeros((nr * reps, lz4_amd.hint_bytes(bs)), dtype=torch.uint8, device=dev)
                mplan = lz4_amd.Plan(ctx, lz4_amd.OP_DECOMPRESS, rtab)
                mplan.attach_hints(mh.data_ptr(), mh.stride(0)); mplan.make_hints(True)
                out.zero_()
                first_ms = mplan.launch_timed(stream)[0][0]
                made = mplan.hints_made()
                out.zero_()
                mplan.launch(stream)
                ok2 = mplan.results(stream) == [bs] * (nr * reps) and torch.equal(out[:nr * bs], data[:nr * bs])
                mms = sum(mplan.launch_timed(stream)[0][0] for _ in range(args.steps)) / args.steps
                used, rejected = mplan.hint_stats()
                foreign["reference_compressed_blocks"]["decoded_again_from_tables_the_first_decode_made"] = {
                    "decompress_GBps": round(Ur / (mms * 1e-3) / 1e9, 2), "avg_ms": round(mms, 4), "bit_exact": bool(ok2),
                    "first_decode_ms": round(first_ms, 4), "tables_made": made, "decodes_from_tables": used, "tables_rejected": rejected,
                    "frac_of_hbm_peak": round((Ur + Cr) / (mms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)}
        except Exception as e:
            foreign["reference_compressed_blocks"] = {"error": str(e)}

    # ---- not part of the step: the batched XXH32 kernel (frame block checksums) over the same blocks
    xplan = lz4_amd.Plan(ctx, lz4_amd.OP_XXH32, lz4_amd.BlockTable([data.data_ptr() + i * bs for i in range(nb)], [bs] * nb, [0] * nb, [0] * nb))
    xplan.launch(stream); xplan.results(stream)
    x_ms = sum(xplan.launch_timed(stream)[1] for _ in range(3)) / 3

    result = None
    if rank == 0:
        copy_gbps = stream_copy_gbps(ctx, lz4_amd, torch, 1 << 30, stream)
        tbytes = table_bytes_written(torch, hints)
        alg = {"compress": U + C, "decompress": U + C}        # U read + C written / C read + U written (the tables' rows, written / read as well, are reported beside: frac_with_table_rows)
        pmc = measured_traffic() if (nb == 256 and bs == 4 << 20 and args.pct == 60) else {}
        traffic = {k: pmc.get(k, {}).get("hbm_bytes_per_launch") for k in ("compress", "decompress")}
        kernels = []
        for name, ms in k_ms.items():
            gbps = alg[name] / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
            kernels.append({"kernel": name, "avg_ms": round(ms, 4), "algorithmic_bytes": alg[name],
                            "GBps": round(gbps, 1), "frac_of_hbm_peak": round(gbps / HBM_PEAK_GBPS, 4),
                            "frac_of_measured_copy": round(gbps / copy_gbps, 4) if copy_gbps else None})
        dom = max(kernels, key=lambda k: k["avg_ms"])
        result = {
            "metric": "GB/s compress + decompress, 4 MB independent blocks (uncompressed bytes through one compress+decompress pass per second)",
            "value": round(bytes_all / t_max / 1e9, 3), "unit": "GB/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "world_size": (dist.get_world_size() if (dist is not None and dist.is_initialized()) else 1), "rccl_version": rccl_version(torch),
            "ms_per_step": round(t_max / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic (datagen -P%d restated in tools/datagen.c, md5-pinned to the reference tool)" % args.pct,
            "config": {"workload": "%s: %d independent %d-byte blocks per GPU (%.2f GiB%s), datagen -P%d -s<rank>, block compress + decompress, device resident"
                                   % ("configs[1]" if world == 1 else "configs[4] (%d GPUs)" % world, nb, bs, U / 2**30,
                                      ": %d datagen streams of 1 GiB, seeds 1000 * rank + k" % n_streams if n_streams > 1 else "", args.pct),
                       "blocks_per_gpu": nb, "block_bytes": bs, "parallelism": "blocks sharded over %d GPU(s), no collective inside the codec" % world},
            "compress_GBps": round(U / (c_total * 1e-3) / 1e9, 2),
            "decompress_GBps": round(U / (d_total * 1e-3) / 1e9, 2),
            "ratio": round(U / C, 4), "compressed_bytes": C,
            # the step's decode is handed the compressor's entry-point tables (an interface of this library, out of band: plain LZ4
            # blocks - frames, LZ4_decompress_safe, foreign data - come without).  The same round trip without them, and with blocks
            # the reference compressed, beside it:
            "value_note": "`value` = compress (writes the tables) + decompress from the tables; `value_without_tables` = the same compress + decode of "
                          "the same blocks as plain LZ4 blocks; `value_reference_input` = the same compress + decode of blocks made by the reference's "
                          "LZ4_compress_default (what LZ4_decompress_safe / LZ4F_decompress callers get)",
            "value_without_tables": (round(U / ((c_total + foreign["own_blocks_without_table"]["avg_ms"]) * 1e-3) / 1e9, 3)
                                     if "own_blocks_without_table" in foreign else None),
            "value_reference_input": (round(U / ((c_total + foreign["reference_compressed_blocks"]["avg_ms"]) * 1e-3) / 1e9, 3)
                                      if "avg_ms" in foreign.get("reference_compressed_blocks", {}) else None),
            "decompress_GBps_without_tables": foreign.get("own_blocks_without_table", {}).get("decompress_GBps"),
            "decompress_GBps_reference_input": foreign.get("reference_compressed_blocks", {}).get("decompress_GBps"),
            "ratio_with_tables": round(U / (C + tbytes), 4) if hints is not None else None,
            "roofline": roofline_obj(dom["kernel"], dom["avg_ms"], dom["algorithmic_bytes"], copy_gbps, traffic.get(dom["kernel"]), tbytes),
            "roofline_decompress": roofline_obj("decompress", k_ms["decompress"], alg["decompress"], copy_gbps, traffic.get("decompress"), tbytes),
            "kernels": kernels,
            "entry_point_tables": ({"used": True, "room_bytes_per_block": hstride, "room_bytes_per_launch": hstride * nb,
                                    "bytes_written_per_launch": tbytes, "bytes_written_per_block": tbytes // nb, "fraction_of_compressed_bytes": round(tbytes / C, 4),
                                    "blocks_decoded_from_their_table": table_stats[0], "tables_rejected": table_stats[1],
                                    "launches_counted": 1 + args.warmup + 2 * args.steps,
                                    "note": "optional out-of-band column of the block table (include/lz4amd.h): written by lz4amd_k_compress, checked row by row by the decoder; "
                                            "`decompress_GBps` above is WITH the tables, `decode_of_foreign_blocks` is without"}
                                   if hints is not None else {"used": False}),
            "decode_of_foreign_blocks": foreign,
            "kernel_sources_sha": kernel_sources_sha(),
            "extras": {"xxh32_batch_GBps": round(U / (x_ms * 1e-3) / 1e9, 1), "xxh32_batch_ms": round(x_ms, 3),
                       "note": "XXH32 (seed 0) of every 4 MiB block, one wave per block; not in `value`"},
        }
    # ---- N > 1: the movement configs[4] names, over RCCL (not part of `value`, which stays the kernel-only rate)
    dp = None
    if world > 1 and not args.no_data_path and args.shards == "local":
        # every rank keeps its own shard: the one exchange is the all_gather of the compressed sizes (what a consumer of the sharded corpus needs
        # to address the blocks); no byte of payload crosses a link
        try:
            mine = torch.tensor(list(csizes), dtype=torch.int32, device=dev)
            allsz = [torch.empty_like(mine) for _ in range(world)]
            torch.cuda.synchronize(); dist.barrier()
            t0 = time.perf_counter()
            dist.all_gather(allsz, mine)
            torch.cuda.synchronize()
            t_sizes = aggregate(dist, time.perf_counter() - t0, 0, device=dev)[0]
            if rank == 0:
                result["data_path"] = {"mode": "--shards local: shards stay where they were generated; all_gather of int32 csize[%d] only" % nb,
                                       "world_size": dist.get_world_size(), "backend": dist.get_backend(), "sizes_s": round(t_sizes, 6),
                                       "compressed_bytes_all_ranks": int(sum(int(t.sum().item()) for t in allsz)),
                                       "kernel_only_GBps": round(bytes_all / t_max / 1e9, 3),
                                       "note": "unmeasured on multi-GPU hardware until a SCALE record exists"}
        except Exception as e:
            if rank == 0:
                result["data_path"] = {"error": str(e)}
    elif world > 1 and not args.no_data_path:
        # setup, untimed and outside the watchdog below: rank 0 collects every rank's shard (8 GiB each at the default size: tens of GiB over
        # xGMI); a collective that never returns must still not cost the bench line, so it has a (long) watchdog of its own
        def bail_setup():
            if rank == 0:
                result["data_path"] = {"error": "the setup gather of the shards did not finish within 600 s"}
                print(json.dumps(result), flush=True)
            os._exit(0)
        w0 = threading.Timer(600.0, bail_setup)
        w0.daemon = True
        w0.start()
        try:
            corpus = gather_corpus(dist, torch, rank, world, data)
            torch.cuda.synchronize()
        finally:
            w0.cancel()
        # a collective that never returns must not cost the bench line: after 120 s rank 0 prints what it has and every rank leaves
        def bail():
            if rank == 0:
                result["data_path"] = {"error": "the scatter / all_gather / gather phase did not finish within 120 s"}
                print(json.dumps(result), flush=True)
            os._exit(0)
        watchdog = threading.Timer(120.0, bail)
        watchdog.daemon = True
        watchdog.start()
        try:
            def pack_gpu(comp_, cs_):
                # LZ4AMD_OP_GATHER: one launch copies the used part of every slot to its place in the packed payload
                packed = torch.empty(sum(cs_), dtype=torch.uint8, device=dev)
                offs, o = [], 0
                for c in cs_:
                    offs.append(o); o += c
                gt = lz4_amd.BlockTable([comp_.data_ptr() + i * stride for i in range(nb)], cs_,
                                        [packed.data_ptr() + off for off in offs], cs_)
                gp = lz4_amd.Plan(ctx, lz4_amd.OP_GATHER, gt)
                gp.launch(stream)
                assert gp.results(stream) == cs_, "gather op failed"
                return packed
            r = data_path(dist, torch, dev, rank, world, data, comp, csizes, pack=pack_gpu, shards=corpus)
            ok = bool(torch.equal(r["received"], data))               # every rank got its own shard back from the root
            if rank == 0:
                # the root decodes a sample of EVERY rank's gathered blocks with its own decoder and compares with the shard it holds
                ok = ok and len(r["blocks"]) == world * nb
                sample = list(range(0, nb, max(1, nb // 8)))
                for rr in range(world):
                    blks = [r["blocks"][rr * nb + i] for i in sample]
                    cs = [int(b.numel()) for b in blks]
                    so = torch.empty(len(sample) * bs, dtype=torch.uint8, device=dev)
                    vt = lz4_amd.BlockTable([b.data_ptr() for b in blks], cs, [so.data_ptr() + k * bs for k in range(len(sample))], [bs] * len(sample))
                    vp = lz4_amd.Plan(ctx, lz4_amd.OP_DECOMPRESS, vt)
                    vp.launch(stream)
                    ok = ok and vp.results(stream) == [bs] * len(sample)
                    for k, i in enumerate(sample):
                        ok = ok and bool(torch.equal(so[k * bs:(k + 1) * bs], r["shards"][rr][i * bs:(i + 1) * bs]))
            dp = {k: aggregate(dist, r[k], 0, device=dev)[0] for k in ("scatter_s", "sizes_s", "pack_s", "gather_s")}
            okt = torch.tensor([1.0 if ok else 0.0], dtype=torch.float64, device=dev)
            dist.all_reduce(okt, op=dist.ReduceOp.MIN)
            names = [None] * world
            dist.all_gather_object(names, torch.cuda.get_device_name(dev))
            dp.update({"ok": bool(okt.item() == 1.0), "scatter_bytes": r["scatter_bytes"], "gather_bytes": r["gather_bytes"], "devices": names})
        except Exception as e:
            dp = {"error": str(e)}
        finally:
            watchdog.cancel()

    if rank == 0:
        if dp is not None:
            if "error" not in dp:
                move_s = dp["scatter_s"] + dp["sizes_s"] + dp["pack_s"] + dp["gather_s"]
                step_s = t_max / args.steps
                e2e = U * world / (step_s + move_s) / 1e9
                backend = dist.get_backend()
                dp = {"experiment": "configs[4]: %d GPUs x %d blocks of %d bytes (%.1f GiB per GPU, %.1f GiB in all), shards datagen -s0..%d held by rank 0"
                                    % (world, nb, bs, U / 2**30, U * world / 2**30, world - 1),
                      "world_size": dist.get_world_size(), "launched_by": "torchrun / external launcher" if os.environ.get("TORCHELASTIC_RUN_ID") or os.environ.get("GROUP_RANK") else "bench.py --gpus N (launch_ranks) or an external launcher",
                      "backend": backend + (" (RCCL over xGMI)" if backend == "nccl" else " (loopback check of the code path, not a measurement)"),
                      "devices": dp["devices"],
                      "collectives": "scatter of the %d distinct shards from rank 0; all_gather of int32 csize[%d]; payloads packed by LZ4AMD_OP_GATHER and sent to rank 0 with their exact lengths" % (world, nb),
                      "scatter_s": round(dp["scatter_s"], 5), "sizes_s": round(dp["sizes_s"], 5), "pack_s": round(dp["pack_s"], 5), "gather_s": round(dp["gather_s"], 5),
                      "scatter_GBps": round(dp["scatter_bytes"] / dp["scatter_s"] / 1e9, 2) if dp["scatter_s"] > 0 else None,
                      "gather_GBps": round(dp["gather_bytes"] / dp["gather_s"] / 1e9, 2) if dp["gather_s"] > 0 else None,
                      "kernel_only_GBps": round(bytes_all / t_max / 1e9, 3),
                      "kernel_only_frac_of_hbm_peak": round((U + C) * 2 * world / step_s / 1e9 / (HBM_PEAK_GBPS * world), 4),
                      "end_to_end_GBps": round(e2e, 3),
                      "end_to_end_note": "uncompressed bytes of all ranks / (scatter + one compress+decompress step + sizes + pack + gather); the movement is bound by rank 0's xGMI links, not by HBM",
                      "payloads_reassembled_and_decoded": dp["ok"],
                      "note": "unmeasured on multi-GPU hardware until a SCALE record exists"}
            result["data_path"] = dp
        if world == 1 and not args.no_hc and U % (256 << 10) == 0:
            try:
                result["hc"] = bench_hc(ctx, lz4_amd, torch, data, out, stream, args.pct, plan_s["seed"], copy_gbps,
                                        with_cpu=not args.no_cpu_baseline)
            except Exception as e:                           # the side measurement never kills the bench line
                result["hc"] = {"error": str(e)}
        if world == 1 and not args.no_extras and bs == 4 << 20:
            try:
                byc = bench_by_compressibility(ctx, lz4_amd, torch, stream, bs, hints is not None)
                own = {"blocks": nb, "ratio": result["ratio"], "compress_GBps": result["compress_GBps"], "decompress_GBps": result["decompress_GBps"]}
                if "_ref_ratio" in foreign:
                    own["ratio_vs_reference"] = foreign["_ref_ratio"]
                byc["P%d" % args.pct] = own
                result["by_compressibility"] = byc
            except Exception as e:
                result["by_compressibility"] = {"error": str(e)}
        foreign.pop("_ref_ratio", None)
        if world == 1 and not args.no_extras and nb == 256 and bs == 4 << 20:
            try:
                result["shape_64K"] = bench_shape_small(ctx, lz4_amd, torch, data, stream)
            except Exception as e:
                result["shape_64K"] = {"error": str(e)}
            try:
                result["shape_2048"] = bench_shape_2048(ctx, lz4_amd, torch, data, stream, bs, copy_gbps, hints is not None)
            except Exception as e:
                result["shape_2048"] = {"error": str(e)}
            try:
                result["frame"] = bench_frame(lz4_amd, host)
            except Exception as e:
                result["frame"] = {"error": str(e)}
            try:
                result["frame"]["device_resident"] = bench_frame_device(ctx, lz4_amd, torch, data, out, stream, bs, copy_gbps)
            except Exception as e:
                result["frame"]["device_resident"] = {"error": str(e)}
        if world == 1 and not args.no_cpu_baseline:
            cb = cpu_baseline(bs, args.pct, plan_s["seed"])
            result["cpu_baseline"] = cb
            if cb and "ref_comp_bytes_per_unique" in cb:
                ours = sum(csizes[:min(nb, cb["unique_blocks"])])
                result["ratio_vs_reference"] = round(cb["ref_comp_bytes_per_unique"] / ours, 4)
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
