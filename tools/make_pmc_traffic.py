#!/usr/bin/env python3
"""Build profiles/pmc_traffic.json from one tools/gpu_round.sh visit (gpurun_out/<tag>_pmc_*), and copy the
summaries the numbers come from into profiles/<round>_*.  Usage: make_pmc_traffic.py <tag> <name>  (e.g. r02e r02_v4)

Units: rocprofv3's FETCH_SIZE / WRITE_SIZE are KiB per dispatch.  Calibration rides in the same passes: the
stream-copy kernel (1 GiB in, 1 GiB out, 16 B/lane) reports FETCH_SIZE = 0.5 GiB and WRITE_SIZE = 1.0 GiB, i.e.
the guide's gfx950 x2 on wide reads and x1 on writes; the factors below are derived from that kernel's rows."""
import json, os, re, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench

def counters(path):
    out = {}
    for line in open(path):
        m = re.match(r"\s+(lz4amd_k_\w+)\(.*?\s+(FETCH_SIZE|WRITE_SIZE)\s+([0-9.]+)\s+\(n=(\d+)\)", line)
        if m:
            out[m.group(1)] = float(m.group(3))
        m = re.match(r"\s+(?:void )?(lz4amd_k_stream_copy)<.*?\s+(FETCH_SIZE|WRITE_SIZE)\s+([0-9.]+)\s+\(n=(\d+)\)", line)      # (the calibration kernel's shapes all move 1 GiB)
        if m and m.group(1) not in out:
            out[m.group(1)] = float(m.group(3))
    return out

def main():
    tag, name = sys.argv[1], sys.argv[2]
    g = os.path.join(ROOT, "gpurun_out")
    files = {"kernel_stats": f"{tag}_prof/{tag}_results.txt", "pmc_fetch": f"{tag}_pmc_fetch/{tag}f_results.txt",
             "pmc_write": f"{tag}_pmc_write/{tag}w_results.txt", "pmc_sq1": f"{tag}_pmc_sq1/{tag}s1_results.txt",
             "pmc_sq2": f"{tag}_pmc_sq2/{tag}s2_results.txt", "hc_kernel_stats": f"{tag}_prof_hc/{tag}hc_results.txt",
             "hc_pmc_fetch": f"{tag}_pmc_fetch_hc/{tag}hf_results.txt", "hc_pmc_write": f"{tag}_pmc_write_hc/{tag}hw_results.txt",
             "hc_pmc_sq1": f"{tag}_pmc_sq1_hc/{tag}hs_results.txt",
             "nohints_kernel_stats": f"{tag}_prof_nohints/{tag}nh_results.txt", "nohints_pmc_fetch": f"{tag}_pmc_fetch_nohints/{tag}nhf_results.txt",
             "nohints_pmc_write": f"{tag}_pmc_write_nohints/{tag}nhw_results.txt", "nohints_pmc_sq1": f"{tag}_pmc_sq1_nohints/{tag}nhs_results.txt",
             "refdec_kernel_stats": f"{tag}_prof_refdec/{tag}rd_results.txt", "refdec_pmc_fetch": f"{tag}_pmc_fetch_refdec/{tag}rdf_results.txt",
             "refdec_pmc_write": f"{tag}_pmc_write_refdec/{tag}rdw_results.txt", "refdec_pmc_sq1": f"{tag}_pmc_sq1_refdec/{tag}rds_results.txt",
             "frame_kernel_stats": f"{tag}_prof_frame/{tag}fr_results.txt"}
    for k, f in files.items():
        if os.path.exists(os.path.join(g, f)):
            shutil.copy(os.path.join(g, f), os.path.join(ROOT, "profiles", f"{name}_rocprof_{k}.txt"))
    if os.path.exists(os.path.join(g, f"{tag}_bench.json")):
        shutil.copy(os.path.join(g, f"{tag}_bench.json"), os.path.join(ROOT, "profiles", f"{name}_bench.json"))
    fe, wr = counters(os.path.join(g, files["pmc_fetch"])), counters(os.path.join(g, files["pmc_write"]))
    gib_kib = float(1 << 20)
    fcorr = gib_kib / fe["lz4amd_k_stream_copy"] if "lz4amd_k_stream_copy" in fe else 2.0
    wcorr = gib_kib / wr["lz4amd_k_stream_copy"] if "lz4amd_k_stream_copy" in wr else 1.0
    doc = {"kernel_sources_sha": {k: bench.kernel_sources_sha(k) for k in bench.KERNEL_FILES},
           "source": f"profiles/{name}_rocprof_pmc_fetch.txt, _pmc_write.txt (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, "
                     f"bench.py --no-hc --no-extras: configs[1]); compress_hc: profiles/{name}_rocprof_hc_pmc_fetch.txt / _write.txt "
                     "(tools/prof_hc.py 4096 262144 60 9: configs[3])",
           "note": "KiB per dispatch; corrections calibrated on the stream-copy kernel of the same passes (1 GiB read + 1 GiB "
                   "written with 16 B/lane accesses), applied to every kernel: all of them read HBM with 16 B/lane loads",
           "fetch_correction": round(fcorr, 4), "write_correction": round(wcorr, 4)}
    for key, kern in (("compress", "lz4amd_k_compress"), ("decompress", "lz4amd_k_decompress"), ("xxh32", "lz4amd_k_xxh32")):
        if kern in fe and kern in wr:
            doc[key] = {"FETCH_SIZE_KiB": fe[kern], "WRITE_SIZE_KiB": wr[kern],
                        "hbm_bytes_per_launch": int((fe[kern] * fcorr + wr[kern] * wcorr) * 1024)}
    hf, hw = os.path.join(g, files["hc_pmc_fetch"]), os.path.join(g, files["hc_pmc_write"])
    if os.path.exists(hf) and os.path.exists(hw):
        a, b = counters(hf).get("lz4amd_k_compress_hc"), counters(hw).get("lz4amd_k_compress_hc")
        if a and b:
            doc["compress_hc"] = {"FETCH_SIZE_KiB": a, "WRITE_SIZE_KiB": b, "hbm_bytes_per_launch": int((a * fcorr + b * wcorr) * 1024)}
    # the decoder on plain LZ4 blocks (no tables): the step's own blocks (bench.py --no-hints) and blocks the reference compressed (tools/prof_refdec.py)
    for key, fk, wk in (("decompress_without_tables", "nohints_pmc_fetch", "nohints_pmc_write"), ("decompress_reference_input", "refdec_pmc_fetch", "refdec_pmc_write")):
        a, b = os.path.join(g, files[fk]), os.path.join(g, files[wk])
        if os.path.exists(a) and os.path.exists(b):
            x, y = counters(a).get("lz4amd_k_decompress"), counters(b).get("lz4amd_k_decompress")
            if x and y:
                doc[key] = {"FETCH_SIZE_KiB": x, "WRITE_SIZE_KiB": y, "hbm_bytes_per_launch": int((x * fcorr + y * wcorr) * 1024),
                            "source": f"profiles/{name}_rocprof_{fk}.txt, _{wk.split('_', 1)[1]}.txt"}
    doc["canonical_set"] = f"profiles/{name}_* (one tools/gpu_round.sh visit on the kernel sources named above); older r0N_* files are earlier rounds' records"
    # VALU issue: SQ_INSTS_VALU wave-instructions per dispatch and the kernel's average duration in the same pass.  A SIMD's
    # VALU pipe takes one wave64 instruction per ~4.2 cycles in mixed code (measured: profiles/r05_valu_issue.txt; 2.4 only for runs
    # of plain add / logic / shift), so instructions x 4.2 / (CUs x 4 SIMDs) are the pipe-cycles each SIMD spends; against duration x 2.4 GHz
    # (the engine's top clock: the fraction is a lower bound) that is how full the pipes are.
    sq = os.path.join(g, files["pmc_sq1"])
    if os.path.exists(sq):
        dur, valu = {}, {}
        for line in open(sq):
            m = re.match(r"\s+(lz4amd_k_\w+)\(.*?\s+(\d+)\s+([0-9.]+)\s+([0-9.]+)\s+([0-9.]+)\s*$", line)
            if m:
                dur[m.group(1)] = float(m.group(4))
            m = re.match(r"\s+(lz4amd_k_\w+)\(.*?\s+SQ_INSTS_VALU\s+([0-9.]+)\s+\(n=", line)
            if m:
                valu[m.group(1)] = float(m.group(2))
        for key, kern in (("compress", "lz4amd_k_compress"), ("decompress", "lz4amd_k_decompress")):
            if key in doc and kern in dur and kern in valu:
                pipe = valu[kern] * 4.2 / (256 * 4)
                doc[key]["valu"] = {"SQ_INSTS_VALU": valu[kern], "kernel_us_in_that_pass": dur[kern],
                                    "pipe_cycles_per_simd": int(pipe), "frac_of_kernel_cycles_at_2.4GHz": round(pipe / (dur[kern] * 2400.0), 3),
                                    "source": f"profiles/{name}_rocprof_pmc_sq1.txt"}
    json.dump(doc, open(os.path.join(ROOT, "profiles", "pmc_traffic.json"), "w"), indent=1)
    json.dump(doc, open(os.path.join(ROOT, "profiles", f"{name}_pmc_traffic.json"), "w"), indent=1)
    print(json.dumps(doc, indent=1))

main()
