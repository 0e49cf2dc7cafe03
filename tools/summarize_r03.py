"""gpurun_out/prof_r03/ (tools/prof_r03.sh) -> the small files kept under profiles/: per-kernel stats CSVs, the C2 traffic
JSON, and per secondary workload a PMC text summary + the JSON bench.py cites in its `roofline` blocks.
usage: python tools/summarize_r03.py <prof dir> <out dir>"""
import collections, csv, glob, json, os, sys

src, dst = sys.argv[1], sys.argv[2]
os.makedirs(dst, exist_ok=True)


def kernel_stats(d, out):
    f = glob.glob(f"{d}/**/*_kernel_stats.csv", recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    with open(out, "w") as o:
        w = csv.writer(o)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
        for r in rows:
            name = r["Name"] if len(r["Name"]) <= 120 else r["Name"][:117] + "..."
            w.writerow([name] + [r[k] for k in ("Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev")])
    return rows


def pmc(d, substr):
    fs = glob.glob(f"{d}/**/*_counter_collection.csv", recursive=True)
    agg = collections.defaultdict(list)
    for f in fs:
        for r in csv.DictReader(open(f)):
            if substr in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in agg.items()}


# ---- C2 ----
rows = kernel_stats(f"{src}/c2_trace", f"{dst}/r03_c2_decode_kernel_stats.csv")
sub = "decode_mfma16_kernel"
avg_ns = next(float(r["AverageNs"]) for r in rows if sub in r["Name"])
fetch, write = pmc(f"{src}/c2_fetch", sub)["FETCH_SIZE"], pmc(f"{src}/c2_write", sub)["WRITE_SIZE"]
json.dump({"kernel": sub, "avg_kernel_ns_under_rocprof": avg_ns, "FETCH_SIZE_KiB_raw_avg": fetch[0],
           "FETCH_SIZE_dispatches": fetch[1], "WRITE_SIZE_KiB_raw_avg": write[0], "WRITE_SIZE_dispatches": write[1],
           "correction": "hbm_bytes = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024 (gfx950: FETCH_SIZE counts 128-B requests as 64 B)",
           "hbm_bytes_per_launch": 2 * fetch[0] * 1024 + write[0] * 1024},
          open(f"{dst}/r03_c2_decode_traffic.json", "w"), indent=1)

# ---- C3 / C4 ----
for w, sub, flops in (("c3", "batch_prefill_fp8_kernel", 16 * (2 * 8192 - 2048) * 2048 * 32 * 2 * 128),
                      ("c4", "group_gemm_fp8_big_kernel", 2 * 8 * 4096 * 14336 * 4096)):
    rows = kernel_stats(f"{src}/{w}_trace", f"{dst}/r03_{w}_kernel_stats.csv")
    # the GEMM launches two instantiations per call (hardware-scale and fold); the one not selected returns at once
    cand = [r for r in rows if sub in r["Name"]]
    main = max(cand, key=lambda r: float(r["AverageNs"]))
    avg_ns = float(main["AverageNs"])
    c = {}
    for i in (1, 2, 3, 4):
        for k, v in pmc(f"{src}/{w}_pmc{i}", sub).items():
            c[k] = v
    def val(k):
        return c[k][0] if k in c else float("nan")
    # with two GEMM instantiations per call the per-kernel averages mix a real launch and an empty one: use sums per call
    ndisp = c["SQ_WAVE_CYCLES"][1] if "SQ_WAVE_CYCLES" in c else 1
    per_call = 2 if w == "c4" else 1
    scale = per_call  # average over dispatches x dispatches per call = per call
    grbm = val("GRBM_GUI_ACTIVE") * scale
    clock_ghz = grbm / 8 / avg_ns  # rocprofv3 sums the 8 XCDs
    mfma_busy = val("SQ_VALU_MFMA_BUSY_CYCLES") * scale / (1024 * avg_ns * clock_ghz)
    rd_req = val("TCC_EA0_RDREQ_sum") * scale
    wr_req = val("TCC_EA0_WRREQ_sum") * scale
    hit, miss = val("TCC_HIT_sum") * scale, val("TCC_MISS_sum") * scale
    out = {
        "kernel": main["Name"][:100], "avg_kernel_ns_under_rocprof": avg_ns,
        "achieved_TFLOPs_under_rocprof": flops / avg_ns / 1e3,
        "clock_ghz": clock_ghz, "clock_note": "GRBM_GUI_ACTIVE / 8 XCDs / kernel time",
        "mfma_busy": mfma_busy, "mfma_busy_note": "SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel cycles)",
        "valu_per_mfma": (val("SQ_INSTS_VALU") - val("SQ_INSTS_MFMA")) / val("SQ_INSTS_MFMA"),
        "l2_hit_rate": hit / (hit + miss), "l2_requests": val("TCC_REQ_sum") * scale,
        "traffic": rd_req * 128 + wr_req * 64,
        "traffic_note": "fabric-side bytes per launch: TCC_EA0_RDREQ x 128 B (upper bound: requests are 64 or 128 B) + "
                        "TCC_EA0_WRREQ x 64 B; includes Infinity-Cache hits",
        "wave_cycles_share": {"active": val("SQ_ACTIVE_INST_ANY") / val("SQ_WAVE_CYCLES"),
                              "wait_any": val("SQ_WAIT_ANY") / val("SQ_WAVE_CYCLES"),
                              "wait_inst_any": val("SQ_WAIT_INST_ANY") / val("SQ_WAVE_CYCLES")},
        "lds_bank_conflict_share": val("SQ_LDS_BANK_CONFLICT") / max(val("SQ_LDS_IDX_ACTIVE"), 1.0),
        "dispatches_averaged": ndisp, "harness": "tools/prof_secondary_once.py (bench.py inputs, 256 MB L2 flush per launch)",
    }
    json.dump(out, open(f"{dst}/r03_{w}_pmc.json", "w"), indent=1)
    with open(f"{dst}/r03_{w}_pmc.txt", "w") as o:
        for k, v in sorted(c.items()):
            o.write(f"{k:32s} n={v[1]:3d} avg={v[0]:.4g}\n")
        o.write("# " + json.dumps(out) + "\n")
    print(w, json.dumps(out, indent=1))
