"""Condense rocprofv3 output directories (gpurun_out/...) into small tracked summaries under profiles/.
usage: python tools/summarize_prof.py <trace_dir> <pmc_fetch_dir> <pmc_write_dir> <out_prefix> [kernel_substr]"""
import collections, csv, glob, json, sys

def kernel_stats(d, out):
    f = glob.glob(f"{d}/**/*_kernel_stats.csv", recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    with open(out, "w") as o:
        w = csv.writer(o)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
        for r in rows:
            name = r["Name"]
            if len(name) > 120:
                name = name[:117] + "..."
            w.writerow([name] + [r[k] for k in ("Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev")])
    return rows

def pmc(d, substr):
    f = glob.glob(f"{d}/**/*_counter_collection.csv", recursive=True)[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if substr in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in agg.items()}

if __name__ == "__main__":
    trace, fdir, wdir, prefix = sys.argv[1:5]
    substr = sys.argv[5] if len(sys.argv) > 5 else "batch_decode_kernel"
    rows = kernel_stats(trace, prefix + "_kernel_stats.csv")
    avg_ns = next(float(r["AverageNs"]) for r in rows if substr in r["Name"])
    fetch = pmc(fdir, substr)["FETCH_SIZE"]
    write = pmc(wdir, substr)["WRITE_SIZE"]
    # MI355X_MICROARCH.md (HBM): FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports exactly
    # half of the bytes of a wide (16 B/lane) coalesced streaming read -> doubled; WRITE_SIZE is exact.
    hbm = 2 * fetch[0] * 1024 + write[0] * 1024
    out = {
        "kernel": substr,
        "avg_kernel_ns_under_rocprof": avg_ns,
        "FETCH_SIZE_KiB_raw_avg": fetch[0], "FETCH_SIZE_dispatches": fetch[1],
        "WRITE_SIZE_KiB_raw_avg": write[0], "WRITE_SIZE_dispatches": write[1],
        "correction": "hbm_bytes = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024 (gfx950: FETCH_SIZE counts 128-B requests as 64 B)",
        "hbm_bytes_per_launch": hbm,
    }
    json.dump(out, open(prefix + "_traffic.json", "w"), indent=1)
    print(json.dumps(out, indent=1))
