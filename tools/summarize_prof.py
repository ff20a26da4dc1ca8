"""Turn the rocprofv3 CSVs under gpurun_out/ into the small summaries committed under profiles/.

usage: python tools/summarize_prof.py <round-tag> <config> <stats_dir> <fetch_dir> <write_dir> [kernel-substr]

Traffic per launch follows MI355X_MICROARCH.md §HBM: FETCH_SIZE / WRITE_SIZE come from separate
--pmc passes, are in KiB, and on gfx950 FETCH_SIZE reports exactly half of a wide (16 B/lane)
coalesced streaming read, so it is doubled; WRITE_SIZE is exact for streaming stores.
"""
import collections, csv, glob, json, os, sys

# rocprofv3's VGPR_Count on gfx950 wave64 kernels reads HALF the per-lane register count the compiler reports
# (build/resource_usage.txt "VGPRs", the number the occupancy table of MI355X_MICROARCH.md is indexed by)
VGPR_UNIT = "rocprofv3 VGPR_Count as reported: half of the compiler's per-lane VGPR count (build/resource_usage.txt)"

tag, config, stats_dir, fetch_dir, write_dir = sys.argv[1:6]
needle = sys.argv[6] if len(sys.argv) > 6 else "matrix_kernel"
commit = sys.argv[7] if len(sys.argv) > 7 else None  # commit of the tree the counters were collected on
# launches of the kernel per step of the workload: a batch beyond 10 Mi systems goes out as several launches on slices
# (C4: 40 M systems = 4 launches); per-step figures are the sums over them
lps = int(sys.argv[8]) if len(sys.argv) > 8 else 1
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out_dir = os.path.join(ROOT, "profiles")
os.makedirs(out_dir, exist_ok=True)


def one(d, pat):
    g = glob.glob(os.path.join(d, "**", pat), recursive=True)
    assert g, (d, pat)
    return max(g, key=os.path.getmtime)  # gpurun merges runs into the same directory: take the newest


stats = list(csv.DictReader(open(one(stats_dir, "*_kernel_stats.csv"))))
with open(os.path.join(out_dir, f"{tag}_{config}_kernel_stats.csv"), "w") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
    for r in stats:
        w.writerow([r["Name"], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"],
                    r["MaxNs"], r["StdDev"]])
kern = [r for r in stats if needle in r["Name"]][0]


def pmc(d, counter):
    rows = list(csv.DictReader(open(one(d, "*_counter_collection.csv"))))
    vals = [float(r["Counter_Value"]) for r in rows if needle in r["Kernel_Name"] and r["Counter_Name"] == counter]
    meta = [r for r in rows if needle in r["Kernel_Name"]][0]
    return vals, meta


fetch, meta = pmc(fetch_dir, "FETCH_SIZE")
write, _ = pmc(write_dir, "WRITE_SIZE")
fetch_kib = sum(fetch) / len(fetch)
write_kib = sum(write) / len(write)
traffic = fetch_kib * 1024 * 2 + write_kib * 1024
summary = {
    "kernel": kern["Name"], "calls": int(kern["Calls"]), "avg_ns": float(kern["AverageNs"]),
    "min_ns": float(kern["MinNs"]), "max_ns": float(kern["MaxNs"]),
    "vgpr": int(meta["VGPR_Count"]), "vgpr_unit": VGPR_UNIT, "sgpr": int(meta["SGPR_Count"]),
    "lds_bytes": int(meta["LDS_Block_Size"]),
    "scratch": int(meta["Scratch_Size"]), "grid": int(meta["Grid_Size"]), "workgroup": int(meta["Workgroup_Size"]),
    "FETCH_SIZE_KiB_raw": fetch_kib, "WRITE_SIZE_KiB": write_kib,
    "read_bytes_corrected_x2": fetch_kib * 1024 * 2, "write_bytes": write_kib * 1024,
    "traffic_bytes_per_launch": traffic, "pmc_launches": len(fetch), "measured_at_commit": commit,
}
if lps > 1:
    summary["launches_per_step"] = lps
    summary["avg_ns_per_step"] = float(kern["TotalDurationNs"]) / (int(kern["Calls"]) / lps)
    summary["traffic_bytes_per_step"] = (sum(fetch) * 1024 * 2) / (len(fetch) / lps) + (sum(write) * 1024) / (len(write) / lps)
    traffic = summary["traffic_bytes_per_step"]
with open(os.path.join(out_dir, f"{tag}_{config}_summary.json"), "w") as f:
    json.dump(summary, f, indent=1)
tpath = os.path.join(out_dir, "traffic.json")
t = json.load(open(tpath)) if os.path.exists(tpath) else {}
t[config] = {"traffic_bytes_per_launch": traffic, "source": f"profiles/{tag}_{config}_summary.json",
             "measured_at_commit": commit}
if lps > 1:
    t[config]["note"] = f"per step = {lps} launches of the kernel on slices of the batch"
json.dump(t, open(tpath, "w"), indent=1)
print(json.dumps(summary, indent=1))
