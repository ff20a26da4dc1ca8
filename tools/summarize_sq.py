"""Summarise rocprofv3 SQ-counter passes + a --kernel-trace --stats pass of one kernel into profiles/.

usage: python tools/summarize_sq.py <out.json> <kernel-substr> <n_sys> <stats_dir> <pmc_dir> [<pmc_dir> ...] [-- note]

Counters are averaged over the launches of the kernel found in each pass (the newest CSV in each directory,
because gpurun merges successive runs into the same place)."""
import collections, csv, glob, json, os, sys

# rocprofv3's VGPR_Count on gfx950 wave64 kernels reads HALF the per-lane register count the compiler reports
# (build/resource_usage.txt "VGPRs", the number the occupancy table of MI355X_MICROARCH.md is indexed by)
VGPR_UNIT = "rocprofv3 VGPR_Count as reported: half of the compiler's per-lane VGPR count (build/resource_usage.txt)"

args = sys.argv[1:]
note = ""
if "--" in args:
    i = args.index("--")
    note = " ".join(args[i + 1:])
    args = args[:i]
out, needle, n_sys, stats_dir, pmc_dirs = args[0], args[1], int(args[2]), args[3], args[4:]


def newest(d, pat):
    g = glob.glob(os.path.join(d, "**", pat), recursive=True)
    assert g, (d, pat)
    return max(g, key=os.path.getmtime)


stats = [r for r in csv.DictReader(open(newest(stats_dir, "*_kernel_stats.csv"))) if needle in r["Name"]]
kern = max(stats, key=lambda r: float(r["TotalDurationNs"]))
pmc = collections.defaultdict(list)
meta = None
for d in pmc_dirs:
    for r in csv.DictReader(open(newest(d, "*_counter_collection.csv"))):
        if r["Kernel_Name"] == kern["Name"]:
            pmc[r["Counter_Name"]].append(float(r["Counter_Value"]))
            meta = r
c = {k: sum(v) / len(v) for k, v in sorted(pmc.items())}
res = {
    "kernel": kern["Name"], "calls": int(kern["Calls"]), "avg_ns": float(kern["AverageNs"]),
    "min_ns": float(kern["MinNs"]), "max_ns": float(kern["MaxNs"]),
    "vgpr": int(meta["VGPR_Count"]), "vgpr_unit": VGPR_UNIT, "accum_vgpr": int(meta["Accum_VGPR_Count"]),
    "sgpr": int(meta["SGPR_Count"]),
    "lds_bytes": int(meta["LDS_Block_Size"]), "scratch": int(meta["Scratch_Size"]),
    "grid": int(meta["Grid_Size"]), "workgroup": int(meta["Workgroup_Size"]),
    "pmc_per_launch": c,
}
if "SQ_THREAD_CYCLES_VALU" in c and "SQ_ACTIVE_INST_VALU" in c:
    res["lanes_active_per_valu_inst"] = c["SQ_THREAD_CYCLES_VALU"] / (64 * c["SQ_ACTIVE_INST_VALU"])
if "SQ_INSTS_VALU" in c:
    res["wave_valu_insts_per_system"] = c["SQ_INSTS_VALU"] / n_sys
    # 16-lane SIMDs: 4 cycles per wave64 VALU op; 256 CUs x 4 SIMDs at 2.4 GHz
    res["valu_issue_slot_frac"] = c["SQ_INSTS_VALU"] * 4 / (float(kern["AverageNs"]) * 1e-9 * 1024 * 2.4e9)
if note:
    res["note"] = note
if os.environ.get("TDR_PROFILE_COMMIT"):
    res["measured_at_commit"] = os.environ["TDR_PROFILE_COMMIT"]
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
