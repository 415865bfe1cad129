"""Second half of tools/final_profiles.sh: python tools/final_profiles_summarise.py TAG HASH -- gpurun_out/TAG_* -> profiles/TAG_*."""
import collections, csv, glob, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import lib_hash
tag, h = sys.argv[1], sys.argv[2]
go = os.path.join(ROOT, "gpurun_out")
def newest(pattern):
    """gpurun MERGES a run's files into the local gpurun_out/: a directory may still hold an earlier run's files -- only the newest counts"""
    fs = sorted(glob.glob(pattern, recursive=True), key=os.path.getmtime)
    return fs[-1:]


stats = newest(os.path.join(go, tag + "_stats", "**", "*kernel_stats.csv"))
assert stats, "no kernel_stats.csv under gpurun_out/%s_stats" % tag
dst = os.path.join(ROOT, "profiles", "%s_kernel_stats_B8192.csv" % tag)
shutil.copy(stats[0], dst)
with open(dst, "a") as f:
    f.write('"# lib_source_hash %s git_head %s: rocprofv3 --kernel-trace --stats -- python3 tools/pmc_period.py (B = 8192: 2 warm-up + 14 solve_steps of one solve_problem period)",,,,,,,\n' % (h, lib_hash.git_head()))
meta = None
for l in open(os.path.join(go, tag + "_pmc_FETCH_SIZE.log")):
    if l.startswith("PMC_PERIOD "):
        meta = json.loads(l[len("PMC_PERIOD "):])
assert meta, "tools/pmc_period.py did not print its counters"
tot = collections.defaultdict(lambda: collections.defaultdict(float)); launches = collections.Counter()
for G in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in newest(os.path.join(go, "%s_pmc_%s" % (tag, G), "**", "*_counter_collection.csv")):
        seen = set()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            if "scvx" not in k or r["Counter_Name"] != G:
                continue
            tot[k][G] += float(r["Counter_Value"])
            if G == "FETCH_SIZE" and r["Dispatch_Id"] not in seen:
                seen.add(r["Dispatch_Id"]); launches[k] += 1
pm = {"note": "default bench workload (exo, B = 8192): 2 warm-up + one solve_problem period of 14 solve_steps; KiB summed over the launches",
      "lib_source_hash": h, "git_head": lib_hash.git_head(), "period": meta,
      "kernels": {k: dict(v, launches_in_fetch_pass=launches[k]) for k, v in tot.items()}}
json.dump(pm, open(os.path.join(ROOT, "profiles", "%s_pmc_mix_B8192.json" % tag), "w"), indent=1)
ks = [k for k in tot if "socp_kernel" in k]
assert ks, "no socp_kernel records in the PMC passes"
k = ks[0]
traffic = (2.0 * tot[k]["FETCH_SIZE"] + tot[k]["WRITE_SIZE"]) * 1024.0
iters = meta["ipm_iters"] + meta["warmup_ipm_iters"]
solves = meta["solves"] + meta["warmup_solves"]
model = {"bytes_per_ipm_iteration": traffic / iters, "bytes_per_solve": 0.0, "lib_source_hash": h, "git_head": lib_hash.git_head(),
         "lib_build_switches": lib_hash.lib_build_switches(),   # of the .so in the tree (the one the profiled run loaded: it travels with the snapshot)
         "source": "tools/final_profiles.sh %s: (2 x FETCH_SIZE + WRITE_SIZE) of %s over %d launches of the bench mix / the interior-point iterations "
                   "those launches executed (device-side counters); per-solve overheads are inside the per-iteration figure" % (tag, k, launches[k]),
         "calibration": {"traffic_bytes": traffic, "launches": launches[k], "ipm_iterations": iters, "solves": solves,
                         "traffic_bytes_per_launch": traffic / max(launches[k], 1), "bytes_per_solve_of_the_mix": traffic / max(solves, 1)}}
json.dump(model, open(os.path.join(ROOT, "profiles", "%s_k4_traffic_model.json" % tag), "w"), indent=1)
print(json.dumps(model, indent=1))
for row in csv.DictReader(open(stats[0])):
    if "socp" in row["Name"] or "linearize" in row["Name"]:
        print("%-60s calls %s avg %.3f ms" % (row["Name"][:60], row["Calls"], float(row["AverageNs"]) * 1e-6))
