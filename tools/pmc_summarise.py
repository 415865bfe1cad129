"""Sum the rocprofv3 --pmc passes written by tools/pmc_passes.sh (+ pmc_icache.sh) per kernel:
python tools/pmc_summarise.py <tag> <out.json> [note].  A counter collected in more than one pass is averaged over passes."""
import collections, csv, glob, json, sys
tag, out = sys.argv[1], sys.argv[2]
note = sys.argv[3] if len(sys.argv) > 3 else ""
per_file = collections.defaultdict(lambda: collections.defaultdict(list))   # kernel -> counter -> [sum in each pass]
launches = collections.Counter()
for f in glob.glob("gpurun_out/%s_*/**/*_counter_collection.csv" % tag, recursive=True):
    tot = collections.defaultdict(lambda: collections.defaultdict(float))
    seen = set()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if "scvx" not in k:
            continue
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "FETCH_SIZE" and r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"]); launches[k] += 1
    for k, cs in tot.items():
        for c, v in cs.items():
            per_file[k][c].append(v)
res = {"note": note, "kernels": {k: dict(sorted((c, sum(v) / len(v)) for c, v in cs.items()), launches_in_fetch_pass=launches[k])
                                 for k, cs in per_file.items()}}
json.dump(res, open(out, "w"), indent=1)
for k, v in res["kernels"].items():
    n = max(1, v["launches_in_fetch_pass"])
    print("%-40s launches %d  FETCH %.1f MiB  WRITE %.1f MiB per launch" % (k, n, v.get("FETCH_SIZE", 0) / 1024 / n, v.get("WRITE_SIZE", 0) / 1024 / n))
