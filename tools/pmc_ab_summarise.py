"""python tools/pmc_ab_summarise.py TAG [out.json]: per variant of tools/pmc_ab.sh, traffic of the conic kernel = 2 x FETCH_SIZE + WRITE_SIZE
(profiles/r03_stream_ceiling.md) summed over the 16 launches of a pass, with the pass's own iteration counters."""
import collections, csv, glob, json, re, sys
tag = sys.argv[1]
out = {}
n = 0
while glob.glob("gpurun_out/%s_%d_FETCH_SIZE.log" % (tag, n)):
    meta = None
    for l in open("gpurun_out/%s_%d_FETCH_SIZE.log" % (tag, n)):
        if l.startswith("PMC_PERIOD "):
            meta = json.loads(l[len("PMC_PERIOD "):])
    tot = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for G in ("FETCH_SIZE", "WRITE_SIZE"):
        for f in glob.glob("gpurun_out/%s_%d_%s/**/*_counter_collection.csv" % (tag, n, G), recursive=True):
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"].split("(")[0]
                if r["Counter_Name"] == G:
                    tot[k][G] += float(r["Counter_Value"]); 
                    if G == "FETCH_SIZE": cnt[k] += 1
    ks = [k for k in tot if "socp" in k]
    e = dict(meta or {})
    for k in ks:
        fe, wr = tot[k]["FETCH_SIZE"] * 1024.0, tot[k]["WRITE_SIZE"] * 1024.0
        e.update(kernel=k, records=cnt[k], FETCH_bytes=fe, WRITE_bytes=wr, traffic_bytes=2 * fe + wr)
        if meta:
            # the pass holds 16 launches (2 warm-up + 14 counted by the device-side counters); warm-up = 2 cold-ish steps: scale by launches
            e["traffic_per_launch"] = (2 * fe + wr) / 16.0
            if "warmup_ipm_iters" in meta:
                e["bytes_per_ipm_iteration"] = (2 * fe + wr) / (meta["ipm_iters"] + meta["warmup_ipm_iters"])
    out[str(n)] = e
    print(n, json.dumps(e))
    n += 1
if len(sys.argv) > 2:
    json.dump(out, open(sys.argv[2], "w"), indent=1)
