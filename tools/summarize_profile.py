"""Condenses gpurun_out/prof_<tag>/ (tools/profile_round.sh) into profiles/<prefix>_<tag>_*.  The PMC byte
totals follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE is reported in KiB and at
half the bytes of coalesced streaming reads on gfx950 (x2), WRITE_SIZE in KiB, exact."""
import collections, csv, json, os, re, shutil, sys

tag, prefix, key = sys.argv[1], sys.argv[2], sys.argv[3]   # key into profiles/traffic.json
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", "prof_" + tag)
dst = os.path.join(root, "profiles")
shutil.copy(os.path.join(src, "stats", "stats_kernel_stats.csv"), os.path.join(dst, f"{prefix}_{tag}_kernel_stats.csv"))
vals = {}
for kind, ctr in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(os.path.join(src, kind, kind + "_counter_collection.csv"))):
        if r["Counter_Name"] == ctr:
            acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    name = max((k for k in acc if "k_stream_collide" in k or "k_aa_step" in k), key=lambda k: len(acc[k]))
    v = acc[name]
    vals[ctr] = (sum(v) / len(v), len(v), name)
fetch_b = vals["FETCH_SIZE"][0] * 1024 * 2
write_b = vals["WRITE_SIZE"][0] * 1024
line = [l for l in open(os.path.join(src, "bench_stats.log")) if l.startswith("{")][-1]
b = json.loads(line)
alg = b["roofline"]["algorithmic_bytes_per_launch"]
stats = [r for r in csv.DictReader(open(os.path.join(src, "stats", "stats_kernel_stats.csv")))
         if "k_stream_collide" in r["Name"] or "k_aa_step" in r["Name"]]
with open(os.path.join(dst, f"{prefix}_{tag}_pmc_traffic.txt"), "w") as f:
    f.write(f"workload: {b['config']['workload']}\n")
    f.write("rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (tools/profile_round.sh), per dispatch of\n")
    f.write(f"  {vals['FETCH_SIZE'][2][:90]}\n")
    f.write(f"FETCH_SIZE avg {vals['FETCH_SIZE'][0]:.1f} KiB over {vals['FETCH_SIZE'][1]} dispatches -> x1024 x2 (gfx950 half-count) = {fetch_b:.0f} B\n")
    f.write(f"WRITE_SIZE avg {vals['WRITE_SIZE'][0]:.1f} KiB over {vals['WRITE_SIZE'][1]} dispatches -> x1024 = {write_b:.0f} B\n")
    f.write(f"traffic per launch = {fetch_b + write_b:.0f} B; algorithmic (152 B x {b['roofline']['fluid_cells_per_launch']} fluid cells) = {alg} B; ratio {(fetch_b + write_b) / alg:.4f}\n")
    f.write(f"reads / algorithmic reads = {fetch_b / (alg / 2):.4f}; writes / algorithmic writes = {write_b / (alg / 2):.4f}\n")
    # k_stream_collide<PRESET, 1, TUNE> is the moments-only pass bench.py runs once after the timed region
    def inst(name):
        m = re.search(r"(k_stream_collide|k_aa_step)<([^>]*)>", name)
        return (m.group(1), m.group(2)) if m else (name, "")
    stats = [r for r in stats if not (inst(r["Name"])[0] == "k_stream_collide" and inst(r["Name"])[1].split(",")[1].strip() == "1")]
    for r in stats:
        f.write(f"kernel-trace stats {inst(r['Name'])[0]}<{inst(r['Name'])[1]}>: {r['Calls']} calls, average {float(r['AverageNs']) / 1e3:.2f} us (min {float(r['MinNs']) / 1e3:.2f}, max {float(r['MaxNs']) / 1e3:.2f}); in-bench HIP events: {b['roofline']['launch_ms'] * 1e3:.2f} us\n")
    if len(stats) > 1:  # AA: the even and the odd phase are two instantiations; bench.py times their mix
        calls = sum(int(r["Calls"]) for r in stats)
        mean = sum(int(r["Calls"]) * float(r["AverageNs"]) for r in stats) / calls
        f.write(f"kernel-trace stats, all {calls} step-kernel launches: average {mean / 1e3:.2f} us\n")
    f.write("bench line of the kernel-trace pass: " + line)
tp = os.path.join(dst, "traffic.json")
t = json.load(open(tp)) if os.path.exists(tp) else {}
t[key] = int(fetch_b + write_b)
json.dump(t, open(tp, "w"), indent=1, sort_keys=True)
print(open(os.path.join(dst, f"{prefix}_{tag}_pmc_traffic.txt")).read())
