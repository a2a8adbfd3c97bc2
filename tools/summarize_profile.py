"""Condenses gpurun_out/prof_<tag>/ (tools/profile_round.sh) into profiles/<prefix>_<tag>_*.  The PMC byte
totals follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE is reported in KiB and at
half the bytes of coalesced streaming reads on gfx950 (x2), WRITE_SIZE in KiB, exact.
usage: summarize_profile.py <tag> <prefix>"""
import collections, csv, json, os, re, shutil, sys

tag, prefix = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", "prof_" + tag)
dst = os.path.join(root, "profiles")
shutil.copy(os.path.join(src, "stats", "stats_kernel_stats.csv"), os.path.join(dst, f"{prefix}_{tag}_kernel_stats.csv"))
line = [l for l in open(os.path.join(src, "bench_stats.log")) if l.startswith("{")][-1]
b = json.loads(line)
dominant = b["roofline"]["kernel"].split()[0]          # k_fused2 | k_stream_collide | k_aa_step
STEP_KERNELS = ("k_fused2", "k_stream_collide", "k_aa_step")


def inst(name):
    m = re.search(r"(k_fused2|k_stream_collide|k_aa_step)<([^>]*)>", name)
    return (m.group(1), m.group(2)) if m else (name, "")


def is_moments_only(name):   # k_stream_collide<PRESET, 1, TUNE>: the on-demand rho/u pass bench.py triggers once after the timed region
    k, a = inst(name)
    return k == "k_stream_collide" and a.split(",")[1].strip() == "1"


vals = {}
per_kernel = {}
for kind, ctr in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(os.path.join(src, kind, kind + "_counter_collection.csv"))):
        if r["Counter_Name"] == ctr:
            acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    name = max((k for k in acc if dominant in k and not is_moments_only(k)), key=lambda k: len(acc[k]))
    v = acc[name]
    vals[ctr] = (sum(v) / len(v), len(v), name)
    for k, v in acc.items():
        if any(s in k for s in STEP_KERNELS) and not is_moments_only(k):
            per_kernel.setdefault(k, {})[ctr] = (sum(v) / len(v), len(v))
fetch_b = vals["FETCH_SIZE"][0] * 1024 * 2
write_b = vals["WRITE_SIZE"][0] * 1024
alg = b["roofline"]["algorithmic_bytes_per_launch"]
stats = [r for r in csv.DictReader(open(os.path.join(src, "stats", "stats_kernel_stats.csv")))
         if any(s in r["Name"] for s in STEP_KERNELS) and not is_moments_only(r["Name"])]
with open(os.path.join(dst, f"{prefix}_{tag}_pmc_traffic.txt"), "w") as f:
    f.write(f"workload: {b['config']['workload']}\n")
    f.write("rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (tools/profile_round.sh), per dispatch of\n")
    f.write(f"  {vals['FETCH_SIZE'][2][:110]}\n")
    f.write(f"FETCH_SIZE avg {vals['FETCH_SIZE'][0]:.1f} KiB over {vals['FETCH_SIZE'][1]} dispatches -> x1024 x2 (gfx950 half-count) = {fetch_b:.0f} B\n")
    f.write(f"WRITE_SIZE avg {vals['WRITE_SIZE'][0]:.1f} KiB over {vals['WRITE_SIZE'][1]} dispatches -> x1024 = {write_b:.0f} B\n")
    tail = b["roofline"].get("tail_cell_updates_per_launch", 0)   # the rim's first step riding in the fused launch: one update per cell
    if tail:
        f.write(f"traffic per launch = {fetch_b + write_b:.0f} B; algorithmic (76 B x {b['roofline']['cell_updates_per_launch'] - tail} updates of the blocks' cells "
                f"+ 152 B x {tail} updates of the rim's first step in the launch's tail) = {alg} B; ratio {(fetch_b + write_b) / alg:.4f}\n")
    else:
        f.write(f"traffic per launch = {fetch_b + write_b:.0f} B; algorithmic ({b['roofline']['bytes_per_cell_update']} B x {b['roofline']['cell_updates_per_launch']} cell updates) = {alg} B; ratio {(fetch_b + write_b) / alg:.4f}\n")
    f.write(f"reads / algorithmic reads = {fetch_b / (alg / 2):.4f}; writes / algorithmic writes = {write_b / (alg / 2):.4f}\n")
    if dominant == "k_fused2":
        cells = (b["roofline"]["cell_updates_per_launch"] - tail) / 2 + tail
        f.write(f"per cell the launch touches (two updates; one for the cells of a rim tail): {fetch_b / cells:.1f} B read + {write_b / cells:.1f} B written "
                f"(one read + one write of the 19 populations = 76 + 76 B; the rest of the reads is the tile halo and the two lead-in layers of a z chunk)\n")
    for k, d in sorted(per_kernel.items()):
        if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
            f.write(f"  {inst(k)[0]}<{inst(k)[1]}>: {d['FETCH_SIZE'][1]} dispatches, avg read {d['FETCH_SIZE'][0] * 2048:.0f} B, avg written {d['WRITE_SIZE'][0] * 1024:.0f} B\n")
    for r in stats:
        f.write(f"kernel-trace stats {inst(r['Name'])[0]}<{inst(r['Name'])[1]}>: {r['Calls']} calls, average {float(r['AverageNs']) / 1e3:.2f} us (min {float(r['MinNs']) / 1e3:.2f}, max {float(r['MaxNs']) / 1e3:.2f})\n")
    f.write(f"in-bench HIP events, {dominant}: {b['roofline']['launch_ms'] * 1e3:.2f} us per launch\n")
    same = [r for r in stats if inst(r["Name"])[0] == dominant]
    if len(same) > 1:  # AA: the even and the odd phase are two instantiations; bench.py times their mix
        calls = sum(int(r["Calls"]) for r in same)
        mean = sum(int(r["Calls"]) * float(r["AverageNs"]) for r in same) / calls
        f.write(f"kernel-trace stats, all {calls} {dominant} launches: average {mean / 1e3:.2f} us\n")
    f.write("bench line of the kernel-trace pass: " + line)

# machine-readable twin for bench.py's `roofline.traffic` / `hbm_frac` (bench.py counter_traffic()): quoted there only
# while the kernel sources still hash to what this profile was taken on
sys.path.insert(0, root)
import bench  # noqa: E402
dom_stats = [r for r in stats if r["Name"] == vals["FETCH_SIZE"][2]]
with open(os.path.join(dst, f"{prefix}_{tag}_traffic.json"), "w") as f:
    json.dump({"kernel": dominant, "kernel_symbol": "%s<%s>" % inst(vals["FETCH_SIZE"][2]), "workload": b["config"]["workload"],
               "cell_updates_per_launch": b["roofline"]["cell_updates_per_launch"],
               "first_step_loads": b["roofline"].get("first_step_loads"),
               "fetch_bytes": fetch_b, "write_bytes": write_b, "dispatches": vals["FETCH_SIZE"][1],
               "launch_us": round(float(dom_stats[0]["AverageNs"]) / 1e3, 2) if dom_stats else None,
               "corrections": "FETCH_SIZE KiB x1024 x2 (gfx950 counts wide streaming reads at half), WRITE_SIZE KiB x1024; separate --pmc passes",
               "source_hashes": bench.source_hashes()}, f, indent=1)
    f.write("\n")
print(open(os.path.join(dst, f"{prefix}_{tag}_pmc_traffic.txt")).read())
