"""Add the kernels' average durations (one-stream rocprofv3 --kernel-trace --stats summary, tools/prof_summary.py format) to a PMC traffic file
(tools/pmc_traffic.py), in place: kernels[name].avg_us / .ms_per_step / .gb_s.  bench.py turns them into the `roofline.hbm_kernels` rows.
usage: python tools/pmc_merge_durations.py gpurun_out/r05_pmc_hbm_traffic.json gpurun_out/r05_z_kernel_stats.txt"""
import json
import sys

traffic_file, stats_file = sys.argv[1], sys.argv[2]
d = json.load(open(traffic_file))
stats = {}
for line in open(stats_file):
    if line.startswith(("#", "calls")) or not line.strip():
        continue
    _calls, _total, avg, _pct, ms, name = line.rstrip("\n").split(",", 5)
    stats[name.split("(")[0]] = (float(avg), float(ms))
hit = 0
for name, v in d["kernels"].items():
    s = stats.get(name.split("(")[0])
    if s is not None:
        v["avg_us"], v["ms_per_step"] = s
        v["gb_s"] = round(v["hbm_bytes_per_launch"] / (s[0] * 1e-6) / 1e9)
        hit += 1
d["durations_from"] = stats_file.split("/")[-1]
json.dump(d, open(traffic_file, "w"), indent=1)
print(f"{traffic_file}: durations for {hit} of {len(d['kernels'])} kernels from {stats_file}")
