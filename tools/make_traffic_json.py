"""Builds profiles/<tag>_traffic.json from the summaries tools/collect_profiles.sh wrote
(<dir>/<tag>_pmc_FETCH_SIZE.txt, _pmc_WRITE_SIZE.txt, _calib_*.txt, _kernel_stats.csv,
bench_kernel_trace.log).  HBM traffic per launch = corr_fetch x FETCH_SIZE + corr_write x WRITE_SIZE (KB),
the corrections taken from the calibration stream of known size (tools/traffic_calib.hip: 1 GiB)."""
import csv, json, re, sys

d, tag = sys.argv[1], sys.argv[2]

def per_dispatch(path, kernel, counter):
    lines = open(path).read().splitlines()
    for i, l in enumerate(lines):
        if l.startswith(kernel + " dispatches"):
            for l2 in lines[i + 1:i + 4]:
                m = re.match(r"\s+%s\s+\S+\s+per-dispatch (\S+)" % counter, l2)
                if m:
                    return float(m.group(1)), int(l.split()[-1])
    return None, 0

GIB_KB = 1048576.0
cal_r, _ = per_dispatch("%s/%s_calib_FETCH_SIZE.txt" % (d, tag), "calib_read_dword", "FETCH_SIZE")
cal_w, _ = per_dispatch("%s/%s_calib_WRITE_SIZE.txt" % (d, tag), "calib_write_dword", "WRITE_SIZE")
corr_r, corr_w = GIB_KB / cal_r, GIB_KB / cal_w
stats = {}
for row in csv.DictReader(open("%s/%s_kernel_stats.csv" % (d, tag))):
    n = row["Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    stats[n] = (int(row["Calls"]), float(row["AverageNs"]) / 1e3)
bench = None
for l in open("%s/bench_kernel_trace.log" % d):
    if l.startswith('{"metric"'):
        bench = json.loads(l)

def kernel(name):
    f, nf = per_dispatch("%s/%s_pmc_FETCH_SIZE.txt" % (d, tag), name, "FETCH_SIZE")
    w, nw = per_dispatch("%s/%s_pmc_WRITE_SIZE.txt" % (d, tag), name, "WRITE_SIZE")
    calls, avg_us = stats.get(name, (0, 0.0))
    return {"fetch_size_kb_per_launch_raw": f, "write_size_kb_per_launch_raw": w,
            "traffic_bytes_per_launch": 1024.0 * (corr_r * f + corr_w * w),
            "rocprof_avg_launch_us": avg_us, "rocprof_calls": calls}

form = 2 if "k_relax<0, 0, false, 2>" in stats else 1      # 8 waves per tile visit and 512 workgroups, or 16 and 256
main = kernel("k_relax<0, 0, false, %d>" % form)
out = {
    "kernel": "k_relax<FD,LOWER,resident> = k_relax<0, 0, false, %d> in %s_kernel_stats.csv (%s): the resident lowering kernel, one launch per plan (the whole lowering phase); the launch bench.py brackets with HIP events" % (
        form, tag, "8 waves per tile visit, 512 workgroups" if form == 2 else "16 waves per tile visit, 256 workgroups"),
    "fetch_correction": corr_r, "write_correction": corr_w,
    "calibration": "tools/traffic_calib.hip: 1 GiB dword stream -> FETCH_SIZE %.0f KB, WRITE_SIZE %.0f KB" % (cal_r, cal_w),
    "command": "python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes; tools/collect_profiles.sh)",
    "other_instantiations": {
        "k_relax<0, 0, false, 0> (lowering, launch chain, short queues: after the resident kernel and for the replans the block kernel leaves)": kernel("k_relax<0, 0, false, 0>"),
        "k_relax<0, 1, false, 0> (invalidation, launch chain)": kernel("k_relax<0, 1, false, 0>"),
        "k_replan_region<0> (block-resident replan: both phases of a replan in one workgroup)": kernel("k_replan_region<0>"),
    },
}
out.update(main)
reg = out["other_instantiations"]["k_replan_region<0> (block-resident replan: both phases of a replan in one workgroup)"]
out["region_traffic_bytes_per_launch"] = reg["traffic_bytes_per_launch"]      # bench.py: roofline_replans.traffic
if bench and "roofline_replans" in bench:
    out["region_bench_avg_launch_us_same_run"] = bench["roofline_replans"]["avg_launch_us"]
    out["region_traffic_over_algorithmic"] = reg["traffic_bytes_per_launch"] / bench["roofline_replans"]["algorithmic_bytes_per_launch"]
if bench and "roofline" in bench:
    out["bench_avg_launch_us_same_run"] = bench["roofline"]["avg_launch_us"]
    out["algorithmic_bytes_per_launch_same_run"] = bench["roofline"]["algorithmic_bytes_per_launch"]
    out["traffic_over_algorithmic"] = out["traffic_bytes_per_launch"] / bench["roofline"]["algorithmic_bytes_per_launch"]
print(json.dumps(out, indent=1))
