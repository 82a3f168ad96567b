#!/usr/bin/env python3
"""Summarise a tools/profile.sh run into profiles/:
   <round>_kernel_stats.csv  -- this repo's kernels from rocprofv3 --kernel-trace --stats
   <round>_pmc_traffic.json  -- FETCH_SIZE / WRITE_SIZE per launch per kernel, with the gfx950 correction
                                (FETCH_SIZE doubled, MI355X_MICROARCH.md "HBM"; the counters are in KiB)
usage: tools/profile_summary.py gpurun_out/<tag> profiles/<round> [workload]"""
import csv, glob, json, os, sys

OURS = ("k_scan", "k_seg", "k_hist", "k_cluster_filter", "k_flagged", "k_cand", "k_apply", "k_tile", "k_unpack", "k_ko",
        "k_pack", "k_consensus", "k_tail", "k_reads", "k_record", "k_phase", "k_lcp", "k_snapshot", "k_job", "k_img")


def ours(name):
    return any(name.startswith(p) or (" " + p) in name for p in OURS)


def short(name):
    return name.replace("void ", "").split("(")[0]


def main(src, dst, workload="c2"):
    st = glob.glob(os.path.join(src, "stats", "**", "*kernel_stats.csv"), recursive=True)[0]
    rows = [r for r in csv.DictReader(open(st)) if ours(r["Name"])]
    with open(dst + "_kernel_stats.csv", "w") as f:
        f.write("# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 10 --warmup 2 --cpu-sample 0 --workload <workload>   (MI355X, workload %s)\n" % workload)
        f.write("# only this repo's kernels are listed; the rest of the trace is torch building the synthetic EGSA\n")
        f.write("Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs,StdDev\n")
        for r in rows:
            f.write('"%s",%s,%s,%s,%s,%s,%s,%s\n' % (r["Name"], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"],
                                                     r["MinNs"], r["MaxNs"], r["StdDev"]))
    pmc = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        fs = glob.glob(os.path.join(src, c, "**", "*counter_collection.csv"), recursive=True)
        if not fs:
            continue
        acc = {}
        for r in csv.DictReader(open(fs[0])):
            if r["Counter_Name"] != c or not ours(r["Kernel_Name"]):
                continue
            acc.setdefault(short(r["Kernel_Name"]), []).append(float(r["Counter_Value"]))
        for k, v in acc.items():
            kib = sum(v) / len(v)
            b = kib * 1024.0 * (2.0 if c == "FETCH_SIZE" else 1.0)
            pmc.setdefault(k, {})[c + "_raw_KiB_per_launch"] = kib
            pmc[k][("fetch" if c == "FETCH_SIZE" else "write") + "_bytes_per_launch"] = b
            pmc[k]["launches"] = len(v)
    for k, v in pmc.items():
        v["hbm_bytes_per_launch"] = v.get("fetch_bytes_per_launch", 0.0) + v.get("write_bytes_per_launch", 0.0)
    json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 bench.py --steps 3 --warmup 1 --cpu-sample 0",
               "corrections": "counter unit KiB; FETCH_SIZE doubled on gfx950 (128-B requests tallied at 64 B)",
               "workload": workload, "kernels": pmc}, open(dst + "_pmc_traffic.json", "w"), indent=1)
    ck = os.path.join(src, "clocks_under_load.txt")
    if os.path.exists(ck):
        # sclk / mclk samples taken by rocm-smi every half second while the profiled bench ran: the box's sustained clocks
        import re
        sc = [int(x) for x in re.findall(r"sclk clock level: \d+: \((\d+)Mhz\)", open(ck).read())]
        mc = [int(x) for x in re.findall(r"mclk clock level: \d+: \((\d+)Mhz\)", open(ck).read())]
        json.dump({"source": "rocm-smi --showclocks sampled every 0.5 s during the rocprofv3 --stats run", "sclk_mhz_samples": len(sc),
                   "sclk_mhz_max": max(sc) if sc else None, "sclk_mhz_median": sorted(sc)[len(sc) // 2] if sc else None,
                   "mclk_mhz_max": max(mc) if mc else None}, open(dst + "_clocks.json", "w"), indent=1)
    bj = os.path.join(src, "bench_under_rocprof.json")
    if os.path.exists(bj) and os.path.getsize(bj):
        open(dst + "_bench_under_rocprof.json", "w").write(open(bj).read())


if __name__ == "__main__":
    main(*sys.argv[1:4])
