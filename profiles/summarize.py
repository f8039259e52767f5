#!/usr/bin/env python3
"""Turns rocprofv3 output directories (under gpurun_out/) into the committed summaries.

    python profiles/summarize.py r01 gpurun_out/prof2 gpurun_out/prof2_fetch gpurun_out/prof2_write

* <tag>_kernel_stats.csv   per-kernel calls / total / average / min / max (from --kernel-trace --stats)
* <tag>_hbm_counters.csv   per kernel and grid: FETCH_SIZE / WRITE_SIZE averages (KiB as reported;
                           on gfx950 FETCH_SIZE counts half of a wide coalesced read -- see
                           MI355X_MICROARCH.md, HBM -- the corrected column doubles it)
* traffic.json             HBM bytes per semiglobal aggregation (8 sgm_pass launches + repeats) for
                           the two bench workloads, read by bench.py into roofline.traffic
"""
import collections
import csv
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))


def short(name):
    name = name.replace("void stm::(anonymous namespace)::", "").replace("stm::(anonymous namespace)::", "")
    return name.split("(")[0]


def load(path):
    with open(path) as f:
        return list(csv.DictReader(f))


def main():
    tag, trace_dir, fetch_dir, write_dir = sys.argv[1:5]
    prefix = sys.argv[5] if len(sys.argv) > 5 else "r01"
    trace = load(os.path.join(trace_dir, prefix + "_kernel_trace.csv"))
    agg = collections.defaultdict(list)
    for r in trace:
        agg[short(r["Kernel_Name"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    total = sum(sum(v) for v in agg.values())
    with open(os.path.join(HERE, tag + "_kernel_stats.csv"), "w") as f:
        f.write("kernel,calls,total_us,avg_us,min_us,max_us,percent\n")
        for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
            f.write("%s,%d,%.1f,%.2f,%.2f,%.2f,%.2f\n" % (k.replace(",", ";"), len(v), sum(v) / 1e3,
                                                       sum(v) / len(v) / 1e3, min(v) / 1e3,
                                                       max(v) / 1e3, 100.0 * sum(v) / total))
    counters = {}
    for name, d in (("FETCH_SIZE", fetch_dir), ("WRITE_SIZE", write_dir)):
        per = collections.defaultdict(list)
        for r in load(os.path.join(d, prefix + "_counter_collection.csv")):
            if r["Counter_Name"] == name:
                per[(short(r["Kernel_Name"]), int(r["Grid_Size"]))].append(float(r["Counter_Value"]))
        counters[name] = {k: sum(v) / len(v) for k, v in per.items()}
    keys = sorted(set(counters["FETCH_SIZE"]) | set(counters["WRITE_SIZE"]))
    with open(os.path.join(HERE, tag + "_hbm_counters.csv"), "w") as f:
        f.write("kernel,grid_size,fetch_kib_raw,fetch_kib_corrected_x2,write_kib\n")
        for k in keys:
            fr = counters["FETCH_SIZE"].get(k, 0.0)
            f.write("%s,%d,%.1f,%.1f,%.1f\n" % (k[0].replace(",", ";"), k[1], fr, 2 * fr,
                                                 counters["WRITE_SIZE"].get(k, 0.0)))

    # HBM bytes per aggregation: the SGM launches of one bench step, identified by VPL
    # (sgm_pass_kernel<2,...> = Teddy D=128, <4,...> = 1080p D=256), summed over one step.
    def per_step(vpl):
        fetch = write = 0.0
        steps = None
        for name, d, scale in (("FETCH_SIZE", fetch_dir, 2.0), ("WRITE_SIZE", write_dir, 1.0)):
            rows = [r for r in load(os.path.join(d, prefix + "_counter_collection.csv"))
                    if r["Counter_Name"] == name and ("sgm_" in r["Kernel_Name"])]
            mine = [r for r in rows if ("sgm_pass_kernel<%d," % vpl) in r["Kernel_Name"]]
            firsts = sum(1 for r in mine if short(r["Kernel_Name"]).endswith(", true>"))
            steps = max(firsts, 1)
            tot = sum(float(r["Counter_Value"]) for r in mine) * 1024.0 * scale / steps
            if name == "FETCH_SIZE":
                fetch = tot
            else:
                write = tot
        return fetch + write, fetch, write
    traffic = {}
    for wl, vpl in (("teddy", 2), ("1080p", 4)):
        tot, fe, wr = per_step(vpl)
        traffic[wl] = round(tot)
        traffic[wl + "_detail"] = {"fetch_bytes_corrected": round(fe), "write_bytes": round(wr)}
    traffic["_note"] = ("HBM bytes per semiglobal aggregation of one bench step (Teddy: 16 pairs, "
                        "1080p: 2 pairs), rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; "
                        "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts 64 B per 128 B request); "
                        "tag " + tag)
    with open(os.path.join(HERE, "traffic.json"), "w") as f:
        json.dump(traffic, f, indent=1)
    print(json.dumps(traffic, indent=1))


if __name__ == "__main__":
    main()
