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


def csrc_sha16():
    """Hash of the kernel sources (same function as bench.py's): stamped into traffic.json so
    that bench.py can tell a PMC figure measured on other kernels from a current one.  Run
    this script on the tree the profile was taken on."""
    import hashlib
    d = os.path.join(os.path.dirname(HERE), "stereo-matching_amd", "csrc")
    h = hashlib.sha256()
    for fn in sorted(os.listdir(d)):
        if fn.endswith((".hip", ".h")) or fn == "Makefile":
            with open(os.path.join(d, fn), "rb") as f:
                h.update(fn.encode() + b"\0" + f.read() + b"\0")
    return h.hexdigest()[:16]


def short(name):
    name = name.replace("void stm::(anonymous namespace)::", "").replace("stm::(anonymous namespace)::", "")
    return name.split("(")[0]


def load(path):
    with open(path) as f:
        return list(csv.DictReader(f))


def main():
    tag, trace_dir, fetch_dir, write_dir = sys.argv[1:5]
    prefix = os.environ.get("STM_PROF_PREFIX", "r01")
    # pairs per step of the two workloads in the profiled bench runs (per-pair figures are stored)
    batches = {"teddy": int(sys.argv[5]) if len(sys.argv) > 5 else 64,
               "1080p": int(sys.argv[6]) if len(sys.argv) > 6 else 4}
    trace = load(os.path.join(trace_dir, prefix + "_kernel_trace.csv"))
    agg = collections.defaultdict(list)
    for r in trace:
        k = short(r["Kernel_Name"])
        # (the skewed-tile kernels of the two disparity counts carry the same name: four waves per
        # workgroup at D = 128, eight at D = 256)
        if k.startswith("sgm_skew_"):
            k += " [D=%d]" % (128 if int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", "256"))) == 256 else 256)
        agg[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
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
                k = short(r["Kernel_Name"])
                if k.startswith("sgm_skew_"):
                    k += " [D=%d]" % (128 if int(r.get("Workgroup_Size", r.get("Workgroup_Size_X", "256"))) == 256 else 256)
                per[(k, int(r["Grid_Size"]))].append(float(r["Counter_Value"]))
        counters[name] = {k: sum(v) / len(v) for k, v in per.items()}
    keys = sorted(set(counters["FETCH_SIZE"]) | set(counters["WRITE_SIZE"]))
    with open(os.path.join(HERE, tag + "_hbm_counters.csv"), "w") as f:
        f.write("kernel,grid_size,fetch_kib_raw,fetch_kib_corrected_x2,write_kib\n")
        for k in keys:
            fr = counters["FETCH_SIZE"].get(k, 0.0)
            f.write("%s,%d,%.1f,%.1f,%.1f\n" % (k[0].replace(",", ";"), k[1], fr, 2 * fr,
                                                 counters["WRITE_SIZE"].get(k, 0.0)))

    # HBM bytes per aggregation call: dispatches in order; a call begins with sgm_hwalk_kernel
    # (tiled form: + row sweeps + sgm_tile_kernel) or sgm_ckpt_kernel (paired form: + 4 pair
    # launches + repeat kernels).  Calls whose last kernel is the fused disparity-only variant
    # (sgm_tile_kernel<VPL, 2> / sgm_tile_h2_kernel<2> / sgm_pair_kernel<.., FINAL = 2, ..>) are the
    # ones bench.py times.
    def per_step(vpl):
        out = {}
        for name, d, scale in (("FETCH_SIZE", fetch_dir, 2.0), ("WRITE_SIZE", write_dir, 1.0)):
            rows = [r for r in load(os.path.join(d, prefix + "_counter_collection.csv"))
                    if r["Counter_Name"] == name and "sgm_" in r["Kernel_Name"]]
            rows.sort(key=lambda r: int(r["Dispatch_Id"]))
            groups, cur = [], None
            for r in rows:
                k = short(r["Kernel_Name"])
                if k.startswith(("sgm_ckpt_kernel", "sgm_hwalk_kernel", "sgm_bwalk_kernel")):
                    cur = {"vpl": k, "bytes": 0.0, "fused": False,
                           "form": "tiled (sweeps)" if k.startswith("sgm_hwalk") else
                                   ("tiled (walks)" if k.startswith("sgm_bwalk") else "paired")}
                    groups.append(cur)
                if k.startswith("sgm_skew_rev_kernel"):
                    # (the skewed-tile form: reverse pass, then forward pass; both disparity counts'
                    # kernels carry the same name -- four waves per workgroup at D = 128, eight at 256)
                    wg = int(r.get("Workgroup_Size", r.get("Workgroup_Size_X", "256")))
                    cur = {"vpl": "skew<%d>" % (2 if wg == 256 else 4), "bytes": 0.0, "fused": False,
                           "form": "skewed tiles"}
                    groups.append(cur)
                if cur is None:
                    continue
                cur["bytes"] += float(r["Counter_Value"]) * 1024.0 * scale
                if k.startswith(("sgm_pair_kernel", "sgm_tile_kernel", "sgm_tile_h2_kernel", "sgm_skew_fwd_kernel")):
                    args = [a.strip() for a in k[k.index("<") + 1:k.rindex(">")].split(",")]
                    if k.startswith(("sgm_tile_h2_kernel", "sgm_skew_fwd_kernel")):  # <FINAL>
                        final = args[0]
                    else:
                        final = args[3] if k.startswith("sgm_pair_kernel") and len(args) >= 4 else args[1]
                    if final == "2":
                        cur["fused"] = True
            mine = [g for g in groups if g["vpl"].split("<")[1].split(",")[0].rstrip(">").strip() == str(vpl)]
            out["form"] = mine[-1]["form"] if mine else None
            for kind in (True, False):
                sel = [g["bytes"] for g in mine if g["fused"] == kind]
                out[(name, kind)] = sum(sel) / len(sel) if sel else 0.0
        return out
    traffic = {}
    for wl, vpl in (("teddy", 2), ("1080p", 4)):
        o = per_step(vpl)
        nb = batches[wl]
        traffic[wl + "_per_pair"] = round((o[("FETCH_SIZE", True)] + o[("WRITE_SIZE", True)]) / nb)
        traffic[wl + "_pairs_per_step"] = nb
        traffic[wl + "_form"] = o["form"]
        traffic[wl + "_detail_per_pair"] = {
            "pairs_per_step_when_profiled": nb,
            "fused_fetch_bytes_corrected": round(o[("FETCH_SIZE", True)] / nb),
            "fused_write_bytes": round(o[("WRITE_SIZE", True)] / nb),
            "unfused_fetch_bytes_corrected": round(o[("FETCH_SIZE", False)] / nb),
            "unfused_write_bytes": round(o[("WRITE_SIZE", False)] / nb)}
    traffic["_note"] = ("HBM bytes per stereo pair of the fused aggregation+WTA stage (bench.py "
                        "multiplies by its pairs per step), rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; "
                        "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts 64 B per 128 B request); "
                        "tag " + tag)
    traffic["csrc_sha16"] = csrc_sha16()
    with open(os.path.join(HERE, "traffic.json"), "w") as f:
        json.dump(traffic, f, indent=1)
    print(json.dumps(traffic, indent=1))


if __name__ == "__main__":
    main()
