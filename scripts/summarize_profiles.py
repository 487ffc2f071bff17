#!/usr/bin/env python3
"""Condenses rocprofv3 output under gpurun_out/ into small committed files under profiles/.

  kernel stats : gpurun_out/prof_<tag>/**/**_kernel_stats.csv  -> profiles/<round>_<tag>_kernel_stats.csv
  PMC passes   : gpurun_out/pmc_<tag>_<COUNTER>/**/*_counter_collection.csv -> profiles/<round>_<tag>_pmc.json
                 (per kernel, per counter: launches + average per launch; FETCH_SIZE / WRITE_SIZE are
                 in KiB as rocprofv3 reports them, plus hbm_read_bytes = FETCH_SIZE * 1024 * 2 -- the
                 gfx950 half-count correction of MI355X_MICROARCH.md for wide coalesced streams -- and
                 hbm_write_bytes = WRITE_SIZE * 1024)
usage: scripts/summarize_profiles.py <round> <tag> [<tag> ...]
"""
import collections
import csv
import glob
import json
import pathlib
import shutil
import sys

ROOT = pathlib.Path(__file__).resolve().parent.parent
OUT = ROOT / "profiles"


def short(name):
    for key, tag in (("eval_chain_kernel", "scan_eval_chain"), ("eval_spec_kernel", "scan_eval_spec"), ("eval_generic_kernel", "scan_eval_generic"),
                     ("reduce_totals_kernel", "reduce_totals"), ("merge_slots_kernel", "merge_slots"), ("append_range_kernel", "append_range"), ("probe_kernel", "index_probe")):
        if key in name:
            if key in ("eval_spec_kernel", "eval_chain_kernel", "eval_generic_kernel"):
                return tag + name[name.index("<"):name.index(">") + 1].replace(" ", "")
            return tag
    return None


def main():
    rnd, tags = sys.argv[1], sys.argv[2:]
    OUT.mkdir(exist_ok=True)
    for tag in tags:
        import os
        name = tag if tag.startswith(rnd + "_") else f"{rnd}_{tag}"      # (tags may carry the round already: prof_r04_s1_100m)
        stats = glob.glob(str(ROOT / "gpurun_out" / f"prof_{tag}" / "**" / "*_kernel_stats.csv"), recursive=True)
        if stats:
            shutil.copy(max(stats, key=os.path.getmtime), OUT / f"{name}_kernel_stats.csv")
            print("kernel stats ->", OUT / f"{name}_kernel_stats.csv")
        bench = ROOT / "gpurun_out" / f"prof_{tag}.json"
        if bench.exists():
            shutil.copy(bench, OUT / f"{name}_bench_under_rocprof.json")
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        import os
        for d in glob.glob(str(ROOT / "gpurun_out" / f"pmc_{tag}_*")):
            files = glob.glob(d + "/**/*_counter_collection.csv", recursive=True)
            if not files:
                continue
            f = max(files, key=os.path.getmtime)                 # newest run of this counter group only
            for r in csv.DictReader(open(f)):
                k = short(r["Kernel_Name"])
                if k:
                    agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        if agg:
            out = {}
            for k, ctrs in agg.items():
                out[k] = {c: {"launches": len(v), "avg_per_launch": sum(v) / len(v)} for c, v in ctrs.items()}
                if "FETCH_SIZE" in ctrs:
                    out[k]["hbm_read_bytes_per_launch"] = out[k]["FETCH_SIZE"]["avg_per_launch"] * 1024 * 2
                if "WRITE_SIZE" in ctrs:
                    out[k]["hbm_write_bytes_per_launch"] = out[k]["WRITE_SIZE"]["avg_per_launch"] * 1024
            (OUT / f"{name}_pmc.json").write_text(json.dumps(out, indent=1, sort_keys=True))
            print("pmc ->", OUT / f"{name}_pmc.json")


if __name__ == "__main__":
    main()
