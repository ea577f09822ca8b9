#!/usr/bin/env python3
"""Development tool: turn the two PMC passes of tools/prof_bench.sh (FETCH_SIZE and WRITE_SIZE, collected in separate
rocprofv3 runs of the default `python bench.py --no-cpu-baseline`) into profiles/traffic.json, the file bench.py reads
`roofline.traffic` from, and copy the rows of the hot-path kernels into profiles/ as evidence.

HBM bytes per launch = FETCH_SIZE x 1024 x 2 + WRITE_SIZE x 1024  (the counters are in KiB; on gfx950 FETCH_SIZE reports half of
the bytes of wide coalesced streaming reads — guides/MI355X_MICROARCH.md, HBM section; calibrated in profiles/r01e_summary.md).
Usage: pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <tag>   (tag: e.g. r02b)
"""
import csv
import json
import os
import re
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def op_values():
    txt = open(os.path.join(ROOT, "include", "hanabi_amd.h")).read()
    body = txt[txt.index("typedef enum HnbOp {"):txt.index("} HnbOp;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = re.findall(r"\b(HNB_OP_[A-Z0-9_]+)\b", body)
    return {n: i for i, n in enumerate(names)}


def main():
    fetch_csv, write_csv, tag = sys.argv[1:4]
    ops = op_values()
    sig = lambda *names: ", ".join(f"{ops['HNB_OP_M_' + n]}u" for n in names)
    # dominant kernel of each bench configuration (bench.py CONFIGS), by its static op sequence
    configs = {
        "c2:16777216x1": ("k_update_slots_stream", sig("AGE_TICK", "VEL_SCALE", "VEL_ADD", "EULER")),
        "c3:8388608x1": ("k_update_slots_stream", sig("AGE_TICK", "CONFORM_SPHERE", "CONFORM_SPHERE", "KILL_AABB", "KILL_SPHERE", "EULER")),
        "c4:65536x512": ("k_update_slots_stream", sig("AGE_TICK", "EULER")),
        "c5:4194304x1": ("k_update_slots_stream_age", ""),   # (r6: the age-only update has a kernel of its own)
    }

    def load(path, counter):
        rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter]
        return rows

    f_rows, w_rows = load(fetch_csv, "FETCH_SIZE"), load(write_csv, "WRITE_SIZE")
    out = {}
    keep = re.compile(r"k_update_slots|k_init|k_list_rows|k_compact|k_sort|k_emit|k_order")
    for key, (kern, s) in configs.items():
        sel = lambda rows: [float(r["Counter_Value"]) for r in rows if kern in r["Kernel_Name"] and s in r["Kernel_Name"]]
        fv, wv = sel(f_rows), sel(w_rows)
        if not fv or not wv:
            continue
        # steady state: the median launch (the frame after a burst reads the lifetimes once more; warm-up frames of the churn differ)
        f_kib, w_kib = statistics.median(fv), statistics.median(wv)
        out[key] = {"bytes_per_launch": f_kib * 1024 * 2 + w_kib * 1024, "fetch_size_kib_median": f_kib, "write_size_kib_median": w_kib,
                    "launches": [len(fv), len(wv)], "kernel": f"{kern}<ProgStatic<{s.strip('<>')}>>",
                    "source": f"profiles/{tag}_pmc_FETCH_SIZE.csv + profiles/{tag}_pmc_WRITE_SIZE.csv (FETCH_SIZE x2 on gfx950)"}
    for path, name in ((fetch_csv, "FETCH_SIZE"), (write_csv, "WRITE_SIZE")):
        rd = csv.DictReader(open(path))
        cols = ["Dispatch_Id", "Grid_Size", "Kernel_Name", "Workgroup_Size", "VGPR_Count", "Counter_Name", "Counter_Value", "Start_Timestamp", "End_Timestamp"]
        with open(os.path.join(ROOT, "profiles", f"{tag}_pmc_{name}.csv"), "w", newline="") as f:
            w = csv.DictWriter(f, fieldnames=cols)
            w.writeheader()
            for r in rd:
                if keep.search(r["Kernel_Name"]) and r["Counter_Name"] == name and "k_sort" not in r["Kernel_Name"]:
                    r = {c: r[c] for c in cols}
                    r["Kernel_Name"] = re.sub(r"\(.*", "", r["Kernel_Name"])[:160]
                    w.writerow(r)
    json.dump(out, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
    for k, v in out.items():
        print(f"{k}: {v['bytes_per_launch']:.4e} B per launch ({v['fetch_size_kib_median']:.1f} KiB x2 fetched, {v['write_size_kib_median']:.1f} KiB written, {v['launches']} launches)")


if __name__ == "__main__":
    main()
