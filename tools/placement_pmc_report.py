"""Development tool: table of per-slab counter means from the CSVs written by tools/placement_pmc.sh (see placement_pmc.py)."""
import csv, glob, sys, collections
n = int(sys.argv[1]); files = sys.argv[2:]
table = collections.defaultdict(dict)   # (slab, mode) -> {counter: mean}
for path in files:
    rows = [r for r in csv.DictReader(open(path)) if "k_update_slots_stream" in r["Kernel_Name"]]
    by_counter = collections.defaultdict(list)
    for r in rows:
        by_counter[r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    for cname, lst in by_counter.items():
        lst.sort()
        lst = lst[n:]                      # the burst frames at creation
        assert len(lst) == n * 24, (cname, len(lst))
        for s in range(n):
            for m, mode in enumerate(("one-dir", "alternating")):
                seg = lst[s * 24 + m * 12 + 3: s * 24 + (m + 1) * 12]   # skip 3 frames after the mode switch
                table[(s, mode)][cname] = sum(v for _, v, _ in seg) / len(seg)
                table[(s, mode)]["us(" + path.split("pmc_")[-1][:6] + ")"] = sum(d for _, _, d in seg) / len(seg) / 1e3
cols = sorted({c for v in table.values() for c in v})
print("slab mode        " + "  ".join("%22s" % c[:22] for c in cols))
for (s, mode), v in sorted(table.items()):
    print("%3d  %-11s " % (s, mode) + "  ".join("%22.4g" % v.get(c, float("nan")) for c in cols))
