"""Regenerate the numbers in README.md / DESIGN.md from a bench record.

    python tools/fill_docs.py profiles/r06_bench.json

README's "Measured on one MI355X ... ## Layout" section is rebuilt from tools/README_numbers.tmpl.md; DESIGN.md's table
between the R06_TABLE markers is rebuilt from the line."""
import json, os, sys, re
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
line = json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith('{"metric"')][-1])
c = line["configs"]
def sci(v):
    m, e = ("%.2e" % v).split("e"); return f"{m} × 10{''.join('⁰¹²³⁴⁵⁶⁷⁸⁹'[int(d)] for d in str(int(e)))}"
ms = lambda v: ("%.4f" % v).rstrip("0") if v < 0.1 else "%.3f" % v
b3 = lambda v: "%.3g" % v
ro = line["roofline"]
sub = {"R06_C2_MS": ms(line["ms_per_step"]), "R06_C2_V": sci(line["value"]), "R06_C2_F": "%.2f" % ro["frac"], "R06_C2_B": b3(ro["moved_bytes_per_update"]), "R06_C2_WS": "%.2f" % ro["algorithmic_whole_step_over_peak"],
       "R06_VIEW_C": ms(c["c2_view"]["consumer_ms"]), "R06_VIEW_S": ms(c["c2_view"]["sim_only_ms"]), "R06_VIEW_K": ms(c["c2_view"]["kernel_ms"]),
       "R06_RB_I": ms(c["c2_reburst"]["reburst_init_ms"]), "R06_RB_IF": "%.2f" % c["c2_reburst"]["reburst_init_frac"],
       "R06_C3_I": "%.2f" % c["c3"]["init_frac"], "R06_C4_I": "%.2f" % c["c4"]["init_frac"],
       "R06_BI_MS": ms(line["burst_init"]["kernel_ms"]), "R06_BI_F": "%.2f" % line["burst_init"]["frac"],
       "R06_SC_MS": ms(line["small_effects_scene"]["ms_per_frame_wall"]), "R06_SCI_MS": ms(line["small_effects_scene"]["ms_interpreters"]),
       "R06_CPU_T": str(line["cpu_baseline"]["threads"]), "R06_CPU_Q": "%g" % (line["cpu_baseline"].get("cpu_quota") or 0), "R06_CPU": sci(line["cpu_baseline"]["value"])}
for tag, k in (("LEAN", "c2_lean"), ("INT", "c2_interop"), ("VIEW", "c2_view"), ("RB", "c2_reburst"), ("MIX", "c2_mixed"), ("DIE", "c2_dieoff"), ("EV", "c2_events"), ("C3", "c3"), ("C4", "c4"), ("C5", "c5")):
    v = c[k]
    sub.update({f"R06_{tag}_MS": ms(v["ms_per_step"]), f"R06_{tag}_V": sci(v["value"]), f"R06_{tag}_F": "%.2f" % v["frac"], f"R06_{tag}_B": b3(v["B_upd"]), f"R06_{tag}_WS": "%.2f" % v["ws68"]})
readme = open(ROOT + "/README.md").read()
a, b = readme.index("Measured on one MI355X (`python bench.py`"), readme.index("## Layout")
readme = readme[:a] + open(ROOT + "/tools/README_numbers.tmpl.md").read() + readme[b:]
for k in sorted(sub, key=len, reverse=True): readme = readme.replace(k, sub[k])
assert "R06_" not in readme, re.findall(r"R06_\w+", readme)
open(ROOT + "/README.md", "w").write(readme)
rows = [("**c2** firework burst 16,777,216 (headline; library defaults)", line["ms_per_step"], line["value"], ro["kernel_ms_avg"], ro.get("kernel_ms_rocprof"), ro["moved_bytes_per_update"], ro["frac"], ro["algorithmic_whole_step_over_peak"], "cohorts + ages written in the kernel, lifetimes culled, no lists")]
notes = {"c2_lean": "`LEAN`: AGE stale (the headline of r2–r5)", "c2_interop": "cohorts off", "c2_view": "+ consumer %.3f ms; simulation %.3f" % (c["c2_view"]["consumer_ms"], c["c2_view"]["sim_only_ms"]),
         "c2_reburst": "re-burst init %.3f ms = %.2f of HBM; stages %.3f + %.3f + %.3f" % ((c["c2_reburst"]["reburst_init_ms"], c["c2_reburst"]["reburst_init_frac"]) + tuple(c["c2_reburst"]["stages_ms"])),
         "c2_mixed": "init %.3f + update %.3f + lists %.3f" % tuple(c["c2_mixed"]["stages_ms"]), "c2_dieoff": "per SURVIVING particle",
         "c2_events": "3 effects, trails' kernel", "c3": "burst init frac %.2f" % c["c3"].get("init_frac", 0), "c4": "burst init frac %.2f" % c["c4"].get("init_frac", 0), "c5": "three launches, device-bound"}
names = {"c2_lean": "c2_lean", "c2_interop": "c2_interop", "c2_view": "c2_view", "c2_reburst": "**c2_reburst** burst(capacity, period), 4-frame cycle", "c2_mixed": "**c2_mixed** rate spawner steady state", "c2_dieoff": "c2_dieoff frames 48–70",
         "c2_events": "c2_events real firework.rs", "c3": "c3 force field 8.4M", "c4": "c4 512 × 65,536", "c5": "c5 ribbon 4.19M"}
for k in ("c2_lean", "c2_interop", "c2_view", "c2_reburst", "c2_mixed", "c2_dieoff", "c2_events", "c3", "c4", "c5"):
    v = c[k]; rows.append((names[k], v["ms_per_step"], v["value"], v["kernel_ms"], v.get("kernel_ms_rocprof"), v["B_upd"], v["frac"], v["ws68"], notes[k]))
t = "| config (1 MI355X, `python bench.py`, median of 25 windows × 30 steps) | step ms | updates/s | kernel ms (HIP events) | moved B/update | kernel frac of 8 TB/s | whole step, 68 B × updates ÷ 8 TB/s | |\n|---|---|---|---|---|---|---|---|\n"
for n, m, v, k, kr, b, f, w, note in rows:
    t += f"| {n} | {ms(m)} | {v:.3g} | {ms(k)}{' / ' + ms(kr) + ' (rocprofv3)' if kr else ''} | {b:.3g} | {f:.2f} | {w:.2f} | {note} |\n"
sc = line["small_effects_scene"]
t += f"| 26-effect scene (set module / interpreters) | {ms(sc['ms_per_frame_wall'])} / {ms(sc['ms_interpreters'])} per frame | — | — | — | four dependent launches | — | |\n"
t += f"\nFirst burst of c2 (`k_init_slots`): {ms(line['burst_init']['kernel_ms'])} ms = {line['burst_init']['frac']:.2f} of 8 TB/s on 44 B per spawn. CPU port: {line['cpu_baseline']['value']:.3g} updates/s with {line['cpu_baseline']['threads']} threads on a {line['cpu_baseline'].get('cpu_quota')}-CPU quota. `comm`: {line['comm']['library'].split('/')[-1]}, 1 rank, total = {line['comm']['alive_total']}.\n"
d = open(ROOT + "/DESIGN.md").read()
d = re.sub(r"<!-- R06_TABLE_BEGIN -->.*?<!-- R06_TABLE_END -->", "<!-- R06_TABLE_BEGIN -->\n" + t.replace("\\", "\\\\") + "<!-- R06_TABLE_END -->", d, flags=re.S)
open(ROOT + "/DESIGN.md", "w").write(d)
print(len(d), len(readme))
