"""Regenerate the numbers in README.md / DESIGN.md from a bench record.

    python tools/fill_docs.py profiles/r05_bench.json profiles/r05_bench_full.json

README's "Measured on one MI355X ... ## Layout" section is rebuilt from tools/README_numbers.tmpl.md; DESIGN.md's table
between the R05_TABLE markers is rebuilt from the line."""
import json, os, sys, re
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
line = json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith('{"metric"')][-1])
full = json.load(open(sys.argv[2]))
c = line["configs"]
def sci(v): 
    m, e = ("%.2e" % v).split("e"); return f"{m} × 10{''.join('⁰¹²³⁴⁵⁶⁷⁸⁹'[int(d)] for d in str(int(e)))}"
ms = lambda v: ("%.4f" % v).rstrip("0") if v < 0.1 else "%.3f" % v
sub = {
 "R05_C2_MS": ms(line["ms_per_step"]), "R05_C2_V": sci(line["value"]), "R05_C2_F": "%.2f" % line["roofline"]["frac"], "R05_C2_WS": "%.2f" % line["roofline"]["algorithmic_whole_step_over_peak"],
 "R05_VIEW_MS": ms(c["c2_view"]["ms_per_step"]), "R05_IVIEW_MS": ms(c["c2_interop_view"]["ms_per_step"]), "R05_VIEW_V": sci(c["c2_view"]["value"]), "R05_VIEW_K": ms(c["c2_view"]["kernel_ms"]),
 "R05_INT_MS": ms(c["c2_interop"]["ms_per_step"]), "R05_INT_V": sci(c["c2_interop"]["value"]), "R05_INT_F": "%.2f" % c["c2_interop"]["frac"],
 "R05_MIX_MS": ms(c["c2_mixed"]["ms_per_step"]), "R05_MIX_V": sci(c["c2_mixed"]["value"]), "R05_MIX_F": "%.2f" % c["c2_mixed"]["frac"], "R05_MIX_WS": "%.2f" % c["c2_mixed"]["ws68"],
 "R05_DIE_MS": ms(c["c2_dieoff"]["ms_per_step"]), "R05_DIE_V": sci(c["c2_dieoff"]["value"]), "R05_DIE_F": "%.2f" % c["c2_dieoff"]["frac"],
 "R05_EV_MS": ms(c["c2_events"]["ms_per_step"]), "R05_EV_V": sci(c["c2_events"]["value"]), "R05_EV_F": "58.4 B, %.2f (trails)" % c["c2_events"]["frac"],
 "R05_C3_MS": ms(c["c3"]["ms_per_step"]), "R05_C3_V": sci(c["c3"]["value"]), "R05_C3_F": "%.2f" % c["c3"]["frac"],
 "R05_C4_MS": ms(c["c4"]["ms_per_step"]), "R05_C4_V": sci(c["c4"]["value"]), "R05_C4_F": "%.2f" % c["c4"]["frac"],
 "R05_C5_MS": ms(c["c5"]["ms_per_step"]), "R05_C5_V": sci(c["c5"]["value"]),
 "R05_SC_MS": ms(line["small_effects_scene"]["ms_per_frame_wall"]), "R05_SCI_MS": ms(line["small_effects_scene"]["ms_interpreters"]),
 "R05_CPU": sci(line["cpu_baseline"]["value"]),
}
readme = open(ROOT + "/README.md").read()
a, b = readme.index("Measured on one MI355X (`python bench.py`"), readme.index("## Layout")
readme = readme[:a] + open(ROOT + "/tools/README_numbers.tmpl.md").read() + readme[b:]
for k in sorted(sub, key=len, reverse=True): readme = readme.replace(k, sub[k])
assert "R05_" not in readme, re.findall(r"R05_\w+", readme)
open(ROOT + "/README.md", "w").write(readme)
rows = [("**c2** firework burst 16,777,216 (headline; `LEAN`)", line["ms_per_step"], line["value"], line["roofline"]["kernel_ms_avg"], line["roofline"].get("kernel_ms_rocprof"), line["roofline"]["moved_bytes_per_update"], line["roofline"]["frac"], line["roofline"]["algorithmic_whole_step_over_peak"], "ages per chunk, lifetimes culled, no lists")]
notes = {"c2_mixed": "init %.3f + update %.3f + lists %.3f" % tuple(c["c2_mixed"]["stages_ms"]), "c2_dieoff": "per SURVIVING particle", "c2_interop": "cohorts off", "c2_view": "`AUTO`: cohorts + materialise + consumer kernel", "c2_interop_view": "cohorts off + the same consumer",
         "c2_events": "3 effects, trails' kernel", "c3": "burst init frac %.2f" % c["c3"].get("init_frac", 0), "c4": "burst init frac %.2f" % c["c4"].get("init_frac", 0), "c5": "three launches"}
names = {"c2_mixed": "**c2_mixed** rate spawner steady state", "c2_dieoff": "c2_dieoff frames 48–70", "c2_interop": "c2_interop", "c2_view": "**c2_view** the asset end to end", "c2_interop_view": "c2_interop_view", "c2_events": "c2_events real firework.rs", "c3": "c3 force field 8.4M", "c4": "c4 512 × 65,536", "c5": "c5 ribbon 4.19M"}
for k in ("c2_view", "c2_interop_view", "c2_interop", "c2_mixed", "c2_dieoff", "c2_events", "c3", "c4", "c5"):
    v = c[k]; rows.append((names[k], v["ms_per_step"], v["value"], v["kernel_ms"], v.get("kernel_ms_rocprof"), v["B_upd"], v["frac"], v["ws68"], notes[k]))
t = "| config (1 MI355X, `python bench.py`, median of 25 windows × 30 steps) | step ms | updates/s | kernel ms (HIP events / rocprofv3) | moved B/update | kernel frac of 8 TB/s | whole step, 68 B × updates ÷ 8 TB/s | |\n|---|---|---|---|---|---|---|---|\n"
for n, m, v, k, kr, b, f, w, note in rows:
    t += f"| {n} | {ms(m)} | {v:.3g} | {ms(k)} / {ms(kr) if kr else '—'} | {b:.3g} | {f:.2f} | {w:.2f} | {note} |\n"
sc = line["small_effects_scene"]
t += f"| 26-effect scene (set module / interpreters) | {ms(sc['ms_per_frame_wall'])} / {ms(sc['ms_interpreters'])} per frame | — | — | — | launch-bound | — | |\n"
t += f"\nBurst init of c2: {ms(line['burst_init']['kernel_ms'])} ms = {line['burst_init']['frac']:.2f} of 8 TB/s on 44 B per spawn. CPU port: {line['cpu_baseline']['value']:.3g} updates/s with {line['cpu_baseline']['threads']} threads on a {line['cpu_baseline'].get('cpu_quota')}-CPU quota. `comm`: {line['comm']['library'].split('/')[-1]}, 1 rank, total = {line['comm']['alive_total']}.\n"
d = open(ROOT + "/DESIGN.md").read()
d = re.sub(r"<!-- R05_TABLE_BEGIN -->.*?<!-- R05_TABLE_END -->", "<!-- R05_TABLE_BEGIN -->\n" + t.replace("\\", "\\\\") + "<!-- R05_TABLE_END -->", d, flags=re.S)
open(ROOT + "/DESIGN.md", "w").write(d)
print(len(d), len(readme))
