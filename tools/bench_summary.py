"""Development tool: one line per configuration of a bench.py JSON line."""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
def line(k, v):
    r, st = v["roofline"], v["stages"]
    print("%-11s value %.3e ms/step %.4f (min %.4f) | init %.4f upd %.4f lists %.4f | moved %.1f B/upd kernel frac %.3f | whole step: alg/8TB/s %.3f moved-frac %s | %s" % (
        k, v["value"], v["ms_per_step"], v["windows"]["min_ms_per_step"], st["init_ms_avg"], st["update_ms_avg"], st["lists_ms_avg"],
        r.get("moved_bytes_per_update") or 0, r["frac"], r["algorithmic"]["whole_step_over_peak"],
        ("%.3f" % r["whole_step"]["frac"]) if r.get("whole_step") else "-", r["traffic_source"][:28]))
    if v.get("init"):
        print("            init burst %.4f ms frac %.3f" % (v["init"]["kernel_ms"], v["init"]["frac"]))
    if st.get("per_program"):
        for n, t in st["per_program"].items():
            print("            %-14s init %.4f update %.4f lists %.4f" % (n, t["init_ms_avg"], t["update_ms_avg"], t["compact_ms_avg"]))
line(d["config"]["name"], d)
for k, v in d.get("configs", {}).items():
    print(k, v) if "error" in v else line(k, v)
if "cpu_baseline" in d:
    print("cpu", {k: d["cpu_baseline"].get(k) for k in ("value", "threads", "host_physical_cores", "error")})
