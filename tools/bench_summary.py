#!/usr/bin/env python
"""One screen of a bench.py line: python tools/bench_summary.py FILE"""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
c = d["config"]
print("headline ms/step %.3f  value %.0f %s  roofline frac %.3f (stage %.3f)  regions %s minmax %s" % (d["ms_per_step"], d["value"], d["unit"], d["roofline"]["frac"], d["roofline"].get("scan_stage", {}).get("frac", 0), d.get("regions"), d.get("ms_per_step_minmax")))
print("config_wall_ms_measured", c.get("config_wall_ms_measured"), (c.get("config_measured") or {}).get("minmax"))
cp = c.get("cached_pass") or {}
print("cached_pass", cp.get("ms_per_step"), cp.get("ms_per_step_minmax"), "kernels", cp.get("scan_kernels_ms"), "one-query", (cp.get("one_query_pass") or {}).get("ms_per_pass"))
print("batch_setup_ms", c.get("batch_setup_ms"))
for k, v in (c.get("other_workloads") or {}).items():
    if not isinstance(v, dict): continue
    extra = ""
    if k == "cli":
        cc = v.get("config", {}); extra = " warm %s first %s rows_equal %s" % (cc.get("warm_runs_ms"), (cc.get("first_run") or {}).get("wall_ms"), cc.get("rows_equal_library_calls"))
    print(k, v.get("ms_per_step", v.get("value")), v.get("ms_per_step_minmax"), v.get("error", ""), extra)
cb = d.get("cpu_baseline") or {}
print("cpu_baseline", cb.get("value"), cb.get("unit"), "cores", cb.get("cores"), cb.get("kind"))
print("box", json.dumps(c.get("box"))[:300] if c.get("box") else "")
