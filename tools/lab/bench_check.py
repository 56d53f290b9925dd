"""lab: the figures of one bench.py JSON line that get quoted in DESIGN.md, one per line"""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("headline", d["value"], "ms/step", d["ms_per_step"], "roofline", d["roofline"]["frac"])
if "roofline_scan" in d:
    big = d["roofline_scan"]["at_large_batch"]
    print("scan large", big["frac"], big["avg_launch_us"])
for k, v in d.get("other_configs", {}).items():
    print(k, {kk: (round(vv, 2) if isinstance(vv, float) else vv) for kk, vv in v.items() if kk.startswith("us_") or kk == "spot_check"})
if "cpu_baseline" in d:
    print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"])
