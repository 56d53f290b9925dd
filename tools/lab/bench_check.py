import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("headline", d["value"], "ms/step", d["ms_per_step"], "roofline", d["roofline"]["frac"], "scan large", d["roofline_scan"]["at_large_batch"]["frac"], d["roofline_scan"]["at_large_batch"]["avg_launch_us"])
for k, v in d["other_configs"].items():
    print(k, {kk: (round(vv, 2) if isinstance(vv, float) else vv) for kk, vv in v.items() if kk.startswith("us_") or kk == "spot_check"})
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"])
