#!/bin/bash
# lab (round 4): bench.py's other_configs (cfg3 through doa_root_pipeline, flowgraph shape, cfg4) with the quad EVD on / off
export DOA_HIP_LIB=$PWD/_ab/libdoa_hip_lab.so
for rep in 1 2; do
for q in 0 1; do
    export DOA_EVD_QUAD=$q
    python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-scan-roofline > gpurun_out/r04/bench_quad_$q.json 2> gpurun_out/r04/bench_quad_$q.err
    python3 - gpurun_out/r04/bench_quad_$q.json $q <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("quad", sys.argv[2], "headline us/step", round(d["ms_per_step"] * 1e3, 2))
for k, v in d.get("other_configs", {}).items():
    if "error" in v: print("  ", k, v["error"]); continue
    print("  ", k, "serial", round(v["us_per_step_serial"], 2), "overlapped", round(v["us_per_step_overlapped"], 2) if v["us_per_step_overlapped"] else None, "check", (v.get("spot_check") or {}).get("ok"))
PY
done
done
