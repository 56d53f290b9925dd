"""lab (round 4): bench.py's pipeline_config, extracted as text and run with single edits, to find what costs 6 us per step"""
import os, sys, time, textwrap
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gr-doa_amd", "python")]
import torch, doa
src = open(os.path.join(ROOT, "bench.py")).read().splitlines()
def block(start_marker):
    i = next(k for k, l in enumerate(src) if l.startswith("    def " + start_marker))
    j = i + 1
    while j < len(src) and (src[j].startswith("        ") or not src[j].strip()):
        j += 1
    return textwrap.dedent("\n".join(src[i:j]))
code = block("timed") + "\n" + block("pipeline_config")
edits = {
    "verbatim": [],
    "no_serial": [("us_serial = timed(serial, reps)", "us_serial = 1.0")],
    "no_serial_at_all": [("us_serial = timed(serial, reps)", "us_serial = 1.0"), ("    serial(1)\n", "    pass\n")],
    "use_make": [("    pipe = doa.music_pipeline(", "    import flow_bisect_common as fc; dd = fc.make(nbuf); bufs, ptrs, cov, spec, mx, am = dd['bufs'], dd['ptrs'], dd['cov'], dd['spec'], dd['mx'], dd['am']\n    pipe = doa.music_pipeline(")],
    "literal_handle": [("doa.music_pipeline(N, K, ovl, fb, d, M, P, B)", "doa.music_pipeline(4, 2048, 512, 1, 0.4, 2, 1024, 4096)")],
    "del_src": [("    pipe = doa.music_pipeline(", "    src = None\n    pipe = doa.music_pipeline(")],
    "best_of_8": [("for _ in range(3):", "for _ in range(8):")],
    "no_20": [("us_lanes = timed(lanes_fn, reps) if ok_lanes else None", "us_lanes = 1.0")],
}
mode = sys.argv[1]
c = code
for a, b in edits[mode]:
    assert a in c, a
    c = c.replace(a, b)
ns = dict(torch=torch, doa=doa, time=time, B=4096, out={}, st=torch.cuda.current_stream(), lanes=4, spot=lambda *a: None, SNR_DB=20.0)
exec(c, ns)
ns["pipeline_config"]("flow", 4, 2048, 512, 1, 0.4, 2, 1024, 8, 20, 100, "x")
print(mode, {k: round(v, 2) for k, v in ns["out"]["flow"].items() if k.startswith("us_")}, flush=True)
