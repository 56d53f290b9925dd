"""MFMA scan kernel against the lean kernel on the same covariances (child processes: the switch is read once per
process).  usage: python tools/exp/mfma_check.py [--n 4096] [--P 1024]"""
import argparse, os, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=4099)
ap.add_argument("--P", type=int, default=1024)
ap.add_argument("--N", type=int, default=4)
ap.add_argument("--child", default="")
a = ap.parse_args()
if a.child:
    sys.path[:0] = [os.path.join(ROOT, "gr-doa_amd", "python")]
    import doa
    N, K, P, n = a.N, 64, a.P, a.n
    x, _ = doa.sim.make_batch_streams(N, K, n, 0.5, 1, 20.0, seed=3)
    x = [np.ascontiguousarray(t) for t in x]
    x[0][5 * K + 3] = np.nan                          # one irregular row
    x[1][(n - 1) * K + 1] = np.inf                    # and one in the last (ragged) group
    pipe = doa.music_pipeline(N, K, 0, 0, 0.5, 1, P, n)
    mx, am = np.empty((n, 1), np.float32), np.empty((n, 1), np.float32)
    sp = np.empty((n, P), np.float32)
    try:
        pipe.work(n, x, mx, am, spectrum_out=sp)
    except Exception as e:
        print("work raised:", e)
    blk = doa.MUSIC_lin_array(0.5, 1, N, P)
    ac = doa.autocorrelate(N, K, 0, 0)
    R = np.empty((n, N * N), np.complex64); ac.general_work(n, x, [R])
    sb = np.empty((n, P), np.float32)
    try:
        blk.work(n, [R], [sb])
    except Exception as e:
        print("block raised:", e)
    np.savez(a.child, mx=mx, am=am, sp=sp, sb=sb)
    sys.exit(0)
with tempfile.TemporaryDirectory() as tmp:
    out = {}
    for tag, env in (("lean", {"DOA_SCAN_MFMA": "0"}), ("mfma", {"DOA_SCAN_MFMA": "1"})):
        f = os.path.join(tmp, tag + ".npz")
        r = subprocess.run([sys.executable, __file__, "--child", f, "--n", str(a.n), "--P", str(a.P), "--N", str(a.N)],
                           env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
        print(tag, r.stdout.strip()[-300:], r.stderr.strip()[-300:])
        out[tag] = dict(np.load(f))
    l, m = out["lean"], out["mfma"]
    fin = np.isfinite(l["sp"]) & np.isfinite(m["sp"])
    print("non-finite pattern equal:", np.array_equal(np.isfinite(l["sp"]), np.isfinite(m["sp"])))
    d = np.abs(l["sp"] - m["sp"])[fin]
    print("spectrum max |diff| dB:", d.max(), "rows differing:", int((np.where(fin, l["sp"] != m["sp"], False)).any(axis=1).sum()), "of", a.n)
    print("block == pipeline (mfma):", np.array_equal(m["sp"], m["sb"], equal_nan=True))
    print("max equal:", np.array_equal(l["mx"], m["mx"], equal_nan=True), "argmax equal:", np.array_equal(l["am"], m["am"], equal_nan=True),
          "argmax mismatches:", int((l["am"] != m["am"]).sum()))
    zl, zm = (l["sp"] == 0).sum(axis=1), (m["sp"] == 0).sum(axis=1)
    print("zeros per row equal:", np.array_equal(zl, zm), "rows with != 1 zero:", int((zm != 1).sum()))
