"""Per-launch record of the constitutive kernel inside a real Newton/PCG solve (128^3 FCC Voce by default): duration (EXA_MODEL_TIMES) and
evaluation-count histogram (EXA_NFEV_LOG) of every launch, for a list of tail-split settings (EXA_NEWTON_CAP values).  Run on the GPU box:
    python scripts/transition_study.py [--n 128] [--steps 14] [--caps off,auto,6,8] [--model fcc_voce] > gpurun_out/transition.txt
Each setting runs in a fresh process (the library reads the switches once)."""
import argparse
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, json, time
import numpy as np
sys.path.insert(0, ROOT)
import exaconstit_amd.lib as L
N, steps, model = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
xt, sl = model.split("_", 1)
pfile = {"voce": "props_cp_voce.txt", "voce_nl": "props_cp_vocenl.txt", "kmdd": "props_cp_mts.txt"}[sl]
props = np.loadtxt(os.path.join(ROOT, "tests", "golden", "refdata", pfile)).ravel()
rng = np.random.default_rng(20240928)
quats = rng.standard_normal((N ** 3, 4)); quats /= np.linalg.norm(quats, axis=1, keepdims=True)
sched = np.loadtxt(os.path.join(ROOT, "tests", "golden", "refdata", "custom_dt.txt")).ravel()[:steps]
drv = L.Driver.synthetic(N, props, quats.ravel(), sched, bcc=(xt == "bcc"), slip={"voce": 0, "voce_nl": 1, "kmdd": 2}[sl])
rows = []
for ti in range(1, steps + 1):
    drv.reset_timers()
    sys.stderr.write("step %d\n" % ti); sys.stderr.flush()
    assert drv.step(ti)
    tm = drv.timers(); nw, kr, mc = drv.stats()
    rows.append((ti, float(sched[ti - 1]), int(nw[-1]), int(kr[-1]), int(mc[-1]), tm["model_ms"]))
print(json.dumps({"rows": rows, "szz": [float(x) for x in drv.avgs(0, 6)[:, 2]]}))
'''


def run(cap, n, steps, model):
    env = dict(os.environ, EXA_MODEL_TIMES="1", EXA_NFEV_LOG="1")
    if cap != "default":
        env["EXA_NEWTON_CAP"] = cap
    p = subprocess.run([sys.executable, "-c", "ROOT=%r\n" % ROOT + WORKER, str(n), str(steps), model], env=env, capture_output=True, text=True)
    if p.returncode != 0:
        return None, p.stderr[-2000:]
    out = json.loads(p.stdout.strip().splitlines()[-1])
    launches = []; step = 0
    times = {}
    for line in p.stderr.splitlines():
        m = re.match(r"step (\d+)", line)
        if m:
            step = int(m.group(1)); continue
        m = re.match(r"model_launch call (\d+): ([\d.]+) ms", line)
        if m:
            times[int(m.group(1))] = float(m.group(2)); continue
        m = re.match(r"nfev_hist call (\d+) dt (\S+) cap (\d+) tail (\d+):(.*)", line)
        if m:
            h = {int(a): int(b) for a, b in (x.split(":") for x in m.group(5).split())}
            launches.append(dict(call=int(m.group(1)), step=step, dt=float(m.group(2)), cap=int(m.group(3)), tail=int(m.group(4)), hist=h))
    for l in launches:
        l["ms"] = times.get(l["call"])
    return dict(rows=out["rows"], szz=out["szz"], launches=launches), None


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=128); ap.add_argument("--steps", type=int, default=14)
    ap.add_argument("--caps", default="off,auto"); ap.add_argument("--model", default="fcc_voce")
    ap.add_argument("--json", default=None)
    a = ap.parse_args()
    allres = {}
    for cap in a.caps.split(","):
        res, err = run(cap, a.n, a.steps, a.model)
        if res is None:
            print("cap", cap, "FAILED", err); continue
        allres[cap] = res
        tot = sum(l["ms"] for l in res["launches"] if l["ms"]); n = len(res["launches"])
        print(f"== EXA_NEWTON_CAP={cap}: {n} launches, {tot:.1f} ms, mean {tot / max(n, 1):.3f} ms/launch; sigma_zz last {res['szz'][-1]:.9g}")
        for l in res["launches"]:
            tp = sum(l["hist"].values()); mean = sum(k * v for k, v in l["hist"].items()) / max(tp, 1); mx = max(l["hist"])
            big = {k: v for k, v in l["hist"].items() if v > 0.002 * tp}
            print(f"  step {l['step']:2d} call {l['call']:3d} dt {l['dt']:.3f} cap {l['cap']:2d} tail {l['tail']:8d}  {l['ms'] if l['ms'] is not None else float('nan'):7.3f} ms  mean {mean:.2f} max {mx:2d}  {big}")
    if a.json:
        json.dump(allres, open(a.json, "w"))
