"""Time-free fit of the Time.Auto golden file (mtsdd_full_auto_stress.txt) in stress space.  CPU only, test infrastructure only.

The golden file holds no times, but at constant applied velocity the averaged stress follows one curve in (sigma_33, sigma_23, sigma_13,
sigma_12) space whatever the step sizes were.  For every golden row the oracle's curve (fixed small steps) is interpolated at the row's
sigma_33 and the three shear averages are compared; the last row adds sigma_33(t = 10).  A Levenberg-Marquardt loop over property
multipliers then answers: is there ANY parameter set of the implemented law that traces the golden curve?  (If a set of round numbers
did, the props file would simply not be the one the golden file was made with.)

usage: python scripts/auto_case_study/trajectory_fit.py eval name=value ...       one run, prints the misfit
       python scripts/auto_case_study/trajectory_fit.py fit name[=start],name,... [iters]  LM fit over the named properties (4 runs at a time)
names: c_1 tau_a p q gam_wo gam_ro wrD go s k1 k2o ninv gamma_o rho0   (values are multipliers; tau_a/go also accept +x additive as 'a+x')
"""
import multiprocessing as mp
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "tests"))
os.environ.setdefault("OMP_NUM_THREADS", "2")

IDX = dict(mu=6, tK_ref=7, c_1=8, tau_a=9, p=10, q=11, gam_wo=12, gam_ro=13, wrD=14, go=15, s=16, k1=17, k2o=18, ninv=19, gamma_o=20, rho0=21)
NSTEP, DT = 80, 0.125


def curve(mods):
    import numpy as np
    import orc
    case = orc.load_case("mtsdd_full_auto.toml")
    p = case["props"].copy()
    for k, v in mods.items():
        p[IDX[k]] *= v
    case["props"] = p
    case["auto"] = None
    case["dts"] = np.full(NSTEP, DT)
    out = orc.run_case(case)
    return out["avg_stress"][:, 2:].copy()


def residual(mods):
    import numpy as np
    import orc
    g = orc.golden("mtsdd_full_auto_stress.txt")[:, 2:]
    c = curve(mods)
    s33 = -c[:, 0]
    res = []
    for i in range(11, 71):   # plastic rows
        x = -g[i, 0]
        if x > s33[-1]:
            # beyond the oracle's range: penalise with the end point (keeps the residual continuous)
            sh = c[-1, 1:]; extra = (x - s33[-1]) * 0.05
        else:
            sh = np.array([np.interp(x, s33, c[:, k]) for k in (1, 2, 3)]); extra = 0.0
        res.extend(list(sh - g[i, 1:]) + [extra])
    res.append((c[-1, 0] - g[-1, 0]) * 0.2)   # sigma_33 at t = 10, weight 0.2 (MPa -> units comparable with the shear misfits)
    return np.array(res)


def _work(args):
    return residual(args)


def main():
    import numpy as np
    if sys.argv[1] == "eval":
        mods = {kv.split("=")[0]: float(kv.split("=")[1]) for kv in sys.argv[2:]}
        r = residual(mods)
        sh = r[:-1].reshape(-1, 4)
        print(mods, "rms shear misfit %.3f MPa, max %.3f, rows beyond range %d, sigma_33(t=10) misfit %.2f MPa" %
              (np.sqrt(np.mean(sh[:, :3] ** 2)), np.abs(sh[:, :3]).max(), int(np.sum(sh[:, 3] > 0)), r[-1] / 0.2))
        return
    spec = sys.argv[2].split(",")      # name or name=start (multiplier)
    names = [s.split("=")[0] for s in spec]
    iters = int(sys.argv[3]) if len(sys.argv) > 3 else 6
    x = np.array([np.log(float(s.split("=")[1])) if "=" in s else 0.0 for s in spec])   # log multipliers
    lam = 1e-2
    pool = mp.Pool(4)
    r0 = residual(dict(zip(names, np.exp(x))))
    print("start cost", float(r0 @ r0), flush=True)
    for it in range(iters):
        h = 0.05
        jobs = [dict(zip(names, np.exp(x + h * np.eye(len(names))[k]))) for k in range(len(names))]
        rs = pool.map(_work, jobs)
        J = np.array([(r - r0) / h for r in rs]).T
        while True:
            dx = np.linalg.solve(J.T @ J + lam * np.diag(np.diag(J.T @ J) + 1e-12), -J.T @ r0)
            dx = np.clip(dx, -1.0, 1.0)
            r1 = residual(dict(zip(names, np.exp(x + dx))))
            if r1 @ r1 < r0 @ r0:
                x = x + dx; r0 = r1; lam = max(lam / 3, 1e-6); break
            lam *= 5
            if lam > 1e4:
                break
        sh = r0[:-1].reshape(-1, 4)
        print("iter", it, "cost %.4f" % float(r0 @ r0), "rms shear %.3f max %.3f beyond %d s33(10) %.2f" %
              (np.sqrt(np.mean(sh[:, :3] ** 2)), np.abs(sh[:, :3]).max(), int(np.sum(sh[:, 3] > 0)), r0[-1] / 0.2),
              dict(zip(names, np.round(np.exp(x), 4))), flush=True)
        if lam > 1e4:
            break


if __name__ == "__main__":
    main()
