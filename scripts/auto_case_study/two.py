import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', 'tests'))
import numpy as np, orc
from knee import run, setp, G, dts
s,_ = run(setp(), dts[:2]); rate=(s[1]-s[0])/dts[1]; d2=dts.copy(); d2[1]=(G[1]-s[0])/rate
rows = list(d2) + [0.064129115905, 0.059378751644, 0.054980270616, 0.057271057954]
beta = float(sys.argv[1]); which = sys.argv[2]
def mod(p):
    p[8] *= 3.7771
    if which == "k1": p[17] *= beta
    elif which == "s": p[16] *= beta
    elif which == "rho0": p[21] *= beta
s,f = run(mod, np.array(rows))
print(which, beta, "rows 9-15 residual", np.round((s-G[:15])[8:],4), flush=True)
case = orc.load_case("mtsdd_full_auto.toml"); p = case["props"].copy(); mod(p); case["props"]=p; case["auto"]=None; case["dts"]=np.full(20,0.5)
out = orc.run_case(case); print(which, beta, "final", np.round(out["avg_stress"][-1][2:],3), "golden [-773.13 9.574 -3.797 -4.292]", flush=True)
