import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', 'tests'))
import numpy as np, orc
def run(mod):
    case = orc.load_case("mtsdd_full_auto.toml")
    p = case["props"].copy(); mod(p); case["props"]=p
    case["auto"]=None; case["dts"]=np.full(20,0.5)
    out = orc.run_case(case)
    return out["avg_stress"][-1][2:], out["failed"]
sa = float(sys.argv[1])
for kb in [float(x) for x in sys.argv[2:]]:
    def mod(p): p[16]*=sa; p[17]*=kb
    s,f = run(mod); print("s x", sa, "k1 x", kb, np.round(s,3), f, flush=True)
