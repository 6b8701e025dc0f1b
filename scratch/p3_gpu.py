import sys, os, time
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests"); sys.path.insert(0, "oracle")
import orc
import exaconstit_amd.lib as L
txt = open(os.path.join(orc.REFDATA, "voce_pa.toml")).read()
for fl in ("props_cp_voce.txt", "state_cp_voce.txt", "voce_quats.ori", "grains.txt", "custom_dt.txt"):
    txt = txt.replace('"%s"' % fl, '"%s"' % os.path.join(orc.REFDATA, fl))
os.makedirs("/tmp/p3out", exist_ok=True)
for p in (3, 4):
  for assembly, integ in (("PA","FULL"),("EA","FULL"),("EA","BBAR")):
    t = txt.replace('assembly = "PA"', 'assembly = "%s"\n    integ_model = "%s"' % (assembly, integ)).replace("prefinement = 1", "p_refinement = %d" % p).replace("ref_ser = 1", "ref_ser = 0")
    open("/tmp/p3.toml","w").write(t)
    try:
        d = L.Driver.from_toml("/tmp/p3.toml", out_dir="/tmp/p3out")
        ok = [d.step(ti) for ti in range(1, 4)]
        s = d.avgs(0, 6)
        print(p, assembly, integ, ok, s[:,2], d.stats()[0], d.stats()[1], flush=True)
    except Exception as e:
        print(p, assembly, integ, "FAILED", repr(e)[:300], flush=True)
    if p == 3:
        case = orc.load_case("/tmp/p3.toml"); ref = orc.run_case(case, nsteps=3)
        print("  oracle", ref["avg_stress"][:,2], ref["newton_iters"], ref["krylov_iters"], "rel", np.linalg.norm(s[:,2:]-ref["avg_stress"][:,2:])/np.linalg.norm(ref["avg_stress"][:,2:]), flush=True)
