"""Full-length GPU runs of the reference's regression option files vs its golden stress files: number of printed sigma_33 rows that differ
from the golden text (6 significant digits, the reference's own acceptance criterion) and the largest difference in units of the last digit."""
import os, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import exaconstit_amd.lib as L, orc
for name, gold in [("voce_pa", "voce_pa"), ("voce_ea", "voce_ea"), ("voce_bcc", "voce_bcc"), ("voce_nl_full", "voce_full"), ("mtsdd_full", "mtsdd_full"), ("mtsdd_bcc", "mtsdd_bcc"),
                   ("voce_ea_cs", "voce_ea_cs"), ("voce_full_cyclic", "voce_full_cyclic"), ("voce_full_cyclic_cs", "voce_full_cyclic_cs"), ("voce_full_cyclic_csm", "voce_full_cyclic_csm")]:
    d = L.Driver.from_toml(os.path.join(orc.REFDATA, name + ".toml"), out_dir=tempfile.mkdtemp())
    n = d.run()
    s = d.avgs(0, 6); g = orc.golden(gold + "_stress.txt")
    m = min(len(s), len(g))
    unit = 10.0 ** (np.floor(np.log10(np.abs(g[:m, 2]))) - 5)
    u = np.abs(orc.fmt6(s[:m, 2]) - g[:m, 2]) / unit
    print(f"{name:24s} steps {n:3d}/{len(g):3d}  printed sigma_33 rows that differ: {int((u > 0.5).sum()):2d} (max {u.max():.0f} unit)   rel-L2 {np.linalg.norm(s[:m,2]-g[:m,2])/np.linalg.norm(g[:m,2]):.2e}   max|ds33|/max|g33| = {np.max(np.abs(s[:m,2]-g[:m,2]))/np.abs(g[:,2]).max():.2e}", flush=True)
    d.close()
