"""Generates tests/golden/oracle_curves/*.npz: full-length runs of the CPU oracle on the reference's regression option files.
They are the evidence that the oracle (and through it the ExaCMech restatement) is pinned to the reference's golden curves
over the WHOLE load history; tests/test_oracle_golden.py re-runs only the first steps and checks these stored curves.
Usage: python tests/golden/make_oracle_curves.py [case ...]      (about 15 minutes for all cases on one core)
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import orc  # noqa: E402

CASES = {"voce_pa": None, "voce_bcc": None, "voce_ea": None, "voce_nl_full": None, "mtsdd_full": None, "mtsdd_bcc": None, "voce_full_cyclic": None,
         "voce_ea_cs": None, "voce_full_cyclic_cs": None, "voce_full_cyclic_csm": None}

if __name__ == "__main__":
    names = sys.argv[1:] or list(CASES)
    orc.build()
    for name in names:
        case = orc.load_case(name + ".toml")
        out = orc.run_case(case)
        assert out["failed"] == 0, (name, out["failed"])
        np.savez(os.path.join(HERE, "oracle_curves", name + ".npz"), avg_stress=out["avg_stress"], avg_def_grad=out["avg_def_grad"],
                 avg_pl_work=out["avg_pl_work"], avg_dp_tensor=out["avg_dp_tensor"], newton_iters=out["newton_iters"],
                 krylov_iters=out["krylov_iters"])
        print(name, "done", out["t_total"], flush=True)
