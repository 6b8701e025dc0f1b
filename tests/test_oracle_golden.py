"""Pins the CPU oracle to the reference's own golden vectors (reference test/data/*_stress.txt etc., 6 printed digits;
the reference's own check is equality of the printed text, test/test_mechanics.py:22-30).

Bar: every printed number of every golden file is reproduced to within ONE unit of its last printed digit (most rows
are identical text), and sigma_33 within 3e-6 rel-L2 (the 6-digit quantisation floor is 4e-7 ... 2.4e-6).

Two layers so that the CPU suite stays within minutes:
  * live: the first steps of every regression case are re-run on the oracle and compared with the golden rows;
  * stored: full-length oracle runs (tests/golden/oracle_curves/*.npz, produced by tests/golden/make_oracle_curves.py) are
    compared with the complete golden files, and the live rows must reproduce the stored rows.
"""
import os

import numpy as np
import pytest

CURVES = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_curves")


def printed_ulps(orc, s, g, floor=1e-6):
    """|round6(s) - g| in units of g's last printed digit, for the entries that are not round-off noise
    (|g| > floor * max|g| of the file); returns (ulps of those entries, max abs difference of the rest)."""
    s = np.asarray(s, dtype=np.float64).reshape(np.shape(g))
    big = np.abs(g) > floor * np.abs(g).max()
    unit = 10.0 ** (np.floor(np.log10(np.abs(np.where(big, g, 1.0)))) - 5)
    u = np.abs(orc.fmt6(s) - g) / unit
    rest = np.abs(s - g)[~big].max() if (~big).any() else 0.0
    return u[big], rest


def col_unit(col):
    """One unit of the last printed digit of the column's largest entry."""
    return 10.0 ** (np.floor(np.log10(np.abs(col).max())) - 5)


def check_file(orc, s, gold_name, max_ulp=1.0, rest_tol=2e-8):
    g = orc.golden(gold_name)
    u, rest = printed_ulps(orc, s, g)
    assert u.max() <= max_ulp + 1e-6, (gold_name, u.max(), int((u > 0).sum()), u.size)
    assert rest < rest_tol * max(np.abs(g).max(), 1e-30) + 1e-300, (gold_name, rest)
    return u


LIVE = [("voce_pa", "voce_pa", 4), ("voce_ea_cs", "voce_ea_cs", 3), ("voce_bcc", "voce_bcc", 3), ("voce_nl_full", "voce_full", 3),
        ("mtsdd_full", "mtsdd_full", 5), ("mtsdd_bcc", "mtsdd_bcc", 5)]


@pytest.mark.parametrize("name,gold,nsteps", LIVE)
def test_live_steps_match_golden(oracle, name, gold, nsteps):
    orc = oracle
    out = orc.run_case(orc.load_case(name + ".toml"), nsteps=nsteps)
    assert out["failed"] == 0
    g = orc.golden(gold + "_stress.txt")[:nsteps]
    s = out["avg_stress"]
    u, _ = printed_ulps(orc, s[:, 2], g[:, 2])
    assert u.max() <= 1.0 + 1e-6
    gfull = orc.golden(gold + "_stress.txt")
    for c in (3, 4, 5):     # shear averages cross zero: measure against the column's own scale
        assert np.max(np.abs(orc.fmt6(s[:, c]) - g[:, c])) <= 1.001 * col_unit(gfull[:, c])
    f = os.path.join(CURVES, name + ".npz")
    if os.path.exists(f):
        st = np.load(f)["avg_stress"][:nsteps]
        assert np.allclose(s, st, rtol=1e-9, atol=1e-16)


STORED = [("voce_pa", "voce_pa"), ("voce_bcc", "voce_bcc"), ("voce_nl_full", "voce_full"), ("voce_ea", "voce_ea"), ("voce_ea_cs", "voce_ea_cs"),
          ("mtsdd_full", "mtsdd_full"), ("mtsdd_bcc", "mtsdd_bcc"),
          ("voce_full_cyclic", "voce_full_cyclic"), ("voce_full_cyclic_cs", "voce_full_cyclic_cs"), ("voce_full_cyclic_csm", "voce_full_cyclic_csm")]


@pytest.mark.parametrize("name,gold", STORED)
def test_stored_full_curves_match_golden(oracle, name, gold):
    orc = oracle
    f = os.path.join(CURVES, name + ".npz")
    if not os.path.exists(f):
        pytest.skip("stored curve not generated")
    s = np.load(f)["avg_stress"]
    g = orc.golden(gold + "_stress.txt")
    assert s.shape == g.shape
    # sigma_33: the north-star bar (1e-6 of the CPU reference is below the files' own 6-digit resolution; 3e-6 rel-L2)
    assert np.linalg.norm(s[:, 2] - g[:, 2]) / np.linalg.norm(g[:, 2]) < 3e-6
    u, _ = printed_ulps(orc, s[:, 2], g[:, 2])
    assert u.max() <= 1.0 + 1e-6, (name, u.max())
    # monotonic cases: the printed sigma_33 column is reproduced digit for digit
    if "cyclic" not in name:
        assert int((u > 0).sum()) == 0, (name, int((u > 0).sum()))
    else:
        assert int((u > 0).sum()) <= 3
    # shear averages (1e-4 of sigma_33; they cross zero, so measure against the column's own scale)
    for c in (3, 4, 5):
        assert np.max(np.abs(orc.fmt6(s[:, c]) - g[:, c])) <= 1.001 * col_unit(g[:, c]), (name, c)
    # sigma_11, sigma_22 are solver noise around zero in both
    assert np.abs(s[:, :2]).max() < 1e-6 * np.abs(g[:, 2]).max() and np.abs(g[:, :2]).max() < 1e-6 * np.abs(g[:, 2]).max()


@pytest.mark.parametrize("name", ["voce_ea", "voce_ea_cs"])
def test_stored_extra_outputs(oracle, name):
    """def_grad / pl_work / dp_tensor files of the EA cases (reference src/system_driver.cpp:470-553), velocity- and
    velocity-gradient-driven: every printed number within one unit of its last digit."""
    orc = oracle
    f = os.path.join(CURVES, name + ".npz")
    if not os.path.exists(f):
        pytest.skip("stored curve not generated")
    z = np.load(f)
    check_file(orc, z["avg_def_grad"], name + "_def_grad.txt")
    check_file(orc, z["avg_pl_work"], name + "_pl_work.txt")
    check_file(orc, z["avg_dp_tensor"], name + "_dp_tensor.txt")


def test_cyclic_reversal_branches(oracle):
    """Load reversals (BC-change corrector, reference src/system_driver.cpp:293-319).  The elastic unloading branches and the
    re-yield transients are what pins the strain state's a_V scaling (oracle/ecmech_port.hpp, struct Problem): with the
    begin-of-step strain converted by the END-of-step a_V every branch is reproduced to the printed digits."""
    orc = oracle
    f = os.path.join(CURVES, "voce_full_cyclic.npz")
    if not os.path.exists(f):
        pytest.skip("stored curve not generated")
    s = np.load(f)["avg_stress"]
    g = orc.golden("voce_full_cyclic_stress.txt")
    assert np.max(np.abs(s[:, 2] - g[:, 2])) < 3e-6 * np.abs(g[:, 2]).max()
    for lo, hi in ((10, 14), (30, 34), (50, 54)):      # elastic unloading branches
        assert np.max(np.abs(s[lo:hi, 2] - g[lo:hi, 2])) < 1e-6 * np.abs(g[:, 2]).max()


def test_auto_time_stepping_case_elastic_branch(oracle):
    """mtsdd_full_auto (Time.Auto, IN625-like Kocks-Mecking properties, compression): the golden file holds stresses only, the step sizes
    depend on the Newton iteration counts k_n of the reference's FULL-assembly + BoomerAMG solve (dt_{n+1} = dt_n * 25 * 0.333333 / k_n,
    src/system_driver.cpp:263-269), which are not reproducible here.  The replay mode of the oracle (oracle/driver_port.hpp,
    run_case_replay) infers k_n row by row: it tries the 25 admissible step sizes and keeps the one whose stress INCREMENT matches the golden
    row.  On the elastic branch (rows 1-11) this pins the rule and the response: every increment is reproduced to < 0.02 MPa with the integer
    sequence k = 2, 24, 6, 6, 15, 6, 6, 9, 21, 7 (24 Newton iterations on an elastic step: the reference's linear solver, not the material).
    Row 2 of the golden file carries a one-off deficit of 2.01 MPa that persists unchanged through the elastic rows - an artefact of that
    barely-converged step of the reference, not a material response (no slip below |tau| = tau_a = 260 MPa)."""
    orc = oracle
    g = orc.golden("mtsdd_full_auto_stress.txt")[:, 2]
    out = orc.run_case(orc.load_case("mtsdd_full_auto.toml"), replay_target33=g[:11], replay_increments=True)
    s = out["avg_stress"][:, 2]
    assert len(s) == 11 and list(out["ks"][1:]) == [2, 24, 6, 6, 15, 6, 6, 9, 21, 7]
    assert abs(s[0] / g[0] - 1.0) < 2e-6
    off = s - g[:11]
    assert np.all(np.abs(off[1:9] - off[1]) < 0.01) and abs(off[1] + 2.01) < 0.01      # rows 2-9 (sigma from -107 to -439 MPa): < 3e-5 of the increments
    assert abs(off[9] - off[1]) < 0.06                                                  # row 10: first slip in the best-oriented grains


def _auto_inferred_steps(orc, g):
    """Step sizes of rows 1-11 of the Time.Auto case from the inferred Newton counts; dt_2 calibrated on row 2 (the reference moved the
    boundary 2.3 % less than v dt in that step: a one-off deficit of 2.01 MPa that every later elastic row carries)."""
    ks = [2, 24, 6, 6, 15, 6, 6, 9, 21, 7]
    dts = [0.1]
    for k in ks:
        dts.append(dts[-1] * 25 * 0.333333 / k)
    dts = np.array(dts)
    case = orc.load_case("mtsdd_full_auto.toml")
    case["auto"] = None; case["dts"] = dts[:2].copy()
    s = orc.run_case(case)["avg_stress"][:, 2]
    dts[1] = (g[1] - s[0]) / ((s[1] - s[0]) / dts[1])
    return dts


def test_auto_time_stepping_case_rows_with_inferred_steps(oracle):
    """With those step sizes the ABSOLUTE sigma_33 of rows 1-8 (elastic, -21 ... -380 MPa) is the golden file's to 0.003 MPa and row 9
    (first slip in the best-oriented grains) to 0.005 MPa - a much sharper statement than the increments of the replay."""
    orc = oracle
    g = orc.golden("mtsdd_full_auto_stress.txt")[:, 2]
    dts = _auto_inferred_steps(orc, g)
    assert abs(dts[1] / (0.1 * 25 * 0.333333 / 2) - 0.9771) < 2e-4
    case = orc.load_case("mtsdd_full_auto.toml")
    case["auto"] = None; case["dts"] = dts[:9].copy()
    out = orc.run_case(case)
    assert out["failed"] == 0
    d = out["avg_stress"][:, 2] - g[:9]
    assert np.abs(d[:8]).max() < 0.003 and abs(d[8]) < 0.006, d


@pytest.mark.xfail(reason="OPEN: thermally activated Kocks-Mecking regime (p = 0.8, q = 1.4, c_e = 26) is not pinned: the last row of the golden file is at "
                          "t = t_final = 10 exactly, where the oracle's sigma_33 is -725 MPa against the file's -773 MPa (the response there does not depend on "
                          "the step sizes: 20 or 200 steps agree to 0.1 MPa).  Bounded, not closed: structural variants of the law, replays and a stress-space fit "
                          "(no parameter set of this law traces the golden curve: 0.23 MPa rms at best) in DESIGN.md section 5, tools under scripts/auto_case_study", strict=False)
def test_auto_time_stepping_case_plastic_branch(oracle):
    orc = oracle
    g = orc.golden("mtsdd_full_auto_stress.txt")
    # incipient plasticity with the known step sizes: rows 10 and 11 are 0.04 and 0.13 MPa softer than the file (the reference slips
    # less at low rates; s x 2.9, c_1 x 6 or gam_wo / 3e4 each zero both rows, none of them reaches the final row)
    case = orc.load_case("mtsdd_full_auto.toml")
    case["auto"] = None; case["dts"] = _auto_inferred_steps(orc, g[:, 2])
    d = orc.run_case(case)["avg_stress"][:, 2] - g[:11, 2]
    assert np.abs(d[9:]).max() < 0.01, d[9:]
    case = orc.load_case("mtsdd_full_auto.toml")
    case["auto"] = None; case["dts"] = np.full(20, 0.5)
    out = orc.run_case(case)
    assert out["failed"] == 0
    assert abs(out["avg_stress"][-1, 2] / g[-1, 2] - 1.0) < 2e-3
