"""The multi-rank path of the stand-alone driver (block decomposition, halo-sum, weighted dots, reductions, BC masks and
velocity-gradient origin on partitions) on ONE GPU: R drivers, one host thread each, exchange through the in-process loopback
group (exa_loopback_group_create) instead of RCCL — RCCL refuses two ranks on one device.  Every line of the driver that the
RCCL build runs is exercised except the RCCL calls themselves.  Results must match the one-rank run: same Newton iteration
counts, volume averages to summation-order round-off."""
import ctypes as C
import os
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _run_ranks(L, toml, nranks, nsteps, tmp_path, jacobi=False):
    os.makedirs(str(tmp_path), exist_ok=True)
    gid = (C.c_ubyte * 128)()
    assert L.exa_loopback_group_create(nranks, gid) == 0
    drivers = [None] * nranks
    errors = []

    def work(r):
        try:
            d = L.Driver.from_toml(toml, out_dir=str(tmp_path), rank=r, nranks=nranks, uid=gid, jacobi=jacobi, write_files=(r == 0))
            drivers[r] = d
            for ti in range(1, nsteps + 1):
                if not d.step(ti):
                    raise RuntimeError(f"rank {r}: Newton failed at step {ti}")
        except Exception as e:   # noqa: BLE001
            errors.append((r, repr(e)))

    th = [threading.Thread(target=work, args=(r,)) for r in range(nranks)]
    [t.start() for t in th]
    [t.join(timeout=600) for t in th]
    assert not errors, errors
    assert all(not t.is_alive() for t in th), "a rank hung"
    out = [(d.avgs(0, 6), d.stats()) for d in drivers]
    for d in drivers:
        d.close()
    L.exa_loopback_group_destroy(gid)
    return out


@pytest.mark.parametrize("case,nranks", [("voce_pa", 2), ("voce_pa", 3), ("voce_pa", 8), ("voce_ea_cs", 4), ("voce_full_cyclic", 2), ("mtsdd_bcc", 4), ("mtsdd_full", 2)])
def test_partitioned_run_matches_single_rank(oracle, tmp_path, case, nranks):
    import exaconstit_amd.lib as L
    orc = oracle
    toml = os.path.join(orc.REFDATA, case + ".toml")
    nsteps = 12 if case == "voce_full_cyclic" else 5     # the cyclic case reverses the load at step 11 (BC-change corrector on partitions)
    ref = _run_ranks(L, toml, 1, nsteps, tmp_path / "r1")[0]
    got = _run_ranks(L, toml, nranks, nsteps, tmp_path / f"r{nranks}")
    for s, st in got:                                      # every rank reports the same global averages and solver history
        assert np.max(np.abs(s - ref[0])) < 1e-9 * np.abs(ref[0]).max()
        assert list(st[0]) == list(ref[1][0])              # Newton iterations


@pytest.mark.parametrize("case,nranks", [("voce_pa", 8), ("voce_ea", 4)])
def test_halo_overlap_switch(oracle, tmp_path, monkeypatch, case, nranks):
    """Several ranks: the action runs the element blocks that touch shared nodes first, starts the halo exchange on a second stream and
    computes the interior blocks meanwhile (NonlinearMechOperator::GradMult, Comm::halo_begin / halo_end).  EXA_HALO_OVERLAP=off runs
    the plain sequence (whole action, then exchange) on the same boundary-first element order: same Newton history, same averages."""
    import exaconstit_amd.lib as L
    orc = oracle
    toml = os.path.join(orc.REFDATA, case + ".toml")
    ref = _run_ranks(L, toml, 1, 5, tmp_path / "r1")[0]
    on = _run_ranks(L, toml, nranks, 5, tmp_path / "on")
    monkeypatch.setenv("EXA_HALO_OVERLAP", "off")
    off = _run_ranks(L, toml, nranks, 5, tmp_path / "off")
    # The loopback exchange is stream-asynchronous by default (host/driver.hip, Comm::exchange: peers' copies ordered by events, no stream is
    # drained - how the choreography of halo_begin / halo_end runs over RCCL); EXA_LOOPBACK_SYNC=1 is the host-synchronous form of rounds 1-4
    monkeypatch.delenv("EXA_HALO_OVERLAP")
    monkeypatch.setenv("EXA_LOOPBACK_SYNC", "1")
    sync = _run_ranks(L, toml, nranks, 5, tmp_path / "sync")
    for (s, st), (s2, st2), (s3, st3) in zip(on, off, sync):
        assert np.max(np.abs(s - s2)) < 1e-10 * np.abs(s).max() and np.max(np.abs(s - s3)) < 1e-10 * np.abs(s).max()
        assert list(st[0]) == list(st2[0]) == list(st3[0]) == list(ref[1][0])
        assert np.max(np.abs(s - ref[0])) < 1e-9 * np.abs(ref[0]).max()


def test_partitioned_order2_bbar_matches_single_rank(oracle, tmp_path):
    """BASELINE config 5 ingredients (p = 2, B-bar, element assembly, NRLS) on 2 and 4 ranks vs one rank."""
    import exaconstit_amd.lib as L
    from test_gpu_driver import _variant_toml
    edits = [('assembly = "PA"', 'assembly = "EA"\n    integ_model = "BBAR"'), ("prefinement = 1", "p_refinement = 2"), ("ref_ser = 1", "ref_ser = 0"),
             ("[Solvers.NR]", '[Solvers.NR]\n        nl_solver = "NRLS"')]
    os.makedirs(str(tmp_path), exist_ok=True)
    toml = _variant_toml(tmp_path, "voce_pa.toml", edits, "p2bbar")
    ref = _run_ranks(L, toml, 1, 3, tmp_path / "r1")[0]
    for nranks in (2, 4):
        for s, st in _run_ranks(L, toml, nranks, 3, tmp_path / f"r{nranks}"):
            assert np.max(np.abs(s - ref[0])) < 1e-9 * np.abs(ref[0]).max()
            assert list(st[0]) == list(ref[1][0])


def test_partitioned_order3_matches_single_rank(oracle, tmp_path):
    """p_refinement = 3 (run-time-order kernels, Gauss-Lobatto nodes) on 2 and 3 ranks vs one rank: shared faces carry 16 nodes per element face."""
    import exaconstit_amd.lib as L
    from test_gpu_driver import _variant_toml
    os.makedirs(str(tmp_path), exist_ok=True)
    toml = _variant_toml(tmp_path, "voce_pa.toml", [("prefinement = 1", "p_refinement = 3"), ("ref_ser = 1", "ref_ser = 0")], "p3")
    ref = _run_ranks(L, toml, 1, 3, tmp_path / "r1")[0]
    for nranks in (2, 3):
        for s, st in _run_ranks(L, toml, nranks, 3, tmp_path / f"r{nranks}"):
            assert np.max(np.abs(s - ref[0])) < 1e-9 * np.abs(ref[0]).max()
            assert list(st[0]) == list(ref[1][0])


@pytest.mark.parametrize("mesh,nranks,p", [("cube5_shuffled.mesh", 2, 1), ("cube5_shuffled.mesh", 3, 1), ("cube5_nodes.mesh", 5, 1), ("cube5_shuffled.mesh", 3, 2), ("cube5_shuffled.mesh", 2, 3)])
def test_file_mesh_partitioned_matches_single_rank(oracle, tmp_path, mesh, nranks, p):
    """Mesh.type = "other" on several ranks: recursive-coordinate-bisection partition of the file's elements (unstructured neighbour
    lists, nodes shared by up to 8 ranks) gives the one-rank run."""
    import exaconstit_amd.lib as L
    from test_gpu_driver import _variant_toml
    orc = oracle
    os.makedirs(str(tmp_path), exist_ok=True)
    toml = _variant_toml(tmp_path, "voce_pa.toml", [("ref_ser = 1", "ref_ser = 0"), ("prefinement = 1", "p_refinement = %d" % p), ('type = "auto"', 'type = "other"'),
                                                    ('floc = "../../data/cube-hex-ro.mesh"', 'floc = "%s"' % os.path.join(orc.REFDATA, mesh))], "file5")
    ref = _run_ranks(L, toml, 1, 4, tmp_path / "r1")[0]
    for s, st in _run_ranks(L, toml, nranks, 4, tmp_path / f"r{nranks}"):
        assert np.max(np.abs(s - ref[0])) < 1e-9 * np.abs(ref[0]).max()
        assert list(st[0]) == list(ref[1][0])


def test_deterministic_mode_is_bit_reproducible(oracle, tmp_path, monkeypatch):
    """EXA_DETERMINISTIC=1 (ordered E->L sums and halo additions): two runs of the same partitioned case give the same bits - averages,
    Newton and Krylov counts - on 1 and on 4 ranks; and the answers are the atomic path's to round-off."""
    import exaconstit_amd.lib as L
    orc = oracle
    toml = os.path.join(orc.REFDATA, "voce_pa.toml")
    base = _run_ranks(L, toml, 1, 4, tmp_path / "base")[0]
    monkeypatch.setenv("EXA_DETERMINISTIC", "1")
    for nranks in (1, 4):
        a = _run_ranks(L, toml, nranks, 4, tmp_path / f"a{nranks}")
        b = _run_ranks(L, toml, nranks, 4, tmp_path / f"b{nranks}")
        for (sa, sta), (sb, stb) in zip(a, b):
            assert np.array_equal(sa, sb)
            assert all(list(x) == list(y) for x, y in zip(sta, stb))
        assert np.max(np.abs(a[0][0] - base[0])) < 1e-9 * np.abs(base[0]).max()
