"""Host logic that needs no GPU: the options.toml reader and the block domain decomposition of the stand-alone driver."""
import ctypes as C
import os

import numpy as np
import pytest

import partition_util as pu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "tests", "golden", "refdata")


def _q(name):
    import exaconstit_amd.lib as L
    out = np.zeros(20); err = C.create_string_buffer(512)
    rc = L.exa_options_query(os.path.join(REF, name).encode(), out.ctypes.data_as(C.POINTER(C.c_double)), err, 512)
    return rc, out, err.value.decode()


def test_options_reader_matches_reference_schema():
    rc, o, _ = _q("voce_pa.toml")
    assert rc == 0
    assert o[0] == 298 and o[1] == 17 and o[2] == 500 and o[3] == 0 and o[4] == 0
    assert o[5] == 1 and o[7] == 40                     # Time.Custom wins (src/mechanics_driver.cpp:842-851)
    assert o[8] == 0 and o[9] == 0 and o[10] == 25 and o[11] == 5e-5 and o[12] == 5e-10
    assert o[13] == 1000 and o[14] == 1e-7 and o[15] == 1e-27 and o[16] == 1 and o[17] == 5 and o[19] == 1
    rc, o, _ = _q("mtsdd_bcc.toml")
    assert rc == 0 and o[3] == 1 and o[4] == 2 and o[1] == 24 and o[8] == 1      # FULL -> element assembly operator
    rc, o, _ = _q("voce_full_cyclic.toml")
    assert rc == 0 and o[5] == 0 and o[7] == 70 and o[19] == 5
    rc, o, _ = _q("voce_ea.toml")
    assert rc == 0 and o[8] == 1 and o[18] == 1
    rc, o, _ = _q("mtsdd_full_auto.toml")
    assert rc == 0 and o[6] == 1


def test_options_reader_rejects_unsupported(tmp_path):
    txt = open(os.path.join(REF, "voce_ea_cs.toml")).read()
    import exaconstit_amd.lib as L

    def q(text):
        for fl in ("props_cp_voce.txt", "state_cp_voce.txt", "voce_quats.ori", "grains.txt", "custom_dt.txt"):
            text = text.replace('"%s"' % fl, '"%s"' % os.path.join(REF, fl))
        bad = tmp_path / "bad.toml"
        bad.write_text(text)
        out = np.zeros(20); err = C.create_string_buffer(512)
        return L.exa_options_query(str(bad).encode(), out.ctypes.data_as(C.POINTER(C.c_double)), err, 512), err.value.decode()

    rc, _, msg = _q("voce_ea_cs.toml")
    assert rc == 0                                           # velocity-gradient BCs (negative essential_comps) parse
    rc, msg = q(txt.replace("essential_vel_grad", "unused_key"))
    assert rc == -1 and "essential_vel_grad was not provided" in msg      # reference src/option_parser.cpp:217-219
    rc, msg = q(txt.replace('mech_type = "exacmech"', 'mech_type = "umat"'))
    assert rc == -1 and "exacmech" in msg
    rc, msg = q(txt.replace('type = "auto"', 'type = "other"').replace('floc = "../../data/cube-hex-ro.mesh"', 'floc = "%s"' % os.path.join(REF, "cube5_nodes.mesh")).replace("ref_ser = 1", "ref_ser = 0"))
    assert rc == 0, msg                                      # MFEM mesh v1.0 file meshes (reference src/mechanics_driver.cpp:239-241)
    rc, msg = q(txt.replace('type = "auto"', 'type = "other"'))
    assert rc == -1 and "ref_ser" in msg                     # uniform refinement of file meshes is not built: loud
    rc, msg = q(txt.replace('assembly = "EA"', 'assembly = "PA"\n    integ_model = "BBAR"'))
    assert rc == -1 and "BBAR" in msg                        # no partial-assembly gradient for B-bar (reference README.md:20)
    # solver keys are validated like the reference does (src/option_parser.cpp:616-662): nothing silently becomes NR / CG
    rc, msg = q(txt.replace('solver = "PCG"', 'solver = "GMRES"'))
    assert rc == -1 and "GMRES" in msg.upper() and "PCG" in msg
    rc, msg = q(txt.replace('solver = "PCG"', 'solver = "MINRES"'))
    assert rc == -1 and "not built" in msg
    rc, msg = q("\n".join(l for l in txt.splitlines() if not l.strip().startswith("solver =")))
    assert rc == -1 and "GMRES" in msg.upper()              # the reference's default when the key is missing
    rc, msg = q(txt.replace('solver = "PCG"', 'solver = "BiCG"'))
    assert rc == -1 and "valid type" in msg
    rc, msg = q(txt.replace("[Solvers.NR]", '[Solvers.NR]\n        nl_solver = "newton"'))
    assert rc == -1 and "nl_solver" in msg


def test_auto_time_stepping_with_changing_bcs_is_refused(tmp_path):
    """reference src/option_parser.cpp:509-511"""
    import exaconstit_amd.lib as L
    text = open(os.path.join(REF, "voce_full_cyclic.toml")).read()
    for fl in ("props_cp_voce.txt", "state_cp_voce.txt", "voce_quats.ori", "grains.txt", "custom_dt.txt"):
        text = text.replace('"%s"' % fl, '"%s"' % os.path.join(REF, fl))
    assert "[Time.Fixed]" in text
    text = text.replace("[Time.Fixed]", "[Time.Auto]\n        dt_start = 0.1\n        dt_min = 0.05\n        t_final = 7.0\n    [Time.Fixed]")
    bad = tmp_path / "bad.toml"; bad.write_text(text)
    out = np.zeros(20); err = C.create_string_buffer(512)
    rc = L.exa_options_query(str(bad).encode(), out.ctypes.data_as(C.POINTER(C.c_double)), err, 512)
    assert rc == -1 and "changing boundary conditions" in err.value.decode()


@pytest.mark.parametrize("nranks,order", [(1, 1), (2, 1), (3, 1), (4, 1), (8, 1), (1, 2), (4, 2)])
def test_block_decomposition_is_a_partition(nranks, order):
    N = (6, 5, 4)
    parts = [pu.query(N, r, nranks, order) for r in range(nranks)]
    gids = np.concatenate([p["gid"] for p in parts])
    assert sorted(gids.tolist()) == list(range(N[0] * N[1] * N[2]))            # every element exactly once
    nn_glob = (N[0] * order + 1) * (N[1] * order + 1) * (N[2] * order + 1)
    wsum = np.zeros(nn_glob)
    for p in parts:
        g = pu.global_node_ids(p, N)
        assert len(set(g.tolist())) == p["NN"]
        np.add.at(wsum, g, p["weight"])
        # connectivity: native vertex order, unit cells
        X = p["X"]; c = p["conn"]
        d = X[:, c[:, 6]] - X[:, c[:, 0]]
        assert np.allclose(d, np.array([[1 / N[0]], [1 / N[1]], [1 / N[2]]]))
        if order == 2:      # native order: the volume-interior node (index 26) is the cell centre, edge 0-1 midpoint is node 8
            assert np.allclose(X[:, c[:, 26]], 0.5 * (X[:, c[:, 0]] + X[:, c[:, 6]]))
            assert np.allclose(X[:, c[:, 8]], 0.5 * (X[:, c[:, 0]] + X[:, c[:, 1]]))
            assert np.allclose(X[:, c[:, 20]], 0.25 * (X[:, c[:, 0]] + X[:, c[:, 1]] + X[:, c[:, 2]] + X[:, c[:, 3]]))
    assert np.allclose(wsum, 1.0)                                               # duplicated nodes count once in dot products
    # neighbour lists are symmetric and enumerate the same global dofs in the same order on both sides
    for r, p in enumerate(parts):
        g = pu.global_node_ids(p, N)
        for (r2, dofs) in p["nbrs"]:
            q = parts[r2]
            back = [d for (rr, d) in q["nbrs"] if rr == r]
            mine = [(g[d % p["NN"]], d // p["NN"]) for d in dofs]
            g2 = pu.global_node_ids(q, N)
            found = any([(g2[d % q["NN"]], d // q["NN"]) for d in b] == mine for b in back)
            assert found, (r, r2)


def test_tail_split_controller():
    """Histograms measured on the 24^3 RVE (scripts/nfev_hist.py): the controller leaves the narrow Voce distribution uncapped, cuts the
    BCC Kocks-Mecking launch at 4 evaluations (1.3 % of the points carry counts of 5-13) and the broad FCC one at 6."""
    import exaconstit_amd.lib as L

    def cap(counts, w):
        h = (C.c_int * 64)(*([0] * 64))
        for n, c in counts.items():
            h[n] = c
        return L.exa_choose_newton_cap(h, w)
    assert cap({5: 97598, 6: 12531, 7: 463}, 4.0) == 0
    assert cap({3: 87029, 4: 21603, 5: 404, 6: 191, 7: 196, 8: 42, 9: 1, 10: 8, 11: 160, 12: 197, 13: 222}, 1.5) == 4
    assert cap({3: 36464, 4: 29018, 5: 16278, 6: 8254, 7: 2499, 8: 1509, 9: 2919, 10: 4377, 11: 4438, 12: 2875, 13: 1529}, 1.5) == 6
    assert cap({}, 1.5) == 0 and cap({4: 1000}, 1.5) == 0

    # resumed tail points (exa_set_newton_caps): first and second cap; 128^3 histograms of profiles/r03_bench_n128_{fcc,bcc}_kmdd.json
    def caps(counts, w):
        h = (C.c_int * 64)(*([0] * 64))
        for n, c in counts.items():
            h[n] = c
        k1, k2 = C.c_int(-1), C.c_int(-1)
        assert L.exa_choose_newton_caps(h, w, C.byref(k1), C.byref(k2)) == 0
        return k1.value, k2.value
    assert caps({5: 15417772, 6: 1353423, 7: 6021}, 4.0) == (0, 0)
    fcc = {3: 5727094, 4: 5375245, 5: 1931232, 6: 1105587, 7: 320681, 8: 209811, 9: 440309, 10: 584807, 11: 542477, 12: 332173, 13: 168195, 14: 36806, 15: 2359, 16: 383, 17: 57}
    bcc = {3: 13666837, 4: 2823624, 5: 48561, 6: 32733, 7: 29108, 8: 5166, 9: 328, 10: 1903, 11: 24858, 12: 28558, 13: 33761, 14: 33583, 15: 28132, 16: 12739, 17: 4533, 18: 2587, 19: 200, 20: 4, 21: 1}
    k1, k2 = caps(fcc, 1.0)
    assert 3 <= k1 <= 6 and k1 + 2 <= k2 <= 13, (k1, k2)      # broad main mode, second mode near 10: two dense launches
    k1, k2 = caps(bcc, 1.0)
    assert k1 in (3, 4) and (k2 == 0 or k2 >= k1 + 2), (k1, k2)
    assert caps({}, 1.0) == (0, 0) and caps({4: 1000}, 1.0) == (0, 0)


@pytest.mark.parametrize("mesh,nranks", [("cube5_shuffled.mesh", 2), ("cube5_shuffled.mesh", 3), ("cube5_nodes.mesh", 8)])
def test_file_mesh_partition_invariants(mesh, nranks):
    """Partition of an MFEM mesh file (recursive coordinate bisection, exa_mesh_partition_query): every element on exactly one rank,
    balanced counts, weights of a node's copies sum to one, neighbour lists symmetric and in the same (global node id) order on
    both sides, local coordinates = the file's coordinates."""
    import ctypes as C
    import exaconstit_amd.lib as L
    path = os.path.join(REF, mesh).encode()
    vp = lambda a: a.ctypes.data_as(C.c_void_p)

    def query(rank, nr):
        info = (C.c_int64 * 8)(); err = C.create_string_buffer(256)
        assert L.exa_mesh_partition_query(path, rank, nr, info, None, None, None, None, None, None, None, err, 256) == 0, err.value
        E, NN, nnb, shared, n = info[0], info[1], info[2], info[6], info[7]
        conn = np.zeros(n * E, np.int32); X = np.zeros(3 * NN); gid = np.zeros(E, np.int64); w = np.zeros(NN)
        nrk = np.zeros(max(nnb, 1), np.int32); ncnt = np.zeros(max(nnb, 1), np.int32); nd = np.zeros(max(shared, 1), np.int32)
        assert L.exa_mesh_partition_query(path, rank, nr, info, vp(conn), vp(X), vp(gid), vp(w), vp(nrk), vp(ncnt), vp(nd), err, 256) == 0
        nb = {}; off = 0
        for i in range(nnb):
            nb[int(nrk[i])] = nd[off:off + ncnt[i]].copy(); off += ncnt[i]
        return dict(E=E, NN=NN, conn=conn.reshape(E, n), X=X.reshape(3, NN), gid=gid, w=w, nb=nb)

    whole = query(0, 1)
    assert whole["E"] == 125 and whole["NN"] == 216 and not whole["nb"] and np.all(whole["w"] == 1.0)
    parts = [query(r, nranks) for r in range(nranks)]
    gids = np.concatenate([p["gid"] for p in parts])
    assert sorted(gids.tolist()) == list(range(125))
    counts = [p["E"] for p in parts]
    assert max(counts) - min(counts) <= max(1, nranks // 2)
    key = lambda X: [tuple(np.round(X[:, i], 9)) for i in range(X.shape[1])]
    wsum = {}
    for p in parts:
        # local element geometry = the file's element geometry
        for le, ge in enumerate(p["gid"]):
            assert np.allclose(p["X"][:, p["conn"][le]], whole["X"][:, whole["conn"][ge]], atol=0)
        for k, w in zip(key(p["X"]), p["w"]):
            wsum[k] = wsum.get(k, 0.0) + w
    assert len(wsum) == 216 and all(abs(v - 1.0) < 1e-12 for v in wsum.values())
    for r, p in enumerate(parts):
        for r2, dofs in p["nb"].items():
            other = parts[r2]["nb"][r]
            assert len(dofs) == len(other) and len(dofs) % 3 == 0
            m = len(dofs) // 3
            a = p["X"][:, dofs[:m] % p["NN"]]; b = parts[r2]["X"][:, other[:m] % parts[r2]["NN"]]
            assert np.array_equal(a, b)                                   # same nodes in the same order on both sides
            assert np.array_equal(dofs[m:2 * m] - dofs[:m], np.full(m, p["NN"]))
    err = C.create_string_buffer(256); info = (C.c_int64 * 8)()
    assert L.exa_mesh_partition_query(b"/nonexistent.mesh", 0, 2, info, None, None, None, None, None, None, None, err, 256) == -1 and b"Cannot open" in err.value


@pytest.mark.parametrize("style", ["mpich", "torchrun", "exa"])
def test_launcher_bootstrap_rendezvous(style):
    """`mpirun -np N mechanics -opt ...` without MPI in the executable (reference src/mechanics_driver.cpp:119-150): rank and size are read
    from whatever the launcher exports, the RCCL unique id travels from rank 0 to every rank over the TCP rendez-vous of
    csrc/host/bootstrap.cpp.  Three real processes, started in reverse order (peers before rank 0: they must retry), no GPU."""
    import socket
    import subprocess
    import sys
    import time
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "bootstrap_worker.py")
    names = {"mpich": ("PMI_RANK", "PMI_SIZE", "MPI_LOCALRANKID"), "torchrun": ("RANK", "WORLD_SIZE", "LOCAL_RANK"), "exa": ("EXA_RANK", "EXA_NRANKS", "EXA_LOCAL_RANK")}[style]
    procs = []
    for rank in (2, 1, 0):
        env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "PMI_RANK", "PMI_SIZE", "EXA_RANK", "EXA_NRANKS", "MASTER_PORT", "MASTER_ADDR")}
        env.update({names[0]: str(rank), names[1]: "3", names[2]: str(rank), "EXA_MASTER_PORT": str(port)})
        if style == "torchrun":
            env["MASTER_ADDR"] = "127.0.0.1"      # RANK / WORLD_SIZE count only with a rendez-vous address, which torchrun always exports
        procs.append((rank, subprocess.Popen([sys.executable, worker], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)))
        if rank == 1:
            time.sleep(0.3)
    for rank, p in procs:
        out, err = p.communicate(timeout=120)
        assert p.returncode == 0, (rank, out, err)
        assert "rank %d of 3 local %d payload_ok 1" % (rank, rank) in out


def _ident_table(recs):
    import ctypes as C
    t = (C.c_ubyte * (96 * len(recs)))()
    for i, (h, d) in enumerate(recs):
        for k, b in enumerate(h.encode()[:63]): t[96 * i + k] = b
        for k, b in enumerate(d.encode()[:31]): t[96 * i + 64 + k] = b
    return t


def test_transport_decided_from_device_identities(monkeypatch):
    """exa_bootstrap picks the transport from (host name, PCI bus id) of every rank, not from rank and device COUNTS: 16 ranks on 2 nodes x 8
    GPUs (more ranks than any rank sees devices) and one visible device per rank (srun --gpus-per-task=1) are RCCL launches; only ranks of one
    host that share a device get the shared-device transport, and that transport is refused across hosts."""
    import ctypes as C
    import exaconstit_amd.lib as L
    monkeypatch.delenv("EXA_TRANSPORT", raising=False)
    err = C.create_string_buffer(256)
    dec = lambda recs: L.exa_transport_from_identities(_ident_table(recs), len(recs), err, 256)
    gpus = ["0000:%02x:00.0" % b for b in (0x05, 0x15, 0x65, 0x75, 0x85, 0x95, 0xe5, 0xf5)]
    assert dec([("nodeA", g) for g in gpus]) == 1                                             # one node, 8 GPUs, 8 ranks
    assert dec([("nodeA", g) for g in gpus] + [("nodeB", g) for g in gpus]) == 1              # 2 x 8: same bus ids on the other host are other GPUs
    assert dec([("box", gpus[0]), ("box", gpus[0])]) == 2                                     # two ranks on the one GPU of a test box
    assert dec([("box", gpus[0])] * 8) == 2
    assert dec([("nodeA", gpus[0]), ("nodeA", gpus[0]), ("nodeB", gpus[0])]) == -1 and b"one host" in err.value      # shared device + several hosts
    monkeypatch.setenv("EXA_TRANSPORT", "ipc")
    assert dec([("nodeA", g) for g in gpus]) == 2                                             # forced, one host: allowed (plumbing runs)
    assert dec([("nodeA", gpus[0]), ("nodeB", gpus[0])]) == -1
    monkeypatch.setenv("EXA_TRANSPORT", "rccl")
    assert dec([("box", gpus[0]), ("box", gpus[0])]) == -1 and b"same device" in err.value    # RCCL refuses two ranks on a device: say so early
    assert dec([("nodeA", gpus[0]), ("nodeB", gpus[0])]) == 1


@pytest.mark.parametrize("case", ["two_nodes", "shared_device"])
def test_bootstrap_gathers_identities(case):
    """The rendez-vous in its gather-and-reply form (what exa_bootstrap runs): four real processes contribute an identity each, rank 0
    decides, every rank receives the same decision and table.  Peers start before rank 0 (they retry).  No GPU."""
    import socket
    import subprocess
    import sys
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "bootstrap_worker.py")
    ident = {"two_nodes": [("n0", "0000:05:00.0"), ("n0", "0000:15:00.0"), ("n1", "0000:05:00.0"), ("n1", "0000:15:00.0")],
             "shared_device": [("box", "0000:05:00.0")] * 4}[case]
    procs = []
    for rank in (3, 2, 1, 0):
        env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "PMI_RANK", "PMI_SIZE", "EXA_RANK", "EXA_NRANKS", "MASTER_PORT", "MASTER_ADDR", "EXA_TRANSPORT")}
        env.update({"EXA_RANK": str(rank), "EXA_NRANKS": "4", "EXA_MASTER_PORT": str(port), "FAKE_HOST": ident[rank][0], "FAKE_PCI": ident[rank][1]})
        procs.append((rank, subprocess.Popen([sys.executable, worker, "gather"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)))
    want = "kind %d hosts %s pcis %s" % (1 if case == "two_nodes" else 2, ",".join(h for h, _ in ident), ",".join(d for _, d in ident))
    for rank, p in procs:
        out, err = p.communicate(timeout=120)
        assert p.returncode == 0, (rank, out, err)
        assert ("rank %d of 4 " % rank) + want in out, (out, err)


def test_bootstrap_without_launcher():
    import ctypes as C
    import exaconstit_amd.lib as L
    saved = {k: os.environ.pop(k) for k in ("RANK", "WORLD_SIZE", "PMI_RANK", "PMI_SIZE", "EXA_RANK", "EXA_NRANKS", "SLURM_PROCID", "SLURM_NTASKS", "OMPI_COMM_WORLD_RANK", "OMPI_COMM_WORLD_SIZE") if k in os.environ}
    try:
        r, n, l = C.c_int(-1), C.c_int(-1), C.c_int(-1)
        assert L.exa_bootstrap_env(C.byref(r), C.byref(n), C.byref(l)) == 0 and (r.value, n.value, l.value) == (0, 1, 0)
        os.environ["EXA_RANK"] = "5"; os.environ["EXA_NRANKS"] = "4"
        assert L.exa_bootstrap_env(C.byref(r), C.byref(n), C.byref(l)) != 0      # rank outside the group
        os.environ.pop("EXA_RANK"); os.environ.pop("EXA_NRANKS")
        # generic variables that also exist outside a launch do not turn a plain run into rank 0 of N (an sbatch script without srun, a
        # container that exports WORLD_SIZE): they count only inside an srun step / with a rendez-vous address
        for extra, want in (({"SLURM_PROCID": "0", "SLURM_NTASKS": "4"}, (0, 1, 0)), ({"SLURM_PROCID": "1", "SLURM_NTASKS": "4", "SLURM_STEP_ID": "0", "SLURM_LOCALID": "1"}, (1, 4, 1)),
                            ({"RANK": "0", "WORLD_SIZE": "8"}, (0, 1, 0)), ({"RANK": "3", "WORLD_SIZE": "8", "LOCAL_RANK": "3", "MASTER_ADDR": "127.0.0.1"}, (3, 8, 3)),
                            ({"PMI_RANK": "2", "PMI_SIZE": "4", "WORLD_SIZE": "16"}, (2, 4, 2))):      # rank and size from ONE family
            keep = {k: os.environ.pop(k) for k in ("MASTER_ADDR", "EXA_MASTER_ADDR", "SLURM_STEP_ID") if k in os.environ}
            os.environ.update(extra)
            try:
                assert L.exa_bootstrap_env(C.byref(r), C.byref(n), C.byref(l)) == 0 and (r.value, n.value, l.value) == want, (extra, r.value, n.value, l.value)
            finally:
                for k in extra:
                    os.environ.pop(k, None)
                os.environ.update(keep)
        for bad in ({"EXA_RANK": "1", "EXA_NRANKS": "2x"}, {"PMI_RANK": "1"}):      # trailing garbage / half a family
            os.environ.update(bad)
            try:
                assert L.exa_bootstrap_env(C.byref(r), C.byref(n), C.byref(l)) != 0, bad
            finally:
                for k in bad:
                    os.environ.pop(k, None)
    finally:
        os.environ.pop("EXA_RANK", None); os.environ.pop("EXA_NRANKS", None); os.environ.update(saved)


@pytest.mark.parametrize("nranks", [2, 8, 12])
def test_boundary_first_element_order(nranks):
    """Several ranks: the driver permutes its local elements so that those touching a node shared with another rank come first
    (Partition::order_boundary_first) - the operator action computes these blocks, starts the halo exchange and overlaps it with the
    interior blocks.  The permutation keeps the set of elements, and the split is exact."""
    import ctypes as C
    import exaconstit_amd.lib as L
    N = (8, 8, 12)
    Nc = (C.c_int * 3)(*N)
    for r in range(nranks):
        base = pu.query(N, r, nranks)
        out = (C.c_int64 * 2)()
        conn = np.zeros(8 * base["E"], np.int32); gid = np.zeros(base["E"], np.int64)
        assert L.exa_partition_query_boundary_first(Nc, r, nranks, 1, out, conn.ctypes.data_as(C.c_void_p), gid.ctypes.data_as(C.c_void_p)) == 0
        E, Eb = out[0], out[1]
        assert E == base["E"] and sorted(gid.tolist()) == sorted(base["gid"].tolist())
        shared = np.zeros(base["NN"], bool)
        for _, dofs in base["nbrs"]:
            shared[dofs % base["NN"]] = True
        touches = shared[conn.reshape(E, 8)].any(axis=1)
        assert touches[:Eb].all() and not touches[Eb:].any() and 0 < Eb < E
        # connectivity rows moved with their elements
        lookup = {int(g): base["conn"][i] for i, g in enumerate(base["gid"])}
        assert all(np.array_equal(conn.reshape(E, 8)[i], lookup[int(g)]) for i, g in enumerate(gid))
        # stable: the original relative order is kept inside both groups
        pos = {int(g): i for i, g in enumerate(base["gid"])}
        for seg in (gid[:Eb], gid[Eb:]):
            p = [pos[int(g)] for g in seg]
            assert p == sorted(p)


def _mesh_query(path, rank, nr, order):
    import ctypes as C
    import exaconstit_amd.lib as L
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    info = (C.c_int64 * 8)(); err = C.create_string_buffer(256)
    assert L.exa_mesh_partition_query_order(path, rank, nr, order, info, None, None, None, None, None, None, None, err, 256) == 0, err.value
    E, NN, nnb, shared, n = info[0], info[1], info[2], info[6], info[7]
    conn = np.zeros(n * E, np.int32); X = np.zeros(3 * NN); gid = np.zeros(E, np.int64); w = np.zeros(NN)
    nrk = np.zeros(max(nnb, 1), np.int32); ncnt = np.zeros(max(nnb, 1), np.int32); nd = np.zeros(max(shared, 1), np.int32)
    assert L.exa_mesh_partition_query_order(path, rank, nr, order, info, vp(conn), vp(X), vp(gid), vp(w), vp(nrk), vp(ncnt), vp(nd), err, 256) == 0
    nb = {}; off = 0
    for i in range(nnb):
        nb[int(nrk[i])] = nd[off:off + ncnt[i]].copy(); off += ncnt[i]
    return dict(E=E, NN=NN, n=n, conn=conn.reshape(E, n), X=X.reshape(3, NN), gid=gid, w=w, nb=nb)


def _generated(N, order):
    import ctypes as C
    import exaconstit_amd.lib as L
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    Nn = (C.c_int * 3)(N, N, N); info = (C.c_int64 * 8)(); info[7] = order
    assert L.exa_partition_query(Nn, 0, 1, info, None, None, None, None, None, None, None) == 0
    E, NN, n = info[0], info[1], info[7]
    conn = np.zeros(n * E, np.int32); X = np.zeros(3 * NN); gid = np.zeros(E, np.int64); w = np.zeros(NN)
    info[7] = order
    assert L.exa_partition_query(Nn, 0, 1, info, vp(conn), vp(X), vp(gid), vp(w), None, None, None) == 0
    return dict(E=E, NN=NN, n=n, conn=conn.reshape(E, n), X=X.reshape(3, NN))


def test_file_mesh_at_order_two_equals_the_generated_mesh():
    """p_refinement = 2 on an MFEM mesh file: one node per edge / face / element of the trilinear mesh, numbered like the generated
    triquadratic mesh (vertices, edges, faces, centre) - every element of the 5^3 file mesh carries the node coordinates of the generated
    5^3 p = 2 mesh, and a partition of it keeps the invariants of the p = 1 partition."""
    path = os.path.join(REF, "cube5_nodes.mesh").encode()
    f1 = _mesh_query(path, 0, 1, 1); f2 = _mesh_query(path, 0, 1, 2)
    assert f1["n"] == 8 and f2["n"] == 27 and f2["E"] == 125 and f2["NN"] == 11 ** 3
    g = _generated(5, 2)
    assert g["n"] == 27 and g["NN"] == 11 ** 3
    scale = f1["X"].max()
    for e in range(125):
        assert np.allclose(f2["X"][:, f2["conn"][e]], scale * g["X"][:, g["conn"][e]], atol=1e-14 * scale)
    assert np.array_equal(f2["conn"][:, :8], f1["conn"])
    path = os.path.join(REF, "cube5_shuffled.mesh").encode()
    for nranks in (2, 3):
        parts = [_mesh_query(path, r, nranks, 2) for r in range(nranks)]
        assert sorted(np.concatenate([p["gid"] for p in parts]).tolist()) == list(range(125))
        wsum = {}
        for p in parts:
            for i in range(p["NN"]):
                k = tuple(np.round(p["X"][:, i], 9)); wsum[k] = wsum.get(k, 0.0) + p["w"][i]
        assert len(wsum) == 11 ** 3 and all(abs(v - 1.0) < 1e-12 for v in wsum.values())
        for r, p in enumerate(parts):
            for r2, dofs in p["nb"].items():
                other = parts[r2]["nb"][r]; m = len(dofs) // 3
                assert len(dofs) == len(other)
                assert np.array_equal(p["X"][:, dofs[:m] % p["NN"]], parts[r2]["X"][:, other[:m] % parts[r2]["NN"]])


@pytest.mark.parametrize("order", [3, 4])
def test_file_mesh_at_higher_order_equals_the_generated_mesh(order):
    """p_refinement = 3, 4 on an MFEM mesh file (host/mesh.hpp, elevate_to_order): (p-1) nodes per edge, (p-1)^2 per face, (p-1)^3 per element at the
    Gauss-Lobatto points, shared nodes identified independently of the elements' local directions - checked on the file whose elements AND local
    vertex orders are shuffled: every element carries the node coordinates of the generated mesh's element of the same global index, the node
    count is the conforming one ((5p+1)^3: no duplicated edge / face node), and a partition keeps the invariants of the p = 1 partition."""
    p = order
    for name in ("cube5_nodes.mesh", "cube5_shuffled.mesh"):
        path = os.path.join(REF, name).encode()
        f1 = _mesh_query(path, 0, 1, 1); fp = _mesh_query(path, 0, 1, p)
        assert fp["n"] == (p + 1) ** 3 and fp["E"] == 125 and fp["NN"] == (5 * p + 1) ** 3
        g = _generated(5, p)
        scale = f1["X"].max()
        # element correspondence by centroid (the shuffled file permutes elements and rotates their local vertex order)
        cen_g = {tuple(np.round(scale * g["X"][:, g["conn"][e]].mean(axis=1), 9)): e for e in range(125)}
        for e in range(125):
            xe = fp["X"][:, fp["conn"][e]]
            eg = cen_g[tuple(np.round(xe.mean(axis=1), 9))]
            xg = scale * g["X"][:, g["conn"][eg]]
            if name == "cube5_nodes.mesh":      # same local orientation: node by node
                assert np.allclose(xe, xg, atol=1e-13 * scale)
            a = {tuple(r) for r in np.round(xe.T, 9)}; b = {tuple(r) for r in np.round(xg.T, 9)}
            assert a == b                          # the same (p+1)^3 points in either orientation
        assert np.array_equal(fp["conn"][:, :8], f1["conn"])
    path = os.path.join(REF, "cube5_shuffled.mesh").encode()
    for nranks in (2, 3):
        parts = [_mesh_query(path, r, nranks, p) for r in range(nranks)]
        assert sorted(np.concatenate([q["gid"] for q in parts]).tolist()) == list(range(125))
        wsum = {}
        for q in parts:
            for i in range(q["NN"]):
                k = tuple(np.round(q["X"][:, i], 9)); wsum[k] = wsum.get(k, 0.0) + q["w"][i]
        assert len(wsum) == (5 * p + 1) ** 3 and all(abs(v - 1.0) < 1e-12 for v in wsum.values())
        for r, q in enumerate(parts):
            for r2, dofs in q["nb"].items():
                other = parts[r2]["nb"][r]; m = len(dofs) // 3
                assert len(dofs) == len(other)
                assert np.array_equal(q["X"][:, dofs[:m] % q["NN"]], parts[r2]["X"][:, other[:m] % parts[r2]["NN"]])


def test_generated_mesh_nodes_sit_at_gauss_lobatto_points():
    """Orders above 2: the nodes of a generated mesh are the Gauss-Lobatto-Legendre points of each element (MFEM's H1 basis), not equispaced."""
    g = _generated(2, 4)
    assert g["n"] == 125 and g["NN"] == 9 ** 3
    xs = np.unique(np.round(g["X"][0], 12))
    gll5 = 0.5 * (1.0 + np.array([-1.0, -np.sqrt(3.0 / 7.0), 0.0, np.sqrt(3.0 / 7.0), 1.0]))
    expect = np.unique(np.round(np.concatenate([0.5 * gll5, 0.5 + 0.5 * gll5]), 12))
    assert np.allclose(xs, expect, atol=1e-12)
    g3 = _generated(1, 3)
    assert np.allclose(np.unique(np.round(g3["X"][0], 12)), [0.0, 0.5 - 0.5 / np.sqrt(5.0), 0.5 + 0.5 / np.sqrt(5.0), 1.0], atol=1e-12)
