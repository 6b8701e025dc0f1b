"""N > 1 path on CPU: two `gloo` ranks reproduce the driver's multi-GPU algebra — block decomposition, duplicated interface
nodes, neighbour halo-SUM after the local scatter-add (Comm::halo_sum: pack all, exchange, unpack-add) and 1/multiplicity
weighted dot products + all-reduce — with the oracle's element kernels standing in for the HIP kernels, and must match the
single-domain result.  (RCCL itself can only be exercised on the multi-GPU box; the exchange pattern and index lists are the
same objects the GPU path uses: exa_partition_query exposes the driver's Partition.)"""
import ctypes as C
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _local_operator(orc, part, seed_by_gid):
    """element 'stiffness' matrices of the local elements from the oracle's AssembleEA with a seeded SPD tangent per element"""
    E = part["E"]; n = 8; Q = 8
    G = np.zeros(n * 3 * Q); W = np.zeros(Q); orc.lib().orc_ref_elem(1, orc._p(G), orc._p(W))
    conn = part["conn"]
    xe = np.zeros((E, 3, n))
    for c in range(3):
        xe[:, c, :] = part["X"][c][conn]
    J = np.zeros(9 * E * Q); orc.lib().orc_jacobians(1, E, orc._p(np.ascontiguousarray(xe.ravel())), orc._p(J))
    Cm = np.zeros((E, Q, 6, 6))
    for e in range(E):
        rng = np.random.default_rng(1000 + int(seed_by_gid[e]))
        A = rng.standard_normal((6, 6)); S = A @ A.T + 6 * np.eye(6)
        Cm[e, :] = S
    emat = np.zeros(9 * n * n * E)
    orc.lib().orc_assemble_ea(Q, E, n, C.c_double(0.1), orc._p(W), orc._p(G), orc._p(J), orc._p(np.ascontiguousarray(Cm.ravel())), orc._p(emat))
    return emat


def _apply_local(orc, part, emat, xL):
    E, NN = part["E"], part["NN"]; n = 8
    conn = part["conn"]
    xe = np.zeros((E, 3, n))
    for c in range(3):
        xe[:, c, :] = xL[conn + NN * c]
    ye = np.zeros(3 * n * E)
    orc.lib().orc_ea_mult(E, n, orc._p(emat), orc._p(np.ascontiguousarray(xe.ravel())), orc._p(ye))
    yL = np.zeros(3 * NN)
    for c in range(3):
        np.add.at(yL, conn + NN * c, ye.reshape(E, 3, n)[:, c, :])
    return yL


def _worker(rank, world, port, N, q):
    try:
        sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
        import torch
        import torch.distributed as dist
        import orc
        import partition_util as pu
        os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        part = pu.query(N, rank, world)
        ser = pu.query(N, 0, 1)
        gl = pu.global_node_ids(part, N)
        nn_g = ser["NN"]
        rng = np.random.default_rng(42)
        x_glob = rng.standard_normal(3 * nn_g)
        xL = np.concatenate([x_glob[gl + nn_g * c] for c in range(3)])
        emat = _local_operator(orc, part, part["gid"])
        yL = _apply_local(orc, part, emat, xL)
        # ---- halo-sum exactly as Comm::halo_sum: pack every buffer first, exchange, then unpack-add
        sends = [torch.from_numpy(yL[d].copy()) for (_, d) in part["nbrs"]]
        recvs = [torch.zeros(len(d), dtype=torch.float64) for (_, d) in part["nbrs"]]
        reqs = []
        for (r2, _), s, r in zip(part["nbrs"], sends, recvs):
            reqs.append(dist.isend(s, r2)); reqs.append(dist.irecv(r, r2))
        for rq in reqs:
            rq.wait()
        for (_, d), r in zip(part["nbrs"], recvs):
            np.add.at(yL, d, r.numpy())
        # ---- serial reference on the whole domain
        emat_s = _local_operator(orc, ser, ser["gid"])
        y_ser = _apply_local(orc, ser, emat_s, x_glob)
        y_ref = np.concatenate([y_ser[gl + nn_g * c] for c in range(3)])
        err = np.linalg.norm(yL - y_ref) / np.linalg.norm(y_ref)
        # ---- weighted dot + all-reduce
        w3 = np.tile(part["weight"], 3)
        t = torch.tensor([float(np.sum(w3 * xL * yL))], dtype=torch.float64)
        dist.all_reduce(t)
        derr = abs(t.item() - float(x_glob @ y_ser)) / abs(float(x_glob @ y_ser))
        q.put((rank, err, derr, None))
        dist.destroy_process_group()
    except Exception as e:   # pragma: no cover
        import traceback
        q.put((rank, 1.0, 1.0, traceback.format_exc()))


@pytest.mark.parametrize("N", [(4, 3, 4), (3, 3, 5)])
def test_two_rank_halo_sum_and_dot_match_serial(oracle, N):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + N[2]
    procs = [ctx.Process(target=_worker, args=(r, 2, port, N, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, err, derr, tb in res:
        assert tb is None, tb
        assert err < 1e-13 and derr < 1e-13, (rank, err, derr)
