"""Test plumbing for the GPU parity tests: synthetic RVE inputs (numpy) and thin wrappers that move them through the
C ABI of libexaconstit_hip.so using torch tensors as device buffers.  The oracle (tests/orc.py) is the checker."""
import ctypes as C

import numpy as np


def rel_l2(a, b):
    a = np.asarray(a, dtype=np.float64).ravel()
    b = np.asarray(b, dtype=np.float64).ravel()
    d = np.linalg.norm(b)
    return np.linalg.norm(a - b) / (d if d > 0 else 1.0)


def random_quats(n, seed=20240928):
    rng = np.random.default_rng(seed)
    q = rng.standard_normal((n, 4))
    return q / np.linalg.norm(q, axis=1, keepdims=True)


def make_rve(orc, N, p=1, distort=0.0, seed=1):
    """Cartesian N^3 mesh of the unit cube through the oracle's mesh builder (x fastest), optionally distorted."""
    n = (p + 1) ** 3
    E = N ** 3
    NN = (N * p + 1) ** 3
    conn = np.zeros(n * E, dtype=np.int32)
    X = np.zeros(NN * 3)
    orc.lib().orc_mesh(p, N, N, N, C.c_double(1.0), C.c_double(1.0), C.c_double(1.0), orc._ip(conn), orc._p(X))
    if distort > 0:
        rng = np.random.default_rng(seed)
        X = X + distort / (N * p) * rng.uniform(-1, 1, X.shape)
    G = np.zeros(n * 3 * n)
    W = np.zeros(n)
    orc.lib().orc_ref_elem(p, orc._p(G), orc._p(W))
    return dict(N=N, p=p, n=n, Q=n, E=E, NN=NN, conn=conn, X=X, G=G, W=W)


def l_to_e(rve, L):
    n, E, NN = rve["n"], rve["E"], rve["NN"]
    conn = rve["conn"].reshape(E, n)
    out = np.zeros((E, 3, n))
    for c in range(3):
        out[:, c, :] = L[conn + NN * c]
    return out.ravel()


def e_to_l(rve, Ev):
    """transpose of l_to_e: sums the element contributions (E, 3, n) into an L-vector (byNODES)"""
    n, E, NN = rve["n"], rve["E"], rve["NN"]
    conn = rve["conn"].reshape(E, n)
    out = np.zeros(3 * NN)
    for c in range(3):
        np.add.at(out, conn + NN * c, np.asarray(Ev).reshape(E, 3, n)[:, c, :])
    return out


def velocity_field(rve, scale=1.0, seed=7):
    """Nodal velocity of a perturbed uniaxial tension: v = L0 x + noise (SURVEY 8(d) kernel micro-benchmark shape)."""
    rng = np.random.default_rng(seed)
    NN = rve["NN"]
    X = rve["X"].reshape(3, NN)
    L0 = np.diag([-0.45e-3, -0.45e-3, 1.0e-3]) + 1e-4 * np.array([[0, 0.3, -0.2], [-0.3, 0, 0.1], [0.2, -0.1, 0]])
    v = (L0 @ X) * scale
    v += 1.0e-4 * scale / rve["N"] * rng.uniform(-1, 1, v.shape)
    return v.ravel()


class Dev:
    """torch-backed device buffers (allocator only)."""

    def __init__(self):
        import torch
        self.torch = torch
        if not torch.cuda.is_available():
            raise RuntimeError("GPU parity tests need a HIP device")
        self.dev = torch.device("cuda:0")

    def up(self, a, dtype=None):
        t = self.torch.from_numpy(np.ascontiguousarray(a))
        if dtype is not None:
            t = t.to(dtype)
        return t.to(self.dev)

    def zeros(self, n, dtype=None):
        return self.torch.zeros(int(n), dtype=dtype or self.torch.float64, device=self.dev)

    def sync(self):
        self.torch.cuda.synchronize()


def ptr(t):
    return C.c_void_p(t.data_ptr())
