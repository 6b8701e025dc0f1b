"""The `mechanics -opt options.toml` executable (reference src/mechanics_driver.cpp:112-1022; launched as `mpirun -np N mechanics -opt x.toml`
by the reference's own regression script, test/test_mechanics.py:38) - C++ only, no Python in the process: bootstrap from the launcher's
environment, run of a whole regression case, the reference's output files."""
import os
import shutil
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "exaconstit_amd", "mechanics")
REFDATA = os.path.join(ROOT, "tests", "golden", "refdata")


def _stage(tmp_path, name, nsteps=None):
    """copy the reference's option / property / grain files; nsteps: run only the first steps of the case's custom schedule"""
    for f in os.listdir(REFDATA):
        if f.endswith((".txt", ".ori", ".toml", ".mesh")) and not f.endswith("_stress.txt"):
            shutil.copy(os.path.join(REFDATA, f), str(tmp_path))
    toml = os.path.join(str(tmp_path), name + ".toml")
    if nsteps is not None:
        t = open(toml).read()
        assert "nsteps = 40" in t
        open(toml, "w").write(t.replace("nsteps = 40", "nsteps = %d" % nsteps, 1))
    return toml


def _mpirun():
    for c in ("mpirun", "/opt/conda/bin/mpirun", "mpiexec"):
        p = shutil.which(c) or (c if os.path.exists(c) else None)
        if p:
            return p
    return None


def _check_against_golden(tmp_path, name, rows=None):
    g = np.loadtxt(os.path.join(REFDATA, name + "_stress.txt"))[:rows]
    s = np.loadtxt(os.path.join(str(tmp_path), "test_" + name + "_stress.txt"))
    assert s.shape == g.shape
    unit = 10.0 ** (np.floor(np.log10(np.abs(g[:, 2]))) - 5)
    assert np.max(np.abs(s[:, 2] - g[:, 2]) / unit) <= 1.0 + 1e-6      # the reference's acceptance: the printed text


def test_executable_plain(tmp_path):
    assert os.path.exists(EXE), "build() links exaconstit_amd/mechanics"
    toml = _stage(tmp_path, "voce_pa")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "PMI_RANK", "PMI_SIZE")}
    r = subprocess.run([EXE, "-opt", toml], cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr
    assert "The process took" in r.stdout and "ranks 1" in r.stdout
    _check_against_golden(tmp_path, "voce_pa")
    t = np.loadtxt(os.path.join(str(tmp_path), "time", "time_solve.0.txt"))      # reference: ./time/time_solve.<rank>.txt, one wall time per step
    assert t.shape == (40,) and np.all(t > 0)


def test_executable_under_mpirun_one_rank(tmp_path):
    mpirun = _mpirun()
    if mpirun is None:
        pytest.skip("no mpirun in the image")
    toml = _stage(tmp_path, "mtsdd_bcc")
    r = subprocess.run([mpirun, "-np", "1", EXE, "-opt", toml], cwd=str(tmp_path), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout, r.stderr)
    _check_against_golden(tmp_path, "mtsdd_bcc")


@pytest.mark.parametrize("np_", [2, 8])
def test_executable_ranks(tmp_path, np_):
    """The reference's regression command line, `mpirun -np N mechanics -opt voce_pa.toml` (test/test_mechanics.py:38), N = 2 and N = 8 (the 10^3
    mesh on 2 x 2 x 2 blocks).  One GPU per rank: RCCL; a one-GPU box: the ranks hand the identity of their device to rank 0 in the rendez-vous,
    which finds them on one device and answers with the id of the shared-device inter-process transport instead of a RCCL id (stderr says so)."""
    mpirun = _mpirun()
    nsteps = 40 if np_ == 2 else 4      # (eight processes share one GPU through the host-synchronous transport: the first 4 of the 40 steps there)
    toml = _stage(tmp_path, "voce_pa", nsteps=None if nsteps == 40 else nsteps)
    env = dict(os.environ, EXA_MASTER_PORT=str(29533 + np_))
    if mpirun:
        cmd = [mpirun, "-np", str(np_), EXE, "-opt", toml]
        r = subprocess.run(cmd, cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=1500)
        assert r.returncode == 0, (r.stdout, r.stderr)
        assert "exa_bootstrap: %d ranks, transport" % np_ in r.stderr and ("ranks %d" % np_) in r.stdout
    else:
        ps = [subprocess.Popen([EXE, "-opt", toml], cwd=str(tmp_path), env=dict(env, EXA_RANK=str(k), EXA_NRANKS=str(np_))) for k in range(np_)]
        assert all(p.wait(timeout=1500) == 0 for p in ps)
    _check_against_golden(tmp_path, "voce_pa", rows=nsteps)
    for k in range(np_):
        assert os.path.exists(os.path.join(str(tmp_path), "time", "time_solve.%d.txt" % k))
