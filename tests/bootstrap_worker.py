"""One rank of the rendez-vous test (tests/test_host_logic.py): rank / size come from the environment a launcher would set, rank 0's 128-byte
payload must arrive on every rank.  No GPU involved (exa_bootstrap_env + exa_bootstrap_bcast only)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import exaconstit_amd.lib as L

r, n, l = C.c_int(-1), C.c_int(-1), C.c_int(-1)
assert L.exa_bootstrap_env(C.byref(r), C.byref(n), C.byref(l)) == 0
buf = (C.c_ubyte * 128)()
if r.value == 0:
    for i in range(128):
        buf[i] = (37 * i + 11) % 251
err = C.create_string_buffer(256)
rc = L.exa_bootstrap_bcast(r.value, n.value, buf, 128, 30.0, err, 256)
assert rc == 0, err.value
ok = all(buf[i] == (37 * i + 11) % 251 for i in range(128))
print("rank %d of %d local %d payload_ok %d" % (r.value, n.value, l.value, ok), flush=True)
sys.exit(0 if ok else 3)
