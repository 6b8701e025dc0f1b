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
if len(sys.argv) > 1 and sys.argv[1] == "gather":
    # the general form: every rank contributes a 96-byte identity record (host name[64], PCI bus id[32]; here taken from the environment),
    # rank 0 decides the transport from the table (exa_transport_from_identities) and answers with the decision + the whole table
    mine = (C.c_ubyte * 96)()
    for i, b in enumerate(os.environ["FAKE_HOST"].encode()[:63]): mine[i] = b
    for i, b in enumerate(os.environ["FAKE_PCI"].encode()[:31]): mine[64 + i] = b
    nr = n.value
    reply = (C.c_ubyte * (4 + 96 * nr))()

    def decide(all_p, nranks, nbytes, reply_p, reply_bytes, user):
        e = C.create_string_buffer(256)
        kind = L.exa_transport_from_identities(all_p, nranks, e, 256)
        out = (C.c_ubyte * reply_bytes).from_address(reply_p)
        out[0:4] = list(int(kind).to_bytes(4, "little", signed=True))
        C.memmove(reply_p + 4, all_p, nranks * nbytes)
        return 0
    cb = L.exa_bootstrap_reply_fn(decide)
    rc = L.exa_bootstrap_gather_reply(r.value, nr, mine, 96, reply, 4 + 96 * nr, cb, None, 30.0, err, 256)
    assert rc == 0, err.value
    kind = int.from_bytes(bytes(reply[0:4]), "little", signed=True)
    table = bytes(reply[4:])
    hosts = [table[96 * i: 96 * i + 64].split(b"\0")[0].decode() for i in range(nr)]
    pcis = [table[96 * i + 64: 96 * i + 96].split(b"\0")[0].decode() for i in range(nr)]
    print("rank %d of %d kind %d hosts %s pcis %s" % (r.value, nr, kind, ",".join(hosts), ",".join(pcis)), flush=True)
    sys.exit(0)
rc = L.exa_bootstrap_bcast(r.value, n.value, buf, 128, 30.0, err, 256)
assert rc == 0, err.value
ok = all(buf[i] == (37 * i + 11) % 251 for i in range(128))
print("rank %d of %d local %d payload_ok %d" % (r.value, n.value, l.value, ok), flush=True)
sys.exit(0 if ok else 3)
