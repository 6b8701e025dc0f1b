"""Worker of tests/test_gpu_rccl.py: runs a regression case on the stand-alone driver with the RCCL transport and writes the averaged
stresses and solver history as JSON.  world = 1: EXA_FORCE_RCCL=1 makes the one-rank communicator run every RCCL call (comm init,
all-reduce, grouped send/recv with an empty neighbour list).  world > 1 (torchrun, one rank per GPU): the real multi-rank path.
Usage: python rccl_worker.py <case.toml> <nsteps> <out.json> [jacobi]"""
import ctypes as C
import json
import os
import sys

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    toml, nsteps, out = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    jacobi = len(sys.argv) > 4 and sys.argv[4] == "jacobi"
    import torch
    import exaconstit_amd.lib as L
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1")); local = int(os.environ.get("LOCAL_RANK", "0"))
    ndev = torch.cuda.device_count()
    shared = world > ndev                 # more ranks than devices: RCCL (and torch's nccl backend) refuse two ranks on one device
    torch.cuda.set_device(local % ndev)
    uid = None
    if world > 1:
        import torch.distributed as dist
        if shared:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        buf = (C.c_ubyte * 128)()
        if rank == 0:
            assert L.exa_comm_unique_id(buf, world) == 0      # a RCCL id, or the id of the shared-device inter-process transport
        t = torch.tensor(list(buf), dtype=torch.uint8, device="cpu" if shared else "cuda")
        dist.broadcast(t, 0)
        uid = (C.c_ubyte * 128)(*t.cpu().tolist())
    d = L.Driver.from_toml(toml, out_dir=os.path.dirname(out), rank=rank, nranks=world, uid=uid, jacobi=jacobi, write_files=False)
    ok = True
    for ti in range(1, nsteps + 1):
        ok = ok and d.step(ti)
    s = d.avgs(0, 6)
    newton, krylov, calls = d.stats()
    comm_ranks, transport = d.comm_info()
    if rank == 0:
        with open(out, "w") as f:
            json.dump(dict(ok=bool(ok), world=world, comm_ranks=comm_ranks, transport=transport, forced=bool(os.environ.get("EXA_FORCE_RCCL")), avg_stress=s.tolist(), comm=d.comm_details(),
                           newton=[int(x) for x in newton], krylov=[int(x) for x in krylov]), f)
    d.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
