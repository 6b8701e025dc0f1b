"""Copies the evidence set of scripts/profile_round6.sh from gpurun_out/ into profiles/ - and refuses any record that was not produced by the kernel build of the
library in this tree (record hygiene: bench lines and counter files of one round must carry one kernel_build_id).
    python scripts/collect_round6.py [--allow-other-build]"""
import glob, json, os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.chdir(ROOT)
# the id of the sources in the tree (what `make` stamps into the library): recomputed the way csrc/Makefile does, no GPU and no import needed
kid = subprocess.check_output(["make", "-s", "-C", "exaconstit_amd/csrc", "--eval", "printkid: ; @echo $(KERNEL_ID)", "printkid"]).decode().strip()
bad = []; n = 0
for f in sorted(glob.glob("gpurun_out/r06_*")):
    if os.path.isdir(f):
        continue
    rid = None
    try:
        if f.endswith(".json"):
            d = json.load(open(f)); rid = d.get("kernel_build_id") or d.get("library", {}).get("kernel_build_id") or d.get("roofline", {}).get("kernel_build_id")
        elif f.endswith(".txt"):
            head = open(f).readline().split(); rid = head[-1] if head[:1] == ["kernel_build_id"] or head[:2] == ["kernel", "build"] else None
    except Exception as e:
        bad.append((f, f"unreadable: {e}")); continue
    if rid is not None and rid != kid and "--allow-other-build" not in sys.argv:
        bad.append((f, f"kernel build {rid}, tree is {kid}")); continue
    if f.endswith((".log", ".err")) and not f.endswith("_gpu_tests.log"):
        continue
    shutil.copy(f, os.path.join("profiles", os.path.basename(f))); n += 1
print(f"tree kernel build id {kid}: copied {n} files into profiles/")
for f, why in bad:
    print("REFUSED", f, "-", why)
sys.exit(1 if bad else 0)
