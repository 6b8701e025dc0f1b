"""Register / spill / scratch metadata of the k_model_setup instantiations (cross-compiled, no GPU needed): python scripts/kernel_meta.py [extra hipcc flags]
EXA_META_SRC=model_kernels_aos.hip lists the staged AOS instantiations (all of them; the default lists the element-blocked fused p = 1 ones)."""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "exaconstit_amd", "csrc", os.environ.get("EXA_META_SRC", "model_kernels.hip"))
show_all = "EXA_META_SRC" in os.environ
with tempfile.TemporaryDirectory() as d:
    s = os.path.join(d, "m.s")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-mllvm", "-amdgpu-use-amdgpu-trackers=1",
                           "-fno-signed-zeros", "-fno-trapping-math", "-fno-math-errno", "-S", "--cuda-device-only", "-o", s, src] + sys.argv[1:], stderr=subprocess.DEVNULL)
    t = open(s).read()
names = {"0": "Voce", "1": "VoceNL", "2": "KM-FCC", "3": "KM-BCC", "6": "KM-FCC p=q=1", "7": "KM-BCC p=q=1", "8": "Voce x^49", "9": "VoceNL x^49"}
for m in re.finditer(r'\.name:\s+(_Z13k_model_setupILi(\d+)ELb(\d)ELi(\d+)ELb(\d)ELb(\d)ELb(\d)EE\S*)\n((?:.*\n){1,14})', t):
    kin, lvec, nfix, qb, rec, stg, body = m.group(2), m.group(3), m.group(4), m.group(5), m.group(6), m.group(7), m.group(8)
    if not show_all and not (lvec == "1" and nfix == "8" and qb == "1"):
        continue
    g = lambda k: int(re.search(k + r':\s+(\d+)', body).group(1))
    ag = re.search(r'\.agpr_count:\s+(\d+)', body)
    print("%-13s lvec=%s n=%s qb=%s stg=%s rec=%s  vgpr %d  agpr %s  sgpr-spills %d  vgpr-spills %d  scratch %d B" % (names.get(kin, kin), lvec, nfix, qb, stg, rec, g(r'\.vgpr_count'), ag.group(1) if ag else "?", g(r'\.sgpr_spill_count'), g(r'\.vgpr_spill_count'), g(r'\.private_segment_fixed_size')))
