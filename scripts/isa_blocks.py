"""Basic-block statistics of one kernel in an AMDGPU assembly listing (hipcc -S --cuda-device-only): per block the loop annotation LLVM
prints, instruction count, VALU / FP64 / scratch / LDS / global / lane-spill counts.  usage: isa_blocks.py file.s mangled-name-prefix"""
import collections, re, sys
src, pref = sys.argv[1], sys.argv[2]
lines = open(src).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith(pref) and l.rstrip().endswith(":") or (l.startswith(pref) and ": ;" in l))
blocks = []; cur = None
for l in lines[start + 1:]:
    if ".end_amdhsa_kernel" in l or l.startswith("\t.section"):
        break
    m = re.match(r'^(\.LBB\d+_\d+):\s*(;.*)?$', l) or re.match(r'^; %bb\.(\d+):\s*(;.*)?$', l)
    if m:
        cur = dict(name=m.group(1), note=(m.group(2) or ""), ops=collections.Counter()); blocks.append(cur); continue
    if cur is None:
        cur = dict(name="entry", note="", ops=collections.Counter()); blocks.append(cur)
    s = l.strip()
    if not s or s.startswith(";") or s.startswith("."):
        if s.startswith(";") and "Loop" in s and not cur["ops"]:
            cur["note"] += " " + s
        continue
    cur["ops"][s.split()[0]] += 1
def cls(ops):
    n = sum(ops.values()); valu = sum(c for o, c in ops.items() if o.startswith("v_")); f64 = sum(c for o, c in ops.items() if "f64" in o)
    lane = sum(c for o, c in ops.items() if o.startswith(("v_readlane", "v_writelane"))); scr = sum(c for o, c in ops.items() if o.startswith("scratch_"))
    mov = sum(c for o, c in ops.items() if o.startswith(("v_mov", "v_accvgpr"))); sel = sum(c for o, c in ops.items() if o.startswith("v_cndmask"))
    lds = sum(c for o, c in ops.items() if o.startswith("ds_")); glb = sum(c for o, c in ops.items() if o.startswith("global_"))
    return n, valu, f64, mov, sel, lane, scr, lds, glb
print("%-12s %6s %6s %6s %5s %5s %5s %5s %4s %4s  note" % ("block", "n", "valu", "f64", "mov", "sel", "lane", "scr", "lds", "glb"))
for b in blocks:
    c = cls(b["ops"])
    if c[0] >= int(sys.argv[3]) if len(sys.argv) > 3 else 15:
        print("%-12s %6d %6d %6d %5d %5d %5d %5d %4d %4d  %s" % ((b["name"],) + c + (b["note"].strip()[:70],)))
