"""Program-order walk of one kernel in a -gline-tables-only device .s: runs of instructions per source line with counts.
usage: asm_walk.py file.s kernel [first_line last_line]"""
import re, sys
path, kern = sys.argv[1], sys.argv[2]
lines = open(path, errors="replace").read().split("\n")
files = {}
for l in lines:
    m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
    if m: files[int(m.group(1))] = (m.group(3) or m.group(2)).split("/")[-1].replace("rt_", "").replace(".h", "")
start = next(i for i, l in enumerate(lines) if l.startswith(kern + ":"))
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
cur, runs = "?", []
for i in range(start, end):
    l = lines[i]
    m = re.match(r'\s*\.loc\s+(\d+)\s+(\d+)', l)
    if m:
        cur = "%s:%s" % (files.get(int(m.group(1)), "?"), m.group(2)); continue
    t = l.strip()
    if t.endswith(":") and t.startswith(".LBB"):
        runs.append(["== " + t, 0, 0, 0]); continue
    if not t or t.startswith((";", ".")) or t.endswith(":"): continue
    op = t.split()[0]
    if not re.match(r'^[a-z]', op): continue
    isv = op.startswith("v_"); isl = op.startswith("ds_")
    if "s_cbranch" in op or "s_branch" in op:
        runs.append(["   -> " + t, 0, 0, 0]); continue
    if runs and runs[-1][0] == cur: runs[-1][1] += 1; runs[-1][2] += isv; runs[-1][3] += isl
    else: runs.append([cur, 1, int(isv), int(isl)])
for r in runs:
    if r[0].startswith(("==", "   ->")): print(r[0])
    else: print("   %-22s n=%-3d valu=%-3d lds=%d" % (r[0], r[1], r[2], r[3]))
