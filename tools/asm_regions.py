"""Walk one kernel of a -gline-tables-only device .s in program order and print runs of instructions per source file:line
range bucket (coarse regions), with VALU/f64/LDS/SALU counts.  usage: asm_regions.py file.s kernel"""
import re, sys
path, kern = sys.argv[1], sys.argv[2]
lines = open(path, errors="replace").read().split("\n")
files = {}
for l in lines:
    m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
    if m:
        files[int(m.group(1))] = (m.group(3) or m.group(2)).split("/")[-1]
start = next(i for i, l in enumerate(lines) if l.startswith(kern + ":"))
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
def region(f, ln):
    if f == "rt_kernels.h":
        for lo, hi, name in ((0, 170, "stage/unit"), (170, 230, "K.prologue"), (230, 330, "K.refill"), (330, 350, "K.scan-call"), (350, 420, "K.hit-glue"), (420, 460, "K.finish/shadow-state"), (460, 2000, "K.tail")):
            if lo <= ln < hi: return name
    if f == "rt_scan.h":
        for lo, hi, name in ((0, 230, "S.helpers"), (230, 300, "S.ops/post"), (300, 345, "S.util"), (345, 420, "S.filter"), (420, 480, "S.drainB"), (480, 545, "S.phaseA"), (545, 2000, "S.tree")):
            if lo <= ln < hi: return name
    if f == "rt_shade.h":
        for lo, hi, name in ((0, 60, "H.tex/mat"), (60, 135, "H.scatter"), (135, 175, "H.shade"), (175, 2000, "H.shadowq")):
            if lo <= ln < hi: return name
    if f == "rt_device_math.h":
        for lo, hi, name in ((0, 62, "M.vec/normalize"), (62, 112, "M.reflect/refract/fresnel"), (112, 145, "M.sincos"), (145, 202, "M.pow"), (202, 216, "M.halton"), (216, 260, "M.rng")):
            if lo <= ln < hi: return name
    if f == "rt_params.h": return "P.raygen"
    return f
cur = None
runs = []
for l in lines[start:end]:
    m = re.match(r'\s*\.loc\s+(\d+)\s+(\d+)', l)
    if m:
        cur = region(files.get(int(m.group(1)), "?"), int(m.group(2)))
        continue
    t = l.strip()
    if not t or t.startswith((";", ".")) or t.endswith(":"):
        if t.endswith(":") and t.startswith(".LBB"):
            runs.append(("LABEL " + t, {}))
        continue
    op = t.split()[0]
    if not re.match(r'^[a-z]', op): continue
    k = "f64" if ("_f64" in op) else "mfma" if op.startswith("v_mfma") else "valu" if op.startswith("v_") else "lds" if op.startswith("ds_") else "salu" if op.startswith("s_") else "vmem"
    if runs and runs[-1][0] == cur:
        runs[-1][1][k] = runs[-1][1].get(k, 0) + 1
    else:
        runs.append((cur, {k: 1}))
# merge tiny runs into summary by region
import collections
tot = collections.defaultdict(lambda: collections.Counter())
for name, c in runs:
    if name and not name.startswith("LABEL"):
        tot[name].update(c)
for name, c in sorted(tot.items()):
    print("%-28s %s  total %d" % (name, dict(c), sum(c.values())))
