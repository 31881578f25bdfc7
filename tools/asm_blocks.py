"""Basic blocks of one kernel of build/asm/rt_dev_g.o with >= N VALU instructions: VALU count, dominant RT_SITE region, leaf source lines.
usage: python tools/asm_blocks.py [c2|c5] [min_valu] [site-filter]   (after tools/phase_budget.py static)"""
import collections, re, sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import phase_budget as P
cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
minv = int(sys.argv[2]) if len(sys.argv) > 2 else 14
filt = sys.argv[3] if len(sys.argv) > 3 else None
obj = os.path.join(P.ROOT, "build", "asm", "rt_dev_g.o")
txt = P.sh([P.LLVM + "/llvm-objdump", "-d", "--no-show-raw-insn", "--disassemble-symbols=" + P.KERNELS[cfg], obj])
ins = []
for l in txt.split("\n"):
    m = re.match(r"^\s+([a-z][a-z0-9_]*)\b(.*?)//\s*([0-9A-Fa-f]+):\s*([0-9A-F ]+)(<.*\+0x([0-9a-f]+)>)?", l)
    if m:
        ins.append((int(m.group(3), 16), m.group(1), m.group(2).strip(), m.group(6)))
base = ins[0][0]
targets = {base + int(t, 16) for a, op, args, t in ins if (op.startswith("s_cbranch") or op == "s_branch") and t}
sym = P.sh([P.LLVM + "/llvm-symbolizer", "--obj=" + obj, "--inlining", "--functions=short", "--basenames"], input="\n".join("0x%x" % a for a, _, _, _ in ins) + "\n")
stacks = []
for grp in sym.strip().split("\n\n"):
    ls = grp.split("\n")
    fr = []
    for k in range(0, len(ls) - 1, 2):
        m = re.match(r"(.*):(\d+):(\d+)$", ls[k + 1])
        fr.append((ls[k][:22], m.group(1), int(m.group(2))))
    stacks.append(fr)
regions = P.site_regions()
blocks, cur = [], []
for i, (a, op, args, t) in enumerate(ins):
    if a in targets and cur:
        blocks.append(cur)
        cur = []
    cur.append(i)
    if op.startswith("s_cbranch") or op in ("s_branch", "s_endpgm"):
        blocks.append(cur)
        cur = []
if cur:
    blocks.append(cur)
rows = []
for b in blocks:
    nv = sum(1 for i in b if ins[i][1].startswith("v_"))
    c = collections.Counter()
    for i in b:
        for fn, f, line in stacks[i]:
            if line > 0:
                s = P.innermost_site(regions, f, line)
                if s:
                    c[s] += 1
                    break
    if nv < minv or (filt and not any(filt in s for s in c)):
        continue
    leafs = collections.Counter((stacks[i][0][0], stacks[i][0][1], stacks[i][0][2]) for i in b)
    rows.append((nv, len(b), "0x%x" % (ins[b[0]][0] - base), c.most_common(2), leafs.most_common(3)))
for r in sorted(rows, key=lambda r: -r[0]):
    print(r)
