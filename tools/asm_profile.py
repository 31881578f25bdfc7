"""Static instruction histogram of one kernel by source line (device .s compiled with -gline-tables-only).
usage: python tools/asm_profile.py build/asm2/rt_capi_dev.s <mangled kernel name> [top]
Counts are static (not execution-weighted): they show where the code size of the loop body sits."""
import re, sys, collections
path, kern = sys.argv[1], sys.argv[2]
top = int(sys.argv[3]) if len(sys.argv) > 3 else 60
files = {}
lines = open(path, errors="replace").read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith(kern + ":"))
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
for l in lines:
    m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
    if m:
        files[int(m.group(1))] = (m.group(3) or m.group(2)).split("/")[-1]
cur = ("?", 0)
hist = collections.Counter()
kinds = collections.Counter()
bykind_line = collections.defaultdict(collections.Counter)
for l in lines[start:end]:
    m = re.match(r'\s*\.loc\s+(\d+)\s+(\d+)', l)
    if m:
        cur = (files.get(int(m.group(1)), m.group(1)), int(m.group(2)))
        continue
    t = l.strip()
    if not t or t.startswith((";", ".", "_")) or t.endswith(":"):
        continue
    op = t.split()[0]
    if not re.match(r'^[a-z]', op):
        continue
    kind = ("mfma" if op.startswith("v_mfma") else "valu" if op.startswith("v_") else "salu" if op.startswith("s_") and not op.startswith(("s_waitcnt", "s_nop", "s_cbranch", "s_branch"))
            else "lds" if op.startswith("ds_") else "vmem" if op.startswith(("global_", "buffer_", "flat_", "scratch_")) else "ctl")
    hist[cur] += 1
    kinds[kind] += 1
    bykind_line[cur][kind] += 1
tot = sum(hist.values())
print("kernel %s: %d instructions  %s" % (kern[:60], tot, dict(kinds)))
byfile = collections.Counter()
for (f, ln), c in hist.items():
    byfile[f] += c
print("by file:", dict(byfile))
for (f, ln), c in hist.most_common(top):
    print("%5d  %5.1f%%  %s:%d  %s" % (c, 100.0 * c / tot, f, ln, dict(bykind_line[(f, ln)])))
