#!/usr/bin/env python3
"""Dynamic instruction mix of the trace kernel: every mnemonic of every marked region (tools/phase_budget.py static: the listing of
the SHIPPED kernel attributed through inline stacks) weighed by the region's trip count from the counting build (phase_budget.py
dynamic).  It answers what the wave-level counters cannot: how much of a wave's instruction stream is not vector arithmetic at all --
exec-mask bookkeeping of divergent branches, s_nop hazards, s_waitcnt -- each of which costs the wave one issue slot like a VALU
instruction does.

The VALU total is checked against SQ_INSTS_VALU by phase_budget.py combine; the scalar classes are an UPPER bound (the reconvergence
code of every branch of a region is counted whether or not the branch was taken; SQ_INSTS_SALU, which leaves out s_nop, s_waitcnt
and branches, is about 20 % below the comparable sum).

usage: python tools/opcode_mix.py [c2|c5]        (needs build/phase_static_<cfg>.json and gpurun_out/phase_dynamic_<cfg>.json)
"""
import collections
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def klass(op):
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op in ("s_nop", "s_waitcnt", "s_sleep", "s_barrier"):
        return op
    if op.startswith("s_cbranch") or op in ("s_branch", "s_setpc_b64"):
        return "branch"
    if op.startswith("s_"):
        return "salu"
    return "vmem"


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
    st = json.load(open(os.path.join(ROOT, "build", "phase_static_%s.json" % cfg)))
    dy = json.load(open(os.path.join(ROOT, "gpurun_out", "phase_dynamic_%s.json" % cfg)))
    sites = dy.get("sites", dy)
    samples = dy.get("samples") or 1
    ops, per_site = collections.Counter(), []
    for site, mix in st["region_opcodes"].items():
        rec = sites.get(site)
        visits = rec.get("visits", 0) if isinstance(rec, dict) else 0
        copies = st["regions"][site].get("copies", 1)
        k = collections.Counter()
        for op, n in mix.items():
            ops[op] += n * visits / copies
            k[klass(op)] += n * visits / copies
        per_site.append((sum(k.values()), site, visits, k))
    total = sum(ops.values())
    by_class = collections.Counter()
    for op, n in ops.items():
        by_class[klass(op)] += n
    print("config %s, kernel %s" % (cfg, st["kernel"]))
    print("wave-instructions per launch %.4e = %.1f per sample" % (total, total / samples))
    print("classes: " + ", ".join("%s %.1f %%" % (c, 100 * n / total) for c, n in by_class.most_common()))
    print("\nregions by wave-instructions (share of all; of which valu / salu+branch / s_nop / s_waitcnt / lds):")
    for n, site, visits, k in sorted(per_site, reverse=True)[:24]:
        print("  %-18s %5.1f %%   valu %5.1f  scalar %5.1f  nop %4.1f  waitcnt %4.1f  lds %4.1f   (per visit; %.3e visits)"
              % (site, 100 * n / total, k["valu"] / max(visits, 1), (k["salu"] + k["branch"]) / max(visits, 1), k["s_nop"] / max(visits, 1),
                 k["s_waitcnt"] / max(visits, 1), k["lds"] / max(visits, 1), visits))
    print("\nmnemonics:")
    for op, n in ops.most_common(40):
        print("  %-28s %.3e  %5.2f %%" % (op, n, 100 * n / total))


if __name__ == "__main__":
    main()
