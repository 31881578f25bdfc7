#!/usr/bin/env python3
"""Per-phase instruction budget of the trace megakernel.

Two measurements, multiplied:
  static   (here, no GPU)  every instruction of the SHIPPED kernel (device code object rebuilt with -g: same code, checked
                           against the -g-less build opcode by opcode) is attributed, through its inline stack
                           (llvm-symbolizer --inlining), to the innermost RT_SITE-marked source region its frames lie in;
  dynamic  (GPU box)       librt_hip_phases.so (-DRT_PHASES) counts, per marked region, wave visits and active lanes over a
                           whole launch of the workload (the counts are algorithmic: the shipped build takes the same trips).
budget = sum over regions of static VALU instructions x visits (wave-instructions) and x active lanes (lane-operations),
grouped into phases, and compared with SQ_INSTS_VALU / SQ_THREAD_CYCLES_VALU of the committed PMC pass of the same workload.

usage:
  python tools/phase_budget.py static  [c2|c5] -> build/phase_static_<cfg>.json        (needs hipcc; ~2 min)
  python tools/phase_budget.py dynamic [c2|c5] -> gpurun_out/phase_dynamic_<cfg>.json  (GPU box; librt_hip_phases.so)
  python tools/phase_budget.py combine [c2|c5] [pmc_summary.json] -> profiles/r04_phase_budget[_c5].json
"""
import collections
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "cpuraytracer_amd", "csrc")
LLVM = "/opt/rocm/lib/llvm/bin"
KERNELS = {  # the instantiation rt_render launches for the config (RT_VERBOSE prints the variant)
    "c2": "_ZN3rtd15rt_trace_kernelILb1ELi1024ELi1ELb1ELb1ELb0ELb1ELb1ELb0ELb0EEEvNS_11TraceParamsE",
    "c5": "_ZN3rtd15rt_trace_kernelILb0ELi1024ELi3ELb1ELb1ELb0ELb1ELb0ELb0ELb0EEEvNS_11TraceParamsE",
}
WORKLOADS = {"c2": ("cover", 1200, 800, 128, -1.0), "c5": ("grid10k", 4096, 4096, 64, -1.0)}
SITE_FILES = ("rt_kernels.h", "rt_scan.h", "rt_shade.h", "rt_params.h", "rt_device_math.h")

# region -> phase of the budget (VERDICT r3 #1a's list)
PHASES = collections.OrderedDict([
    ("ray generation", ["K_GEN", "K_GEN_LANE", "K_PATHLIST", "K_PARTIAL", "P_HALTON"]),
    ("slow paths of the divisions and square roots (range guards failed)", ["M_DIV_SLOW", "M_SQRT_SLOW", "M_ROOT_SLOW"]),
    ("queue (block claims)", ["K_NEXTBLOCK", "K_CLAIM"]),
    ("stash push / pop", ["K_POP", "K_POP_LANE", "K_PUSH", "K_PUSH_LANE"]),
    ("scan: operand build, bitmap exchange, result read", ["S_SCAN"]),
    ("scan: MFMA + mfma_post", ["S_TILEPAIR"]),
    ("scan: one-sphere groups straight to the exact list", ["S_SINGLE", "S_SINGLE_PUSH0", "S_SINGLE_PUSH1"]),
    ("scan: pooled (ray, group) list build", ["S_PASS", "S_TAKE", "S_TAKE_PUSH0", "S_TAKE_PUSH1", "S_PUSH_WORD"]),
    ("scan: pooled phase A (ray fetch + four one-sphere bound tests + survivor push)", ["S_ASTEP", "S_ASTEP2", "S_APUSH"]),
    ("scan: exact Sphere::Intersect (phase B)", ["S_DRAIN", "S_BSTEP", "S_BSTEP2"]),
    ("scan: ds_min_u64 merge", ["S_BMIN"]),
    ("grid scan: ray clip, big spheres, result read", ["G_SCAN", "G_BIG", "G_BIGPUSH"]),
    ("grid scan: feed (slab items listed)", ["G_FEED", "G_FEED_LANE"]),
    ("grid scan: item round (ray fetch, slab rows, cell starts)", ["G_ROUND"]),
    ("grid scan: one-sphere bound tests, four per step + survivor push", ["G_STEP", "G_PUSH"]),
    ("grid scan: exact Sphere::Intersect", ["G_DRAIN", "G_BSTEP"]),
    ("grid scan: ds_min_u64 merge", ["G_BMIN"]),
    ("transitions (miss / hit record / far-hit shadow state)", ["K_TRANS_MISS", "K_TRANS_HIT", "K_TRANS_SHADOW"]),
    ("hit processing: glue (material load, normal, state update)", ["K_PROCESS", "H_PROCESS", "H_MAT16", "H_INDEXED", "H_FARHIT", "H_MULTI"]),
    ("hit processing: scatter_only", ["H_SCATTER", "H_TRANSPARENT", "H_METAL", "H_OPAQUE", "H_OPAQUE_DIFFUSE"]),
    ("hit processing: shadow_query", ["H_SHADOWQ", "H_SQ_WALK", "H_SQ_CONSIDER", "H_SQ_GROUND", "H_SQ_GTAIL", "H_SQ_ROUND", "H_SQ_TAIL1", "H_SQ_CELL", "H_SQ_ROOTS", "H_SQ_FULL"]),
    ("hit processing: shade_value", ["H_SHADEV", "H_SHADE"]),
    ("sample store (finishPath)", ["K_FINISH"]),
    ("loop control, refill ballots, exit test", ["K_ITER"]),
    ("prologue / epilogue (staging, counters)", ["K_WAVE", "OTHER"]),
])


def sh(cmd, **kw):
    return subprocess.run(cmd, check=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True, **kw).stdout


def hipflags():
    mk = open(os.path.join(CSRC, "Makefile")).read()
    m = re.search(r"^HIPFLAGS = (.*?)(?<!\\)\n", mk, re.S | re.M)
    return m.group(1).replace("\\\n", " ").replace("$(ARCH)", "gfx950").split()


def strip_code(text):
    """Source text with comments and string/char literals blanked (same length, same line breaks): for brace matching."""
    out = list(text)
    i, n = 0, len(text)
    while i < n:
        c = text[i]
        if text.startswith("//", i):
            j = text.find("\n", i)
            j = n if j < 0 else j
            for k in range(i, j):
                out[k] = " "
            i = j
        elif text.startswith("/*", i):
            j = text.find("*/", i + 2)
            j = n if j < 0 else j + 2
            for k in range(i, j):
                if out[k] != "\n":
                    out[k] = " "
            i = j
        elif c in "\"'":
            j = i + 1
            while j < n and text[j] != c:
                j += 2 if text[j] == "\\" else 1
            for k in range(i + 1, min(j, n)):
                out[k] = " "
            i = j + 1
        else:
            i += 1
    return "".join(out)


def site_regions():
    """{file: [(first_line, last_line, site)]}: the innermost { } block around every RT_SITE(NAME) marker (1-based lines)."""
    regions = collections.defaultdict(list)
    for fn in SITE_FILES:
        text = open(os.path.join(CSRC, fn)).read()
        code = strip_code(text)
        line_of = [1] * (len(code) + 1)
        ln = 1
        for k, ch in enumerate(code):
            line_of[k] = ln
            if ch == "\n":
                ln += 1
        for m in re.finditer(r"\bRT_SITE\((\w+)\)", code):
            if code.rfind("#define", 0, m.start()) > code.rfind("\n", 0, m.start()):
                continue  # the macro's own definition
            depth, k = 0, m.start()
            while k >= 0:  # backwards to the unmatched '{'
                if code[k] == "}":
                    depth += 1
                elif code[k] == "{":
                    if depth == 0:
                        break
                    depth -= 1
                k -= 1
            assert k >= 0, (fn, m.group(1))
            depth, e = 0, k
            while e < len(code):  # forwards to its '}'
                if code[e] == "{":
                    depth += 1
                elif code[e] == "}":
                    depth -= 1
                    if depth == 0:
                        break
                e += 1
            regions[fn].append((line_of[k], line_of[e], m.group(1)))
    return regions


def innermost_site(regions, fn, line, in_lambda=False):
    """The innermost marked region around (file, line).  in_lambda: the frame is a lambda's operator(); a region whose { } block lies
    AROUND the lambda's body does not count then -- the lambda runs where it is CALLED (testItem, exactPair, pushWord are defined once
    and called from differently marked places), so the caller's frame decides; a region inside the lambda's body does count."""
    best = None
    for lo, hi, name in regions.get(fn, ()):
        if lo <= line <= hi and (best is None or hi - lo < best[0]):
            best = (hi - lo, name, lo, hi)
    if best and in_lambda:
        for llo, lhi in lambda_bodies(fn):
            if llo <= line <= lhi and best[2] <= llo and lhi <= best[3] and (best[2], best[3]) != (llo, lhi):
                return None
    return best[1] if best else None


_LAMBDAS = {}


def lambda_bodies(fn):
    """[(first_line, last_line)] of the bodies of the lambdas defined in a source file (`[&](...) ... {` to its `}`)."""
    if fn not in _LAMBDAS:
        out = []
        path = os.path.join(CSRC, fn)
        if os.path.exists(path):
            code = strip_code(open(path).read())
            for m in re.finditer(r"\[&\]\s*\(", code):
                k = code.find("{", m.end())
                # the parameter list and attributes come first: the body's brace is the first '{' after the matching ')'
                depth, j = 1, m.end()
                while j < len(code) and depth:
                    depth += code[j] == "("
                    depth -= code[j] == ")"
                    j += 1
                k = code.find("{", j)
                if k < 0:
                    continue
                depth, e = 0, k
                while e < len(code):
                    depth += code[e] == "{"
                    depth -= code[e] == "}"
                    if depth == 0:
                        break
                    e += 1
                out.append((code.count("\n", 0, k) + 1, code.count("\n", 0, e) + 1))
        _LAMBDAS[fn] = out
    return _LAMBDAS[fn]


def classify(op):
    if op.startswith("v_mfma") or op.startswith("v_smfma"):
        return "mfma"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith("s_"):
        return "salu"
    if op.startswith(("global_", "flat_", "buffer_", "scratch_")):
        return "vmem"
    return "other"


def disassemble(obj, kernel, with_loop=False):
    txt = sh([LLVM + "/llvm-objdump", "-d", "--no-show-raw-insn", "--disassemble-symbols=" + kernel, obj])
    ins, branches = [], []
    for l in txt.split("\n"):
        m = re.match(r"^\s+([a-z][a-z0-9_]*)\b(.*?)//\s*([0-9A-Fa-f]+):\s*[0-9A-F ]+(?:<.*\+0x([0-9a-f]+)>)?", l)
        if m:
            ins.append((int(m.group(3), 16), m.group(1), m.group(2).strip()))
            if m.group(4) and (m.group(1).startswith("s_cbranch") or m.group(1) == "s_branch"):
                branches.append((int(m.group(3), 16), int(m.group(4), 16)))
    if not with_loop:
        return ins
    # basic blocks: cut at branch targets and behind branches
    base = ins[0][0]
    targets = {base + t for _a, t in branches}
    blocks, cur = [], []
    for k, (a, op, _args) in enumerate(ins):
        if a in targets and cur:
            blocks.append(cur)
            cur = []
        cur.append(k)
        if op.startswith("s_cbranch") or op in ("s_branch", "s_endpgm", "s_setpc_b64"):
            blocks.append(cur)
            cur = []
    if cur:
        blocks.append(cur)
    return ins, blocks


def do_static(cfg):
    kernel = KERNELS[cfg]
    out_dir = os.path.join(ROOT, "build", "asm")
    os.makedirs(out_dir, exist_ok=True)
    flags = hipflags()
    src = os.path.join(CSRC, "rt_capi.hip")
    objs = {"g": os.path.join(out_dir, "rt_dev_g.o"), "plain": os.path.join(out_dir, "rt_dev.o")}
    newest = max(os.path.getmtime(os.path.join(CSRC, f)) for f in os.listdir(CSRC) if f.endswith((".h", ".hip", ".inc")))
    procs = []
    for kind, obj in objs.items():
        if not os.path.exists(obj) or os.path.getmtime(obj) < newest:
            cmd = ["/opt/rocm/bin/hipcc"] + flags + (["-g"] if kind == "g" else []) + ["--cuda-device-only", "--no-gpu-bundle-output", "-c", "-o", obj, src]
            procs.append(subprocess.Popen(cmd, stderr=subprocess.DEVNULL))
    for p in procs:
        assert p.wait() == 0
    ins, blocks = disassemble(objs["g"], kernel, with_loop=True)
    plain = disassemble(objs["plain"], kernel)
    # -g is not perfectly codegen-neutral in this compiler (register allocation and a handful of instructions move: 7,110 vs
    # 7,106 on the headline kernel); the attribution is valid while the two builds have the same instruction mix to 0.5 %
    hg, hp = collections.Counter(classify(o) for _, o, _ in ins), collections.Counter(classify(o) for _, o, _ in plain)
    for c in set(hg) | set(hp):
        assert abs(hg[c] - hp[c]) <= max(2, 0.005 * hp[c]), ("-g changed the instruction mix", c, hg[c], hp[c])
    same_code = [(o, a) for _, o, a in ins] == [(o, a) for _, o, a in plain]
    sym = sh([LLVM + "/llvm-symbolizer", "--obj=" + objs["g"], "--inlining", "--functions=short", "--basenames"],
             input="\n".join("0x%x" % a for a, _, _ in ins) + "\n")
    stacks = []
    for grp in sym.strip().split("\n\n"):
        ls = grp.split("\n")
        fr = []
        for k in range(0, len(ls) - 1, 2):
            m = re.match(r"(.*):(\d+):(\d+)$", ls[k + 1])
            fr.append((ls[k], m.group(1), int(m.group(2))) if m else (ls[k], "?", 0))
        stacks.append(fr)
    assert len(stacks) == len(ins), (len(stacks), len(ins))
    regions = site_regions()
    sites, paths = [], []
    for fr in stacks:
        site, path = None, ()
        for k, (_fn, f, line) in enumerate(fr):  # leaf first
            if line > 0:
                site = innermost_site(regions, f, line, in_lambda=_fn.startswith("operator()"))
                if site:
                    path = tuple((ff, ll) for _n, ff, ll in fr[k + 1:])  # the call sites above the matched frame
                    break
        sites.append(site)
        paths.append(path if site else None)
    # instructions without a usable line (line 0, compiler generated) take their nearest attributed neighbour's region
    known = [k for k, s in enumerate(sites) if s is not None or any(l > 0 for _, _, l in stacks[k])]
    for k, s in enumerate(sites):
        if s is None and not any(l > 0 for _, _, l in stacks[k]):
            near = min(known, key=lambda j: abs(j - k)) if known else None
            sites[k] = sites[near] if near is not None else None
    # Loop-invariant code the compiler hoisted in front of the persistent loop keeps the source lines of the loop body, but it runs
    # once per wave: a basic block most of whose instructions belong to the once-per-wave code (K_WAVE, or no region at all) is
    # once-per-wave code as a whole, whatever the lines of the rest say.
    hoisted = 0
    for b in blocks:
        once = sum(1 for k in b if sites[k] in (None, "K_WAVE"))
        if 2 * once >= len(b):
            for k in b:
                if sites[k] not in (None, "K_WAVE"):
                    sites[k] = "K_WAVE"
                    paths[k] = None
                    hoisted += 1
    # A region inside a lambda or function that is inlined at several call sites exists in several COPIES, of which a visit runs
    # one (drainB is called from the phase-A loop and after it; finishPath from two places; consider() from six): the region's
    # counters add the visits of all copies, so the static count per visit is the total over the number of copies = distinct call
    # paths above the matched frame.
    copies = collections.defaultdict(set)
    for site, path in zip(sites, paths):
        if site and path is not None:  # (instructions that took a neighbour's region have no stack of their own)
            copies[site].add(path)
    per = collections.defaultdict(collections.Counter)
    per_ops = collections.defaultdict(collections.Counter)  # mnemonics per region (tools/opcode_mix.py weighs them by the trip counts)
    for (addr, op, _a), site in zip(ins, sites):
        c = classify(op)
        per[site or "OTHER"][c] += 1
        per_ops[site or "OTHER"][op] += 1
        if c == "valu" and "_f64" in op:
            per[site or "OTHER"]["valu_f64"] += 1
        if op.startswith("ds_bpermute"):
            per[site or "OTHER"]["ds_bpermute"] += 1
    out = {"config": cfg, "kernel": kernel, "instructions": len(ins), "instructions_shipped_build": len(plain), "debug_build_code_identical": same_code, "instructions_outside_the_persistent_loop_with_loop_lines": hoisted,
           "instruction_mix_debug_build": dict(hg), "instruction_mix_shipped_build": dict(hp), "regions": {k: dict(v, copies=max(1, len(copies.get(k, ())))) for k, v in sorted(per.items())},
           "region_opcodes": {k: dict(v) for k, v in sorted(per_ops.items())},
           "totals": dict(sum((collections.Counter(v) for v in per.values()), collections.Counter())),
           "site_lines": {fn: [(lo, hi, n) for lo, hi, n in r] for fn, r in regions.items()}}
    path = os.path.join(ROOT, "build", "phase_static_%s.json" % cfg)
    json.dump(out, open(path, "w"), indent=1, sort_keys=True)
    with open(os.path.join(ROOT, "build", "phase_listing_%s.txt" % cfg), "w") as f:  # the kernel, instruction by instruction, with its region
        for (addr, op, args), site, fr in zip(ins, sites, stacks):
            f.write("%6x  %-14s %-28s %-60s %s\n" % (addr, site or "-", op, args, " < ".join("%s:%d" % (n[:24], l) for n, _f, l in fr[:3])))
    print(path)
    for k, v in sorted(per.items()):
        print("  %-18s x%d %s" % (k, max(1, len(copies.get(k, ()))), dict(v)))
    print("  totals", out["totals"])


def do_dynamic(cfg):
    import ctypes as C
    sys.path.insert(0, ROOT)
    from cpuraytracer_amd import _capi
    _capi.LIB_PATH = os.path.join(ROOT, "cpuraytracer_amd", "lib", "librt_hip_phases.so")
    from cpuraytracer_amd import HipRenderer, scenes
    scene, W, H, spp, ap = WORKLOADS[cfg]
    r = HipRenderer(0)
    r.upload(scenes.build_scene(scene, 1, W, H, aperture=ap))
    L = _capi.load()
    L.rt_debug_sites.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong), C.c_uint32, C.POINTER(C.c_char_p)]
    names = C.c_char_p()
    buf = (C.c_ulonglong * 512)()
    r.render(W, H, 1, 2, 50, 1)  # warm-up (tile order, first launch)
    assert L.rt_debug_sites(r._h, buf, 512, C.byref(names)) == 0  # read and clear
    st = r.render(W, H, 1, 1 + spp, 50, 1)
    assert L.rt_debug_sites(r._h, buf, 512, C.byref(names)) == 0
    nm = [n for n in names.value.decode().split(",") if n]
    out = {"config": cfg, "workload": {"scene": scene, "W": W, "H": H, "spp": spp}, "samples": int(st.samples), "traversals": int(st.traversals),
           "segments": int(st.segments), "ms_render_counting_build": st.ms_render,
           "sites": {n: {"visits": int(buf[2 * k]), "lanes": int(buf[2 * k + 1])} for k, n in enumerate(nm)}}
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    path = os.path.join(ROOT, "gpurun_out", "phase_dynamic_%s.json" % cfg)
    json.dump(out, open(path, "w"), indent=1, sort_keys=True)
    print(path, "samples", st.samples, "K_ITER", out["sites"].get("K_ITER"))


def do_combine(cfg, pmc_path=None):
    st = json.load(open(os.path.join(ROOT, "build", "phase_static_%s.json" % cfg)))
    dy = json.load(open(os.path.join(ROOT, "gpurun_out", "phase_dynamic_%s.json" % cfg)))
    S = dy["sites"]
    samples = dy["samples"]
    waves = None
    rows = []
    used = set()
    tot_wi = tot_lo = tot_mfma = tot_lds = 0.0
    for phase, names in PHASES.items():
        wi = lo = mf = lds = bperm = 0.0
        static_valu = 0
        for n in names:
            r = st["regions"].get(n)
            if not r:
                continue
            used.add(n)
            if n == "OTHER":  # staging functions and compiler-generated code outside every marked region: once per wave
                v = S.get("K_WAVE", {"visits": 0, "lanes": 0})
            else:
                v = S.get(n, {"visits": 0, "lanes": 0})
            nc = float(r.get("copies", 1))
            valu = r.get("valu", 0) / nc
            static_valu += valu
            wi += valu * v["visits"]
            lo += valu * v["lanes"]
            mf += r.get("mfma", 0) / nc * v["visits"]
            lds += r.get("lds", 0) / nc * v["visits"]
            bperm += r.get("ds_bpermute", 0) / nc * v["visits"]
        rows.append({"phase": phase, "regions": [n for n in names if n in st["regions"]], "static_valu_instructions": static_valu,
                     "valu_wave_instructions_per_launch": wi, "valu_lane_ops_per_launch": lo,
                     "valu_wave_instructions_per_sample": wi / samples, "valu_lane_ops_per_sample": lo / samples,
                     "lane_utilisation": (lo / (64.0 * wi)) if wi else None,
                     "mfma_wave_instructions_per_launch": mf, "lds_wave_instructions_per_launch": lds, "ds_bpermute_per_launch": bperm})
        tot_wi += wi
        tot_lo += lo
        tot_mfma += mf
        tot_lds += lds
    missing = sorted(set(st["regions"]) - used)
    for r in rows:
        r["share_of_valu_wave_instructions"] = r["valu_wave_instructions_per_launch"] / tot_wi
    it = S.get("K_ITER", {"visits": 1})["visits"]
    out = {"config": cfg, "workload": dy["workload"], "kernel": st["kernel"], "samples_per_launch": samples,
           "method": "static VALU instructions per marked region of the shipped kernel (llvm-symbolizer inline stacks) x trip counts of "
                     "the regions from the -DRT_PHASES counting build (tools/phase_budget.py)",
           "phases": rows,
           "totals": {"valu_wave_instructions_per_launch": tot_wi, "valu_lane_ops_per_launch": tot_lo, "lane_utilisation": tot_lo / (64.0 * tot_wi),
                      "valu_wave_instructions_per_sample": tot_wi / samples, "valu_lane_ops_per_sample": tot_lo / samples,
                      "mfma_wave_instructions_per_launch": tot_mfma, "lds_wave_instructions_per_launch": tot_lds},
           "unattributed_regions": missing,
           "per_wave_iteration": {"wave_iterations_per_launch": it, "valu_wave_instructions": tot_wi / it,
                                  "live_rays_per_scan": (S["S_SCAN"]["visits"] and dy["traversals"] / S["S_SCAN"]["visits"]) if "S_SCAN" in S and S["S_SCAN"]["visits"] else None},
           "scan_statistics": {}}
    T = float(dy["traversals"])  # ray-vs-list scans incl. the shadow rays answered by the index
    seg = float(dy["segments"])
    ss = out["scan_statistics"]
    if S.get("S_SCAN", {}).get("visits"):
        scans = T - S.get("H_INDEXED", {"lanes": 0})["lanes"]  # scans that ran through the filter (closest hit + far-hit shadow scans)
        ss.update({"ray_scans_through_the_filter": scans, "wave_scans": S["S_SCAN"]["visits"],
                   "candidate_groups_per_scan": sum(S.get(n, {"lanes": 0})["lanes"] for n in ("S_TAKE_PUSH0", "S_TAKE_PUSH1", "S_SINGLE_PUSH0", "S_SINGLE_PUSH1", "S_PUSH_WORD")) / scans,
                   "push_loop_iterations_per_wave_scan": sum(S.get(n, {"visits": 0})["visits"] for n in ("S_TAKE_PUSH0", "S_TAKE_PUSH1", "S_SINGLE_PUSH0", "S_SINGLE_PUSH1", "S_PUSH_WORD")) / S["S_SCAN"]["visits"],
                   "phase_A_items_per_scan": S["S_ASTEP"]["lanes"] / scans,
                   "phase_A_steps_per_wave_scan": S["S_ASTEP"]["visits"] / S["S_SCAN"]["visits"],
                   "phase_A_lane_utilisation": S["S_ASTEP"]["lanes"] / (64.0 * S["S_ASTEP"]["visits"]),
                   "exact_tests_per_scan": S["S_BSTEP"]["lanes"] / scans,
                   "exact_steps_per_wave_scan": S["S_BSTEP"]["visits"] / S["S_SCAN"]["visits"],
                   "exact_lane_utilisation": S["S_BSTEP"]["lanes"] / (64.0 * S["S_BSTEP"]["visits"]),
                   "accepted_roots_per_scan (ds_min_u64 merges)": S["S_BMIN"]["lanes"] / scans,
                   "hits_per_scan": S["K_TRANS_HIT"]["lanes"] / seg if seg else None,
                   "note": "exact_lane_utilisation counts lanes of the wave that executed the step (all 64 do: `has` only masks the merge); "
                           "exact_tests_per_scan is the number of (ray, sphere) pairs evaluated"})
    if S.get("G_SCAN", {}).get("visits"):
        scans = T - S.get("H_INDEXED", {"lanes": 0})["lanes"]
        ss.update({"ray_scans_through_the_grid": scans, "wave_scans": S["G_SCAN"]["visits"],
                   "slab_items_per_scan": S["G_ROUND"]["lanes"] / scans, "item_rounds_per_wave_scan": S["G_ROUND"]["visits"] / S["G_SCAN"]["visits"],
                   "item_round_lane_utilisation": S["G_ROUND"]["lanes"] / (64.0 * S["G_ROUND"]["visits"]),
                   "bound_test_steps_per_wave_scan": S["G_STEP"]["visits"] / S["G_SCAN"]["visits"],
                   "bound_test_step_lane_utilisation": S["G_STEP"]["lanes"] / (64.0 * S["G_STEP"]["visits"]),
                   "exact_tests_per_scan": S["G_BSTEP"]["lanes"] / scans, "exact_steps_per_wave_scan": S["G_BSTEP"]["visits"] / S["G_SCAN"]["visits"],
                   "survivors_pushed_per_scan": S["G_PUSH"]["lanes"] / scans,
                   "accepted_roots_per_scan (ds_min_u64 merges)": S["G_BMIN"]["lanes"] / scans,
                   "hits_per_scan": S["K_TRANS_HIT"]["lanes"] / seg if seg else None})
    if pmc_path:
        d = json.load(open(pmc_path))
        t = d["raw_counters"]["trace"]
        k = d["derived_trace_kernel"]
        launches = max(1, int(k.get("launches_in_pmc_pass", 1)))
        out["pmc_check"] = {"source": os.path.relpath(pmc_path, ROOT), "SQ_INSTS_VALU_per_launch": t["SQ_INSTS_VALU"] / launches,
                            "SQ_THREAD_CYCLES_VALU_per_launch": t["SQ_THREAD_CYCLES_VALU"] / launches,
                            "budget_over_measured_wave_instructions": tot_wi / (t["SQ_INSTS_VALU"] / launches),
                            "budget_over_measured_lane_ops": tot_lo / (t["SQ_THREAD_CYCLES_VALU"] / launches),
                            "SQ_INSTS_MFMA_per_launch": t.get("SQ_INSTS_MFMA", 0) / launches, "budget_mfma": tot_mfma,
                            "SQ_INSTS_LDS_per_launch": t.get("SQ_INSTS_LDS", 0) / launches, "budget_lds": tot_lds,
                            "kernel_sources_sha256_of_pmc": d.get("kernel_sources_sha256")}
    path = os.path.join(ROOT, "profiles", "r04_phase_budget%s.json" % ("" if cfg == "c2" else "_" + cfg))
    json.dump(out, open(path, "w"), indent=1)
    print(path)
    for r in rows:
        print("  %-84s %6.1f wi/sample %7.1f lane-ops/sample  util %s  share %.3f" % (r["phase"], r["valu_wave_instructions_per_sample"] * 64, r["valu_lane_ops_per_sample"],
              ("%.2f" % r["lane_utilisation"]) if r["lane_utilisation"] else " -  ", r["share_of_valu_wave_instructions"]))
    print("  total lane-ops/sample %.1f, lane utilisation %.3f" % (tot_lo / samples, tot_lo / (64.0 * tot_wi)))
    if missing:
        print("  regions in no phase:", missing)
    if "pmc_check" in out:
        print("  budget / measured: wave-instructions %.3f, lane-ops %.3f" % (out["pmc_check"]["budget_over_measured_wave_instructions"], out["pmc_check"]["budget_over_measured_lane_ops"]))
    print("  scan statistics:", json.dumps(ss))


if __name__ == "__main__":
    mode = sys.argv[1]
    cfg = sys.argv[2] if len(sys.argv) > 2 else "c2"
    if mode == "static":
        do_static(cfg)
    elif mode == "dynamic":
        do_dynamic(cfg)
    else:
        do_combine(cfg, sys.argv[3] if len(sys.argv) > 3 else None)
