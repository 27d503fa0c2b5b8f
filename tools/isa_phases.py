"""Static instruction counts of one kernel between consecutive s_memtime instructions (the PROF_MARK boundaries of a -DITD_PROF=1
build; file order of the assembly, loops counted once — backward branches are listed so they can be weighted by hand).
usage: python tools/isa_phases.py file.s <mangled-name regex>"""
import re
import sys

src, pat = sys.argv[1], re.compile(sys.argv[2])
infn, seg, segs, labels = False, None, [], {}
for ln, line in enumerate(open(src)):
    if not infn:
        m = re.match(r"^(\S+):", line)
        if m and pat.search(m.group(1)) and not m.group(1).startswith("."):
            infn, seg = True, {"valu": 0, "salu": 0, "lds": 0, "vmem": 0, "smem": 0, "wait": 0, "branch": 0, "f64": 0, "trans": 0, "back": [], "first": ln}
            segs.append(seg)
        continue
    m = re.match(r"^(\.LBB\S+):", line)
    if m:
        labels[m.group(1)] = len(segs) - 1
        continue
    t = line.split()
    if not t or not re.match(r"^[a-z_0-9]+$", t[0]):
        continue
    op = t[0]
    if op == "s_memtime":
        seg = {"valu": 0, "salu": 0, "lds": 0, "vmem": 0, "smem": 0, "wait": 0, "branch": 0, "f64": 0, "trans": 0, "back": [], "first": ln}
        segs.append(seg)
        continue
    if op.startswith("v_"):
        seg["valu"] += 1
        if "f64" in op:
            seg["f64"] += 1
        if re.match(r"v_(rcp|rsq|sqrt|div_fixup|div_fmas|div_scale)", op):
            seg["trans"] += 1
    elif op.startswith("ds_"):
        seg["lds"] += 1
    elif op.startswith("buffer_") or op.startswith("global_") or op.startswith("flat_") or op.startswith("scratch_"):
        seg["vmem"] += 1
    elif op in ("s_waitcnt", "s_nop", "s_barrier"):
        seg["wait"] += 1
    elif op.startswith("s_cbranch") or op == "s_branch":
        seg["branch"] += 1
        tgt = t[1] if len(t) > 1 else ""
        if tgt in labels:
            seg["back"].append("%s->seg%d" % (tgt, labels[tgt]))
    elif op.startswith("s_load") or op.startswith("s_buffer_load"):
        seg["smem"] += 1
    elif op == "s_endpgm":
        break
    elif op.startswith("s_"):
        seg["salu"] += 1
print("%-4s %6s %6s %6s %6s %6s %6s %6s %6s %6s  %s" % ("seg", "valu", "f64", "div*", "salu", "lds", "vmem", "smem", "wait", "branch", "backward branches (loops)"))
for i, d in enumerate(segs):
    print("%-4d %6d %6d %6d %6d %6d %6d %6d %6d %6d  %s" % (i, d["valu"], d["f64"], d["trans"], d["salu"], d["lds"], d["vmem"], d["smem"], d["wait"], d["branch"], " ".join(d["back"])))
tot = {k: sum(d[k] for d in segs) for k in ("valu", "salu", "lds", "vmem")}
print("total (static)", tot)
