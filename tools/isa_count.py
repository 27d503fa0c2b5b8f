"""Static instruction mix of one kernel of a `-save-temps` build, whole function (early s_endpgm exits do not end the count) and per
basic block (label), so loop bodies can be read off.  usage: python tools/isa_count.py file.s <mangled-name regex> [--blocks]"""
import re
import sys

src, pat = sys.argv[1], re.compile(sys.argv[2])
blocks = "--blocks" in sys.argv
infn = False
cur = None
order = []


def new(name):
    d = {"name": name, "valu": 0, "f64": 0, "div": 0, "salu": 0, "lds": 0, "vmem": 0, "smem": 0, "wait": 0, "branch": 0, "to": []}
    order.append(d)
    return d


for line in open(src):
    if not infn:
        m = re.match(r"^(\S+):", line)
        if m and pat.search(m.group(1)) and not m.group(1).startswith("."):
            infn = True
            cur = new("entry")
        continue
    if re.match(r"^\.Lfunc_end", line):
        break
    m = re.match(r"^(\.LBB\S+):", line)
    if m:
        cur = new(m.group(1))
        continue
    t = line.split()
    if not t or not re.match(r"^[a-z_0-9]+$", t[0]):
        continue
    op = t[0]
    if op.startswith("v_"):
        cur["valu"] += 1
        if "f64" in op:
            cur["f64"] += 1
        if re.match(r"v_(rcp|rsq|sqrt|div_fixup|div_fmas|div_scale)", op):
            cur["div"] += 1
    elif op.startswith("ds_"):
        cur["lds"] += 1
    elif re.match(r"^(buffer_|global_|flat_|scratch_)", op):
        cur["vmem"] += 1
    elif op in ("s_waitcnt", "s_nop", "s_barrier"):
        cur["wait"] += 1
    elif op.startswith("s_cbranch") or op == "s_branch":
        cur["branch"] += 1
        if len(t) > 1:
            cur["to"].append(t[1])
    elif op.startswith("s_load") or op.startswith("s_buffer_load"):
        cur["smem"] += 1
    elif op.startswith("s_"):
        cur["salu"] += 1
keys = ("valu", "f64", "div", "salu", "lds", "vmem", "smem", "wait", "branch")
if blocks:
    print("%-14s" % "block" + "".join("%7s" % k for k in keys) + "  branches to")
    for d in order:
        print("%-14s" % d["name"] + "".join("%7d" % d[k] for k in keys) + "  " + " ".join(d["to"]))
print("total (static): " + ", ".join("%s %d" % (k, sum(d[k] for d in order)) for k in keys))
