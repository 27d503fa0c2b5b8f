"""Static instruction counts of one kernel attributed to source lines (.loc directives of a -gline-tables-only assembly):
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -gline-tables-only -S --offload-device-only -o engine.s pyitd_amd/csrc/itd_engine.hip
  python tools/isa_lines.py engine.s '<mangled-name regex>' [min-count]
Loops are counted once (file order); inlined code is attributed to the innermost source line."""
import collections
import re
import sys

src, pat = sys.argv[1], re.compile(sys.argv[2])
thresh = int(sys.argv[3]) if len(sys.argv) > 3 else 4
files, infn, cur = {}, False, None
cnt = collections.defaultdict(lambda: collections.Counter())
for line in open(src, errors="replace"):
    m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', line)
    if m:
        files[m.group(1)] = (m.group(3) or m.group(2)).split("/")[-1]
        continue
    if not infn:
        m = re.match(r"^(\S+):", line)
        if m and pat.search(m.group(1)) and not m.group(1).startswith("."):
            infn = True
        continue
    m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", line)
    if m:
        cur = (files.get(m.group(1), m.group(1)), int(m.group(2)))
        continue
    t = line.split()
    if not t or not re.match(r"^[a-z_0-9]+$", t[0]):
        continue
    op = t[0]
    if op == "s_endpgm":
        cnt[cur]["salu"] += 1
        break
    kind = ("valu" if op.startswith("v_") else "lds" if op.startswith("ds_") else
            "vmem" if re.match(r"(buffer|global|flat|scratch)_", op) else "wait" if op in ("s_waitcnt", "s_nop") else "salu")
    cnt[cur][kind] += 1
    if op.startswith("v_mov") or op.startswith("v_cndmask") or op.startswith("v_readlane") or op.startswith("v_writelane") or op.startswith("v_accvgpr"):
        cnt[cur]["mov"] += 1
tot = collections.Counter()
for k, c in cnt.items():
    tot.update(c)
print("total", dict(tot))
print("%-28s %5s %5s %5s %5s %5s" % ("file:line", "valu", "(mov)", "salu", "lds", "vmem"))
for k in sorted(cnt, key=lambda k: (k[0], k[1]) if k else ("", 0)):
    c = cnt[k]
    if c["valu"] + c["salu"] + c["lds"] + c["vmem"] >= thresh:
        print("%-28s %5d %5d %5d %5d %5d" % ("%s:%d" % k if k else "?", c["valu"], c["mov"], c["salu"], c["lds"], c["vmem"]))
