"""Static instruction counts of the chain kernel between its CHAIN_MARK comments (file order of the assembly: approximate, the
compiler moves blocks).  usage: python tools/isa_regions.py file.s"""
import re, sys
cur, acc, order = "start", {}, []
infn = False
for line in open(sys.argv[1]):
    if re.match(r"^_ZN3itd7k_chain.*:", line):
        infn = True
    if not infn:
        continue
    m = re.search(r"CHAIN_MARK (\d+)", line)
    if m:
        cur = m.group(1)
        if cur not in acc:
            order.append(cur)
        continue
    t = line.split()
    if not t or not re.match(r"^[a-z_0-9]+$", t[0]) or t[0].startswith("."):
        continue
    op = t[0]
    kind = "valu" if op.startswith("v_") else "salu" if op.startswith("s_") else "lds" if op.startswith("ds_") else "vmem"
    d = acc.setdefault(cur, {"valu": 0, "salu": 0, "lds": 0, "vmem": 0})
    if cur not in order:
        order.append(cur)
    d[kind] += 1
    if "s_endpgm" in line:
        break
print("%-8s %6s %6s %6s %6s" % ("after", "valu", "salu", "lds", "vmem"))
for k in order:
    d = acc[k]
    print("%-8s %6d %6d %6d %6d" % (k, d["valu"], d["salu"], d["lds"], d["vmem"]))
