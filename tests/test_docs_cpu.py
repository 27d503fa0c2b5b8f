"""The documents point at evidence: every profiles/... and tools/... path that DESIGN.md, README.md, INTEGRATION.md and
profiles/README.md name must exist in the tree (a renamed or deleted file must not leave a dangling reference), and no file
under profiles/ may claim more than the HBM peak."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DOCS = ["DESIGN.md", "README.md", "INTEGRATION.md", os.path.join("profiles", "README.md"), os.path.join("tools", "README.md")]


def _paths(text, base):
    out = set()
    for m in re.finditer(r"`((?:profiles|tools|tests|oracle|include|pyitd_amd)/[A-Za-z0-9_./\-]+)`", text):
        out.add(m.group(1))
    # profiles/README.md and tools/README.md name their own files without the directory
    ext = {"profiles": "txt|csv|json|log", "tools": "py|sh|hip|c"}.get(base)
    if ext:
        for m in re.finditer(r"`((?:r0[0-9]/)?[A-Za-z0-9_\-]+\.(?:%s))`" % ext, text):
            name = m.group(1)
            out.add(name if name in ("bench.py",) else os.path.join(base, name))     # (the root's bench.py is named everywhere)
    return out


# named on purpose although absent from the tree: the reference build that cannot exist, binaries built on the GPU box
ABSENT_ON_PURPOSE = {"oracle/_ref", "tools/membench", "tools/membench2", "tools/membench3", "tools/dispatch_bench", "tools/c_timing"}


def test_every_named_evidence_file_exists():
    missing = []
    for doc in DOCS:
        text = open(os.path.join(ROOT, doc)).read()
        base = os.path.dirname(doc)
        for p in sorted(_paths(text, base)):
            if "*" in p or p.endswith("/") or "..." in p or p in ABSENT_ON_PURPOSE:
                continue
            full = os.path.join(ROOT, p)
            if not os.path.exists(full):
                # profiles/README.md keeps rows for older rounds' raw session files by prefix (r02/session1_*): checked by pattern above
                missing.append("%s: %s" % (doc, p))
    assert not missing, "dangling references:\n" + "\n".join(missing)


def test_no_profile_claims_more_than_the_peak():
    bad = []
    for dirpath, _, files in os.walk(os.path.join(ROOT, "profiles")):
        for f in files:
            if not f.endswith((".txt", ".json")):
                continue
            text = open(os.path.join(dirpath, f), errors="replace").read()
            for m in re.finditer(r"= (\d+\.\d+) of peak", text):
                if float(m.group(1)) > 1.0:
                    bad.append("%s: %s" % (f, m.group(0)))
            for m in re.finditer(r'"frac[a-z_]*": (\d+\.\d+)', text):
                if float(m.group(1)) > 1.0:
                    bad.append("%s: %s" % (f, m.group(0)))
    assert not bad, "\n".join(bad[:20])


def test_no_profile_is_a_crashed_run():
    """An evidence file that holds a Python traceback (round 4 shipped one: a diagnostic library that no longer matched the ABI) is
    not evidence: whatever writes into profiles/ must have run to the end."""
    bad = []
    for dirpath, _, files in os.walk(os.path.join(ROOT, "profiles")):
        for f in files:
            if f.endswith((".txt", ".json", ".log", ".csv", ".md")):
                text = open(os.path.join(dirpath, f), errors="replace").read()
                if "Traceback (most recent call last)" in text or "undefined symbol" in text:
                    bad.append(os.path.relpath(os.path.join(dirpath, f), ROOT))
    assert not bad, "crashed runs kept as evidence: " + ", ".join(bad)


def test_design_quotes_the_last_bench_line_s_operator_figures():
    """DESIGN.md section 5 quotes the SURVEY 8f operators' figures of the round's bench line: MEITD's milliseconds and the block-wise
    microseconds must be the ones in profiles/r06/bench_default_form.json (round 5 shipped a 7.0 ms that every other place had at 2.7)."""
    import json
    text = open(os.path.join(ROOT, "DESIGN.md")).read()
    line = json.load(open(os.path.join(ROOT, "profiles", "r06", "bench_default_form.json")))["f_rows"]
    m = re.search(r"MEITD on the golden two-tone \+\s+noise signal ([0-9.]+) ms", text)
    assert m, "DESIGN.md section 5 no longer quotes MEITD's time"
    assert abs(float(m.group(1)) - line["meitd_two_tone_noise_3000"]["ms"]) <= 0.15 * line["meitd_two_tone_noise_3000"]["ms"]
    m = re.search(r"block-wise ([0-9.]+) / ([0-9.]+) µs per 4096-sample block", text)
    assert m, "DESIGN.md section 5 no longer quotes the block-wise operators' time"
    assert abs(float(m.group(1)) - line["stream_cubic_block4096"]["us_per_block"]) <= 0.15 * line["stream_cubic_block4096"]["us_per_block"]
    assert abs(float(m.group(2)) - line["stream_linear_block4096"]["us_per_block"]) <= 0.15 * line["stream_linear_block4096"]["us_per_block"]
