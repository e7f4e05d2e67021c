#!/usr/bin/env python3
"""Per-kernel-NAME table of arbitrary rocprofv3 --pmc counters (one or more result databases, e.g. two SQ passes):
sums per kernel name, printed per launch and -- for the SQ wave-state counters -- as fractions of SQ_WAVE_CYCLES.
usage: tools/pmc_sq.py <out.txt> <pass1.db> [pass2.db ...] [--min-ms X] [--match substring]"""
import collections, re, sqlite3, sys

args = [a for a in sys.argv[1:] if not a.startswith("--")]
opts = sys.argv[1:]
min_ms = float(opts[opts.index("--min-ms") + 1]) if "--min-ms" in opts else 0.0
match = opts[opts.index("--match") + 1] if "--match" in opts else ""
args = [a for a in args if a not in (str(min_ms), match)] if ("--min-ms" in opts or "--match" in opts) else args
out, dbs = args[0], [a for a in args[1:] if a.endswith(".db")]


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"\(.*$", "", n)
    return n[:80]


tot = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(set))
dur = collections.defaultdict(float)
for db in dbs:
    con = sqlite3.connect(db)
    seen = set()
    for did, k, c, v, d in con.execute("select dispatch_id, kernel_name, counter_name, value, duration from counters_collection"):
        k = short(k)
        tot[k][c] += v
        cnt[k][c].add(did)
        if (db, did) not in seen:
            seen.add((db, did))
            if db == dbs[0]:
                dur[k] += d
lines = []
for k in sorted(tot, key=lambda k: -dur[k]):
    if dur[k] / 1e6 < min_ms or (match and match not in k):
        continue
    n = max(len(s) for s in cnt[k].values())
    per = {c: tot[k][c] / max(len(cnt[k][c]), 1) for c in tot[k]}
    wc = per.get("SQ_WAVE_CYCLES", 0.0)
    lines.append(f"{k}  launches={n} avg={dur[k] / max(n, 1) / 1e3:.1f}us")
    for c in sorted(per):
        frac = f"  = {per[c] / wc:.3f} of SQ_WAVE_CYCLES" if wc and c.startswith("SQ_") and ("WAIT" in c or "ACTIVE" in c or "BUSY_CYCLES" == c[-11:]) and c != "SQ_WAVE_CYCLES" else ""
        lines.append(f"    {c:32s} {per[c]:16.0f} per launch{frac}")
    if "SQ_VALU_MFMA_BUSY_CYCLES" in per and "GRBM_GUI_ACTIVE" in per and per["GRBM_GUI_ACTIVE"] > 0:
        lines.append(f"    MFMA busy = {per['SQ_VALU_MFMA_BUSY_CYCLES'] / (per['GRBM_GUI_ACTIVE'] / 8.0 * 1024):.3f}")
open(out, "w").write("\n".join(lines) + "\n")
print("\n".join(lines[:200]))
