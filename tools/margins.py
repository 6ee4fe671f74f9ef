"""How far every tolerance check of the parity suite sits from its bound:
    RDRF_MARGINS=/tmp/m.tsv python -m pytest tests -m gpu -q ; python tools/margins.py /tmp/m.tsv [top]
(tests/_util.py record_margin writes one `test <tab> check <tab> error / tolerance` line per comparison.)"""
import collections
import sys

rows = [l.rstrip("\n").split("\t") for l in open(sys.argv[1])]
top = int(sys.argv[2]) if len(sys.argv) > 2 else 25
rows = [(float(r[2]), r[0], r[1]) for r in rows if len(r) == 3]
print(f"{len(rows)} checks; {sum(r[0] > 0.5 for r in rows)} above 0.5 of their tolerance, {sum(r[0] > 0.8 for r in rows)} above 0.8")
per_test = collections.defaultdict(float)
for v, t, c in rows:
    per_test[t] = max(per_test[t], v)
print("worst checks:")
for v, t, c in sorted(rows, reverse=True)[:top]:
    print(f"  {v:6.3f}  {t}  {c}")
print("worst per test:")
for t, v in sorted(per_test.items(), key=lambda kv: -kv[1])[:top]:
    print(f"  {v:6.3f}  {t}")
