"""Residency timeline of the SAD candidates grid (PCREG_SAD_TIMELINE dump): blocks per CU, lifetimes."""
import sys, collections
rows = [tuple(int(x) for x in l.split()) for l in open(sys.argv[1])]
t0 = min(r[1] for r in rows); t1 = max(r[2] for r in rows)
print(f"blocks {len(rows)}, span {(t1 - t0) / 100:.1f} us")
life = [(r[2] - r[1]) / 100 for r in rows]
print(f"lifetime us: min {min(life):.0f} mean {sum(life)/len(life):.0f} max {max(life):.0f}")
late = [r for r in rows if (r[1] - t0) / 100 > 50]
print(f"blocks starting > 50 us after the first: {len(late)}")
def cu_of(hw, xcc): return (xcc & 0xF, (hw >> 13) & 0x7, (hw >> 8) & 0xF)      # (xcc, se, cu)
per = collections.Counter(cu_of(r[3], r[4]) for r in rows)
print("distinct (xcc,se,cu):", len(per), "blocks/CU histogram:", sorted(collections.Counter(per.values()).items()))
first = collections.Counter(cu_of(r[3], r[4]) for r in rows if (r[1] - t0) / 100 <= 50)
print("first-round blocks/CU histogram:", sorted(collections.Counter(first.values()).items()))
