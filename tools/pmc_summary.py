"""Mean PMC counter values per kernel from a rocprofv3 results database (rocpd sqlite) or counter_collection.csv.
usage: python tools/pmc_summary.py <dir-or-file> [kernel-substring]"""
import glob, os, sqlite3, sys, csv, collections

def from_db(path, sub):
    c = sqlite3.connect(path)
    cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
    rows = c.execute("select * from counters_collection").fetchall()
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    ik, ic, iv = cols.index("kernel_name") if "kernel_name" in cols else cols.index("name"), cols.index("counter_name"), cols.index("value")
    for r in rows:
        if sub in r[ik]:
            acc[r[ik]][r[ic]].append(float(r[iv]))
    return acc, cols

def main():
    target, sub = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
    files = [target] if os.path.isfile(target) else glob.glob(os.path.join(target, "**", "*.db"), recursive=True)
    for f in files:
        acc, cols = from_db(f, sub)
        for k, d in acc.items():
            print(k[:90])
            for cn, v in sorted(d.items()):
                print(f"   {cn:28s} n={len(v):4d} mean={sum(v)/len(v):.6g}")

if __name__ == "__main__":
    main()
