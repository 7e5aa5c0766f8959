"""Summarises the rocprofv3 CSVs written by tools/profile.sh into one text file for profiles/."""
import csv, glob, os, sys, collections

def load(pattern):
    rows = []
    for f in glob.glob(pattern, recursive=True):
        with open(f) as fh:
            rows += list(csv.DictReader(fh))
    return rows

def main(d, out):
    lines = []
    cmd = os.path.join(d, "command.txt")
    if os.path.exists(cmd):
        lines.append("command under rocprofv3 (every pass): " + open(cmd).read().strip())
    st = load(os.path.join(d, "trace", "**", "*kernel_stats.csv"))
    lines.append("== rocprofv3 --kernel-trace --stats (kernel_stats.csv) ==")
    for r in st:
        lines.append("  %-90s calls %5s  avg %12.1f ns  total %14s ns  %6s%%" % (
            r["Name"][:90], r["Calls"], float(r["AverageNs"]), r["TotalDurationNs"], r["Percentage"]))
    for p in sorted(glob.glob(os.path.join(d, "pmc*"))):
        rows = load(os.path.join(p, "**", "*counter_collection.csv"))
        agg = collections.defaultdict(lambda: collections.defaultdict(float))
        cnt = collections.defaultdict(lambda: collections.defaultdict(int))
        for r in rows:
            k = r["Kernel_Name"][:60]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[k][r["Counter_Name"]] += 1
        lines.append("== %s: per-dispatch AVERAGE of each counter ==" % os.path.basename(p))
        for k in agg:
            lines.append("  " + k)
            for c in sorted(agg[k]):
                lines.append("      %-28s %18.1f   (%d dispatches)" % (c, agg[k][c] / cnt[k][c], cnt[k][c]))
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))

if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
