"""Condense rocprofv3 CSV output (kernel stats + PMC counter_collection) into small text/JSON
summaries that are committed under profiles/.

    python tools/summarize_prof.py stats  <kernel_stats.csv>              > profiles/xxx_kernel_stats.txt
    python tools/summarize_prof.py pmc    <counter_collection.csv> [...]  > profiles/xxx_pmc.json

Either kind of input may also be a rocpd ``*_results.db`` (rocprofv3's default output format in
ROCm 7.2): the ``kernels`` / ``counters_collection`` views hold the same rows.
"""
import collections
import csv
import json
import re
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([A-Za-z_0-9:]+(<[^(]*>)?)", name)
    return m.group(1) if m else name[:60]


def db_rows(path, query):
    import sqlite3
    con = sqlite3.connect(path)
    cur = con.execute(query)
    names = [d[0] for d in cur.description]
    return [dict(zip(names, r)) for r in cur.fetchall()]


def stats_rows_from_db(path):
    per = collections.defaultdict(list)
    for r in db_rows(path, "select name, duration from kernels"):
        per[r["name"]].append(float(r["duration"]))
    total = sum(sum(v) for v in per.values())
    rows = [{"Name": k, "Calls": len(v), "AverageNs": sum(v) / len(v), "MinNs": min(v), "MaxNs": max(v),
             "MedianNs": sorted(v)[len(v) // 2], "Percentage": 100.0 * sum(v) / total, "_total": sum(v)}
            for k, v in per.items()]
    rows.sort(key=lambda r: -r["_total"])
    return rows


def stats(path):
    rows = stats_rows_from_db(path) if path.endswith(".db") else list(csv.DictReader(open(path)))
    print("%-44s %7s %12s %12s %12s %12s %7s" % ("kernel", "calls", "avg_us", "median_us", "min_us", "max_us", "pct"))
    for r in rows:
        med = "%12.2f" % (float(r["MedianNs"]) / 1e3) if "MedianNs" in r else "%12s" % "-"
        print("%-44s %7s %12.2f %s %12.2f %12.2f %7.2f" % (short(r["Name"]), r["Calls"], float(r["AverageNs"]) / 1e3, med,
                                                          float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3,
                                                          float(r["Percentage"])))


def pmc(paths):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for path in paths:
        if path.endswith(".db"):
            rows = [{"Kernel_Name": r["kernel_name"], "Counter_Name": r["counter_name"], "Counter_Value": r["value"]}
                    for r in db_rows(path, "select kernel_name, counter_name, value from counters_collection")]
        else:
            rows = csv.DictReader(open(path))
        for r in rows:
            agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out = {}
    for kern, counters in agg.items():
        out[kern] = {}
        for c, v in counters.items():
            v.sort()
            out[kern][c] = {"launches": len(v), "median": v[len(v) // 2], "min": v[0], "max": v[-1]}
    print(json.dumps(out, indent=1, sort_keys=True))


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2])
    else:
        pmc(sys.argv[2:])
