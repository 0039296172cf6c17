"""Summarise a rocprofv3 rocpd sqlite database: per-kernel calls / total / average duration
(the --stats table) and, if present, PMC counter sums per kernel.

    python tools/rocpd_summary.py gpurun_out/prof/x_results.db [out.md]
"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(.*", "", name)
    m = re.search(r"(k_[a-z_]+)(<[^>]*>)?", name)
    return (m.group(1) + (m.group(2) or "")) if m else name[:60]


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    rows = cur.execute("select name, start, end from kernels").fetchall()
    agg = {}
    for name, s, e in rows:
        k = short(name)
        a = agg.setdefault(k, [0, 0.0, 1e30, 0.0])
        d = (e - s) / 1e3
        a[0] += 1
        a[1] += d
        a[2] = min(a[2], d)
        a[3] = max(a[3], d)
    total = sum(a[1] for a in agg.values())
    lines = ["| kernel | calls | total us | avg us | min us | max us | % |", "|---|---|---|---|---|---|---|"]
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        lines.append(f"| {k} | {a[0]} | {a[1]:.1f} | {a[1] / a[0]:.2f} | {a[2]:.2f} | {a[3]:.2f} | {100 * a[1] / total:.1f} |")
    try:
        pmc = cur.execute("select * from counters_collection limit 1").fetchall()
        ccols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
        if pmc:
            q = cur.execute("select kernel_name, counter_name, sum(value), count(*) from counters_collection "
                            "group by kernel_name, counter_name").fetchall()
            lines += ["", "| kernel | counter | sum | dispatches |", "|---|---|---|---|"]
            for kn, cn, v, n in q:
                lines.append(f"| {short(kn)} | {cn} | {v:.6g} | {n} |")
    except sqlite3.Error as ex:
        lines.append(f"(no counters: {ex})")
    text = "\n".join(lines)
    print(text)
    if len(sys.argv) > 2:
        with open(sys.argv[2], "w") as f:
            f.write(text + "\n")


if __name__ == "__main__":
    main()
