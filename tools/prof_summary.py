"""Summarise a rocprofv3 rocpd sqlite database (kernel-trace) into per-kernel totals: python tools/prof_summary.py db [out.md]"""
import re
import sqlite3
import sys


def summarize(path, steps=None):
    db = sqlite3.connect(path)
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    cols = [r[1] for r in cur.execute(f"pragma table_info({kd})")]
    scol = [r[1] for r in cur.execute(f"pragma table_info({ks})")]
    name_col = "display_name" if "display_name" in scol else ("kernel_name" if "kernel_name" in scol else scol[-1])
    q = f"select s.{name_col}, count(*), sum(d.end - d.start), min(d.end - d.start), max(d.end - d.start) from {kd} d join {ks} s on d.kernel_id = s.id group by s.{name_col} order by 3 desc"
    rows = list(cur.execute(q))
    total = sum(r[2] for r in rows)
    lines = ["| kernel | calls | total ms | avg us | min us | max us | % |", "|---|---|---|---|---|---|---|"]
    for n, c, t, mn, mx in rows:
        n = re.sub(r"\(anonymous namespace\)::", "", n)
        n = re.sub(r"\(.*", "", n)[:90]
        lines.append(f"| {n} | {c} | {t / 1e6:.3f} | {t / c / 1e3:.1f} | {mn / 1e3:.1f} | {mx / 1e3:.1f} | {100.0 * t / total:.1f} |")
    lines.append(f"\ntotal kernel time {total / 1e6:.2f} ms over {sum(r[1] for r in rows)} dispatches")
    return "\n".join(lines)


if __name__ == "__main__":
    out = summarize(sys.argv[1])
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(out + "\n")
    print(out)
