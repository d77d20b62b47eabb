#!/usr/bin/env python
"""Summarise a rocprofv3 result (rocpd SQLite `*_results.db`, the default output of
`rocprofv3 --kernel-trace --stats`) as a small markdown table: calls, total / mean / min / max
duration per kernel.  Usage: tools/rocpd_summary.py <results.db> [title] > profiles/<name>.md"""
import re
import sqlite3
import sys


def short(name: str) -> str:
    name = re.sub(r"\(.*", "", name)           # drop the argument list
    name = name.replace("void ", "")
    return name if len(name) <= 70 else name[:67] + "..."


def main():
    db = sys.argv[1]
    title = sys.argv[2] if len(sys.argv) > 2 else db
    con = sqlite3.connect(db)
    rows = con.execute(
        "select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
        "from kernels group by name order by 3 desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    print(f"# {title}\n")
    print("Source: `rocprofv3 --kernel-trace --stats` (rocpd database), durations in microseconds.\n")
    print("| kernel | calls | total us | mean us | min us | max us | % |")
    print("|---|---:|---:|---:|---:|---:|---:|")
    other = [0, 0.0]
    for name, n, tot, avg, mn, mx in rows:
        if "lsr::" in name or tot / total > 0.01:
            print(f"| `{short(name)}` | {n} | {tot/1e3:.1f} | {avg/1e3:.2f} | {mn/1e3:.2f} | {mx/1e3:.2f} | {100*tot/total:.1f} |")
        else:
            other[0] += n
            other[1] += tot
    print(f"| (other: torch fills/copies, {other[0]} launches) | {other[0]} | {other[1]/1e3:.1f} | | | | {100*other[1]/total:.1f} |")


if __name__ == "__main__":
    main()
