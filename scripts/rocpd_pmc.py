#!/usr/bin/env python
"""Per-kernel PMC counter totals (sum over all counter instances, averaged over dispatches) from a
rocprofv3 rocpd SQLite database produced with --pmc."""
import sqlite3
import sys


def main(path, like='%osqp%'):
    db = sqlite3.connect(path)
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    ev = [t for t in tabs if t.startswith('rocpd_pmc_event')][0]
    inf = [t for t in tabs if t.startswith('rocpd_info_pmc')][0]
    kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
    ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
    q = (f"select s.kernel_name, i.name, sum(e.value), count(distinct d.id) from {ev} e "
         f"join {inf} i on e.pmc_id = i.id join {kd} d on e.event_id = d.event_id "
         f"join {ks} s on d.kernel_id = s.id where s.kernel_name like ? group by s.kernel_name, i.name")
    for name, ctr, total, nd in cur.execute(q, (like,)):
        print(f'{name[:60]:<60} {ctr:<24} per-dispatch total {total / nd:.6g}  ({nd} dispatches)')


if __name__ == '__main__':
    main(*sys.argv[1:])
