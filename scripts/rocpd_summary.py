#!/usr/bin/env python
"""Per-kernel summary (count / avg / min / max duration, launch geometry) from a rocprofv3 rocpd
SQLite database (what `rocprofv3 --kernel-trace --stats` writes on this ROCm build)."""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
    ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
    q = (f"select s.kernel_name, count(*), avg(d.end-d.start), min(d.end-d.start), max(d.end-d.start), "
         f"sum(d.end-d.start), max(d.workgroup_size_x), max(d.grid_size_x), max(d.group_segment_size), "
         f"max(d.private_segment_size) from {kd} d join {ks} s on d.kernel_id = s.id "
         f"group by s.kernel_name order by 6 desc")
    rows = list(cur.execute(q))
    tot = sum(r[5] for r in rows) or 1
    print(f'{"kernel":<70} {"calls":>5} {"avg_ns":>14} {"min_ns":>14} {"max_ns":>14} {"%":>6} '
          f'{"wg":>5} {"grid":>8} {"lds_B":>7} {"scratch_B":>9}')
    for r in rows:
        print(f'{r[0][:70]:<70} {r[1]:>5} {r[2]:>14.0f} {r[3]:>14} {r[4]:>14} {100*r[5]/tot:>6.2f} '
              f'{r[6]:>5} {r[7]:>8} {r[8]:>7} {r[9]:>9}')


if __name__ == '__main__':
    main(sys.argv[1])
