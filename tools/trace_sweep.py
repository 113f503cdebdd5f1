# -*- coding: utf-8 -*-
"""Print the kernel timeline of one sweep from a rocprofv3 --kernel-trace database (rocpd sqlite).
   python tools/trace_sweep.py <results.db> [marker kernel substring = k_row_stats] [which occurrence from the end = 5]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    marker = sys.argv[2] if len(sys.argv) > 2 else 'k_row_stats'
    back = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    t = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    kd = [x for x in t if 'kernel_dispatch' in x][0]
    ks = [x for x in t if 'kernel_symbol' in x][0]
    rows = list(db.execute("select s.kernel_name, d.start, d.end, d.grid_size_x, d.grid_size_y, d.workgroup_size_x "
                           "from %s d join %s s on d.kernel_id = s.id order by d.start" % (kd, ks)))
    idx = [i for i, r in enumerate(rows) if marker in r[0]]
    a, b = idx[-back - 1], idx[-back]
    t0 = rows[a][1]
    busy = 0.0
    for r in rows[a:b]:
        busy += (r[2] - r[1]) / 1e3
        print('%8.1f %7.1f  %-64s grid %d x %d / %d' % ((r[1] - t0) / 1e3, (r[2] - r[1]) / 1e3, r[0][:64], r[3], r[4], r[5]))
    print('sweep %.1f us from start to start, %.1f us of kernels, %d launches' % ((rows[b][1] - t0) / 1e3, busy, b - a))


if __name__ == '__main__':
    main()
