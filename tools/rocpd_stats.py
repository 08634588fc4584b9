#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd sqlite database (the default output of `rocprofv3
--kernel-trace --stats`) as a per-kernel table: calls, total / average / min / max duration.

    python tools/rocpd_stats.py gpurun_out/prof/x_results.db > profiles/r01_x_kernel_stats.txt
"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    name = re.sub(r'^void ', '', name)
    return name if len(name) <= 110 else name[:107] + '...'


def main(path, skip_first=0):
    db = sqlite3.connect(path)
    rows = db.execute('select name, start, end from kernels order by start').fetchall()
    stats = {}
    for name, s, e in rows:
        d = stats.setdefault(name, [])
        d.append(e - s)
    total = sum(sum(v) for v in stats.values())
    print('# %s: %d kernel dispatches, %.3f ms total GPU kernel time' % (path, len(rows), total / 1e6))
    print('%-112s %8s %11s %10s %10s %10s %6s' % ('kernel', 'calls', 'total_ms', 'avg_us', 'min_us',
                                                   'max_us', '%'))
    for name, v in sorted(stats.items(), key=lambda kv: -sum(kv[1])):
        print('%-112s %8d %11.3f %10.2f %10.2f %10.2f %6.2f' % (
            short(name), len(v), sum(v) / 1e6, sum(v) / len(v) / 1e3, min(v) / 1e3, max(v) / 1e3,
            100.0 * sum(v) / total))


if __name__ == '__main__':
    main(sys.argv[1])
