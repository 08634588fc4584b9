#!/usr/bin/env python3
"""Kernel sequence (name, duration, gap to the previous kernel) for dispatches [i0, i0+n)."""
import re
import sqlite3
import sys


def main(path, i0, n):
    db = sqlite3.connect(path)
    rows = db.execute('select name, start, end from kernels order by start').fetchall()
    prev = None
    for name, s, e in rows[i0:i0 + n]:
        nm = re.sub(r'\(anonymous namespace\)::|^void |sf::', '', name)[:70]
        print('%7.1f us  gap %6.1f  %s' % ((e - s) / 1e3, (s - prev) / 1e3 if prev else 0.0, nm))
        prev = e


if __name__ == '__main__':
    main(sys.argv[1], int(sys.argv[2]), int(sys.argv[3]))
