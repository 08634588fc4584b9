#!/usr/bin/env python3
"""Every dispatch of kernels whose name contains <match>: duration and grid, in launch order."""
import sqlite3
import sys


def main(path, match):
    db = sqlite3.connect(path)
    cols = [r[1] for r in db.execute('pragma table_info(kernels)')]
    gx = [c for c in cols if 'grid' in c.lower()]
    q = 'select name, start, end%s from kernels order by start' % (''.join(', ' + c for c in gx))
    for row in db.execute(q):
        if match in row[0]:
            print('%8.1f us  grid %s' % ((row[2] - row[1]) / 1e3, row[3:]))


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2])
