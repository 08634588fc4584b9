#!/usr/bin/env python3
"""Per-kernel average of every PMC counter in a rocprofv3 rocpd database (--pmc run)."""
import sqlite3
import sys
from collections import defaultdict


def main(path, match=''):
    db = sqlite3.connect(path)
    cols = [r[1] for r in db.execute("pragma table_info(counters_collection)")]
    rows = db.execute('select * from counters_collection').fetchall()
    name_i = cols.index('kernel_name') if 'kernel_name' in cols else None
    cn_i, val_i = cols.index('counter_name'), cols.index('value')
    did_i = cols.index('dispatch_id')
    acc = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(set)
    for r in rows:
        k = r[name_i]
        if match and match not in k:
            continue
        acc[k][r[cn_i]] += r[val_i]
        cnt[k].add(r[did_i])
    for k in acc:
        n = len(cnt[k])
        print('%s  (%d dispatches)' % (k[:100], n))
        for c, v in sorted(acc[k].items()):
            print('    %-34s %16.1f per dispatch' % (c, v / n))


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else '')
