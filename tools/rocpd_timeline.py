#!/usr/bin/env python3
"""Kernel timeline (start offset, duration, queue) of the dispatches whose start lies in a window of
the LAST `back_ms` milliseconds of the trace: shows what actually overlaps."""
import re, sqlite3, sys
def main(path, back_us, span_us):
    db = sqlite3.connect(path)
    cols = [r[1] for r in db.execute('pragma table_info(kernels)')]
    q = 'queue_id' if 'queue_id' in cols else ('stream_id' if 'stream_id' in cols else None)
    rows = db.execute('select name, start, end%s from kernels order by start' % (', ' + q if q else '')).fetchall()
    t_end = rows[-1][2]
    t0 = t_end - back_us * 1000
    for r in rows:
        if t0 <= r[1] < t0 + span_us * 1000:
            nm = re.sub(r'\(anonymous namespace\)::|^void |sf::', '', r[0])[:60]
            print('%9.1f +%6.1f us  q%-3s %s' % ((r[1] - t0) / 1e3, (r[2] - r[1]) / 1e3, r[3] if q else '?', nm))
if __name__ == '__main__':
    main(sys.argv[1], float(sys.argv[2]), float(sys.argv[3]))
