#!/usr/bin/env python3
"""Phases of the last training iteration in a rocprofv3 kernel trace: the iteration is cut at its adam_kernel
launches; prints, per phase (forward encoder / decode steps / backward through time / weight gradients + encoder
backward / optimizer), wall time, per-queue busy time and the time no queue is busy."""
import re, sqlite3, sys
def main(path):
    db = sqlite3.connect(path)
    cols = [r[1] for r in db.execute('pragma table_info(kernels)')]
    q = 'queue_id' if 'queue_id' in cols else 'stream_id'
    rows = db.execute('select name, start, end, %s from kernels order by start' % q).fetchall()
    adam = [i for i, r in enumerate(rows) if 'adam_kernel' in r[0]]
    # iterations end with two adam launches
    end = adam[-1]
    begin = adam[-3] + 1
    it = rows[begin:end + 1]
    t0 = it[0][1]
    def nm(n): return re.sub(r'\(anonymous namespace\)::|^void |sf::', '', n).split('(')[0][:40]
    marks = {}
    for name, s, e, qq in it:
        n = nm(name)
        if 'enc_persist_kernel' in n: marks['encoder end'] = e
        if 'reduce_terms' in n: marks['forward end'] = e
        if 'ctx_grad_kernel' in n: marks['bptt end'] = e
        if 'enc_bwd_persist_kernel' in n: marks['enc bwd end'] = e
    last = it[-1][2]
    print('iteration: %.1f us, %d kernels' % ((last - t0) / 1e3, len(it)))
    prev = t0
    for k in ('encoder end', 'forward end', 'bptt end', 'enc bwd end'):
        if k in marks:
            seg = [(s, e, qq) for _, s, e, qq in it if s >= prev and s < marks[k]]
            busy = {}
            for s, e, qq in seg: busy[qq] = busy.get(qq, 0) + (e - s)
            # union of busy intervals
            iv = sorted((s, e) for s, e, _ in seg)
            covered, cur_s, cur_e = 0, None, None
            for s, e in iv:
                if cur_e is None or s > cur_e:
                    if cur_e is not None: covered += cur_e - cur_s
                    cur_s, cur_e = s, e
                else: cur_e = max(cur_e, e)
            if cur_e is not None: covered += cur_e - cur_s
            wall = marks[k] - prev
            print('%-14s wall %8.1f us  any-queue-busy %8.1f us  idle %7.1f us  per-queue busy %s  kernels %d'
                  % (k, wall / 1e3, covered / 1e3, (wall - covered) / 1e3, {qq: round(v / 1e3, 1) for qq, v in busy.items()}, len(seg)))
            prev = marks[k]
    print('tail (adam etc) %.1f us' % ((last - prev) / 1e3))
if __name__ == '__main__':
    main(sys.argv[1])
