"""The navigation geometry of the 90 Matterport3D scans of R2R (10 567 included viewpoints), shipped as one
compact table file (data/r2r_connectivity.npz, built by tools/make_nav_geometry.py from the connectivity
data files) and turned back into a `<scan>_connectivity.json` directory on demand.

Only the fields the navigation-only simulator (sim/mattersim_nav.cpp:143-170) and the planner
(env.NavGraph; utils.py:26-51) read are kept: image_id, the translation of the pose, included,
unobstructed.  Positions are float64 and written with repr(), so a regenerated file parses to exactly the
numbers of the original one.
"""
import hashlib
import json
import os
import tempfile

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
GEOMETRY = os.path.join(_PKG, 'data', 'r2r_connectivity.npz')


def load_geometry(path=GEOMETRY):
    """{scan: dict(ids [n] str, included [n] bool, pos [n,3] float64, unobstructed [n,n] bool)}."""
    z = np.load(path)
    out, o, ob = {}, 0, 0
    for s, n, nb in zip(z['scans'], z['n'], z['bits_n']):
        n, nb = int(n), int(nb)
        un = np.unpackbits(z['unobstructed_bits'][ob:ob + nb])[:n * n].reshape(n, n).astype(bool)
        out[str(s)] = dict(ids=[i.decode() for i in z['ids'][o:o + n]], included=z['included'][o:o + n],
                           pos=z['pos'][o:o + n], unobstructed=un)
        o += n
        ob += nb
    return out


def connectivity_dir(path=GEOMETRY, scans=None, out_dir=None):
    """Writes `<scan>_connectivity.json` (+ scans.txt) for `scans` (default: all 90) and returns the directory.
    Default location: a per-user cache under the system temp directory keyed by the table file's hash, so a
    second call (or process) finds the files in place."""
    geo = load_geometry(path)
    if out_dir is None:
        h = hashlib.sha1(open(path, 'rb').read()).hexdigest()[:12]
        out_dir = os.path.join(tempfile.gettempdir(), 'sf_connectivity_%s_%d' % (h, os.getuid()))
    os.makedirs(out_dir, mode=0o700, exist_ok=True)
    st = os.stat(out_dir)
    if st.st_uid != os.getuid() or (st.st_mode & 0o022):
        # a predictable path under a shared temp directory: contents are trusted only if the directory is ours alone
        raise RuntimeError('%s is not owned by this user or is writable by others; remove it or pass out_dir' % out_dir)
    names = list(geo) if scans is None else list(scans)
    for s in names:
        f = os.path.join(out_dir, s + '_connectivity.json')
        if os.path.exists(f):
            continue
        g = geo[s]
        rows = []
        for i, vid in enumerate(g['ids']):
            x, y, z = (float(v) for v in g['pos'][i])
            rows.append('{"image_id":"%s","pose":[1,0,0,%r,0,1,0,%r,0,0,1,%r,0,0,0,1],"included":%s,"unobstructed":[%s]}'
                        % (vid, x, y, z, 'true' if g['included'][i] else 'false',
                           ','.join('true' if u else 'false' for u in g['unobstructed'][i])))
        tmp = f + '.tmp%d' % os.getpid()
        with open(tmp, 'w') as fh:
            fh.write('[' + ',\n'.join(rows) + ']')
        os.replace(tmp, f)                      # atomic: concurrent ranks may build the same cache
    # scans.txt lists what the directory HOLDS (the union over all callers), written atomically
    present = sorted(f[:-len('_connectivity.json')] for f in os.listdir(out_dir) if f.endswith('_connectivity.json'))
    tmp = os.path.join(out_dir, 'scans.txt.tmp%d' % os.getpid())
    with open(tmp, 'w') as fh:
        fh.write('\n'.join(present) + '\n')
    os.replace(tmp, os.path.join(out_dir, 'scans.txt'))
    return out_dir


def row_index(geo):
    """'scan_viewpoint' -> feature-table row, scans in file order, every listed viewpoint (included or not):
    the layout of a table built from the reference's TSV is by viewpoint id, any consistent map works here."""
    row_of, n = {}, 0
    for s, g in geo.items():
        for v, inc in zip(g['ids'], g['included']):
            if inc:
                row_of[s + '_' + v] = n
                n += 1
    return row_of, n
