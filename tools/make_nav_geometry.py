#!/usr/bin/env python3
"""Build container only: the navigation geometry of all 90 Matterport3D scans as ONE compact table file.

Reads the connectivity DATA files under /root/reference/connectivity (image_id, pose translation, included,
unobstructed -- the four fields the navigation-only simulator and the shortest-path planner use) and writes
speaker_follower_amd/data/r2r_connectivity.npz (ids, included flags, float64 positions, bit-packed
unobstructed matrices; < 1 MB).  speaker_follower_amd/nav_data.py turns it back into a connectivity
directory on any machine, so the full 10 567-viewpoint environment exists on the GPU box too.
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(os.environ.get('SF_REFERENCE', '/root/reference'), 'connectivity')
OUT = os.path.join(ROOT, 'speaker_follower_amd', 'data', 'r2r_connectivity.npz')


def main():
    scans = [s.strip() for s in open(os.path.join(REF, 'scans.txt')) if s.strip()]
    ids, included, pos, bits, n_of = [], [], [], [], []
    for s in scans:
        data = json.load(open(os.path.join(REF, s + '_connectivity.json')))
        n = len(data)
        n_of.append(n)
        un = np.zeros((n, n), bool)
        for i, d in enumerate(data):
            ids.append(d['image_id'])
            included.append(bool(d['included']))
            pos.append([d['pose'][3], d['pose'][7], d['pose'][11]])
            assert len(d['unobstructed']) == n
            un[i] = d['unobstructed']
        bits.append(np.packbits(un.reshape(-1)))
    np.savez_compressed(
        OUT, scans=np.array(scans), n=np.array(n_of, np.int32), ids=np.array(ids, dtype='S32'),
        included=np.array(included, bool), pos=np.array(pos, np.float64),
        unobstructed_bits=np.concatenate(bits), bits_n=np.array([len(b) for b in bits], np.int64))
    print('%d scans, %d viewpoints (%d included) -> %s (%d bytes)'
          % (len(scans), len(ids), sum(included), OUT, os.path.getsize(OUT)))


if __name__ == '__main__':
    sys.exit(main())
