"""HBM-resident image-feature store and index-form observations.

Replaces the reference's per-step host work -- dict lookup of a 36x2048 block per
sample (env.py:380-383), concatenation with the location embedding (env.py:773),
construction of the candidate-action embeddings (env.py:60-75), np.stack and the
H2D copy (env.py:330-332, follower.py:291-320) -- by ONE [n_viewpoints,36,2048]
fp32 tensor that lives in HBM (3.1 GB for the full R2R set; 288 GB available) plus a
36x36x128 location table; per step only a few int32 indices per sample travel.
"""
import base64
import csv
import ctypes as C
import json
import sys

import numpy as np
import torch

from . import _lib
from ._lib import call
from .runtime import ptr, stream

NUM_VIEWS = 36          # env.py:285
MEAN_POOLED_DIM = 2048  # env.py:286
LOC_DIM = 128           # env.py:62, 87
ANGLE_INC = np.pi / 6.0  # env.py:57


def build_loc_table(n_views=NUM_VIEWS, loc=LOC_DIM):
    """[n_views (agent view index), n_views (absolute view), loc]: sin/cos of each view's
    heading / elevation relative to the agent's view (env.py:78-101, 12 headings x 3
    elevations), each repeated loc/4 times."""
    g = loc // 4
    tab = np.zeros((n_views, n_views, loc), np.float32)
    for view in range(n_views):
        for abs_view in range(n_views):
            rel = (abs_view - view) % 12 + (abs_view // 12) * 12
            h = (rel % 12) * ANGLE_INC
            e = (rel // 12 - 1) * ANGLE_INC
            tab[view, abs_view, 0:g] = np.sin(h)
            tab[view, abs_view, g:2 * g] = np.cos(h)
            tab[view, abs_view, 2 * g:3 * g] = np.sin(e)
            tab[view, abs_view, 3 * g:] = np.cos(e)
    return tab


def cand_sincos(rel_heading, rel_elevation):
    """[..., 4] fp32 = sin h, cos h, sin e, cos e, evaluated in float64 like the
    reference does on the simulator's python floats (env.py:69-74)."""
    h = np.asarray(rel_heading, np.float64)
    e = np.asarray(rel_elevation, np.float64)
    return np.stack((np.sin(h), np.cos(h), np.sin(e), np.cos(e)), axis=-1).astype(np.float32)


def tsv_to_bin(tsv_path, bin_path):
    """One-shot converter (N4): the reference's ResNet TSV -> flat little-endian fp32 file
    [n][36][2048] + '<bin>.json' {ids, shape}.  Streams row by row (the full table is 3.1 GB)."""
    csv.field_size_limit(sys.maxsize)
    names = ['scanId', 'viewpointId', 'image_w', 'image_h', 'vfov', 'features']
    ids = []
    with open(tsv_path, 'rt') as f, open(bin_path, 'wb') as out:
        for item in csv.DictReader(f, delimiter='\t', fieldnames=names):
            buf = base64.b64decode(item['features'])
            if len(buf) != NUM_VIEWS * MEAN_POOLED_DIM * 4:
                raise ValueError('row %s_%s: %d feature bytes' % (item['scanId'], item['viewpointId'], len(buf)))
            out.write(buf)
            ids.append(item['scanId'] + '_' + item['viewpointId'])
    with open(bin_path + '.json', 'w') as f:
        json.dump({'ids': ids, 'shape': [len(ids), NUM_VIEWS, MEAN_POOLED_DIM]}, f)
    return len(ids)


class FeatureStore:
    """The feature table in HBM + viewpoint-id index."""

    def __init__(self, table, ids=None, device='cuda', loc=LOC_DIM):
        if isinstance(table, np.ndarray):
            table = torch.from_numpy(np.ascontiguousarray(table, np.float32))
        self.table = table.to(device=device, dtype=torch.float32).contiguous()
        self.n, self.V, self.IMG = self.table.shape
        self.LOC = loc
        self.F = self.IMG + loc
        self.device = self.table.device
        self.loc_table = torch.from_numpy(build_loc_table(self.V, loc)).to(self.device)
        self.index = {k: i for i, k in enumerate(ids)} if ids is not None else None

    @classmethod
    def from_tsv(cls, path, device='cuda'):
        """Reads the reference's ResNet-152 TSV (scanId, viewpointId, image_w, image_h, vfov,
        base64 fp32 36x2048; env.py:359-370, scripts/precompute_img_features.py:31)."""
        csv.field_size_limit(sys.maxsize)
        names = ['scanId', 'viewpointId', 'image_w', 'image_h', 'vfov', 'features']
        ids, rows = [], []
        with open(path, 'rt') as f:
            for item in csv.DictReader(f, delimiter='\t', fieldnames=names):
                ids.append(item['scanId'] + '_' + item['viewpointId'])       # env.py:377-378
                buf = base64.b64decode(item['features'])
                rows.append(np.frombuffer(buf, np.float32).reshape(NUM_VIEWS, MEAN_POOLED_DIM))
        return cls(np.stack(rows), ids, device)

    @classmethod
    def from_bin(cls, path, device='cuda', chunk_rows=512):
        """Flat table written by `tsv_to_bin` (path + '.json' holds the ids and shape): the file is
        memory-mapped and uploaded in chunks straight into ONE preallocated HBM tensor, so start-up
        costs a sequential read instead of minutes of TSV / base64 parsing."""
        with open(path + '.json') as f:
            meta = json.load(f)
        n, V, IMG = meta['shape']
        mm = np.memmap(path, dtype=np.float32, mode='r', shape=(n, V, IMG))
        table = torch.empty(n, V, IMG, dtype=torch.float32, device=device)
        for r0 in range(0, n, chunk_rows):
            r1 = min(n, r0 + chunk_rows)
            table[r0:r1].copy_(torch.from_numpy(np.array(mm[r0:r1])))
        return cls(table, meta['ids'], device)

    def row(self, scan_id, viewpoint_id):
        return self.index[scan_id + '_' + viewpoint_id]

    # ---- pointer structs for the C ABI (tensors must stay alive while the call is enqueued) -----
    def pano(self, vp, view):
        return _lib.Pano(None, self.table.data_ptr(), self.loc_table.data_ptr(), vp.data_ptr(),
                         view.data_ptr(), self.V, self.IMG, self.LOC)

    def cands(self, vp, cand_view, sincos, a_num, A):
        return _lib.Cands(None, self.table.data_ptr(), vp.data_ptr(), cand_view.data_ptr(),
                          sincos.data_ptr(), a_num.data_ptr(), A, self.V, self.IMG, self.LOC)

    # ---- dense materialisation (what the reference agent feeds the modules) ---------------------
    def gather_panorama(self, vp, view):
        """[B,V,F] = features || location embedding (follower.py:291-298)."""
        B = vp.shape[0]
        out = torch.empty(B, self.V, self.F, device=self.device, dtype=torch.float32)
        p = self.pano(vp, view)
        call('sf_gather_panorama', C.byref(p), B, ptr(out), stream())
        return out

    def gather_candidates(self, vp, cand_view, sincos, a_num):
        """all_u_t [B,A,F], is_valid [B,A] (follower.py:300-320)."""
        B, A = cand_view.shape
        all_u = torch.empty(B, A, self.F, device=self.device, dtype=torch.float32)
        is_valid = torch.empty(B, A, device=self.device, dtype=torch.float32)
        c = self.cands(vp, cand_view, sincos, a_num, A)
        call('sf_gather_candidates', C.byref(c), B, ptr(all_u), ptr(is_valid), stream())
        return all_u, is_valid

    def gather_actions(self, vp, act_view, sincos, act):
        """[B,F] embeddings of one chosen action per sample (speaker.py:104); act <= 0 or
        vp < 0 gives zeros.  act_view/sincos are [B,1]-shaped candidate lists."""
        B = vp.shape[0]
        out = torch.empty(B, self.F, device=self.device, dtype=torch.float32)
        a_num = torch.full((B,), 2, dtype=torch.int32, device=self.device)
        cv = torch.stack((torch.zeros_like(act_view), act_view), 1).contiguous()
        sc = torch.stack((torch.zeros_like(sincos), sincos), 1).contiguous()
        c = self.cands(vp, cv, sc, a_num, 2)
        call('sf_gather_actions', C.byref(c), B, ptr(act), ptr(out), stream())
        return out
