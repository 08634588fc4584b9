#!/usr/bin/env python3
"""N1 over ALL scans (BUILD container only): runs the "Navigable Locations" walk of the reference's
Catch test (src/test/main.cpp:169-299, restated as tests/test_mattersim_nav.py::walk_scan) with this
repo's navigation-only MatterSim over every connectivity graph under /root/reference/connectivity
(90 scans; 8 of them are committed as fixtures) and writes the per-scan record of the walk -- only
reached if every assert of the walk held -- to tests/golden/n1_nav_walk_all_scans.json.

    python tests/golden/make_nav_summary.py
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
REF_CONN = os.path.join(os.environ.get('SF_REFERENCE', '/root/reference'), 'connectivity')

import test_mattersim_nav as T          # noqa: E402


def main():
    from speaker_follower_amd.build import build_sim
    build_sim()
    import speaker_follower_amd.sim as sim_mod
    MatterSim = sim_mod.load()
    scans = open(os.path.join(REF_CONN, 'scans.txt')).read().split()
    out = {}
    for scan in scans:
        sim = MatterSim.Simulator()
        sim.setCameraResolution(20, 20)
        sim.setCameraVFOV(T.rad(90))
        sim.setRenderingEnabled(False)
        sim.setDiscretizedViewingAngles(False)
        sim.setNavGraphPath(REF_CONN)
        sim.setSeed(1)
        sim.init()
        out[scan] = T.walk_scan(sim, scan, REF_CONN)
    summary = dict(n_scans=len(out), all_asserts_held=True,
                   total_checks=sum(r['checks'] for r in out.values()),
                   total_viewpoints=sum(r['viewpoints'] for r in out.values()), scans=out)
    path = os.path.join(HERE, 'n1_nav_walk_all_scans.json')
    with open(path, 'w') as f:
        json.dump(summary, f, indent=0, sort_keys=True)
    print('%d scans, %d viewpoints, %d membership checks, all asserts held -> %s (%d bytes)'
          % (summary['n_scans'], summary['total_viewpoints'], summary['total_checks'], path, os.path.getsize(path)))


if __name__ == '__main__':
    main()
