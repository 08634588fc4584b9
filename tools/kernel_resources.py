"""Register / LDS / spill table of every kernel of one csrc/*.hip file (hipcc -Rpass-analysis=kernel-resource-usage).
    python tools/kernel_resources.py sf_persist.hip [name-filter]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'speaker_follower_amd', 'csrc')
src = os.path.join(CSRC, sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 else ''
res = subprocess.run(['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-I' + CSRC, '-c', src, '--cuda-device-only',
                      '-o', '/dev/null', '-Rpass-analysis=kernel-resource-usage'], capture_output=True, text=True)
rows, cur = [], None
for line in res.stderr.splitlines():
    m = re.search(r'remark: +([^:]+): +(.*?) *\[-Rpass', line)
    if not m:
        continue
    k, v = m.group(1).strip(), m.group(2).strip()
    if k == 'Function Name':
        cur = {'name': subprocess.run(['c++filt', v], capture_output=True, text=True).stdout.strip()[:90]}
        rows.append(cur)
    elif cur is not None:
        cur[k] = v
for r in rows:
    if flt in r['name']:
        print('%-90s VGPR %3s AGPR %3s spill %3s SGPR %3s scratch %3s LDS %6s occ %s' % (
            r['name'], r.get('VGPRs'), r.get('AGPRs'), r.get('VGPRs Spill'), r.get('TotalSGPRs'),
            r.get('ScratchSize [bytes/lane]'), r.get('LDS Size [bytes/block]'), r.get('Occupancy [waves/SIMD]')))
