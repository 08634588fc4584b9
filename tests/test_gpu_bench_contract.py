"""GPU: the driver contract of bench.py -- one JSON line with the required keys -- on a reduced
table (the default run uses the full 10 567-viewpoint table)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_prints_one_json_line_with_the_contract_keys():
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '2', '--warmup', '1',
                          '--n-viewpoints', '96', '--cpu-reps', '1'], capture_output=True, text=True,
                         timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better',
              'scaling', 'vs_baseline', 'dtype', 'data', 'config', 'roofline', 'cpu_baseline'):
        assert k in d, k
    assert d['n_gpus'] == 1 and d['steps'] == 2 and d['warmup'] == 1 and d['higher_is_better'] is True
    assert d['scaling'] == 'weak' and d['vs_baseline'] is None and d['dtype'] == 'f32'
    assert d['unit'] == 'agent-steps/s' and d['value'] > 0
    assert abs(d['value'] - 100 * 20 / (d['ms_per_step'] * 1e-3)) < 1e-3 * d['value']
    r = d['roofline']
    for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic'):
        assert k in r, k
    assert r['bound'] in ('hbm', 'mfma') and abs(r['frac'] - r['achieved'] / r['peak']) < 1e-9
    # the roofline names the top kernel by time of the profiled rollout and carries the table
    ks = r['kernels']
    assert len(ks) >= 5 and ks[0]['kernel'] in r['kernel']
    assert all(ks[i]['share'] >= ks[i + 1]['share'] for i in range(len(ks) - 1))
    for k in ks[:5]:
        assert k['avg_us'] > 0 and k['calls_per_rollout'] > 0
    priced = [k for k in ks if 'mfma_frac' in k]
    assert len(priced) >= 4 and all(0 < k['mfma_frac'] < 1 and 0 < k['hbm_frac'] < 1 for k in priced)
    assert 0 < r['rollout']['flops_frac'] < 1 and 0 < r['rollout']['hbm_frac'] < 1
    # kernel time of a rollout (sum of event pairs) cannot exceed its wall time by more than noise
    assert r['kernel_time_ms_per_rollout'] < 1.5 * d['ms_per_step']
    for extra in ('speaker_decode', 'search_step', 'train_iteration', 'cpu_baseline_all_cores'):
        assert extra in d, extra
    assert 'error' not in d['speaker_decode'], d['speaker_decode']
    assert 'error' not in d['search_step'], d['search_step']
    c = d['cpu_baseline']
    for k in ('value', 'unit', 'cores', 'kind', 'sample'):
        assert k in c, k
    assert c['kind'] in ('port', 'reference') and c['cores'] >= 1
    assert d['parity_vs_cpu_port']['actions_bit_exact'] is True
    assert 'workload' in d['config'] and 'model' not in d['config']
