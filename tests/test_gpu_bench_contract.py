"""GPU: the driver contract of bench.py -- ONE compact JSON line (< 4 KB, strict JSON) with the required keys,
the full object in the extras file beside it -- on a reduced table, and once as the DEFAULT command the driver runs."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONTRACT = ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better',
            'scaling', 'vs_baseline', 'dtype', 'data', 'config', 'roofline', 'cpu_baseline')


def _run(argv, tmp_path):
    extras = str(tmp_path / 'extras.json')
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + argv + ['--extras-out', extras],
                         capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = out.stdout.splitlines()
    assert len(lines) == 1 and lines[0].startswith('{'), out.stdout[:400]      # stdout is the line and nothing else
    assert len(lines[0]) < 4096, len(lines[0])

    def no_constants(name):
        raise AssertionError('non-finite constant %s in the line' % name)
    d = json.loads(lines[0], parse_constant=no_constants)
    return d, json.load(open(extras))


def _check_headline(d, steps, warmup):
    for k in CONTRACT:
        assert k in d, k
    assert d['n_gpus'] == 1 and d['steps'] == steps and d['warmup'] == warmup and d['higher_is_better'] is True
    assert d['scaling'] == 'weak' and d['vs_baseline'] is None and d['dtype'] == 'f32'
    assert d['unit'] == 'agent-steps/s' and d['value'] > 0
    assert abs(d['value'] - 100 * 20 / (d['ms_per_step'] * 1e-3)) < 1e-3 * d['value']
    r = d['roofline']
    for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'kernel', 'launch_us', 'flops_per_launch',
              'bytes_per_launch'):
        assert k in r, k
    assert r['bound'] in ('hbm', 'mfma') and abs(r['frac'] - r['achieved'] / r['peak']) < 1e-9
    assert ' ' not in r['kernel'].split('<')[0]              # the kernel's NAME, no prose
    if 'gemm_nt_split' in r['kernel']:
        # the bf16-split gate product is priced BOTH ways: algorithmic fp32 work against the fp32 MFMA peak (frac), and
        # the 6x bf16 work it executes against the dense bf16 peak (executed.frac)
        e = r['executed']
        assert e['dtype'] == 'bf16' and e['peak'] == 2500.0 and 0 < e['frac'] < 1
        assert abs(e['tflops'] - 6 * r['achieved']) < 1e-4 * e['tflops']      # (the line carries six significant digits)
    assert 0 < r['rollout']['flops_frac'] < 1 and 0 < r['rollout']['hbm_frac'] < 1
    assert r['kernel_time_ms_per_rollout'] < 1.5 * d['ms_per_step']
    c = d['cpu_baseline']
    for k in ('value', 'unit', 'cores', 'kind', 'sample'):
        assert k in c, k
    assert c['kind'] in ('port', 'reference') and c['cores'] >= 1
    assert d['parity_vs_cpu_port']['actions_bit_exact'] is True
    assert d['parity_vs_cpu_port']['max_abs_logit_diff'] < 1e-4
    assert 'workload' in d['config'] and 'model' not in d['config']
    assert d['persistent_launch_faults'] == 0


def test_bench_prints_one_compact_json_line_with_the_contract_keys(tmp_path):
    d, full = _run(['--steps', '2', '--warmup', '1', '--n-viewpoints', '96', '--cpu-reps', '1'], tmp_path)
    _check_headline(d, 2, 1)
    assert d['extras_file']
    # the full object (side file) carries the per-kernel table and the other configs
    for k in CONTRACT:
        assert k in full, k
    assert abs(full['value'] - d['value']) <= 1e-5 * d['value']          # (the line carries six significant digits)
    ks = full['roofline']['kernels']
    assert len(ks) >= 5 and ks[0]['kernel'] in full['roofline']['kernel']
    assert all(ks[i]['share'] >= ks[i + 1]['share'] for i in range(len(ks) - 1))
    for k in ks[:5]:
        assert k['avg_us'] > 0 and k['calls_per_rollout'] > 0
    priced = [k for k in ks if 'mfma_frac' in k]
    assert len(priced) >= 4 and all(0 < k['mfma_frac'] < 1 and 0 < k['hbm_frac'] < 1 for k in priced)
    for extra in ('speaker_decode', 'search_step', 'train_iteration', 'cpu_baseline_all_cores'):
        assert extra in full, extra
    assert 'error' not in full['speaker_decode'], full['speaker_decode']
    assert 'error' not in full['search_step'], full['search_step']


def test_the_default_command_fits_one_driver_line(tmp_path):
    """`python bench.py` exactly as the driver runs it (full 10 567-viewpoint table, every extra)."""
    d, full = _run([], tmp_path)
    _check_headline(d, 20, 5)
    assert 'train_iteration' in d['extras'] and 'speaker_sweep' in d['extras']
    assert 'real_env_full' in full and 'pragmatic_inference' in full
