"""CPU: the line bench.py prints stays a compact, strict-JSON headline whatever the extras hold (round 5's 20 KB line was
not parsed by the driver).  The full object of a real default run (profiles/r06_zzz_bench_default_extras.json) and a
hostile one (NaN / Infinity, kilobytes of prose, a failed extra) go through bench.print_result."""
import importlib.util
import io
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONTRACT = ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
            'vs_baseline', 'dtype', 'data', 'config', 'roofline', 'cpu_baseline')


def _bench():
    argv = sys.argv
    sys.argv = ['bench.py']
    try:
        spec = importlib.util.spec_from_file_location('sf_bench_under_test', os.path.join(ROOT, 'bench.py'))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
    finally:
        sys.argv = argv
    return mod


def _line(bench, full, tmp_path):
    buf = io.StringIO()
    bench.print_result(full, buf, str(tmp_path / 'extras.json'))
    text = buf.getvalue()
    assert text.endswith('\n') and text.count('\n') == 1
    line = text[:-1]
    assert len(line) < 4096

    def no_constants(name):
        raise AssertionError('non-finite constant %s in the line' % name)
    return json.loads(line, parse_constant=no_constants), line, json.load(open(tmp_path / 'extras.json'))


def test_a_real_default_run_prints_a_compact_line(tmp_path):
    bench = _bench()
    full = json.load(open(os.path.join(ROOT, 'profiles', 'r06_zzz_bench_default_extras.json')))
    assert len(json.dumps(full)) > 15000                       # (what round 5 printed)
    d, line, side = _line(bench, full, tmp_path)
    for k in CONTRACT:
        assert k in d, k
    r = d['roofline']
    assert r['frac'] == r['achieved'] / r['peak'] and r['bound'] in ('mfma', 'hbm') and ' ' not in r['kernel'].split('<')[0]
    assert r['executed']['dtype'] == 'bf16' and 0 < r['executed']['frac'] < 1 and 0 < r['mfma_busy_frac'] < 1
    assert d['cpu_baseline']['kind'] == 'port' and d['cpu_baseline']['cores'] == 1
    assert abs(d['value'] - full['value']) <= 1e-5 * full['value']
    assert 'extras' in d and 'parity_speaker_g9' in d and d['parity_vs_cpu_port']['actions_bit_exact'] is True
    assert side['value'] == full['value'] and 'kernels' in side['roofline']        # the side file keeps everything


def test_a_hostile_object_still_prints_a_valid_line(tmp_path):
    bench = _bench()
    full = json.load(open(os.path.join(ROOT, 'profiles', 'r06_zzz_bench_default_extras.json')))
    full['loss'] = float('nan')
    full['roofline']['traffic'] = float('inf')
    full['parity_vs_cpu_port']['loss_abs_diff'] = float('nan')
    full['config']['workload'] = 'x' * 5000
    full['roofline']['kernel'] = 'some_kernel<1, 2> (' + 'prose ' * 400 + ')'
    full['cpu_baseline']['sample'] = 'y' * 3000
    full['speaker_decode'] = dict(error='Z' * 4000)
    full['train_dp'] = dict(value=1.0, error='E' * 5000, strong=dict(value=2.0), health=dict(persistent_launch_faults=[0, 1]))
    d, line, _ = _line(bench, full, tmp_path)
    for k in CONTRACT:
        assert k in d, k
    assert d['roofline']['kernel'] == 'some_kernel<1, 2>'
    assert d['roofline']['traffic'] is None or isinstance(d['roofline']['traffic'], (int, float))
    assert len(d['config']['workload']) <= 260
