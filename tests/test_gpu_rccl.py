"""GPU: the RCCL path executes.  One rank -- the box has one GPU -- but real collectives on the real
device: tests/rccl_worker.py runs the data-parallel training iteration with `init_process_group('nccl')`, the bucketed
async all-reduces launched from the backward and the count-table all-reduce, and compares it bit for bit with the same
iteration without a process group.  And `bench.py --gpus 1 --force-collectives` puts `train_dp` into the bench line."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env():
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    return env


def test_rccl_one_rank_training_iteration_is_bit_identical_to_no_collectives():
    res = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'rccl_worker.py'), 'nccl'], env=_env(),
                         capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    line = [l for l in res.stdout.splitlines() if l.startswith('RCCL_WORKER ')][-1]
    out = json.loads(line[len('RCCL_WORKER '):])
    print(out)
    assert out['backend'] == 'nccl' and out['world'] == 1
    assert out['finite'] and out['grad_abs_max'] > 0
    assert out['losses_equal'] and out['grads_bit_identical'] and out['weights_bit_identical'], out
    assert out['counts_identity'] and out['buckets'] == 3
    # run(backward=True) with a forced persistent-launch fault under the nccl group (advisor, round 4): flag reduced with
    # MAX, buckets aborted and re-armed, the re-issued iteration's gradients equal an undisturbed iteration's
    f = out['fault']
    assert f['fallbacks'] == [0, 1] and f['finite'], f
    assert abs(f['losses'][0] - f['losses'][1]) <= 1e-4 * abs(f['losses'][0]), f
    assert f['grad_rel_diff'] <= 1e-4, f


def test_bench_force_collectives_reports_train_dp(tmp_path):
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--backend', 'nccl', '--force-collectives',
           '--steps', '3', '--warmup', '2', '--n-viewpoints', '512', '--no-extras', '--no-cpu-baseline',
           '--extras-out', str(tmp_path / 'x.json')]
    res = subprocess.run(cmd, env=_env(), capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    out = json.loads([l for l in res.stdout.splitlines() if l.startswith('{')][-1])
    assert out['train_dp']['ms_per_iteration'] > 0 and out['train_dp']['allreduce_bytes'] > 50e6      # the compact line
    td = json.load(open(tmp_path / 'x.json'))['train_dp']
    assert 'forced_collectives' in td and td['allreduce_bytes'] > 50e6
    assert td['allreduce_total_ms_blocking'] > 0 and td['ms_per_iteration'] > 0
    assert out['persistent_launch_faults'] == 0
    print({k: td[k] for k in ('ms_per_iteration', 'allreduce_total_ms_blocking', 'allreduce_exposed_ms_overlapped',
                              'ms_per_iteration_no_exchange')})
