"""CPU: `bench.py --gpus N` is its own launcher (the reference has no multi-GPU path --
the slot is between backward() and step() at follower.py:1014-1018).  The parent must build one
child per GPU with the torch.distributed environment, before anything touches a GPU."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench          # noqa: E402


def test_launch_plan_has_one_child_per_gpu_with_rendezvous_env():
    argv = ['--gpus', '4', '--steps', '3', '--warmup', '1']
    plan = bench.launch_plan(4, argv, env={'PATH': '/usr/bin', 'KEEP': 'me'}, port=29777)
    assert len(plan) == 4
    for r, (cmd, env) in enumerate(plan):
        assert cmd[0] == sys.executable and cmd[1] == os.path.join(ROOT, 'bench.py')
        assert cmd[2:] == argv                               # the children see the same flags
        assert env['RANK'] == env['LOCAL_RANK'] == str(r)
        assert env['WORLD_SIZE'] == env['LOCAL_WORLD_SIZE'] == '4'
        assert env['MASTER_ADDR'] == '127.0.0.1' and env['MASTER_PORT'] == '29777'
        assert env['HSA_ENABLE_IPC_MODE_LEGACY'] == '0'      # dmabuf IPC for RCCL
        assert env['KEEP'] == 'me'
    assert len({id(e) for _, e in plan}) == 4                # separate dicts


def test_launcher_is_chosen_only_without_world_size(monkeypatch):
    calls = []
    monkeypatch.setattr(bench, 'run_launcher', lambda args, argv: calls.append((args.gpus, list(argv))) or 0)
    monkeypatch.delenv('WORLD_SIZE', raising=False)
    try:
        bench.main(['--gpus', '2', '--steps', '1'])
    except SystemExit as e:
        assert e.code == 0
    assert calls == [(2, ['--gpus', '2', '--steps', '1'])]


def test_plan_prints_the_eight_rank_launch_without_touching_a_gpu(capsys, monkeypatch):
    """`bench.py --gpus 8 --plan`: the child commands and their rendezvous environment as JSON -- what the first
    8-GPU run will start, reviewable on a box with no GPU at all."""
    import json
    monkeypatch.delenv('WORLD_SIZE', raising=False)
    monkeypatch.setattr(bench, 'run_launcher', lambda *a: (_ for _ in ()).throw(AssertionError('must not launch')))
    bench.main(['--gpus', '8', '--plan', '--steps', '5'])
    d = json.loads(capsys.readouterr().out)
    assert d['launcher'] == 'self' and d['collective'] == 'RCCL over xGMI' and len(d['ranks']) == 8
    ports = {r['env']['MASTER_PORT'] for r in d['ranks']}
    assert len(ports) == 1
    for i, r in enumerate(d['ranks']):
        assert r['cmd'][1].endswith('bench.py') and '--plan' not in r['cmd'] and r['cmd'][-2:] == ['--steps', '5']
        assert r['env']['RANK'] == r['env']['LOCAL_RANK'] == str(i) and r['env']['WORLD_SIZE'] == '8'
        assert r['env']['MASTER_ADDR'] == '127.0.0.1' and r['env']['HSA_ENABLE_IPC_MODE_LEGACY'] == '0'


def test_launcher_returns_first_failure_and_stops_the_rest(monkeypatch):
    """Children are plain processes: a failing rank ends the run with its exit code and the other
    ranks are terminated by PID (never by pattern)."""
    script = ("import os, sys, time\n"
              "r = int(os.environ['RANK'])\n"
              "sys.exit(7) if r == 1 else time.sleep(60)\n")
    monkeypatch.setattr(bench, 'launch_plan',
                        lambda n, argv, env=None, port=None: [
                            ([sys.executable, '-c', script], dict(os.environ, RANK=str(r))) for r in range(n)])
    args = bench.parse(['--gpus', '3'])
    import time
    t0 = time.time()
    assert bench.run_launcher(args, ['--gpus', '3']) == 7
    assert time.time() - t0 < 30


def test_kernel_work_prices_the_gate_product_like_survey_8d():
    fl, by, _ = bench.kernel_work('sf::gemm_nt_tiled_kernel<7>(sf::NtArgs)', 100, 20, 80, 5.0, (512, 2176, 256, 36))
    assert fl == 2.0 * 100 * (2 * 2176 + 512) * 2048          # 1.992 GFLOP (19.92 MFLOP / sample)
    assert abs(by - 42.6e6) < 0.1e6
    fl, by, _ = bench.kernel_work('lstm_step_wide_kernel<2>', 100, 20, 80, 5.0, (512, 2176, 256, 36))
    assert fl == 2.0 * 100 * 512 * 2048


def test_plan_shows_the_strong_scaling_mode(capsys):
    """`--scaling strong`: ONE global batch of 100 (train.py:28) split over the ranks with dp.shard_rows -- the plan says
    how many rows each rank takes; the default stays weak (100 rows per GPU)."""
    import json
    from speaker_follower_amd import dp
    bench.main(['--gpus', '8', '--plan', '--scaling', 'strong'])
    plan = json.loads(capsys.readouterr().out)
    assert plan['scaling'] == 'strong' and plan['global_batch'] == 100
    assert plan['rows_per_rank'] == [13, 13, 13, 13, 12, 12, 12, 12]
    assert plan['rows_per_rank'] == [(lambda s: s.stop - s.start)(dp.shard_rows(100, r, 8)) for r in range(8)]
    assert all(r['cmd'][-2:] == ['--scaling', 'strong'] for r in plan['ranks'])
    bench.main(['--gpus', '8', '--plan'])
    plan = json.loads(capsys.readouterr().out)
    assert plan['scaling'] == 'weak' and plan['global_batch'] == 800 and plan['rows_per_rank'] == [100] * 8
