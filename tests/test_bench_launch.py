"""`--gpus N` of the three benches starts N ranks (bench_launch.py): the children's environment, the forwarding of
rank 0's output, the propagation of a failing rank's exit code, and that the benches call the launcher before they
import torch or the library.  No GPU: the children are stub scripts."""
import ast
import json
import os
import subprocess
import sys
import textwrap
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench_launch  # noqa: E402


def _stub(tmp_path, body):
    f = tmp_path / 'stub.py'
    f.write_text('import json, os, sys, time\n' + textwrap.dedent(body))
    return str(f)


def _run_launcher(tmp_path, stub, world, extra_env=None, grace_s=1.0):
    """a fresh interpreter that calls launch_ranks on the stub, as `python bench.py --gpus N` does"""
    code = (f'import sys; sys.path.insert(0, {ROOT!r}); import bench_launch; '
            f'sys.exit(bench_launch.launch_ranks({world}, argv=[{stub!r}, "--gpus", "{world}"], grace_s={grace_s}))')
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_PORT')}
    env.update(extra_env or {})
    return subprocess.run([sys.executable, '-c', code], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                          timeout=120)


def test_rank_environment_and_rank0_stdout(tmp_path):
    out = tmp_path / 'ranks'
    out.mkdir()
    stub = _stub(tmp_path, f'''
        keys = ['RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'LOCAL_WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT',
                'HSA_ENABLE_IPC_MODE_LEGACY']
        env = {{k: os.environ.get(k) for k in keys}}
        env['argv'] = sys.argv[1:]
        open(os.path.join({str(out)!r}, env['RANK'] + '.json'), 'w').write(json.dumps(env))
        print(json.dumps({{'line_of_rank': env['RANK']}}))
    ''')
    r = _run_launcher(tmp_path, stub, 4)
    assert r.returncode == 0, r.stderr.decode()
    # only rank 0's JSON line reaches the launcher's stdout
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.strip()]
    assert lines == [json.dumps({'line_of_rank': '0'})]
    envs = [json.loads((out / f'{k}.json').read_text()) for k in range(4)]
    assert [e['RANK'] for e in envs] == ['0', '1', '2', '3']
    assert [e['LOCAL_RANK'] for e in envs] == ['0', '1', '2', '3']
    assert {e['WORLD_SIZE'] for e in envs} == {'4'} and {e['LOCAL_WORLD_SIZE'] for e in envs} == {'4'}
    assert {e['MASTER_ADDR'] for e in envs} == {'127.0.0.1'}
    assert len({e['MASTER_PORT'] for e in envs}) == 1 and int(envs[0]['MASTER_PORT']) > 0
    assert {e['HSA_ENABLE_IPC_MODE_LEGACY'] for e in envs} == {'0'}
    assert all(e['argv'] == ['--gpus', '4'] for e in envs)       # the same command line in every rank


def test_failing_rank_sets_exit_code_and_peers_are_ended(tmp_path):
    stub = _stub(tmp_path, '''
        if os.environ['RANK'] == '2':
            sys.exit(7)
        time.sleep(600)            # a peer waiting in a collective that will never complete
    ''')
    t0 = time.monotonic()
    r = _run_launcher(tmp_path, stub, 3, grace_s=1.0)
    assert r.returncode == 7
    assert time.monotonic() - t0 < 60
    assert b'rank 2 of 3 exited with 7' in r.stderr


def test_rank_killed_by_signal_gives_nonzero(tmp_path):
    stub = _stub(tmp_path, '''
        import signal
        if os.environ['RANK'] == '1':
            os.kill(os.getpid(), signal.SIGKILL)
        print('{}')
    ''')
    r = _run_launcher(tmp_path, stub, 2)
    assert r.returncode == 128 + 9


def test_maybe_launch_is_identity_in_a_rank_and_for_one_gpu(monkeypatch):
    monkeypatch.delenv('WORLD_SIZE', raising=False)
    monkeypatch.delenv('RANK', raising=False)
    assert bench_launch.maybe_launch(1) is None
    monkeypatch.setenv('WORLD_SIZE', '8')
    monkeypatch.setenv('RANK', '3')
    assert bench_launch.is_rank()
    assert bench_launch.maybe_launch(8) is None               # already a rank (torch.distributed.run): no second fan-out


def test_benches_call_the_launcher_before_torch():
    """In each bench's main() `bench_launch.maybe_launch(args.gpus)` comes before the first import of torch / the
    package, and bench_launch itself imports neither: the launcher has made no GPU call when it starts the ranks."""
    for name in ('bench.py', 'bench_corpus.py', 'bench_fit.py'):
        tree = ast.parse(open(os.path.join(ROOT, name)).read())
        main = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == 'main')
        first_call = first_torch = None
        for node in ast.walk(main):
            if isinstance(node, ast.Call) and isinstance(node.func, ast.Attribute) and node.func.attr == 'maybe_launch':
                first_call = node.lineno if first_call is None else min(first_call, node.lineno)
            if isinstance(node, (ast.Import, ast.ImportFrom)):
                mods = [a.name for a in node.names] + [getattr(node, 'module', None) or '']
                if any(m.split('.')[0] in ('torch', 'kwiiyatta_amd') for m in mods):
                    first_torch = node.lineno if first_torch is None else min(first_torch, node.lineno)
        assert first_call is not None, name
        assert first_torch is None or first_call < first_torch, name
        top = [a.name.split('.')[0] for n in tree.body if isinstance(n, ast.Import) for a in n.names]
        assert 'torch' not in top and 'kwiiyatta_amd' not in top, name
    src = open(os.path.join(ROOT, 'bench_launch.py')).read()
    mods = [a.name.split('.')[0] for n in ast.walk(ast.parse(src)) if isinstance(n, ast.Import) for a in n.names]
    assert 'torch' not in mods and 'kwiiyatta_amd' not in mods
    calls = [n.func.attr for n in ast.walk(ast.parse(src)) if isinstance(n, ast.Call) and isinstance(n.func, ast.Attribute)]
    assert not [c for c in calls if c.startswith('exec') or c.startswith('spawn')]      # children, never a replaced process


def test_bench_gpus_flag_reaches_the_launcher(tmp_path):
    """`python bench.py --gpus 2 ...` outside a rank starts two children of the SAME command line: checked with a fake
    interpreter (KWY_BENCH_PYTHON is not a thing -- the launcher uses sys.executable -- so run bench.py's argument
    parsing under a stubbed bench_launch)."""
    code = textwrap.dedent(f'''
        import sys
        sys.path.insert(0, {ROOT!r})
        import bench_launch
        seen = []
        bench_launch.maybe_launch = lambda gpus: (seen.append(gpus), sys.exit(42))[1]
        sys.argv = ['bench.py', '--gpus', '2', '--steps', '1']
        import runpy
        try:
            runpy.run_path({os.path.join(ROOT, 'bench.py')!r}, run_name='__main__')
        except SystemExit as e:
            assert e.code == 42 and seen == [2], (e.code, seen)
            assert 'torch' not in sys.modules, 'torch imported before the launcher ran'
            print('ok')
    ''')
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE')}
    r = subprocess.run([sys.executable, '-c', code], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert r.stdout.decode().strip() == 'ok', r.stderr.decode()
