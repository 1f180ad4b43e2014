"""`python bench.py --gpus N` must create N ranks by itself (the driver's plain command has no launcher in front), and
must refuse to report n_gpus for a world that does not exist.  CPU, gloo, the kernel-free --stub step: the launcher,
the rendezvous, the shard -> all_gather -> max-over-ranks skeleton and the JSON contract."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, 'bench.py')


def _run(args, env_extra=None, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, BENCH] + args, env=env, capture_output=True, text=True, timeout=timeout)


def _json_line(out):
    lines = [l for l in out.splitlines() if l.startswith('{')]
    assert len(lines) == 1, out
    return json.loads(lines[0])


def test_plain_invocation_spawns_two_ranks():
    r = _run(['--gpus', '2', '--steps', '3', '--warmup', '1', '--stub', '--backend', 'gloo'])
    assert r.returncode == 0, r.stderr[-2000:]
    line = _json_line(r.stdout)
    assert line['n_gpus'] == 2 and line['ranks_seen'] == 2
    assert line['gathered_in_item_order'] is True
    assert line['allgather_payload_bytes_per_rank'] == 16 * 5 * 8 * 8 * 4
    assert line['steps'] == 3 and line['warmup'] == 1 and line['scaling'] == 'weak'


def test_single_rank_stub():
    line = _json_line(_run(['--stub', '--steps', '2', '--warmup', '0']).stdout)
    assert line['n_gpus'] == 1 and line['ranks_seen'] == 1 and line['allgather_payload_bytes_per_rank'] == 0


def test_world_mismatch_is_refused():
    """A launcher that started 1 rank while --gpus says 2 must not produce a line claiming 2 GPUs."""
    r = _run(['--gpus', '2', '--stub', '--backend', 'gloo'], {'WORLD_SIZE': '1', 'RANK': '0', 'LOCAL_RANK': '0'})
    assert r.returncode != 0 and 'refusing' in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith('{')]


def test_eight_ranks_stub():
    """The driver's 8-GPU command line, on CPU: `bench.py --gpus 8` must bring up 8 ranks, shard 8 x 16 objects and gather
    them in item order (gloo stands in for RCCL; the measured path refuses gloo)."""
    r = _run(['--gpus', '8', '--steps', '2', '--warmup', '1', '--stub', '--backend', 'gloo'], timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = _json_line(r.stdout)
    assert line['n_gpus'] == 8 and line['ranks_seen'] == 8 and line['gathered_in_item_order'] is True
    assert line['allgather_payload_bytes_per_rank'] == 16 * 5 * 8 * 8 * 4
