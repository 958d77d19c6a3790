"""`python bench.py --gpus N` starts N ranks by itself: without a launcher around it the
script re-executes under torch.distributed.run, one rank per device (backend nccl = RCCL on a GPU node).
Here, without a GPU: the launcher + rendezvous + max-over-ranks plumbing with ``--dry-run`` over gloo, and
the loud failure when N GPUs are asked for and absent."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, **env):
    e = dict(os.environ)
    e.pop('WORLD_SIZE', None)
    e.pop('RANK', None)
    e.update(env)
    return subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + args, env=e, stdout=subprocess.PIPE,
                          stderr=subprocess.PIPE, universal_newlines=True, timeout=240)


def test_gpus_2_spawns_two_ranks_and_reports_n_gpus_2():
    r = _run(['--gpus', '2', '--steps', '2', '--warmup', '1', '--dry-run'], EGONET_AMD_DIST_BACKEND='gloo')
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1                                   # rank 0 alone prints
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['ranks'] == 2 and out['steps'] == 2 and out['dry_run'] is True


def test_gpus_1_needs_no_launcher():
    r = _run(['--gpus', '1', '--steps', '1', '--dry-run'])
    assert r.returncode == 0, r.stderr[-2000:]
    assert json.loads(r.stdout.strip().splitlines()[-1])['n_gpus'] == 1


def test_more_gpus_than_visible_fails_loudly():
    import torch
    if torch.cuda.device_count() >= 8:
        return
    r = _run(['--gpus', '8', '--steps', '1'])
    assert r.returncode == 2 and 'only %d GPU(s) are visible' % torch.cuda.device_count() in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith('{')]       # no bench line under a false n_gpus


def test_help_prints_every_option():
    """argparse expands `%` in help strings: a literal per-cent sign there made `--help` raise (found in round 4)."""
    r = _run(['--help'])
    assert r.returncode == 0, r.stderr[-2000:]
    for opt in ('--gpus', '--steps', '--warmup', '--batch', '--pipelined', '--live-traffic', '--profile-json'):
        assert opt in r.stdout
