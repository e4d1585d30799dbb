"""bench.py end to end on the GPU at a reduced stream count: one JSON line with the contract fields, a positive
throughput, the roofline and CPU-baseline objects, and the queued / threaded pipeline shutting down cleanly."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_bench_runs_and_prints_one_contract_line():
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--streams', '64', '--steps', '4', '--warmup', '24'],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    assert d['n_gpus'] == 1 and d['steps'] == 4 and d['warmup'] == 24 and d['value'] > 0
    assert d['config']['streams_per_gpu'] == 64 and d['msckf_in_step'] is True
    assert d['roofline']['bound'] == 'hbm' and 0 < d['roofline']['frac'] < 1 and d['roofline']['traffic'] > 0
    assert d['cpu_baseline']['kind'] == 'port' and d['cpu_baseline']['cores'] == 1 and d['cpu_baseline']['value'] > 0
    assert abs(d['value'] - 64 / (d['ms_per_step'] * 1e-3)) / d['value'] < 1e-6
