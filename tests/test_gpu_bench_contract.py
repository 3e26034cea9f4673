"""bench.py prints ONE JSON line with the keys the driver reads (small volume, no CPU baseline leg)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_json_line_contract():
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--volume', '192', '--steps', '1', '--warmup', '1',
                          '--no-cpu-baseline'], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1
    j = json.loads(lines[0])
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
              'vs_baseline', 'dtype', 'data', 'config', 'roofline'):
        assert k in j, k
    assert j['n_gpus'] == 1 and j['steps'] == 1 and j['warmup'] == 1 and j['higher_is_better'] is True
    assert j['vs_baseline'] is None and j['unit'] == 'patches/s' and j['data'] == 'synthetic' and j['dtype'] == 'f16'
    assert 'workload' in j['config'] and 'model' not in j['config']
    r = j['roofline']
    for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic'):
        assert k in r, k
    assert r['bound'] in ('hbm', 'mfma') and abs(r['frac'] - r['achieved'] / r['peak']) < 1e-3
    assert j['value'] > 0 and abs(j['value'] - 1000.0 * _patches(j) / j['ms_per_step']) / j['value'] < 1e-3


def _patches(j):
    import re
    return int(re.search(r'(\d+) patches/volume', j['config']['workload']).group(1))
