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
    # round 4: the shader clock the device held during one extra step (fnn_clock_probe_*: one sleeping wave, s_memtime over
    # s_memrealtime)
    assert 0.5 < r['clock_ghz'] < 3.2 and r['clock_sampled_s'] > 0
    assert abs(r['frac_at_clock'] - r['achieved'] / (r['peak'] * r['clock_ghz'] / 2.4)) < 2e-3


def test_clock_probe_ends_by_itself_and_reads_a_plausible_clock():
    """fnn_clock_probe_start / _stop (include/fnn.h): the probe stops after max_seconds of the 100 MHz counter without a
    flag from the host, and on request earlier; an idle device reads its idle or boost clock."""
    import time
    from fast_nnunet_amd import capi
    h = capi.clock_probe_start(0, 0.05)
    time.sleep(0.2)
    ghz, sec = capi.clock_probe_stop(h)
    assert 0.045 < sec < 0.08 and 0.05 < ghz < 3.5, (ghz, sec)
    h = capi.clock_probe_start(0, 20.0)
    t0 = time.perf_counter()
    ghz, sec = capi.clock_probe_stop(h)                          # the flag in mapped host memory ends it
    assert time.perf_counter() - t0 < 5.0 and sec < 5.0 and 0.05 < ghz < 3.5, (ghz, sec)


def _patches(j):
    import re
    return int(re.search(r'(\d+) patches/volume', j['config']['workload']).group(1))


def test_bench_force_sharded_single_rank_runs_the_multi_gpu_path():
    """`--gpus 1 --force-sharded`: the rank-sharded code path (RCCL process group, patch-activation exchange, the owner's box
    formed by fnn_gather_box) on one GPU - the line the driver would get from every rank 0 at N > 1.  Round 4: the timed
    step is the N = 1 step sharded (fp16 logits of the owned box resident in HBM; the metric string says so), the
    assembled-labels step is a named extra key, and the line carries rank 0's roofline block."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--volume', '192', '--steps', '1', '--warmup', '1',
                          '--gpus', '1', '--force-sharded'], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j['n_gpus'] == 1 and 'patch-sharded x1' in j['config']['parallelism']
    assert 'nothing assembled' in j['config']['step_output'] and 'fp16 logits resident in HBM' in j['metric']
    # (single-step timings of a 7 ms step on a shared box: the same order of magnitude is all that can be asserted)
    assert j['value'] > 0 and 0 < j['ms_per_step_labels_assembled'] <= j['ms_per_step'] * 3 + 50
    ph = j['phases_profiled_step']
    assert ph['rccl_ranks'] == 1 and ph['mode'] == 'gather' and ph['gather_box_ms'] > 0 and ph['per_rank']['n_interior_patches'][0] > 0
    r = j['roofline']
    assert r['bound'] == 'mfma' and r['achieved'] > 0 and abs(r['frac'] - r['achieved'] / r['peak']) < 1e-3 and r['traffic'] is None
    assert 'cpu_baseline' not in j


def test_bench_fold_ensemble_line():
    """`--folds 2`: BASELINE configs[3] is a fold ENSEMBLE (predict_from_raw_data.py:483-500); the folds stay resident and
    one step is one volume through all of them."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--volume', '192', '--steps', '1', '--warmup', '1',
                          '--workload', 'iso128_r2', '--folds', '2', '--no-cpu-baseline'], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    j = json.loads([l for l in out.stdout.splitlines() if l.startswith('{')][0])
    assert j['config']['folds'] == 2 and abs(j['sec_per_volume_per_fold'] * 2 - j['sec_per_volume']) < 2e-4
    assert abs(j['value'] - 1000.0 * 2 * _patches(j) / j['ms_per_step']) / j['value'] < 1e-3


def test_bench_refuses_a_world_size_that_contradicts_gpus():
    env = dict(os.environ, WORLD_SIZE='1', RANK='0', LOCAL_RANK='0')
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1'], capture_output=True,
                         text=True, timeout=300, cwd=ROOT, env=env)
    assert out.returncode != 0 and 'WORLD_SIZE=1' in (out.stderr + out.stdout)
