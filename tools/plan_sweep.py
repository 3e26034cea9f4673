#!/usr/bin/env python3
"""Plan sweep: the engine on realistic nnU-Net configurations off the BASELINE shapes.

    python tools/plan_sweep.py [--plans tools/plans/a.json ...] [--out profiles/r06_plan_sweep] [--steps 3]

Runs `bench.py --plan <file>` for every plan file (one process per plan), collects the JSON lines and writes
  <out>.json   every line as bench.py printed it (with its per-layer table)
  <out>.txt    per plan: patches/s, conv-family fraction of the MFMA peak, and per layer the kernel the launch rules
               picked with its microseconds per forward, algorithmic TFLOP/s and TB/s; then the list of CLIFFS:
               layers that take > 5 % of their forward and run below 10 % of the MFMA peak - or, for layers whose
               arithmetic intensity puts them on the HBM side (16-channel tensors), below 2.5 TB/s - and every plan that
               reached the generic fallback kernel.
The topologies come from bench.py's restatement of the reference's planning rule (network_topology.py:30-108), which
tests/test_host_cpu.py pins to the reference-made tests/golden/topology.json.
"""
import argparse
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PEAK = 2500.0


def is_cliff(r):
    """r: a row of bench.py's per-layer table.  HBM-side layers: at most 16 channels on the wide side of the layer."""
    if r['share'] <= 0.05 or not r['us']:
        return None
    thin = min(r['cin'], r['cout']) <= 16 and r['type'].startswith(('conv', 'stem', 'input'))
    if thin or r['type'].startswith('tconv'):
        if r['tbps'] is not None and r['tbps'] < 2.5 and (r['tflops'] or 0) < 0.10 * PEAK:
            pad = f" (with the channels padded to 16 as stored: {r['tbps_padded']:.2f} TB/s)" if r.get('tbps_padded') else ''
            return f"{r['tbps']:.2f} TB/s < 2.5{pad}"
        return None
    if r['type'].startswith('conv') and r['tflops'] is not None and r['tflops'] < 0.10 * PEAK:
        return f"{r['tflops']:.0f} TFLOP/s < {0.10 * PEAK:.0f}"
    return None


def fmt_plan(j):
    out = []
    c = j['config']
    rf = j.get('roofline', {})
    out.append(f"== {c['workload']}")
    out.append(f"   {j['value']:.1f} patches/s, {j['ms_per_step']:.1f} ms per volume, batch {c['patches_per_forward']}, "
               f"{c['gflop_per_patch']} GFLOP per patch, conv family {rf.get('achieved')} TFLOP/s = {rf.get('frac')} of peak, "
               f"whole net {rf.get('whole_net_tflops')} TFLOP/s; hidden by batches in flight "
               f"{rf.get('schedules', {}).get('hidden_by_batches_in_flight')}")
    ts = rf.get('time_share_ms', {})
    out.append('   profiled step, ms: ' + ', '.join(f'{k} {v}' for k, v in ts.items()))
    out.append(f"   {'layer':>5} {'type':<14} {'cin':>4} {'cout':>4} {'k':<6} {'s':<6} {'out':<12} {'us':>8} {'share':>6} {'TFLOP/s':>8} {'TB/s':>6}  kernel")
    for r in j.get('layers', []):
        mark = '  <-- CLIFF: ' + is_cliff(r) if is_cliff(r) else ''
        padded = f" [{r['tbps_padded']} TB/s as stored]" if r.get('tbps_padded') else ''
        out.append(f"   {r['layer']:>5} {r['type']:<14} {r['cin']:>4} {r['cout']:>4} {r['kernel']:<6} {r['stride']:<6} {r['out']:<12} "
                   f"{r['us']:>8.1f} {100 * r['share']:>5.1f}% {r['tflops'] if r['tflops'] is not None else '-':>8} "
                   f"{r['tbps'] if r['tbps'] is not None else '-':>6}  {r['picked']}{padded}{mark}")
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--plans', nargs='*', default=None)
    ap.add_argument('--out', default=os.path.join(ROOT, 'profiles', 'r06_plan_sweep'))
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--timeout', type=int, default=420)
    ap.add_argument('--extra', nargs=argparse.REMAINDER, default=[], help='further bench.py arguments (after --extra)')
    args = ap.parse_args()
    plans = args.plans or sorted(glob.glob(os.path.join(ROOT, 'tools', 'plans', '*.json')))
    lines, failed = [], []
    for pf in plans:
        cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--plan', pf, '--steps', str(args.steps), '--warmup', str(args.warmup),
               '--no-cpu-baseline', '--no-clock-probe'] + args.extra
        print('+', ' '.join(cmd), flush=True)
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=args.timeout)
        except subprocess.TimeoutExpired:
            failed.append((pf, 'timeout'))
            continue
        last = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
        if r.returncode != 0 or not last:
            failed.append((pf, (r.stderr or r.stdout)[-600:]))
            print(r.stderr[-600:], flush=True)
            continue
        j = json.loads(last[-1])
        lines.append(j)
        print(f"  {j['value']:.1f} patches/s, frac {j.get('roofline', {}).get('frac')}", flush=True)
    txt = ['Plan sweep (tools/plan_sweep.py): bench.py --plan <file> --steps %d --warmup %d per plan; per-layer rows = median over the' % (args.steps, args.warmup),
           'profiled step\'s full batches (HIP events around every launch, one stream); TFLOP/s = 2*MACs, TB/s = every input read once +',
           'the output written once (fp16); peak 2500 TFLOP/s dense f16.', '']
    txt.append(f"{'plan':<28} {'patches/s':>10} {'ms/vol':>9} {'GFLOP/patch':>12} {'family TFLOP/s':>15} {'frac':>7} {'net TFLOP/s':>12}")
    for j in lines:
        name = j['config']['workload'].split(':')[0]
        rf = j.get('roofline', {})
        txt.append(f"{name:<28} {j['value']:>10.1f} {j['ms_per_step']:>9.1f} {j['config']['gflop_per_patch']:>12} {rf.get('achieved', 0):>15} "
                   f"{rf.get('frac', 0):>7} {rf.get('whole_net_tflops', 0):>12}")
    txt.append('')
    cliffs, generic = [], []
    for j in lines:
        name = j['config']['workload'].split(':')[0]
        for r in j.get('layers', []):
            why = is_cliff(r)
            if why:
                cliffs.append(f"{name}: layer {r['layer']} {r['type']} {r['cin']}->{r['cout']} k {r['kernel']} s {r['stride']} out {r['out']} "
                              f"{r['us']:.0f} us = {100 * r['share']:.1f} % of the forward, {why}  [{r['picked']}]")
        for k in j.get('roofline', {}).get('launches_by_kernel', {}):
            if 'generic' in k:
                generic.append(f'{name}: {k}')
    txt.append(f'CLIFFS ({len(cliffs)}): layers > 5 % of their forward below 10 % of the MFMA peak (or, 16-channel / transposed layers, below 2.5 TB/s)')
    txt += ['  ' + c for c in cliffs] or ['  none']
    txt.append(f'GENERIC FALLBACK KERNEL reached by {len(generic)} plan(s)')
    txt += ['  ' + g for g in generic]
    if failed:
        txt.append(f'FAILED ({len(failed)}):')
        txt += [f'  {pf}: {why}' for pf, why in failed]
    txt.append('')
    for j in lines:
        txt += fmt_plan(j) + ['']
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out + '.txt', 'w') as f:
        f.write('\n'.join(txt) + '\n')
    with open(args.out + '.json', 'w') as f:
        for j in lines:
            f.write(json.dumps(j) + '\n')
    print('\n'.join(txt[:len(lines) + 12 + len(cliffs) + len(generic)]))
    sys.exit(1 if failed else 0)


if __name__ == '__main__':
    main()
