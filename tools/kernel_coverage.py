"""Which compiled kernels does a traced run never launch?
GPU box:   cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/suite_trace -- python3 -m pytest $REPO/tests -q -m gpu
anywhere:  python tools/kernel_coverage.py gpurun_out/suite_trace
Compiled kernels = the `Function Name` lines of hipcc's resource reports (fast-nnunet_amd/csrc/*.res, written by the Makefile);
launched kernels = the Name column of every *_kernel_stats.csv under the trace directory (one per traced process)."""
import csv
import glob
import os
import re
import subprocess
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')


def norm(s):
    s = re.sub(r'^void ', '', s.replace('(anonymous namespace)::', ''))
    depth, out = 0, ''
    for ch in s:
        depth += (ch == '<') - (ch == '>')
        if ch == '(' and depth == 0:
            break
        out += ch
    return out.replace(' ', '').replace('true', '1').replace('false', '0')


def main(trace_dir):
    names = set()
    for f in glob.glob(os.path.join(ROOT, 'fast-nnunet_amd', 'csrc', '*.res')):
        names.update(re.findall(r'Function Name: (\S+)', open(f, errors='replace').read()))
    names = sorted(names)
    dem = subprocess.run(['c++filt'], input='\n'.join(names), capture_output=True, text=True, check=True).stdout.splitlines()
    compiled = {norm(d) for d in dem}
    launched = {}
    for f in glob.glob(os.path.join(trace_dir, '**', '*_kernel_stats.csv'), recursive=True):
        for row in csv.DictReader(open(f)):
            k = norm(row['Name'])
            launched[k] = launched.get(k, 0) + int(row['Calls'])
    never = sorted(k for k in compiled if k not in launched)
    print(f'{len(compiled)} kernels compiled into libfnn_hip.so, {len(compiled) - len(never)} launched by the traced run, {len(never)} never:')
    for k in never:
        print('  ', k)
    print('launches per compiled kernel (fewest first):')
    for k in sorted(compiled - set(never), key=lambda k: launched[k])[:15]:
        print(f'  {launched[k]:8d}  {k}')


if __name__ == '__main__':
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, 'gpurun_out', 'suite_trace'))
