#!/usr/bin/env python
"""Code bytes of every kernel in libcentertrack_hip.so, beside what a launch of it costs in the pool's two box states.

    python tools/code_sizes.py [--fast profiles/r05_a_kstats_fast_box.txt] [--slow profiles/r05_a_kstats_slow_box.txt]

CPU only (build container): the gfx950 code object of every translation unit is taken out of centertrack_amd/build/*.o
(`.hip_fatbin` section -> clang-offload-bundler), its kernel symbols are sized by llvm-readelf and their registers / LDS read
from the code-object metadata.  The two rocprofv3 kernel summaries (same command, a fast and a slow lease) give the average
launch duration per kernel; the table joins the three and fits  slow - fast = a + b * code_KB  over the kernels of the frame.

Why: the one box probe that differs between the states is cold instruction fetch (DESIGN.md section 4,
`box_calibration.launch_us.ifetch_64KB_code_256wg`).  If instruction fetch is what the slow state slows down, the per-launch
penalty should grow with the code a launch has to fetch and not with the bytes it moves.
"""
import argparse
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = '/opt/rocm/lib/llvm/bin'
TARGET = 'hipv4-amdgcn-amd-amdhsa--gfx950'


def sh(*cmd):
    return subprocess.run(cmd, check=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True).stdout


def short(name):
    name = name.replace('(anonymous namespace)::', '').replace('void ', '')
    return re.sub(r'\s+', ' ', name).strip()


def kernels_of(obj, tmp):
    """[(demangled kernel, code bytes, vgprs, sgprs, lds bytes, scratch bytes)] of one host object."""
    fb = os.path.join(tmp, os.path.basename(obj) + '.fb')
    co = os.path.join(tmp, os.path.basename(obj) + '.co')
    try:
        sh(LLVM + '/llvm-objcopy', '--dump-section', '.hip_fatbin=' + fb, obj, fb + '.copy')     # (an explicit output: without one objcopy rewrites `obj` in place and the build takes it for fresh)
    except subprocess.CalledProcessError:
        return []
    if not os.path.exists(fb) or os.path.getsize(fb) == 0:
        return []
    sh(LLVM + '/clang-offload-bundler', '--unbundle', '--type=o', '--input=' + fb, '--targets=' + TARGET, '--output=' + co)
    sizes = {}
    for line in sh(LLVM + '/llvm-readelf', '-sW', co).splitlines():
        f = line.split()
        if len(f) >= 8 and f[3] == 'FUNC':
            sizes[f[7]] = int(f[2])
    meta = {}
    cur = None
    for line in sh(LLVM + '/llvm-readelf', '--notes', co).splitlines():
        m = re.match(r'\s*(-\s+)?\.(\w+):\s*(.*)$', line)
        if not m:
            continue
        key, val = m.group(2), m.group(3).strip().strip("'")
        if key == 'agpr_count' or (m.group(1) and key != 'name'):       # first key of a kernel entry (keys are sorted)
            pass
        if key == 'name' and val in sizes:
            cur = meta.setdefault(val, {})
        elif key == 'symbol':
            cur = meta.setdefault(val[:-3] if val.endswith('.kd') else val, {})
        if cur is not None and key in ('vgpr_count', 'sgpr_count', 'group_segment_fixed_size', 'private_segment_fixed_size'):
            cur.setdefault(key, int(val))
    out = []
    names = list(sizes)
    dem = sh('c++filt', *names).splitlines() if names else []
    for mangled, d in zip(names, dem):
        m = meta.get(mangled, {})
        out.append((short(d), sizes[mangled], m.get('vgpr_count'), m.get('sgpr_count'), m.get('group_segment_fixed_size'),
                    m.get('private_segment_fixed_size')))
    return out


def kstats(path):
    rows = {}
    for line in open(path):
        m = re.match(r'^(.*?\S)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s*$', line)
        if m:
            rows[short(m.group(1))] = (int(m.group(2)), float(m.group(4)))
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--fast', default=os.path.join(ROOT, 'profiles', 'r05_a_kstats_fast_box.txt'))
    ap.add_argument('--slow', default=os.path.join(ROOT, 'profiles', 'r05_a_kstats_slow_box.txt'))
    ap.add_argument('--all', action='store_true', help='list every kernel of the library, not only the profiled ones')
    a = ap.parse_args()
    objdir = os.path.join(ROOT, 'centertrack_amd', 'build')
    ks = []
    with tempfile.TemporaryDirectory() as tmp:
        for o in sorted(os.listdir(objdir)):
            if o.endswith('.o'):
                ks += [(o[:-2],) + k for k in kernels_of(os.path.join(objdir, o), tmp)]
    fast, slow = kstats(a.fast), kstats(a.slow)
    print('%-62s %8s %5s %5s %7s %7s | %6s %8s %8s %7s %6s' % ('kernel', 'code_B', 'vgpr', 'sgpr', 'lds_B', 'scr_B', 'calls', 'fast_us',
                                                              'slow_us', 'delta', 'ratio'))
    pts = []
    total = 0
    for unit, name, size, vg, sg, lds, scr in sorted(ks, key=lambda k: -k[2]):
        key = name
        if key not in fast:                      # rocprofv3 prints the parameter list without namespaces; match on it
            cands = [k for k in fast if k.split('(')[0] == name.split('(')[0]]
            key = cands[0] if len(cands) == 1 else None
        if key is None or key not in slow:
            if a.all:
                print('%-62s %8d %5s %5s %7s %7s |' % (name[:62], size, vg, sg, lds, scr))
            continue
        calls, f = fast[key]
        s = slow[key][1]
        if name.startswith('calib_') or calls < 100:
            continue
        total += size
        pts.append((size / 1024.0, s - f, calls))
        print('%-62s %8d %5s %5s %7s %7s | %6d %8.2f %8.2f %+7.2f %6.2f' % (name[:62], size, vg, sg, lds, scr, calls, f, s, s - f, s / f))
    n = len(pts)
    if n >= 3:
        mx = sum(p[0] for p in pts) / n
        my = sum(p[1] for p in pts) / n
        sxx = sum((p[0] - mx) ** 2 for p in pts)
        sxy = sum((p[0] - mx) * (p[1] - my) for p in pts)
        syy = sum((p[1] - my) ** 2 for p in pts)
        b = sxy / sxx
        print('\n%d profiled kernels of the frame, %.0f KB of code between them' % (n, total / 1024.0))
        print('slow - fast [us per launch] = %.2f + %.3f * code_KB      (Pearson r = %.2f)' % (my - b * mx, b, sxy / (sxx * syy) ** 0.5))
        small = [p for p in pts if p[0] < 16]
        big = [p for p in pts if p[0] >= 24]
        if small and big:
            print('mean penalty per launch: %.2f us over the %d kernels under 16 KB, %.2f us over the %d kernels of 24 KB and more' % (
                sum(p[1] for p in small) / len(small), len(small), sum(p[1] for p in big) / len(big), len(big)))


if __name__ == '__main__':
    sys.exit(main())
