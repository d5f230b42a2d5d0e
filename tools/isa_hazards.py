#!/usr/bin/env python
"""Lint of the SHIPPED device code for the one hardware hazard hipcc does not pad on gfx950 (found in round 6):

    a vector-memory store of more than 64 bits (`*_store_dwordx3 / x4`) whose data registers are overwritten by a VALU
    instruction within the next two issue slots.

The published rule exempts buffer stores that carry a scalar offset REGISTER, LLVM's hazard recognizer follows the
publication (GCNHazardRecognizer::createsVALUHazard: no hazard if `soffset` is a register), and gfx950 does not honour the
exemption: `buffer_store_dwordx4 v[2:5], v23, s[8:11], s19 offen` followed by `v_or_b32 v2, 2, v11` stored the OR's result
in lanes 12-15 of every 16, once per ~15 launches of the Winograd OFFSETS kernel at 4 streams (tools/determinism.py;
profiles/r06_an_store_hazard.txt).  The kernels therefore keep the scalar-offset field of wide stores at the constant 0 (the
compiler then pads), and this lint disassembles every code object of centertrack_amd/build/*.o and looks at what the
compiler actually emitted.  CPU only; tests/test_cabi.py runs it.

(The analogous LDS pattern -- `ds_write_b128` followed directly by a vector write of its data registers, which LLVM exempts unless the
instruction uses GDS -- occurs 35 times in the shipped code, 29 of them in the stride-2 conv shapes every plan runs, and those kernels are
bit-reproducible over > 10 000 runs (tests/test_hip_determinism.py, profiles/r06_an_store_hazard.txt): that exemption holds on gfx950,
LDS writes are not linted.)

    python tools/isa_hazards.py            # prints findings, exit status 1 if any
"""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = '/opt/rocm/lib/llvm/bin'
TARGET = 'hipv4-amdgcn-amd-amdhsa--gfx950'
WIDE_STORE = re.compile(r'^(buffer|global|flat|scratch)_store_(dwordx3|dwordx4|format_xyz|format_xyzw)\b')


def sh(*cmd):
    return subprocess.run(cmd, check=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True).stdout


def code_object(obj, tmp):
    fb = os.path.join(tmp, os.path.basename(obj) + '.fb')
    co = os.path.join(tmp, os.path.basename(obj) + '.co')
    try:
        sh(LLVM + '/llvm-objcopy', '--dump-section', '.hip_fatbin=' + fb, obj, fb + '.copy')     # (an explicit output: without one objcopy rewrites `obj` in place and the build takes it for fresh)
    except subprocess.CalledProcessError:
        return None
    if not os.path.exists(fb) or os.path.getsize(fb) == 0:
        return None
    sh(LLVM + '/clang-offload-bundler', '--unbundle', '--type=o', '--input=' + fb, '--targets=' + TARGET, '--output=' + co)
    return co


def vregs(op):
    """set of VGPR numbers named by one operand (v7, v[2:5]); empty for anything else"""
    m = re.match(r'^v(\d+)$', op)
    if m:
        return {int(m.group(1))}
    m = re.match(r'^v\[(\d+):(\d+)\]$', op)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    return set()


def parse(line):
    """(mnemonic, [operands]) of one objdump line, or None"""
    line = line.split('//')[0].strip()
    if not line or line.endswith(':') or line.startswith(('.', 'Disassembly', '/')):
        return None
    parts = line.split(None, 1)
    mn = parts[0]
    ops = [o.strip() for o in parts[1].split(',')] if len(parts) > 1 else []
    return mn, ops


def store_data(mn, ops):
    if mn.startswith('buffer_'):
        return vregs(ops[0])
    return vregs(ops[1]) if len(ops) > 1 else set()       # global / flat / scratch: address first, data second


def valu_writes(mn, ops):
    if not mn.startswith('v_') or mn.startswith(('v_cmp', 'v_readfirstlane', 'v_readlane', 'v_nop')):
        return set()
    w = vregs(ops[0]) if ops else set()
    if mn.startswith('v_swap') and len(ops) > 1:
        w |= vregs(ops[1])
    return w


def lint(co):
    """[(kernel, store line, offending line)] of one code object"""
    out = []
    kernel = '?'
    insts = []
    for line in sh(LLVM + '/llvm-objdump', '-d', '--no-show-raw-insn', co).splitlines():
        m = re.match(r'^[0-9a-f]+ <(.+)>:$', line.strip())
        if m:
            kernel = m.group(1)
            continue
        p = parse(line)
        if p:
            insts.append((kernel, p[0], p[1], line.split('//')[0].strip()))
    for i, (k, mn, ops, txt) in enumerate(insts):
        if not WIDE_STORE.match(mn):
            continue
        data = store_data(mn, ops)
        states = 0
        for k2, mn2, ops2, txt2 in insts[i + 1:i + 4]:
            if k2 != k or states >= 2:
                break
            if mn2 == 's_nop':
                states += int(ops2[0], 0) + 1
                continue
            if valu_writes(mn2, ops2) & data:
                out.append((k, txt, txt2))
                break
            states += 1
    return out


def run(objs=None):
    objs = objs or sorted(glob.glob(os.path.join(ROOT, 'centertrack_amd', 'build', '*.o')))
    findings, n = [], 0
    with tempfile.TemporaryDirectory() as tmp:
        for obj in objs:
            co = code_object(obj, tmp)
            if co is None:
                continue
            n += 1
            findings += [(os.path.basename(obj),) + f for f in lint(co)]
    return n, findings


def main():
    n, findings = run()
    for f in findings:
        print('HAZARD %s  %s\n    %s\n    %s' % f)
    print('%d code objects, %d wide stores overwritten within two issue slots' % (n, len(findings)))
    return 1 if findings else 0


if __name__ == '__main__':
    sys.exit(main())
