#!/usr/bin/env python
"""Static check of the F(4x4,3x3) kernels' compiled ISA (csrc/conv_wino4.hip, conv_wino4w.hip, conv_wino4h.hip, conv_wino4r.hip).

The kernel loads its MFMA B operands (the transformed filter) with raw `buffer_load_dword` instructions and
waits for them with hand-counted `s_waitcnt vmcnt(N)`: the compiler does not know that the destination
registers are written asynchronously.  The source is arranged so that a loaded register reaches its wait
untouched; this script ASSERTS it on the ISA hipcc produced: between a `buffer_load_dword[x4] vN` and the
`s_waitcnt vmcnt` that retires it, no instruction may read or write vN (a register copy there would move a
value that has not arrived).  It also checks that no scratch is used (a spill would do the same).

    python tools/check_wino4_isa.py [conv_wino4.s]        # without an argument: compiles the file with hipcc -S
Exit code 0 = clean.  Linear scan in program order (the stage loop is straight-line code by construction).
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def compile_isa(name='conv_wino4'):
    src = os.path.join(ROOT, 'egonet_amd', 'csrc', name + '.hip')
    out = os.path.join(tempfile.mkdtemp(prefix='w4isa'), name + '.s')
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    # (conv_wino4h.hip's and conv_wino4r.hip's kernels exist in probe builds only)
    subprocess.run([hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-c', '-S', '--cuda-device-only'] +
                   (['-DEGN_PROBES'] if name in ('conv_wino4h', 'conv_wino4r') else []) + ['-o', out, src],
                   check=True, stderr=subprocess.DEVNULL)
    return out


def regs_of(operand_text):
    """Set of VGPR numbers named in an operand string (v12, v[4:7])."""
    out = set()
    for m in re.finditer(r'\bv\[(\d+):(\d+)\]|\bv(\d+)\b', operand_text):
        if m.group(3) is not None:
            out.add(int(m.group(3)))
        else:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    return out


def inflight(queue):
    out = set()
    for q in queue:
        if q is not None:
            out |= q
    return out


# the geometries of the body (GEO 0 / 1 / 2, the last one without and with the K split), product builds
KERNELS = ('conv_wino4_kernelILi0E', 'conv_wino4b_kernelILi0E', 'conv_wino4bk_kernelILi0E', 'conv_wino4c_kernelILi0ELi1E',
           'conv_wino4c_kernelILi0ELi2E',
           # the training tape's builds (BatchNorm statistics in the item end) [round 5]
           'conv_wino4s_kernelILi0ELi1E', 'conv_wino4s_kernelILi1ELi1E', 'conv_wino4s_kernelILi1ELi2E',
           'conv_wino4s_kernelILi2ELi1E', 'conv_wino4s_kernelILi2ELi2E')


# conv_wino4w.hip [round 6]: 96 output channels per item
KERNELS_W = ('conv_wino4w_kernelILi0E',)
# conv_wino4h.hip [round 6]: half-size blocks, two per CU
KERNELS_H = ('conv_wino4h_kernelILi0E', 'conv_wino4d_kernelILi0E')
# conv_wino4r.hip [round 6]: row-owner waves, one exchange round
KERNELS_R = ('conv_wino4r_kernelILi0E',)
FILES = (('conv_wino4', KERNELS), ('conv_wino4w', KERNELS_W), ('conv_wino4h', KERNELS_H), ('conv_wino4r', KERNELS_R))


def check(path, kernel='conv_wino4_kernelILi0E'):
    text = open(path).read()
    m = re.search(r'^(_Z\d+%s\w*):' % kernel, text, re.M)
    assert m, 'kernel symbol not found: ' + kernel
    body = text[m.end():text.index('s_endpgm', m.end())]
    queue = []          # outstanding vector-memory operations, oldest first: destination VGPRs or None
    lds = []            # outstanding LDS / scalar-memory operations (lgkmcnt), oldest first
    problems, nload, nwait = [], 0, 0
    for ln in body.split('\n'):
        t = ln.split(';')[0].strip()
        if not t or t.endswith(':') or t.startswith('.'):
            continue
        op, _, rest = t.partition(' ')
        if op.startswith('buffer_load') or op.startswith('buffer_store') or op.startswith('buffer_atomic'):
            dst = None
            if op.startswith('buffer_load') and ' lds' not in t:
                dst = frozenset(regs_of(rest.split(',', 1)[0]))
                nload += 1
            used = regs_of(rest.split(',', 1)[1] if dst is not None else rest)     # address / data operands
            hit = used & inflight(queue)
            if hit:
                problems.append('%s  <- uses in-flight v%s' % (t, sorted(hit)))
            queue.append(dst)
            continue
        if op == 's_waitcnt':
            mm = re.search(r'vmcnt\((\d+)\)', rest)
            if mm:
                nwait += 1
                keep = int(mm.group(1))
                while len(queue) > keep:
                    queue.pop(0)
            mm = re.search(r'lgkmcnt\((\d+)\)', rest)
            if mm:                                  # LDS operations complete in order
                keep = int(mm.group(1))
                while len(lds) > keep:
                    lds.pop(0)
            continue
        if op.startswith('ds_read') or op.startswith('ds_write') or op.startswith('s_load') or op.startswith('s_buffer_load'):
            dst = frozenset(regs_of(rest.split(',', 1)[0])) if op.startswith('ds_read') else None
            used = regs_of(rest.split(',', 1)[1]) if op.startswith('ds_read') else regs_of(rest)
            hit = used & (inflight(queue) | inflight(lds))
            if hit:
                problems.append('%s  <- uses in-flight v%s' % (t, sorted(hit)))
            lds.append(dst)
            continue
        hit = regs_of(rest) & (inflight(queue) | inflight(lds))
        if hit:
            problems.append('%s  <- touches in-flight v%s' % (t, sorted(hit)))
    for scratch in re.finditer(r'\.private_segment_fixed_size:\s*(\d+)', text):
        if int(scratch.group(1)) != 0:
            problems.append('scratch in use: %s bytes per lane' % scratch.group(1))
    return problems, nload, nwait


if __name__ == '__main__':
    bad = 0
    for name, kernels in FILES:
        if len(sys.argv) > 1 and os.path.splitext(os.path.basename(sys.argv[1]))[0] != name:
            continue
        path = sys.argv[1] if len(sys.argv) > 1 else compile_isa(name)
        for kern in kernels:
            problems, nload, nwait = check(path, kern)
            print('%s: %d filter / residual loads, %d vmcnt waits, %d problems' % (kern, nload, nwait, len(problems)))
            for p in problems[:40]:
                print('  ' + p)
            bad += len(problems)
    sys.exit(1 if bad else 0)
