#!/usr/bin/env python3
"""Static instruction audit of one kernel of a `hipcc -S --cuda-device-only` listing.

    hipcc <flags of csrc/Makefile> -S --cuda-device-only prep.hip -o /tmp/prep.s
    python tools/probes/isa_audit.py /tmp/prep.s prep_fast32_kernel [--blocks]

Prints, per region between two s_barrier instructions (the kernel's phases) and per basic block, the number of
vector-ALU, scalar, vector-memory (load / store), LDS and other instructions, and the most frequent vector
opcodes -- the table the round-5 review asked for (which instructions make up the ~87 lane operations per
element of the prep kernel).  Static counts: a loop body counts once; trip counts are the reader's.
"""
import collections
import re
import sys


def classify(op):
    if op.startswith(('v_mfma', 'v_smfmac')):
        return 'mfma'
    if op.startswith(('global_load', 'buffer_load', 'flat_load', 'scratch_load')):
        return 'vmem_ld'
    if op.startswith(('global_store', 'buffer_store', 'flat_store', 'scratch_store')):
        return 'vmem_st'
    if op.startswith(('global_atomic', 'buffer_atomic', 'flat_atomic')):
        return 'vmem_at'
    if op.startswith('ds_'):
        return 'lds'
    if op.startswith('v_'):
        return 'valu'
    if op.startswith('s_waitcnt'):
        return 'wait'
    if op.startswith('s_barrier'):
        return 'barrier'
    if op.startswith(('s_load', 's_buffer_load')):
        return 'smem'
    if op.startswith(('s_cbranch', 's_branch')):
        return 'branch'
    if op.startswith('s_'):
        return 'salu'
    return 'other'


def main():
    path, name = sys.argv[1], sys.argv[2]
    per_block = '--blocks' in sys.argv
    lines = open(path).read().split('\n')
    start = next(i for i, l in enumerate(lines) if re.match(r'^_Z\S*' + re.escape(name) + r'\S*:', l))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith('.Lfunc_end'))
    region, block = 0, 'entry'
    reg_cnt = collections.defaultdict(collections.Counter)
    reg_ops = collections.defaultdict(collections.Counter)
    blk_cnt = collections.OrderedDict()
    for l in lines[start + 1:end + 1]:
        s = l.strip()
        if not s or s.startswith((';', '.')) and not s.startswith('.LBB'):
            continue
        m = re.match(r'^(\.LBB\S+):', s)
        if m:
            block = m.group(1)
            continue
        op = s.split()[0]
        k = classify(op)
        key = (region, block)
        blk_cnt.setdefault(key, collections.Counter())[k] += 1
        reg_cnt[region][k] += 1
        if k == 'valu':
            reg_ops[region][op] += 1
        if k == 'barrier':
            region += 1
    kinds = ['valu', 'salu', 'vmem_ld', 'vmem_st', 'vmem_at', 'lds', 'mfma', 'wait', 'branch', 'smem']
    print('kernel', name, ': static instruction counts per barrier-delimited region')
    print('%-8s' % 'region', ' '.join('%8s' % k for k in kinds))
    tot = collections.Counter()
    for r in sorted(reg_cnt):
        print('%-8d' % r, ' '.join('%8d' % reg_cnt[r][k] for k in kinds))
        tot.update(reg_cnt[r])
    print('%-8s' % 'total', ' '.join('%8d' % tot[k] for k in kinds))
    for r in sorted(reg_ops):
        top = ', '.join('%s x%d' % kv for kv in reg_ops[r].most_common(14))
        print('region %d vector opcodes: %s' % (r, top))
    if per_block:
        print('\nper basic block (region, label): counts')
        for (r, b), c in blk_cnt.items():
            print('  r%-2d %-12s' % (r, b), ' '.join('%s=%d' % (k, c[k]) for k in kinds if c[k]))


if __name__ == '__main__':
    main()
