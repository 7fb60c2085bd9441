#!/usr/bin/env python3
"""Instruction / register statistics per kernel of a hipcc -S --cuda-device-only listing."""
import re, sys
s = open(sys.argv[1]).read()
for m in re.finditer(r'^(_Z\w+):\s*; @.*?\n(.*?)\.Lfunc_end\d+:', s, re.S | re.M):
    name, body = m.group(1), m.group(2)
    if len(sys.argv) > 2 and sys.argv[2] not in name:
        continue
    c = lambda pat: len(re.findall(pat, body))
    meta = {k: (re.search(r'\.set %s\.%s, (\d+)' % (re.escape(name), k), s) or [None, '?'])[1] for k in ('num_vgpr', 'num_agpr', 'numbered_sgpr', 'private_seg_size')}
    print(name[-40:], meta, 'fma', c(r'v_fma_f64'), 'mul', c(r'v_mul_f64'), 'add', c(r'v_add_f64'), 'mov64', c(r'v_mov_b64'),
          'mov32', c(r'v_mov_b32'), 'xlane', c(r'dpp|swizzle|bpermute'), 'acc', c(r'v_accvgpr'), 'scratch', c(r'scratch_'), 'lines', body.count('\n'))
