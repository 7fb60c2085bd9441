#!/usr/bin/env python3
"""Per-basic-block instruction mix of one kernel in a hipcc -S listing (blocks with spills or heavy FP64)."""
import re, sys
s = open(sys.argv[1]).read()
m = re.search(r'^(_Z\w*%s\w*):\s*; @.*?\n(.*?)\.Lfunc_end\d+:' % re.escape(sys.argv[2]), s, re.S | re.M)
blocks, cur = [], ['<entry>', []]
for l in m.group(2).split('\n'):
    if re.match(r'^\.LBB\d+_\d+:', l):
        blocks.append(cur); cur = [l.strip(), []]
    else:
        cur[1].append(l)
blocks.append(cur)
for name, ls in blocks:
    c = lambda pat: sum(1 for l in ls if re.search(pat, l))
    if c('scratch_') or c('v_mov_b64') > 4 or c('v_(fma|mul|add)_f64') > 30:
        print(name[:14], name[-28:], 'lines', len(ls), 'f64', c('v_(fma|mul|add)_f64'), 'mov64', c('v_mov_b64'), 'xlane', c('dpp|swizzle|bpermute'),
              'ds', c(r'\bds_'), 'scratch_ld', c('scratch_load'), 'scratch_st', c('scratch_store'), 'barrier', c('s_barrier'))
