#!/usr/bin/env python3
"""Where a kernel's register spills sit: scratch loads / stores per loop depth of a gfx950 assembly listing
(hipcc -S --cuda-device-only), and every block inside a loop nest of depth >= 3 that holds one.

usage: tools/hot_spills.py listing.s kernel-name-substring

The sweep kernels of the register engine run in one of two modes that differ by 3-4 % (measured at 12 and 13
qubits with otherwise equivalent builds): either the block of the phase loop that holds the matrix instructions
(v_mfma) is free of scratch traffic, or the allocator has put two reloads there.  Unrelated changes of the
per-sample code flip a template instance from one mode to the other; this tells which one a build is in."""
import re
import sys

src, key = open(sys.argv[1]).read(), sys.argv[2]
m = re.search(r'^(_Z\w*%s\w*):\s*; @.*?\n(.*?)\.Lfunc_end\d+:' % re.escape(key), src, re.S | re.M)
if not m:
    sys.exit("no kernel matches " + key)
blocks, cur = [], ['<entry>', []]
for line in m.group(2).split('\n'):
    if re.match(r'^\.LBB\d+_\d+:', line):
        blocks.append(cur)
        cur = [line.strip(), []]
    else:
        cur[1].append(line)
blocks.append(cur)
by_depth, hot = {}, 0
for name, lines in blocks:
    count = lambda pat: sum(1 for l in lines if re.search(pat, l))
    d = re.search(r'Depth=(\d+)', name)
    depth = int(d.group(1)) if d else 0
    acc = by_depth.setdefault(depth, [0, 0, 0])
    acc[0] += count('scratch_load')
    acc[1] += count('scratch_store')
    acc[2] += len(lines)
    if depth >= 3 and count('scratch_'):
        if count('v_mfma'):
            hot += count('scratch_')
        print("%-12s depth %d  lines %4d  f64 %3d  mfma %2d  scratch loads %2d  stores %2d  lds %2d  barriers %d" % (
            name.split(':')[0], depth, len(lines), count(r'v_(fma|mul|add)_f64'), count('v_mfma'), count('scratch_load'),
            count('scratch_store'), count(r'\bds_'), count('s_barrier')))
for depth in sorted(by_depth):
    print("depth %d: %d scratch loads, %d stores in %d lines" % (depth, *by_depth[depth]))
print("scratch operations in blocks with matrix instructions (the phase loop): %d" % hot)
