#!/usr/bin/env python3
"""Instruction mix after the last MFMA (= the epilogue) of one kernel in a `hipcc -S --cuda-device-only` dump.
Usage: tools/epi_mix.py <file.s> <mangled-name-prefix>"""
import collections, sys
lines = open(sys.argv[1]).read().split('\n')
key = sys.argv[2]
i0 = next(i for i, l in enumerate(lines) if l.startswith(key) and ': ' in l[:len(key) + 60] or (l.startswith(key) and l.rstrip().endswith(':')))
i1 = next(i for i in range(i0, len(lines)) if lines[i].startswith('.Lfunc_end'))
body = lines[i0:i1]
last = max(i for i, l in enumerate(body) if 'v_mfma' in l)
c = collections.Counter()
for l in body[last + 1:]:
    t = l.strip()
    if not t or t.startswith(';') or t.startswith('.'):
        continue
    c[t.split()[0]] += 1
print('instructions after the last MFMA:', sum(c.values()))
print('  '.join(f'{k}:{v}' for k, v in c.most_common(int(sys.argv[3]) if len(sys.argv) > 3 else 45)))
