#!/usr/bin/env python3
"""Print the run-length-compressed opcode stream around the MFMA loop of one kernel in a hipcc -S dump.
Usage: tools/asm_loop.py <file.s> <mangled-name-substring>"""
import sys
lines = open(sys.argv[1]).read().split('\n')
key = sys.argv[2]
starts = [i for i, l in enumerate(lines) if l.startswith('_Z') and key in l and l.rstrip().endswith(':') or (l.startswith('_Z') and key in l and ': ' in l)]
i0 = starts[0]
i1 = next(i for i in range(i0, len(lines)) if lines[i].startswith('.Lfunc_end'))
body = lines[i0:i1]
idx = [i for i, l in enumerate(body) if 'v_mfma' in l]
seg = body[max(0, idx[0] - 12):idx[-1] + 8]
out = []
for l in seg:
    t = l.strip()
    if not t or t.startswith(';') or (t.startswith('.') and not t.startswith('.LBB')):
        continue
    op = t.split()[0]
    out.append(op if not op.startswith(('s_waitcnt', '.LBB', 's_barrier', 's_cbranch')) else t.split(';')[0].strip())
res, prev, cnt = [], None, 0
for o in out:
    if o == prev:
        cnt += 1
    else:
        if prev:
            res.append(f"{prev} x{cnt}" if cnt > 1 else prev)
        prev, cnt = o, 1
res.append(f"{prev} x{cnt}" if cnt > 1 else prev)
print(' | '.join(res))
for l in lines[i1:i1 + 400]:
    if any(k in l for k in ('.vgpr_count', '.sgpr_count', 'vgpr_spill', '.private_segment_fixed_size')):
        print(l.strip())
    if l.startswith('_Z'):
        break
