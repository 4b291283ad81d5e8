"""Opcode sequence around the MFMAs of one kernel in hipcc -S output: python tools/isa_loop.py file.s <mangled-name-substring> [before] [after]"""
import sys
s = open(sys.argv[1]).read().split('\n')
key = sys.argv[2]
st = [i for i, l in enumerate(s) if l.startswith('_ZN') and key in l and ':' in l][0]
en = [i for i, l in enumerate(s) if i > st and 's_endpgm' in l][0]
body = s[st:en]
mf = [i for i, l in enumerate(body) if 'v_mfma' in l]
before = int(sys.argv[3]) if len(sys.argv) > 3 else 40
after = int(sys.argv[4]) if len(sys.argv) > 4 else 30
seq = []
for l in body[max(0, mf[0] - before):mf[-1] + after]:
    l = l.strip()
    if not l or l.startswith(';'): continue
    op = l.split()[0]
    if op.startswith('v_mfma'): op = 'M'
    elif op.startswith('ds_read'): op = 'R'
    elif op.startswith('ds_write'): op = 'W'
    elif op == 's_waitcnt': op = '[' + l.split(None, 1)[1].split(';')[0].strip() + ']'
    seq.append(op)
print(' '.join(seq))
