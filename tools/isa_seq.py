"""Compact opcode sequence of a kernel's main loop from hipcc -S output (M = 32x32x2 MFMA, m = 4x4x1 MFMA, x = v_max,
r/w/v = accvgpr read/write/mov, n = s_nop, R/W = ds read/write, c = cndmask, p = cmp, * = mul/pk_mul, + = add)."""
import re, sys, collections
path, key = sys.argv[1], sys.argv[2]
s = open(path).read().split('\n')
start = [i for i, l in enumerate(s) if l.startswith('_ZN4gngf') and key in l and l.rstrip().endswith(':') is False and ':' in l][0]
end = [i for i, l in enumerate(s) if i > start and l.strip().startswith('s_endpgm')][0]
body = s[start:end]
idx = [i for i, l in enumerate(body) if 'v_mfma' in l]
lo, hi = idx[0] - int(sys.argv[3]) if len(sys.argv) > 3 else idx[0] - 30, idx[-1] + (int(sys.argv[4]) if len(sys.argv) > 4 else 60)
short = {'v_mfma_f32_32x32x2_f32': 'M', 'v_mfma_f32_4x4x1_16b_f32': 'm', 'v_max_f32_e32': 'x', 'v_accvgpr_read_b32': 'r',
         'v_accvgpr_write_b32': 'w', 'v_accvgpr_mov_b32': 'v', 's_nop': 'n', 'v_cndmask_b32_e64': 'c', 'v_cndmask_b32_e32': 'c',
         'v_cmp_lt_f32_e64': 'p', 'v_cmp_lt_f32_e32': 'p', 'v_cmp_gt_f32_e64': 'p', 'v_cmp_gt_f32_e32': 'p', 'v_mul_f32_e32': '*', 'v_pk_mul_f32': '*', 'v_add_f32_e32': '+', 'v_pk_add_f32': '+',
         'v_add_u32_e32': 'a', 's_waitcnt': 'WAIT'}
seq, cnt = [], collections.Counter()
for l in body[lo:hi]:
    l = l.strip()
    if not l or l.startswith(';') or l.startswith('.'):
        continue
    op = l.split()[0]
    cnt[op] += 1
    if op.startswith('ds_read'): op = 'R'
    elif op.startswith('ds_write'): op = 'W'
    elif op == 's_waitcnt': op = 'WAIT[' + l.split(None, 1)[1].split(';')[0].strip() + ']'
    seq.append(short.get(op, op))
out, prev, n = [], None, 0
for o in seq + [None]:
    if o == prev: n += 1
    else:
        if prev: out.append(prev + (str(n) if n > 1 else ''))
        prev, n = o, 1
print(' '.join(out))
print('instructions:', sum(cnt.values()), ' non-MFMA:', sum(v for k, v in cnt.items() if 'mfma' not in k))
if '-v' in sys.argv:
    for k, v in cnt.most_common(30): print(f'  {k:30s}{v}')
