"""Instruction mix of the innermost loops of selected kernels in a hipcc -S listing (analysis helper).
usage: isa_loop_stats.py listing.s PREFIX [PREFIX ...]"""
import re, sys
s = open(sys.argv[1]).read().split('\n')
def loop_stats(prefix):
    starts = [i for i, l in enumerate(s) if l.startswith(prefix)]
    if not starts:
        print(prefix, 'not found'); return
    start = starts[0]
    end = [i for i in range(start, len(s)) if s[i].startswith('.Lfunc_end')][0]
    labels = {m.group(1): i for i in range(start, end) for m in [re.match(r'^(\.LBB\d+_\d+):', s[i])] if m}
    loops = []
    for i in range(start, end):
        m = re.match(r'\s+s_c?branch\w*\s+(\.LBB\d+_\d+)', s[i])
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            loops.append((labels[m.group(1)], i))
    # innermost = loops not containing another loop
    inner = [l for l in loops if not any(o != l and l[0] <= o[0] and o[1] <= l[1] for o in loops)]
    for h, e in inner:
        if e - h < 60: continue
        cnt = {}
        ops = {}
        for i in range(h, e + 1):
            m = re.match(r'\s+([a-z_0-9]+)', s[i])
            if not m: continue
            op = m.group(1)
            k = ('pk_fma' if op.startswith('v_pk_fma') else 'fma' if op.startswith(('v_fma', 'v_fmac')) else 'lds' if op.startswith('ds_')
                 else 'vmem' if op.startswith(('global_', 'buffer_', 'scratch_')) else 'valu' if op.startswith('v_')
                 else 'wait' if op.startswith('s_waitcnt') else 'salu' if op.startswith('s_') else 'other')
            cnt[k] = cnt.get(k, 0) + 1
            if k == 'valu': ops[op] = ops.get(op, 0) + 1
        print(prefix[:60], 'lines', e - h, cnt)
        print('    ', sorted(ops.items(), key=lambda x: -x[1])[:18])
for p in sys.argv[2:]:
    loop_stats(p)
