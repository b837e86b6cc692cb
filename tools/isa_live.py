"""VGPR liveness over the MFMA loop of one kernel in a hipcc -S dump: max live registers and where (straight-line approximation of
the loop body taken as a cycle).  python tools/isa_live.py file.s <kernel name filter>"""
import re, sys
txt = open(sys.argv[1]).read().split('\n')
flt = sys.argv[2]
starts = [i for i, l in enumerate(txt) if re.match(r'^_Z\w+:', l)]
s = [i for i in starts if flt in txt[i]][0]
e = [i for i in range(s, len(txt)) if txt[i].startswith('.Lfunc_end')][0]
body = txt[s:e]
labels = {m.group(1): i for i, l in enumerate(body) if (m := re.match(r'^(\.LBB\d+_\d+):', l))}
best = None
for i, l in enumerate(body):
    m = re.search(r's_cbranch_\w+\s+(\.LBB\d+_\d+)', l) or re.search(r's_branch\s+(\.LBB\d+_\d+)', l)
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        seg = body[labels[m.group(1)]:i]
        if sum('v_mfma' in x for x in seg) and (best is None or len(seg) > len(best)): best = seg
ins = [x.strip() for x in best if re.match(r'^\s+[a-z]', x) and not x.strip().startswith(('.', ';'))]
def regs(tok):
    out = []
    for m in re.finditer(r'\bv\[(\d+):(\d+)\]|\bv(\d+)\b', tok):
        if m.group(1): out += list(range(int(m.group(1)), int(m.group(2)) + 1))
        else: out.append(int(m.group(3)))
    return out
NODEF = ('v_cmp', 'scratch_store', 'ds_write', 'global_store', 'buffer_store', 'global_load_lds', 'v_readfirstlane', 'v_readlane', 's_', 'ds_bpermute_nodef')
du = []
for x in ins:
    op = x.split()[0]
    ops = x[len(op):].split(';')[0]
    parts = [p.strip() for p in ops.split(',')]
    if op.startswith(NODEF) or not parts or not regs(parts[0]):
        du.append((set(), set(r for p in parts for r in regs(p))))
    else:
        du.append((set(regs(parts[0])), set(r for p in parts[1:] for r in regs(p))))
live = set()
for _ in range(3):                       # iterate the cycle to a fixed point
    trace = []
    for (d, u) in reversed(du):
        live = (live - d) | u
        trace.append(len(live))
trace.reverse()
mx = max(trace); at = trace.index(mx)
print(f'{len(ins)} instructions in the loop; max live VGPRs {mx} at instruction {at}: {ins[at]}')
print('live at the barriers:', [trace[i] for i, x in enumerate(ins) if x.startswith('s_barrier')])
if len(sys.argv) > 3:                    # third argument: where the registers live at the maximum were last written (by opcode)
    live = set()
    lives = []
    for _ in range(3):
        lives = []
        for (d, u) in reversed(du):
            live = (live - d) | u
            lives.append(set(live))
    lives.reverse()
    L = lives[at]
    import collections
    src = collections.Counter()
    for r in L:
        k = at
        for back in range(1, len(du) + 1):
            k = (at - back) % len(du)
            if r in du[k][0]:
                break
        src[(ins[k].split()[0], 'before' if k < at else 'wraps')] += 1
    print('defined by:', src.most_common(25))
