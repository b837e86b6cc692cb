"""Instruction mix of the MFMA loop of each kernel in a hipcc -S dump: python tools/isa_loop.py file.s [name filter]
(the loop = the innermost backward branch whose body holds MFMAs)"""
import re, sys
txt = open(sys.argv[1]).read().split('\n')
flt = sys.argv[2] if len(sys.argv) > 2 else ''
starts = [i for i, l in enumerate(txt) if re.match(r'^_Z\w+:', l)]
for si, s in enumerate(starts):
    name = txt[s].split(':')[0]
    if flt not in name: continue
    e = starts[si + 1] if si + 1 < len(starts) else len(txt)
    body = txt[s:e]
    labels = {m.group(1): i for i, l in enumerate(body) if (m := re.match(r'^(\.LBB\d+_\d+):', l))}
    best = None
    for i, l in enumerate(body):
        m = re.search(r's_cbranch_\w+\s+(\.LBB\d+_\d+)', l) or re.search(r's_branch\s+(\.LBB\d+_\d+)', l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            seg = body[labels[m.group(1)]:i]
            n = sum('v_mfma' in x for x in seg)
            if n and (best is None or len(seg) < len(best)): best = seg
    tot = dict(mfma=sum('v_mfma' in x for x in body), scratch=sum('scratch_' in x for x in body))
    if best is None:
        print(name[:90], 'no MFMA loop', tot); continue
    def cnt(pat): return sum(bool(re.search(pat, x)) for x in best)
    ins = [x for x in best if re.match(r'^\s+[a-z]', x) and not x.strip().startswith(('.', ';'))]
    print(name[:100])
    print(f'   loop: {len(ins)} instr | mfma {cnt("v_mfma")} | ds_read {cnt("ds_read")} | glds {cnt("global_load_lds|buffer_load.* lds")} | valu {sum(bool(re.match(chr(92)+"s+v_", x)) and "v_mfma" not in x for x in best)} '
          f'| salu {sum(bool(re.match(chr(92)+"s+s_", x)) for x in best)} | waitcnt {cnt("s_waitcnt")} | barrier {cnt("s_barrier")} | scratch {cnt("scratch_")} | accvgpr {cnt("v_accvgpr")} || whole kernel: {tot}')
