"""Instruction-class mix per basic block of the kernels in a hipcc -S listing (dev tool).
usage: isa_mix.py file.s [name-substring]"""
import re
import sys

lines = open(sys.argv[1]).read().split('\n')
pat = sys.argv[2] if len(sys.argv) > 2 else ''
starts = [i for i, l in enumerate(lines) if re.match(r'^_Z\w+:', l) and pat in l]
for st in starts:
    end = next(i for i in range(st, len(lines)) if 's_endpgm' in lines[i])
    print(lines[st].split(':')[0][:90])
    blk, name = [], 'entry'

    def flush():
        ins = [l.strip().split()[0] for l in blk if l.strip() and not l.strip().startswith((';', '.'))]
        if len(ins) > 15:
            c = lambda p: sum(i.startswith(p) for i in ins)
            print(f"   {name[:14]:14s} total {len(ins):4d} mfma {c('v_mfma'):3d} valu {c('v_') - c('v_mfma'):4d} "
                  f"(exp {c('v_exp')}, mov {c('v_mov')}, pk {c('v_pk')}) salu {c('s_'):3d} ds {c('ds_'):3d} "
                  f"vmem {c('buffer_') + c('global_'):3d}")

    for l in lines[st + 1:end]:
        if re.match(r'^\.LBB\d+_\d+:', l):
            flush(); blk = []; name = l.split(':')[0]
        else:
            blk.append(l)
    flush()
