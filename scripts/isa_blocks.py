"""Dev: per-basic-block instruction census of one kernel in a hipcc -S listing (MFMAs, scratch traffic, LDS-direct loads,
fragment reads): where do the spills sit?  Usage: python scripts/isa_blocks.py file.s KERNEL_NAME_SUBSTRING"""
import re
import sys

s = open(sys.argv[1]).read()
key = sys.argv[2]
m = re.search(r"^(\S*" + re.escape(key) + r"\S*): *;[^\n]*\n(.*?)s_endpgm", s, re.S | re.M)
print(m.group(1))
blk, stats, order = None, {}, []
for ln in m.group(2).split('\n'):
    if re.match(r'^\.LBB\d+_\d+:', ln):
        blk = ln.strip(); order.append(blk)
        stats[blk] = dict(n=0, mfma=0, scr_ld=0, scr_st=0, dma=0, ds_read=0, vmem=0, waitcnt=0)
    elif blk and ln.strip() and not ln.strip().startswith(';'):
        st = stats[blk]; st['n'] += 1
        st['mfma'] += 'v_mfma' in ln
        st['scr_ld'] += 'scratch_load' in ln
        st['scr_st'] += 'scratch_store' in ln
        st['dma'] += ('buffer_load' in ln and ' lds' in ln)
        st['ds_read'] += 'ds_read' in ln
        st['vmem'] += ('buffer_' in ln)
        st['waitcnt'] += 's_waitcnt' in ln
for b in order:
    if stats[b]['n'] > 12:
        print(b, stats[b])
