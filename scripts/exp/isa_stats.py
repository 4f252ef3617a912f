#!/usr/bin/env python3
"""Register / scratch table of the kernels in a -save-temps .s file, and a run-length view of one kernel's instruction classes."""
import re, sys, textwrap
s = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else None
for m in re.finditer(r'\.name:\s+(\S+)\n(.*?)\.wavefront_size', s, re.S):
    nm, blk = m.group(1), m.group(0)
    g = lambda k: re.search(r'\.' + k + r':\s+(\d+)', blk)
    if 'kernel' in nm and (not pat or pat in nm):
        print(nm, 'vgpr', g('vgpr_count').group(1), 'scratch', g('private_segment_fixed_size').group(1), 'sgpr', g('sgpr_count').group(1))
if len(sys.argv) > 3:
    i = s.index(sys.argv[3] + ':')
    j = s.index('.Lfunc_end', i)
    def cls(op):
        if op.startswith('v_rcp_f32'): return 'RCP'
        if op.startswith('v_mfma'): return 'MFMA'
        if op.startswith('v_log') or op.startswith('v_exp'): return 'LOG'
        if op.startswith('v_pk_'): return 'PK'
        if op.startswith('ds_read'): return 'DSR'
        if op.startswith('ds_write'): return 'DSW'
        if op.startswith('global_load_lds'): return 'DMA'
        if 'f64' in op: return 'F64'
        if op.startswith('s_barrier'): return 'BARRIER'
        if op.startswith('s_waitcnt'): return 'WAIT'
        if op.startswith('s_cbranch') or op.startswith('s_branch'): return 'BR'
        if 'dpp' in op: return 'DPP'
        if op.startswith('v_readlane'): return 'RDLANE'
        if op.startswith('s_'): return 's'
        if op.startswith('v_'): return 'v'
        return op
    out, prev, cnt = [], None, 0
    for ln in s[i:j].splitlines():
        t = ln.strip()
        if not t or t.startswith(';') or t.startswith('.'):
            if t.startswith('.LBB'):
                if prev: out.append(f"{prev}x{cnt}" if cnt > 1 else prev)
                out.append('|'); prev, cnt = None, 0
            continue
        c = cls(t.split()[0])
        if c == prev: cnt += 1
        else:
            if prev: out.append(f"{prev}x{cnt}" if cnt > 1 else prev)
            prev, cnt = c, 1
    if prev: out.append(f"{prev}x{cnt}" if cnt > 1 else prev)
    print('\n'.join(textwrap.wrap(' '.join(out), 220)))
