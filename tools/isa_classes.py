"""Condensed view of a kernel's ISA: run lengths of instruction classes (MFMA / TRANS / valu / ds / vmem /
waits / barriers / branches), to see what the scheduler interleaved.
python tools/isa_classes.py file.s mangled_kernel_name"""
import sys

s = open(sys.argv[1]).read()
a = s.index(sys.argv[2] + ':')
b = s.index('.end_amdhsa_kernel', a)
out, prev, cnt = [], None, 0
for ln in s[a:b].split('\n'):
    ln = ln.strip()
    if not ln or ln.startswith(';') or ln.startswith('.') and not ln.startswith('.LBB'):
        continue
    m = ln.split()[0]
    if m.endswith(':'):
        cls = '\n' + m
    elif m.startswith('v_mfma'):
        cls = 'MFMA'
    elif m.startswith('v_exp') or m.startswith('v_rcp'):
        cls = 'TRANS'
    elif m.startswith('v_'):
        cls = 'valu'
    elif m.startswith('ds_'):
        cls = 'ds'
    elif m.startswith('s_barrier'):
        cls = 'BARRIER'
    elif m.startswith('s_waitcnt'):
        cls = 'wait'
    elif m.startswith('global_') or m.startswith('buffer_'):
        cls = 'vmem'
    elif m.startswith('s_cbranch') or m.startswith('s_branch'):
        cls = 'BR'
    else:
        cls = 's'
    if cls == prev:
        cnt += 1
    else:
        if prev:
            out.append(prev + ('x%d' % cnt if cnt > 1 else ''))
        prev, cnt = cls, 1
out.append(prev + ('x%d' % cnt if cnt > 1 else ''))
print(' '.join(out))
