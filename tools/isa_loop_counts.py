"""VALU / MFMA / LDS / VMEM / SALU instruction counts of every MFMA-carrying loop block of a kernel in a .s file
(hipcc -S --cuda-device-only): the per-step instruction budget of the recurrence kernels.
python tools/isa_loop_counts.py file.s mangled_kernel_name [min_mfma]"""
import collections
import sys

s = open(sys.argv[1]).read()
a = s.index(sys.argv[2] + ':')
b = s.index('.end_amdhsa_kernel', a)
floor = int(sys.argv[3]) if len(sys.argv) > 3 else 70
blocks, cur, name = [], [], 'entry'
for ln in s[a:b].split('\n'):
    t = ln.strip()
    if t.endswith(':') and t.startswith('.LBB'):
        blocks.append((name, cur)); name, cur = t, []
    else:
        cur.append(t)
blocks.append((name, cur))
for name, ins in blocks:
    ops = [i.split()[0] for i in ins if i and not i.startswith(';') and not i.startswith('.')]
    if sum(1 for o in ops if o.startswith('v_mfma')) < floor:
        continue
    c = collections.Counter()
    for o in ops:
        if o.startswith('v_mfma'):
            c['mfma'] += 1
        elif o.startswith('v_'):
            c['valu'] += 1
            if 'f64' in o or o.startswith('v_mov_b64'):
                c['valu64'] += 1
            c[o] += 1
        elif o.startswith('ds_'):
            c['ds'] += 1
        elif o.startswith('global_'):
            c['vmem'] += 1
        elif o.startswith('s_'):
            c['salu'] += 1
    print(name, 'mfma', c['mfma'], 'valu', c['valu'], '64-bit', c['valu64'], 'ds', c['ds'], 'vmem', c['vmem'], 'salu', c['salu'])
    print('   ', dict(sorted((k, v) for k, v in c.items() if k.startswith('v_') and not k.startswith('v_mfma'))))
