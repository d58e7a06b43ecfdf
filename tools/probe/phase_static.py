"""Static instruction counts per phase of one eval_kernel instance (hipcc -S of a -DMP_PHASES build: the phase clocks'
s_memtime reads mark the boundaries).  usage: phase_static.py /tmp/mp_ph.s <mangled-prefix> [dump-dir]"""
import collections, re, sys
lines = open(sys.argv[1]).read().splitlines()
pref = sys.argv[2]
start = [i for i, l in enumerate(lines) if l.startswith(pref)][0]
end = start
while not lines[end].startswith('.Lfunc_end'): end += 1
seg, cur = [], []
for l in lines[start:end]:
    t = l.strip()
    if t.startswith(('s_memtime', 's_memrealtime')):
        seg.append(cur); cur = []
    else:
        cur.append(l)
seg.append(cur)
def stats(s):
    c = collections.Counter()
    for l in s:
        t = l.strip()
        if not t or t[0] in ';.' or t.endswith(':'): continue
        op = t.split()[0]
        if op.startswith('v_'):
            c['valu'] += 1
            if '_f64' in op and not op.startswith(('v_cmp', 'v_cvt')): c['f64'] += 1
            if op.startswith('v_cndmask'): c['cnd'] += 1
            if op.startswith('v_mov'): c['mov'] += 1
        elif op.startswith('s_'):
            if op.startswith('s_waitcnt'): c['wait'] += 1
            elif op.startswith('s_load'): c['sload'] += 1
            else: c['salu'] += 1
        elif op.startswith('ds_'): c['ds'] += 1
        elif op.startswith(('global', 'scratch', 'flat')): c['vmem'] += 1
    return dict(c)
for i, s in enumerate(seg):
    print(i, len(s), stats(s))
    if len(sys.argv) > 3:
        open(f"{sys.argv[3]}/seg{i}.s", "w").write("\n".join(s))
