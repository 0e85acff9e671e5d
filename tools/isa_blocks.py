"""Per-basic-block instruction profile of one kernel in a hipcc -save-temps .s file (MFMA / LDS / global / scratch / lane moves).
    python tools/isa_blocks.py file.s kernel_substring [min_instructions]"""
import re
import sys

s = open(sys.argv[1]).read()
want = sys.argv[2]
minn = int(sys.argv[3]) if len(sys.argv) > 3 else 40
for f in re.split(r'\n(?=_Z\w+:)', s):
    name = f.split(':', 1)[0]
    if want not in name or '\n' in name:
        continue
    keys = ['n', 'mfma', 'valu', 'ds_r', 'ds_w', 'gl', 'gs', 'sst', 'sld', 'rdl', 'wrl', 'bar', 'wait']
    blk, cur, out = 'entry', dict.fromkeys(keys, 0), []
    for ln in f.split('\n'):
        m = re.match(r'^(\.LBB\d+_\d+):', ln)
        if m:
            out.append((blk, cur)); blk, cur = m.group(1), dict.fromkeys(keys, 0)
            continue
        t = ln.strip()
        if not t or t[0] in ';.' or t.endswith(':'):
            continue
        cur['n'] += 1
        op = t.split()[0]
        if op.startswith('v_mfma'): cur['mfma'] += 1
        elif op.startswith('v_readlane') or op.startswith('v_readfirstlane'): cur['rdl'] += 1
        elif op.startswith('v_writelane'): cur['wrl'] += 1
        elif op.startswith('v_'): cur['valu'] += 1
        elif op.startswith('ds_read') or op.startswith('ds_load'): cur['ds_r'] += 1
        elif op.startswith('ds_write') or op.startswith('ds_store'): cur['ds_w'] += 1
        elif op.startswith('global_load') or op.startswith('buffer_load'): cur['gl'] += 1
        elif op.startswith('global_store') or op.startswith('buffer_store'): cur['gs'] += 1
        elif op.startswith('scratch_store'): cur['sst'] += 1
        elif op.startswith('scratch_load'): cur['sld'] += 1
        elif op.startswith('s_barrier'): cur['bar'] += 1
        elif op.startswith('s_waitcnt'): cur['wait'] += 1
    out.append((blk, cur))
    print(name)
    tot = dict.fromkeys(keys, 0)
    for b, c in out:
        for k in keys: tot[k] += c[k]
        if c['n'] >= minn:
            print(f"  {b:12s}", ' '.join(f"{k}={c[k]}" for k in keys if c[k]))
    print("  total       ", ' '.join(f"{k}={tot[k]}" for k in keys if tot[k]))
