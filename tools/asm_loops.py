"""Innermost loops of one function of a device assembly file and their instruction mix: tools/asm_loops.py file.s name-substring [min-instrs]
(a loop = a label and the nearest later backward branch to it with no other label in between)"""
import re, sys
from collections import Counter
s = open(sys.argv[1]).read().split('\n')
want = sys.argv[2]; minn = int(sys.argv[3]) if len(sys.argv) > 3 else 20
cur, body = None, []
for l in s:
    m = re.match(r'^(_Z\w+):', l)
    if m:
        cur = m.group(1); continue
    if l.startswith('.Lfunc_end'):
        cur = None
    if cur and want in cur:
        body.append(l)
labels = {}
for i, l in enumerate(body):
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m: labels[m.group(1)] = i
for i, l in enumerate(body):
    m = re.match(r'\s+s_cbranch_\w+\s+(\.LBB\d+_\d+)', l)
    if not m or m.group(1) not in labels: continue
    a = labels[m.group(1)]
    if a >= i: continue
    if any(labels[k] for k in labels if a < labels[k] < i): continue        # not innermost / has inner labels
    ins = [x.strip() for x in body[a + 1:i + 1] if x.startswith('\t') and not x.strip().startswith(('.', ';'))]
    if len(ins) < minn: continue
    c = Counter(x.split()[0] for x in ins)
    valu = sum(v for k, v in c.items() if k.startswith('v_'))
    f64 = sum(v for k, v in c.items() if k in ('v_fma_f64', 'v_mul_f64', 'v_add_f64', 'v_fmac_f64_e32'))
    print(f"{m.group(1)} lines {a}-{i}: {len(ins)} instrs, VALU {valu} (f64 arith {f64}), SALU {sum(v for k, v in c.items() if k.startswith('s_') and k != 's_waitcnt')}, waitcnt {c['s_waitcnt']}, LDS {sum(v for k, v in c.items() if k.startswith('ds_'))}, VMEM {sum(v for k, v in c.items() if k.startswith(('global_', 'scratch_', 'flat_', 'buffer_')))}")
    print("    " + "  ".join(f"{k}={v}" for k, v in c.most_common(14)))
