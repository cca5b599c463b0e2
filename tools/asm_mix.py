"""Static instruction mix of the functions of a device assembly file (-save-temps *.s): tools/asm_mix.py file.s [name-substring ...]"""
import re, sys
from collections import Counter
s = open(sys.argv[1]).read().split('\n')
want = sys.argv[2:]
cur, funcs = None, {}
for l in s:
    m = re.match(r'^(_Z\w+):', l)
    if m:
        cur = m.group(1); funcs[cur] = []
        continue
    if l.startswith('.Lfunc_end'):
        cur = None
    if cur and l.startswith('\t') and not l.strip().startswith(('.', ';')):
        funcs[cur].append(l.strip())
KEYS = ('s_load_dwordx2', 's_load_dwordx4', 's_load_dword', 's_load_dwordx8', 's_load_dwordx16', 's_waitcnt', 'v_accvgpr_write_b32', 'v_accvgpr_read_b32',
        'global_load_dwordx2', 'global_load_dwordx4', 'global_store_dwordx2', 'scratch_load_dwordx2', 'scratch_store_dwordx2', 'ds_read_b64', 'ds_write_b64', 'v_cndmask_b32_e32', 'v_cndmask_b32_e64', 'v_mov_b32_e32',
        's_mov_b32', 's_mov_b64', 'v_fma_f64', 'v_mul_f64', 'v_add_f64', 'v_div_scale_f64', 'v_div_fmas_f64', 'v_div_fixup_f64', 'v_rcp_f64_e32', 'v_exp_f32_e32', 'v_ldexp_f64', 's_swappc_b64', 'v_readlane_b32', 'v_writelane_b32')
for name, ins in funcs.items():
    if want and not any(w in name for w in want):
        continue
    c = Counter(i.split()[0] for i in ins)
    print(name[:90], 'static instrs', len(ins))
    print('   ' + '  '.join(f'{k}={c[k]}' for k in KEYS if c.get(k)))
