import sys, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
from conftest import golden, model_spec
from parity_util import make_sampler, oracle_teacher_forced
from oracle import oracle as O
from klhr_amd._lib import DIAG_COLS as DC
col={k:i for i,k in enumerate(DC)}
for fn in ["g8_funnel10_s7_w1000.npz","g8_glmm45_s0_w100.npz"]:
    R=oracle_teacher_forced(O, fn, max_records=400); g,M=R['g'],R['M']
    s=make_sampler(R['stem'],R['data'],M,warmup=R['warmup'],**R['flags'])
    s.set('theta',g['theta_before'][:M]); s.set('rng',g['state_before'][:M])
    for f in R['fields']: s.set(f,R['before'][f])
    s.set('counters',R['cnt_before'])
    rho=s.random_direction()
    eta=s.fit(g['rho'][:M]); d=s.get('diag')
    od=R['diag_fit']
    err=(np.abs(eta-R['eta'])/(1+np.abs(R['eta']))).max(axis=1)
    idx=np.argsort(-err)[:12]
    print(fn)
    for i in idx:
        print(i, "err %.3g"%err[i], "gpu nfev", d[i,col['nfev1']], d[i,col['nfev2']], "st", d[i,col['status1']], d[i,col['status2']], "fb", d[i,col['fallbacks']], "fun2 %.12g"%d[i,col['fun2']],
              "| orc nfev", od[i]['nfev1'], od[i]['nfev2'], "st", od[i]['status1'], od[i]['status2'], "fb", od[i]['ls_fallbacks'], "fun2 %.12g"%od[i]['fun2'], "| ref nfev", g['nfev1'][i], g['nfev2'][i], "fun2 %.12g"%g['fun2'][i])
        print("    eta gpu", eta[i], "orc", R['eta'][i], "ref", g['eta'][i])
    fd=np.array([x['fun2'] for x in od]); 
    print("fun2 rel diff quantiles", np.quantile(np.abs(d[:,col['fun2']]-fd)/(1+np.abs(fd)), [0.5,0.9,0.99,1.0]))
    print("passes mean", d[:,col['passes']].mean(), "nfev1+nfev2 mean", (d[:,col['nfev1']]+d[:,col['nfev2']]).mean())
