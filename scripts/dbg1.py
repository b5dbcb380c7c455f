import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np
from neklab_amd import host
from test_gpu_linop import setup_case, load_pair
from oracle.lns import ExptA, LNSConfig
ctx = host.Context(0)
for dim in (2,):
    hm, sem, gm, oA, gA, rng = setup_case(ctx, dim)
    print('ediag err', np.abs(gm.get('ediag',2)-sem.e_diag().ravel()).max()/np.abs(sem.e_diag()).max())
    hd = sem.gs(sem.helm_diag_local(0.02, 30.0)); print('hdiag err', np.abs(gm.get('hdiag:0.02:30.0')-hd.ravel()).max()/np.abs(hd).max())
    ov, gv = load_pair(sem, gm, rng)
    for (to, fv, fp, ns) in [(1,12,2,1),(1,12,3,1),(1,12,5,1),(1,12,10,1),(1,12,20,1),(1,12,30,1),(1,12,400,1),(3,30,600,4)]:
        kw = dict(re=50., torder=to, tau=0.01*ns, dt=0.01, vtol=1e-13, ptol=1e-13, fixed_iters_v=fv, fixed_iters_p=fp)
        oA = ExptA(sem, oA.U, LNSConfig(**kw))
        gb = host.nek_dvector(gm)
        for i in range(dim): gb.set_field(i, oA.U[i])
        gA = host.exptA_linop(kw['tau'], gb, **{k:v for k,v in kw.items() if k!='tau'}); gA.init()
        gout = host.nek_dvector(gm); gA.matvec(gv, gout); oout = oA.matvec(ov)
        e = [np.abs(gout.get_field(i)-oout.v[i].ravel()).max() for i in range(dim)] + [np.abs(gout.get_field(3)-oout.pr.ravel()).max()]
        e1 = [np.abs(gout.get_field(i,1)-oout.v_rst[0][i].ravel()).max() for i in range(dim)] if to>1 else []
        print(to,fv,fp,ns, gA.info()['nsteps'], oA.nsteps, ['%.2e'%x for x in e], ['%.2e'%x for x in e1], 'scale', np.abs(oout.v[0]).max(), np.abs(oout.pr).max())
