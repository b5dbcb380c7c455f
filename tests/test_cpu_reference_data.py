"""The oracle's operators against DATA produced by the reference's own Nek5000 discretisation.

The fixture is the base-flow field file of the reference's only integration test (cylinder, Re = 50, curved
mesh, E = 1996, lx1 = 6): a steady Navier-Stokes solution in Nek5000's Pn-Pn-2 discretisation.  If the oracle
restates that discretisation correctly, then with the ORACLE's operators
  * the discrete divergence D u on the lx2 = lx1 - 2 Gauss mesh vanishes to the accuracy of the stored solution
    (the collocated divergence on the velocity mesh does not: it is 1e7 times larger),
  * the steady momentum residual  (U.grad)U + nu A U - D^T p  (dealiased convection on lxd = 9 points, pressure
    taken back to the Gauss mesh) vanishes at all nodes off the domain boundary to the solver tolerance.
This pins GLL/GL nodes, geometry factors, the divergence / gradient pair, the Helmholtz stiffness, the dealiased
convective operator, the mass matrix and the gather-scatter of oracle/sem.py against the reference.
"""
import numpy as np

from oracle.sem import SEM
from refdata import load_cylinder


def test_reference_base_flow_is_discretely_solenoidal_and_steady():
    hm, ux, uy, p, re, lxd, interior = load_cylinder()
    sem = SEM(hm, lxd=lxd)
    u = [ux.reshape(sem.shape1), uy.reshape(sem.shape1)]
    # divergence on the pressure mesh
    div = sem.opdiv(u) / sem.bm2
    l2 = np.sqrt(np.sum(div ** 2 * sem.bm2) / sem.volvm2)
    g = [sem.gradm1(a) for a in u]
    col = g[0][0] + g[1][1]
    l2col = np.sqrt(np.sum(col ** 2 * sem.bm1) / sem.volvm1)
    assert l2 < 1e-10 and np.abs(div).max() < 1e-8, (l2, np.abs(div).max())
    assert l2col > 1e5 * l2                       # the check discriminates: a different divergence would fail
    # steady momentum residual
    p2 = sem.to_mesh2(p.reshape(sem.shape1))      # Nek writes the pressure interpolated to mesh 1
    N = sem.lns_conv_weak(u, u)                   # = 2 (U.grad) U
    gp = sem.opgradt(p2)
    inter = interior.reshape(sem.shape1)
    for i in range(2):
        r = sem.gs(0.5 * N[i] + (1.0 / re) * sem.axhelm_local(u[i], 1.0, 0.0) - gp[i]) * sem.binvm1
        scale = np.abs(sem.gs(0.5 * N[i]) * sem.binvm1)[inter].max()
        assert scale > 0.5
        assert np.sqrt(np.mean(r[inter] ** 2)) < 1e-6 * scale and np.abs(r[inter]).max() < 1e-5 * scale
    # without dealiasing-consistent quadrature (convection collocated on the GLL mesh) the residual is far larger
    conv_col = u[0] * g[0][0] + u[1] * g[0][1]
    r_col = sem.gs(sem.bm1 * conv_col + (1.0 / re) * sem.axhelm_local(u[0], 1.0, 0.0) - gp[0]) * sem.binvm1
    assert np.sqrt(np.mean(r_col[inter] ** 2)) > 100 * 1e-6
