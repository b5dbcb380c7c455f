#!/usr/bin/env python3
"""Extracts a DATA fixture from the reference's own test case (no reference source is copied):
/root/reference/examples/cylinder/stability/direct/BF_1cyl0.f00001 -- the Nek5000 base-flow field file
(format "#std 8 6 6 1 1996 ..."; fields X, U, P in fp64) that the reference's only integration test loads
(test/neklabTests.py:16-47, 1cyl.usr:15).  It is the output of Nek5000's own discretisation: a steady
Navier-Stokes solution at Re = 50 on the curved cylinder mesh.  tests/test_cpu_reference_data.py checks the
oracle's operators against it (discrete divergence ~1e-11, steady momentum residual ~1e-7).

Round 3 adds the two other base-flow files the reference ships next to a case that uses the hot path:
  examples/back_fstep/transient_growth/BF_bfs0.f00001 (+ bfs.re2: midside-node edges 'm', boundary ids; the tags the ids stand
      for are the four setbc calls of bfs.usr:112-115) -- the base flow of the reference's transient-growth (svds) case,
      E = 2760, lx1 = 6, Re = 600 (bfs.par:31), run with an explicit filter of weight 0.01 (bfs.par:16-18);
  examples/rayBen/baseflow/BF_rayBen0.f00001 (+ rayBen.re2) -- fields XUPT on a 10 x 4 box, lx1 = 10.
-> reference_bfs_baseflow.npz, reference_rayben_baseflow.npz (numbers only: coordinates, fields, element map, boundary records).

  examples/cylinder/newton/Re40_fixed_point/BF.fld -- the initial guess of the reference's Newton-Krylov example (a DNS snapshot at
      Re = 40, t = 80, fp32, on the mesh of the stability case) together with the numbers read off the convergence plot the reference
      ships next to it (residual.png: Newton residuals and GMRES residual histories of that very run) -> reference_cyl_re40_guess.npz.

  examples/thermosyphon/baseflow/tsyphon.re2 -- the annulus 1 <= r <= 2 (8 x 32 elements with circular-arc sides, genbox + curved sides;
      velocity 'W', temperature 't' on both walls, periodic in the angle) of the reference's temperature-coupled Newton example, the
      parameters of tsyphon.par / tsyphon.usr, and the Newton residuals read off the convergence plot shipped with the case
      (residual.png; its initial guess BF_Ra500_tsyphon0.f00001 is NOT shipped) -> reference_tsyphon_mesh.npz.

Run from the repo root (needs /root/reference):  python tests/golden/make_reference_fixture.py
"""
import os

import numpy as np

SRC = "/root/reference/examples/cylinder/stability/direct/BF_1cyl0.f00001"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_cyl_baseflow.npz")


def read_fld(path):
    raw = open(path, "rb").read()
    hdr = raw[:132].decode().split()
    wd, nx, ny, nz, nel = int(hdr[1]), int(hdr[2]), int(hdr[3]), int(hdr[4]), int(hdr[5])
    assert hdr[0] == "#std" and hdr[11] == "XUP" and nz == 1
    assert abs(np.frombuffer(raw[132:136], dtype=np.float32)[0] - 6.54321) < 1e-5    # endianness tag
    off = 136 + 4 * nel
    dt = np.float64 if wd == 8 else np.float32
    npt = nx * ny * nz

    def rd(nc):
        nonlocal off
        a = np.frombuffer(raw[off: off + wd * nel * nc * npt], dtype=dt).reshape(nel, nc, npt)
        off += wd * nel * nc * npt
        return a.astype(np.float64)

    X, U, P = rd(2), rd(2), rd(1)
    return nx, X, U, P


if __name__ == "__main__":
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    from neklab_amd.nekio import read_fld as rf, read_re2_bcs
    n, X, U, P = read_fld(SRC)
    elmap = rf(SRC)["elmap"]
    nel, dim, bcs = read_re2_bcs(SRC.replace("BF_1cyl0.f00001", "1cyl.re2"))
    assert nel == X.shape[0] and dim == 2
    np.savez_compressed(OUT, n=np.array(n), x=X[:, 0], y=X[:, 1], ux=U[:, 0], uy=U[:, 1], p=P[:, 0],
                        re=np.array(50.0), lxd=np.array(9), elmap=elmap,
                        bc_elem=np.array([b[0] for b in bcs]), bc_face=np.array([b[1] for b in bcs]),
                        bc_tag=np.array([b[2] for b in bcs]))
    print(OUT, os.path.getsize(OUT), "bytes")
    # the mesh files of the same case as data: vertex coordinates and circular-arc records of 1cyl.re2, global vertex
    # ids and bisection keys of 1cyl.ma2 (what Nek5000's genxyz / set_vert start from)
    from neklab_amd.nekio import read_ma2, read_re2
    r = read_re2(SRC.replace("BF_1cyl0.f00001", "1cyl.re2"))
    m = read_ma2(SRC.replace("BF_1cyl0.f00001", "1cyl.ma2"))
    assert all(c[3] == "C" for c in r["curves"])
    out2 = OUT.replace("reference_cyl_baseflow", "reference_cyl_mesh")
    np.savez_compressed(out2, xc=r["xc"], yc=r["yc"], curve_elem=np.array([c[0] for c in r["curves"]]),
                        curve_edge=np.array([c[1] for c in r["curves"]]), curve_par=np.array([c[2] for c in r["curves"]]),
                        vert=m["vert"], pmap=m["pmap"])
    print(out2, os.path.getsize(out2), "bytes")

    # ---- backward-facing step (transient growth) ------------------------------------------------------------------------
    from neklab_amd.nekio import re2_gll_coords
    bdir = "/root/reference/examples/back_fstep/transient_growth/"
    fb = rf(bdir + "BF_bfs0.f00001")
    rb = read_re2(bdir + "bfs.re2")
    raw = open(bdir + "bfs.re2", "rb").read()
    off = 84 + 8 * (1 + 2 * 4) * rb["nel"]
    nc = int(np.frombuffer(raw[off: off + 8], dtype=np.float64)[0])
    off += 8 + 64 * nc
    nbc = int(np.frombuffer(raw[off: off + 8], dtype=np.float64)[0])
    off += 8
    bid = []                       # gmsh boundary id of every record (fifth parameter; the tag in the file is 'MSH')
    for _ in range(nbc):
        bid.append(int(np.frombuffer(raw[off: off + 56], dtype=np.float64)[6]))
        off += 64
    assert all(np.array_equal(fb[k].astype(np.float32).astype(np.float64), fb[k]) for k in ("ux", "uy"))
    out3 = OUT.replace("reference_cyl_baseflow", "reference_bfs_baseflow")
    np.savez_compressed(out3, n=np.array(6), x=fb["x"], y=fb["y"], ux=fb["ux"].astype(np.float32), uy=fb["uy"].astype(np.float32),
                        p=fb["p"], re=np.array(600.0), lxd=np.array(9), elmap=fb["elmap"],
                        bc_elem=np.array([b[0] for b in rb["bcs"]]), bc_face=np.array([b[1] for b in rb["bcs"]]), bc_id=np.array(bid),
                        # bfs.usr:112-115  setbc(5,1,'W  '), setbc(2,1,'v  '), setbc(3,1,'v  '), setbc(4,1,'SYM')
                        id_list=np.array([5, 2, 3, 4]), id_tag=np.array(["W", "v", "v", "SYM"]),
                        xc=rb["xc"], yc=rb["yc"], curve_elem=np.array([c[0] for c in rb["curves"]]),
                        curve_edge=np.array([c[1] for c in rb["curves"]]), curve_par=np.array([c[2] for c in rb["curves"]]))
    print(out3, os.path.getsize(out3), "bytes")
    # ---- Rayleigh-Benard box ----------------------------------------------------------------------------------------------
    rdir = "/root/reference/examples/rayBen/baseflow/"
    fr = rf(rdir + "BF_rayBen0.f00001")
    rr = read_re2(rdir + "rayBen.re2")
    out4 = OUT.replace("reference_cyl_baseflow", "reference_rayben_baseflow")
    np.savez_compressed(out4, n=np.array(10), x=fr["x"], y=fr["y"], ux=fr["ux"], uy=fr["uy"], p=fr["p"], t=fr["t"], lxd=np.array(15),
                        elmap=fr["elmap"], bc_elem=np.array([b[0] for b in rr["bcs"]]), bc_face=np.array([b[1] for b in rr["bcs"]]),
                        bc_tag=np.array([b[2] for b in rr["bcs"]]),
                        # rayBen.par:5-6 userParam05 = Pr, userParam06 = Ra; rayBen.usr:98 ffy = temp * Ra * Pr
                        prandtl=np.array(1.0), rayleigh=np.array(1900.0))
    print(out4, os.path.getsize(out4), "bytes")
    # ---- Newton-Krylov example at Re = 40: initial guess + the published convergence history ----------------------------------
    ndir = "/root/reference/examples/cylinder/newton/Re40_fixed_point/"
    assert open(ndir + "1cyl.re2", "rb").read() == open(SRC.replace("BF_1cyl0.f00001", "1cyl.re2"), "rb").read()   # same mesh as the stability case
    fn = rf(ndir + "BF.fld")
    # the file was written with another element order: bring it to the order of reference_cyl_baseflow.npz (same global elements)
    pos = {int(g): k for k, g in enumerate(fn["elmap"])}
    perm = np.array([pos[int(g)] for g in elmap])
    for k in ("x", "y", "ux", "uy", "p"):
        fn[k] = fn[k][perm]
    assert np.max(np.abs(fn["x"] - X[:, 0])) < 1e-5 and np.max(np.abs(fn["y"] - X[:, 1])) < 1e-5    # same points (fp32 coordinates)
    out5 = OUT.replace("reference_cyl_baseflow", "reference_cyl_re40_guess")
    import digitize_reference_plots as dgp
    dg = dgp.cylinder_re40()
    assert [len(v) for v in dg["gmres"]] == [21, 19, 3] and len(dg["newton"]) == 3 and dg["rel_err"] < 0.01
    np.savez_compressed(out5, ux=fn["ux"].astype(np.float32), uy=fn["uy"].astype(np.float32), p=fn["p"].astype(np.float32),
                        re=np.array(40.0),                         # 1cyl.par: viscosity = -40
                        tau=np.array(1.0), dt=np.array(0.009),     # endTime = 1.0, dt = 0.009, bdf3
                        solver_tol=np.array(1.0e-8),               # [PRESSURE] / [VELOCITY] residualTol
                        newton_tol=np.array(1.0e-6),               # 1cyl.usr: tol = 1.0e-6_dp
                        # digitised from residual.png by tests/golden/digitize_reference_plots.py (marker centroids against the tick marks of the
                        # log axes; one-sigma relative error `plot_rel_err`, about 0.8 %): Newton residuals at the start of steps 1 - 3 ...
                        plot_newton_residuals=dg["newton"], plot_rel_err=np.array(dg["rel_err"]),
                        # ... the GMRES residuals of every Newton step ("init step", then the inner steps) and the inner-step counts
                        plot_gmres_step1=dg["gmres"][0], plot_gmres_step2=dg["gmres"][1], plot_gmres_step3=dg["gmres"][2],
                        # the same number read on both axes (GMRES init residual of step k / Newton residual of step k - 1): the digitisation's own check
                        plot_cross_check=dg["cross_check"],
                        plot_gmres_inner_steps=np.array([len(v) - 1 for v in dg["gmres"]]))
    print(out5, os.path.getsize(out5), "bytes")
    # ---- thermosyphon: mesh, parameters, published Newton residuals ---------------------------------------------------------
    tdir = "/root/reference/examples/thermosyphon/baseflow/"
    rt = read_re2(tdir + "tsyphon.re2")
    assert rt["nel"] == 256 and all(c[3] == "C" for c in rt["curves"]) and len(rt["bcs_fields"]) == 2
    out6 = OUT.replace("reference_cyl_baseflow", "reference_tsyphon_mesh")
    dgt = dgp.thermosyphon()
    assert len(dgt["newton"]) == 9
    np.savez_compressed(out6, xc=rt["xc"], yc=rt["yc"], curve_elem=np.array([c[0] for c in rt["curves"]]),
                        curve_edge=np.array([c[1] for c in rt["curves"]]), curve_par=np.array([c[2] for c in rt["curves"]]),
                        bc_elem=np.array([b[0] for b in rt["bcs_fields"][0]]), bc_face=np.array([b[1] for b in rt["bcs_fields"][0]]),
                        bc_tag=np.array([b[2] for b in rt["bcs_fields"][0]]), tbc_tag=np.array([b[2] for b in rt["bcs_fields"][1]]),
                        n=np.array(8), lxd=np.array(12),                         # SIZE: lx1 = 8, lxd = 12
                        # tsyphon.par: viscosity = -5 (nu = 1/5), conductivity = 1, rhocp = 1, userparam06 = 510 (Ra), endTime = 1, bdf3,
                        # tolerances 1e-8; tsyphon.usr: ffy = T * |param(2) * uparam(6)| = T nu Ra, wall temperature 0.5 (1 + tanh(-20 y)),
                        # newton_fixed_point_iteration(sys, bf, 1e-6, tol_mode = 2) from the Ra = 500 base flow
                        nu=np.array(0.2), conductivity=np.array(1.0), rhocp=np.array(1.0), rayleigh=np.array(510.0), rayleigh_guess=np.array(500.0),
                        tau=np.array(1.0), newton_tol=np.array(1.0e-6),
                        # digitised from residual.png (digitize_reference_plots.thermosyphon; left axes only -- the GMRES curves of its nine
                        # Newton steps overlap too much for marker segmentation): Newton residuals at the start of steps 1 - 9
                        plot_newton_residuals=dgt["newton"], plot_rel_err=np.array(dgt["rel_err"]),
                        # GMRES of Newton step 1: residual at the start and after inner steps 1 - 3; tolerance lines of steps 1, 3, 4 (dashed)
                        plot_gmres_step1=np.array([4.2e-1, 6.6e-3, 1.75e-4, 2.45e-5]), plot_gmres_tol=np.array([1.0e-4, 6.6e-6, 4.0e-7]))
    print(out6, os.path.getsize(out6), "bytes")
