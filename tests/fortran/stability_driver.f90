!> What a neklab case file does, against the shim: the body of `userchk` in
!! /root/reference/examples/cylinder/stability/direct/1cyl.usr:13-24 followed by the reference's own driver
!! (linear_stability_analysis_fixed_point, src/neklab_analysis.f90:38-105, restated in neklab_analysis.f90 of this
!! directory with the same call sequence).  Nek5000 is replaced by `case.bin` (written by tests/test_gpu_fortran.py):
!! the arrays a Nek5000 host holds in its commons (coordinates, global numbering, masks, the loaded base flow) and the
!! case parameters it reads from the .par file.
program stability_driver
   use iso_c_binding, only: c_int64_t
   use neklab
   implicit none
   integer :: ldim, lx1, nelv, lvn, lpn, kdim, nev, u, device, i
   integer(c_int64_t), allocatable :: glo(:)
   real(dp), allocatable :: xm1(:), ym1(:), zm1(:), v1mask(:), v2mask(:), v3mask(:), vx(:), vy(:), vz(:), pr(:), t(:)
   real(dp) :: tau, re, vtol, ptol
   type(nek_dvector), allocatable :: bf, X0
   type(exptA_linop), allocatable :: exptA

   open (newunit=u, file='case.bin', access='stream', form='unformatted', status='old')
   read (u) ldim, lx1, nelv, kdim, nev, device
   read (u) tau, re, vtol, ptol
   lvn = nelv*lx1**ldim
   lpn = nelv*(lx1 - 2)**ldim
   allocate (xm1(lvn), ym1(lvn), zm1(lvn), v1mask(lvn), v2mask(lvn), v3mask(lvn), vx(lvn), vy(lvn), vz(lvn), glo(lvn), pr(lpn), t(lvn))
   zm1 = 0; v3mask = 0; vz = 0; pr = 0; t = 0
   read (u) xm1, ym1
   if (ldim == 3) read (u) zm1
   read (u) glo
   read (u) v1mask, v2mask
   if (ldim == 3) read (u) v3mask
   read (u) vx, vy                          ! "call load_fld('BF_1cyl0.f00001')"
   if (ldim == 3) read (u) vz
   close (u)

   ! what a Nek5000 host does once (INTEGRATION.md): hand over the mesh and the case parameters of its commons
   call neklab_gpu_init(0)
   call neklab_gpu_set_mesh(ldim, lx1, nelv, xm1, ym1, zm1, glo, v1mask, v2mask, v3mask, .false.)
   call neklab_gpu_set_case(re=re, torder=3, vtol=vtol, ptol=ptol, maxit_v=400, maxit_p=4000)
   device_eigs = device /= 0

   ! ---- 1cyl.usr:13-24 ------------------------------------------------------------------------------------------------
   ! Load baseflow.
   allocate (bf); call nek2vec(bf, vx, vy, vz, pr, t)
   call vec2nek(vx, vy, vz, pr, t, bf)

   ! Exponential propagator.
   exptA = exptA_linop(tau, bf); call exptA%init()

   ! Stability analysis.  (1cyl.usr:23 lets eigs draw the start vector; it is drawn here and handed over as X0 so that the
   ! LightKrylov loop and the device block path start from the same vector: the leading Ritz values of a run depend on
   ! the start vector at the 1e-4 .. 1e-3 level, see DESIGN.md "start vector and restart history")
   allocate (X0); call X0%zero(); call X0%rand(.true.)
   call linear_stability_analysis_fixed_point(exptA, kdim, nev, X0=X0)
   ! ---------------------------------------------------------------------------------------------------------------------

   write (*, '(A,I0)') 'NSTEPS ', exptA%nsteps()
   write (*, '(A,I0)') 'SIZE ', bf%get_size()
   ! the base flow must have survived being copied into the operator (by-value semantics of the reference's types)
   write (*, '(A,ES24.16)') 'BFNORM ', bf%norm()
   write (*, '(A,ES24.16)') 'OPBFNORM ', exptA%baseflow%norm()
   block      ! sourced allocation + assignment: copies must not alias (SURVEY.md 7.3 item 5)
      type(nek_dvector), allocatable :: X(:)
      type(nek_dvector) :: w
      allocate (X(2), source=bf)
      call X(1)%scal(2.0_dp)
      w = X(2)
      call w%scal(3.0_dp)
      write (*, '(A,3ES24.16)') 'COPIES ', bf%norm(), X(1)%norm(), w%norm()
   end block
   deallocate (exptA, bf, X0)
   call neklab_gpu_finalize()
end program stability_driver
