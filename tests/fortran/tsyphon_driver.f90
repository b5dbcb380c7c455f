!> The call sequence of a temperature-coupled neklab case against the shim: the body of `userchk` in
!! /root/reference/examples/thermosyphon/baseflow/tsyphon.usr:29-70 -- nek_system_temp + nek_jacobian_temp, Newton-Krylov
!! for the fixed point (newton_fixed_point_iteration, tol_mode = 2), then the stability of that fixed point through
!! exptA_linop_temp and eigs -- with the reference's type and procedure names.  Nek5000 is replaced by `case.bin` (written by
!! tests/test_gpu_fortran.py): the arrays of its commons and the parameters of its .par file; the flow is a small heated box.
program tsyphon_driver
   use iso_c_binding, only: c_int64_t
   use LightKrylov, only: zero_basis, eigs, save_eigenspectrum
   use neklab
   implicit none
   integer :: ldim, lx1, nelv, lvn, lpn, kdim, nev, u, device, info
   integer(c_int64_t), allocatable :: glo(:)
   real(dp), allocatable :: xm1(:), ym1(:), zm1(:), v1mask(:), v2mask(:), v3mask(:), tmask(:), vx(:), vy(:), vz(:), pr(:), t(:)
   real(dp) :: tau, re, vtol, ptol, conductivity, rhocp, buoy(3), endtime, tol, fnorm
   type(nek_system_temp), allocatable :: sys
   type(exptA_linop_temp), allocatable :: exptA_temp
   type(nek_dvector) :: bf, F
   type(nek_dvector), allocatable :: eigvecs(:)
   complex(dp), allocatable :: eigvals(:)
   real(dp), allocatable :: residuals(:)
   character(len=3) :: file_prefix

   open (newunit=u, file='case.bin', access='stream', form='unformatted', status='old')
   read (u) ldim, lx1, nelv, kdim, nev, device
   read (u) tau, re, vtol, ptol, conductivity, rhocp, buoy, endtime, tol
   lvn = nelv*lx1**ldim
   lpn = nelv*(lx1 - 2)**ldim
   allocate (xm1(lvn), ym1(lvn), zm1(lvn), v1mask(lvn), v2mask(lvn), v3mask(lvn), tmask(lvn), vx(lvn), vy(lvn), vz(lvn), glo(lvn), pr(lpn), t(lvn))
   zm1 = 0; v3mask = 0; vz = 0; pr = 0
   read (u) xm1, ym1
   read (u) glo
   read (u) v1mask, v2mask, tmask
   read (u) vx, vy, t                        ! "call load_fld('BF_Ra500_tsyphon0.f00001')": the initial guess
   close (u)

   call neklab_gpu_init(0)
   call neklab_gpu_set_mesh(ldim, lx1, nelv, xm1, ym1, zm1, glo, v1mask, v2mask, v3mask, .false., tmask=tmask)
   call neklab_gpu_set_case(re=re, torder=3, vtol=vtol, ptol=ptol, maxit_v=600, maxit_p=4000, ifheat=.true., conductivity=conductivity, &
                            rhocp=rhocp, buoy=buoy, endtime=endtime, dt=0.02_dp)      ! (a state at rest has no CFL number to derive dt from)
   device_eigs = device /= 0

   ! ---- tsyphon.usr:33-70 ---------------------------------------------------------------------------------------------
   ! Load initial guess
   call nek2vec(bf, vx, vy, vz, pr, t)

   ! Define system
   sys = nek_system_temp()
   sys%jacobian = nek_jacobian_temp()
   sys%jacobian%X = bf

   ! Compute fixed point
   call newton_fixed_point_iteration(sys, bf, tol, tol_mode=2)

   ! Outpost solution
   call outpost_dnek(bf, "BF_")

   ! Exponential propagator
   exptA_temp = exptA_linop_temp(tau, bf); call exptA_temp%init()
   allocate (eigvecs(nev)); call zero_basis(eigvecs)

   call eigs(exptA_temp, eigvecs, eigvals, residuals, info, kdim=kdim, write_intermediate=.true.)

   ! Transform eigenspectrum to continuous-time representation.
   eigvals = log(eigvals)/exptA_temp%tau
   file_prefix = "dir"
   call save_eigenspectrum(eigvals, residuals, trim(file_prefix)//"_eigenspectrum.npy")
   call outpost_dnek(eigvecs(:nev), file_prefix)
   ! ---------------------------------------------------------------------------------------------------------------------

   ! what the test reads back: the residual of the fixed point, the fixed point itself, the spectrum
   call sys%response(bf, F, 0.1_dp*tol)
   fnorm = F%norm()
   write (*, '(A,ES24.16)') 'FNORM ', fnorm
   write (*, '(A,ES24.16)') 'BFNORM ', bf%norm()
   call vec2nek(vx, vy, vz, pr, t, bf)
   write (*, '(A,3ES24.16)') 'BFMAX ', maxval(abs(vx)), maxval(abs(vy)), maxval(abs(t))
   write (*, '(A,I0)') 'MATVECS ', info
   deallocate (exptA_temp, eigvecs, sys)
   call neklab_gpu_finalize()
end program tsyphon_driver
