!> The resolvent operator through the shim with the reference's names: `resolvent_linop(omega, bf)` acting on `nek_zvector`s
!! (src/linops/neklab_linops.f90:198-205, src/linops/resolvent.f90:17-74), direct and adjoint.  Nek5000 is replaced by `case.bin`
!! (tests/test_gpu_fortran.py); the responses go to `response.bin` for the comparison with the Python mirror.
program resolvent_driver
   use iso_c_binding, only: c_int64_t
   use LightKrylov_AbstractVectors, only: abstract_vector_cdp
   use neklab
   implicit none
   integer :: ldim, lx1, nelv, lvn, lpn, u, adj
   integer(c_int64_t), allocatable :: glo(:)
   real(dp), allocatable :: xm1(:), ym1(:), zm1(:), v1mask(:), v2mask(:), v3mask(:), vx(:), vy(:), vz(:), pr(:), t(:), fre(:, :), fim(:, :), o(:, :)
   real(dp) :: omega, re, vtol, ptol
   type(nek_dvector) :: bf
   type(nek_zvector) :: f, q
   type(resolvent_linop) :: R

   open (newunit=u, file='case.bin', access='stream', form='unformatted', status='old')
   read (u) ldim, lx1, nelv, adj
   read (u) omega, re, vtol, ptol
   lvn = nelv*lx1**ldim
   lpn = nelv*(lx1 - 2)**ldim
   allocate (xm1(lvn), ym1(lvn), zm1(lvn), v1mask(lvn), v2mask(lvn), v3mask(lvn), vx(lvn), vy(lvn), vz(lvn), glo(lvn), pr(lpn), t(1), fre(lvn, 2), fim(lvn, 2), o(lvn, 4))
   zm1 = 0; v3mask = 0; vz = 0; pr = 0; t = 0
   read (u) xm1, ym1
   read (u) glo
   read (u) v1mask, v2mask
   read (u) vx, vy
   read (u) fre, fim
   close (u)

   call neklab_gpu_init(0)
   call neklab_gpu_set_mesh(ldim, lx1, nelv, xm1, ym1, zm1, glo, v1mask, v2mask, v3mask, .false.)
   call neklab_gpu_set_case(re=re, torder=3, vtol=vtol, ptol=ptol, maxit_v=400, maxit_p=4000)

   call nek2vec(bf, vx, vy, vz, pr, t)
   call nek2vec(f%re, fre(:, 1), fre(:, 2), vz, pr, t)
   call nek2vec(f%im, fim(:, 1), fim(:, 2), vz, pr, t)
   R = resolvent_linop(omega, bf)
   if (adj /= 0) then
      call R%rmatvec(f, q)
   else
      call R%matvec(f, q)
   end if
   call vec2nek(o(:, 1), o(:, 2), vz, pr, t, q%re)
   call vec2nek(o(:, 3), o(:, 4), vz, pr, t, q%im)
   open (newunit=u, file='response.bin', access='stream', form='unformatted', status='replace')
   write (u) o
   close (u)
   write (*, '(A,2ES24.16)') 'QNORM ', q%norm(), f%norm()
   call neklab_gpu_finalize()
end program resolvent_driver
