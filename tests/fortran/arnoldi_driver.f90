!> Demo / test driver: what a neklab `userchk` + LightKrylov would do with the shim, written against the
!! ABSTRACT interfaces only (class(abstract_vector_rdp), class(abstract_exptA_linop_rdp)):
!!   bf -> exptA_linop(tau, bf); exptA%init()            (examples/cylinder/stability/direct/1cyl.usr:16-20)
!!   k steps of Arnoldi with per-vector dot / axpby       (LightKrylov's loop structure, SURVEY.md §3.1)
!! Input : mesh.bin written by tests/test_gpu_fortran.py (ldim, lx1, nelv, coordinates, labels, masks,
!!         base flow, start vector).  Output: the Hessenberg matrix on stdout, one entry per line.
program arnoldi_driver
   use iso_c_binding
   use LightKrylov, only: abstract_vector_rdp, abstract_exptA_linop_rdp
   use neklab_gpu                            ! round-1 module name: re-exports `neklab`
   implicit none
   integer :: ldim, lx1, nelv, lvn, lpn, k, kdim, i, j, u
   integer(c_int64_t), allocatable :: glo(:)
   real(dp), allocatable :: x(:), y(:), z(:), m1(:), m2(:), m3(:), bx(:), by(:), bz(:), vx(:), vy(:), vz(:), pr(:), t(:)
   real(dp) :: tau, re, dt, beta
   real(dp), allocatable :: H(:, :)
   type(nek_dvector), allocatable :: bf, Xb(:)
   type(exptA_linop), allocatable :: exptA
   type(nek_dvector) :: wrk

   open (newunit=u, file='mesh.bin', access='stream', form='unformatted', status='old')
   read (u) ldim, lx1, nelv, kdim
   read (u) tau, re, dt
   lvn = nelv*lx1**ldim
   lpn = nelv*(lx1 - 2)**ldim
   allocate (x(lvn), y(lvn), z(lvn), m1(lvn), m2(lvn), m3(lvn), bx(lvn), by(lvn), bz(lvn), vx(lvn), vy(lvn), vz(lvn))
   allocate (glo(lvn), pr(lpn), t(1))
   z = 0; m3 = 0; bz = 0; vz = 0; pr = 0; t = 0
   read (u) x, y
   if (ldim == 3) read (u) z
   read (u) glo
   read (u) m1, m2
   if (ldim == 3) read (u) m3
   read (u) bx, by
   if (ldim == 3) read (u) bz
   read (u) vx, vy
   if (ldim == 3) read (u) vz
   close (u)

   call neklab_gpu_init(0)
   call neklab_gpu_set_mesh(ldim, lx1, nelv, x, y, z, glo, m1, m2, m3, .false.)

   allocate (bf); call nek2vec(bf, bx, by, bz, pr, t)
   allocate (exptA)
   exptA%tau = tau
   exptA%baseflow = bf                       ! deep copy through defined assignment
   call set_cfg(exptA)                       ! this driver fills in the solver configuration itself
   call exptA%init()

   allocate (Xb(kdim + 1))
   call nek2vec(Xb(1), vx, vy, vz, pr, t)
   allocate (H(kdim + 1, kdim)); H = 0.0_dp
   beta = Xb(1)%norm(); call Xb(1)%scal(1.0_dp/beta)
   do k = 1, kdim
      call arnoldi_step(exptA, Xb, H, k)
   end do
   do j = 1, kdim
      do i = 1, kdim + 1
         write (*, '(A,I0,1X,I0,1X,ES24.16)') 'H ', i, j, H(i, j)
      end do
   end do
   ! assignment semantics: a copy must not alias
   wrk = Xb(1)
   call wrk%scal(2.0_dp)
   write (*, '(A,ES24.16)') 'ALIAS ', Xb(1)%norm()
   write (*, '(A,I0)') 'SIZE ', Xb(1)%get_size()
   block      ! object lifetimes of the reference's by-value vectors: move_alloc, and reallocation on assignment -- where the old
      !         elements are finalised AFTER the temporary holding their bitwise copies has been built (handle adoption)
      type(nek_dvector), allocatable :: Y(:), Z(:)
      type(nek_dvector) :: extra
      real(dp) :: n1, n2
      allocate (Y(2))
      Y(1) = Xb(1); Y(2) = Xb(2)
      call Y(2)%scal(3.0_dp)
      n1 = Y(1)%norm(); n2 = Y(2)%norm()
      call move_alloc(Y, Z)
      call Z(1)%scal(1.0_dp)                  ! an inout use after the move: still the owner, nothing cloned
      write (*, '(A,4ES24.16)') 'MOVE ', n1, n2, Z(1)%norm(), Z(2)%norm()
      extra = Xb(1); call extra%scal(5.0_dp)
      Z = [Z, extra]
      call Z(1)%scal(2.0_dp)                  ! first inout use of a moved element: adopts the handle its old self released
      call Z(3)%scal(2.0_dp)                  ! copy of the live `extra`: gets a clone, `extra` keeps its value
      write (*, '(A,I0,4ES24.16)') 'REALLOC ', size(Z), Z(1)%norm(), Z(2)%norm(), Z(3)%norm(), extra%norm()
      write (*, '(A,L1)') 'HASRST ', Z(1)%has_rst_fields()
      ! a moved element that has only been READ since the move, then the pool of released vectors is emptied (as an allocation
      ! failure or nlg_vec_pool_limit would): the element must still hold its data -- also when the reallocation grew the array in
      ! place, i.e. the copy stands at the address of its finalised original and is "the owner" by address
      extra = Xb(2)
      Z = [Z, extra]
      n1 = Z(2)%norm()
      block
         use neklab_gpu_capi, only: c_vec_pool_trim
         integer(c_int) :: rc
         rc = c_vec_pool_trim(c_null_ptr)
      end block
      call Z(2)%scal(2.0_dp)
      write (*, '(A,I0,3ES24.16,1X,L1)') 'TRIM ', size(Z), n1, Z(2)%norm(), Z(4)%norm(), Z(2)%has_rst_fields()
   end block
   deallocate (Xb, exptA, bf)
   call neklab_gpu_finalize()

contains

   subroutine set_cfg(A)
      type(exptA_linop), intent(inout) :: A
      A%cfg%tau = tau; A%cfg%re = re; A%cfg%cfl_limit = 0.5_dp
      A%cfg%vtol = 1.0e-13_dp; A%cfg%ptol = 1.0e-13_dp; A%cfg%dt = dt
      A%cfg%torder = 3; A%cfg%maxit_v = 400; A%cfg%maxit_p = 4000
      A%cfg%fixed_iters_v = 0; A%cfg%fixed_iters_p = 0; A%cfg%pprecond = 0; A%cfg%pproj = 1
      A%cfg%ifheat = 0; A%cfg%conductivity = 1.0_dp; A%cfg%rhocp = 1.0_dp; A%cfg%buoy = 0.0_dp
      A%cfg_set = .true.
   end subroutine

   !> one Arnoldi step written against the abstract API only (CGS2 with k separate dots and axpbys)
   subroutine arnoldi_step(A, X, Hm, kk)
      class(abstract_exptA_linop_rdp), intent(inout) :: A
      class(abstract_vector_rdp), intent(inout) :: X(:)
      real(dp), intent(inout) :: Hm(:, :)
      integer, intent(in) :: kk
      integer :: ii, pass
      real(dp) :: hij
      call A%matvec(X(kk), X(kk + 1))
      do pass = 1, 2
         block
            real(dp) :: hh(kk)
            do ii = 1, kk
               hh(ii) = X(ii)%dot(X(kk + 1))
            end do
            do ii = 1, kk
               call X(kk + 1)%axpby(-hh(ii), X(ii), 1.0_dp)
               Hm(ii, kk) = Hm(ii, kk) + hh(ii)
            end do
         end block
      end do
      hij = X(kk + 1)%norm()
      Hm(kk + 1, kk) = hij
      call X(kk + 1)%scal(1.0_dp/hij)
   end subroutine

end program arnoldi_driver
