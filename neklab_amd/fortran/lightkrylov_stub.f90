!> Minimal stand-in for the parts of LightKrylov that neklab's hot path uses.
!!
!! LightKrylov (nekStab/LightKrylov @ main, un-pinned, /root/reference/LightKrylov_setup.sh:55-57) is not available in
!! this image.  What is reproduced here, with signatures inferred from how the reference implements / calls them:
!!   abstract_vector_rdp : zero, rand, scal, axpby, dot, get_size (deferred) + norm, sub, add
!!       /root/reference/src/vectors/neklab_vectors.f90:39-44 and interfaces :65-93
!!   abstract_linop_rdp / abstract_exptA_linop_rdp : matvec, rmatvec, %tau, finalize_timer
!!       /root/reference/src/linops/neklab_linops.f90:35-62, src/neklab_analysis.f90:84,98
!!   zero_basis, eigs, save_eigenspectrum  as called at /root/reference/src/neklab_analysis.f90:77-90
!!   type_error, stop_error (LightKrylov_Logger) as called at src/vectors/real_vectors.f90:202-204
!! `eigs` here is a plain Arnoldi iteration written against the ABSTRACT interfaces only -- k separate dot / axpby calls
!! per Gram-Schmidt pass, exactly the loop structure LightKrylov imposes on neklab (SURVEY.md 3.1) -- with the Ritz values
!! of the Hessenberg matrix from the library's dense helper.  It is test scaffolding for the drop-in boundary, not the
!! product's eigensolver (that is nlg_eigs, bound as `nek_eigs` in neklab_linops).  With the real LightKrylov on the
!! module path this file is simply left out of the build.
module LightKrylov_Logger
   implicit none
   private
   public :: type_error, stop_error
contains
   subroutine stop_error(msg, module, procedure)
      character(len=*), intent(in) :: msg
      character(len=*), optional, intent(in) :: module, procedure
      if (present(module) .and. present(procedure)) then
         write (*, '(A)') 'ERROR in '//trim(module)//'::'//trim(procedure)//': '//trim(msg)
      else
         write (*, '(A)') 'ERROR: '//trim(msg)
      end if
      error stop 1
   end subroutine
   subroutine type_error(var, type, intent, module, procedure)
      character(len=*), intent(in) :: var, type, intent, module, procedure
      call stop_error("The intent ["//trim(intent)//"] argument '"//trim(var)//"' must be of type '"//trim(type)//"'", module, procedure)
   end subroutine
end module LightKrylov_Logger

module LightKrylov
   use iso_c_binding
   use iso_fortran_env, only: real64
   use LightKrylov_Logger
   implicit none
   private
   integer, parameter, public :: dp = real64
   real(dp), parameter, public :: rtol_dp = 1.4901161193847656e-08_dp, atol_dp = 1.0e-12_dp
   public :: zero_basis, eigs, save_eigenspectrum, innerprod, type_error, stop_error

   type, abstract, public :: abstract_vector_rdp
   contains
      procedure(abstract_zero), pass(self), deferred, public :: zero
      procedure(abstract_rand), pass(self), deferred, public :: rand
      procedure(abstract_scal), pass(self), deferred, public :: scal
      procedure(abstract_axpby), pass(self), deferred, public :: axpby
      procedure(abstract_dot), pass(self), deferred, public :: dot
      procedure(abstract_size), pass(self), deferred, public :: get_size
      procedure, pass(self), public :: norm => vec_norm
      procedure, pass(self), public :: sub => vec_sub
      procedure, pass(self), public :: add => vec_add
   end type

   abstract interface
      subroutine abstract_zero(self)
         import abstract_vector_rdp
         class(abstract_vector_rdp), intent(inout) :: self
      end subroutine
      subroutine abstract_rand(self, ifnorm)
         import abstract_vector_rdp
         class(abstract_vector_rdp), intent(inout) :: self
         logical, optional, intent(in) :: ifnorm
      end subroutine
      subroutine abstract_scal(self, alpha)
         import abstract_vector_rdp, dp
         class(abstract_vector_rdp), intent(inout) :: self
         real(dp), intent(in) :: alpha
      end subroutine
      subroutine abstract_axpby(alpha, vec, beta, self)
         import abstract_vector_rdp, dp
         class(abstract_vector_rdp), intent(inout) :: self
         real(dp), intent(in) :: alpha
         class(abstract_vector_rdp), intent(in) :: vec
         real(dp), intent(in) :: beta
      end subroutine
      function abstract_dot(self, vec) result(alpha)
         import abstract_vector_rdp, dp
         class(abstract_vector_rdp), intent(in) :: self, vec
         real(dp) :: alpha
      end function
      pure function abstract_size(self) result(n)
         import abstract_vector_rdp
         class(abstract_vector_rdp), intent(in) :: self
         integer :: n
      end function
   end interface

   type, abstract, public :: abstract_linop_rdp
   contains
      procedure(abstract_matvec), pass(self), deferred, public :: matvec
      procedure(abstract_matvec), pass(self), deferred, public :: rmatvec
      procedure, pass(self), public :: finalize_timer => linop_finalize_timer
   end type

   type, abstract, extends(abstract_linop_rdp), public :: abstract_exptA_linop_rdp
      real(dp) :: tau = 1.0_dp
   end type

   abstract interface
      subroutine abstract_matvec(self, vec_in, vec_out)
         import abstract_linop_rdp, abstract_vector_rdp
         class(abstract_linop_rdp), intent(inout) :: self
         class(abstract_vector_rdp), intent(in) :: vec_in
         class(abstract_vector_rdp), intent(out) :: vec_out
      end subroutine
   end interface

   interface
      function c_dense_eig(n, A, lda, wr, wi, vr, ldvr) bind(C, name="nlg_dense_eig") result(rc)
         import c_int, c_double
         integer(c_int), value :: n, lda, ldvr
         real(c_double), intent(in) :: A(*)
         real(c_double), intent(out) :: wr(*), wi(*), vr(*)
         integer(c_int) :: rc
      end function
   end interface

contains

   function vec_norm(self) result(alpha)
      class(abstract_vector_rdp), intent(in) :: self
      real(dp) :: alpha
      alpha = sqrt(self%dot(self))
   end function

   subroutine vec_sub(self, vec)
      class(abstract_vector_rdp), intent(inout) :: self
      class(abstract_vector_rdp), intent(in) :: vec
      call self%axpby(-1.0_dp, vec, 1.0_dp)
   end subroutine

   subroutine vec_add(self, vec)
      class(abstract_vector_rdp), intent(inout) :: self
      class(abstract_vector_rdp), intent(in) :: vec
      call self%axpby(1.0_dp, vec, 1.0_dp)
   end subroutine

   subroutine linop_finalize_timer(self)
      class(abstract_linop_rdp), intent(inout) :: self
   end subroutine

   subroutine zero_basis(X)
      class(abstract_vector_rdp), intent(inout) :: X(:)
      integer :: i
      do i = 1, size(X)
         call X(i)%zero()
      end do
   end subroutine

   function innerprod(X, y) result(v)
      class(abstract_vector_rdp), intent(in) :: X(:), y
      real(dp) :: v(size(X))
      integer :: i
      do i = 1, size(X)
         v(i) = X(i)%dot(y)
      end do
   end function

   !> eigs(A, X, eigvals, residuals, info, x0=, kdim=, tolerance=, transpose=, write_intermediate=): see the header.
   subroutine eigs(A, X, eigvals, residuals, info, x0, kdim, tolerance, transpose, write_intermediate)
      class(abstract_linop_rdp), intent(inout) :: A
      class(abstract_vector_rdp), intent(inout) :: X(:)
      complex(dp), allocatable, intent(out) :: eigvals(:)
      real(dp), allocatable, intent(out) :: residuals(:)
      integer, intent(out) :: info
      class(abstract_vector_rdp), optional, intent(in) :: x0
      integer, optional, intent(in) :: kdim
      real(dp), optional, intent(in) :: tolerance
      logical, optional, intent(in) :: transpose, write_intermediate
      class(abstract_vector_rdp), allocatable :: Kb(:)
      real(dp), allocatable :: H(:, :), Hk(:), wr(:), wi(:), vr(:), res(:), h1(:)
      integer, allocatable :: order(:)
      integer :: nev, kd, k, i, j, pass, nconv, u, tmp
      real(dp) :: tol, beta
      logical :: trans, wint

      nev = size(X)
      kd = 4*nev; if (present(kdim)) kd = kdim
      tol = rtol_dp; if (present(tolerance)) tol = tolerance
      trans = .false.; if (present(transpose)) trans = transpose
      wint = .false.; if (present(write_intermediate)) wint = write_intermediate
      allocate (Kb(kd + 1), mold=X(1))
      call zero_basis(Kb)
      if (present(x0)) then
         call Kb(1)%add(x0)
      else
         call Kb(1)%rand(.true.)
      end if
      beta = Kb(1)%norm(); call Kb(1)%scal(1.0_dp/beta)
      allocate (H(kd + 1, kd)); H = 0.0_dp
      allocate (wr(kd), wi(kd), res(kd), order(kd))
      info = 0
      do k = 1, kd
         if (trans) then
            call A%rmatvec(Kb(k), Kb(k + 1))
         else
            call A%matvec(Kb(k), Kb(k + 1))
         end if
         info = info + 1
         ! double Gram-Schmidt: innerprod (k dots), then k axpbys, twice
         do pass = 1, 2
            h1 = innerprod(Kb(1:k), Kb(k + 1))
            do i = 1, k
               call Kb(k + 1)%axpby(-h1(i), Kb(i), 1.0_dp)
            end do
            H(1:k, k) = H(1:k, k) + h1
         end do
         beta = Kb(k + 1)%norm(); H(k + 1, k) = beta
         if (beta > 0.0_dp) call Kb(k + 1)%scal(1.0_dp/beta)
         ! Ritz pairs of H(1:k, 1:k)
         if (allocated(Hk)) deallocate (Hk, vr)
         allocate (Hk(k*k), vr(k*k))
         do j = 1, k
            Hk((j - 1)*k + 1:j*k) = H(1:k, j)
         end do
         if (c_dense_eig(int(k, c_int), Hk, int(k, c_int), wr, wi, vr, int(k, c_int)) /= 0) call stop_error('dense eigensolver failed', 'LightKrylov', 'eigs')
         do i = 1, k
            order(i) = i
         end do
         do i = 2, k      ! insertion sort by decreasing modulus
            tmp = order(i); j = i - 1
            do while (j >= 1)
               if (hypot(wr(order(j)), wi(order(j))) >= hypot(wr(tmp), wi(tmp))) exit
               order(j + 1) = order(j); j = j - 1
            end do
            order(j + 1) = tmp
         end do
         j = 1
         do while (j <= k)      ! residual |h_{k+1,k}| |e_k^T y|; a complex pair shares the modulus of (Re, Im) of the last row
            i = order(j)
            if (wi(i) /= 0.0_dp) then
               if (wi(i) > 0.0_dp) then
                  res(i) = beta*hypot(vr((i - 1)*k + k), vr(i*k + k)); if (i < k) res(i + 1) = res(i)
               else
                  res(i) = beta*hypot(vr((i - 2)*k + k), vr((i - 1)*k + k)); res(i - 1) = res(i)
               end if
            else
               res(i) = beta*abs(vr((i - 1)*k + k))
            end if
            j = j + 1
         end do
         nconv = 0
         do j = 1, k
            if (res(order(j)) < tol) then
               nconv = nconv + 1
            else
               exit
            end if
         end do
         if (wint) then
            open (newunit=u, file='eigs_output.txt', status='replace', action='write')
            write (u, '(A)') '#  iter                     Re                     Im                modulus               residual  conv'
            do j = 1, k
               i = order(j)
               write (u, '(I7,4(1X,ES22.14),3X,A1)') k, wr(i), wi(i), hypot(wr(i), wi(i)), res(i), merge('T', 'F', res(i) < tol)
            end do
            close (u)
         end if
         if (nconv >= nev .or. k == kd) exit
      end do
      k = min(k, kd)
      ! Ritz vectors in the real LAPACK convention (a complex pair occupies two consecutive columns: Re, Im)
      allocate (eigvals(nev), residuals(nev))
      call zero_basis(X)
      do j = 1, nev
         i = order(j)
         eigvals(j) = cmplx(wr(i), wi(i), kind=dp)
         residuals(j) = res(i)
         do pass = 1, k      ! column i: real part (wi > 0), imaginary part (wi < 0: second member of the pair) or the real vector
            call X(j)%axpby(vr((i - 1)*k + pass), Kb(pass), 1.0_dp)
         end do
      end do
   end subroutine eigs

   !> (n, 3) array [Re, Im, residual] in .npy format (the layout examples/*/plot_eigenvalues.py reads)
   subroutine save_eigenspectrum(eigvals, residuals, fname)
      complex(dp), intent(in) :: eigvals(:)
      real(dp), intent(in) :: residuals(:)
      character(len=*), intent(in) :: fname
      character(len=:), allocatable :: dict
      character(len=32) :: shp
      integer :: u, n, padded
      integer(c_int16_t) :: hlen
      real(dp), allocatable :: dat(:, :)
      n = size(eigvals)
      allocate (dat(n, 3))
      dat(:, 1) = real(eigvals); dat(:, 2) = aimag(eigvals); dat(:, 3) = residuals
      write (shp, '(I0)') n
      dict = "{'descr': '<f8', 'fortran_order': True, 'shape': ("//trim(shp)//", 3), }"
      padded = ((10 + len(dict) + 1 + 63)/64)*64 - 10
      hlen = int(padded, c_int16_t)
      open (newunit=u, file=fname, access='stream', form='unformatted', status='replace')
      write (u) achar(147), 'NUMPY', achar(1), achar(0), hlen, dict, repeat(' ', padded - len(dict) - 1), achar(10), dat
      close (u)
   end subroutine

end module LightKrylov

module LightKrylov_AbstractVectors
   use LightKrylov, only: abstract_vector_rdp
   implicit none
   public
end module

module LightKrylov_AbstractLinops
   use LightKrylov, only: abstract_linop_rdp, abstract_exptA_linop_rdp
   implicit none
   public
end module
