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
!!   abstract_system_rdp / abstract_jacobian_linop_rdp, newton, gmres_rdp : src/systems/neklab_systems.f90:42-55 (response,
!!       %jacobian, %X), src/neklab_analysis.f90:186-192 (newton(sys, bf, gmres_rdp, info, atol=, options=, scheduler=))
!!   svds(A, U, S, V, residuals, info, kdim=, write_intermediate=)  as called at src/neklab_analysis.f90:136
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
   public :: zero_basis, eigs, svds, save_eigenspectrum, innerprod, type_error, stop_error, initialize_krylov_subspace
   public :: newton, gmres_rdp

   !> options / metadata of `newton` as the reference uses them (src/neklab_analysis.f90:173-196: newton_dp_opts(maxiter=40,
   !! ifbisect=.false.), meta%input_is_fixed_point); only the components the reference touches are restated
   type, public :: newton_dp_opts
      integer :: maxiter = 100
      logical :: ifbisect = .false.
      integer :: maxstep_bisection = 5
   end type
   type, public :: newton_dp_metadata
      integer :: n_iter = 0
      logical :: converged = .false.
      logical :: input_is_fixed_point = .false.
   end type

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

   !> complex vectors and operators (src/vectors/neklab_vectors.f90: nek_zvector; src/linops/neklab_linops.f90:198-205: resolvent_linop)
   type, abstract, public :: abstract_vector_cdp
   contains
      procedure(abstract_zzero), pass(self), deferred, public :: zero
      procedure(abstract_zrand), pass(self), deferred, public :: rand
      procedure(abstract_zscal), pass(self), deferred, public :: scal
      procedure(abstract_zaxpby), pass(self), deferred, public :: axpby
      procedure(abstract_zdot), pass(self), deferred, public :: dot
      procedure(abstract_zsize), pass(self), deferred, public :: get_size
      procedure, pass(self), public :: norm => zvec_norm
   end type

   abstract interface
      subroutine abstract_zzero(self)
         import abstract_vector_cdp
         class(abstract_vector_cdp), intent(inout) :: self
      end subroutine
      subroutine abstract_zrand(self, ifnorm)
         import abstract_vector_cdp
         class(abstract_vector_cdp), intent(inout) :: self
         logical, optional, intent(in) :: ifnorm
      end subroutine
      subroutine abstract_zscal(self, alpha)
         import abstract_vector_cdp, dp
         class(abstract_vector_cdp), intent(inout) :: self
         complex(dp), intent(in) :: alpha
      end subroutine
      subroutine abstract_zaxpby(alpha, vec, beta, self)
         import abstract_vector_cdp, dp
         class(abstract_vector_cdp), intent(inout) :: self
         complex(dp), intent(in) :: alpha
         class(abstract_vector_cdp), intent(in) :: vec
         complex(dp), intent(in) :: beta
      end subroutine
      function abstract_zdot(self, vec) result(alpha)
         import abstract_vector_cdp, dp
         class(abstract_vector_cdp), intent(in) :: self, vec
         complex(dp) :: alpha
      end function
      pure function abstract_zsize(self) result(n)
         import abstract_vector_cdp
         class(abstract_vector_cdp), intent(in) :: self
         integer :: n
      end function
   end interface

   type, abstract, public :: abstract_linop_cdp
   contains
      procedure(abstract_zmatvec), pass(self), deferred, public :: matvec
      procedure(abstract_zmatvec), pass(self), deferred, public :: rmatvec
   end type

   abstract interface
      subroutine abstract_zmatvec(self, vec_in, vec_out)
         import abstract_linop_cdp, abstract_vector_cdp
         class(abstract_linop_cdp), intent(inout) :: self
         class(abstract_vector_cdp), intent(in) :: vec_in
         class(abstract_vector_cdp), intent(out) :: vec_out
      end subroutine
   end interface

   !> Jacobian of a nonlinear system about the state X (neklab_systems.f90:47-55: `self%X`)
   type, abstract, extends(abstract_linop_rdp), public :: abstract_jacobian_linop_rdp
      class(abstract_vector_rdp), allocatable :: X
   end type

   !> nonlinear system F(X) = 0 with its Jacobian (neklab_systems.f90:42-46: `response`; tsyphon.usr:38-40: `sys%jacobian`)
   type, abstract, public :: abstract_system_rdp
      class(abstract_jacobian_linop_rdp), allocatable :: jacobian
   contains
      procedure(abstract_response), pass(self), deferred, public :: response
      procedure, pass(self), public :: finalize_timer => system_finalize_timer
   end type

   abstract interface
      subroutine abstract_response(self, vec_in, vec_out, atol)
         import abstract_system_rdp, abstract_vector_rdp, dp
         class(abstract_system_rdp), intent(inout) :: self
         class(abstract_vector_rdp), intent(in) :: vec_in
         class(abstract_vector_rdp), intent(out) :: vec_out
         real(dp), intent(in) :: atol
      end subroutine
      subroutine abstract_scheduler(tol, target_tol, rnorm, iter, info)
         import dp
         real(dp), intent(out) :: tol
         real(dp), intent(in) :: target_tol, rnorm
         integer, intent(in) :: iter
         integer, intent(out) :: info
      end subroutine
      subroutine abstract_linear_solver(A, b, x, info, atol)
         import abstract_linop_rdp, abstract_vector_rdp, dp
         class(abstract_linop_rdp), intent(inout) :: A
         class(abstract_vector_rdp), intent(in) :: b
         class(abstract_vector_rdp), intent(inout) :: x
         integer, intent(out) :: info
         real(dp), intent(in) :: atol
      end subroutine
   end interface

   abstract interface
      subroutine abstract_matvec(self, vec_in, vec_out)
         import abstract_linop_rdp, abstract_vector_rdp
         class(abstract_linop_rdp), intent(inout) :: self
         class(abstract_vector_rdp), intent(in) :: vec_in
         class(abstract_vector_rdp), intent(out) :: vec_out
      end subroutine
   end interface

   interface
      function c_symtridiag_eig(n, d, e, Z) bind(C, name="nlg_symtridiag_eig") result(rc)
         import c_int, c_double
         integer(c_int), value :: n
         real(c_double), intent(inout) :: d(*), e(*)
         real(c_double), intent(out) :: Z(*)
         integer(c_int) :: rc
      end function
      function c_dense_eig(n, A, lda, wr, wi, vr, ldvr) bind(C, name="nlg_dense_eig") result(rc)
         import c_int, c_double
         integer(c_int), value :: n, lda, ldvr
         real(c_double), intent(in) :: A(*)
         real(c_double), intent(out) :: wr(*), wi(*), vr(*)
         integer(c_int) :: rc
      end function
   end interface

contains

   function zvec_norm(self) result(alpha)
      class(abstract_vector_cdp), intent(in) :: self
      real(dp) :: alpha
      alpha = sqrt(real(self%dot(self), dp))
   end function

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

   subroutine system_finalize_timer(self)
      class(abstract_system_rdp), intent(inout) :: self
   end subroutine

   subroutine initialize_krylov_subspace(X)
      class(abstract_vector_rdp), intent(inout) :: X(:)
      call zero_basis(X)
   end subroutine

   !> restarted GMRES(30) on A x = b from the x handed in, written against the abstract interfaces (Givens rotations, modified
   !! Gram-Schmidt); stops at |r| < atol.  info = matvecs.
   subroutine gmres_rdp(A, b, x, info, atol)
      class(abstract_linop_rdp), intent(inout) :: A
      class(abstract_vector_rdp), intent(in) :: b
      class(abstract_vector_rdp), intent(inout) :: x
      integer, intent(out) :: info
      real(dp), intent(in) :: atol
      integer, parameter :: kd = 30, maxcycle = 10
      class(abstract_vector_rdp), allocatable :: V(:), w
      real(dp) :: H(kd + 1, kd), cs(kd), sn(kd), g(kd + 1), y(kd), beta, t, d
      integer :: k, i, j, cyc
      info = 0
      allocate (V(kd + 1), mold=b); allocate (w, mold=b)
      do cyc = 1, maxcycle
         call zero_basis(V)
         call A%matvec(x, w); info = info + 1
         call V(1)%add(b); call V(1)%sub(w)                 ! r = b - A x
         beta = V(1)%norm()
         if (beta < atol) return
         call V(1)%scal(1.0_dp/beta)
         H = 0.0_dp; g = 0.0_dp; g(1) = beta
         k = 0
         do j = 1, kd
            call A%matvec(V(j), V(j + 1)); info = info + 1
            do i = 1, j
               H(i, j) = V(i)%dot(V(j + 1)); call V(j + 1)%axpby(-H(i, j), V(i), 1.0_dp)
            end do
            H(j + 1, j) = V(j + 1)%norm()
            if (H(j + 1, j) > 0.0_dp) call V(j + 1)%scal(1.0_dp/H(j + 1, j))
            do i = 1, j - 1
               t = cs(i)*H(i, j) + sn(i)*H(i + 1, j); H(i + 1, j) = -sn(i)*H(i, j) + cs(i)*H(i + 1, j); H(i, j) = t
            end do
            d = hypot(H(j, j), H(j + 1, j)); cs(j) = H(j, j)/d; sn(j) = H(j + 1, j)/d
            H(j, j) = d; H(j + 1, j) = 0.0_dp
            g(j + 1) = -sn(j)*g(j); g(j) = cs(j)*g(j)
            k = j
            if (abs(g(j + 1)) < atol) exit
         end do
         do i = k, 1, -1
            y(i) = (g(i) - dot_product(H(i, i + 1:k), y(i + 1:k)))/H(i, i)
         end do
         do i = 1, k
            call x%axpby(y(i), V(i), 1.0_dp)
         end do
         if (abs(g(k + 1)) < atol) return
      end do
   end subroutine gmres_rdp

   !> Newton iteration on sys%response(X) = 0: X <- X + dx with jacobian dx = -F(X) by `linear_solver`; the scheduler sets the
   !! tolerance the residual and the linear solves are evaluated with (neklab_analysis.f90:186-192).  info = iterations, < 0: not converged.
   subroutine newton(sys, X, linear_solver, info, atol, options, scheduler, meta)
      class(abstract_system_rdp), intent(inout) :: sys
      class(abstract_vector_rdp), intent(inout) :: X
      procedure(abstract_linear_solver) :: linear_solver
      integer, intent(out) :: info
      real(dp), intent(in) :: atol
      type(newton_dp_opts), optional, intent(in) :: options
      procedure(abstract_scheduler), optional :: scheduler
      type(newton_dp_metadata), optional, intent(out) :: meta
      class(abstract_vector_rdp), allocatable :: r, dx
      real(dp) :: tol, rnorm
      integer :: it, nmax, sinfo, linfo
      nmax = 100; if (present(options)) nmax = options%maxiter
      allocate (r, mold=X); allocate (dx, mold=X)
      tol = atol
      info = -1
      if (present(meta)) then
         meta%converged = .false.; meta%input_is_fixed_point = .false.; meta%n_iter = 0
      end if
      do it = 0, nmax
         call sys%response(X, r, tol)
         rnorm = r%norm()
         if (present(scheduler)) then
            call scheduler(tol, atol, rnorm, it, sinfo)
         end if
         if (rnorm < atol) then
            if (tol <= atol*(1.0_dp + 1.0e-12_dp)) then
               info = it
               if (present(meta)) then
                  meta%converged = .true.; meta%input_is_fixed_point = it == 0; meta%n_iter = it
               end if
               return
            end if
            tol = atol; cycle                                ! converged at a loose solver tolerance: re-evaluate at the target
         end if
         if (it == nmax) exit
         if (allocated(sys%jacobian%X)) deallocate (sys%jacobian%X)
         allocate (sys%jacobian%X, source=X)
         call dx%zero(); call r%scal(-1.0_dp)
         call linear_solver(sys%jacobian, r, dx, linfo, 0.1_dp*max(rnorm, atol))
         call X%add(dx)
      end do
      if (present(meta)) meta%n_iter = nmax
   end subroutine newton

   !> Golub-Kahan-Lanczos bidiagonalisation with full re-orthogonalisation through the abstract interfaces; singular values of
   !! the bidiagonal matrix from the tridiagonal eigenproblem of B^T B (the structure of the device twin nlg_svds)
   subroutine svds(A, U, S, V, residuals, info, kdim, tolerance, write_intermediate)
      class(abstract_linop_rdp), intent(inout) :: A
      class(abstract_vector_rdp), intent(inout) :: U(:), V(:)
      real(dp), allocatable, intent(out) :: S(:), residuals(:)
      integer, intent(out) :: info
      integer, optional, intent(in) :: kdim
      real(dp), optional, intent(in) :: tolerance
      logical, optional, intent(in) :: write_intermediate
      class(abstract_vector_rdp), allocatable :: Ub(:), Vb(:)
      real(dp), allocatable :: al(:), be(:), d(:), e(:), Z(:), sig(:), res(:), p(:)
      real(dp) :: tol, h
      integer :: nsv, kd, k, i, j, pass, conv, src
      nsv = size(U)
      kd = 4*nsv; if (present(kdim)) kd = kdim
      tol = rtol_dp; if (present(tolerance)) tol = tolerance
      allocate (Ub(kd + 1), mold=U(1)); allocate (Vb(kd), mold=U(1))
      call zero_basis(Ub); call zero_basis(Vb)
      call Ub(1)%rand(.false.); h = Ub(1)%norm(); call Ub(1)%scal(1.0_dp/h)
      allocate (al(kd), be(kd + 1)); al = 0.0_dp; be = 0.0_dp
      info = 0
      do k = 1, kd
         call A%rmatvec(Ub(k), Vb(k))
         do pass = 1, 2
            do i = 1, k - 1
               h = Vb(i)%dot(Vb(k)); call Vb(k)%axpby(-h, Vb(i), 1.0_dp)
            end do
         end do
         al(k) = Vb(k)%norm(); call Vb(k)%scal(1.0_dp/al(k))
         call A%matvec(Vb(k), Ub(k + 1))
         do pass = 1, 2
            do i = 1, k
               h = Ub(i)%dot(Ub(k + 1)); call Ub(k + 1)%axpby(-h, Ub(i), 1.0_dp)
            end do
         end do
         be(k + 1) = Ub(k + 1)%norm(); call Ub(k + 1)%scal(1.0_dp/be(k + 1))
         info = info + 2
         if (allocated(d)) deallocate (d, e, Z, sig, res)
         allocate (d(k), e(k), Z(k*k), sig(k), res(k)); e = 0.0_dp
         do i = 1, k
            d(i) = al(i)**2; if (i < k) d(i) = d(i) + be(i + 1)**2
            if (i > 1) e(i) = al(i)*be(i)
         end do
         if (c_symtridiag_eig(int(k, c_int), d, e, Z) /= 0) call stop_error('tridiagonal eigensolver failed', 'LightKrylov', 'svds')
         conv = 0
         do i = 1, k                                         ! descending singular values
            src = k + 1 - i
            sig(i) = sqrt(max(d(src), 0.0_dp))
            res(i) = abs(be(k + 1)*Z((k - 1)*k + src))            ! Z row-major [component][eigenvector]
            if (res(i) < tol) conv = conv + 1
         end do
         if (conv >= nsv .or. k == kd) exit
      end do
      k = min(k, kd)
      allocate (S(nsv), residuals(nsv), p(k))
      call zero_basis(U); call zero_basis(V)
      do i = 1, min(nsv, k)
         src = k + 1 - i
         S(i) = sig(i); residuals(i) = res(i)
         do j = 1, k
            call V(i)%axpby(Z((j - 1)*k + src), Vb(j), 1.0_dp)
         end do
         do j = 1, k                                         ! p = B q / sigma
            p(j) = al(j)*Z((j - 1)*k + src); if (j > 1) p(j) = p(j) + be(j)*Z((j - 2)*k + src)
            call U(i)%axpby(p(j)/S(i), Ub(j), 1.0_dp)
         end do
      end do
   end subroutine svds

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
   use LightKrylov, only: abstract_vector_rdp, abstract_vector_cdp
   implicit none
   public
end module

module LightKrylov_AbstractLinops
   use LightKrylov, only: abstract_linop_rdp, abstract_exptA_linop_rdp, abstract_linop_cdp
   implicit none
   public
end module
