!> Drop-in for the reference's module of the same name: `exptA_linop` (and the wavenumber-projected variant) with every
!! type-bound procedure forwarded to libneklab_gpu.so.
!!
!! Mapping (file:line under /root/reference):
!!   type exptA_linop                     src/linops/neklab_linops.f90:35-44   (constructor idiom exptA_linop(1.0_dp, bf),
!!                                        examples/cylinder/stability/direct/1cyl.usr:20: tau is the parent's component,
!!                                        baseflow the first own component -- positional construction works unchanged)
!!     init / matvec / rmatvec            src/linops/exponential_propagator.f90:4-107
!!     compute_rst / get_rst              src/linops/exponential_propagator.f90:109-142
!!   type exptA_proj_linop                src/linops/neklab_linops.f90:130-152, exponential_propagator_proj.f90
!! plus what the reference reaches through LightKrylov and neklab_systems on this path, bound to the device block path:
!!   nek_eigs / nek_svds                  the eigs / svds calls of src/neklab_analysis.f90:80-81, :136 (nlg_eigs, nlg_svds)
!!   nonlinear_map / set_baseflow / set_tolerances   src/systems/fixed_point.f90:4-96, neklab_systems.f90:229-335
!!   integrate_forced                     src/linops/resolvent.f90:80-111, :133-166
module neklab_linops
   use iso_c_binding
   use LightKrylov, only: dp, abstract_vector_rdp, abstract_exptA_linop_rdp, abstract_vector_cdp, abstract_linop_cdp, type_error
   use neklab_gpu_capi
   use neklab_vectors
   implicit none
   private
   character(len=*), parameter, private :: this_module = 'neklab_linops'

   public :: nek_eigs, nek_svds

   !------------------------------------------
   !-----     EXPONENTIAL PROPAGATOR     -----
   !------------------------------------------
   type, extends(abstract_exptA_linop_rdp), public :: exptA_linop
      type(nek_dvector) :: baseflow
      !> solver configuration; taken from the case (neklab_gpu_set_case: what the reference reads from param(.)) at init
      !! unless `cfg_set` says the caller filled it in
      type(nlg_exptA_config) :: cfg = nlg_exptA_config()
      logical :: cfg_set = .false.
      type(c_ptr), private :: h = c_null_ptr
      integer(c_intptr_t), private :: owner = 0
      real(dp), private :: tau_built = -1.0_dp
   contains
      private
      procedure, pass(self), public :: init => init_exptA
      procedure, pass(self), public :: matvec => exptA_matvec
      procedure, pass(self), public :: rmatvec => exptA_rmatvec
      procedure, pass(self), public :: compute_rst => exptA_compute_rst
      procedure, pass(self), public :: get_rst => exptA_get_rst
      ! Newton-Krylov row: what nek_system%eval (nonlinear_map) and nek_jacobian (self%X) forward to
      procedure, pass(self), public :: nonlinear_map => exptA_nonlinear_map
      procedure, pass(self), public :: set_baseflow => exptA_set_baseflow
      procedure, pass(self), public :: set_tolerances => exptA_set_tolerances
      ! resolvent building block (evaluate_rhs / evaluate_imaginary_part)
      procedure, pass(self), public :: integrate_forced => exptA_integrate_forced
      procedure, pass(self), public :: handle => exptA_handle
      procedure, pass(self), public :: nsteps => exptA_nsteps
      final :: finalize_exptA
   end type exptA_linop

   !> exptA_temp_linop (src/linops/neklab_linops.f90:82-93, exponential_propagator_temp.f90: the propagator of the temperature-
   !! coupled equations; the reference's file is a clone of the plain one because the temperature already travels in
   !! nek_dvector%theta) and, under the name the reference's umbrella module and its thermosyphon case use, exptA_linop_temp
   !! (src/neklab.f90:45, examples/thermosyphon/baseflow/tsyphon.usr:13,52 `exptA_linop_temp(1.0_dp, bf)`).  Here the coupling
   !! is a property of the case (neklab_gpu_set_case(ifheat=.true., ..)); the two types insist on it at init().
   type, extends(exptA_linop), public :: exptA_temp_linop
   contains
      procedure, pass(self), public :: init => init_exptA_temp
   end type exptA_temp_linop
   type, extends(exptA_temp_linop), public :: exptA_linop_temp
   end type exptA_linop_temp

   !> exptA_proj_linop(tau=.., baseflow=.., alpha=..) (examples/poiseuille/stability/direct_alpha_1/poiseuille.usr:24).
   !! `set_lines` hands over what Nek5000's gtpp_gs_setup derives from (nelx, nely, nelz): one label per velocity point
   !! naming its line along the homogeneous direction.
   type, extends(exptA_linop), public :: exptA_proj_linop
      real(dp) :: alpha = 0.0_dp
      integer :: idir = 1
   contains
      procedure, pass(self), public :: set_lines => proj_set_lines
      procedure, pass(self), public :: proj => proj_apply
   end type exptA_proj_linop

   !--------------------------------------
   !-----     RESOLVENT OPERATOR     -----
   !--------------------------------------
   !> resolvent_linop(omega, baseflow) (src/linops/neklab_linops.f90:198-205, resolvent.f90): R(omega) f by time stepping -- one
   !! forcing period from rest (nlg_linop_integrate_forced), the real part from (I - exp(T L)) x = b by GMRES(64) at rtol 1e-6
   !! (resolvent.f90:113-131; restated here on the type-bound procedures, LightKrylov's gmres options are not in the reference
   !! tree), the imaginary part = the state a quarter period later; rmatvec integrates the adjoint equations.
   type, extends(abstract_linop_cdp), public :: resolvent_linop
      real(kind=dp) :: omega = 0.0_dp
      type(nek_dvector) :: baseflow
   contains
      private
      procedure, pass(self), public :: matvec => resolvent_matvec
      procedure, pass(self), public :: rmatvec => resolvent_rmatvec
   end type

contains

   function exptA_handle(self) result(h)
      class(exptA_linop), intent(in) :: self
      type(c_ptr) :: h
      if (.not. c_associated(self%h) .or. self%owner /= loc(self)) then
         write (*, '(A)') 'ERROR in '//this_module//': exptA%init() has not been called on this object (1cyl.usr:20)'
         error stop 1
      end if
      h = self%h
   end function

   integer function exptA_nsteps(self) result(n)
      class(exptA_linop), intent(in) :: self
      real(c_double) :: tau, dt, cfl
      integer(c_int) :: ns
      call nlg_check(c_linop_get_info(exptA_handle(self), tau, dt, ns, cfl), 'exptA_nsteps')
      n = ns
   end function

   subroutine init_exptA(self)
      class(exptA_linop), intent(inout) :: self
      integer(c_int) :: rc
      if (c_associated(self%h) .and. self%owner == loc(self)) rc = c_linop_destroy(self%h)
      self%h = c_null_ptr
      if (.not. self%cfg_set) self%cfg = nek_case      ! param(2), param(21/22), |param(27)|, ifheat ... of the Nek5000 host
      self%cfg%tau = self%tau
      call nlg_check(c_linop_create(nlg_mesh, self%cfg, nek_dvector_handle(self%baseflow), self%h), 'init_exptA')
      self%owner = loc(self)
      call nlg_check(c_linop_init(self%h), 'init_exptA')
      self%tau_built = self%tau
   end subroutine

   subroutine init_exptA_temp(self)
      class(exptA_temp_linop), intent(inout) :: self
      if (.not. self%cfg_set) self%cfg = nek_case
      if (self%cfg%ifheat == 0) then
         write (*, '(A)') 'ERROR in '//this_module//': exptA_temp_linop needs the temperature coupling of the case (neklab_gpu_set_case(ifheat=.true.))'
         error stop 1
      end if
      self%cfg_set = .true.
      call init_exptA(self)
   end subroutine

   subroutine sync_tau(self, where)
      class(exptA_linop), intent(inout) :: self
      character(len=*), intent(in) :: where
      if (self%tau /= self%tau_built) then       ! apply_exptA sets A%tau before the call (neklab_linops.f90:252)
         call nlg_check(c_linop_set_tau(exptA_handle(self), self%tau), where); self%tau_built = self%tau
      end if
   end subroutine

   subroutine exptA_matvec(self, vec_in, vec_out)
      class(exptA_linop), intent(inout) :: self
      class(abstract_vector_rdp), intent(in) :: vec_in
      class(abstract_vector_rdp), intent(out) :: vec_out
      call sync_tau(self, 'exptA_matvec')
      select type (vec_in)
      type is (nek_dvector)
         select type (vec_out)
         type is (nek_dvector)
            call nek_dvector_ensure(vec_out)
            call nlg_check(c_linop_matvec(exptA_handle(self), nek_dvector_handle(vec_in), vec_out%h), 'exptA_matvec')
         class default
            call type_error('vec_out', 'nek_dvector', 'OUT', this_module, 'exptA_matvec')
         end select
      class default
         call type_error('vec_in', 'nek_dvector', 'IN', this_module, 'exptA_matvec')
      end select
   end subroutine

   subroutine exptA_rmatvec(self, vec_in, vec_out)
      class(exptA_linop), intent(inout) :: self
      class(abstract_vector_rdp), intent(in) :: vec_in
      class(abstract_vector_rdp), intent(out) :: vec_out
      call sync_tau(self, 'exptA_rmatvec')
      select type (vec_in)
      type is (nek_dvector)
         select type (vec_out)
         type is (nek_dvector)
            call nek_dvector_ensure(vec_out)
            call nlg_check(c_linop_rmatvec(exptA_handle(self), nek_dvector_handle(vec_in), vec_out%h), 'exptA_rmatvec')
         class default
            call type_error('vec_out', 'nek_dvector', 'OUT', this_module, 'exptA_rmatvec')
         end select
      class default
         call type_error('vec_in', 'nek_dvector', 'IN', this_module, 'exptA_rmatvec')
      end select
   end subroutine

   !> exptA_compute_rst (exponential_propagator.f90:109-127).  In the reference the matvec calls it after the final state
   !! has been copied out, to run `nrst` extra steps that fill the history of vec_out.  The device matvec does exactly that
   !! internally (the integrator state lives on the device), so the history is already in place when the matvec returns:
   !! this binding exists for callers that invoke it themselves and only checks that claim.
   subroutine exptA_compute_rst(self, vec_out, nrst)
      class(exptA_linop), intent(inout) :: self
      class(abstract_vector_rdp), intent(inout) :: vec_out
      integer, intent(in) :: nrst
      integer(c_int) :: have
      select type (vec_out)
      type is (nek_dvector)
         call nlg_check(c_vec_nrst(nek_dvector_handle(vec_out), have), 'exptA_compute_rst')
         if (have < nrst) then
            write (*, '(A,I0,A,I0)') 'ERROR in exptA_compute_rst: the vector holds ', have, ' restart fields, expected ', nrst
            error stop 1
         end if
      class default
         call type_error('vec_out', 'nek_dvector', 'OUT', this_module, 'exptA_compute_rst')
      end select
   end subroutine

   !> exptA_get_rst (exponential_propagator.f90:129-142): replay of vec_in's history slot `istep` -- performed inside the
   !! device matvec after time step istep <= nrst whenever vec_in%has_rst_fields(); kept for interface completeness.
   subroutine exptA_get_rst(self, vec_in, istep)
      class(exptA_linop), intent(inout) :: self
      class(abstract_vector_rdp), intent(in) :: vec_in
      integer, intent(in) :: istep
      select type (vec_in)
      type is (nek_dvector)
      class default
         call type_error('vec_in', 'nek_dvector', 'IN', this_module, 'exptA_get_rst')
      end select
   end subroutine

   subroutine exptA_nonlinear_map(self, vec_in, vec_out)
      class(exptA_linop), intent(inout) :: self
      class(abstract_vector_rdp), intent(in) :: vec_in
      class(abstract_vector_rdp), intent(out) :: vec_out
      select type (vec_in)
      type is (nek_dvector)
         select type (vec_out)
         type is (nek_dvector)
            call nek_dvector_ensure(vec_out)
            call nlg_check(c_linop_nonlinear_map(exptA_handle(self), nek_dvector_handle(vec_in), vec_out%h), 'nonlinear_map')
         class default
            call type_error('vec_out', 'nek_dvector', 'OUT', this_module, 'nonlinear_map')
         end select
      class default
         call type_error('vec_in', 'nek_dvector', 'IN', this_module, 'nonlinear_map')
      end select
   end subroutine

   subroutine exptA_set_baseflow(self, X)
      class(exptA_linop), intent(inout) :: self
      class(abstract_vector_rdp), intent(in) :: X
      select type (X)
      type is (nek_dvector)
         call nlg_check(c_linop_set_baseflow(exptA_handle(self), nek_dvector_handle(X)), 'set_baseflow')
      class default
         call type_error('X', 'nek_dvector', 'IN', this_module, 'set_baseflow')
      end select
   end subroutine

   subroutine exptA_set_tolerances(self, vtol, ptol)
      class(exptA_linop), intent(inout) :: self
      real(dp), intent(in) :: vtol, ptol
      call nlg_check(c_linop_set_tolerances(exptA_handle(self), vtol, ptol), 'set_tolerances')
   end subroutine

   !> vec_out = state after one application started from `ic` (absent: rest) under the body force
   !! Re[(f_re + i f_im) exp(+- i omega t)]  (resolvent.f90:80-111, :133-166)
   subroutine exptA_integrate_forced(self, f_re, omega, vec_out, ic, f_im, adjoint)
      class(exptA_linop), intent(inout) :: self
      type(nek_dvector), intent(in) :: f_re
      real(dp), intent(in) :: omega
      type(nek_dvector), intent(inout) :: vec_out
      type(nek_dvector), optional, intent(in) :: ic, f_im
      logical, optional, intent(in) :: adjoint
      type(c_ptr) :: hic, him
      integer(c_int) :: adj
      hic = c_null_ptr; him = c_null_ptr; adj = 0
      if (present(ic)) hic = nek_dvector_handle(ic)
      if (present(f_im)) him = nek_dvector_handle(f_im)
      if (present(adjoint)) adj = merge(1, 0, adjoint)
      call sync_tau(self, 'integrate_forced')
      call nek_dvector_ensure(vec_out)
      call nlg_check(c_linop_integrate_forced(exptA_handle(self), hic, nek_dvector_handle(f_re), him, omega, adj, vec_out%h), 'integrate_forced')
   end subroutine

   subroutine finalize_exptA(self)
      type(exptA_linop), intent(inout) :: self
      integer(c_int) :: rc
      if (c_associated(self%h) .and. self%owner == loc(self)) rc = c_linop_destroy(self%h)
      self%h = c_null_ptr; self%owner = 0
   end subroutine

   !---- exptA_proj_linop ---------------------------------------------------------------------------------------------
   !> after init(): line labels of the velocity points (and, optionally, of the pressure points with their coordinate along
   !! idir: the pressure is then projected as well, see include/neklab_gpu.h nlg_linop_set_projection)
   subroutine proj_set_lines(self, line_label, line_label2, x2)
      class(exptA_proj_linop), intent(inout) :: self
      integer(c_int64_t), target, intent(in) :: line_label(*)
      integer(c_int64_t), target, optional, intent(in) :: line_label2(*)
      real(dp), target, optional, intent(in) :: x2(*)
      type(c_ptr) :: l2, xx
      l2 = c_null_ptr; xx = c_null_ptr
      if (present(line_label2) .and. present(x2)) then
         l2 = c_loc(line_label2); xx = c_loc(x2)
      end if
      call nlg_check(c_linop_set_projection(exptA_handle(self), self%alpha, int(self%idir, c_int), c_loc(line_label), l2, xx), 'exptA_proj set_lines')
   end subroutine

   !> proj_alpha (exponential_propagator_proj.f90:135-173) applied to the state held by `vec`
   subroutine proj_apply(self, vec)
      class(exptA_proj_linop), intent(inout) :: self
      type(nek_dvector), intent(inout) :: vec
      call nek_dvector_ensure(vec)
      call nlg_check(c_linop_project(exptA_handle(self), vec%h), 'exptA_proj proj')
   end subroutine

   !---- resolvent ------------------------------------------------------------------------------------------------------
   subroutine resolvent_apply(self, vec_in, vec_out, adjoint)
      class(resolvent_linop), intent(inout) :: self
      class(abstract_vector_cdp), intent(in) :: vec_in
      class(abstract_vector_cdp), intent(out) :: vec_out
      logical, intent(in) :: adjoint
      real(dp), parameter :: two_pi = 8.0_dp*atan(1.0_dp)
      type(exptA_linop) :: exptA
      type(nek_dvector) :: b
      real(dp) :: tau
      tau = 1.0_dp
      if (self%omega /= 0.0_dp) tau = two_pi/abs(self%omega)
      exptA%tau = tau; exptA%baseflow = self%baseflow
      call exptA%init()
      select type (vec_in)
      type is (nek_zvector)
         select type (vec_out)
         type is (nek_zvector)
            call exptA%integrate_forced(vec_in%re, self%omega, b, f_im=vec_in%im, adjoint=adjoint)      ! evaluate_rhs
            call solve_real_part(exptA, b, vec_out%re, adjoint)
            exptA%tau = 0.25_dp*tau                                                                      ! resolvent.f90:35
            call exptA%integrate_forced(vec_in%re, self%omega, vec_out%im, ic=vec_out%re, f_im=vec_in%im, adjoint=adjoint)
         class default
            call type_error('vec_out', 'nek_zvector', 'OUT', this_module, 'resolvent_matvec')
         end select
      class default
         call type_error('vec_in', 'nek_zvector', 'IN', this_module, 'resolvent_matvec')
      end select
   end subroutine

   subroutine resolvent_matvec(self, vec_in, vec_out)
      class(resolvent_linop), intent(inout) :: self
      class(abstract_vector_cdp), intent(in) :: vec_in
      class(abstract_vector_cdp), intent(out) :: vec_out
      call resolvent_apply(self, vec_in, vec_out, .false.)
   end subroutine

   subroutine resolvent_rmatvec(self, vec_in, vec_out)
      class(resolvent_linop), intent(inout) :: self
      class(abstract_vector_cdp), intent(in) :: vec_in
      class(abstract_vector_cdp), intent(out) :: vec_out
      call resolvent_apply(self, vec_in, vec_out, .true.)
   end subroutine

   !> (I - exp(T L)) x = b, x0 = 0, restarted GMRES(64) with Givens rotations, |r| <= max(1e-6 |b|, 1e-12).  Krylov vectors lose their
   !! restart history (each application starts impulsively, like the forced integration whose periodic state is sought).
   subroutine solve_real_part(exptA, b, x, adjoint)
      type(exptA_linop), intent(inout) :: exptA
      type(nek_dvector), intent(in) :: b
      type(nek_dvector), intent(inout) :: x
      logical, intent(in) :: adjoint
      integer, parameter :: kd = 64, maxcycle = 10
      type(nek_dvector), allocatable :: V(:)
      type(nek_dvector) :: w
      real(dp) :: H(kd + 1, kd), cs(kd), sn(kd), g(kd + 1), y(kd), beta, t, d, tol
      integer :: k, i, j, cyc
      tol = max(1.0e-6_dp*b%norm(), 1.0e-12_dp)
      call x%zero()
      allocate (V(kd + 1))
      do cyc = 1, maxcycle
         call V(1)%zero(); call V(1)%axpby(1.0_dp, b, 0.0_dp)
         if (cyc > 1) then                                   ! r = b - (x - A x)
            if (adjoint) then
               call exptA%rmatvec(x, w)
            else
               call exptA%matvec(x, w)
            end if
            call w%clear_rst_fields()
            call V(1)%axpby(-1.0_dp, x, 1.0_dp); call V(1)%axpby(1.0_dp, w, 1.0_dp)
         end if
         beta = V(1)%norm()
         if (beta <= tol) return
         call V(1)%scal(1.0_dp/beta)
         H = 0.0_dp; g = 0.0_dp; g(1) = beta; k = 0
         do j = 1, kd
            if (adjoint) then
               call exptA%rmatvec(V(j), V(j + 1))
            else
               call exptA%matvec(V(j), V(j + 1))
            end if
            call V(j + 1)%clear_rst_fields()
            call V(j + 1)%axpby(1.0_dp, V(j), -1.0_dp)         ! (I - A) v_j
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
            if (abs(g(j + 1)) <= tol) exit
         end do
         do i = k, 1, -1
            y(i) = (g(i) - dot_product(H(i, i + 1:k), y(i + 1:k)))/H(i, i)
         end do
         do i = 1, k
            call x%axpby(y(i), V(i), 1.0_dp)
         end do
         if (abs(g(k + 1)) <= tol) return
      end do
   end subroutine

   !---- eigs / svds on the device block path ---------------------------------------------------------------------------
   !> Same argument list as the LightKrylov call at src/neklab_analysis.f90:80-81; the Krylov basis lives in one
   !! allocation on the device and every Gram-Schmidt pass is two kernels and one all-reduce (nlg_eigs).
   subroutine nek_eigs(A, X, eigvals, residuals, info, x0, kdim, tolerance, transpose, write_intermediate)
      class(exptA_linop), intent(inout) :: A
      type(nek_dvector), intent(inout) :: X(:)
      complex(dp), allocatable, intent(out) :: eigvals(:)
      real(dp), allocatable, intent(out) :: residuals(:)
      integer, intent(out) :: info
      type(nek_dvector), optional, intent(in) :: x0
      integer, optional, intent(in) :: kdim
      real(dp), optional, intent(in) :: tolerance
      logical, optional, intent(in) :: transpose, write_intermediate
      type(nlg_eigs_opts) :: o
      type(c_ptr), allocatable :: hx(:)
      type(c_ptr) :: h0
      real(dp), allocatable :: re(:), im(:)
      integer(c_int) :: cinfo
      integer :: i, nev
      nev = size(X)
      call nlg_check(c_eigs_opts_default(o), 'nek_eigs')
      if (present(kdim)) o%kdim = kdim
      if (present(tolerance)) o%tol = tolerance
      if (present(transpose)) o%transpose = merge(1, 0, transpose)
      o%write_intermediate = 0
      if (present(write_intermediate)) o%write_intermediate = merge(1, 0, write_intermediate)
      allocate (hx(nev), re(nev), im(nev), residuals(nev), eigvals(nev))
      do i = 1, nev
         call nek_dvector_ensure(X(i)); hx(i) = X(i)%h
      end do
      h0 = c_null_ptr
      if (present(x0)) h0 = nek_dvector_handle(x0)
      call sync_tau(A, 'nek_eigs')
      call nlg_check(c_eigs(exptA_handle(A), hx, int(nev, c_int), re, im, residuals, cinfo, h0, o), 'nek_eigs')
      info = cinfo
      eigvals = cmplx(re, im, kind=dp)
   end subroutine

   !> svds(exptA, U, S, V, residuals, info, kdim=, write_intermediate=) of src/neklab_analysis.f90:136 (nlg_svds)
   subroutine nek_svds(A, U, S, V, residuals, info, u0, kdim, tolerance, write_intermediate)
      class(exptA_linop), intent(inout) :: A
      type(nek_dvector), intent(inout) :: U(:), V(:)
      real(dp), allocatable, intent(out) :: S(:), residuals(:)
      integer, intent(out) :: info
      type(nek_dvector), optional, intent(in) :: u0
      integer, optional, intent(in) :: kdim
      real(dp), optional, intent(in) :: tolerance
      logical, optional, intent(in) :: write_intermediate
      type(nlg_eigs_opts) :: o
      type(c_ptr), allocatable :: hu(:), hv(:)
      type(c_ptr) :: h0
      integer(c_int) :: cinfo
      integer :: i, nsv
      nsv = size(U)
      call nlg_check(c_eigs_opts_default(o), 'nek_svds')
      if (present(kdim)) o%kdim = kdim
      if (present(tolerance)) o%tol = tolerance
      o%write_intermediate = 0
      if (present(write_intermediate)) o%write_intermediate = merge(1, 0, write_intermediate)
      allocate (hu(nsv), hv(nsv), S(nsv), residuals(nsv))
      do i = 1, nsv
         call nek_dvector_ensure(U(i)); hu(i) = U(i)%h
         call nek_dvector_ensure(V(i)); hv(i) = V(i)%h
      end do
      h0 = c_null_ptr
      if (present(u0)) h0 = nek_dvector_handle(u0)
      call sync_tau(A, 'nek_svds')
      call nlg_check(c_svds(exptA_handle(A), hu, hv, int(nsv, c_int), S, residuals, cinfo, h0, o), 'nek_svds')
      info = cinfo
   end subroutine

end module neklab_linops
