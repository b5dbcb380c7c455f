!> Drop-in for the fixed-point part of the reference's module of the same name (SURVEY.md 8f row 3):
!!   nek_system   / nek_jacobian        /root/reference/src/systems/neklab_systems.f90:42-55,  fixed_point.f90:4-96
!!   nek_temp_system / nek_temp_jacobian                      neklab_systems.f90:97-110, fixed_point_temp.f90 (the same with the
!!        temperature: here the case's `ifheat` decides, the types are the first pair under the reference's second names; the
!!        reference's umbrella module and its thermosyphon case spell them nek_system_temp / nek_jacobian_temp,
!!        src/neklab.f90:63-64, examples/thermosyphon/baseflow/tsyphon.usr:13,38-39 -- both spellings exist here)
!!   nek_constant_tol / nek_dynamic_tol  neklab_systems.f90:229-335
!! The reference integrates with Nek5000's global solver state; here each system / Jacobian owns a device propagator
!! (exptA_linop) over the horizon `endtime` of the case (neklab_gpu_set_case(endtime=..): Nek5000's endTime, which
!! setup_nonlinear_solver integrates to) and forwards:
!!   response(X, F, atol)  ->  nlg_linop_nonlinear_map at tolerances 0.1 atol, CFL limit 0.4   (fixed_point.f90:4-38)
!!   jacobian%matvec       ->  nlg_linop_set_baseflow(X), tolerances 0.5 atol, exptA matvec, minus the input
!!                              (fixed_point.f90:40-96); rmatvec likewise with the adjoint propagator
module neklab_systems
   use iso_c_binding
   use LightKrylov, only: dp, atol_dp, abstract_vector_rdp, abstract_system_rdp, abstract_jacobian_linop_rdp, type_error
   use neklab_gpu_capi
   use neklab_vectors
   use neklab_linops
   implicit none
   private
   character(len=*), parameter, private :: this_module = 'neklab_systems'

   public :: nek_constant_tol, nek_dynamic_tol

   type, extends(abstract_system_rdp), public :: nek_system
      type(exptA_linop), allocatable, private :: prop      ! allocatable: `nek_system()` (tsyphon.usr:38) names no component
      logical, private :: ready = .false.
   contains
      private
      procedure, pass(self), public :: response => nonlinear_map
   end type nek_system

   type, extends(abstract_jacobian_linop_rdp), public :: nek_jacobian
      type(exptA_linop), allocatable, private :: prop
      logical, private :: ready = .false.
   contains
      private
      procedure, pass(self), public :: matvec => jac_exptA_matvec
      procedure, pass(self), public :: rmatvec => jac_exptA_rmatvec
   end type nek_jacobian

   type, extends(nek_system), public :: nek_temp_system
   end type
   type, extends(nek_jacobian), public :: nek_temp_jacobian
   end type
   type, extends(nek_system), public :: nek_system_temp
   end type
   type, extends(nek_jacobian), public :: nek_jacobian_temp
   end type

   !> the solver tolerance the schedulers last chose = what Nek5000 keeps in param(21) / param(22) (neklab_systems.f90:261-264)
   !> the reference's param(22) (Nek5000's velocity tolerance), which its schedulers, nonlinear_map and the Jacobian products read and
   !! write in turn (neklab_systems.f90:259-260, fixed_point.f90:14-17, :49-62, :87): ONE variable here too, so that the tolerance of
   !! a Jacobian product depends on who set it last exactly as it does there
   real(dp), save, private :: solver_tol = 1.0e-9_dp

contains

   subroutine make_propagator(prop, about, cfl_limit)
      type(exptA_linop), intent(inout) :: prop
      type(nek_dvector), intent(in) :: about
      real(dp), intent(in) :: cfl_limit
      prop%tau = nek_endtime
      prop%baseflow = about
      prop%cfg = nek_case
      prop%cfg%cfl_limit = cfl_limit
      prop%cfg_set = .true.
      call prop%init()
   end subroutine

   subroutine nonlinear_map(self, vec_in, vec_out, atol)
      class(nek_system), intent(inout) :: self
      class(abstract_vector_rdp), intent(in) :: vec_in
      class(abstract_vector_rdp), intent(out) :: vec_out
      real(dp), intent(in) :: atol
      select type (vec_in)
      type is (nek_dvector)
         select type (vec_out)
         type is (nek_dvector)
            if (.not. self%ready) then
               allocate (self%prop)
               call make_propagator(self%prop, vec_in, 0.4_dp)
               self%ready = .true.
            end if
            call self%prop%set_tolerances(0.1_dp*atol, 0.1_dp*atol)
            solver_tol = 0.1_dp*atol      ! setup_nonlinear_solver(vtol = atol*0.1) leaves param(22) at this value (neklab_nek_setup.f90:228)
            call self%prop%nonlinear_map(vec_in, vec_out)      ! Phi_T(X) - X; the time step follows the CFL number of X
         class default
            call type_error('vec_out', 'nek_dvector', 'OUT', this_module, 'nonlinear_map')
         end select
      class default
         call type_error('vec_in', 'nek_dvector', 'IN', this_module, 'nonlinear_map')
      end select
   end subroutine nonlinear_map

   subroutine jac_apply(self, vec_in, vec_out, transposed)
      class(nek_jacobian), intent(inout) :: self
      class(abstract_vector_rdp), intent(in) :: vec_in
      class(abstract_vector_rdp), intent(out) :: vec_out
      logical, intent(in) :: transposed
      if (.not. allocated(self%X)) then
         write (*, '(A)') 'ERROR in '//this_module//': jacobian%X is not set (tsyphon.usr:40)'
         error stop 1
      end if
      select type (state => self%X)
      type is (nek_dvector)
         ! linearise about the current X on every application, as the reference does (abs_vec2nek(.., self%X) and
         ! setup_linear_solver(recompute_dt = .true.) inside jac_exptA_matvec, fixed_point.f90:52-59): X is updated in place by
         ! Newton, so there is no cheaper way to know that it is still the state of the last call
         if (.not. self%ready) then
            allocate (self%prop)
            call make_propagator(self%prop, state, 0.5_dp)
            self%ready = .true.
         else
            call self%prop%set_baseflow(state)
         end if
         ! fixed_point.f90:49-62: atol = param(22) AS IT STANDS at the call -- the value the scheduler wrote if it ran last, a tenth of it
         ! if nonlinear_map ran last -- the solves run at half of it, and param(22) is put back afterwards (:87)
         call self%prop%set_tolerances(0.5_dp*solver_tol, 0.5_dp*solver_tol)
         if (transposed) then
            call self%prop%rmatvec(vec_in, vec_out)
         else
            call self%prop%matvec(vec_in, vec_out)
         end if
         call vec_out%sub(vec_in)                       ! [exp(T J) - I] dx
      class default
         call type_error('self%X', 'nek_dvector', 'IN', this_module, 'jac_exptA_matvec')
      end select
   end subroutine

   subroutine jac_exptA_matvec(self, vec_in, vec_out)
      class(nek_jacobian), intent(inout) :: self
      class(abstract_vector_rdp), intent(in) :: vec_in
      class(abstract_vector_rdp), intent(out) :: vec_out
      call jac_apply(self, vec_in, vec_out, .false.)
   end subroutine

   subroutine jac_exptA_rmatvec(self, vec_in, vec_out)
      class(nek_jacobian), intent(inout) :: self
      class(abstract_vector_rdp), intent(in) :: vec_in
      class(abstract_vector_rdp), intent(out) :: vec_out
      call jac_apply(self, vec_in, vec_out, .true.)
   end subroutine

   !> constant solver tolerance = the target, never below 10 atol_dp
   subroutine nek_constant_tol(tol, target_tol, rnorm, iter, info)
      real(dp), intent(out) :: tol
      real(dp), intent(in) :: target_tol, rnorm
      integer, intent(in) :: iter
      integer, intent(out) :: info
      tol = max(target_tol, 10.0_dp*atol_dp)
      solver_tol = tol
      info = 0
   end subroutine

   !> solver tolerance a tenth of the current residual, between the target and 1e-4; the target itself once within a factor 10
   subroutine nek_dynamic_tol(tol, target_tol, rnorm, iter, info)
      real(dp), intent(out) :: tol
      real(dp), intent(in) :: target_tol, rnorm
      integer, intent(in) :: iter
      integer, intent(out) :: info
      real(dp), parameter :: loosest = 1.0e-4_dp
      real(dp) :: goal
      goal = min(max(target_tol, 10.0_dp*atol_dp), loosest)
      tol = max(0.1_dp*rnorm, goal)
      if (tol < 10.0_dp*goal) tol = goal
      tol = min(tol, loosest)
      solver_tol = tol
      info = 0
   end subroutine

end module neklab_systems
