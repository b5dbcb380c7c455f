!> Analysis drivers over the device hot path: the two entry points of the reference's module of the same name that consume it,
!!   linear_stability_analysis_fixed_point   (/root/reference/src/neklab_analysis.f90:38-105)
!!   transient_growth_analysis_fixed_point   (/root/reference/src/neklab_analysis.f90:107-156)
!! plus newton_fixed_point_iteration (:158-212) for the base-flow step upstream of them.  Same names, dummy arguments and
!! files written (`dir|adj_eigenspectrum.npy`, `singular_spectrum.dat`, the `dir` / `adj` / `prt` / `rsp` / `nwt` field files),
!! so a case file that calls them needs no edit.  With `device_eigs = .true.` the Krylov loop itself runs on the device
!! (nek_eigs / nek_svds -> nlg_eigs / nlg_svds: basis in one allocation, one fused projection kernel and one all-reduce per
!! Gram-Schmidt pass); with `.false.` LightKrylov's own eigs / svds drive the vectors and the operator through their
!! type-bound procedures.  Logger and timer plumbing of the reference is LightKrylov's and is not repeated here.
!! (In a build against the real LightKrylov the reference's own file works unchanged on top of neklab_vectors / neklab_linops;
!!  this module exists for the device switch.)
module neklab_analysis
   use LightKrylov, only: dp, eigs, svds, save_eigenspectrum, zero_basis, initialize_krylov_subspace, newton, gmres_rdp, &
                          newton_dp_opts, newton_dp_metadata
   use LightKrylov, only: abstract_vector_rdp, abstract_exptA_linop_rdp, abstract_system_rdp
   use neklab_vectors
   use neklab_linops
   use neklab_utils
   use neklab_systems, only: nek_constant_tol, nek_dynamic_tol
   implicit none
   private

   public :: linear_stability_analysis_fixed_point, transient_growth_analysis_fixed_point, newton_fixed_point_iteration
   !> .true.: Arnoldi / Lanczos on the device; .false.: LightKrylov's loops over the type-bound procedures
   logical, save, public :: device_eigs = .false.

contains

   !> Leading eigenpairs of exp(tau L) (adjoint: of its transpose) -> growth rates and frequencies log(mu) / tau.
   subroutine linear_stability_analysis_fixed_point(exptA, kdim, nev, adjoint, X0)
      class(abstract_exptA_linop_rdp), intent(inout) :: exptA
      integer, intent(in) :: kdim, nev
      logical, intent(in), optional :: adjoint
      type(nek_dvector), optional, intent(in) :: X0
      type(nek_dvector), allocatable :: modes(:)
      complex(dp), allocatable :: spectrum(:)
      real(dp), allocatable :: res(:)
      character(len=3) :: tag
      integer :: nmatvec
      logical :: transposed, on_device

      transposed = .false.
      if (present(adjoint)) transposed = adjoint
      tag = 'dir'
      if (transposed) tag = 'adj'

      allocate (modes(nev))
      call zero_basis(modes)

      on_device = .false.
      if (device_eigs) then
         select type (exptA)
         class is (exptA_linop)
            on_device = .true.
            call nek_eigs(exptA, modes, spectrum, res, nmatvec, x0=X0, kdim=kdim, transpose=transposed, write_intermediate=.true.)
         end select
      end if
      if (.not. on_device) then
         call eigs(exptA, modes, spectrum, res, nmatvec, x0=X0, kdim=kdim, transpose=transposed, write_intermediate=.true.)
      end if

      spectrum = log(spectrum)/exptA%tau                       ! multipliers of the propagator -> continuous-time eigenvalues
      call save_eigenspectrum(spectrum, res, tag//'_eigenspectrum.npy')
      call outpost_dnek(modes(:nev), tag)
      call exptA%finalize_timer()
   end subroutine linear_stability_analysis_fixed_point

   !> Largest singular triplets of exp(tau L): optimal perturbations (V, prefix prt), optimal responses (U, prefix rsp).
   subroutine transient_growth_analysis_fixed_point(exptA, nsv, kdim)
      class(abstract_exptA_linop_rdp), intent(inout) :: exptA
      integer, intent(in) :: nsv, kdim
      type(nek_dvector), allocatable :: resp(:), pert(:)
      real(dp), allocatable :: sigma(:), res(:)
      integer :: nmatvec, u
      logical :: on_device

      allocate (resp(nsv), pert(nsv))
      call initialize_krylov_subspace(resp)
      call initialize_krylov_subspace(pert)

      on_device = .false.
      if (device_eigs) then
         select type (exptA)
         class is (exptA_linop)
            on_device = .true.
            call nek_svds(exptA, resp, sigma, pert, res, nmatvec, kdim=kdim, write_intermediate=.true.)
         end select
      end if
      if (.not. on_device) call svds(exptA, resp, sigma, pert, res, nmatvec, kdim=kdim, write_intermediate=.true.)

      open (newunit=u, file='singular_spectrum.dat', status='replace', action='write')
      write (u, *) sigma
      close (u)
      call outpost_dnek(pert(:nsv), 'prt')
      call outpost_dnek(resp(:nsv), 'rsp')
      call exptA%finalize_timer()
   end subroutine transient_growth_analysis_fixed_point

   !> Newton-Krylov iteration for a fixed point of sys, at most 40 iterations, no bisection; tol_mode 1: constant solver
   !! tolerance, otherwise tied to the residual (the two schedulers of neklab_systems).  The converged state is written with the
   !! prefix nwt.
   subroutine newton_fixed_point_iteration(sys, bf, tol, tol_mode, input_is_fixed_point)
      class(abstract_system_rdp), intent(inout) :: sys
      class(abstract_vector_rdp), intent(inout) :: bf
      real(dp), intent(inout) :: tol
      integer, optional, intent(in) :: tol_mode
      logical, optional, intent(out) :: input_is_fixed_point
      integer :: info, mode
      type(newton_dp_opts) :: opts
      type(newton_dp_metadata) :: meta
      mode = 1
      if (present(tol_mode)) mode = tol_mode
      opts = newton_dp_opts(maxiter=40, ifbisect=.false.)      ! LightKrylov's own interface, as the reference calls it (neklab_analysis.f90:186-196)
      if (mode == 1) then
         call newton(sys, bf, gmres_rdp, info, atol=tol, options=opts, scheduler=nek_constant_tol, meta=meta)
      else
         call newton(sys, bf, gmres_rdp, info, atol=tol, options=opts, scheduler=nek_dynamic_tol, meta=meta)
      end if
      if (.not. meta%converged) write (*, '(A)') 'WARNING in newton_fixed_point_iteration: not converged after 40 iterations'
      select type (bf)
      type is (nek_dvector)
         call outpost_dnek(bf, 'nwt')
      end select
      if (present(input_is_fixed_point)) input_is_fixed_point = meta%input_is_fixed_point
      call sys%finalize_timer()
      call sys%jacobian%finalize_timer()
   end subroutine newton_fixed_point_iteration

end module neklab_analysis
