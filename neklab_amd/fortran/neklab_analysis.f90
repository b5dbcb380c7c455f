!> Drop-in for the hot-path driver of the reference's module of the same name:
!! linear_stability_analysis_fixed_point, src/neklab_analysis.f90:38-105.  The body is the reference's call sequence
!! (:77-93) line for line, with the same `use` names; logger / timer plumbing (LightKrylov_Logger, LightKrylov_Timing) is
!! out of scope and left out.  `eigs` is LightKrylov's (here: the stand-in of lightkrylov_stub.f90, which drives the
!! vectors and the operator through their type-bound procedures only); `device_eigs = .true.` switches to the device
!! block path (nek_eigs -> nlg_eigs) with the same arguments.
module neklab_analysis
   use LightKrylov, only: dp, eigs, save_eigenspectrum
   use LightKrylov, only: zero_basis
   use LightKrylov_AbstractVectors, only: abstract_vector_rdp
   use LightKrylov_AbstractLinops, only: abstract_exptA_linop_rdp
   use neklab_vectors
   use neklab_linops
   use neklab_utils
   implicit none
   private
   character(len=*), parameter, private :: this_module = 'neklab_analysis'

   public :: linear_stability_analysis_fixed_point
   logical, save, public :: device_eigs = .false.

contains

   subroutine linear_stability_analysis_fixed_point(exptA, kdim, nev, adjoint, X0)
      class(abstract_exptA_linop_rdp), intent(inout) :: exptA
      !! Operator whose stability properties are to be investigated.
      integer, intent(in) :: kdim
      !! Maximum dimension of the Krylov subspace.
      integer, intent(in) :: nev
      !! Desired number of eigenpairs to converge.
      logical, intent(in), optional :: adjoint
      !! Whether direct or adjoint analysis should be conducted.
      type(nek_dvector), optional, intent(in) :: X0
      !! Initial guess for the eigenvectors

      type(nek_dvector), allocatable :: eigvecs(:)
      complex(kind=dp), allocatable :: eigvals(:)
      real(kind=dp), allocatable :: residuals(:)
      integer :: info
      logical :: adjoint_
      character(len=3) :: file_prefix

      ! Optional parameters.
      if (present(adjoint)) then
         adjoint_ = adjoint
      else
         adjoint_ = .false.
      end if

      ! Allocate eigenvectors and initialize Krylov basis.
      allocate (eigvecs(nev)); call zero_basis(eigvecs)

      ! Run the eigenvalue analysis.
      if (device_eigs) then
         select type (exptA)
         class is (exptA_linop)
            call nek_eigs(exptA, eigvecs, eigvals, residuals, info, x0=X0, kdim=kdim, &
                          transpose=adjoint_, write_intermediate=.true.)
         end select
      else
         call eigs(exptA, eigvecs, eigvals, residuals, info, x0=X0, kdim=kdim, &
                   transpose=adjoint_, write_intermediate=.true.)
      end if

      ! Transform eigenspectrum to continuous-time representation.
      eigvals = log(eigvals)/exptA%tau

      ! Determine the file prefix.
      file_prefix = merge("adj", "dir", adjoint_)

      ! Save eigenspectrum to disk.
      call save_eigenspectrum(eigvals, residuals, trim(file_prefix)//"_eigenspectrum.npy")

      ! Export eigenfunctions to disk.
      call outpost_dnek(eigvecs(:nev), file_prefix)

      ! Finalize exptA timings
      call exptA%finalize_timer()

   end subroutine linear_stability_analysis_fixed_point

end module neklab_analysis
