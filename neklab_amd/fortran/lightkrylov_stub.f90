!> Minimal stand-in for the parts of LightKrylov that neklab's hot path extends.
!!
!! LightKrylov (nekStab/LightKrylov @ main, un-pinned, /root/reference/LightKrylov_setup.sh:55-57) is
!! not available in this image.  Only the abstract types are reproduced here, with the deferred
!! procedure signatures inferred from how the reference implements them:
!!   abstract_vector_rdp : zero, rand, scal, axpby, dot, get_size
!!       /root/reference/src/vectors/neklab_vectors.f90:39-44 and interfaces :65-93
!!   abstract_linop_rdp / abstract_exptA_linop_rdp : matvec, rmatvec, %tau
!!       /root/reference/src/linops/neklab_linops.f90:35-62, src/neklab_analysis.f90:84
!! With the real LightKrylov on the module path this file is simply left out of the build.
module LightKrylov
   use iso_fortran_env, only: real64
   implicit none
   private
   integer, parameter, public :: dp = real64

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
      function abstract_size(self) result(n)
         import abstract_vector_rdp
         class(abstract_vector_rdp), intent(in) :: self
         integer :: n
      end function
   end interface

   type, abstract, public :: abstract_linop_rdp
   contains
      procedure(abstract_matvec), pass(self), deferred, public :: matvec
      procedure(abstract_matvec), pass(self), deferred, public :: rmatvec
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

end module LightKrylov
