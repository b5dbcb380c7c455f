!> Drop-in for the reference's module of the same name: `nek_dvector` with every type-bound procedure forwarded through
!! ISO_C_BINDING to libneklab_gpu.so (include/neklab_gpu.h).
!!
!! Mapping (file:line under /root/reference):
!!   type nek_dvector                     src/vectors/neklab_vectors.f90:26-50
!!   constructor nek_dvector(vx, vy, ..)  src/vectors/neklab_vectors.f90:53-61
!!   zero/rand/scal/axpby/dot/get_size    src/vectors/real_vectors.f90:37-247
!!   save_rst/get_rst/has_rst_fields/clear_rst_fields               :249-346
!!
!! Object semantics.  The reference's vectors are plain static arrays: intrinsic assignment, sourced allocation and
!! structure constructors deep-copy them (SURVEY.md 7.3 item 5).  Here the fields live in HBM behind an opaque handle, so
!!   * `=` is a defined assignment (clone / copy) and the type has a finaliser;
!!   * every object remembers the address it was created at (`owner`) and the generation number of its handle (`gen`).  A
!!     bitwise copy made behind the type's back (allocate(.., source=), an array constructor, reallocation on assignment,
!!     a structure-constructor component) has the same handle at another address.  It is recognised on its first use as an
!!     inout argument and ADOPTS the handle (nlg_vec_adopt, include/neklab_gpu.h): if the original has been finalised in the
!!     meantime -- `X = [X, v]` finalises the old X after the temporary has been built -- the copy simply becomes the owner
!!     (no copy, no leak, no dangling pointer); if the original is alive, the copy gets a clone of its own.  A finaliser
!!     therefore RELEASES its handle instead of destroying it; released buffers are freed by the library (oldest first above
!!     nlg_vec_pool_limit, on allocation failure, at neklab_gpu_finalize).  move_alloc keeps the address: nothing to do.
!!     What cannot be had: a copy READS through the original's handle until its first inout use, so it sees changes made
!!     to the original in between (the reference's copies do not); copy, then modify the original, then read the copy is
!!     the one sequence to avoid -- LightKrylov does not contain it;
!!   * handles are created on first use, so `allocate(X(k))`, `intent(out)` dummies and `mold=` allocations cost nothing.
!! A non-zero return code from the C ABI becomes `error stop` with nlg_last_error(), which is what stop_error does in the
!! reference (src/neklab_nek_setup.f90:406-417); a wrong dynamic type calls type_error as the reference does.
module neklab_vectors
   use iso_c_binding
   use LightKrylov, only: dp, abstract_vector_rdp, abstract_vector_cdp, type_error
   use neklab_gpu_capi
   implicit none
   private
   character(len=*), parameter, private :: this_module = 'neklab_vectors'

   public :: nek_dvector_handle, nek_dvector_ensure

   type, extends(abstract_vector_rdp), public :: nek_dvector
      type(c_ptr) :: h = c_null_ptr
      integer(c_intptr_t) :: owner = 0
      integer(c_int64_t) :: gen = 0
   contains
      private
      procedure, pass(self), public :: zero => nek_dzero
      procedure, pass(self), public :: rand => nek_drand
      procedure, pass(self), public :: scal => nek_dscal
      procedure, pass(self), public :: axpby => nek_daxpby
      procedure, pass(self), public :: dot => nek_ddot
      procedure, pass(self), public :: get_size => nek_dsize
      procedure, pass(self), public :: save_rst => dsave_rst
      procedure, pass(self), public :: get_rst => dget_rst
      procedure, pass(self), public :: has_rst_fields => dhas_rst_fields
      procedure, pass(self), public :: clear_rst_fields => dclear_rst_fields
      procedure, pass(lhs) :: assign_dvector
      generic, public :: assignment(=) => assign_dvector
      final :: finalize_dvector, finalize_dvector_rank1
   end type nek_dvector

   !> Complex vector as a (re, im) pair of real ones (reference: type nek_zvector, src/vectors/neklab_vectors.f90, complex_vectors.f90:
   !! the forcing and the response of the resolvent operator).  The inner product is <a, b> = sum conj(a) b in the mass-weighted
   !! velocity inner product of nek_dvector.
   type, extends(abstract_vector_cdp), public :: nek_zvector
      type(nek_dvector) :: re, im
   contains
      private
      procedure, pass(self), public :: zero => nek_zzero
      procedure, pass(self), public :: rand => nek_zrand
      procedure, pass(self), public :: scal => nek_zscal
      procedure, pass(self), public :: axpby => nek_zaxpby
      procedure, pass(self), public :: dot => nek_zdot
      procedure, pass(self), public :: get_size => nek_zsize
   end type nek_zvector

   ! --> Constructor (reference: construct_nek_dvector, neklab_vectors.f90:53-61; not pure: it allocates device memory)
   interface nek_dvector
      module procedure construct_nek_dvector
   end interface

contains

   !> make sure `self` has a handle of its own (see the header): create on first use, clone if `self` is a bitwise copy
   subroutine nek_dvector_ensure(self)
      class(nek_dvector), intent(inout) :: self
      type(c_ptr) :: hnew
      integer(c_int) :: status
      if (.not. c_associated(self%h)) then
         call nlg_check(c_vec_create(nlg_mesh, int(nek_nscal, c_int), int(nek_lorder, c_int), self%h), 'nek_dvector allocate')
      else if (self%owner /= loc(self)) then      ! a bitwise copy: take over a released handle, clone a live one
         call nlg_check(c_vec_adopt(self%h, self%gen, status), 'nek_dvector adopt')
         if (status == 0) then
            call nlg_check(c_vec_clone(self%h, hnew), 'nek_dvector copy-on-detect')
            self%h = hnew
         end if
      else      ! the owner by address -- or a bitwise copy that landed where its original stood (`X = [X, v]` grown in place), whose
         !        handle the original's finaliser has released: validate the generation and take a released handle back
         call nlg_check(c_vec_adopt(self%h, self%gen, status), 'nek_dvector (owner whose handle has been freed)')
         return
      end if
      self%owner = loc(self)
      call nlg_check(c_vec_generation(self%h, self%gen), 'nek_dvector generation')
   end subroutine

   !> handle for a read-only use (a bitwise copy may read through the original's handle)
   function nek_dvector_handle(self) result(h)
      class(nek_dvector), intent(in) :: self
      type(c_ptr) :: h
      integer(c_int) :: status
      if (.not. c_associated(self%h)) then
         write (*, '(A)') 'ERROR in '//this_module//': use of a nek_dvector that holds no data yet'
         error stop 1
      end if
      if (self%owner /= loc(self)) then      ! a bitwise copy reads through the original's handle: it must still be that handle (same
         !                                     generation); a released one is pinned, so that it cannot be evicted under the reader
         call nlg_check(c_vec_pin(self%h, self%gen), 'nek_dvector (bitwise copy whose original has been freed)')
      else                                   ! owner by address: see nek_dvector_ensure
         call nlg_check(c_vec_adopt(self%h, self%gen, status), 'nek_dvector (owner whose handle has been freed)')
      end if
      h = self%h
   end function

   function construct_nek_dvector(vx, vy, vz, pr, theta) result(out)
      real(dp), intent(in) :: vx(*), vy(*)
      real(dp), optional, intent(in) :: vz(*), pr(*), theta(*)
      type(nek_dvector) :: out
      call nek_dvector_ensure(out)
      call nlg_check(c_vec_zero(out%h), 'construct_nek_dvector')
      call nlg_check(c_vec_set_field(out%h, 0_c_int, 0_c_int, vx, nek_lvn), 'construct_nek_dvector')
      call nlg_check(c_vec_set_field(out%h, 1_c_int, 0_c_int, vy, nek_lvn), 'construct_nek_dvector')
      if (present(vz) .and. nek_ldim == 3) call nlg_check(c_vec_set_field(out%h, 2_c_int, 0_c_int, vz, nek_lvn), 'construct_nek_dvector')
      if (present(pr)) call nlg_check(c_vec_set_field(out%h, 3_c_int, 0_c_int, pr, nek_lpn), 'construct_nek_dvector')
      if (present(theta) .and. nek_nscal > 0) call nlg_check(c_vec_set_field(out%h, 4_c_int, 0_c_int, theta, nek_lvn), 'construct_nek_dvector')
   end function

   !-----------------------------------------
   !-----     TYPE-BOUND PROCEDURES     -----
   !-----------------------------------------
   subroutine nek_dzero(self)
      class(nek_dvector), intent(inout) :: self
      integer(c_int) :: status
      if (c_associated(self%h) .and. self%owner /= loc(self)) then      ! a bitwise copy about to be overwritten: no clone needed
         call nlg_check(c_vec_adopt(self%h, self%gen, status), 'nek_dzero adopt')
         if (status == 1) then
            self%owner = loc(self)
         else
            self%h = c_null_ptr
         end if
      end if
      call nek_dvector_ensure(self)
      call nlg_check(c_vec_zero(self%h), 'nek_dzero')
   end subroutine

   subroutine nek_drand(self, ifnorm)
      class(nek_dvector), intent(inout) :: self
      logical, optional, intent(in) :: ifnorm
      integer(c_int) :: nrm
      integer(c_int64_t), save :: seed = 0
      nrm = 0
      if (present(ifnorm)) nrm = merge(1, 0, ifnorm)
      call nek_dvector_ensure(self)
      seed = seed + 1     ! successive calls draw different fields, like random_number in the reference
      call nlg_check(c_vec_rand(self%h, nrm, seed), 'nek_drand')
   end subroutine

   subroutine nek_dscal(self, alpha)
      class(nek_dvector), intent(inout) :: self
      real(dp), intent(in) :: alpha
      call nek_dvector_ensure(self)
      call nlg_check(c_vec_scal(self%h, alpha), 'nek_dscal')
   end subroutine

   subroutine nek_daxpby(alpha, vec, beta, self)
      class(nek_dvector), intent(inout) :: self
      real(dp), intent(in) :: alpha
      class(abstract_vector_rdp), intent(in) :: vec
      real(dp), intent(in) :: beta
      call nek_dvector_ensure(self)
      select type (vec)
      type is (nek_dvector)
         call nlg_check(c_vec_axpby(alpha, nek_dvector_handle(vec), beta, self%h), 'nek_daxpby')
      class default
         call type_error('vec', 'nek_dvector', 'IN', this_module, 'nek_daxpby')      ! real_vectors.f90:202-204
      end select
   end subroutine

   function nek_ddot(self, vec) result(alpha)
      class(nek_dvector), intent(in) :: self
      class(abstract_vector_rdp), intent(in) :: vec
      real(dp) :: alpha
      alpha = 0.0_dp
      select type (vec)
      type is (nek_dvector)
         call nlg_check(c_vec_dot(nek_dvector_handle(self), nek_dvector_handle(vec), alpha), 'nek_ddot')
      class default
         call type_error('vec', 'nek_dvector', 'IN', this_module, 'nek_ddot')        ! real_vectors.f90:229-231
      end select
   end function

   pure function nek_dsize(self) result(n)
      class(nek_dvector), intent(in) :: self
      integer :: n
      if (c_associated(self%h)) then
         n = int(c_vec_size_checked(self%h, self%gen))
         if (n < 0) error stop 'neklab_vectors::nek_dsize: this nek_dvector is a bitwise copy of a vector that has been freed since'
      else      ! not materialised yet: the size is a property of the mesh and the case (real_vectors.f90:235-247)
         n = int((nek_ldim + nek_nscal)*nek_lvn + nek_lpn)
      end if
   end function

   subroutine dsave_rst(self, vec_rst, irst)
      class(nek_dvector), intent(inout) :: self
      class(abstract_vector_rdp), intent(in) :: vec_rst
      integer, intent(in) :: irst
      call nek_dvector_ensure(self)
      select type (vec_rst)
      type is (nek_dvector)
         call nlg_check(c_vec_save_rst(self%h, nek_dvector_handle(vec_rst), int(irst, c_int)), 'dsave_rst')
      class default
         call type_error('vec_rst', 'nek_dvector', 'IN', this_module, 'dsave_rst')     ! real_vectors.f90:286-288
      end select
   end subroutine

   subroutine dget_rst(self, vec_rst, irst)
      class(nek_dvector), intent(in) :: self
      class(abstract_vector_rdp), intent(inout) :: vec_rst
      integer, intent(in) :: irst
      select type (vec_rst)
      type is (nek_dvector)
         call nek_dvector_ensure(vec_rst)
         call nlg_check(c_vec_get_rst(nek_dvector_handle(self), vec_rst%h, int(irst, c_int)), 'dget_rst')
      class default
         call type_error('vec_rst', 'nek_dvector', 'OUT', this_module, 'dget_rst')     ! real_vectors.f90:329-331
      end select
   end subroutine

   pure function dhas_rst_fields(self) result(has_rst_fields)      ! `pure` as in the reference (neklab_vectors.f90:107-110)
      class(nek_dvector), intent(in) :: self
      logical :: has_rst_fields
      integer(c_int) :: flag
      has_rst_fields = .false.
      if (c_associated(self%h)) then      ! never dereferences a stale handle: unknown handle or another generation -> -1
         flag = c_vec_has_rst_checked(self%h, self%gen)
         if (flag < 0) error stop 'neklab_vectors::dhas_rst_fields: this nek_dvector is a bitwise copy of a vector that has been freed since'
         has_rst_fields = flag /= 0
      end if
   end function

   subroutine dclear_rst_fields(self)
      class(nek_dvector), intent(inout) :: self
      call nek_dvector_ensure(self)
      call nlg_check(c_vec_clear_rst(self%h), 'dclear_rst_fields')
   end subroutine

   !---- nek_zvector: every operation is the complex arithmetic of the pair, through the real vector's procedures ------------
   subroutine nek_zzero(self)
      class(nek_zvector), intent(inout) :: self
      call self%re%zero(); call self%im%zero()
   end subroutine

   subroutine nek_zrand(self, ifnorm)
      class(nek_zvector), intent(inout) :: self
      logical, optional, intent(in) :: ifnorm
      real(dp) :: nrm
      call self%re%rand(.false.); call self%im%rand(.false.)
      if (present(ifnorm)) then
         if (ifnorm) then
            nrm = sqrt(self%re%dot(self%re) + self%im%dot(self%im))
            call self%re%scal(1.0_dp/nrm); call self%im%scal(1.0_dp/nrm)
         end if
      end if
   end subroutine

   subroutine nek_zscal(self, alpha)
      class(nek_zvector), intent(inout) :: self
      complex(dp), intent(in) :: alpha
      type(nek_dvector) :: old_re
      old_re = self%re
      call self%re%axpby(-aimag(alpha), self%im, real(alpha, dp))      ! re <- Re(a) re - Im(a) im
      call self%im%axpby(aimag(alpha), old_re, real(alpha, dp))        ! im <- Re(a) im + Im(a) re
   end subroutine

   subroutine nek_zaxpby(alpha, vec, beta, self)
      class(nek_zvector), intent(inout) :: self
      complex(dp), intent(in) :: alpha
      class(abstract_vector_cdp), intent(in) :: vec
      complex(dp), intent(in) :: beta
      select type (vec)
      type is (nek_zvector)
         call nek_zscal(self, beta)
         call self%re%axpby(real(alpha, dp), vec%re, 1.0_dp); call self%re%axpby(-aimag(alpha), vec%im, 1.0_dp)
         call self%im%axpby(real(alpha, dp), vec%im, 1.0_dp); call self%im%axpby(aimag(alpha), vec%re, 1.0_dp)
      class default
         write (*, '(A)') 'ERROR in '//this_module//"::nek_zaxpby: the intent [IN] argument 'vec' must be of type 'nek_zvector'"
         error stop 1
      end select
   end subroutine

   function nek_zdot(self, vec) result(alpha)
      class(nek_zvector), intent(in) :: self
      class(abstract_vector_cdp), intent(in) :: vec
      complex(dp) :: alpha
      alpha = (0.0_dp, 0.0_dp)
      select type (vec)
      type is (nek_zvector)
         alpha = cmplx(self%re%dot(vec%re) + self%im%dot(vec%im), self%re%dot(vec%im) - self%im%dot(vec%re), kind=dp)
      class default
         write (*, '(A)') 'ERROR in '//this_module//"::nek_zdot: the intent [IN] argument 'vec' must be of type 'nek_zvector'"
         error stop 1
      end select
   end function

   pure function nek_zsize(self) result(n)
      class(nek_zvector), intent(in) :: self
      integer :: n
      n = 2*self%re%get_size()
   end function

   !> intrinsic-assignment semantics of the reference's by-value vectors: deep copy
   subroutine assign_dvector(lhs, rhs)
      class(nek_dvector), intent(inout) :: lhs
      class(nek_dvector), intent(in) :: rhs
      integer(c_int) :: rc, status
      if (loc(lhs) == loc(rhs)) return
      if (c_associated(lhs%h) .and. lhs%owner /= loc(lhs)) then      ! lhs was a bitwise copy: own the handle or drop the alias
         call nlg_check(c_vec_adopt(lhs%h, lhs%gen, status), 'nek_dvector assignment')
         if (status == 1) then
            lhs%owner = loc(lhs)
         else
            lhs%h = c_null_ptr
         end if
      end if
      if (.not. c_associated(rhs%h)) then      ! rhs holds nothing: lhs becomes empty too
         if (c_associated(lhs%h)) rc = c_vec_release(lhs%h)
         lhs%h = c_null_ptr; lhs%owner = 0; lhs%gen = 0
         return
      end if
      if (c_associated(lhs%h)) then
         if (c_associated(lhs%h, rhs%h)) return      ! rhs is a bitwise copy of lhs, which owns the handle: same data already
         call nlg_check(c_vec_copy(lhs%h, rhs%h), 'nek_dvector assignment')
      else
         call nlg_check(c_vec_clone(rhs%h, lhs%h), 'nek_dvector assignment')
      end if
      lhs%owner = loc(lhs)
      call nlg_check(c_vec_generation(lhs%h, lhs%gen), 'nek_dvector assignment')
   end subroutine

   subroutine finalize_dvector(self)
      type(nek_dvector), intent(inout) :: self
      integer(c_int) :: rc
      if (c_associated(self%h)) then
         if (self%owner == loc(self)) then
            rc = c_vec_release(self%h)      ! a bitwise copy may still adopt it
         else
            rc = c_vec_unpin(self%h, self%gen)      ! a reader that pinned a released handle gives it back to the pool
         end if
      end if
      self%h = c_null_ptr; self%owner = 0; self%gen = 0
   end subroutine

   subroutine finalize_dvector_rank1(self)
      type(nek_dvector), intent(inout) :: self(:)
      integer :: i
      integer(c_int) :: rc
      do i = 1, size(self)
         if (c_associated(self(i)%h)) then
            if (self(i)%owner == loc(self(i))) then
               rc = c_vec_release(self(i)%h)
            else
               rc = c_vec_unpin(self(i)%h, self(i)%gen)
            end if
         end if
         self(i)%h = c_null_ptr; self(i)%owner = 0; self(i)%gen = 0
      end do
   end subroutine

end module neklab_vectors
