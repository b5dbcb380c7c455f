!> Drop-in for the parts of the reference's neklab_utils that the hot path uses: the field movers between the host
!! program's arrays and a nek_dvector, and the field-file writer.
!!   nek2vec / vec2nek   src/neklab_utils.f90:84-134 (nopcopy :279-301: pressure and temperature move with the velocity)
!!   outpost_dnek        src/neklab_utils.f90:305-333 (Nek5000 `outpost(vx, vy, vz, pr, t, prefix)`)
!! The host arrays are Nek5000's own (vx, vy, vz, pr, t or vxp, vyp, ... of SIZE/TOTAL): passed as they are, the lengths
!! come from the mesh handed to neklab_gpu_set_mesh.
module neklab_utils
   use iso_c_binding
   use LightKrylov, only: dp
   use neklab_gpu_capi
   use neklab_vectors
   implicit none
   private
   public :: nek2vec, vec2nek, outpost_dnek, set_outpost_session

   interface outpost_dnek
      module procedure outpost_dnek_vector
      module procedure outpost_dnek_basis
   end interface

   character(len=80), save :: session = 'neklab'
   integer, save :: nopen_prefix = 0
   character(len=3), save :: seen_prefix(64) = '   '
   integer, save :: seen_count(64) = 0

contains

   !> the session name Nek5000 puts between prefix and file number (`<prefix><session>0.f%05d`)
   subroutine set_outpost_session(name)
      character(len=*), intent(in) :: name
      session = name
   end subroutine

   subroutine nek2vec(vec, vx_, vy_, vz_, pr_, t_)
      type(nek_dvector), intent(inout) :: vec
      real(dp), intent(in) :: vx_(*), vy_(*), vz_(*), pr_(*), t_(*)
      call vec%zero()      ! intent(out) in the reference: a default-initialised vector (history cleared), then filled
      call nlg_check(c_vec_set_field(vec%h, 0_c_int, 0_c_int, vx_, nek_lvn), 'nek2vec vx')
      call nlg_check(c_vec_set_field(vec%h, 1_c_int, 0_c_int, vy_, nek_lvn), 'nek2vec vy')
      if (nek_ldim == 3) call nlg_check(c_vec_set_field(vec%h, 2_c_int, 0_c_int, vz_, nek_lvn), 'nek2vec vz')
      call nlg_check(c_vec_set_field(vec%h, 3_c_int, 0_c_int, pr_, nek_lpn), 'nek2vec pr')
      if (nek_nscal > 0) call nlg_check(c_vec_set_field(vec%h, 4_c_int, 0_c_int, t_, nek_lvn), 'nek2vec t')
   end subroutine

   subroutine vec2nek(vx_, vy_, vz_, pr_, t_, vec)
      real(dp), intent(inout) :: vx_(*), vy_(*), vz_(*), pr_(*), t_(*)
      type(nek_dvector), intent(in) :: vec
      type(c_ptr) :: h
      h = nek_dvector_handle(vec)
      call nlg_check(c_vec_get_field(h, 0_c_int, 0_c_int, vx_, nek_lvn), 'vec2nek vx')
      call nlg_check(c_vec_get_field(h, 1_c_int, 0_c_int, vy_, nek_lvn), 'vec2nek vy')
      if (nek_ldim == 3) call nlg_check(c_vec_get_field(h, 2_c_int, 0_c_int, vz_, nek_lvn), 'vec2nek vz')
      call nlg_check(c_vec_get_field(h, 3_c_int, 0_c_int, pr_, nek_lpn), 'vec2nek pr')
      if (nek_nscal > 0) call nlg_check(c_vec_get_field(h, 4_c_int, 0_c_int, t_, nek_lvn), 'vec2nek t')
   end subroutine

   !> one field file per call, numbered per prefix like Nek5000's outpost; the coordinates go into the first file of a prefix
   subroutine outpost_dnek_vector(vec, prefix)
      type(nek_dvector), intent(in) :: vec
      character(len=3), intent(in) :: prefix
      character(len=256) :: fname
      integer :: i, slot
      slot = 0
      do i = 1, nopen_prefix
         if (seen_prefix(i) == prefix) slot = i
      end do
      if (slot == 0) then
         nopen_prefix = min(nopen_prefix + 1, 64); slot = nopen_prefix
         seen_prefix(slot) = prefix; seen_count(slot) = 0
      end if
      seen_count(slot) = seen_count(slot) + 1
      write (fname, '(A,A,A,I5.5)') prefix, trim(session), '0.f', seen_count(slot)
      call nlg_check(c_vec_outpost(nek_dvector_handle(vec), trim(fname)//c_null_char, merge(1, 0, seen_count(slot) == 1), &
                                   0.0_c_double, int(seen_count(slot), c_int)), 'outpost_dnek')
   end subroutine

   subroutine outpost_dnek_basis(vec, prefix)
      type(nek_dvector), intent(in) :: vec(:)
      character(len=3), intent(in) :: prefix
      integer :: i
      do i = 1, size(vec)
         call outpost_dnek_vector(vec(i), prefix)
      end do
   end subroutine

end module neklab_utils
