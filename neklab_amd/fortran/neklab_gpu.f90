!> Fortran 2008 shim: the reference's `nek_dvector` and `exptA_linop` with every type-bound procedure
!! forwarded through ISO_C_BINDING to libneklab_gpu.so (include/neklab_gpu.h).
!!
!! Drop-in mapping (file:line under /root/reference):
!!   type nek_dvector                     src/vectors/neklab_vectors.f90:26-50
!!     zero/rand/scal/axpby/dot/get_size  src/vectors/real_vectors.f90:37-247
!!     save_rst/get_rst/has_rst_fields/clear_rst_fields          :249-346
!!   type exptA_linop                     src/linops/neklab_linops.f90:35-44
!!     init/matvec/rmatvec                src/linops/exponential_propagator.f90:4-107
!! The reference's vectors are static arrays, so intrinsic assignment and sourced allocation deep-copy
!! them (SURVEY.md §7.3 item 5); here the fields live in HBM behind an opaque handle, therefore the type
!! carries defined assignment (clone) and a finaliser (destroy).  A non-zero return code from the C ABI
!! becomes `error stop` with nlg_last_error(), which is what LightKrylov's stop_error does in the
!! reference (src/neklab_nek_setup.f90:406-417).
module neklab_gpu
   use iso_c_binding
   use LightKrylov, only: dp, abstract_vector_rdp, abstract_exptA_linop_rdp
   implicit none
   private

   public :: nek_dvector, exptA_linop, nlg_check
   public :: neklab_gpu_init, neklab_gpu_set_mesh, neklab_gpu_finalize
   public :: nek2vec_host, vec2nek_host

   !> process-wide handles: the reference keeps the same information in Nek5000 commons
   type(c_ptr), save, public :: nlg_ctx = c_null_ptr
   type(c_ptr), save, public :: nlg_mesh = c_null_ptr

   type, bind(C), public :: nlg_mesh_desc
      integer(c_int) :: dim, n, lxd
      integer(c_int64_t) :: nelv
      type(c_ptr) :: xm1, ym1, zm1, glo_num, lglel, v1mask, v2mask, v3mask, tmask
      integer(c_int) :: has_outflow
   end type

   type, bind(C), public :: nlg_exptA_config
      real(c_double) :: tau, re, cfl_limit, vtol, ptol, dt
      integer(c_int) :: torder, maxit_v, maxit_p, fixed_iters_v, fixed_iters_p, pprecond, pproj
      integer(c_int) :: ifheat = 0
      real(c_double) :: conductivity = 1.0_c_double, rhocp = 1.0_c_double, buoy(3) = 0.0_c_double
   end type

   interface
      function c_last_error() bind(C, name="nlg_last_error") result(p)
         import c_ptr
         type(c_ptr) :: p
      end function
      function c_ctx_create(device, ctx) bind(C, name="nlg_ctx_create") result(rc)
         import c_int, c_ptr
         integer(c_int), value :: device
         type(c_ptr), intent(out) :: ctx
         integer(c_int) :: rc
      end function
      function c_ctx_destroy(ctx) bind(C, name="nlg_ctx_destroy") result(rc)
         import c_int, c_ptr
         type(c_ptr), value :: ctx
         integer(c_int) :: rc
      end function
      function c_mesh_create(ctx, desc, mesh) bind(C, name="nlg_mesh_create") result(rc)
         import c_int, c_ptr, nlg_mesh_desc
         type(c_ptr), value :: ctx
         type(nlg_mesh_desc), intent(in) :: desc
         type(c_ptr), intent(out) :: mesh
         integer(c_int) :: rc
      end function
      function c_mesh_destroy(mesh) bind(C, name="nlg_mesh_destroy") result(rc)
         import c_int, c_ptr
         type(c_ptr), value :: mesh
         integer(c_int) :: rc
      end function
      function c_vec_create(mesh, nscal, lorder, v) bind(C, name="nlg_vec_create") result(rc)
         import c_int, c_ptr
         type(c_ptr), value :: mesh
         integer(c_int), value :: nscal, lorder
         type(c_ptr), intent(out) :: v
         integer(c_int) :: rc
      end function
      function c_vec_destroy(v) bind(C, name="nlg_vec_destroy") result(rc)
         import c_int, c_ptr
         type(c_ptr), value :: v
         integer(c_int) :: rc
      end function
      function c_vec_clone(src, v) bind(C, name="nlg_vec_clone") result(rc)
         import c_int, c_ptr
         type(c_ptr), value :: src
         type(c_ptr), intent(out) :: v
         integer(c_int) :: rc
      end function
      function c_vec_copy(dst, src) bind(C, name="nlg_vec_copy") result(rc)
         import c_int, c_ptr
         type(c_ptr), value :: dst, src
         integer(c_int) :: rc
      end function
      function c_vec_zero(v) bind(C, name="nlg_vec_zero") result(rc)
         import c_int, c_ptr
         type(c_ptr), value :: v
         integer(c_int) :: rc
      end function
      function c_vec_rand(v, ifnorm, seed) bind(C, name="nlg_vec_rand") result(rc)
         import c_int, c_ptr, c_int64_t
         type(c_ptr), value :: v
         integer(c_int), value :: ifnorm
         integer(c_int64_t), value :: seed
         integer(c_int) :: rc
      end function
      function c_vec_scal(v, alpha) bind(C, name="nlg_vec_scal") result(rc)
         import c_int, c_ptr, c_double
         type(c_ptr), value :: v
         real(c_double), value :: alpha
         integer(c_int) :: rc
      end function
      function c_vec_axpby(alpha, x, beta, self) bind(C, name="nlg_vec_axpby") result(rc)
         import c_int, c_ptr, c_double
         real(c_double), value :: alpha, beta
         type(c_ptr), value :: x, self
         integer(c_int) :: rc
      end function
      function c_vec_dot(a, b, res) bind(C, name="nlg_vec_dot") result(rc)
         import c_int, c_ptr, c_double
         type(c_ptr), value :: a, b
         real(c_double), intent(out) :: res
         integer(c_int) :: rc
      end function
      function c_vec_size(v, n) bind(C, name="nlg_vec_size") result(rc)
         import c_int, c_ptr, c_int64_t
         type(c_ptr), value :: v
         integer(c_int64_t), intent(out) :: n
         integer(c_int) :: rc
      end function
      function c_vec_save_rst(self, v, irst) bind(C, name="nlg_vec_save_rst") result(rc)
         import c_int, c_ptr
         type(c_ptr), value :: self, v
         integer(c_int), value :: irst
         integer(c_int) :: rc
      end function
      function c_vec_get_rst(self, v, irst) bind(C, name="nlg_vec_get_rst") result(rc)
         import c_int, c_ptr
         type(c_ptr), value :: self, v
         integer(c_int), value :: irst
         integer(c_int) :: rc
      end function
      function c_vec_has_rst(self, flag) bind(C, name="nlg_vec_has_rst_fields") result(rc)
         import c_int, c_ptr
         type(c_ptr), value :: self
         integer(c_int), intent(out) :: flag
         integer(c_int) :: rc
      end function
      function c_vec_clear_rst(self) bind(C, name="nlg_vec_clear_rst_fields") result(rc)
         import c_int, c_ptr
         type(c_ptr), value :: self
         integer(c_int) :: rc
      end function
      function c_vec_set_field(v, field, irst, host, count) bind(C, name="nlg_vec_set_field") result(rc)
         import c_int, c_ptr, c_double, c_int64_t
         type(c_ptr), value :: v
         integer(c_int), value :: field, irst
         real(c_double), intent(in) :: host(*)
         integer(c_int64_t), value :: count
         integer(c_int) :: rc
      end function
      function c_vec_get_field(v, field, irst, host, count) bind(C, name="nlg_vec_get_field") result(rc)
         import c_int, c_ptr, c_double, c_int64_t
         type(c_ptr), value :: v
         integer(c_int), value :: field, irst
         real(c_double), intent(out) :: host(*)
         integer(c_int64_t), value :: count
         integer(c_int) :: rc
      end function
      function c_cfg_default(cfg) bind(C, name="nlg_exptA_config_default") result(rc)
         import c_int, nlg_exptA_config
         type(nlg_exptA_config), intent(out) :: cfg
         integer(c_int) :: rc
      end function
      function c_linop_create(mesh, cfg, baseflow, op) bind(C, name="nlg_linop_create") result(rc)
         import c_int, c_ptr, nlg_exptA_config
         type(c_ptr), value :: mesh, baseflow
         type(nlg_exptA_config), intent(in) :: cfg
         type(c_ptr), intent(out) :: op
         integer(c_int) :: rc
      end function
      function c_linop_destroy(op) bind(C, name="nlg_linop_destroy") result(rc)
         import c_int, c_ptr
         type(c_ptr), value :: op
         integer(c_int) :: rc
      end function
      function c_linop_init(op) bind(C, name="nlg_linop_init") result(rc)
         import c_int, c_ptr
         type(c_ptr), value :: op
         integer(c_int) :: rc
      end function
      function c_linop_set_tau(op, tau) bind(C, name="nlg_linop_set_tau") result(rc)
         import c_int, c_ptr, c_double
         type(c_ptr), value :: op
         real(c_double), value :: tau
         integer(c_int) :: rc
      end function
      function c_linop_matvec(op, vin, vout) bind(C, name="nlg_linop_matvec") result(rc)
         import c_int, c_ptr
         type(c_ptr), value :: op, vin, vout
         integer(c_int) :: rc
      end function
      function c_linop_rmatvec(op, vin, vout) bind(C, name="nlg_linop_rmatvec") result(rc)
         import c_int, c_ptr
         type(c_ptr), value :: op, vin, vout
         integer(c_int) :: rc
      end function
      function c_linop_nonlinear_map(op, vin, vout) bind(C, name="nlg_linop_nonlinear_map") result(rc)
         import c_int, c_ptr
         type(c_ptr), value :: op, vin, vout
         integer(c_int) :: rc
      end function
      function c_linop_set_baseflow(op, bf) bind(C, name="nlg_linop_set_baseflow") result(rc)
         import c_int, c_ptr
         type(c_ptr), value :: op, bf
         integer(c_int) :: rc
      end function
      function c_linop_set_tolerances(op, vtol, ptol) bind(C, name="nlg_linop_set_tolerances") result(rc)
         import c_int, c_ptr, c_double
         type(c_ptr), value :: op
         real(c_double), value :: vtol, ptol
         integer(c_int) :: rc
      end function
      function c_strlen(s) bind(C, name="strlen") result(n)
         import c_ptr, c_size_t
         type(c_ptr), value :: s
         integer(c_size_t) :: n
      end function
   end interface

   !----------------------------------------
   !-----     NEK REAL VECTOR TYPE     -----
   !----------------------------------------
   type, extends(abstract_vector_rdp) :: nek_dvector
      type(c_ptr) :: h = c_null_ptr
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

   !------------------------------------------
   !-----     EXPONENTIAL PROPAGATOR     -----
   !------------------------------------------
   type, extends(abstract_exptA_linop_rdp) :: exptA_linop
      type(nek_dvector) :: baseflow
      type(nlg_exptA_config) :: cfg
      type(c_ptr) :: h = c_null_ptr
      real(dp) :: tau_built = -1.0_dp
   contains
      private
      procedure, pass(self), public :: init => init_exptA
      procedure, pass(self), public :: matvec => exptA_matvec
      procedure, pass(self), public :: rmatvec => exptA_rmatvec
      ! Newton-Krylov row: what nek_system%eval (nonlinear_map) and nek_jacobian (self%X) forward to
      ! (src/systems/fixed_point.f90:4-96); the tolerance schedulers of neklab_systems.f90:229-335 call set_tolerances
      procedure, pass(self), public :: nonlinear_map => exptA_nonlinear_map
      procedure, pass(self), public :: set_baseflow => exptA_set_baseflow
      procedure, pass(self), public :: set_tolerances => exptA_set_tolerances
      final :: finalize_exptA
   end type exptA_linop

contains

   subroutine nlg_check(rc, where)
      integer(c_int), intent(in) :: rc
      character(len=*), intent(in) :: where
      type(c_ptr) :: p
      character(kind=c_char), pointer :: msg(:)
      integer :: n, i
      character(len=1024) :: txt
      if (rc == 0) return
      p = c_last_error()
      n = int(c_strlen(p))
      txt = ''
      if (n > 0) then
         call c_f_pointer(p, msg, [n])
         do i = 1, min(n, 1024)
            txt(i:i) = msg(i)
         end do
      end if
      write (*, '(A)') 'ERROR in '//trim(where)//': '//trim(txt)
      error stop 1
   end subroutine

   !> Create the device context (once per rank).
   subroutine neklab_gpu_init(device)
      integer, intent(in) :: device
      call nlg_check(c_ctx_create(int(device, c_int), nlg_ctx), 'neklab_gpu_init')
   end subroutine

   !> Upload what the reference reads from Nek5000's SIZE/TOTAL commons (xm1, ym1, zm1, glo_num, masks).
   subroutine neklab_gpu_set_mesh(ldim, lx1, nelv, xm1, ym1, zm1, glo_num, v1mask, v2mask, v3mask, has_outflow)
      integer, intent(in) :: ldim, lx1, nelv
      real(dp), target, intent(in) :: xm1(*), ym1(*), zm1(*)
      integer(c_int64_t), target, intent(in) :: glo_num(*)
      real(dp), target, intent(in) :: v1mask(*), v2mask(*), v3mask(*)
      logical, intent(in) :: has_outflow
      type(nlg_mesh_desc) :: d
      d%dim = ldim; d%n = lx1; d%lxd = 0; d%nelv = nelv
      d%xm1 = c_loc(xm1); d%ym1 = c_loc(ym1)
      d%zm1 = c_null_ptr; d%v3mask = c_null_ptr
      if (ldim == 3) then
         d%zm1 = c_loc(zm1); d%v3mask = c_loc(v3mask)
      end if
      d%glo_num = c_loc(glo_num); d%lglel = c_null_ptr
      d%v1mask = c_loc(v1mask); d%v2mask = c_loc(v2mask); d%tmask = c_null_ptr
      d%has_outflow = merge(1, 0, has_outflow)
      call nlg_check(c_mesh_create(nlg_ctx, d, nlg_mesh), 'neklab_gpu_set_mesh')
   end subroutine

   subroutine neklab_gpu_finalize()
      integer(c_int) :: rc
      if (c_associated(nlg_mesh)) rc = c_mesh_destroy(nlg_mesh)
      if (c_associated(nlg_ctx)) rc = c_ctx_destroy(nlg_ctx)
      nlg_mesh = c_null_ptr; nlg_ctx = c_null_ptr
   end subroutine

   subroutine ensure(self)
      class(nek_dvector), intent(inout) :: self
      if (.not. c_associated(self%h)) call nlg_check(c_vec_create(nlg_mesh, 0_c_int, 3_c_int, self%h), 'nek_dvector allocate')
   end subroutine

   !> nek2vec (src/neklab_utils.f90:84-108): host fields -> vector
   subroutine nek2vec_host(vec, vx, vy, vz, pr, lvn, lpn, if3d)
      type(nek_dvector), intent(inout) :: vec
      real(dp), intent(in) :: vx(*), vy(*), vz(*), pr(*)
      integer, intent(in) :: lvn, lpn
      logical, intent(in) :: if3d
      call ensure(vec)
      call nlg_check(c_vec_set_field(vec%h, 0_c_int, 0_c_int, vx, int(lvn, c_int64_t)), 'nek2vec vx')
      call nlg_check(c_vec_set_field(vec%h, 1_c_int, 0_c_int, vy, int(lvn, c_int64_t)), 'nek2vec vy')
      if (if3d) call nlg_check(c_vec_set_field(vec%h, 2_c_int, 0_c_int, vz, int(lvn, c_int64_t)), 'nek2vec vz')
      call nlg_check(c_vec_set_field(vec%h, 3_c_int, 0_c_int, pr, int(lpn, c_int64_t)), 'nek2vec pr')
   end subroutine

   !> vec2nek (src/neklab_utils.f90:110-134): vector -> host fields
   subroutine vec2nek_host(vx, vy, vz, pr, vec, lvn, lpn, if3d)
      real(dp), intent(out) :: vx(*), vy(*), vz(*), pr(*)
      type(nek_dvector), intent(in) :: vec
      integer, intent(in) :: lvn, lpn
      logical, intent(in) :: if3d
      call nlg_check(c_vec_get_field(vec%h, 0_c_int, 0_c_int, vx, int(lvn, c_int64_t)), 'vec2nek vx')
      call nlg_check(c_vec_get_field(vec%h, 1_c_int, 0_c_int, vy, int(lvn, c_int64_t)), 'vec2nek vy')
      if (if3d) call nlg_check(c_vec_get_field(vec%h, 2_c_int, 0_c_int, vz, int(lvn, c_int64_t)), 'vec2nek vz')
      call nlg_check(c_vec_get_field(vec%h, 3_c_int, 0_c_int, pr, int(lpn, c_int64_t)), 'vec2nek pr')
   end subroutine

   !-----------------------------------------
   !-----     TYPE-BOUND PROCEDURES     -----
   !-----------------------------------------
   subroutine nek_dzero(self)
      class(nek_dvector), intent(inout) :: self
      call ensure(self)
      call nlg_check(c_vec_zero(self%h), 'nek_dzero')
   end subroutine

   subroutine nek_drand(self, ifnorm)
      class(nek_dvector), intent(inout) :: self
      logical, optional, intent(in) :: ifnorm
      integer(c_int) :: nrm
      integer(c_int64_t), save :: seed = 0
      nrm = 0
      if (present(ifnorm)) nrm = merge(1, 0, ifnorm)
      call ensure(self)
      seed = seed + 1     ! successive calls draw different fields, like random_number in the reference
      call nlg_check(c_vec_rand(self%h, nrm, seed), 'nek_drand')
   end subroutine

   subroutine nek_dscal(self, alpha)
      class(nek_dvector), intent(inout) :: self
      real(dp), intent(in) :: alpha
      call ensure(self)
      call nlg_check(c_vec_scal(self%h, alpha), 'nek_dscal')
   end subroutine

   subroutine nek_daxpby(alpha, vec, beta, self)
      class(nek_dvector), intent(inout) :: self
      real(dp), intent(in) :: alpha
      class(abstract_vector_rdp), intent(in) :: vec
      real(dp), intent(in) :: beta
      call ensure(self)
      select type (vec)
      type is (nek_dvector)
         call nlg_check(c_vec_axpby(alpha, vec%h, beta, self%h), 'nek_daxpby')
      class default
         write (*, '(A)') "type_error: 'vec' must be nek_dvector in nek_daxpby"   ! real_vectors.f90:202-204
         error stop 1
      end select
   end subroutine

   function nek_ddot(self, vec) result(alpha)
      class(nek_dvector), intent(in) :: self
      class(abstract_vector_rdp), intent(in) :: vec
      real(dp) :: alpha
      alpha = 0.0_dp
      select type (vec)
      type is (nek_dvector)
         call nlg_check(c_vec_dot(self%h, vec%h, alpha), 'nek_ddot')
      class default
         write (*, '(A)') "type_error: 'vec' must be nek_dvector in nek_ddot"     ! real_vectors.f90:229-231
         error stop 1
      end select
   end function

   function nek_dsize(self) result(n)
      class(nek_dvector), intent(in) :: self
      integer :: n
      integer(c_int64_t) :: n8
      call nlg_check(c_vec_size(self%h, n8), 'nek_dsize')
      n = int(n8)
   end function

   subroutine dsave_rst(self, vec_rst, irst)
      class(nek_dvector), intent(inout) :: self
      class(abstract_vector_rdp), intent(in) :: vec_rst
      integer, intent(in) :: irst
      select type (vec_rst)
      type is (nek_dvector)
         call nlg_check(c_vec_save_rst(self%h, vec_rst%h, int(irst, c_int)), 'dsave_rst')
      end select
   end subroutine

   subroutine dget_rst(self, vec_rst, irst)
      class(nek_dvector), intent(in) :: self
      class(abstract_vector_rdp), intent(inout) :: vec_rst
      integer, intent(in) :: irst
      select type (vec_rst)
      type is (nek_dvector)
         call ensure(vec_rst)
         call nlg_check(c_vec_get_rst(self%h, vec_rst%h, int(irst, c_int)), 'dget_rst')
      end select
   end subroutine

   function dhas_rst_fields(self) result(has_rst_fields)
      class(nek_dvector), intent(in) :: self
      logical :: has_rst_fields
      integer(c_int) :: flag
      call nlg_check(c_vec_has_rst(self%h, flag), 'dhas_rst_fields')
      has_rst_fields = flag /= 0
   end function

   subroutine dclear_rst_fields(self)
      class(nek_dvector), intent(inout) :: self
      call nlg_check(c_vec_clear_rst(self%h), 'dclear_rst_fields')
   end subroutine

   !> intrinsic-assignment semantics of the reference's by-value vectors: deep copy
   subroutine assign_dvector(lhs, rhs)
      class(nek_dvector), intent(inout) :: lhs
      class(nek_dvector), intent(in) :: rhs
      if (.not. c_associated(rhs%h)) return
      if (c_associated(lhs%h, rhs%h)) return
      if (c_associated(lhs%h)) then
         call nlg_check(c_vec_copy(lhs%h, rhs%h), 'nek_dvector assignment')
      else
         call nlg_check(c_vec_clone(rhs%h, lhs%h), 'nek_dvector assignment')
      end if
   end subroutine

   subroutine finalize_dvector(self)
      type(nek_dvector), intent(inout) :: self
      integer(c_int) :: rc
      if (c_associated(self%h)) rc = c_vec_destroy(self%h)
      self%h = c_null_ptr
   end subroutine

   subroutine finalize_dvector_rank1(self)
      type(nek_dvector), intent(inout) :: self(:)
      integer :: i
      integer(c_int) :: rc
      do i = 1, size(self)
         if (c_associated(self(i)%h)) rc = c_vec_destroy(self(i)%h)
         self(i)%h = c_null_ptr
      end do
   end subroutine

   !---- exptA_linop ------------------------------------------------------------------------------
   subroutine init_exptA(self)
      class(exptA_linop), intent(inout) :: self
      integer(c_int) :: rc
      if (c_associated(self%h)) rc = c_linop_destroy(self%h)
      if (self%cfg%torder == 0) call nlg_check(c_cfg_default(self%cfg), 'init_exptA')
      self%cfg%tau = self%tau
      call nlg_check(c_linop_create(nlg_mesh, self%cfg, self%baseflow%h, self%h), 'init_exptA')
      call nlg_check(c_linop_init(self%h), 'init_exptA')
      self%tau_built = self%tau
   end subroutine

   subroutine exptA_matvec(self, vec_in, vec_out)
      class(exptA_linop), intent(inout) :: self
      class(abstract_vector_rdp), intent(in) :: vec_in
      class(abstract_vector_rdp), intent(out) :: vec_out
      if (self%tau /= self%tau_built) then       ! apply_exptA sets A%tau before the call (neklab_linops.f90:252)
         call nlg_check(c_linop_set_tau(self%h, self%tau), 'exptA_matvec'); self%tau_built = self%tau
      end if
      select type (vec_in)
      type is (nek_dvector)
         select type (vec_out)
         type is (nek_dvector)
            call ensure(vec_out)
            call nlg_check(c_linop_matvec(self%h, vec_in%h, vec_out%h), 'exptA_matvec')
         class default
            write (*, '(A)') "type_error: 'vec_out' must be nek_dvector in exptA_matvec"; error stop 1
         end select
      class default
         write (*, '(A)') "type_error: 'vec_in' must be nek_dvector in exptA_matvec"; error stop 1
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
            call ensure(vec_out)
            call nlg_check(c_linop_nonlinear_map(self%h, vec_in%h, vec_out%h), 'nonlinear_map')
         class default
            write (*, '(A)') "type_error: 'vec_out' must be nek_dvector in nonlinear_map"; error stop 1
         end select
      class default
         write (*, '(A)') "type_error: 'vec_in' must be nek_dvector in nonlinear_map"; error stop 1
      end select
   end subroutine

   subroutine exptA_set_baseflow(self, X)
      class(exptA_linop), intent(inout) :: self
      class(abstract_vector_rdp), intent(in) :: X
      select type (X)
      type is (nek_dvector)
         call nlg_check(c_linop_set_baseflow(self%h, X%h), 'set_baseflow')
      class default
         write (*, '(A)') "type_error: 'X' must be nek_dvector in set_baseflow"; error stop 1
      end select
   end subroutine

   subroutine exptA_set_tolerances(self, vtol, ptol)
      class(exptA_linop), intent(inout) :: self
      real(dp), intent(in) :: vtol, ptol
      call nlg_check(c_linop_set_tolerances(self%h, vtol, ptol), 'set_tolerances')
   end subroutine

   subroutine exptA_rmatvec(self, vec_in, vec_out)
      class(exptA_linop), intent(inout) :: self
      class(abstract_vector_rdp), intent(in) :: vec_in
      class(abstract_vector_rdp), intent(out) :: vec_out
      if (self%tau /= self%tau_built) then
         call nlg_check(c_linop_set_tau(self%h, self%tau), 'exptA_rmatvec'); self%tau_built = self%tau
      end if
      select type (vec_in)
      type is (nek_dvector)
         select type (vec_out)
         type is (nek_dvector)
            call ensure(vec_out)
            call nlg_check(c_linop_rmatvec(self%h, vec_in%h, vec_out%h), 'exptA_rmatvec')
         class default
            write (*, '(A)') "type_error: 'vec_out' must be nek_dvector in exptA_rmatvec"; error stop 1
         end select
      class default
         write (*, '(A)') "type_error: 'vec_in' must be nek_dvector in exptA_rmatvec"; error stop 1
      end select
   end subroutine

   subroutine finalize_exptA(self)
      type(exptA_linop), intent(inout) :: self
      integer(c_int) :: rc
      if (c_associated(self%h)) rc = c_linop_destroy(self%h)
      self%h = c_null_ptr
   end subroutine

end module neklab_gpu
