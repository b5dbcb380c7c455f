!> ISO_C_BINDING view of include/neklab_gpu.h: one bind(C) interface per entry point the Fortran shim forwards to,
!! the two interoperable structs, the process-wide handles and the case parameters.
!!
!! The reference keeps the corresponding state in Nek5000 commons (SIZE / TOTAL / INPUT): lx1, nelv, the coordinates and
!! masks, `param(2)` (viscosity), `param(21/22)` (tolerances), `param(27)` (time order), `ifheat`, `lorder`, `ldimt`
!! (e.g. /root/reference/src/vectors/neklab_vectors.f90:8-14, src/neklab_nek_setup.f90:227-230,
!! src/linops/exponential_propagator.f90:23).  A Nek5000 host calls neklab_gpu_init / neklab_gpu_set_mesh /
!! neklab_gpu_set_case once from `usrdat3` or at the top of `userchk` (INTEGRATION.md).
module neklab_gpu_capi
   use iso_c_binding
   use iso_fortran_env, only: real64
   implicit none
   private :: real64
   integer, parameter, private :: dp = real64

   !> process-wide handles
   type(c_ptr), save, public :: nlg_ctx = c_null_ptr
   type(c_ptr), save, public :: nlg_mesh = c_null_ptr

   type, bind(C), public :: nlg_mesh_desc
      integer(c_int) :: dim, n, lxd
      integer(c_int64_t) :: nelv
      type(c_ptr) :: xm1, ym1, zm1, glo_num, lglel, v1mask, v2mask, v3mask, tmask
      integer(c_int) :: has_outflow
   end type

   !> nlg_exptA_config, every component default-initialised to what nlg_exptA_config_default returns
   type, bind(C), public :: nlg_exptA_config
      real(c_double) :: tau = 1.0_c_double, re = 100.0_c_double, cfl_limit = 0.5_c_double
      real(c_double) :: vtol = 1.0e-9_c_double, ptol = 1.0e-7_c_double, dt = 0.0_c_double
      integer(c_int) :: torder = 3, maxit_v = 200, maxit_p = 2000, fixed_iters_v = 0, fixed_iters_p = 0, pprecond = 0, pproj = 1
      integer(c_int) :: ifheat = 0
      real(c_double) :: conductivity = 1.0_c_double, rhocp = 1.0_c_double, buoy(3) = 0.0_c_double
      integer(c_int) :: no_history = 0
   end type

   type, bind(C), public :: nlg_eigs_opts
      integer(c_int) :: kdim = 0, transpose = 0, max_restarts = 50, write_intermediate = 1
      real(c_double) :: tol = 0.0_c_double
      type(c_ptr) :: logfile = c_null_ptr
      integer(c_int64_t) :: seed = 0
      integer(c_int) :: block_size = 0, warm_start = 0
   end type

   !> what the reference reads from Nek5000's `param(.)` / logical flags: the template every exptA_linop starts from,
   !! the number of active scalars (ifto / ifpsco) and `lorder` of SIZE for every nek_dvector
   type(nlg_exptA_config), save, public :: nek_case
   integer, save, public :: nek_nscal = 0, nek_lorder = 3
   integer(c_int64_t), save, public :: nek_lvn = 0, nek_lpn = 0
   integer, save, public :: nek_ldim = 0, nek_lx1 = 0
   !> Nek5000's endTime (param(10)): the horizon the nonlinear map of the Newton systems integrates over (fixed_point.f90:14-24)
   real(c_double), save, public :: nek_endtime = 1.0_c_double

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
      function c_vec_generation(v, gen) bind(C, name="nlg_vec_generation") result(rc)
         import c_int, c_ptr, c_int64_t
         type(c_ptr), value :: v
         integer(c_int64_t), intent(out) :: gen
         integer(c_int) :: rc
      end function
      function c_vec_release(v) bind(C, name="nlg_vec_release") result(rc)
         import c_int, c_ptr
         type(c_ptr), value :: v
         integer(c_int) :: rc
      end function
      function c_vec_adopt(v, gen, status) bind(C, name="nlg_vec_adopt") result(rc)
         import c_int, c_ptr, c_int64_t
         type(c_ptr), value :: v
         integer(c_int64_t), value :: gen
         integer(c_int), intent(out) :: status
         integer(c_int) :: rc
      end function
      function c_vec_pin(v, gen) bind(C, name="nlg_vec_pin") result(rc)
         import c_int, c_ptr, c_int64_t
         type(c_ptr), value :: v
         integer(c_int64_t), value :: gen
         integer(c_int) :: rc
      end function
      function c_vec_unpin(v, gen) bind(C, name="nlg_vec_unpin") result(rc)
         import c_int, c_ptr, c_int64_t
         type(c_ptr), value :: v
         integer(c_int64_t), value :: gen
         integer(c_int) :: rc
      end function
      pure function c_vec_has_rst_checked(v, gen) bind(C, name="nlg_vec_has_rst_checked") result(flag)
         import c_int, c_ptr, c_int64_t
         type(c_ptr), value :: v
         integer(c_int64_t), value :: gen
         integer(c_int) :: flag
      end function
      pure function c_vec_size_checked(v, gen) bind(C, name="nlg_vec_size_checked") result(n)
         import c_ptr, c_int64_t
         type(c_ptr), value :: v
         integer(c_int64_t), value :: gen
         integer(c_int64_t) :: n
      end function
      function c_vec_pool_trim(freed) bind(C, name="nlg_vec_pool_trim") result(rc)
         import c_int, c_ptr
         type(c_ptr), value :: freed
         integer(c_int) :: rc
      end function
      pure function c_vec_has_rst_value(v) bind(C, name="nlg_vec_has_rst_value") result(flag)
         import c_int, c_ptr
         type(c_ptr), value :: v
         integer(c_int) :: flag
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
         pure function c_vec_size_value(v) bind(C, name="nlg_vec_size_value") result(n)
         import c_ptr, c_int64_t
         type(c_ptr), value :: v
         integer(c_int64_t) :: n
      end function
      function c_vec_nrst(v, n) bind(C, name="nlg_vec_nrst") result(rc)
         import c_int, c_ptr
         type(c_ptr), value :: v
         integer(c_int), intent(out) :: n
         integer(c_int) :: rc
      end function
      function c_vec_outpost(v, path, with_coords, time, istep) bind(C, name="nlg_vec_outpost") result(rc)
         import c_int, c_ptr, c_double, c_char
         type(c_ptr), value :: v
         character(kind=c_char), intent(in) :: path(*)
         integer(c_int), value :: with_coords, istep
         real(c_double), value :: time
         integer(c_int) :: rc
      end function
      function c_mesh_sizes(mesh, lvn, lpn, dim, n) bind(C, name="nlg_mesh_sizes") result(rc)
         import c_int, c_ptr, c_int64_t
         type(c_ptr), value :: mesh
         integer(c_int64_t), intent(out) :: lvn, lpn
         integer(c_int), intent(out) :: dim, n
         integer(c_int) :: rc
      end function
      function c_linop_get_info(op, tau, dt, nsteps, cfl) bind(C, name="nlg_linop_get_info") result(rc)
         import c_int, c_ptr, c_double
         type(c_ptr), value :: op
         real(c_double), intent(out) :: tau, dt, cfl
         integer(c_int), intent(out) :: nsteps
         integer(c_int) :: rc
      end function
      function c_linop_set_projection(op, alpha, idir, lab, lab2, x2) bind(C, name="nlg_linop_set_projection") result(rc)
         import c_int, c_ptr, c_double
         type(c_ptr), value :: op, lab, lab2, x2
         real(c_double), value :: alpha
         integer(c_int), value :: idir
         integer(c_int) :: rc
      end function
      function c_linop_project(op, v) bind(C, name="nlg_linop_project") result(rc)
         import c_int, c_ptr
         type(c_ptr), value :: op, v
         integer(c_int) :: rc
      end function
      function c_linop_integrate_forced(op, ic, f_re, f_im, omega, adjoint, vout) bind(C, name="nlg_linop_integrate_forced") result(rc)
         import c_int, c_ptr, c_double
         type(c_ptr), value :: op, ic, f_re, f_im, vout
         real(c_double), value :: omega
         integer(c_int), value :: adjoint
         integer(c_int) :: rc
      end function
      function c_eigs_opts_default(o) bind(C, name="nlg_eigs_opts_default") result(rc)
         import c_int, nlg_eigs_opts
         type(nlg_eigs_opts), intent(out) :: o
         integer(c_int) :: rc
      end function
      function c_eigs(op, X, nev, eig_re, eig_im, residuals, info, x0, opts) bind(C, name="nlg_eigs") result(rc)
         import c_int, c_ptr, c_double, nlg_eigs_opts
         type(c_ptr), value :: op, x0
         type(c_ptr), intent(in) :: X(*)
         integer(c_int), value :: nev
         real(c_double), intent(out) :: eig_re(*), eig_im(*), residuals(*)
         integer(c_int), intent(out) :: info
         type(nlg_eigs_opts), intent(in) :: opts
         integer(c_int) :: rc
      end function
      function c_svds(op, U, V, nsv, S, residuals, info, u0, opts) bind(C, name="nlg_svds") result(rc)
         import c_int, c_ptr, c_double, nlg_eigs_opts
         type(c_ptr), value :: op, u0
         type(c_ptr), intent(in) :: U(*), V(*)
         integer(c_int), value :: nsv
         real(c_double), intent(out) :: S(*), residuals(*)
         integer(c_int), intent(out) :: info
         type(nlg_eigs_opts), intent(in) :: opts
         integer(c_int) :: rc
      end function
   end interface

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
   subroutine neklab_gpu_set_mesh(ldim, lx1, nelv, xm1, ym1, zm1, glo_num, v1mask, v2mask, v3mask, has_outflow, lxd, tmask)
      integer, intent(in) :: ldim, lx1, nelv
      real(dp), target, intent(in) :: xm1(*), ym1(*), zm1(*)
      integer(c_int64_t), target, intent(in) :: glo_num(*)
      real(dp), target, intent(in) :: v1mask(*), v2mask(*), v3mask(*)
      logical, intent(in) :: has_outflow
      integer, optional, intent(in) :: lxd
      real(dp), target, optional, intent(in) :: tmask(*)
      type(nlg_mesh_desc) :: d
      integer(c_int) :: cdim, cn
      d%dim = ldim; d%n = lx1; d%lxd = 0; d%nelv = nelv
      if (present(lxd)) d%lxd = lxd
      d%xm1 = c_loc(xm1); d%ym1 = c_loc(ym1)
      d%zm1 = c_null_ptr; d%v3mask = c_null_ptr
      if (ldim == 3) then
         d%zm1 = c_loc(zm1); d%v3mask = c_loc(v3mask)
      end if
      d%glo_num = c_loc(glo_num); d%lglel = c_null_ptr
      d%v1mask = c_loc(v1mask); d%v2mask = c_loc(v2mask); d%tmask = c_null_ptr
      if (present(tmask)) d%tmask = c_loc(tmask)
      d%has_outflow = merge(1, 0, has_outflow)
      call nlg_check(c_mesh_create(nlg_ctx, d, nlg_mesh), 'neklab_gpu_set_mesh')
      call nlg_check(c_mesh_sizes(nlg_mesh, nek_lvn, nek_lpn, cdim, cn), 'neklab_gpu_set_mesh')
      nek_ldim = cdim; nek_lx1 = cn
   end subroutine

   !> The case parameters the reference takes from the .par file / SIZE through Nek5000's commons: viscosity (as
   !! Reynolds number), |param(27)| = time order, param(21) / param(22) = pressure / velocity tolerances, ifheat with
   !! its properties, the number of active scalars and `lorder`.  Omitted arguments keep their current value.
   subroutine neklab_gpu_set_case(re, torder, ptol, vtol, dt, cfl_limit, ifheat, conductivity, rhocp, buoy, nscal, lorder, &
                                  maxit_v, maxit_p, pprecond, pproj, endtime)
      real(dp), optional, intent(in) :: re, ptol, vtol, dt, cfl_limit, conductivity, rhocp, buoy(3), endtime
      integer, optional, intent(in) :: torder, nscal, lorder, maxit_v, maxit_p, pprecond, pproj
      logical, optional, intent(in) :: ifheat
      if (present(re)) nek_case%re = re
      if (present(endtime)) nek_endtime = endtime
      if (present(torder)) nek_case%torder = torder
      if (present(ptol)) nek_case%ptol = ptol
      if (present(vtol)) nek_case%vtol = vtol
      if (present(dt)) nek_case%dt = dt
      if (present(cfl_limit)) nek_case%cfl_limit = cfl_limit
      if (present(ifheat)) nek_case%ifheat = merge(1, 0, ifheat)
      if (present(conductivity)) nek_case%conductivity = conductivity
      if (present(rhocp)) nek_case%rhocp = rhocp
      if (present(buoy)) nek_case%buoy = buoy
      if (present(maxit_v)) nek_case%maxit_v = maxit_v
      if (present(maxit_p)) nek_case%maxit_p = maxit_p
      if (present(pprecond)) nek_case%pprecond = pprecond
      if (present(pproj)) nek_case%pproj = pproj
      if (present(nscal)) nek_nscal = nscal
      if (present(lorder)) nek_lorder = lorder
      if (nek_case%ifheat /= 0 .and. nek_nscal < 1) nek_nscal = 1      ! ifto: the vectors carry the temperature
      if (nek_lorder < nek_case%torder) nek_lorder = nek_case%torder
   end subroutine

   subroutine neklab_gpu_finalize()
      integer(c_int) :: rc
      rc = c_vec_pool_trim(c_null_ptr)      ! vectors released by finalisers and never adopted
      if (c_associated(nlg_mesh)) rc = c_mesh_destroy(nlg_mesh)
      if (c_associated(nlg_ctx)) rc = c_ctx_destroy(nlg_ctx)
      nlg_mesh = c_null_ptr; nlg_ctx = c_null_ptr
   end subroutine

end module neklab_gpu_capi
