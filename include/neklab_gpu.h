/*
 * neklab_gpu.h -- C ABI of the MI355X-native hot path of nekStab/neklab.
 *
 * Drop-in boundary: the reference consumes this path through Fortran 2008 type-bound
 * procedures of LightKrylov's abstract_vector_rdp / abstract_exptA_linop_rdp.  A Fortran shim
 * (neklab_amd/fortran/neklab_vectors.f90, neklab_linops.f90, neklab_utils.f90; bind(C) interfaces in neklab_gpu_capi.f90) extends
 * those abstract types under the reference's module and type names and forwards every binding to one
 * entry point of this header through ISO_C_BINDING.  Each declaration cites the reference interface
 * (file:line under /root/reference) it replaces.
 *
 * Conventions
 *   - every function returns int: 0 = ok, non-zero = error (text in nlg_last_error()).  The reference
 *     has no status codes on the vector API and aborts through type_error / stop_error
 *     (src/vectors/real_vectors.f90:202-204, src/neklab_nek_setup.f90:406-417); the shim turns a
 *     non-zero return into stop_error.
 *   - handles are opaque; all field data lives in HBM and is owned by the library.
 *   - host arrays handed in are plain double / int64_t arrays in Nek5000's element-major layout
 *     ijke = ix + n*(iy + n*(iz + n*e))   (src/vectors/real_vectors.f90:69).
 *   - single-threaded per rank, every call collective across ranks (SURVEY.md §8b "Threading").
 *   - there is NO CPU fallback: every entry point fails with an error if no HIP device is usable.
 */
#ifndef NEKLAB_GPU_H
#define NEKLAB_GPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct nlg_ctx nlg_ctx;     /* device + stream + communicator                                   */
typedef struct nlg_mesh nlg_mesh;   /* SEM discretisation: what Nek5000 keeps in SIZE/TOTAL commons     */
typedef struct nlg_vec nlg_vec;     /* nek_dvector  (src/vectors/neklab_vectors.f90:26-50)               */
typedef struct nlg_basis nlg_basis; /* array of nek_dvector as LightKrylov allocates for the Krylov basis */
typedef struct nlg_linop nlg_linop; /* exptA_linop  (src/linops/neklab_linops.f90:35-44)                 */

/* field selectors for nlg_vec_set_field / nlg_vec_get_field */
enum { NLG_VX = 0, NLG_VY = 1, NLG_VZ = 2, NLG_PR = 3, NLG_THETA = 4 /* + scalar index */ };

/* ---------------------------------------------------------------------------------------------- */
/* context                                                                                          */
/* ---------------------------------------------------------------------------------------------- */
const char *nlg_last_error(void);
int nlg_version(void);
int nlg_ctx_create(int device, nlg_ctx **out);
int nlg_ctx_destroy(nlg_ctx *ctx);
int nlg_ctx_sync(nlg_ctx *ctx);
/* RCCL communicator over xGMI: rank 0 calls nlg_comm_unique_id, the 128 bytes are broadcast by the
 * host program (MPI_Bcast in a Nek5000 host, torch.distributed in bench.py), every rank then calls
 * nlg_ctx_comm_init.  Replaces the MPI_Allreduce inside glsc3 (src/vectors/real_vectors.f90:217-224) and gslib's gs_op
 * behind opdssum (real_vectors.f90:100-104): per gather-scatter the library sums the groups that hold a dof another rank
 * shares, packs, exchanges (grouped ncclSend / ncclRecv), sums the interior groups and unpacks; with NLG_HALO_OVERLAP=1 in
 * the environment the exchange runs on a side stream beside the interior groups. */
int nlg_comm_unique_id(void *out128);
int nlg_ctx_comm_init(nlg_ctx *ctx, int rank, int nranks, const void *unique_id128);
int nlg_ctx_rank(const nlg_ctx *ctx, int *rank, int *nranks);
/* VALIDATION transport, not a product path: the same four collectives staged through a POSIX shared-memory
 * segment `name` (slot_bytes per rank), so that several ranks can share ONE GPU — which RCCL refuses — and the
 * whole distributed path except the RCCL calls themselves can be tested on a one-GPU box
 * (tests/test_gpu_multirank.py).  Every wait times out with an error after 120 s. */
int nlg_ctx_comm_init_shm(nlg_ctx *ctx, int rank, int nranks, const char *name, int64_t slot_bytes);
/* Host-side planning step of the multi-rank gather-scatter (gslib's gs_setup behind opdssum,
 * src/vectors/real_vectors.f90:100): given the ascending unique labels of every rank, concatenated rank after
 * rank (counts[q] each), return for `rank` the labels it shares with every other rank (neigh_counts[q], and the
 * labels themselves in shared_out, neighbour after neighbour, ascending).  Pure host code, no GPU needed.
 * Returns the total number of shared entries, or -1 if `capacity` is too small. */
int64_t nlg_halo_plan(int rank, int nranks, const int64_t *counts, const int64_t *labels_concat,
                      int64_t *neigh_counts, int64_t *shared_out, int64_t capacity);
/* The two remaining pure-host steps of that set-up, exported so that the index lists the GPU exchange uses can be
 * checked on the CPU with emulated ranks (tests/test_cpu_dist.py): the ascending unique labels of the element-boundary
 * dofs of one rank (its contribution to labels_concat; returns their number, -1 if capacity is too small), and the
 * lists themselves -- send_idx[e] = local dof packed for entry e of the send buffer (entries ordered neighbour after
 * neighbour, labels ascending), and per shared label l < *nlab_out the positions rpos[roff[l]:roff[l+1]] of the received
 * contributions and the local copies cidx[coff[l]:coff[l+1]] that receive their sum.  Returns the number of entries. */
int64_t nlg_halo_boundary_labels(int n, int dim, int64_t E, const int64_t *glo, int64_t *labels_out, int64_t capacity);
int64_t nlg_halo_lists(int n, int dim, int64_t E, const int64_t *glo, int rank, int nranks, const int64_t *counts,
                       const int64_t *labels_concat, int64_t *neigh_counts, int32_t *send_idx, int64_t cap_send,
                       int32_t *roff, int32_t *rpos, int32_t *coff, int32_t *cidx, int64_t cap_copies, int64_t *nlab_out);
/* Per-kernel-class timing with HIP events recorded on the launch stream (the reference's counterpart
 * is LightKrylov's timer object, src/neklab_analysis.f90:66-67, :98-101).  Classes: "axhelm", "gs",
 * "opgradt", "opdiv", "colmul", "block_dot", "block_axpy", "cg_vec", "conv", "vec_ops". */
int nlg_prof_enable(nlg_ctx *ctx, int class_mask); /* bit i = class i in the order above; -1 = all; 0 = off */
/* time only every stride-th launch of an enabled class (default 1): a pair of events around a 50-us kernel costs ~12 us of
 * stream time, so timing every launch of the dominant class would slow the run it measures by 2 %; nlg_prof_get then returns
 * the number of TIMED launches and their total */
int nlg_prof_sample(nlg_ctx *ctx, int stride);
int nlg_prof_reset(nlg_ctx *ctx);
int nlg_prof_get(nlg_ctx *ctx, const char *name, int64_t *count, double *total_ms);
/* Running totals since the library was loaded: kernel launches issued, and collective sites passed (all-reduce, all-gather and
 * gather-scatter / Schwarz halo exchanges -- counted on one rank too, where they move nothing: what a partitioned run of the same
 * calls issues).  bench.py reports the difference over its timed region per step.  The reference's counterpart: every glsc3 /
 * gs_op / nekgsync inside nek_advance, which it does not count. */
int nlg_counters(int64_t *launches, int64_t *collectives);

/* ---------------------------------------------------------------------------------------------- */
/* mesh: replaces the Nek5000 commons the reference reads through include "SIZE"/"TOTAL"            */
/* (src/vectors/neklab_vectors.f90:8-14: lx1, lelv, lv, lp; real_vectors.f90: xm1, ym1, zm1, bm1,   */
/*  vmult, v1mask..v3mask, lglel; gather-scatter handle behind opdssum/dsavg :100-104)              */
/* ---------------------------------------------------------------------------------------------- */
typedef struct nlg_mesh_desc {
    int dim;                /* ldim: 2 or 3                                           */
    int n;                  /* lx1 (= ly1 = lz1 in 3-D); pressure mesh is lx2 = n - 2 */
    int lxd;                /* dealiasing points, 0 => 3*n/2                          */
    int64_t nelv;           /* local element count                                    */
    const double *xm1;      /* [nelv * n^dim] GLL coordinates                         */
    const double *ym1;
    const double *zm1;      /* NULL in 2-D                                            */
    const int64_t *glo_num; /* [nelv * n^dim] global (assembled) dof labels           */
    const int64_t *lglel;   /* [nelv] global element ids (0-based), NULL => 0..nelv-1 */
    const double *v1mask;   /* [nelv * n^dim] Dirichlet masks, 0/1                    */
    const double *v2mask;
    const double *v3mask;   /* NULL in 2-D                                            */
    const double *tmask;    /* NULL => no scalar mask                                 */
    int has_outflow;        /* 0 => pressure defined up to a constant (Nek ifvcor)    */
} nlg_mesh_desc;

int nlg_mesh_create(nlg_ctx *ctx, const nlg_mesh_desc *desc, nlg_mesh **out);
int nlg_mesh_destroy(nlg_mesh *mesh);
int nlg_mesh_sizes(const nlg_mesh *mesh, int64_t *lvn, int64_t *lpn, int *dim, int *n);
/* read back a derived array by name for verification: "bm1", "binvm1", "vmult", "jac", "bm2",
 * "g11".."g33" (Nek g1m1..g6m1 ordering by index pair), "rxm1"... (as "rst11".."rst33") */
int nlg_mesh_get(const nlg_mesh *mesh, const char *name, double *out, int64_t count);

/* ---------------------------------------------------------------------------------------------- */
/* nek_dvector                                                                                      */
/* ---------------------------------------------------------------------------------------------- */
/* type(nek_dvector) storage, src/vectors/neklab_vectors.f90:26-36; nscal = active scalars (ifto /
 * ifpsco), lorder as in SIZE (restart history holds lorder-1 slots). */
int nlg_vec_create(nlg_mesh *mesh, int nscal, int lorder, nlg_vec **out);
int nlg_vec_destroy(nlg_vec *v);
/* Handle lifetimes for host languages whose objects are copied bit by bit behind the type's back (the reference's vectors are
 * plain static arrays: src/vectors/neklab_vectors.f90:26-36; Fortran intrinsic assignment / sourced allocation / reallocation on
 * assignment duplicate a shim object together with its handle).  The owner's finaliser RELEASES instead of destroying; a copy
 * that is used later ADOPTS: *status = 1 -- the handle had been released and now belongs to the caller (nothing copied, nothing
 * leaked); *status = 0 -- it is still owned by a live object: the caller clones.  `gen` is the generation number the caller read
 * when it last owned or copied the handle (nlg_vec_generation): a handle that was freed in between is an error, not a crash.
 * Released buffers are freed oldest first above nlg_vec_pool_limit bytes (default 32 GiB), when an allocation fails, and by
 * nlg_vec_pool_trim.  nlg_vec_destroy frees at once, as before (hosts with reference semantics: the Python mirror). */
int nlg_vec_generation(const nlg_vec *v, int64_t *gen);
int nlg_vec_release(nlg_vec *v);
int nlg_vec_adopt(nlg_vec *v, int64_t gen, int *status);
/* A copy that only READS (an intent(in) use cannot take ownership) pins a released handle: it leaves the eviction pool and stays
 * adoptable; the finaliser of a non-owning copy un-pins (back into the pool).  Both validate `gen` like nlg_vec_adopt. */
int nlg_vec_pin(nlg_vec *v, int64_t gen);
int nlg_vec_unpin(nlg_vec *v, int64_t gen);
int nlg_vec_pool_limit(int64_t bytes);
int nlg_vec_pool_trim(int64_t *freed_bytes);
/* Fortran intrinsic assignment / sourced allocation of a nek_dvector */
int nlg_vec_clone(const nlg_vec *src, nlg_vec **out);
int nlg_vec_copy(nlg_vec *dst, const nlg_vec *src);
/* nek_dzero   src/vectors/real_vectors.f90:37-50   (interface neklab_vectors.f90:65-67) */
int nlg_vec_zero(nlg_vec *self);
/* nek_drand   real_vectors.f90:52-123   (interface :69-72); seed replaces the compiler RNG */
int nlg_vec_rand(nlg_vec *self, int ifnorm, uint64_t seed);
/* The two halves of nek_drand, exported so that each can be compared with the oracle on its own: the mth_rand noise
 * (neklab_vectors.f90:305-314) added to every active field, real_vectors.f90:62-98 -- and the part that makes the
 * field admissible: opdssum * vmult, dsavg, bcdirvc / bcdirsc, optional normalisation, nrst = 0 (:100-122).
 * nlg_vec_rand = noise + finish. */
int nlg_vec_rand_noise(nlg_vec *self, uint64_t seed);
int nlg_vec_rand_finish(nlg_vec *self, int ifnorm);
/* nek_dscal   real_vectors.f90:125-160  (interface :74-77) */
int nlg_vec_scal(nlg_vec *self, double alpha);
/* nek_daxpby  real_vectors.f90:162-206  (interface :79-84): self = alpha*vec + beta*self */
int nlg_vec_axpby(double alpha, const nlg_vec *vec, double beta, nlg_vec *self);
/* nek_ddot    real_vectors.f90:208-233  (interface :86-89): globally reduced, same on all ranks */
int nlg_vec_dot(const nlg_vec *self, const nlg_vec *vec, double *out);
/* %norm() inherited from abstract_vector_rdp, used at real_vectors.f90:117 */
int nlg_vec_norm(const nlg_vec *self, double *out);
/* nek_dsize   real_vectors.f90:235-247  (interface :91-93) */
int nlg_vec_size(const nlg_vec *self, int64_t *out);
/* the same as a plain function value (-1: NULL handle): the reference declares nek_dsize `pure`
 * (neklab_vectors.f90:91-93), and a pure Fortran function can only call interfaces that are themselves pure */
int64_t nlg_vec_size_value(const nlg_vec *self);
/* dhas_rst_fields as a plain function value for the same reason: the reference declares it `pure` (neklab_vectors.f90:107-110) */
int nlg_vec_has_rst_value(const nlg_vec *self);
/* the two plain-value forms for handles that may be stale (the Fortran shim's bitwise copies): -1 when the handle is unknown or
 * its generation is not `gen`; they never dereference a freed handle */
int64_t nlg_vec_size_checked(const nlg_vec *self, int64_t gen);
int nlg_vec_has_rst_checked(const nlg_vec *self, int64_t gen);
/* outpost_dnek  src/neklab_utils.f90:305-333 (Nek5000 `outpost(vx, vy, vz, pr, t, prefix)`): one vector -> one Nek5000
 * field file at `path` ("#std" header, fp64): GLL coordinates when with_coords != 0 (Nek5000 writes them into the first
 * file of a series only), velocity, the pressure mapped to the velocity mesh (Nek5000's `mappr` for a Pn-Pn-2 run), the
 * first scalar if the vector carries one.  Single rank. */
int nlg_vec_outpost(const nlg_vec *self, const char *path, int with_coords, double time, int istep);
/* dsave_rst / dget_rst / dhas_rst_fields / dclear_rst_fields  real_vectors.f90:249-346 (irst 1-based) */
int nlg_vec_save_rst(nlg_vec *self, const nlg_vec *vec_rst, int irst);
int nlg_vec_get_rst(const nlg_vec *self, nlg_vec *vec_rst, int irst);
int nlg_vec_has_rst_fields(const nlg_vec *self, int *out);
int nlg_vec_clear_rst_fields(nlg_vec *self);
int nlg_vec_nrst(const nlg_vec *self, int *out);
/* nek2vec / vec2nek  src/neklab_utils.f90:84-134 (nopcopy :279-301): move one field between a host
 * array and the vector.  field = NLG_VX.., irst = 0 main field, 1..lorder-1 history slot. */
int nlg_vec_set_field(nlg_vec *self, int field, int irst, const double *host, int64_t count);
int nlg_vec_get_field(const nlg_vec *self, int field, int irst, double *host, int64_t count);
/* Restart history in axpby / block updates.
 * 1 (default): history slot r of self receives alpha * (history slot r of vec).
 * 0           : literal reading of real_vectors.f90:188-192: every slot receives alpha * vec's MAIN field.
 * The literal reading does not reproduce the reference's own known answer (cylinder Re = 50, |mu_1| = 1.0156
 * +- 1e-4, test/neklabTests.py:43-45): it gives 1.0194 (dt = 0.01) / 1.0177 (dt = 0.005), an O(dt) pollution of
 * the BDF start-up of every matvec, while mode 1 gives 1.01578 independent of dt.
 * Process-wide switch. */
int nlg_set_axpby_rst_consistent(int flag);

/* ---------------------------------------------------------------------------------------------- */
/* Krylov basis: what LightKrylov's allocate(X(kdim+1)) + innerprod / linear_combination do with     */
/* k separate dot/axpby calls (call sites src/neklab_analysis.f90:5-8, :77-81), as block kernels     */
/* ---------------------------------------------------------------------------------------------- */
int nlg_basis_create(nlg_mesh *mesh, int nscal, int lorder, int nvec, nlg_basis **out);
int nlg_basis_destroy(nlg_basis *b);
int nlg_basis_vec(nlg_basis *b, int i, nlg_vec **out); /* borrowed view of column i (0-based) */
/* h[0:k] = V(:,0:k)^T B w   (LightKrylov innerprod(X(1:k), w); k allreduces fused into one) */
int nlg_basis_block_dot(const nlg_basis *b, int k, const nlg_vec *w, double *h);
/* w -= V(:,0:k) h           (LightKrylov linear_combination + axpby loop) */
int nlg_basis_block_axpy(const nlg_basis *b, int k, const double *h, nlg_vec *w);
/* classical Gram-Schmidt with one re-orthogonalisation pass, then norm and scale:
 * h[0:k] accumulated coefficients, *beta = ||w|| before normalisation */
int nlg_basis_cgs2(const nlg_basis *b, int k, nlg_vec *w, double *h, double *beta);
/* Block variant for s <= 4 NEW vectors held in the consecutive columns k .. k+s-1 (block Arnoldi, BASELINE.json config 5;
 * the reference advances several perturbations together through Nek5000's lpert / npert, src/neklab_nek_setup.f90:39-247,
 * src/neklab_otd.f90:37-49): classical Gram-Schmidt with one re-orthogonalisation pass against the columns 0 .. k-1 --
 * every pass reads the basis ONCE for all s vectors -- then a Cholesky QR (applied twice) among the s columns.
 * coef is column-major with leading dimension k+s, one column per new vector: rows 0 .. k-1 the projection coefficients
 * (both passes summed), rows k .. k+s-1 the upper-triangular factor R with  W_old = V(:,0:k) coef(0:k,:) + W_new R.
 * Restart history and pressure follow as in axpby / scal (consistent history update only). */
int nlg_basis_block_cgs2(nlg_basis *b, int k, int s, double *coef);
/* Number of columns the last nlg_basis_block_cgs2 on this basis kept.  A column is deflated -- not an error: its coefficients are
 * returned as for any column with a zero diagonal entry in R, and the column becomes the zero vector -- when (1) what the two
 * projections left of it is below 1e-12 of the norm it came with (|w|^2 = |w - V h|^2 + |h|^2: the column lies in span(V) to
 * rounding; the block Krylov space has reached an invariant subspace), or (2) its Cholesky pivot among the columns of the block falls
 * below 1e-14 of its squared norm at entry to the current round of the twofold CholQR (dependent on the columns before it).  The
 * tests are scale-free: columns of any norm may be passed. */
int nlg_basis_last_block_rank(const nlg_basis *b, int *rank);
/* out = sum_j c[j] V(:,j)  over ALL fields (eigenvector reconstruction, LightKrylov eigs tail) */
int nlg_basis_combine(const nlg_basis *b, int k, const double *c, nlg_vec *out);

/* ---------------------------------------------------------------------------------------------- */
/* exptA_linop                                                                                      */
/* ---------------------------------------------------------------------------------------------- */
typedef struct nlg_exptA_config {
    double tau;        /* abstract_exptA_linop_rdp%tau; exptA_linop(1.0_dp, bf) in 1cyl.usr:20       */
    double re;         /* 1/viscosity; 1cyl.par: viscosity = -50                                      */
    double cfl_limit;  /* 0.5 at src/linops/exponential_propagator.f90:12                             */
    double vtol;       /* Helmholtz residual tolerance (param(22), neklab_nek_setup.f90:228)          */
    double ptol;       /* pressure residual tolerance  (param(21), neklab_nek_setup.f90:227)          */
    double dt;         /* > 0: fixed dt, skips the CFL rule (recompute_dt = .false. branch, :218-220) */
    int torder;        /* |param(27)|: 1cyl.par timeStepper = bdf3                                    */
    int maxit_v;
    int maxit_p;
    int fixed_iters_v; /* > 0: run exactly this many PCG iterations (parity / benchmarking mode)      */
    int fixed_iters_p;
    int pprecond;      /* pressure preconditioner: 0 = two-level Schwarz (element FDM with one layer of face overlap
                          in 3-D for lx1 <= 8 + vertex coarse space), 1 = Jacobi, 2 = two-level without overlap    */
    int pproj;         /* pressure residual projection onto the previous increments of the matvec (Nek5000
                          `residualProj = yes`, examples/cylinder/stability/direct/1cyl.par:23): 0 = off, 1 = on    */
    int ifheat;        /* Boussinesq coupling with one scalar (temperature): the vectors carry theta (nscal = 1), the base
                          flow its base temperature.  rhocp (d/dt + U.grad) theta + rhocp u.grad Theta = conductivity
                          lap theta, momentum forcing buoy[i] * theta -- Nek5000's [TEMPERATURE] block and the buoyancy of
                          the case's userf (examples/rayBen/baseflow/rayBen.par:39-45, rayBen.usr:77-103).  rmatvec
                          integrates the adjoint of the coupled operator in the velocity + temperature inner product:
                          u+_t = L_u^+ u+ - theta+ grad Theta,  rhocp theta+_t = rhocp U.grad theta+ + conductivity lap
                          theta+ + rhocp buoy . u+   (exponential_propagator_temp.f90:62-107).                       */
    double conductivity;
    double rhocp;
    double buoy[3];
    int no_history;    /* 1: no restart history -- every matvec starts impulsively (BDF1, then BDF2, ...) from vec_in alone and
                          fills no history slots: vec_in's slots are ignored, vec_out%nrst = 0, vectors of lorder = 1 are accepted.
                          The reference always carries lorder - 1 history copies (src/vectors/neklab_vectors.f90:26-36,
                          exponential_propagator.f90:44,52), which triples the memory of every Krylov vector; this switch is the
                          memory plan for bases that do not fit otherwise (DESIGN.md "memory plan"): the propagator it defines is a
                          slightly different discretisation of exp(tau L) (start-up at first order: +4e-5 in the cylinder's leading
                          multiplier), but ONE linear map for every column of the Arnoldi relation.  0 (default): the reference's
                          protocol. */
} nlg_exptA_config;

int nlg_exptA_config_default(nlg_exptA_config *cfg);
/* exptA_linop(tau, baseflow) constructor (1cyl.usr:20); baseflow is copied (held by value in the
 * reference, neklab_linops.f90:36) */
int nlg_linop_create(nlg_mesh *mesh, const nlg_exptA_config *cfg, const nlg_vec *baseflow, nlg_linop **out);
int nlg_linop_destroy(nlg_linop *op);
/* init_exptA  src/linops/exponential_propagator.f90:4-13: dt / nsteps from CFL, solver set-up */
int nlg_linop_init(nlg_linop *op);
/* exptA_matvec  exponential_propagator.f90:15-60   (interface neklab_linops.f90:52-56) */
int nlg_linop_matvec(nlg_linop *op, const nlg_vec *vec_in, nlg_vec *vec_out);
/* exptA_rmatvec exponential_propagator.f90:62-107  (interface neklab_linops.f90:58-62) */
int nlg_linop_rmatvec(nlg_linop *op, const nlg_vec *vec_in, nlg_vec *vec_out);
/* s <= 4 vectors through the propagator TOGETHER (vec_out[v] = exp(tau L) vec_in[v], transpose != 0: the adjoint), time
 * step by time step: what Nek5000 does with npert > 1 perturbations (src/neklab_nek_setup.f90:39-247, src/neklab_otd.f90:
 * 37-49).  The s velocity solves and the s pressure solves run as independent PCGs in lockstep (own scalars and
 * convergence flag per vector: every vector gets exactly the iteration of nlg_linop_matvec), with the operator
 * applications of an iteration issued together, so that metric factors, base-flow fields of the convective term and
 * preconditioner data are read once per iteration for all vectors.  Each vector keeps its restart-history protocol
 * (replay of vec_in's history, history of vec_out).  Round 4: also with cfg.ifheat (the scalar's solve is lane-batched like the
 * velocity's: exponential_propagator_temp.f90:15-60, :62-107) and with a wavenumber projection (exptA_proj_linop: every projection
 * of the single-vector path applied lane by lane against the operator's tables, exponential_propagator_proj.f90:30-75). */
int nlg_linop_matvec_block(nlg_linop *op, int s, const nlg_vec *const *vec_in, nlg_vec *const *vec_out, int transpose);
/* Newton-Krylov base-flow solver.
 * nonlinear_map  src/systems/fixed_point.f90:4-38 : vec_out = Phi_tau(vec_in) - vec_in with the nonlinear integrator,
 *   time step from the CFL number of vec_in (cfg.cfl_limit; the reference uses 0.4), tolerances cfg.vtol / cfg.ptol.
 *   Afterwards the operator's base-flow dependent set-up belongs to vec_in.
 * set_baseflow   the `self%X` of jac_exptA_matvec (fixed_point.f90:52): replaces the frozen base flow of the linearised
 *   operator and redoes init (dt / nsteps from its CFL number); the Jacobian of the map is then matvec - identity.
 * set_tolerances the tolerance schedulers nek_constant_tol / nek_dynamic_tol (src/systems/neklab_systems.f90:229-335). */
int nlg_linop_nonlinear_map(nlg_linop *op, const nlg_vec *vec_in, nlg_vec *vec_out);
int nlg_linop_set_baseflow(nlg_linop *op, const nlg_vec *baseflow);
int nlg_linop_set_tolerances(nlg_linop *op, double vtol, double ptol);
/* Wavenumber-projected propagator (exptA_proj_linop, src/linops/neklab_linops.f90:130-152,
 * exponential_propagator_proj.f90): after this call matvec / rmatvec project the initial condition and the final state
 * onto the cos(alpha x_idir) / sin(alpha x_idir) content along the homogeneous direction idir (1-based):
 * u <- cv <2 u cv> + sv <2 u sv> with the bm1-weighted average <.> over every line of points along that direction
 * (proj_alpha, :135-173).  line_label[i] identifies the line of local velocity dof i (what Nek5000's gtpp_gs_setup
 * derives from nelx, nely, nelz).  line_label2 / x2 (both or neither): the same for the pressure mesh, with the
 * coordinate along idir of every pressure point -- the pressure is then projected as well (bm2 weights); the reference
 * projects the velocity only, see DESIGN.md for why that is not enough behind an inner product that ignores the
 * pressure.  Several ranks: the labels are GLOBAL line names (the same line carries the same label on every rank
 * that holds a part of it); the weighted sums are all-reduced, as the reference's planar_avg is a global operation
 * (exponential_propagator_proj.f90:146-169).  nlg_linop_project applies the projection to the state held by a vector. */
int nlg_linop_set_projection(nlg_linop *op, double alpha, int idir, const int64_t *line_label, const int64_t *line_label2,
                             const double *x2);
int nlg_linop_project(nlg_linop *op, nlg_vec *v);
/* Resolvent operator by time stepping (resolvent_linop, src/linops/resolvent.f90): the building
 * block of evaluate_rhs (:80-111) and evaluate_imaginary_part (:133-166): vec_out = state after the nsteps of one
 * application started from `ic` (NULL: rest, zero pressure) under the body force Re[(f_re + i f_im) exp(i s omega t)]
 * (velocity fields of f_re / f_im; f_im may be NULL), s = +1 direct, -1 adjoint; the force enters the explicit term and
 * is evaluated at the time level each step starts from.  The driver (real part through GMRES on I - exp(tau L), imaginary
 * part from a quarter period) lives on the host: neklab_amd/host.py resolvent_linop. */
int nlg_linop_integrate_forced(nlg_linop *op, const nlg_vec *ic, const nlg_vec *f_re, const nlg_vec *f_im, double omega, int adjoint,
                               nlg_vec *vec_out);
/* %tau read/written by the driver (src/neklab_analysis.f90:84; apply_exptA neklab_linops.f90:252) */
int nlg_linop_set_tau(nlg_linop *op, double tau);
int nlg_linop_get_info(const nlg_linop *op, double *tau, double *dt, int *nsteps, double *cfl);
/* counters since creation: time steps, Helmholtz iterations, pressure iterations, matvecs
 * (LightKrylov's per-linop matvec counter/timer, src/neklab_analysis.f90:98) */
int nlg_linop_get_stats(const nlg_linop *op, int64_t *steps, int64_t *v_iters, int64_t *p_iters, int64_t *matvecs);

/* Building blocks of the matvec, exposed for operator-level parity tests and for the
 * "operator applies per second" unit of SURVEY.md §8(d).  They act on the fields of nlg_vec objects
 * that live on the same mesh.
 *   helmholtz : out.v_i = [mask_i * QQ^T] (h1 * A + h2 * B) in.v_i    (assemble=0: element-local only)
 *               operator pieces spelled out at src/linops/neklab_linops.f90:332-366
 *   dssum     : v_i <- QQ^T v_i                     (opdssum, real_vectors.f90:100)
 *   cdabdtp   : out.pr = D (mask B^-1 QQ^T) D^T in.pr
 *   opdiv     : out.pr = sum_i D_i in.v_i ;  opgradt: out.v_i = D_i^T in.pr (neklab_linops.f90:368-380)
 *   conv      : out.v_i = weak linearised convective term around `base` (neklab_linops.f90:268-313)
 */
int nlg_op_helmholtz(nlg_mesh *mesh, const nlg_vec *in, nlg_vec *out, double h1, double h2, int assemble);
int nlg_op_dssum(nlg_mesh *mesh, nlg_vec *v);
int nlg_op_cdabdtp(nlg_mesh *mesh, const nlg_vec *in, nlg_vec *out);
/* out%pr = M^-1 in%pr, the two-level Schwarz preconditioner the pressure solve uses for E (Nek5000's role:
   `preconditioner = semg_xxt`, examples/cylinder/stability/direct/1cyl.par:21).  overlap != 0: local solves with one
   layer of face overlap (3-D, lx1 <= 8); with_coarse == 0: local solves only.  M is symmetric positive semi-definite. */
int nlg_op_pprec(nlg_mesh *mesh, const nlg_vec *in, nlg_vec *out, int overlap, int with_coarse);
int nlg_op_opdiv(nlg_mesh *mesh, const nlg_vec *in, nlg_vec *out);
int nlg_op_opgradt(nlg_mesh *mesh, const nlg_vec *in, nlg_vec *out);
int nlg_op_conv(nlg_mesh *mesh, const nlg_vec *base, const nlg_vec *in, nlg_vec *out, int adjoint);
int nlg_op_cfl(nlg_mesh *mesh, const nlg_vec *base, double dt, double *cfl);

/* ---------------------------------------------------------------------------------------------- */
/* eigs: the LightKrylov call at src/neklab_analysis.f90:80-81                                       */
/* ---------------------------------------------------------------------------------------------- */
/* one Arnoldi step on device: basis column k -> column k+1, H(0:k+1, k) written to H (column-major,
 * leading dimension ldh). transpose != 0 uses rmatvec. */
int nlg_arnoldi_step(nlg_linop *op, nlg_basis *basis, int k, double *H, int ldh, int transpose);

/* one BLOCK Arnoldi step: the s columns k .. k+s-1 of the basis -> columns k+s .. k+2s-1 (s matvecs, then
 * nlg_basis_block_cgs2 against the k+s columns before them); H(0:k+2s, k:k+s) written (column-major, ldh >= k+2s). */
int nlg_block_arnoldi_step(nlg_linop *op, nlg_basis *basis, int k, int s, double *H, int ldh, int transpose);

typedef struct nlg_eigs_opts {
    int kdim;               /* kdim=  (1cyl.usr:11: 128)                                          */
    int transpose;          /* transpose= adjoint_                                                 */
    int max_restarts;       /* Krylov-Schur restarts                                               */
    int write_intermediate; /* write_intermediate=.true. -> eigs_output.txt rewritten every step   */
    double tol;             /* tolerance=, <= 0 => sqrt(1e-15) (LightKrylov rtol_dp)               */
    const char *logfile;    /* NULL => "eigs_output.txt"                                           */
    uint64_t seed;          /* start vector seed when x0 == NULL                                   */
    int block_size;         /* > 1: block Arnoldi with this many vectors per step (<= 4): the basis is read once per
                               block in every Gram-Schmidt pass; the first column is x0 / seed, the others are drawn;
                               no restart (kdim = size of the block Krylov space).  0 / 1: Arnoldi + Krylov-Schur  */
    int warm_start;         /* != 0: start from the image A x0 (which carries a restart history like every later Krylov
                               vector) instead of x0 itself; removes the start-vector dependence of the converged Ritz
                               values that the history-free first column causes.  Default 0 = LightKrylov */
} nlg_eigs_opts;

int nlg_eigs_opts_default(nlg_eigs_opts *o);
/* eigs(A, X, eigvals, residuals, info, x0, kdim, transpose, write_intermediate):
 * X[nev] are existing vectors (allocate(eigvecs(nev)); zero_basis, neklab_analysis.f90:77) and are
 * overwritten; eigvals (re, im) and residuals have nev entries; *info = number of matvecs (>0) or
 * a negative error.  Eigenvectors follow the real LAPACK convention (pair = Re, Im consecutive). */
int nlg_eigs(nlg_linop *op, nlg_vec **X, int nev, double *eig_re, double *eig_im, double *residuals,
             int *info, const nlg_vec *x0, const nlg_eigs_opts *opts);

/* svds(A, U, S, V, residuals, info, kdim=, write_intermediate=) -- the LightKrylov call of
 * transient_growth_analysis_fixed_point (src/neklab_analysis.f90:136; examples/back_fstep/transient_growth/
 * bfs.usr:8-21): leading singular triplets of exp(tau L) by Lanczos bidiagonalisation with matvec + rmatvec.
 * U[nsv] (optimal responses) and V[nsv] (optimal perturbations) are existing vectors and are overwritten;
 * S and residuals have nsv entries; *info = number of operator applications.  opts->kdim, tol, seed,
 * write_intermediate, logfile ("svds_output.txt") as for nlg_eigs; u0 (may be NULL) is the start vector. */
int nlg_svds(nlg_linop *op, nlg_vec **U, nlg_vec **V, int nsv, double *S, double *residuals, int *info,
             const nlg_vec *u0, const nlg_eigs_opts *opts);

/* symmetric tridiagonal eigen-decomposition (diagonal d[n], sub-diagonal e[1..n-1]); eigenvalues ascending in d,
 * eigenvectors as columns of the row-major n x n array Z.  Host helper of nlg_svds, exported for unit tests. */
int nlg_symtridiag_eig(int n, double *d, double *e, double *Z);

/* host-side dense helper used by nlg_eigs, exported for unit tests: eigen-decomposition of a real
 * n x n matrix (column-major, lda); vr column-major complex pairs as LAPACK dgeev. */
int nlg_dense_eig(int n, const double *A, int lda, double *wr, double *wi, double *vr, int ldvr);

#ifdef __cplusplus
}
#endif
#endif /* NEKLAB_GPU_H */
