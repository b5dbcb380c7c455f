!> Umbrella module, as the reference's src/neklab.f90:28-93: user code says `use neklab` only
!! (examples/cylinder/stability/direct/1cyl.usr:4).
module neklab
   use LightKrylov, only: dp
   use neklab_gpu_capi, only: neklab_gpu_init, neklab_gpu_set_mesh, neklab_gpu_set_case, neklab_gpu_finalize, nlg_check, &
                              nlg_exptA_config, nek_case, nek_endtime
   use neklab_vectors
   use neklab_linops
   use neklab_utils
   use neklab_systems
   use neklab_analysis
   implicit none
   public
end module neklab

!> name of the round-1 shim module, kept so that existing host programs keep compiling
module neklab_gpu
   use neklab
   implicit none
   public
end module neklab_gpu
