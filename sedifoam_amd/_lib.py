"""Loads libsedifoam_amd.so (the HIP/gfx950 product library) through ctypes.

There is deliberately NO fallback: if the shared library is missing or no HIP device is usable the
import / engine creation raises.  The CPU oracle under oracle/ is test infrastructure and is never
imported from here.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# SF_LIB_PATH: development knob to A/B a differently compiled build of the same library (tests/build_variant.sh)
LIB_PATH = os.environ.get("SF_LIB_PATH") or os.path.join(HERE, "libsedifoam_amd.so")

dp = C.POINTER(C.c_double)
ip = C.POINTER(C.c_int)
vp = C.c_void_p


class SfError(RuntimeError):
    pass


class HaloLayout(C.Structure):
    _fields_ = [("world", C.c_int), ("shift_left", C.c_double), ("shift_right", C.c_double),
                ("soff_l", C.c_longlong), ("soff_r", C.c_longlong),
                ("roff_l", C.c_longlong), ("n_from_left", C.c_longlong),
                ("roff_r", C.c_longlong), ("n_from_right", C.c_longlong),
                ("send_off", C.POINTER(C.c_longlong)), ("send_cnt", C.POINTER(C.c_longlong)),
                ("recv_off", C.POINTER(C.c_longlong)), ("recv_cnt", C.POINTER(C.c_longlong)),
                ("dev_shdr", vp), ("dev_rhdr", vp), ("dev_tx", vp), ("dev_rx", vp)]


class DemInfo(C.Structure):
    _fields_ = [("nlocal", C.c_int), ("nghost", C.c_int), ("capacity", C.c_int),
                ("max_neigh_used", C.c_int), ("max_neigh_cap", C.c_int),
                ("nbuilds", C.c_longlong), ("nsteps", C.c_longlong),
                ("npairs_full", C.c_longlong)]


class DemDeviceView(C.Structure):
    _fields_ = [("xr", vp), ("vm", vp), ("om", vp), ("force", vp), ("torque", vp),
                ("fdrag", vp), ("DuDt", vp), ("vOld", vp), ("tag", vp), ("type", vp),
                ("foamCpuId", vp), ("nlocal", C.c_int), ("nghost", C.c_int),
                ("capacity", C.c_int), ("stream", vp)]


class GranParams(C.Structure):
    _fields_ = [("kn", C.c_double), ("kt", C.c_double), ("gamman", C.c_double),
                ("gammat", C.c_double), ("xmu", C.c_double), ("dampflag", C.c_int)]


class LubParams(C.Structure):
    _fields_ = [("mu", C.c_double), ("flaglog", C.c_int), ("flagfld", C.c_int),
                ("flagHI", C.c_int), ("flagVF", C.c_int), ("cut_inner", C.c_double),
                ("cut_global", C.c_double), ("R0", C.c_double), ("RT0", C.c_double),
                ("RS0", C.c_double), ("vxmu2f", C.c_double)]


class CloudProps(C.Structure):
    _fields_ = [("dragModel", C.c_int), ("subCycles", C.c_int), ("particleDrag", C.c_int),
                ("particlePressureGrad", C.c_int), ("particleBuoyancy", C.c_int),
                ("particleAddedMass", C.c_int), ("particleLift", C.c_int),
                ("lubricationForce", C.c_int), ("gravity", C.c_double * 3),
                ("rhob", C.c_double), ("nub", C.c_double), ("maxPossibleAlpha", C.c_double),
                ("diffusionBandWidth", C.c_double), ("diffusionSteps", C.c_int), ("UfSmooth", C.c_int),
                ("UpSmooth", C.c_int), ("dragSmooth", C.c_int), ("alphaSmooth", C.c_int),
                ("smoothDirection", C.c_double * 3), ("particleHistoryForce", C.c_int),
                ("addParticleOption", C.c_int), ("inletForce", C.c_double * 3), ("inletBox", C.c_double * 9),
                ("eccentricity", C.c_double * 3)]


class CloudMesh(C.Structure):
    _fields_ = [("origin", C.c_double * 3), ("dx", C.c_double * 3), ("n", C.c_int * 3),
                ("faces", C.POINTER(C.c_double) * 3), ("cell_label", C.POINTER(C.c_int)),
                ("periodic", C.c_int * 3), ("slab_nx_global", C.c_int)]


class CloudTimers(C.Structure):
    _fields_ = [("evolve", C.c_double), ("calcTc", C.c_double), ("dragOnParticles", C.c_double),
                ("lammps", C.c_double), ("particleMove", C.c_double), ("scatter", C.c_double)]


_SIGS = {
    "sf_last_error": (C.c_char_p, []),
    "sf_device_check": (C.c_int, []),
    "sf_version": (C.c_char_p, []),
    "sf_lammps_open": (C.c_int, [C.c_int, vp, C.c_ssize_t, C.POINTER(vp)]),
    "sf_lammps_open_world": (C.c_int, [C.c_int, vp, C.c_ssize_t, C.c_int, C.c_int, C.c_char_p, C.POINTER(vp)]),
    "sf_procgrid_choose": (C.c_int, [C.c_int, dp, dp, ip, ip]),
    "sf_lammps_close": (C.c_int, [vp]),
    "sf_lammps_file": (C.c_int, [vp, C.c_char_p]),
    "sf_lammps_command": (C.c_char_p, [vp, C.c_char_p]),
    "sf_lammps_sync": (C.c_int, [vp]),
    "sf_lammps_get_global_n": (C.c_int, [vp]),
    "sf_lammps_get_initial_np": (C.c_int, [vp, ip]),
    "sf_lammps_get_initial_info": (C.c_int, [vp, dp, dp, dp, dp, ip, ip, ip]),
    "sf_lammps_get_local_n": (C.c_int, [vp]),
    "sf_lammps_get_local_domain": (C.c_int, [vp, dp]),
    "sf_lammps_get_local_info": (C.c_int, [vp, dp, dp, ip, ip, ip]),
    "sf_lammps_put_local_info": (C.c_int, [vp, C.c_int, dp, dp, ip, ip]),
    "sf_lammps_step": (C.c_int, [vp, C.c_int]),
    "sf_lammps_set_timestep": (C.c_int, [vp, C.c_double]),
    "sf_lammps_get_timestep": (C.c_double, [vp]),
    "sf_lammps_create_particle": (C.c_int, [vp, C.c_int, dp, dp, C.c_double, C.c_double, C.c_int, dp]),
    "sf_lammps_delete_particle": (C.c_int, [vp, ip, C.c_int]),
    "sf_dem_create_atoms": (C.c_int, [vp, C.c_int, dp, dp, dp, dp, dp, ip, ip]),
    "sf_dem_set_box": (C.c_int, [vp, dp, dp]),
    "sf_dem_get_info": (C.c_int, [vp, C.POINTER(DemInfo)]),
    "sf_dem_device_view_get": (C.c_int, [vp, C.POINTER(DemDeviceView)]),
    "sf_dem_set_profiling": (C.c_int, [vp, C.c_int]),
    "sf_dem_get_profile": (C.c_int, [vp, C.POINTER(C.c_longlong), C.POINTER(C.c_double)]),
    "sf_dem_get_rebuild_profile": (C.c_int, [vp, C.POINTER(C.c_longlong), C.POINTER(C.c_double)]),
    "sf_dem_get_forces": (C.c_int, [vp, dp, dp, dp, ip]),
    "sf_dem_get_history": (C.c_longlong, [vp, C.c_longlong, ip, ip, dp]),
    "sf_dem_get_wall_shear": (C.c_int, [vp, C.c_int, dp]),
    "sf_dem_set_subdomain": (C.c_int, [vp, C.c_int, C.c_int, C.c_double, C.c_double]),
    "sf_dem_setup": (C.c_int, [vp]),
    "sf_dem_run_begin": (C.c_int, [vp]),
    "sf_dem_substep": (C.c_int, [vp, C.c_int]),
    "sf_dem_need_rebuild": (C.c_int, [vp]),
    "sf_dem_substep_k": (C.c_int, [vp, C.c_int, C.c_int]),
    "sf_dem_batch_end": (C.c_int, [vp, C.c_int, C.c_int, ip]),
    "sf_dem_set_flag_buffer": (C.c_int, [vp, vp]),
    "sf_dem_rebuild_begin": (C.c_int, [vp]),
    "sf_dem_rebuild_sort": (C.c_int, [vp]),
    "sf_dem_rebuild_finish": (C.c_int, [vp]),
    "sf_dem_border_pack": (C.c_longlong, [vp, C.c_int, C.c_double, vp, C.c_longlong]),
    "sf_dem_border_unpack": (C.c_int, [vp, C.c_int, vp, C.c_longlong]),
    "sf_dem_forward_pack": (C.c_longlong, [vp, C.c_int, C.c_double, vp]),
    "sf_dem_forward_unpack": (C.c_int, [vp, C.c_int, vp, C.c_longlong]),
    "sf_dem_forward_pack2": (C.c_int, [vp, C.c_double, vp, C.c_double, vp, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]),
    "sf_dem_forward_unpack2": (C.c_int, [vp, vp, C.c_longlong, vp, C.c_longlong]),
    "sf_dem_forward_pack_fused": (C.c_int, [vp, C.c_double, C.c_longlong, C.c_double, C.c_longlong, vp, C.c_int, vp]),
    "sf_dem_forward_unpack_fused": (C.c_int, [vp, vp, C.c_longlong, C.c_longlong, C.c_longlong, C.c_longlong, vp,
                                              C.c_int, C.c_int]),
    "sf_dem_set_overlap": (C.c_int, [vp, C.c_int, vp]),
    "sf_dem_partition_streams": (C.c_int, [vp, C.c_int, C.POINTER(vp), C.POINTER(vp)]),
    "sf_dem_overlap_begin": (C.c_int, [vp]),
    "sf_dem_substep_part": (C.c_int, [vp, C.c_int, C.c_int, C.c_int]),
    "sf_dem_substep_flip": (C.c_int, [vp, C.c_int]),
    "sf_dem_overlap_batch_end": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, ip]),
    "sf_dem_boundary_count": (C.c_int, [vp]),
    "sf_dem_comm_unique_id": (C.c_int, [C.c_char_p]),
    "sf_dem_comm_init": (C.c_int, [vp, C.c_char_p, C.c_int, C.c_int]),
    "sf_dem_halo_run": (C.c_int, [vp, C.c_int, C.c_int, vp, ip]),
    "sf_slab_init": (C.c_int, [vp, C.c_char_p, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int]),
    "sf_brick_init": (C.c_int, [vp, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "sf_brick_pattern": (C.c_int, [C.c_int] * 4 + [C.POINTER(C.c_int)] * 8),
    "sf_cloud_slab_halo_add": (C.c_int, [vp, C.c_int]),
    "sf_cloud_slab_phase": (C.c_int, [vp, C.c_int]),
    "sf_cloud_slab_info": (C.c_int, [vp, C.POINTER(vp), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "sf_slab_setup": (C.c_int, [vp]),
    "sf_slab_rebuild": (C.c_int, [vp]),
    "sf_slab_step": (C.c_int, [vp, C.c_int]),
    "sf_slab_active": (C.c_int, [vp]),
    "sf_slab_direct_halo": (C.c_int, [vp]),
    "sf_slab_allreduce_sum": (C.c_int, [vp, dp, C.c_int]),
    "sf_slab_rebuild_count": (C.c_longlong, [vp]),
    "sf_slab_comm_info": (C.c_int, [vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_char_p, C.c_int]),
    "sf_slab_exchange_profile": (C.c_int, [vp, C.POINTER(C.c_longlong), C.POINTER(C.c_double)]),
    "sf_slab_rebuild_profile": (C.c_int, [vp, C.POINTER(C.c_longlong), C.POINTER(C.c_double)]),
    "sf_slab_layout_get": (C.c_int, [vp, vp]),
    "sf_dem_local_particle_volume": (C.c_int, [vp, dp]),
    "sf_dem_set_global_particle_volume": (C.c_int, [vp, C.c_double]),
    "sf_dem_local_max_radius": (C.c_int, [vp, dp]),
    "sf_dem_set_global_max_radius": (C.c_int, [vp, C.c_double]),
    "sf_dem_migrate_count": (C.c_longlong, [vp]),
    "sf_dem_migrate_pack": (C.c_longlong, [vp, C.c_int, C.c_double, vp, C.c_longlong]),
    "sf_dem_migrate_unpack": (C.c_int, [vp, vp, C.c_longlong]),
    "sf_dem_migrate_record_doubles": (C.c_int, [vp]),
    "sf_dem_migrate_set_slots": (C.c_int, [vp, C.c_int]),
    "sf_dem_ghost_forward_local": (C.c_int, [vp]),
    "sf_dem_set_stream": (C.c_int, [vp, vp]),
    "sf_dev_alloc": (vp, [C.c_size_t]),
    "sf_dev_free": (C.c_int, [vp]),
    "sf_dev_upload": (C.c_int, [vp, vp, C.c_size_t, vp]),
    "sf_dev_download": (C.c_int, [vp, vp, C.c_size_t, vp]),
    "sf_dev_zero": (C.c_int, [vp, C.c_size_t, vp]),
    "sf_dev_sync": (C.c_int, [vp]),
    "sfk_gran_settings": (C.c_int, [C.POINTER(GranParams), C.c_double, C.c_int, C.c_double, C.c_double,
                                    C.c_int, C.c_double, C.c_double, C.c_int, C.c_double]),
    "sfk_pair_gran_history_compute_rigid": (C.c_int, [C.c_int, C.POINTER(GranParams), C.c_double, C.c_int, C.c_int,
                                                      C.c_int] + [vp] * 11 + [C.c_int, vp, vp, vp, vp]),
    "sfk_pair_gran_history_compute": (C.c_int, [C.c_int, C.POINTER(GranParams), C.c_double, C.c_int, C.c_int,
                                                C.c_int] + [vp] * 11 + [C.c_int, vp, vp, vp]),
    "sfk_fix_cohesive_post_force": (C.c_int, [C.c_double] * 4 + [C.c_int, C.c_int, C.c_int] + [vp] * 6 +
                                    [C.c_int, vp, vp]),
    "sfk_pair_lubricate_poly_compute": (C.c_int, [C.POINTER(LubParams), C.c_int] + [vp] * 10),
    "sfk_fix_wall_granfix_post_force": (C.c_int, [C.c_int, C.POINTER(GranParams), C.c_int, C.c_double, C.c_double,
                                                  C.c_double, C.c_double, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, C.c_int,
                                                  vp, vp, vp, vp]),
    "sfk_fix_fluid_drag_post_force": (C.c_int, [C.c_int, C.c_double, C.c_double, vp, vp, vp, vp, C.c_int,
                                                vp, vp, vp, vp, vp]),
    "sfk_drag_model_jd": (C.c_int, [C.c_int, C.c_int, vp, vp, vp, C.c_double, C.c_double, vp, vp]),
    "sfk_cell_owner": (C.c_int, [C.c_int, vp, dp, dp, ip, vp, vp]),
    "sfk_cell_owner_graded": (C.c_int, [C.c_int, vp, dp, dp, ip, C.POINTER(C.c_void_p), vp, vp]),
    "sf_cloud_create": (C.c_int, [vp, C.POINTER(CloudMesh), C.POINTER(CloudProps), C.c_double, C.POINTER(vp)]),
    "sf_cloud_destroy": (C.c_int, [vp]),
    "sf_cloud_set_fluid": (C.c_int, [vp, dp, dp, dp, dp]),
    "sf_cloud_evolve": (C.c_int, [vp]),
    "sf_cloud_calc_tc_fields": (C.c_int, [vp]),
    "sf_cloud_smooth_field": (C.c_int, [vp, dp, C.c_int]),
    "sf_cloud_phase": (C.c_int, [vp, C.c_int]),
    "sf_cloud_smooth_work": (C.c_int, [vp, C.POINTER(dp), ip]),
    "sf_cloud_smooth_xsolve": (C.c_int, [vp, vp, C.c_longlong, C.c_longlong]),
    "sf_cloud_sub_cycling": (C.c_int, [vp, ip, ip]),
    "sf_cloud_device_fields": (C.c_int, [vp, C.POINTER(dp), C.POINTER(dp), C.POINTER(dp), ip]),
    "sf_cloud_get_fields": (C.c_int, [vp, dp, dp, dp, dp]),
    "sf_cloud_get_particles": (C.c_int, [vp, ip, ip, dp, dp]),
    "sf_cloud_particle_count": (C.c_int, [vp]),
    "sf_cloud_average_info": (C.c_int, [vp, dp]),
    "sf_cloud_adjust_timestep": (C.c_int, [C.c_double, C.c_double, C.c_int, dp, ip, ip, ip]),
    "sf_cloud_get_timers": (C.c_int, [vp, C.POINTER(CloudTimers)]),
}

_lib = None


def exported_symbols():
    """every C-ABI symbol include/sedifoam_amd.h declares"""
    return sorted(_SIGS)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SfError("libsedifoam_amd.so is not built: run `python -c 'import __graft_entry__ as g; "
                          "g.build()'` (hipcc, gfx950).  There is no CPU fallback.")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            if os.environ.get("SF_LIB_ALLOW_MISSING") == "1" and not hasattr(L, name):
                continue            # development: A/B against an older build of the library (SF_LIB_PATH)
            fn = getattr(L, name)   # AttributeError if a declared symbol is missing
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc):
    if rc is not None and rc < 0:
        raise SfError(lib().sf_last_error().decode())
    return rc
