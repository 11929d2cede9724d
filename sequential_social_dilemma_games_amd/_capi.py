"""ctypes binding of libssd_hip.so (the C ABI declared in include/ssd.h).

There is no fallback: if the shared library is missing or cannot be loaded this module
raises, and every product entry point (engine, envs, bench) fails with it.
"""
import ctypes as C
import os

_PKG = os.path.dirname(os.path.abspath(__file__))
# SSD_LIB_PATH lets tools/phase_profile.py load the diagnostic (stamped) build of the SAME sources, and the tests that need
# test hooks the test-hook build (libssd_hip_testhooks.so: the product sources with -DSSD_TESTHOOKS).
LIB_PATH = os.environ.get("SSD_LIB_PATH") or os.path.join(_PKG, "libssd_hip.so")

SSD_OK, SSD_E_INVALID, SSD_E_DEVICE, SSD_E_NOMEM, SSD_E_STATE = 0, -1, -2, -3, -4
SSD_HOST_PTRS, SSD_NO_ROTATE, SSD_OBS_F32, SSD_ROLLOUT_FUSED, SSD_AUTO_RESET, SSD_ROLLOUT_AUTO = 1, 2, 4, 8, 16, 128
SSD_PATH_AQL, SSD_PATH_COHERENT, SSD_PATH_SPLIT, SSD_PATH_FUSED, SSD_PATH_SYNC, SSD_PATH_QUEUE_DROPPED, SSD_PATH_FORKED = 1, 2, 4, 8, 16, 32, 64
SSD_ST_BAD_ACTION, SSD_ST_NO_SPAWN, SSD_ST_MOVE_LOOKUP, SSD_ST_WAIT_TIMEOUT = 1, 2, 4, 8
ABI_VERSION = 3

# every symbol include/ssd.h declares
SYMBOLS = ("ssd_create", "ssd_destroy", "ssd_reset", "ssd_step", "ssd_step_random", "ssd_rollout_random", "ssd_rollout_actions", "ssd_rollout_path", "ssd_set_rollout_chains",
           "ssd_profiler_attached", "ssd_observe",
           "ssd_get_state", "ssd_set_state", "ssd_get_waste_count", "ssd_render_full", "ssd_render_frames", "ssd_agent_action_obs", "ssd_set_horizon", "ssd_potential_waste_area",
           "ssd_device_status", "ssd_synchronize", "ssd_last_error", "ssd_abi_version")


class SsdConfig(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("game", C.c_int32), ("height", C.c_int32), ("width", C.c_int32),
                ("base_map", C.c_char_p), ("num_envs", C.c_int32), ("num_agents", C.c_int32),
                ("view_len", C.c_int32), ("beam_len", C.c_int32), ("seed", C.c_uint64),
                ("env_index_base", C.c_uint32), ("device_id", C.c_int32), ("keep_beams", C.c_int32),
                ("color_lut", C.c_void_p), ("harvest_thresholds", C.c_void_p),
                ("cleanup_apple_thresholds", C.c_void_p), ("cleanup_waste_thresholds", C.c_void_p)]


class SsdError(RuntimeError):
    pass


_lib = None


def _preload_torch_hip_runtime():
    """PyTorch-ROCm wheels bundle their own libamdhip64.so (SONAME libamdhip64.so.7).  Two HIP
    runtimes in one process cannot both initialise the GPU, so bind libssd_hip.so to the copy torch
    will use: load it first, and the dynamic linker resolves our NEEDED entry to it by SONAME.
    Without torch installed the system runtime (/opt/rocm) is used."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return None
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        return C.CDLL(cand, mode=C.RTLD_GLOBAL)
    return None


_hip_runtime = None


def lib():
    """The loaded library; raises SsdError when the HIP extension has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SsdError("HIP extension missing: %s (build it with `python -c 'import __graft_entry__ as g; "
                           "g.build()'` or `make -C sequential_social_dilemma_games_amd/csrc`); there is no CPU "
                           "fallback" % LIB_PATH)
        global _hip_runtime
        try:
            _hip_runtime = _preload_torch_hip_runtime()
            L = C.CDLL(LIB_PATH)
        except OSError as exc:
            raise SsdError("cannot load %s: %s (no CPU fallback)" % (LIB_PATH, exc))
        vp, u32, i32 = C.c_void_p, C.c_uint32, C.c_int32
        L.ssd_create.argtypes = [C.POINTER(SsdConfig), C.POINTER(vp)]
        L.ssd_destroy.argtypes = [vp]
        L.ssd_reset.argtypes = [vp, vp, vp, u32, vp]
        L.ssd_step.argtypes = [vp, vp, vp, vp, vp, vp, u32, vp]
        L.ssd_step_random.argtypes = [vp, i32, vp, vp, vp, vp, u32, vp]
        L.ssd_rollout_random.argtypes = [vp, i32, i32, i32, i32, vp, vp, vp, i32, u32, vp]
        L.ssd_rollout_actions.argtypes = [vp, vp, vp, i32, i32, i32, i32, vp, vp, vp, i32, u32, vp]
        L.ssd_profiler_attached.argtypes = []
        L.ssd_rollout_path.argtypes = [vp]
        L.ssd_set_rollout_chains.argtypes = [vp, i32]
        L.ssd_observe.argtypes = [vp, vp, u32, vp]
        L.ssd_get_state.argtypes = [vp] + [vp] * 6
        L.ssd_set_state.argtypes = [vp] + [vp] * 6
        L.ssd_get_waste_count.argtypes = [vp, vp]
        L.ssd_render_full.argtypes = [vp, i32, vp]
        L.ssd_render_frames.argtypes = [vp, i32, i32, vp, u32, vp]
        L.ssd_agent_action_obs.argtypes = [vp, vp, vp, vp, vp, u32, vp]
        L.ssd_set_horizon.argtypes = [vp, i32]
        L.ssd_potential_waste_area.argtypes = [vp]
        L.ssd_device_status.argtypes = [vp, C.POINTER(u32), C.c_int]
        L.ssd_synchronize.argtypes = [vp]
        L.ssd_last_error.argtypes = [vp]
        L.ssd_last_error.restype = C.c_char_p
        L.ssd_abi_version.argtypes = []
        for name in SYMBOLS:
            getattr(L, name)
        if L.ssd_abi_version() != ABI_VERSION:
            raise SsdError("libssd_hip.so ABI %d != expected %d: rebuild" % (L.ssd_abi_version(), ABI_VERSION))
        _lib = L
    return _lib


def check(rc, handle=None):
    if rc != SSD_OK:
        msg = lib().ssd_last_error(handle)
        raise SsdError("libssd_hip call failed (%d): %s" % (rc, msg.decode() if msg else "?"))
