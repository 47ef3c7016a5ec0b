"""Loads librxgpu.so and declares the C ABI of include/rxgpu.h for ctypes."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# $RXGPU_LIB_FLAVOUR=fi: the TEST build with the fault-injection hook (tests/test_gpu_dropin.py); any other name: a scratch build
# librxgpu_<name>.so beside it (A/B of compile-time choices on one box); the product is librxgpu.so
_FLAVOUR = os.environ.get("RXGPU_LIB_FLAVOUR", "")
LIB_PATH = os.path.join(_HERE, "librxgpu_%s.so" % _FLAVOUR if _FLAVOUR.isalnum() else "librxgpu.so")


class RxGpuError(RuntimeError):
    pass


_lib = None

# every symbol include/rxgpu.h declares; tests/test_abi.py checks they are all exported
EXPORTS = [
    "rxgpu_init", "rxgpu_shutdown", "rxgpu_device_count", "rxgpu_last_error", "rxgpu_stream", "rxgpu_sync", "rxgpu_knobs_reload",
    "rxgpu_pin", "rxgpu_unpin",
    "rxgpu_prof_enable", "rxgpu_prof_reset", "rxgpu_prof_get", "rxgpu_diag_stream_rate",
    "rxgpu_full_demod", "rxgpu_callback", "rxgpu_fatal", "rxgpu_deemph_state", "rxgpu_set_demod_functions", "rxgpu_dropin_invalidate", "rxgpu_dropin_block_rms", "rxgpu_dropin_release", "rxgpu_dropin_pin", "rxgpu_dropin_unpin", "rxgpu_dropin_timing",
    "rxgpu_fm_params_init", "rxgpu_fm_plan_settings",
    "rxgpu_fm_stream_create", "rxgpu_fm_stream_destroy", "rxgpu_fm_stream_set_carry", "rxgpu_fm_stream_get_carry",
    "rxgpu_fm_stream_run", "rxgpu_fm_stream_run_async", "rxgpu_fm_stream_wait", "rxgpu_fm_stream_run_host",
    "rxgpu_fm_stream_host_fixups",
    "rxgpu_chan_create", "rxgpu_chan_destroy", "rxgpu_chan_set_carry", "rxgpu_chan_get_carry", "rxgpu_chan_run", "rxgpu_chan_run_async", "rxgpu_chan_wait",
    "rxgpu_chan_set_audio_carry", "rxgpu_chan_get_audio_carry",
    "rxgpu_chan_host_fixups",
    "rxgpu_scan", "rxgpu_scan_sync", "rxgpu_scan_syncs", "rxgpu_scan_zero_copy", "rxgpu_scan_sync_in_place", "rxgpu_scan_rows_cleared", "rxgpu_scan_release", "rxgpu_scan_deferred", "rxgpu_scan_timing", "rxgpu_csv_dbm", "rxgpu_power_plan_range", "rxgpu_sine_table", "rxgpu_window_coefs",
    "rxgpu_power_scan_create", "rxgpu_power_scan_destroy", "rxgpu_power_scan_run",
    "rxgpu_comm_unique_id", "rxgpu_comm_create", "rxgpu_comm_adopt", "rxgpu_comm_destroy", "rxgpu_comm_rank", "rxgpu_comm_world",
    "rxgpu_comm_gathers", "rxgpu_comm_library", "rxgpu_shard_tunes", "rxgpu_power_gather", "rxgpu_power_scan_run_sharded",
    "rxgpu_sdr_in_bytes", "rxgpu_sdr_out_bytes", "rxgpu_sdr_convert", "rxgpu_sdr_convert_host", "rxgpu_wav_header",
]


def lib():
    """The loaded library.  Raises if it has not been built -- there is no fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RxGpuError(
                "librxgpu.so is missing (%s); build it with `make -C rx_tools_amd/csrc` or "
                "__graft_entry__.build().  rx_tools_amd has no CPU fallback." % LIB_PATH)
        # torch (used by tests/bench for device buffers and torch.distributed) bundles its own HIP
        # runtime; if it is going to share the process it has to be loaded first, otherwise the
        # two runtimes' global symbols interleave and device discovery fails.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        L.rxgpu_last_error.restype = C.c_char_p
        L.rxgpu_stream.restype = C.c_void_p
        L.rxgpu_deemph_state.restype = C.POINTER(C.c_int)
        L.rxgpu_deemph_state.argtypes = [C.c_void_p]
        L.rxgpu_init.argtypes = [C.c_int]
        L.rxgpu_knobs_reload.restype = None
        L.rxgpu_scan_release.restype = None
        L.rxgpu_prof_enable.argtypes = [C.c_int]
        L.rxgpu_diag_stream_rate.argtypes = [C.c_int, C.c_size_t, C.c_int, C.POINTER(C.c_double)]
        L.rxgpu_fm_stream_host_fixups.restype = C.c_long
        L.rxgpu_fm_stream_host_fixups.argtypes = [C.c_void_p]
        L.rxgpu_prof_get.argtypes = [C.c_char_p, C.POINTER(C.c_double), C.POINTER(C.c_long)]
        L.rxgpu_fm_stream_create.argtypes = [C.POINTER(C.c_void_p), C.c_void_p, C.c_size_t, C.c_size_t]
        L.rxgpu_fm_stream_destroy.argtypes = [C.c_void_p]
        L.rxgpu_fm_stream_set_carry.argtypes = [C.c_void_p, C.c_void_p]
        L.rxgpu_fm_stream_get_carry.argtypes = [C.c_void_p, C.c_void_p]
        L.rxgpu_fm_stream_run.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_size_t,
                                          C.POINTER(C.c_size_t), C.POINTER(C.c_int)]
        L.rxgpu_fm_stream_run_host.argtypes = L.rxgpu_fm_stream_run.argtypes
        L.rxgpu_fm_stream_run_async.argtypes = L.rxgpu_fm_stream_run.argtypes
        L.rxgpu_fm_stream_wait.argtypes = [C.c_void_p]
        L.rxgpu_chan_create.argtypes = [C.POINTER(C.c_void_p), C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p]
        L.rxgpu_chan_destroy.argtypes = [C.c_void_p]
        L.rxgpu_chan_set_carry.argtypes = [C.c_void_p, C.c_void_p]
        L.rxgpu_chan_get_carry.argtypes = [C.c_void_p, C.c_void_p]
        L.rxgpu_chan_set_audio_carry.argtypes = [C.c_void_p, C.c_void_p]
        L.rxgpu_chan_get_audio_carry.argtypes = [C.c_void_p, C.c_void_p]
        L.rxgpu_chan_run.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
        L.rxgpu_chan_run_async.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_size_t]
        L.rxgpu_chan_wait.argtypes = [C.c_void_p, C.POINTER(C.c_size_t)]
        L.rxgpu_chan_host_fixups.argtypes = [C.c_void_p]
        L.rxgpu_chan_host_fixups.restype = C.c_long
        L.rxgpu_power_scan_create.argtypes = [C.POINTER(C.c_void_p), C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.rxgpu_power_scan_destroy.argtypes = [C.c_void_p]
        L.rxgpu_power_scan_run.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.rxgpu_power_plan_range.argtypes = [C.c_char_p, C.c_double, C.c_int, C.c_void_p]
        L.rxgpu_sine_table.argtypes = [C.c_int, C.c_void_p]
        L.rxgpu_window_coefs.argtypes = [C.c_char_p, C.c_int, C.c_void_p]
        L.rxgpu_scan.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.rxgpu_scan_sync.argtypes = [C.c_void_p, C.c_int]
        L.rxgpu_scan_syncs.restype = C.c_long
        L.rxgpu_scan_deferred.argtypes = [C.c_int]
        L.rxgpu_csv_dbm.argtypes = [C.c_void_p, C.c_void_p]
        L.rxgpu_full_demod.argtypes = [C.c_void_p]
        L.rxgpu_set_demod_functions.argtypes = [C.c_void_p] * 5
        L.rxgpu_callback.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
        L.rxgpu_sdr_in_bytes.restype = C.c_size_t
        L.rxgpu_sdr_in_bytes.argtypes = [C.c_int, C.c_size_t]
        L.rxgpu_sdr_out_bytes.restype = C.c_size_t
        L.rxgpu_sdr_out_bytes.argtypes = [C.c_int, C.c_size_t]
        L.rxgpu_sdr_convert.argtypes = [C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]
        L.rxgpu_sdr_convert_host.argtypes = [C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]
        L.rxgpu_pin.argtypes = [C.c_void_p, C.c_size_t]
        L.rxgpu_unpin.argtypes = [C.c_void_p]
        L.rxgpu_dropin_invalidate.argtypes = [C.c_void_p]
        L.rxgpu_dropin_invalidate.restype = None
        L.rxgpu_dropin_block_rms.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        L.rxgpu_dropin_release.argtypes = [C.c_void_p]
        L.rxgpu_dropin_timing.argtypes = [C.POINTER(C.c_double), C.c_int]
        L.rxgpu_dropin_pin.argtypes = [C.c_void_p, C.c_void_p]
        L.rxgpu_dropin_unpin.argtypes = [C.c_void_p, C.c_void_p]
        L.rxgpu_fm_params_init.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_int)]
        L.rxgpu_fm_plan_settings.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.rxgpu_comm_unique_id.argtypes = [C.c_void_p]
        L.rxgpu_comm_create.argtypes = [C.POINTER(C.c_void_p), C.c_void_p, C.c_int, C.c_int]
        L.rxgpu_comm_adopt.argtypes = [C.POINTER(C.c_void_p), C.c_void_p, C.c_int, C.c_int]
        L.rxgpu_comm_destroy.argtypes = [C.c_void_p]
        L.rxgpu_comm_destroy.restype = None
        L.rxgpu_comm_rank.argtypes = [C.c_void_p]
        L.rxgpu_comm_world.argtypes = [C.c_void_p]
        L.rxgpu_comm_gathers.argtypes = [C.c_void_p]
        L.rxgpu_comm_gathers.restype = C.c_long
        L.rxgpu_comm_library.restype = C.c_char_p
        L.rxgpu_shard_tunes.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.rxgpu_power_gather.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
        L.rxgpu_power_scan_run_sharded.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                                   C.c_int, C.c_void_p, C.c_void_p, C.c_int]
        L.rxgpu_wav_header.restype = None
        L.rxgpu_wav_header.argtypes = [C.c_int, C.c_int, C.c_void_p]
        _lib = L
    return _lib


def check(rc):
    if rc != 0:
        raise RxGpuError("librxgpu error %d: %s" % (rc, lib().rxgpu_last_error().decode()))
