"""ctypes mirrors of the reference's state structs (x86-64 glibc layout).

These are the structs the drop-in entry points take by pointer:
  struct dongle_state   /root/reference/src/rtl_fm.c:104-122
  struct demod_state    /root/reference/src/rtl_fm.c:124-159   (1 049 160 bytes)
  struct tuning_state   /root/reference/src/rtl_power.c:89-108
The C declarations the library itself is compiled against are in
include/rxgpu_ref_structs.h; tests check that both agree with the sizes/offsets the
reference's own translation unit reports (oracle/_ref).
"""
import ctypes as C

MAXIMUM_BUF_LENGTH = 16 * 16384          # rtl_fm.c:80-82

# glibc x86-64 opaque pthread object sizes
_PTHREAD_T = C.c_ulong
_RWLOCK = C.c_byte * 56
_COND = C.c_byte * 48
_MUTEX = C.c_byte * 40


class DemodState(C.Structure):
    pass


class DongleState(C.Structure):
    _fields_ = [
        ("exit_flag", C.c_int),
        ("thread", _PTHREAD_T),
        ("dev", C.c_void_p),
        ("stream", C.c_void_p),
        ("channel", C.c_size_t),
        ("dev_query", C.c_char_p),
        ("freq", C.c_uint32),
        ("rate", C.c_uint32),
        ("bandwidth", C.c_uint32),
        ("gain_str", C.c_char_p),
        ("buf16", C.c_int16 * MAXIMUM_BUF_LENGTH),
        ("ppm_error", C.c_int),
        ("offset_tuning", C.c_int),
        ("direct_sampling", C.c_int),
        ("mute", C.c_int),
        ("demod_target", C.POINTER(DemodState)),
    ]


DemodState._fields_ = [
    ("exit_flag", C.c_int),
    ("thread", _PTHREAD_T),
    ("lowpassed", C.c_int16 * MAXIMUM_BUF_LENGTH),
    ("lp_len", C.c_int),
    ("lp_i_hist", (C.c_int16 * 6) * 10),
    ("lp_q_hist", (C.c_int16 * 6) * 10),
    ("result", C.c_int16 * MAXIMUM_BUF_LENGTH),
    ("droop_i_hist", C.c_int16 * 9),
    ("droop_q_hist", C.c_int16 * 9),
    ("result_len", C.c_int),
    ("rate_in", C.c_int),
    ("rate_out", C.c_int),
    ("rate_out2", C.c_int),
    ("now_r", C.c_int),
    ("now_j", C.c_int),
    ("pre_r", C.c_int),
    ("pre_j", C.c_int),
    ("prev_index", C.c_int),
    ("downsample", C.c_int),
    ("post_downsample", C.c_int),
    ("output_scale", C.c_int),
    ("squelch_level", C.c_int),
    ("conseq_squelch", C.c_int),
    ("squelch_hits", C.c_int),
    ("terminate_on_squelch", C.c_int),
    ("squelch_zero", C.c_int),
    ("downsample_passes", C.c_int),
    ("comp_fir_size", C.c_int),
    ("custom_atan", C.c_int),
    ("deemph", C.c_int),
    ("deemph_a", C.c_int),
    ("now_lpr", C.c_int),
    ("prev_lpr_index", C.c_int),
    ("dc_block_audio", C.c_int),
    ("dc_avg", C.c_int),
    ("adc_block_const", C.c_int),
    ("dc_block_raw", C.c_int),
    ("dc_avgI", C.c_int),
    ("dc_avgQ", C.c_int),
    ("rdc_block_const", C.c_int),
    ("mode_demod", C.c_void_p),
    ("rw", _RWLOCK),
    ("ready", _COND),
    ("ready_m", _MUTEX),
    ("output_target", C.c_void_p),
]


class TuningState(C.Structure):
    _fields_ = [
        ("freq", C.c_int64),
        ("rate", C.c_int),
        ("bin_e", C.c_int),
        ("avg", C.POINTER(C.c_int64)),
        ("samples", C.c_int),
        ("downsample", C.c_int),
        ("downsample_passes", C.c_int),
        ("crop", C.c_double),
        ("buf16", C.POINTER(C.c_int16)),
        ("buf_len", C.c_int),
    ]


SIZEOF_DEMOD_STATE = 1049160     # SURVEY.md section 8(a) F-T, re-checked against oracle/_ref in tests
assert C.sizeof(DemodState) == SIZEOF_DEMOD_STATE, C.sizeof(DemodState)
