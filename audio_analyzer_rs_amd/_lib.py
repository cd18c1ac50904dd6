"""ctypes loader for libaa_stft.so (the C ABI declared in include/aa_stft.h).

The library is the product: if it is missing the package fails loudly — there is
no CPU or PyTorch fallback for any compute entry point.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libaa_stft.so")

AA_OK = 0
AA_ERR_ARG = -1
AA_ERR_LEN = -2
AA_ERR_HIP = -3
AA_ERR_NOMEM = -4
AA_ERR_UNSUPPORTED = -5
AA_ERR_NCCL = -6

AA_WINDOW_HANN_PERIODIC = 0
AA_WINDOW_RECT = 1
AA_WINDOW_CUSTOM = 2

AA_OUT_COMPLEX = 0
AA_OUT_MAG = 1
AA_OUT_POWER = 2
AA_OUT_LOGPOWER = 3
AA_OUT_LOGMEL = 4
AA_OUT_MFCC = 5

OUTPUT_KINDS = {
    "complex": AA_OUT_COMPLEX,
    "mag": AA_OUT_MAG,
    "power": AA_OUT_POWER,
    "logpower": AA_OUT_LOGPOWER,
    "logmel": AA_OUT_LOGMEL,
    "mfcc": AA_OUT_MFCC,
}
WINDOW_KINDS = {"hann": AA_WINDOW_HANN_PERIODIC, "rect": AA_WINDOW_RECT, "custom": AA_WINDOW_CUSTOM}


class AaError(RuntimeError):
    """A non-zero status from libaa_stft.so (the reference panics instead: fft.rs:69)."""

    def __init__(self, code, message):
        super().__init__(f"libaa_stft status {code}: {message}")
        self.code = code


class StftConfig(ctypes.Structure):
    """Mirror of `aa_stft_config` (include/aa_stft.h)."""

    _fields_ = [
        ("n_fft", ctypes.c_int32),
        ("hop", ctypes.c_int32),
        ("window", ctypes.c_int32),
        ("output", ctypes.c_int32),
        ("sample_rate", ctypes.c_int32),
        ("n_mels", ctypes.c_int32),
        ("n_mfcc", ctypes.c_int32),
        ("fmin", ctypes.c_float),
        ("fmax", ctypes.c_float),
        ("log_floor", ctypes.c_float),
    ]


class TunerOutputStruct(ctypes.Structure):
    """Mirror of `aa_tuner_output` (include/aa_stft.h)."""

    _fields_ = [
        ("label", ctypes.c_char * 256),
        ("cents", ctypes.c_float),
        ("n_notes", ctypes.c_int32),
        ("notes", (ctypes.c_char * 8) * 32),
        ("accuracies", ctypes.c_float * 32),
        ("mode", ctypes.c_int32),
        ("system", ctypes.c_int32),
        ("base_freq", ctypes.c_float),
        ("beat_position", ctypes.c_double),
    ]


_f32p = ctypes.POINTER(ctypes.c_float)
_i64p = ctypes.POINTER(ctypes.c_int64)
_vpp = ctypes.POINTER(ctypes.c_void_p)

# name -> (restype, argtypes); this is also the export list the CPU test checks
# against include/aa_stft.h.
# aa_piece_fn (include/aa_stft.h): int (*)(void *user, int64_t frame_begin, int64_t frame_end, void *stream)
PIECE_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p)

SIGNATURES = {
    "aa_last_error": (ctypes.c_char_p, []),
    "aa_abi_version": (ctypes.c_int, []),
    "aa_device_count": (ctypes.c_int, [ctypes.POINTER(ctypes.c_int)]),
    "aa_hann_window": (ctypes.c_int, [ctypes.c_int, _f32p]),
    "aa_fft_create": (ctypes.c_int, [ctypes.c_int, _vpp]),
    "aa_fft_destroy": (ctypes.c_int, [ctypes.c_void_p]),
    "aa_fft_len": (ctypes.c_int, [ctypes.c_void_p]),
    "aa_fft_forward": (ctypes.c_int, [ctypes.c_void_p, _f32p, _f32p]),
    "aa_fft_inverse": (ctypes.c_int, [ctypes.c_void_p, _f32p, _f32p]),
    "aa_fft_forward_batch_device": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                                   ctypes.c_int64, ctypes.c_void_p]),
    "aa_fft_inverse_batch_device": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                                   ctypes.c_int64, ctypes.c_void_p]),
    "aa_stft_plan_create": (ctypes.c_int, [ctypes.POINTER(StftConfig), _f32p, _vpp]),
    "aa_stft_plan_destroy": (ctypes.c_int, [ctypes.c_void_p]),
    "aa_stft_num_frames": (ctypes.c_int64, [ctypes.c_void_p, ctypes.c_int64]),
    "aa_stft_out_cols": (ctypes.c_int32, [ctypes.c_void_p]),
    "aa_stft_kernel_name": (ctypes.c_char_p, [ctypes.c_void_p]),
    "aa_stft_mel_path_name": (ctypes.c_char_p, [ctypes.c_void_p]),
    "aa_stft_execute_device": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64,
                                              ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p,
                                              ctypes.c_void_p]),
    "aa_stft_execute_channels_device": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64,
                                                       ctypes.c_int64, ctypes.c_int32, ctypes.c_void_p,
                                                       ctypes.c_int64, ctypes.c_void_p]),
    "aa_stft_execute_host": (ctypes.c_int, [ctypes.c_void_p, _f32p, ctypes.c_int64, _f32p]),
    "aa_host_register": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_size_t]),
    "aa_host_unregister": (ctypes.c_int, [ctypes.c_void_p]),
    "aa_downmix_device": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32,
                                         ctypes.c_void_p, ctypes.c_void_p]),
    "aa_onset_flux_device": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_void_p,
                                            ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "aa_onset_gates_device": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_void_p,
                                             ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32,
                                             ctypes.c_void_p, ctypes.c_int32,
                                             ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                             ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "aa_tuner_pitches_device": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32,
                                               ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_void_p,
                                               ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "aa_onset_flux_batch_device": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64,
                                                  ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                                  ctypes.c_void_p]),
    "aa_onset_gates_batch_device": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64,
                                                   ctypes.c_int32] + [ctypes.c_void_p] * 4 + [ctypes.c_int32] +
                                    [ctypes.c_void_p] * 9),
    "aa_tuner_floor_batch_device": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64,
                                                   ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                                   ctypes.c_void_p, ctypes.c_void_p]),
    "aa_conditioner_create": (ctypes.c_int, [ctypes.c_float, ctypes.c_int32, ctypes.POINTER(ctypes.c_void_p)]),
    "aa_conditioner_destroy": (ctypes.c_int, [ctypes.c_void_p]),
    "aa_conditioner_process": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p,
                                              ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "aa_pitch_tracker_create": (ctypes.c_int, [ctypes.POINTER(ctypes.c_void_p)]),
    "aa_pitch_tracker_destroy": (ctypes.c_int, [ctypes.c_void_p]),
    "aa_pitch_tracker_process": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                                ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p,
                                                ctypes.c_void_p]),
    "aa_tuner_floor_device": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_void_p,
                                             ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p]),
    "aa_note_from_freq": (ctypes.c_int, [ctypes.c_float, ctypes.c_float, ctypes.c_char_p, _f32p]),
    "aa_note_to_freq": (ctypes.c_int, [ctypes.c_char_p, ctypes.c_float, ctypes.c_float, _f32p]),
    "aa_midi_from_freq": (ctypes.c_int, [ctypes.c_float, ctypes.c_float, ctypes.POINTER(ctypes.c_int32), _f32p]),
    "aa_midi_to_freq": (ctypes.c_int, [ctypes.c_int32, ctypes.c_float, ctypes.c_float, _f32p]),
    "aa_interval": (ctypes.c_int, [_f32p, ctypes.c_int32, ctypes.c_int32, ctypes.c_char_p, _f32p]),
    "aa_tuner_output_build": (ctypes.c_int, [_f32p, _f32p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                                             ctypes.c_float, ctypes.c_double, ctypes.POINTER(TunerOutputStruct)]),
    "aa_tuner_output_json": (ctypes.c_int, [ctypes.POINTER(TunerOutputStruct), ctypes.c_char_p, ctypes.c_char_p,
                                            ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t)]),
    "aa_shard_frames": (ctypes.c_int, [ctypes.c_int64, ctypes.c_int32, ctypes.c_int32, _i64p, _i64p]),
    "aa_shard_samples": (ctypes.c_int, [ctypes.c_int64, ctypes.c_int64, ctypes.c_int32,
                                        ctypes.c_int32, _i64p, _i64p]),
    "aa_comm_unique_id": (ctypes.c_int, [ctypes.c_void_p]),
    "aa_comm_create": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, _vpp]),
    "aa_comm_destroy": (ctypes.c_int, [ctypes.c_void_p]),
    "aa_comm_world": (ctypes.c_int32, [ctypes.c_void_p]),
    "aa_comm_rank": (ctypes.c_int32, [ctypes.c_void_p]),
    "aa_comm_backend": (ctypes.c_char_p, []),
    "aa_comm_barrier": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p]),
    "aa_scatter_samples": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32,
                                          ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p, ctypes.c_int64,
                                          ctypes.c_void_p]),
    "aa_gather_rows": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32,
                                      ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p]),
    "aa_shard_piece": (ctypes.c_int, [ctypes.c_int64, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32, _i64p, _i64p]),
    "aa_scatter_samples_piece": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32,
                                                ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p, ctypes.c_int64,
                                                ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p]),
    "aa_gather_rows_piece": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32,
                                            ctypes.c_int32, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32,
                                            ctypes.c_void_p]),
    "aa_sharded_pipeline": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64,
                                           ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                                           ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p,
                                           ctypes.c_int32, PIECE_FN, ctypes.c_void_p, ctypes.c_void_p]),
    "aa_stft_execute_sharded": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                               ctypes.c_int64, ctypes.c_int32, ctypes.c_void_p, ctypes.c_int64,
                                               ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p]),
    "aa_stream_create": (ctypes.c_int, [ctypes.POINTER(StftConfig), ctypes.c_int32, _vpp]),
    "aa_stream_destroy": (ctypes.c_int, [ctypes.c_void_p]),
    "aa_stream_push": (ctypes.c_int, [ctypes.c_void_p, _f32p, ctypes.c_int32, _f32p, _i64p,
                                      ctypes.c_int64, _i64p]),
    "aa_rows_alloc": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.POINTER(ctypes.c_void_p),
                                     ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float)]),
    "aa_rows_free": (ctypes.c_int, [ctypes.c_void_p]),
    "aa_diag_flux_div3_device": (ctypes.c_int, [ctypes.c_uint32, ctypes.c_uint64, ctypes.POINTER(ctypes.c_uint64)]),
    "aa_diag_stream_copy_device": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32,
                                                  ctypes.c_void_p]),
}

_lib = None


def load():
    """Load libaa_stft.so once; raise if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; "
            "g.build()'` or `make -C audio_analyzer_rs_amd/csrc`. There is no CPU fallback.")
    # One HIP runtime per process: PyTorch (used for device memory, streams and torch.distributed)
    # bundles its own libamdhip64; loaded after libaa_stft.so has pulled in the system one, the two
    # runtimes disagree about the devices ("no ROCm-capable device is detected").  Import torch
    # first so that the library binds to the runtime torch already loaded.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(status):
    if status != AA_OK:
        msg = load().aa_last_error()
        raise AaError(status, msg.decode("utf-8", "replace") if msg else "")
    return status


def f32_ptr(arr):
    return arr.ctypes.data_as(_f32p)
