"""ctypes binding of include/gmat_hip.h.

``load()`` returns the product library (hipcc build).  ``load(path)`` binds any other build of the
same ABI; the tests use that for the CPU-emulated build under tests/hipemu (test infrastructure).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))

PIX_FMT = dict(yuv420p=0, rgb24=2, bgr24=3, yuv444p=5, nv12=23, yuv444p16le=49, yuv420p16le=45, yuv420p10le=62, rgba=26, bgra=28, rgba64le=105, bgra64le=107, hip=117, rgb0=119, bgr0=121, p010le=159, p016le=170, rgbpf32le=179)
SWS = dict(fast_bilinear=1, bilinear=2, bicubic=4, x=8, point=0x10, area=0x20, bicublin=0x40, gauss=0x80, sinc=0x100,
           lanczos=0x200, spline=0x400, full_chr_h_int=0x2000, full_chr_h_inp=0x4000, accurate_rnd=0x40000,
           bitexact=0x80000, hwaccel=0x1000000)


class GmatError(RuntimeError):
    pass


class GmatFrame(C.Structure):
    _fields_ = [("data", C.c_void_p * 4), ("linesize", C.c_int * 4), ("width", C.c_int), ("height", C.c_int),
                ("format", C.c_int), ("sw_format", C.c_int), ("pts", C.c_int64), ("colorspace", C.c_int),
                ("hw_frames_ctx", C.c_void_p), ("buf", C.c_void_p)]


def lib_path():
    return os.path.join(_HERE, "lib", "libgmat_hip.so")


_P4 = C.c_void_p * 4
_I4 = C.c_int * 4

_SIGS = {
    # name: (restype, argtypes)
    "gmat_sws_getContext": (C.c_void_p, [C.c_int] * 7 + [C.POINTER(C.c_double)]),
    "gmat_sws_scale": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.c_int, C.c_int,
                                 C.POINTER(C.c_void_p), C.POINTER(C.c_int)]),
    "gmat_sws_setStream": (None, [C.c_void_p, C.c_void_p]),
    "gmat_sws_freeContext": (None, [C.c_void_p]),
    "gmat_sws_setColorspace": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "gmat_sws_setRange": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "gmat_sws_setChromaPos": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]),
    "gmat_sws_setFused": (C.c_int, [C.c_void_p, C.c_int]),
    "gmat_sws_getFilter": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int)]),
    "gmat_sws_lastKernel": (C.c_char_p, [C.c_void_p]),
    "gmat_sws_lastLaunchFrames": (C.c_int, [C.c_void_p]),
    "gmat_sws_streamHandoffs": (C.c_int, [C.c_void_p]),
    "gmat_sws_setProfileBuffer": (C.c_int, [C.c_void_p, C.c_void_p]),
    "yuv2rgb_cuda": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.POINTER(C.c_void_p), C.POINTER(C.c_int),
                               C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "rgb2yuv_cuda": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.POINTER(C.c_void_p), C.POINTER(C.c_int),
                               C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "yuv2yuv_cuda": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.POINTER(C.c_void_p), C.POINTER(C.c_int),
                               C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "rgb24tobgr24_cuda": (None, [C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.POINTER(C.c_int),
                                 C.c_int, C.c_int, C.c_void_p]),
    "rgb2rgb_init_cuda": (None, []),
    "SwscaleCuda_Nv12ToRgbpf32_Init": (C.c_void_p, [C.c_int, C.c_int]),
    "SwscaleCuda_Nv12ToRgbpf32_Convert": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int,
                                                    C.c_int, C.c_void_p]),
    "SwscaleCuda_Nv12ToRgbpf32_Delete": (None, [C.c_void_p]),
    "gmat_hwframe_ctx_create": (C.c_void_p, [C.c_int] * 5),
    "gmat_hwframe_ctx_free": (None, [C.c_void_p]),
    "gmat_hwframe_ctx_info": (C.c_int, [C.c_void_p] + [C.POINTER(C.c_int)] * 4),
    "gmat_hwframe_get_buffer": (C.c_int, [C.c_void_p, C.POINTER(GmatFrame)]),
    "gmat_frame_alloc": (C.POINTER(GmatFrame), []),
    "gmat_frame_free": (None, [C.POINTER(C.POINTER(GmatFrame))]),
    "gmat_frame_unref": (None, [C.POINTER(GmatFrame)]),
    "gmat_hwframe_transfer_data": (C.c_int, [C.POINTER(GmatFrame), C.POINTER(GmatFrame), C.c_void_p]),
    "gmat_host_frame_alloc": (C.c_int, [C.POINTER(GmatFrame), C.c_int, C.c_int, C.c_int]),
    "gmat_host_frame_free": (None, [C.POINTER(GmatFrame)]),
    "gmat_filter_alloc": (C.c_void_p, [C.c_char_p]),
    "gmat_filter_set_option": (C.c_int, [C.c_void_p, C.c_char_p, C.c_char_p]),
    "gmat_filter_init": (C.c_int, [C.c_void_p]),
    "gmat_filter_config_props": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "gmat_filter_out_frames": (C.c_void_p, [C.c_void_p]),
    "gmat_filter_frame": (C.c_int, [C.c_void_p, C.POINTER(GmatFrame), C.POINTER(C.POINTER(GmatFrame))]),
    "gmat_filter_free": (None, [C.c_void_p]),
    "gmat_filter_send_frame": (C.c_int, [C.c_void_p, C.POINTER(GmatFrame)]),
    "gmat_filter_receive_frame": (C.c_int, [C.c_void_p, C.POINTER(C.POINTER(GmatFrame))]),
    "gmat_filter_flush": (C.c_int, [C.c_void_p]),
    "gmat_transpose": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "gmat_flip": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "gmat_crop": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "gmat_gauss_blur": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                  C.c_double, C.c_double, C.c_int, C.c_void_p]),
    "gmat_median3x3": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "gmat_smooth3x3": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                 C.POINTER(C.c_int), C.c_float, C.c_float, C.c_void_p]),
    "gmat_rotate": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                              C.c_double, C.c_int, C.c_void_p, C.c_void_p]),
    "gmat_rotate2": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int,
                               C.c_double, C.c_double, C.c_void_p, C.c_void_p]),
    "gmat_rotate_shift_translation": (None, [C.c_double, C.c_double, C.c_double, C.c_int, C.c_int, C.c_int, C.c_int,
                                             C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "gmat_median": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "gmat_rotate_flip_smooth": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "gmat_op_batch": (C.c_int, [C.c_int, C.c_int, C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "gmat_rotate2_batch": (C.c_int, [C.c_int, C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                     C.c_double, C.c_int, C.c_double, C.c_double, C.c_void_p, C.c_void_p]),
    "gmat_set_log_callback": (None, [C.c_void_p]),
    "gmat_knobs_reload": (None, []),
    "gmat_device_count": (C.c_int, []),
    "gmat_set_device": (C.c_int, [C.c_int]),
    "gmat_device_numa_node": (C.c_int, [C.c_int]),
    "gmat_device_compute_units": (C.c_int, [C.c_int]),
    "gmat_bind_thread_to_device": (C.c_int, [C.c_int]),
    "gmat_version": (C.c_char_p, []),
    "gmat_malloc": (C.c_int, [C.POINTER(C.c_void_p), C.c_size_t]),
    "gmat_free": (C.c_int, [C.c_void_p]),
    "gmat_memcpy_h2d": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "gmat_memcpy_d2h": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "gmat_memset": (C.c_int, [C.c_void_p, C.c_int, C.c_size_t]),
    "gmat_stream_create": (C.c_int, [C.POINTER(C.c_void_p)]),
    "gmat_stream_destroy": (C.c_int, [C.c_void_p]),
    "gmat_stream_sync": (C.c_int, [C.c_void_p]),
    "gmat_device_sync": (C.c_int, []),
    "gmat_event_create": (C.c_int, [C.POINTER(C.c_void_p)]),
    "gmat_event_record": (C.c_int, [C.c_void_p, C.c_void_p]),
    "gmat_stream_wait_event": (C.c_int, [C.c_void_p, C.c_void_p]),
    "gmat_event_sync": (C.c_int, [C.c_void_p]),
    "gmat_event_destroy": (None, [C.c_void_p]),
    "gmat_timer_create": (C.c_int, [C.POINTER(C.c_void_p)]),
    "gmat_timer_begin": (C.c_int, [C.c_void_p, C.c_void_p]),
    "gmat_timer_end": (C.c_int, [C.c_void_p, C.c_void_p]),
    "gmat_timer_elapsed_ms": (C.c_int, [C.c_void_p, C.POINTER(C.c_float)]),
    "gmat_timer_destroy": (None, [C.c_void_p]),
    "gmat_sws_graph_create": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_int),
                                        C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]),
    "gmat_sws_scale_batch": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_int),
                                       C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.POINTER(C.c_void_p), C.c_int, C.c_int]),
    "gmat_pipeline_create": (C.c_void_p, [C.c_int] * 9),
    "gmat_pipeline_host_input": (C.c_int, [C.c_void_p, C.c_int64, C.POINTER(GmatFrame)]),
    "gmat_pipeline_host_output": (C.c_int, [C.c_void_p, C.c_int64, C.POINTER(GmatFrame)]),
    "gmat_pipeline_submit": (C.c_int64, [C.c_void_p]),
    "gmat_pipeline_wait": (C.c_int, [C.c_void_p, C.c_int64]),
    "gmat_pipeline_drain": (C.c_int, [C.c_void_p]),
    "gmat_pipeline_free": (None, [C.c_void_p]),
    "gmat_graph_launch": (C.c_int, [C.c_void_p, C.c_void_p]),
    "gmat_graph_destroy": (None, [C.c_void_p]),
}

ABI_SYMBOLS = tuple(_SIGS)


def load(path=None):
    """Load a build of the gmat_hip ABI and attach prototypes.  Raises GmatError when missing."""
    path = path or lib_path()
    if not os.path.exists(path):
        raise GmatError(
            f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
    try:
        lib = C.CDLL(path)
    except OSError as e:  # pragma: no cover
        raise GmatError(f"cannot load {path}: {e}") from e
    for name, (res, args) in _SIGS.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise GmatError(f"{path} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    return lib


def planes(ptrs):
    a = (C.c_void_p * 4)()
    for i, p in enumerate(ptrs):
        a[i] = p
    return C.cast(a, C.POINTER(C.c_void_p))


def ints(vals):
    a = (C.c_int * 4)()
    for i, v in enumerate(vals):
        a[i] = v
    return C.cast(a, C.POINTER(C.c_int))
