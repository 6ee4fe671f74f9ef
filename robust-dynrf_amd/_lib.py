"""ctypes binding of librodynrf.so (C ABI declared in include/rodynrf.h).

The library is the product: there is NO fallback.  If it is missing or fails to load the import
raises, and every op raises if the tensors are not on a HIP device.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# RDRF_DETERMINISTIC=1: the debugging twin (fixed-point gradient accumulation, bit-reproducible gradients; csrc/Makefile)
DETERMINISTIC = os.environ.get("RDRF_DETERMINISTIC", "0") == "1"
LIB_PATH = os.environ.get("RDRF_LIB", os.path.join(_HERE, "librodynrf_det.so" if DETERMINISTIC else "librodynrf.so"))  # RDRF_LIB: A/B builds

ABI_VERSION = 6   # include/rodynrf.h RDRF_ABI_VERSION: the parameter structs below are read to their full length

RAY_TYPES = {"ndc": 0, "contract": 1}
ACTS = {"relu": 0, "softplus": 1}
HEADS = {"MLP_Fea": 0, "MLP_Fea_TimeEmbedding": 1}

fp = C.POINTER(C.c_float)


class RdrfVM(C.Structure):
    _fields_ = [("plane", C.c_void_p * 3), ("line", C.c_void_p * 3), ("C", C.c_int * 3),
                ("H", C.c_int * 3), ("W", C.c_int * 3), ("L", C.c_int * 3),
                ("sH", C.c_int * 3), ("sW", C.c_int * 3)]


class RdrfFieldCfg(C.Structure):
    _fields_ = [("aabb", C.c_float * 6), ("distance_scale", C.c_float), ("weight_thres", C.c_float),
                ("density_shift", C.c_float), ("act", C.c_int), ("ray_type", C.c_int),
                ("static_head", C.c_int)]


class RdrfStaticParams(C.Structure):
    _fields_ = [("density", RdrfVM), ("app", RdrfVM), ("basis", C.c_void_p),
                ("w1", C.c_void_p), ("b1", C.c_void_p), ("w2", C.c_void_p), ("b2", C.c_void_p),
                ("w3", C.c_void_p), ("b3", C.c_void_p), ("packed_fwd", C.c_void_p), ("packed_bwd", C.c_void_p)]


class RdrfDynamicParams(C.Structure):
    _fields_ = [("density", RdrfVM), ("blending", RdrfVM), ("app", RdrfVM), ("basis", C.c_void_p),
                ("rw1", C.c_void_p), ("rb1", C.c_void_p), ("rw2", C.c_void_p), ("rb2", C.c_void_p),
                ("rwv", C.c_void_p), ("rbv", C.c_void_p),
                ("l1w", C.c_void_p), ("l1b", C.c_void_p), ("l2w", C.c_void_p), ("l2b", C.c_void_p),
                ("l3w", C.c_void_p), ("l3b", C.c_void_p), ("l4w", C.c_void_p), ("l4b", C.c_void_p),
                ("l5w", C.c_void_p), ("l5b", C.c_void_p),
                ("dw1", C.c_void_p), ("db1", C.c_void_p), ("dw2", C.c_void_p), ("db2", C.c_void_p),
                ("bw1", C.c_void_p), ("bb1", C.c_void_p), ("bw2", C.c_void_p), ("bb2", C.c_void_p),
                ("sfw", C.c_void_p * 4), ("sfb", C.c_void_p * 4), ("packed_fwd", C.c_void_p), ("packed_bwd", C.c_void_p)]


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build the HIP library first "
            "(python -c 'import __graft_entry__ as g; g.build()' or make -C robust-dynrf_amd/csrc)")
    lib = C.CDLL(LIB_PATH)
    # the version check comes FIRST: a library of another version may lack symbols bound below, and the caller should see
    # "version mismatch", not a raw AttributeError
    if not hasattr(lib, "rdrf_abi_version") or lib.rdrf_abi_version() != ABI_VERSION:
        got = lib.rdrf_abi_version() if hasattr(lib, "rdrf_abi_version") else None
        raise ImportError(f"{LIB_PATH}: ABI version {got}, this binding is built against {ABI_VERSION} (include/rodynrf.h): rebuild "
                          "with make -C robust-dynrf_amd/csrc")
    lib.rdrf_last_error.restype = C.c_char_p
    lib.rdrf_workspace_bytes.restype = C.c_size_t
    lib.rdrf_workspace_bytes.argtypes = [C.c_int, C.c_int]
    lib.rdrf_forward_workspace_bytes.restype = C.c_size_t
    lib.rdrf_forward_workspace_bytes.argtypes = [C.c_int, C.c_int]
    lib.rdrf_saved_bytes.restype = C.c_size_t
    lib.rdrf_saved_bytes.argtypes = [C.c_int, C.c_int, C.c_int]
    lib.rdrf_saved_row_bytes.restype = C.c_size_t
    lib.rdrf_saved_row_bytes.argtypes = [C.c_int]
    for name, args in (("rdrf_features_saved_bytes", [C.c_int, C.c_int]),
                       ("rdrf_features_workspace_bytes", [C.c_int]),
                       ("rdrf_features_bwd_workspace_bytes", [C.c_int])):
        getattr(lib, name).restype = C.c_size_t
        getattr(lib, name).argtypes = args
    lib.rdrf_adam_step.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t,
                                   C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int, C.c_float,
                                   C.c_void_p]
    lib.rdrf_pack_floats.restype = C.c_size_t
    lib.rdrf_pack_floats.argtypes = []
    lib.rdrf_loss_terms_workspace_floats.restype = C.c_size_t
    lib.rdrf_loss_terms_workspace_floats.argtypes = [C.c_int]
    lib.rdrf_frame_depth_loss_workspace_bytes.restype = C.c_size_t
    lib.rdrf_frame_depth_loss_workspace_bytes.argtypes = [C.c_int, C.c_int]
    lib.rdrf_frame_depth_loss_fwd.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float,
                                              C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    lib.rdrf_frame_depth_loss_bwd.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_void_p, C.c_void_p]
    lib.rdrf_render_workspace_bytes.restype = C.c_size_t
    lib.rdrf_render_workspace_bytes.argtypes = [C.c_int, C.c_int]
    lib.rdrf_render_chunks_workspace_bytes.restype = C.c_size_t
    lib.rdrf_render_chunks_workspace_bytes.argtypes = [C.c_int, C.c_int, C.c_int]
    lib.rdrf_prof_get.argtypes = [C.c_char_p, C.POINTER(C.c_double), C.POINTER(C.c_int)]
    lib.rdrf_det_bind.argtypes = [C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
    lib.rdrf_det_finish.argtypes = [C.c_int, C.c_void_p]
    if DETERMINISTIC and not lib.rdrf_deterministic():
        raise ImportError(f"RDRF_DETERMINISTIC=1 but {LIB_PATH} is the product build")
    return lib


lib = _load()

# every symbol include/rodynrf.h declares (tests check that the library exports all of them)
SYMBOLS = [
    "rdrf_abi_version", "rdrf_last_error", "rdrf_workspace_bytes", "rdrf_forward_workspace_bytes", "rdrf_saved_bytes", "rdrf_saved_row_bytes",
    "rdrf_generate_rays",
    "rdrf_generate_rays_bwd", "rdrf_generate_rays_uv", "rdrf_generate_rays_uv_bwd", "rdrf_sample_ndc", "rdrf_sample_contract", "rdrf_sample_bwd",
    "rdrf_static_fwd", "rdrf_static_bwd", "rdrf_dynamic_fwd", "rdrf_dynamic_bwd",
    "rdrf_features_saved_bytes", "rdrf_features_workspace_bytes", "rdrf_features_bwd_workspace_bytes",
    "rdrf_static_features_fwd", "rdrf_static_features_bwd", "rdrf_dynamic_features_fwd",
    "rdrf_dynamic_features_bwd",
    "rdrf_scene_flow_fwd", "rdrf_scene_flow_bwd", "rdrf_composite_fwd", "rdrf_composite_bwd",
    "rdrf_induce_flow_fwd", "rdrf_induce_flow_bwd", "rdrf_rows_scatter_add", "rdrf_distloss_fwd", "rdrf_distloss_bwd",
    "rdrf_tv_fwd", "rdrf_tv_bwd", "rdrf_tv_grad", "rdrf_adam_step", "rdrf_upsample_bilinear", "rdrf_dense_l1_fwd",
    "rdrf_dense_l1_bwd", "rdrf_pack_floats", "rdrf_static_pack", "rdrf_dynamic_pack", "rdrf_loss_terms_workspace_floats", "rdrf_loss_terms_fwd", "rdrf_loss_terms_bwd",
    "rdrf_loss_terms_stats", "rdrf_loss_terms_finish", "rdrf_deterministic", "rdrf_det_bind", "rdrf_det_finish",
    "rdrf_render_fused_fwd", "rdrf_render_sequence_fwd",
    "rdrf_frame_depth_loss_workspace_bytes", "rdrf_frame_depth_loss_fwd", "rdrf_frame_depth_loss_bwd",
    "rdrf_render_workspace_bytes", "rdrf_render_fwd", "rdrf_render_chunks_workspace_bytes", "rdrf_render_chunks_fwd",
    "rdrf_set_scatter_mode", "rdrf_selftest_mlp", "rdrf_prof_reset",
    "rdrf_prof_enable", "rdrf_prof_get",
]


SCATTER_MODES = {"ray": 0, "sorted": 1, "auto": 2, "sorted_plain": 3}


def set_scatter_mode(mode):
    """rdrf_set_scatter_mode: "auto" (default: sorted from 300 k samples per launch) | "ray" | "sorted" | "sorted_plain"
    (sorted without the LDS plane windows of k_scatter_tiled) """
    check(lib.rdrf_set_scatter_mode(SCATTER_MODES[mode]), "rdrf_set_scatter_mode")


class RdrfTensor4(C.Structure):
    _fields_ = [("x", C.c_void_p), ("g", C.c_void_p), ("C", C.c_int), ("H", C.c_int), ("W", C.c_int),
                ("sC", C.c_longlong), ("sH", C.c_longlong), ("sW", C.c_longlong)]


TV_MAX = 16


class RdrfLossTerm(C.Structure):
    _fields_ = [("x", C.c_void_p), ("y", C.c_void_p), ("w", C.c_void_p), ("gx", C.c_void_p), ("gy", C.c_void_p),
                ("rows", C.c_longlong), ("cols", C.c_int), ("kind", C.c_int), ("norm", C.c_int),
                ("ysign", C.c_float), ("coef", C.c_float), ("coef_dev", C.c_void_p)]


MAX_LOSS_TERMS = 32


class RdrfError(RuntimeError):
    pass


def check(rc, what):
    if rc != 0:
        raise RdrfError(f"{what} failed (rc={rc}): {lib.rdrf_last_error().decode()}")


def require_device(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RdrfError("rodynrf ops run on the MI355X only: got a CPU tensor "
                            "(there is no CPU fallback; the oracle lives under oracle/ for tests)")


def ptr(t):
    return C.c_void_p(0 if t is None else t.data_ptr())


def stream_of(t):
    return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


_ws = {}          # (device, stream) -> scratch buffer, most recently used last
_WS_MAX = 20      # the default stream + the 16 side streams of render_chunks + slack: older entries are dropped


def workspace(device, nbytes):
    """scratch of the C entry points, one buffer per (device, stream): calls on different streams (render_chunks) must
    not share scratch.  The cache is an LRU of _WS_MAX entries (a stream that is no longer used gives its buffer back to
    the caching allocator, which keeps it ordered behind that stream's work); growing a buffer drops the old one first."""
    key = (device, torch.cuda.current_stream(device).cuda_stream)
    buf = _ws.pop(key, None)
    if buf is None or buf.numel() < nbytes:
        buf = None   # free before allocating: the peak is one buffer, not two
        buf = torch.empty(int(nbytes * 1.25) + 1024, dtype=torch.uint8, device=device)
    _ws[key] = buf
    while len(_ws) > _WS_MAX:
        _ws.pop(next(iter(_ws)))
    return buf


def release_workspaces():
    """drop every cached scratch buffer (they return to torch's caching allocator)"""
    _ws.clear()


def zeros_like_many(tensors, want=None):
    """torch.zeros_like for several tensors with ONE fill launch: contiguous fp32 views (256-byte aligned) of a
    single zeroed buffer; entries that are None (or whose `want` flag is false) come back as None."""
    want = [t is not None for t in tensors] if want is None else [bool(w) and t is not None for t, w in zip(tensors, want)]
    sizes = [((t.numel() + 63) // 64) * 64 if w else 0 for t, w in zip(tensors, want)]
    total = sum(sizes)
    if total == 0:
        return [None] * len(tensors)
    dev = next(t.device for t, w in zip(tensors, want) if w)
    flat = torch.zeros(total, dtype=torch.float32, device=dev)
    out, o = [], 0
    for t, w, n in zip(tensors, want, sizes):
        out.append(flat[o: o + t.numel()].view(t.shape) if w else None)
        o += n
    return out


def f32c(t):
    """contiguous fp32 view (no copy when already so)"""
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()
