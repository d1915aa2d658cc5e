"""Loader for libdfgnn.so -- the C-ABI library of the HIP kernels (include/dfgnn.h).

libdfgnn.so is a forwarder without any HIP dependency (csrc/gen_shim.py): it opens the kernels library libdfgnn_hip.so
next to it at the first call that needs it, so the order in which a process loads this library and torch does not matter.
There is deliberately NO fallback: if either shared library is missing or a symbol is absent this module (or the first
operator call) raises, so a GPU box can never silently run a non-HIP path.
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
# DFGNN_LIB=<file name>: an A/B build of the library next to the shipped one (tools/diag/build_variant.sh)
LIB_PATH = os.path.join(_HERE, os.environ.get("DFGNN_LIB", "libdfgnn.so"))
CSRC = os.path.join(_HERE, "csrc")

_vp, _i, _f = ctypes.c_void_p, ctypes.c_int, ctypes.c_float

# symbol -> argtypes; mirrors include/dfgnn.h one to one (checked by tests/test_capi_symbols.py)
SIGNATURES = {
    "dfgnn_plan_build": [_i, _i, _i] + [_vp] * 5,
    "dfgnn_preprocess_hyper": [_i, _i, _vp, _vp, _i] + [_vp] * 8 + [ctypes.c_size_t, _vp],
    "dfgnn_gt_hyper_fwd": [_i, _i, _i, _i] + [_vp] * 13,
    "dfgnn_gt_bwd": [_i, _i, _i, _i] + [_vp] * 19,
    "dfgnn_gt_stats_applies": [_i, _i, _i, _i, _vp],
    "dfgnn_gt_hyper_fwd_stats": [_i, _i, _i, _i] + [_vp] * 12,
    "dfgnn_gt_bwd_stats": [_i, _i, _i, _i] + [_vp] * 15,
    "dfgnn_plan_dense_weights": [_i, _i] + [_vp] * 6,
    "dfgnn_gt_hyper_fwd_ranked": [_i, _i, _i, _i] + [_vp] * 10,
    "dfgnn_gt_bwd_ranked": [_i, _i, _i, _i] + [_vp] * 13,
    "dfgnn_gt_bwd_rows": [_i, _i, _i, _i] + [_vp] * 11,
    "dfgnn_gt_bwd_cols": [_i, _i, _i, _i] + [_vp] * 11,
    "dfgnn_gt_tiling_fwd": [_i, _i, _i, _i] + [_vp] * 8,
    "dfgnn_gt_csr_fwd": [_i, _i, _i, _i] + [_vp] * 9,
    "dfgnn_gt_csr_gm_fwd": [_i, _i, _i, _i] + [_vp] * 9,
    "dfgnn_gat_recompute_fwd": [_i, _i, _i, _i] + [_vp] * 4 + [_f] + [_vp] * 3,
    "dfgnn_gt_softmax_fwd": [_i, _i, _i, _i] + [_vp] * 10,
    "dfgnn_gt_softmax_gm_fwd": [_i, _i, _i, _i] + [_vp] * 10,
    "dfgnn_gat_hyper_fwd": [_i, _i, _i, _i] + [_vp] * 5 + [_f] + [_vp] * 6,
    "dfgnn_gat_softmax_fwd": [_i, _i, _i, _i] + [_vp] * 5 + [_f] + [_vp] * 4,
    "dfgnn_gat_softmax_gm_fwd": [_i, _i, _i, _i] + [_vp] * 5 + [_f] + [_vp] * 4,
    "dfgnn_gat_tiling_fwd": [_i, _i, _i, _i] + [_vp] * 4 + [_f] + [_vp] * 3,
    "dfgnn_gat_tiling_chunked_fwd": [_i, _i, _i, _i, _i] + [_vp] * 4 + [_f] + [_vp] * 3 + [ctypes.c_size_t, _vp],
    "dfgnn_gat_attn_scores": [_i, _i, _i] + [_vp] * 6,
    "dfgnn_gat_fwd_train": [_i, _i, _i, _i] + [_vp] * 5 + [_f] + [_vp] * 2 + [_f] + [_vp] * 6,
    "dfgnn_gat_bwd": [_i, _i, _i, _i] + [_vp] * 8 + [_f] + [_vp] * 4 + [_f] + [_vp] * 8,
}

EXT_PATH = os.path.join(_HERE, "_dfgnn_ext.so")   # torch C++ extension over the same C ABI (csrc/torch_ext.cpp)

_lib = None
_ext = False


def source_hash():
    """sha256 (16 hex digits) over the library's sources in the Makefile's order; equals dfgnn_build_id() of a library
    built from exactly these files."""
    import hashlib
    import re
    mk = open(os.path.join(CSRC, "Makefile")).read()
    names = []
    for var in ("SRCS", "HDRS", "EXTRA"):
        names += re.search(rf"^{var}\s*:=\s*(.*)$", mk, flags=re.M).group(1).split()
    h = hashlib.sha256()
    for n in names + ["Makefile"]:
        h.update(open(os.path.join(CSRC, n), "rb").read())
    return h.hexdigest()[:16]


def build_id(path=LIB_PATH):
    """dfgnn_build_id() of a built library: answered by the forwarder itself, no HIP runtime is loaded (round 2 had to
    import torch first here: the library then linked libamdhip64 and, opened ahead of torch, brought /opt/rocm's runtime
    into the process next to torch's own)."""
    L = ctypes.CDLL(path)
    L.dfgnn_build_id.restype = ctypes.c_char_p
    return L.dfgnn_build_id().decode()


def build(force=False, jobs=8):
    """Compile libdfgnn.so for gfx950 with hipcc (recipe: csrc/Makefile; incremental).  Raises if the result does not
    carry the hash of the sources on disk (a stale library whose ABI number still matches)."""
    cmd = ["make", "-C", CSRC, "-j", str(jobs)]
    if force:
        cmd.append("-B")
    subprocess.check_call(cmd, stdout=subprocess.DEVNULL)
    if build_id() != source_hash():
        raise RuntimeError(f"{LIB_PATH} is stale: built from sources {build_id()}, on disk {source_hash()}")
    return LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback for the fused kernels.")
        # (no import order to respect: libdfgnn.so has no HIP dependency, libdfgnn_hip.so is opened by it at the first
        # call that needs the kernels -- csrc/gen_shim.py)
        L = ctypes.CDLL(LIB_PATH)
        for name, argtypes in SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError if the library is stale
            fn.argtypes = argtypes
            fn.restype = ctypes.c_int
        L.dfgnn_error_string.argtypes = [ctypes.c_int]
        L.dfgnn_error_string.restype = ctypes.c_char_p
        L.dfgnn_abi_version.restype = ctypes.c_int
        L.dfgnn_build_id.restype = ctypes.c_char_p
        L.dfgnn_plan_ints.argtypes = [ctypes.c_int, ctypes.c_int]
        L.dfgnn_plan_ints.restype = ctypes.c_size_t
        L.dfgnn_plan_applies.argtypes = [ctypes.c_int] * 4 + [ctypes.c_void_p]
        L.dfgnn_plan_applies.restype = ctypes.c_int
        L.dfgnn_plan_dense_weights_floats.argtypes = [ctypes.c_int]
        L.dfgnn_plan_dense_weights_floats.restype = ctypes.c_size_t
        L.dfgnn_preprocess_ws_bytes.argtypes = [ctypes.c_int, ctypes.c_int]
        L.dfgnn_preprocess_ws_bytes.restype = ctypes.c_size_t
        L.dfgnn_gat_tiling_chunked_ws_bytes.argtypes = [ctypes.c_int] * 4
        L.dfgnn_gat_tiling_chunked_ws_bytes.restype = ctypes.c_size_t
        _lib = L
    return _lib


def ext():
    """The torch C++ extension (csrc/torch_ext.cpp -> _dfgnn_ext.so, built in-tree by build()): the reference-style
    pybind11 binding over the same C ABI, ~5 us of host time per operator call instead of ~25-60 for ctypes.  None when
    it is absent, fails to import, was compiled for another library build (its baked-in ABI number and source hash are
    compared with the library's), or DFGNN_BINDING=ctypes asks for the ctypes path (tests)."""
    global _ext
    if _ext is False:
        _ext = None
        if os.environ.get("DFGNN_BINDING", "ext") != "ctypes" and os.path.exists(EXT_PATH) and \
                os.path.basename(LIB_PATH) == "libdfgnn.so":
            import importlib.util
            L = lib()  # libdfgnn.so first: the extension resolves its dfgnn_* symbols against it
            try:
                import torch  # noqa: F401  (the extension links libtorch)
                spec = importlib.util.spec_from_file_location("_dfgnn_ext", EXT_PATH)
                mod = importlib.util.module_from_spec(spec)
                spec.loader.exec_module(mod)
                # the extension's OWN compile-time constants against the library's: an extension left over from another
                # build (or built against another torch: ImportError) is not used, the ctypes path serves instead
                if mod.abi_version() == L.dfgnn_abi_version() and mod.build_id() == L.dfgnn_build_id().decode():
                    _ext = mod
            except (ImportError, OSError, AttributeError):
                _ext = None
    return _ext


def check(code, what):
    if code != 0:
        msg = lib().dfgnn_error_string(code).decode()
        raise RuntimeError(f"{what} failed: {msg} (code {code})")
