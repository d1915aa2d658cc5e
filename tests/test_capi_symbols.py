"""CPU: libdfgnn.so loads and exports every symbol include/dfgnn.h declares (no compute calls)."""
import ctypes
import os
import re
import sys

from conftest import ROOT


def _declared():
    text = open(os.path.join(ROOT, "include", "dfgnn.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dfgnn_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported():
    import dfgnn_native
    assert os.path.exists(dfgnn_native.LIB_PATH), "libdfgnn.so missing: run __graft_entry__.build()"
    lib = ctypes.CDLL(dfgnn_native.LIB_PATH)
    names = _declared()
    assert len(names) >= 11
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/dfgnn.h but not exported"


def test_loader_signatures_cover_header():
    import dfgnn_native
    compute = [n for n in _declared() if n not in ("dfgnn_abi_version", "dfgnn_error_string", "dfgnn_build_id",
                                                       "dfgnn_plan_ints", "dfgnn_plan_applies", "dfgnn_preprocess_ws_bytes",
                                                       "dfgnn_gat_tiling_chunked_ws_bytes", "dfgnn_plan_dense_weights_floats")]
    assert sorted(dfgnn_native.SIGNATURES) == compute
    lib = dfgnn_native.lib()
    assert lib.dfgnn_abi_version() == 11
    assert b"bad argument" in lib.dfgnn_error_string(-1)
    assert b"unsupported" in lib.dfgnn_error_string(-2)
    assert lib.dfgnn_plan_ints(10, 40) >= 12 + 7 * 10 + 4 + 20 + 2 * 8 * 10  # header + lists + scratch (+ rocPRIM temporary storage) + edge coordinates + the two edge bitmaps


def test_plan_fallback_predicate():
    """dfgnn_plan_applies (host only): a plan is used only for the (m, nnz, f) it was built for, not for low-degree graphs,
    and not for feature matrices of 4 GiB or more -- the plan kernels address feature rows with 32-bit byte offsets, the
    general kernels they fall back to with size_t."""
    import ctypes
    import dfgnn_native
    lib = dfgnn_native.lib()
    budget = 160 * 1024 - 16 * 64 * 8 - 256      # kBlockLdsBudget (dfgnn_launch.hpp)

    def applies(m, nnz, h, f, meta):
        arr = (ctypes.c_int * 12)(*meta)
        return lib.dfgnn_plan_applies(m, nnz, h, f, ctypes.addressof(arr))

    m, nnz, f = 120000, 6000000, 128
    meta = [1000, 0, 186, 15000, m, nnz, f, budget, 0, 1000, 300, 900000]
    assert applies(m, nnz, 1, f, meta) == 1
    assert applies(m, nnz, 8, f, meta) == 1                     # 0.49 GB of features
    assert applies(m, nnz, 70, f, meta) == 0                    # 120000 x 70 x 128 x 4 B >= 4 GiB
    assert applies(m + 1, nnz, 1, f, meta) == 0                 # another graph
    assert applies(m, nnz, 1, 64, meta) == 0                    # another width
    assert applies(m, nnz, 1, f, [0] + meta[1:]) == 0           # no fit range
    low = [1000, 0, 186, 15000, m, 7 * m, f, budget, 0, 0, 0, 0]
    assert applies(m, 7 * m, 1, f, low) == 0                    # fewer than 8 edges per row: the lane-group kernels
    assert lib.dfgnn_plan_applies(m, nnz, 1, f, None) == 0


def test_header_arity_matches_loader():
    """Argument counts in the header equal the ctypes argtypes (catches a drifted binding)."""
    import dfgnn_native
    text = open(os.path.join(ROOT, "include", "dfgnn.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    for name, argtypes in dfgnn_native.SIGNATURES.items():
        m = re.search(r"\bint\s+" + name + r"\s*\((.*?)\)\s*;", text, flags=re.S)
        assert m, name
        assert len([a for a in m.group(1).split(",") if a.strip()]) == len(argtypes), name


def test_library_is_built_from_the_sources_on_disk():
    """dfgnn_build_id() (sha256 of the sources, baked in at build time) equals the hash of csrc/ as it is now: a stale
    libdfgnn.so cannot pass for a fresh one just because its ABI number matches."""
    import dfgnn_native
    assert dfgnn_native.build_id() == dfgnn_native.source_hash()


def test_torch_extension_binds_the_same_c_abi():
    """_dfgnn_ext.so (csrc/torch_ext.cpp, the torch/extension.h shim) imports nothing but dfgnn_* symbols that
    include/dfgnn.h declares and libdfgnn.so exports, and was built against this library build."""
    import subprocess
    import dfgnn_native
    assert os.path.exists(dfgnn_native.EXT_PATH), "_dfgnn_ext.so missing: run __graft_entry__.build()"
    out = subprocess.run(["nm", "-D", "--undefined-only", dfgnn_native.EXT_PATH], capture_output=True, text=True, check=True).stdout
    used = sorted({ln.split()[-1] for ln in out.splitlines() if " dfgnn_" in ln})
    assert "dfgnn_gt_hyper_fwd" in used and "dfgnn_gt_bwd" in used and "dfgnn_gat_softmax_fwd" in used
    assert set(used) <= set(_declared())
    ext = dfgnn_native.ext()
    assert ext is not None and ext.abi_version() == 11 and ext.build_id() == dfgnn_native.source_hash()


def test_graft_entry_build_passes():
    """__graft_entry__.build() -- what the driver runs on the CPU box every round: make is a no-op on an up-to-date tree,
    the stale-library check, the ABI check against include/dfgnn.h and the package imports must all pass."""
    import __graft_entry__ as g
    g.build()


def test_library_opened_ahead_of_torch_brings_no_hip_runtime():
    """libdfgnn.so is a forwarder (csrc/gen_shim.py): opening it, asking for its ABI number, its build id or the text of an
    argument error loads neither torch nor a HIP runtime, so a host may link or open it in any order relative to torch (a
    library that itself linked libamdhip64 brought /opt/rocm's runtime in ahead of torch's own: round 2's 'bad argument'
    on every torch device pointer).  The kernels library libdfgnn_hip.so is opened at the first call that needs it; the
    GPU suite runs an operator in a process that opened libdfgnn.so BEFORE importing torch."""
    import subprocess
    code = ("import sys, ctypes; sys.path.insert(0, %r); import dfgnn_native as n; "
            "L = ctypes.CDLL(n.LIB_PATH); L.dfgnn_build_id.restype = ctypes.c_char_p; "
            "L.dfgnn_error_string.restype = ctypes.c_char_p; L.dfgnn_error_string.argtypes = [ctypes.c_int]; "
            "assert L.dfgnn_abi_version() == 11 and L.dfgnn_build_id().decode() == n.source_hash() == n.build_id(); "
            "assert b'bad argument' in L.dfgnn_error_string(-1); "
            "maps = open('/proc/self/maps').read(); "
            "assert 'torch' not in sys.modules and 'libamdhip64' not in maps and 'libdfgnn_hip' not in maps, maps[-3000:]; "
            "n.lib().dfgnn_plan_ints(10, 40); assert 'libdfgnn_hip.so' in open('/proc/self/maps').read()"
            ) % os.path.join(ROOT, "df-gnn_amd")
    subprocess.run([sys.executable, "-c", code], check=True, timeout=300)


def test_forwarder_has_no_hip_dependency_and_covers_the_header():
    """readelf / nm on the two libraries: libdfgnn.so needs no HIP library and exports every symbol of include/dfgnn.h;
    libdfgnn_hip.so exports them too (it is what the forwarder resolves them in)."""
    import subprocess
    import dfgnn_native
    hip = os.path.join(os.path.dirname(dfgnn_native.LIB_PATH), "libdfgnn_hip.so")
    needed = subprocess.run(["readelf", "-d", dfgnn_native.LIB_PATH], capture_output=True, text=True, check=True).stdout
    assert "amdhip" not in needed and "hsa" not in needed and "RUNPATH" not in needed and "RPATH" not in needed, needed
    for path in (dfgnn_native.LIB_PATH, hip):
        out = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True, check=True).stdout
        exported = {ln.split()[-1] for ln in out.splitlines()}
        missing = [n for n in _declared() if n not in exported]
        assert not missing, (path, missing)
