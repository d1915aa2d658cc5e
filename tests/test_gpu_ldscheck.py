"""The bounds-checking build of the matrix-core kernels (csrc `make ldscheck`: every LDS address the dfgnn_dense.hpp helpers
form and every body's LDS carve-up is compared with the workgroup's allocation; SURVEY.md section 5, sanitizers -- GPU
AddressSanitizer is not available on this pool) runs every range class and head layout without a violation, and does
count a deliberate one."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "df-gnn_amd", "libdfgnn_ldscheck.so")


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_checked_lds_addressing_finds_nothing_on_every_range_class():
    assert os.path.exists(LIB), "build it: make -C df-gnn_amd/csrc ldscheck (__graft_entry__.build() does)"
    env = dict(os.environ, DFGNN_LIB="libdfgnn_ldscheck.so")
    run = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "diag", "lds_check_run.py")], env=env,
                         capture_output=True, text=True, timeout=550)
    assert run.returncode == 0, run.stderr[-3000:]
    rep = json.loads(run.stdout.strip().splitlines()[-1])
    assert rep["build_id"] == rep["source_hash"], "libdfgnn_ldscheck.so was built from other sources"
    assert rep["start"]["units"] >= 2                       # gt_dense.hip and gt_dense_stats.hip carry the checks
    st = rep["selftest"]                                    # the deliberate store past 4096 bytes of dynamic LDS
    assert (st["violations"], st["end"], st["limit"]) == (1, 4096 + 16, 4096) and st["readback"] == 2.0, st
    assert rep["launches"] > 200
    for case in rep["cases"]:
        assert case["violations"] == 0, case
