"""CPU tests of the drop-in boundary: the shared library loads without a GPU, exports every
symbol include/mapalign.h declares, and the ctypes mirror matches the header."""
import ctypes as C
import os
import re

import pytest

from conftest import ROOT

HEADER = os.path.join(ROOT, "include", "mapalign.h")


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ma_[a-z_0-9]+)\s*\(", src)))


def test_header_declares_the_survey_abi():
    names = declared_functions()
    for required in ["ma_create", "ma_destroy", "ma_set_gold", "ma_set_source", "ma_set_source_from_depth",
                     "ma_icp", "ma_icp_step", "ma_nn", "ma_last_error", "ma_version"]:
        assert required in names


def test_library_exports_every_declared_symbol(lib_built):
    from mapalign import _lib
    raw = C.CDLL(str(_lib.LIB_PATH))
    for name in declared_functions():
        assert hasattr(raw, name), f"libmapalign.so does not export {name}"
        assert name in _lib.ABI, f"ctypes mirror lacks {name}"
    assert sorted(_lib.ABI) == declared_functions()


def test_version_and_status_strings(lib_built):
    src = open(HEADER).read()
    assert lib_built.ma_version() == int(re.search(r"#define MA_VERSION (\d+)", src).group(1))
    assert lib_built.ma_status_string(0) == b"ok"
    assert b"device" in lib_built.ma_status_string(-2)


def test_struct_layouts_match_header(tmp_path):
    """sizeof/offsetof as gcc sees include/mapalign.h == the ctypes mirror."""
    import subprocess
    from mapalign import _lib
    prog = tmp_path / "layout.c"
    prog.write_text(r"""
#include <stdio.h>
#include <stddef.h>
#include "mapalign.h"
int main(void) {
  printf("%zu %zu %zu %zu\n", sizeof(ma_params), offsetof(ma_params, tol), offsetof(ma_params, max_corr_dist), (size_t)0);
  printf("%zu %zu %zu %zu\n", sizeof(ma_stats), offsetof(ma_stats, rmse), offsetof(ma_stats, inliers), offsetof(ma_stats, gpu_ms));
  printf("%zu %zu %zu %zu\n", sizeof(ma_grid_info), offsetof(ma_grid_info, n_cells), offsetof(ma_grid_info, origin), offsetof(ma_grid_info, build_ms));
  printf("%zu %zu %zu %zu\n", sizeof(ma_mem_info), offsetof(ma_mem_info, device_frees), offsetof(ma_mem_info, bytes_live), offsetof(ma_mem_info, bytes_peak));
  printf("%d %d %d %d\n", MA_COMM_ID_BYTES, MA_ERR_COMM, MA_ERR_NO_MEMORY, MA_VERSION);
  return 0;
}
""")
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), "-o", str(exe), str(prog)], check=True)
    rows = [[int(v) for v in line.split()] for line in subprocess.check_output([str(exe)]).decode().splitlines()]
    P, S, G, M = _lib.ma_params, _lib.ma_stats, _lib.ma_grid_info, _lib.ma_mem_info
    assert rows[3] == [C.sizeof(M), M.device_frees.offset, M.bytes_live.offset, M.bytes_peak.offset]
    assert rows[4][:3] == [_lib.MA_COMM_ID_BYTES, _lib.MA_ERR_COMM, _lib.MA_ERR_NO_MEMORY]
    assert rows[0][:3] == [C.sizeof(P), P.tol.offset, P.max_corr_dist.offset]
    assert rows[1] == [C.sizeof(S), S.rmse.offset, S.inliers.offset, S.gpu_ms.offset]
    assert rows[2] == [C.sizeof(G), G.n_cells.offset, G.origin.offset, G.build_ms.offset]
    src = open(HEADER).read()
    assert int(re.search(r"#define MA_ACC_LEN (\d+)", src).group(1)) == _lib.MA_ACC_LEN
    assert int(re.search(r"#define MA_TRACE_STRIDE (\d+)", src).group(1)) == _lib.MA_TRACE_STRIDE


def test_no_cpu_fallback_without_gpu(lib_built):
    """Without a HIP device the product path must fail loudly, never compute on the CPU."""
    n = C.c_int(0)
    hip = C.CDLL("libamdhip64.so.7")   # by soname: the runtime libmapalign is already bound to (system, or PyTorch's bundled copy)
    if hip.hipGetDeviceCount(C.byref(n)) == 0 and n.value > 0:
        pytest.skip("a GPU is present")
    import mapalign
    with pytest.raises(mapalign.MapAlignError) as e:
        mapalign.MapAligner(0)
    assert e.value.status == -2
    import numpy as np
    with pytest.raises(mapalign.MapAlignError):
        mapalign.align(np.zeros((10, 3), np.float32), np.zeros((10, 3), np.float32))


def test_missing_library_fails_loudly(tmp_path):
    from mapalign import _lib
    with pytest.raises(_lib.MapAlignLibraryError):
        _lib.load(tmp_path / "libmapalign.so")


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "3d-localisation-and-mapping_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in text and "from oracle" not in text, f


def _build_c_example(tmp_path):
    """gcc-compile and link examples/align_c_abi.c against include/mapalign.h and the in-tree library."""
    import subprocess
    libdir = os.path.join(ROOT, "3d-localisation-and-mapping_amd", "lib")
    exe = str(tmp_path / "align_c_abi")
    cmd = ["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "examples", "align_c_abi.c"), "-L", libdir, "-lmapalign",
           f"-Wl,-rpath,{libdir}", "-lm", "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


def test_plain_c_caller_compiles_and_links(lib_built, tmp_path):
    """The boundary is a C ABI: a C99 program using only include/mapalign.h builds against the library
    (no C++, no torch, no Python in the signatures)."""
    _build_c_example(tmp_path)


@pytest.mark.gpu
def test_plain_c_caller_aligns_a_known_transform(lib_built, tmp_path):
    import subprocess
    exe = _build_c_example(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, (r.stdout, r.stderr)
    assert "iterations 50" in r.stdout
