"""CPU-only checks of the host side: the C-ABI library loads and exports exactly what
include/manytor_hip.h declares, the ctypes mirror matches the C struct, the parity-mode RNG
streams match the reference's draw order, and the product refuses to run without a GPU."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "manytor_hip.h")


@pytest.fixture(scope="module")
def lib():
    from manytor_amd import build, _lib
    build.build_library()            # hipcc cross-compiles gfx950 without a GPU
    return _lib.load()


def declared_symbols():
    text = open(HEADER).read()
    return sorted(set(re.findall(r"^MT_API\s+[\w\s\*]+?\b(mt_\w+)\s*\(", text, flags=re.M)))


def test_library_exports_every_declared_symbol(lib):
    from manytor_amd import _lib
    names = declared_symbols()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), f"{n} declared in manytor_hip.h but not exported"
    assert sorted(_lib.PROTOTYPES) == names, "ctypes prototype table and header disagree"


def test_version_and_status_strings(lib):
    text = open(HEADER).read()
    assert lib.mt_version() == int(re.search(r"#define MT_VERSION (\d+)", text).group(1))
    assert lib.mt_status_string(0) == b"ok"
    assert b"invalid" in lib.mt_status_string(-1)


def test_ctypes_struct_matches_c_layout(lib, tmp_path):
    from manytor_amd import _lib
    src = tmp_path / "sz.c"
    src.write_text(
        '#include <stdio.h>\n#include <stddef.h>\n#include "manytor_hip.h"\n'
        'int main(void){printf("%zu %zu %zu %zu %zu %zu\\n", sizeof(mt_config), offsetof(mt_config,n_envs),'
        ' offsetof(mt_config,dof), offsetof(mt_config,pickup_tol), offsetof(mt_config,dh_table),'
        ' offsetof(mt_config,return_ring));return 0;}\n')
    exe = tmp_path / "sz"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    c = [int(v) for v in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()]
    S = _lib.MtConfig
    assert c == [ctypes.sizeof(S), S.n_envs.offset, S.dof.offset, S.pickup_tol.offset, S.dh_table.offset,
                 S.return_ring.offset]
    # the enums mirrored in _lib.py
    text = open(HEADER).read()
    for name, val in (("MT_F_JOINTS", _lib.F_JOINTS), ("MT_F_TOTAL_REWARD", _lib.F_TOTAL_REWARD), ("MT_F_DONE_BITS", _lib.F_DONE_BITS),
                      ("MT_F_RETURN_RING", _lib.F_RETURN_RING), ("MT_F_TRACE", _lib.F_TRACE)):
        assert int(re.search(rf"{name} = (\d+)", text).group(1)) == val


def test_plain_c_program_links_and_calls_the_abi(lib, tmp_path):
    """The boundary is a C ABI: a C99 translation unit (no C++, no Python) includes the header, links the shared
    library and gets sane answers from the calls that need no GPU."""
    from manytor_amd import _lib
    src = tmp_path / "abi.c"
    src.write_text(r'''
#include <stdio.h>
#include <string.h>
#include "manytor_hip.h"
int main(void) {
  mt_config cfg;
  memset(&cfg, 0, sizeof cfg);
  mt_handle h = 0;
  int64_t total = -1;
  uint64_t bad = 0;
  char msg[256];
  int rc_create, rc[4];
  cfg.struct_size = 4; /* wrong on purpose */
  rc_create = mt_create(&h, &cfg);
  strncpy(msg, mt_last_error(0), sizeof msg - 1); /* the message belongs to the library until the next failing call */
  msg[sizeof msg - 1] = 0;
  rc[0] = mt_gather_returns(0, MT_F_TOTAL_REWARD, 0, 0, 0);
  rc[1] = mt_comm_total_envs(0, &total);
  rc[2] = mt_bad_action_count(0, &bad);
  rc[3] = mt_env_step(0, 0, 0, 0, 0, 0);
  {
    mt_return_stats st;
    if (mt_reduce_returns(0, MT_F_TOTAL_REWARD, 0, &st) != MT_ERR_INVALID_ARG || sizeof st != 40) return 2;
  }
  printf("%d|%s|%d|%s|%d|%d|%d|%d\n", mt_version(), mt_status_string(MT_ERR_STATE), rc_create, msg, rc[0], rc[1], rc[2], rc[3]);
  return h != 0;
}
''')
    exe = tmp_path / "abi"
    libdir = os.path.dirname(_lib.LIB_PATH)
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe),
                    _lib.LIB_PATH, f"-Wl,-rpath,{libdir}"], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.strip().split("|")
    assert out[0] == str(lib.mt_version()) and out[1] == "invalid call order"
    assert int(out[2]) == _lib.MT_ERR_INVALID_ARG and "struct_size" in out[3]
    assert [int(v) for v in out[4:]] == [_lib.MT_ERR_INVALID_ARG] * 4


def test_no_gpu_means_loud_failure_not_fallback(lib):
    import manytor_amd as m
    if m.device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(m.ManytorError):
        m.Multienv((3, 2), 7)
    with pytest.raises(m.ManytorError):
        m.Environment(10)
    with pytest.raises(m.ManytorError):
        m.fk(4, [0, 0, 0, 0])


def test_missing_extension_is_a_loud_import_error(lib, monkeypatch):
    """No .so, no product: nothing falls back to numpy."""
    from manytor_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", os.path.join(ROOT, "manytor_amd", "no_such_library.so"))
    with pytest.raises(ImportError, match="no CPU fallback"):
        _lib.load()
    import manytor_amd as m
    with pytest.raises(ImportError):
        m.StepEngine(4, 2)
    with pytest.raises(ImportError):
        m.fk(4, [0, 0, 0, 0])


def test_invalid_config_is_rejected_before_touching_a_device(lib):
    from manytor_amd import _lib
    cfg = _lib.MtConfig()
    h = _lib._HANDLE()
    cfg.struct_size = 4                       # wrong on purpose
    assert lib.mt_create(ctypes.byref(h), ctypes.byref(cfg)) == _lib.MT_ERR_INVALID_ARG
    assert b"struct_size" in lib.mt_last_error(None)
    cfg.struct_size = ctypes.sizeof(_lib.MtConfig)
    cfg.n_envs, cfg.dof, cfg.n_targets, cfg.substeps = 0, 4, 7, 25           # empty batch
    assert lib.mt_create(ctypes.byref(h), ctypes.byref(cfg)) == _lib.MT_ERR_INVALID_ARG
    assert b"n_envs" in lib.mt_last_error(None)
    cfg.n_envs = 1 << 31                                                     # beyond the 32-bit row offsets
    assert lib.mt_create(ctypes.byref(h), ctypes.byref(cfg)) == _lib.MT_ERR_INVALID_ARG
    cfg.n_envs, cfg.dof, cfg.n_targets, cfg.substeps = 16, 9, 7, 25
    assert lib.mt_create(ctypes.byref(h), ctypes.byref(cfg)) == _lib.MT_ERR_INVALID_ARG
    assert b"dof" in lib.mt_last_error(None)
    cfg.dof, cfg.n_targets = 4, 33
    assert lib.mt_create(ctypes.byref(h), ctypes.byref(cfg)) == _lib.MT_ERR_INVALID_ARG
    cfg.n_targets, cfg.substeps = 7, 1
    assert lib.mt_create(ctypes.byref(h), ctypes.byref(cfg)) == _lib.MT_ERR_INVALID_ARG
    assert lib.mt_step(None) == _lib.MT_ERR_INVALID_ARG


def test_parity_mode_rng_streams_match_reference_order(golden):
    """manytor_amd.rng consumes numpy's global stream exactly like the reference (fixture F6)."""
    from manytor_amd import rng
    g = golden("f6_rng_streams")
    np.random.seed(int(g["seed"]))
    np.testing.assert_array_equal(rng.draw_targets(5, 7), g["points_a"])
    np.testing.assert_array_equal(rng.draw_actions(5), g["actions_a"])
    np.testing.assert_array_equal(rng.draw_actions(5), g["actions_b"])
    np.testing.assert_array_equal(rng.draw_targets(5, 7), g["points_b"])
    np.testing.assert_array_equal(rng.draw_actions(5), g["actions_c"])
    np.testing.assert_array_equal(np.random.random_sample(4), g["tail"])


def test_parity_mode_targets_match_f4_resets(golden):
    from manytor_amd import rng
    g = golden("f4_multienv_trace")
    np.random.seed(int(g["seed"]))
    np.testing.assert_array_equal(rng.draw_targets(6, 7), g["points"][0])
    for t in range(50):
        np.testing.assert_array_equal(rng.draw_actions(6), g["action"][t])
    np.testing.assert_array_equal(rng.draw_targets(6, 7), g["points"][1])


def test_product_does_not_import_the_oracle():
    """The oracle is test infrastructure: nothing under manytor_amd/ may reference it."""
    pkg = os.path.join(ROOT, "manytor_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip")):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), f
    code = "import sys; import manytor_amd; assert not any(m == 'oracle' or m.startswith('oracle.') for m in sys.modules)"
    subprocess.run([sys.executable, "-c", code], check=True, cwd=ROOT)


def test_dh7_table_matches_fixture(golden):
    import manytor_amd as m
    np.testing.assert_allclose(np.array(m.DH7_TABLE), golden("f7_dh7_kat")["table"], atol=0)
    from oracle import manytor_oracle as mo
    np.testing.assert_allclose(np.array(m.REF_DH_TABLE), mo.REF_DH_TABLE, atol=0)
