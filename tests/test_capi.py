"""CPU: the C-ABI library builds for gfx950, loads, exports every symbol that
include/mrl_envs.h declares, and fails loudly (no CPU fallback) when there is
no GPU.  No compute calls here."""
import ctypes
import os
import re

import pytest
import torch

from conftest import REPO


def declared_symbols():
    text = open(os.path.join(REPO, "include", "mrl_envs.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mrl_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_list_agree():
    from madrona_rl_envs_playground_amd import _lib
    assert declared_symbols() == sorted(_lib.SYMBOLS)


def test_library_exports_every_declared_symbol(hip_lib):
    for sym in declared_symbols():
        assert hasattr(hip_lib, sym), f"libmrl_envs.so does not export {sym}"
    from madrona_rl_envs_playground_amd import _lib
    assert hip_lib.mrl_abi_version() == _lib.ABI_VERSION == 4


def test_library_is_gfx950_only():
    from madrona_rl_envs_playground_amd import _lib
    blob = open(_lib.LIB_PATH, "rb").read()
    assert b"gfx950" in blob
    for other in (b"gfx942", b"gfx90a", b"sm_90"):
        assert other not in blob


def test_null_handle_and_bad_slot_are_errors_not_crashes(hip_lib):
    assert hip_lib.mrl_step(None, None) != 0
    assert b"null simulator" in hip_lib.mrl_last_error()
    assert hip_lib.mrl_num_worlds(None) == 0


def test_debug_knobs_are_an_explicit_call_not_the_environment(hip_lib):
    """Tuning knobs go through mrl_debug_set; unknown keys are errors; the shipped sources read no
    environment variable (a stray variable must not change the product's code path)."""
    from madrona_rl_envs_playground_amd import _lib
    _lib.debug_set("overcooked.wpw", 4)
    _lib.debug_set(None, 0)
    with pytest.raises(_lib.MrlError, match="unknown key"):
        _lib.debug_set("overcooked.typo", 1)
    csrc = os.path.join(REPO, "madrona_rl_envs_playground_amd", "csrc")
    for f in os.listdir(csrc):
        if os.path.isfile(os.path.join(csrc, f)):
            assert "getenv" not in open(os.path.join(csrc, f)).read(), f


def test_abi_version_mismatch_is_refused(hip_lib, monkeypatch):
    from madrona_rl_envs_playground_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "ABI_VERSION", 999)
    with pytest.raises(_lib.MrlError, match="ABI version"):
        _lib.lib()
    monkeypatch.undo()
    assert _lib.lib().mrl_abi_version() == _lib.ABI_VERSION


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU failure mode")
def test_no_gpu_means_loud_failure_not_cpu_fallback():
    from madrona_rl_envs_playground_amd import layouts
    from madrona_rl_envs_playground_amd.simulators import CartpoleSimulator, ExecMode, OvercookedSimulator
    params = layouts.get_base_layout_params("cramped_room", 400)
    with pytest.raises(RuntimeError, match="HIP|device|GPU"):
        OvercookedSimulator(exec_mode=ExecMode.CUDA, gpu_id=0, num_worlds=4, **params)
    with pytest.raises(RuntimeError):
        CartpoleSimulator(exec_mode=ExecMode.CUDA, gpu_id=0, num_worlds=4)
    with pytest.raises(NotImplementedError):
        CartpoleSimulator(exec_mode=ExecMode.CPU, gpu_id=0, num_worlds=4)


def test_product_never_imports_the_oracle():
    """oracle/ is test infrastructure: nothing under the package may reference it."""
    pkg = os.path.join(REPO, "madrona_rl_envs_playground_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                text = open(os.path.join(root, f)).read()
                assert "import oracle" not in text and "from oracle" not in text and "mrl_oracle" not in text, f


def test_header_lists_every_debug_key():
    """include/mrl_envs.h names the keys mrl_debug_set accepts: the list must be the one csrc/capi.hip checks against."""
    capi = open(os.path.join(REPO, "madrona_rl_envs_playground_amd", "csrc", "capi.hip")).read()
    table = capi[capi.index("kDebugKeys[] = {"):capi.index("};", capi.index("kDebugKeys[] = {"))]
    in_code = set(re.findall(r'^\s*"([a-z_.]+)"', table, flags=re.M))
    header = open(os.path.join(REPO, "include", "mrl_envs.h")).read()
    doc = header[header.index("Keys (with their meanings"):header.index("key == NULL forgets")]
    in_header = set(re.findall(r"\b((?:overcooked|hanabi|cartpole)\.[a-z_]+|fused_step|fused_heal_test|inject_scan_timeout|ablate|stamps)\b", doc))
    assert in_code == in_header and len(in_code) >= 20


def test_library_proves_which_sources_it_was_built_from(hip_lib, tmp_path, monkeypatch):
    """The hash of the sources is compiled into the library (csrc/Makefile -> mrl_build_hash()).  A library lying beside
    sources it was not built from is refused -- or rebuilt, where hipcc is at hand -- whatever the file times say."""
    import shutil
    import subprocess
    from madrona_rl_envs_playground_amd import _lib
    assert hip_lib.mrl_build_hash().decode() == _lib.source_hash() == _lib.embedded_hash(_lib.LIB_PATH)
    # a copy of the package's native part: sources, header, the built library
    pkg = tmp_path / "pkg"
    shutil.copytree(_lib.CSRC, pkg / "csrc", ignore=shutil.ignore_patterns("build", "build_diag"))
    (tmp_path / "include").mkdir()
    shutil.copy(_lib.HEADER, tmp_path / "include" / "mrl_envs.h")
    shutil.copy(_lib.LIB_PATH, pkg / "libmrl_envs.so")
    monkeypatch.setattr(_lib, "CSRC", str(pkg / "csrc"))
    monkeypatch.setattr(_lib, "HEADER", str(tmp_path / "include" / "mrl_envs.h"))
    monkeypatch.setattr(_lib, "LIB_PATH", str(pkg / "libmrl_envs.so"))
    monkeypatch.setattr(_lib, "_lib", None)
    assert _lib.embedded_hash(_lib.LIB_PATH) == _lib.source_hash()           # the copy is consistent (not loaded: dlopen would keep it)
    # one byte of one header changes; the library is given the newest file time of all
    hpp = pkg / "csrc" / "random_policy.hpp"
    text = hpp.read_bytes()
    at = text.index(b"0x9E3779B9")
    hpp.write_bytes(text[:at] + b"0x9E3779B1" + text[at + 10:])
    os.utime(pkg / "libmrl_envs.so")
    before = _lib.source_hash()
    assert before != _lib.embedded_hash(_lib.LIB_PATH)
    monkeypatch.setattr(_lib, "_lib", None)
    real_run = subprocess.run
    monkeypatch.setattr(subprocess, "run", lambda *a, **k: (_ for _ in ()).throw(OSError("no compiler here")))
    with pytest.raises(_lib.MrlError, match="built from other sources"):
        _lib.lib()
    monkeypatch.setattr(subprocess, "run", real_run)
    # with the compiler at hand the same call rebuilds -- make alone would have called the newer file up to date
    L = _lib.lib()
    assert L.mrl_build_hash().decode() == before == _lib.embedded_hash(_lib.LIB_PATH)
    monkeypatch.undo()
    assert _lib.lib().mrl_build_hash().decode() == _lib.source_hash()
