"""CPU: the C-ABI shared library loads and exports every symbol include/gclm.h declares.
No compute calls here (no GPU in this container): gclm_create must FAIL LOUDLY without a device."""
import ctypes as C
import os
import re
import subprocess

import pytest

from conftest import ROOT

from geocalib_amd import _lib

HEADER = os.path.join(ROOT, "include", "gclm.h")


def declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(gclm_[a-z_0-9]+)\s*\(", src)))


def test_library_is_built_in_tree():
    assert os.path.exists(_lib.LIB_PATH), "run `python __graft_entry__.py` (build()) first"
    assert os.path.realpath(_lib.LIB_PATH).startswith(os.path.realpath(ROOT))


def test_header_symbols_are_exported_and_bound():
    names = declared_symbols()
    assert len(names) >= 15
    lib = C.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/gclm.h but not exported"
    assert set(names) == set(_lib.EXPORTED_SYMBOLS), "ctypes binding and header disagree"
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout
    exported = set(re.findall(r" T (gclm_\w+)", out))
    assert exported == set(names), exported ^ set(names)


def test_code_object_targets_gfx950():
    """The embedded device code must be gfx950 (no other arch, no generic fallback)."""
    blob = open(_lib.LIB_PATH, "rb").read()
    assert b"gfx950" in blob
    for other in (b"gfx942", b"gfx90a", b"gfx1100", b"sm_90"):
        assert other not in blob


def test_default_config_matches_reference_defaults():
    lib = _lib.load()
    assert lib.gclm_version() == 610 == _lib.ABI_VERSION
    header = open(HEADER).read()
    assert re.search(r"#define GCLM_VERSION\s+(\d+)", header).group(1) == "610"
    cfg = _lib.GclmConfig()
    assert lib.gclm_default_config(C.byref(cfg)) == 0
    assert (cfg.struct_size, cfg.abi_version, cfg.device) == (C.sizeof(_lib.GclmConfig), 610, 0)
    assert lib.gclm_abi_config_size() == C.sizeof(_lib.GclmConfig)
    # LMOptimizer.default_conf, lm_optimizer.py:144-162
    assert (cfg.camera_model, cfg.shared_intrinsics, cfg.num_steps, cfg.fix_lambda, cfg.early_stop) == (0, 0, 30, 0, 1)
    assert cfg.lambda0 == pytest.approx(0.1) and cfg.atol == pytest.approx(1e-8) and cfg.rtol == pytest.approx(1e-8)
    assert cfg.use_spherical_manifold == 1 and cfg.use_log_focal == 1
    assert cfg.up_loss_fn_scale == pytest.approx(1e-2) and cfg.lat_loss_fn_scale == pytest.approx(1e-2)
    assert cfg.estimate_gravity == cfg.estimate_focal == cfg.estimate_dist == cfg.compute_uncertainty == 1
    assert cfg.heuristic_init == 0 and C.sizeof(_lib.GclmConfig) == 21 * 4


def test_create_rejects_bad_config_with_message():
    lib = _lib.load()
    cfg = _lib.GclmConfig()
    lib.gclm_default_config(C.byref(cfg))
    cfg.camera_model = 7
    h = C.c_void_p()
    assert lib.gclm_create(C.byref(h), C.byref(cfg)) != 0 and not h
    assert "camera_model" in _lib.last_error(None)
    lib.gclm_default_config(C.byref(cfg))
    cfg.num_steps = 100000
    assert lib.gclm_create(C.byref(h), C.byref(cfg)) != 0
    assert "num_steps" in _lib.last_error(None)


def test_create_rejects_a_stale_caller():
    """A caller compiled against another include/gclm.h (round 1/2: ABI 100, an 18-field struct without the stamp) is
    refused with a message that names both sides, before any field of its struct is trusted."""
    lib = _lib.load()
    h = C.c_void_p()
    cfg = _lib.GclmConfig()
    lib.gclm_default_config(C.byref(cfg))
    cfg.abi_version = 100
    assert lib.gclm_create(C.byref(h), C.byref(cfg)) == -5 and not h
    assert "ABI mismatch" in _lib.last_error(None) and "100" in _lib.last_error(None) and "610" in _lib.last_error(None)
    lib.gclm_default_config(C.byref(cfg))
    cfg.struct_size = 18 * 4
    assert lib.gclm_create(C.byref(h), C.byref(cfg)) == -5 and "struct_size 72" in _lib.last_error(None)
    # a round-2 struct starts with camera_model = 0, shared_intrinsics = 0: both stamps read as 0
    zero = _lib.GclmConfig()
    assert lib.gclm_create(C.byref(h), C.byref(zero)) == -5


def test_rccl_is_bound_at_run_time_not_at_link_time():
    """libgeocalib_hip.so is not linked against librccl: which RCCL it talks to is decided on first use (the one the process
    already holds -- torch's -- else ROCm's), so two RCCL instances never meet in one process by an accident of load order."""
    out = subprocess.run(["readelf", "-d", _lib.LIB_PATH], capture_output=True, text=True).stdout
    assert "librccl" not in out, out
    lib = _lib.load()
    comp, run = C.c_int(0), C.c_int(0)
    assert lib.gclm_comm_versions(C.byref(comp), C.byref(run)) == 0
    assert comp.value >= 22000                       # NCCL_VERSION_CODE of the rccl.h of the build
    assert run.value == 0 or run.value // 10000 == comp.value // 10000     # 0: no librccl on this box at all


def test_no_cpu_fallback():
    """Without a HIP device the product path must refuse to run (and say why), never fall back."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    lib = _lib.load()
    cfg = _lib.GclmConfig()
    lib.gclm_default_config(C.byref(cfg))
    h = C.c_void_p()
    assert lib.gclm_create(C.byref(h), C.byref(cfg)) != 0
    assert "no HIP device" in _lib.last_error(None)
    from geocalib_amd import LMOptimizer
    data = {"up_field": torch.zeros(1, 2, 8, 8), "latitude_field": torch.zeros(1, 1, 8, 8)}
    with pytest.raises(RuntimeError, match="HIP device"):
        LMOptimizer({})(data)


def test_product_package_never_touches_the_oracle():
    """The oracle is test infrastructure: nothing under geocalib_amd/ may import, link or load it."""
    pkg = os.path.join(ROOT, "geocalib_amd")
    pat = re.compile(r"lm_oracle|from\s+oracle|import\s+oracle|oracle/|liblm_oracle|ref_import")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")) or f == "Makefile":
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert not pat.search(txt), (dirpath, f, pat.search(txt).group(0))


def test_documented_bindings_match_the_header():
    """The ctypes stub shown in INTEGRATION.md and struct gclm_config of include/gclm.h list the same fields, in the
    order of the binding the package itself uses (a stale stub would let gclm_default_config write past the struct)."""
    import re
    from conftest import ROOT
    header = open(os.path.join(ROOT, "include", "gclm.h")).read()
    body = header[header.index("typedef struct gclm_config {"):header.index("} gclm_config;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    in_header = [n for decl in re.findall(r"(?:int32_t|float)\s+([^;]+);", body) for n in re.split(r"\s*,\s*", decl.strip())]
    ours = [n for n, _ in _lib.GclmConfig._fields_]
    assert in_header == ours, (in_header, ours)
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    stub = doc[doc.index("class Config(C.Structure):"):doc.index("lib.gclm_last_error.restype")]
    assert re.findall(r'\("(\w+)", C\.c_', stub) == ours
