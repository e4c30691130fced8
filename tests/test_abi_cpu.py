"""CPU: the C-ABI library builds, loads and exports every symbol the header declares;
the product path has no CPU fallback and never touches the oracle."""
import ctypes as C
import os
import pathlib
import re
import subprocess

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from sonar_amd import _lib, build

    build.build(verbose=False)
    return _lib.load()


def _header_functions():
    src = open(os.path.join(ROOT, "include", "sonar_mi355.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return set(re.findall(r"\b(smi_[a-z0-9_]+)\s*\(", src))


def test_header_symbols_exported_and_bound(lib):
    from sonar_amd import _lib

    declared = _header_functions()
    assert declared, "no functions parsed from the header"
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    out = subprocess.run(["nm", "-D", "--defined-only", str(_lib.LIB_PATH)], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r"\bT (smi_[a-z0-9_]+)", out))
    assert declared <= exported, declared - exported


def test_version_and_helpers(lib):
    assert b"gfx950" in lib.smi_version()
    assert lib.smi_xsim_padded_rows(1) == 256 and lib.smi_xsim_padded_rows(256) == 256 and lib.smi_xsim_padded_rows(257) == 512
    assert lib.smi_xsim_workspace_bytes(0, 5, 1, 1024) == 0
    # 8 chunks of partial lists + the tile-major copies of both padded matrices (every k runs on the 256x256 engine since
    # the per-row lists moved to LDS, r04 experiment 2)
    assert lib.smi_xsim_workspace_bytes(1000, 100000, 4, 1024) == 8 * 1024 * 4 * 8 + (1024 + 100096) * 1024 * 2
    assert lib.smi_xsim_workspace_bytes(1000, 100000, 8, 1024) == 8 * 1024 * 8 * 8 + (1024 + 100096) * 1024 * 2


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-device behaviour")
def test_no_device_fails_loudly(lib):
    from sonar_amd import _lib

    assert lib.smi_device_count() == 0
    assert lib.smi_init(0) == -3
    assert b"no HIP device" in lib.smi_last_error()
    with pytest.raises(_lib.SmiError):
        _lib.check(lib.smi_init(0))
    # handle setters refuse a null handle instead of dereferencing it
    assert lib.smi_text_decoder_set_beam_logits_dtype(None, _lib.SMI_F16) != 0
    assert lib.smi_text_decoder_set_chains(None, 1) != 0


def test_engine_refuses_cpu_and_missing_library(monkeypatch, tmp_path):
    from oracle import text_encoder as O
    from sonar_amd import _lib
    from sonar_amd.text_encoder import SonarTextEncoderConfig, TextEncoderEngine, VocabularyInfo

    cfg = SonarTextEncoderConfig(model_dim=256, num_encoder_layers=1, num_encoder_attn_heads=4,
                                 ffn_inner_dim=256, vocab_info=VocabularyInfo(size=50))
    ocfg = O.OracleTextEncoderConfig(model_dim=256, num_layers=1, num_heads=4, ffn_inner_dim=256, vocab_size=50)
    with pytest.raises(RuntimeError, match="HIP device only"):
        TextEncoderEngine(cfg, O.make_synthetic_params(ocfg), device="cpu")
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", tmp_path / "nope.so")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.load()


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "sonar_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), os.path.join(dirpath, f)


def test_tuning_registry_and_no_environment_reads(monkeypatch):
    """Round 5: every A/B switch is an integer of the tuning registry set through the C-ABI; the library
    itself has no `getenv` (nm: the symbol is not even imported), so a production process cannot be steered -- or raced --
    through os.environ.  The Python loader forwards SMI_<NAME> variables once, at load."""
    import ctypes as C
    import subprocess

    from sonar_amd import _lib

    lib = _lib.load()
    names = _lib.tuning_names()
    assert "LONE" in names and "DEC_KS_OUT" in names and "DEC_SLAB_F16" in names and len(set(names)) == len(names)
    und = subprocess.run(["nm", "-D", "--undefined-only", str(_lib.LIB_PATH)], capture_output=True, text=True).stdout
    assert "getenv" not in und
    for src in (p for p in (pathlib.Path(ROOT) / "sonar_amd" / "csrc").iterdir() if p.suffix in (".hip", ".hpp", ".cpp")):
        code = re.sub(r"//.*", "", src.read_text())
        assert "getenv" not in code, src
    v, st = C.c_int32(7), C.c_int32(7)
    assert lib.smi_tuning_get(b"DEC_KS_OUT", C.byref(v), C.byref(st)) == 0 and st.value == 0
    with _lib.tuning(DEC_KS_OUT=2, SMI_G2_AUTO_MIN=-5):          # with or without the SMI_ prefix; negative values survive
        assert lib.smi_tuning_get(b"SMI_DEC_KS_OUT", C.byref(v), C.byref(st)) == 0 and (v.value, st.value) == (2, 1)
        assert lib.smi_tuning_get(b"G2_AUTO_MIN", C.byref(v), C.byref(st)) == 0 and (v.value, st.value) == (-5, 1)
        with _lib.tuning(DEC_KS_OUT=4):
            lib.smi_tuning_get(b"DEC_KS_OUT", C.byref(v), C.byref(st))
            assert v.value == 4
        lib.smi_tuning_get(b"DEC_KS_OUT", C.byref(v), C.byref(st))
        assert v.value == 2
    assert lib.smi_tuning_get(b"DEC_KS_OUT", C.byref(v), C.byref(st)) == 0 and st.value == 0
    assert lib.smi_tuning_set(b"NO_SUCH_SWITCH", 1) != 0 and b"NO_SUCH_SWITCH" in lib.smi_last_error()
    # the loader's one-time forwarding
    monkeypatch.setenv("SMI_LONE_KS", "4")
    _lib._forward_env_switches(lib)
    lib.smi_tuning_get(b"LONE_KS", C.byref(v), C.byref(st))
    assert (v.value, st.value) == (4, 1)
    _lib.set_tuning(LONE_KS=None)
    monkeypatch.setenv("SMI_LONE_KS", "four")
    with pytest.raises(RuntimeError, match="integers"):
        _lib._forward_env_switches(lib)
