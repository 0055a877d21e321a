"""The C-ABI library loads and exports every symbol include/cat_sim.h declares (no compute calls:
no GPU here); the product package never touches the oracle; compute entry points refuse to run
without a device."""
import ctypes as C
import re
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]


def _declared_symbols():
    text = (ROOT / "include" / "cat_sim.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(cat_[a-z_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from as_cops_and_thieves_amd import _native
    _native.build()
    L = _native.lib()
    declared = _declared_symbols()
    assert len(declared) >= 13
    for sym in declared:
        assert hasattr(L, sym), f"libcat_sim.so does not export {sym}"
    assert set(declared) == set(_native.EXPORTED_SYMBOLS)
    assert L.cat_abi_version() == 1


def test_lstm_library_exports_every_declared_symbol_and_struct_layouts_match():
    """include/cat_lstm.h <-> libcat_learn.so <-> the ctypes mirror (no compute call: no GPU here)."""
    from as_cops_and_thieves_amd import _learn_native as ln
    ln.build()
    L = ln.lib()
    text = (ROOT / "include" / "cat_lstm.h").read_text()
    code = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    declared = sorted(set(re.findall(r"\b(cat_lstm_[a-z_0-9]+)\s*\(", code)))
    assert set(declared) == set(ln.EXPORTED_SYMBOLS) and all(hasattr(L, s) for s in declared)
    for struct, cls in (("cat_lstm_dims", ln.Dims), ("cat_lstm_fwd", ln.FwdArgs), ("cat_lstm_bwd", ln.BwdArgs)):
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (struct, struct), code, re.S).group(1)
        names = [n for decl in body.split(";") for n in re.findall(r"\b([A-Za-z_0-9]+)\s*(?=,|$)", decl.strip())]
        assert names == [f[0] for f in cls._fields_], (struct, names)
    d = ln.Dims(3, 16, 100, 0)                                   # 7 blocks of 16 sequences
    assert L.cat_lstm_saved_acts_bytes(C.byref(d)) == 16 * 3 * 7 * 16 * 512 * 2
    assert L.cat_lstm_saved_cell_bytes(C.byref(d)) == 16 * 3 * 7 * 16 * 256 * 2
    bad = ln.FwdArgs()                                           # argument checks come before any device call
    assert L.cat_lstm_seq_forward(C.byref(bad), None) == -1 and b"dimensions" in L.cat_lstm_last_error()


@pytest.mark.parametrize("header,prefix,symbols,structs", [
    ("cat_trunk.h", "cat_trunk_", "TRUNK_SYMBOLS", {"cat_trunk_dims": "TrunkDims", "cat_trunk_params": "TrunkParams", "cat_trunk_fwd": "TrunkFwd",
                                                    "cat_trunk_bwd": "TrunkBwd", "cat_trunk_finish_args": "TrunkFinish"}),
    ("cat_ppo.h", "cat_ppo_", "PPO_SYMBOLS", {"cat_ppo_loss": "PpoLoss", "cat_ppo_adam": "PpoAdam"}),
    ("cat_rollout.h", "cat_rollout_", "ROLLOUT_SYMBOLS", {"cat_rollout_pack_args": "PackArgs", "cat_rollout_sample_args": "SampleArgs", "cat_rollout_post_args": "PostArgs"}),
    ("cat_dense.h", "cat_dense_", "DENSE_SYMBOLS", {"cat_dense_dims": "DenseDims", "cat_dense_wgrad_args": "WgradArgs", "cat_dense_gemm_args": "GemmArgs", "cat_dense_sum_job": "SumJob"})])
def test_learner_kernel_headers_match_the_library_and_the_ctypes_mirror(header, prefix, symbols, structs):
    from as_cops_and_thieves_amd import _learn_native as ln
    ln.build()
    L = ln.lib()
    code = re.sub(r"/\*.*?\*/", "", (ROOT / "include" / header).read_text(), flags=re.S)
    declared = sorted(set(re.findall(r"\b(%s[a-z_0-9]+)\s*\(" % prefix, code)))
    assert set(declared) == set(getattr(ln, symbols)) and all(hasattr(L, s) for s in declared)
    for struct, cls in structs.items():
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (struct, struct), code, re.S).group(1)
        body = re.sub(r"\[[^\]]*\]", "", body)                                  # array extents
        names = [n for decl in body.split(";") for n in re.findall(r"\b([A-Za-z_0-9]+)\s*(?=,|$)", decl.strip())]
        assert names == [f[0] for f in getattr(ln, cls)._fields_], (struct, names)


def test_learner_kernels_reject_bad_arguments_before_touching_a_device():
    from as_cops_and_thieves_amd import _learn_native as ln
    L = ln.lib()
    d = ln.TrunkDims(3, 1000, 4, 64)
    assert L.cat_trunk_out_positions(C.byref(d)) == 9 and L.cat_trunk_supported(C.byref(d)) == 1
    assert L.cat_trunk_supported(C.byref(ln.TrunkDims(3, 1000, 4, 90))) == 1          # the reference's default sensor
    assert L.cat_trunk_supported(C.byref(ln.TrunkDims(3, 1000, 4, 200))) == 0         # does not fit the LDS (dense path)
    assert L.cat_trunk_forward(C.byref(ln.TrunkFwd()), None) == -1 and b"dimensions" in L.cat_trunk_last_error()
    assert L.cat_ppo_loss_grad(C.byref(ln.PpoLoss()), None) == -1 and L.cat_ppo_adam_step(C.byref(ln.PpoAdam()), None) == -1
    assert L.cat_dense_bias_act(C.byref(ln.DenseDims(1, 1, 3, 0)), None, None, 0, None) == -1        # out = 3: neither 1 nor 4 k


def test_struct_layouts_match_header_field_order():
    from as_cops_and_thieves_amd import _native
    text = (ROOT / "include" / "cat_sim.h").read_text()
    for struct, cls in (("cat_config", _native.CatConfig), ("cat_outputs", _native.CatOutputs),
                        ("cat_state", _native.CatState), ("cat_tables", _native.CatTables)):
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (struct, struct), text, re.S).group(1)
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        names = re.findall(r"\*?\s*([a-z_0-9]+);", body)
        assert names == [f[0] for f in cls._fields_], struct
    assert C.sizeof(_native.CatConfig) == 8 * 4 + 8 + 8 + 11 * 8


def test_no_device_is_a_loud_error_not_a_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from as_cops_and_thieves_amd import _native
    from as_cops_and_thieves_amd.config import SimConfig
    from as_cops_and_thieves_amd.maps import load_preset
    from as_cops_and_thieves_amd.sim import CatSim, CatSimError
    with pytest.raises(CatSimError):
        CatSim(SimConfig(), [load_preset("squarinth").compile()])
    # and straight through the C ABI: cat_create reports CAT_ERR_NO_DEVICE
    L = _native.lib()
    cfg = _native.CatConfig()
    cfg.n_envs, cfg.n_cops, cfg.n_thieves, cfg.n_rays = 1, 2, 1, 8
    blob = load_preset("squarinth").compile().to_blob()
    import numpy as np
    dx = np.zeros(8); lut = np.zeros(32768, np.float32)
    tabs = _native.CatTables(dx.ctypes.data, dx.ctypes.data, lut.ctypes.data, lut.ctypes.data)
    arr = (C.c_char_p * 1)(blob); sizes = (C.c_size_t * 1)(len(blob)); h = C.c_void_p()
    rc = L.cat_create(C.byref(cfg), C.byref(tabs), arr, sizes, 1, None, 0, C.byref(h))
    assert rc == -4 and b"no CPU path" in L.cat_last_error(None)


def test_missing_library_raises(monkeypatch, tmp_path):
    from as_cops_and_thieves_amd import _native
    monkeypatch.setattr(_native, "_lib", None)
    monkeypatch.setattr(_native, "LIB_PATH", tmp_path / "nope.so")
    with pytest.raises(_native.NativeLibraryMissing):
        _native.lib()


def test_product_package_never_references_the_oracle():
    pkg = ROOT / "as_cops_and_thieves_amd"
    offenders = []
    for f in list(pkg.rglob("*.py")) + list(pkg.rglob("*.hip")) + list((ROOT / "include").glob("*.h")):
        txt = f.read_text()
        if re.search(r"\boracle\b|cat_oracle|cato_", txt):
            offenders.append(str(f.relative_to(ROOT)))
    assert offenders == [], offenders
