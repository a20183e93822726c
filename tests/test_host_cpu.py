"""CPU suite: host logic, config round trips, synthetic data determinism, and that the C-ABI library
loads and exports every symbol include/csm_hip.h declares (no compute calls without a GPU)."""
import os
import re

import pytest
import torch

import csm_hf_amd
from csm_hf_amd import CSMConfig, CSMModel
from csm_hf_amd.engine import EXPORTS, load_library, rope_tables, llama3_inv_freq
from csm_hf_amd.synth import synth_state_dict, synth_context, state_dict_spec, hash_uniform

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "csm_hip.h")).read()
    declared = set(re.findall(r"\b(csm_[a-z0-9_]+)\s*\(", hdr))
    lib = load_library()
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in csm_hip.h but not exported"
    assert declared == set(EXPORTS)
    from csm_hf_amd.engine import ABI_VERSION
    assert lib.csm_abi_version() == ABI_VERSION == 7


def test_no_gpu_fails_loudly():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    m = CSMModel(CSMConfig.tiny())
    m.load_state_dict(synth_state_dict(CSMConfig.tiny()))
    ids, mask = synth_context(m.config, 1, 2, 2, seed=0)
    with pytest.raises(RuntimeError):
        m.generate(ids, mask, max_new_frames=1)


def test_config_defaults_and_roundtrip(tmp_path):
    c = CSMConfig()
    assert (c.text_vocab_size, c.audio_vocab_size, c.audio_num_codebooks, c.max_seq_len) == (128256, 2051, 32, 2048)
    assert c.backbone_config.hidden_size == 2048 and c.backbone_config.head_dim == 64
    assert c.decoder_config.hidden_size == 1024 and c.decoder_config.head_dim == 128
    assert c.decoder_config.max_position_embeddings == 32 and c.backbone_config.max_position_embeddings == 2048
    assert sum(int(torch.tensor(s).prod()) for _, s, _ in state_dict_spec(c)) == 1552791552
    c.save_pretrained(tmp_path)
    c2 = CSMConfig.from_pretrained(tmp_path)
    assert c2.to_dict() == c.to_dict()
    assert c2.model_type == "csm"
    # transformers>=5 style nested rope_parameters
    d = c.to_dict()
    rs = d["backbone_config"].pop("rope_scaling")
    d["backbone_config"]["rope_parameters"] = dict(rope_type="llama3", rope_theta=500000.0, **{k: v for k, v in rs.items() if k != "type"})
    c3 = CSMConfig.from_dict(d)
    assert c3.backbone_config.rope_scaling["factor"] == 32.0 and c3.backbone_config.rope_theta == 500000.0


def test_state_dict_layout_and_safetensors_roundtrip(tmp_path):
    cfg = CSMConfig.tiny()
    m = CSMModel(cfg)
    keys = set(m.state_dict().keys())
    assert keys == {k for k, _, _ in state_dict_spec(cfg)}
    assert len(list(state_dict_spec(CSMConfig()))) == 187          # SURVEY.md section 5 (checkpoint/resume)
    sd = synth_state_dict(cfg, seed=3)
    m.load_state_dict(sd)
    m.save_pretrained(tmp_path)
    m2 = CSMModel.from_pretrained(str(tmp_path))
    for k, v in m.state_dict().items():
        assert torch.equal(v, m2.state_dict()[k])


def test_synth_is_deterministic_and_device_independent_by_construction():
    a = hash_uniform("x", 1000, 7)
    b = hash_uniform("x", 1000, 7, chunk=128)
    assert torch.equal(a, b) and a.abs().max() < 1 and abs(float(a.mean())) < 0.1
    assert not torch.equal(a, hash_uniform("y", 1000, 7))
    cfg = CSMConfig.tiny()
    ids, mask = synth_context(cfg, 2, 4, 6, seed=1)
    assert ids.shape == (2, 10, 33) and int(mask[:, :4, 32].sum()) == 8 and int(mask[:, 4:, :32].sum()) == 2 * 6 * 32
    assert int(ids[:, :, :32].max()) < cfg.audio_vocab_size and int(ids[:, :, 32].max()) < cfg.text_vocab_size


def test_rope_table_matches_oracle_formula():
    from oracle import csm_oracle as O
    cfg = CSMConfig()
    for lc in (cfg.backbone_config, cfg.decoder_config):
        inv = llama3_inv_freq(lc.head_dim, lc.rope_theta, lc.rope_scaling)
        assert torch.equal(inv, O.llama3_inv_freq(lc.head_dim, lc.rope_theta, lc.rope_scaling))
        cos, sin = rope_tables(lc, 40)
        c2, s2 = O.rope_cos_sin(inv, torch.arange(40)[None], torch.float32)
        assert torch.equal(cos, c2[0, :, : lc.head_dim // 2]) and torch.equal(sin, s2[0, :, : lc.head_dim // 2])
    assert abs(float(llama3_inv_freq(64, 500000.0, cfg.backbone_config.rope_scaling)[1]) - 0.6636) < 1e-3


def test_api_surface_matches_reference_signatures():
    import inspect
    g = inspect.signature(CSMModel.generate).parameters
    # the reference's parameters, in order; extensions (seed) are keyword-only so positional callers are unaffected
    pos = [k for k, v in g.items() if v.kind == v.POSITIONAL_OR_KEYWORD]
    assert pos[1:] == ["input_ids", "attention_mask", "max_new_frames", "temperature", "topk", "use_cache", "stop_on_all_zeros"]
    assert all(v.kind == v.KEYWORD_ONLY and v.default in (None, False) for k, v in g.items() if k not in pos)
    assert (g["max_new_frames"].default, g["temperature"].default, g["topk"].default) == (100, 1.0, 50)
    f = inspect.signature(CSMModel.generate_frame).parameters
    f = {k: v for k, v in f.items() if v.kind != v.KEYWORD_ONLY}      # extensions are keyword-only
    assert list(f)[1:] == ["input_ids", "attention_mask", "position_ids", "temperature", "topk", "past_key_values",
                           "use_cache", "output_attentions", "output_hidden_states", "return_dict"]
    fw = inspect.signature(CSMModel.forward).parameters
    assert list(fw)[1:] == ["input_ids", "attention_mask", "position_ids", "past_key_values", "use_cache", "output_attentions",
                            "output_hidden_states", "return_dict", "temperature", "topk", "generate_frame", "labels"]
    assert hasattr(csm_hf_amd, "sample_topk") and hasattr(CSMModel, "setup_caches") and hasattr(CSMModel, "reset_caches")


def test_kv_start_validation():
    m = torch.ones(2, 5, 33, dtype=torch.int32)
    m[1, :2] = 0
    assert CSMModel._kv_starts(m, 2, 5) == [0, 2]
    m[0, 3] = 0
    with pytest.raises(ValueError):
        CSMModel._kv_starts(m, 2, 5)


def test_header_is_plain_c(tmp_path):
    """include/csm_hip.h is the drop-in boundary: it must compile as C (no C++ / torch types)."""
    import shutil
    import subprocess
    gcc = shutil.which("gcc")
    if gcc is None:
        import pytest
        pytest.skip("no gcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "h.c"
    src.write_text('#include "csm_hip.h"\nint main(void) { return 0; }\n')
    r = subprocess.run([gcc, "-std=c99", "-fsyntax-only", "-Wall", "-I", os.path.join(root, "include"), str(src)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr



def test_graft_entry_build_runs():
    """the driver's build check: __graft_entry__.build() compiles (or finds) the library and its ABI assertion tracks
    engine.ABI_VERSION -- it asserted a literal 3 after the ABI had moved to 4 and failed the check for a while"""
    import __graft_entry__ as g
    g.build()


def test_every_engine_option_is_documented_in_the_header():
    """include/csm_hip.h calls its option list COMPLETE: every name csm_set_option accepts (csrc/engine.hip) must appear there."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(root, "csm-hf_amd", "csrc", "engine.hip")).read()
    body = src[src.index('extern "C" int csm_set_option'):]
    body = body[:body.index("\n}\n")]
    names = set(re.findall(r'strcmp\(name, "([a-z_0-9]+)"\)', body))
    assert len(names) > 40, names
    hdr = open(os.path.join(root, "include", "csm_hip.h")).read()
    missing = sorted(n for n in names if f'"{n}"' not in hdr)
    assert not missing, f"options accepted by csm_set_option but not documented in include/csm_hip.h: {missing}"
