"""JLD2 reader / writer (conditional-ude_amd/cude/jld2.py) against a file JLD2.jl itself wrote: the reference's
`source_data/ude_neural_parameters.jld2` (1455 bytes, committed as a data fixture) and, when the reference
checkout is present (the build container), every other .jld2 file it ships."""
import glob
import os

import numpy as np
import pytest

from cude import jld2

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FIXTURE = os.path.join(GOLD, "ude_neural_parameters.jld2")
REF = "/root/reference"


def test_lookup3_known_answers():
    # Bob Jenkins' lookup3.c self-test values for hashlittle()
    assert jld2.lookup3(b"") == 0xDEADBEEF
    assert jld2.lookup3(b"Four score and seven years ago", 0) == 0x17770551
    assert jld2.lookup3(b"Four score and seven years ago", 1) == 0xCD628161


def test_reader_decodes_a_file_written_by_jld2_jl():
    f = jld2.JLD2File(FIXTURE)
    assert f.julia_version == "1.11.1"
    assert f.keys() == ["width", "depth", "parameters"]
    assert f["width"] == 6 and f["depth"] == 2
    p = f["parameters"]
    assert p.dtype == np.float64 and p.shape == (61,)            # 1 -> 6 -> 6 -> 1 network: 12 + 42 + 7
    assert p[0] == 0.11227807183121734 and np.all(np.isfinite(p))   # bytes 43 b1 92 76 41 be bc 3f at offset 816
    with pytest.raises(KeyError):
        f["betas"]


def test_writer_regenerates_the_reference_file_byte_for_byte():
    content = jld2.load(FIXTURE)
    out = os.path.join(os.environ.get("TMPDIR", "/tmp"), "cude_jld2_regen.jld2")
    jld2.save(out, content, julia_version="1.11.1")
    assert open(out, "rb").read() == open(FIXTURE, "rb").read()
    os.remove(out)


def test_round_trip_of_a_training_checkpoint(tmp_path):
    rng = np.random.default_rng(0)
    ckpt = {"width": 6, "depth": 2, "parameters": [rng.standard_normal(67) for _ in range(25)],
            "betas": rng.standard_normal((57, 25)), "best_model_index": 14, "sigma": 0.25,
            "group_data": rng.standard_normal((3, 8, 37)), "subject_numbers": np.arange(1, 58),
            "a_rather_long_entry_name_for_the_link_message": 1.0}
    path = str(tmp_path / "ckpt.jld2")
    jld2.save(path, ckpt)
    back = jld2.load(path)
    assert list(back) == list(ckpt)
    assert back["width"] == 6 and back["best_model_index"] == 14 and back["sigma"] == 0.25
    assert np.array_equal(back["parameters"], np.stack(ckpt["parameters"], axis=1))      # one column per vector
    assert np.array_equal(back["betas"], ckpt["betas"]) and np.array_equal(back["group_data"], ckpt["group_data"])
    assert back["subject_numbers"].dtype == np.int64 and np.array_equal(back["subject_numbers"], np.arange(1, 58))
    # column-major on disk: Julia's group_data[1, 2, 1] is the second double of the dataset
    raw = open(path, "rb").read()
    at = raw.index(ckpt["group_data"][0, 0, 0].tobytes())
    assert np.frombuffer(raw, "<f8", 2, at)[1] == ckpt["group_data"][1, 0, 0]
    # a flipped bit in an object header is detected by its checksum
    bad = bytearray(raw)
    bad[jld2.HEADER_BYTES + 48 + 20] ^= 1
    (tmp_path / "bad.jld2").write_bytes(bytes(bad))
    with pytest.raises(ValueError):
        jld2.load(str(tmp_path / "bad.jld2"))
    with pytest.raises(TypeError):
        jld2.save(path, {"name": "text"})


def test_api_checkpoint_helpers(tmp_path):
    """save_parameters / load_parameters mirror the jldopen blocks of c-peptide/02-conditional.jl:44-57."""
    from cude import api
    ref = api.load_parameters(FIXTURE)
    assert ref.width == 6 and ref.depth == 2 and ref.parameters.shape == (61,) and ref.betas is None
    rng = np.random.default_rng(1)
    nets = [rng.standard_normal(37) for _ in range(25)]
    betas = [rng.standard_normal(57) for _ in range(25)]
    path = str(tmp_path / "cude_neural_parameters.jld2")
    api.save_parameters(path, 4, 2, nets, betas, best_model_index=14, sigma=0.3)
    back = api.load_parameters(path)
    assert back.width == 4 and back.depth == 2 and back.best_model_index == 14 and back.extra == {"sigma": 0.3}
    assert len(back.parameters) == 25 and all(np.array_equal(a, b) for a, b in zip(back.parameters, nets))
    assert all(np.array_equal(a, b) for a, b in zip(back.betas, betas))
    assert api.load_data(path)["width"] == 4


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference checkout not present")
def test_every_reference_file_is_readable_and_simple_ones_regenerate(tmp_path):
    files = sorted(glob.glob(os.path.join(REF, "**", "*.jld2"), recursive=True))
    assert len(files) >= 8
    regenerated = 0
    for path in files:
        f = jld2.JLD2File(path)
        content = {k: f[k] for k in f.keys()}
        assert content
        plain = all(isinstance(v, (int, float)) or (isinstance(v, np.ndarray) and v.dtype.kind in "fi")
                    for v in content.values())
        if plain:                      # scalars and dense arrays only: the writer's schema
            out = str(tmp_path / "regen.jld2")
            jld2.save(out, content, julia_version=f.julia_version)
            assert open(out, "rb").read() == open(path, "rb").read(), path
            regenerated += 1
    assert regenerated >= 3
    g = np.load(os.path.join(GOLD, "ohashi_cude.npz"))
    cude = jld2.load(os.path.join(REF, "source_data", "cude_neural_parameters.jld2"))
    assert np.array_equal(np.stack(cude["parameters"]), g["nn_2x4x4x1"])
    assert np.array_equal(np.stack(cude["betas"]), g["betas_train"]) and cude["best_model_index"] == 14
    prepared = jld2.load(os.path.join(REF, "data", "ohashi.jld2"))
    assert prepared["train"]["glucose"].shape == (82, 5) and prepared["test"]["types"][0] in ("NGT", "IGT", "T2DM")
