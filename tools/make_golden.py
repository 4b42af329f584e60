"""Extracts small DATA fixtures (inputs + stored results, no source text) from the reference checkout
into tests/golden/.  Run in the build container (needs /root/reference); the GPU box only sees the
committed files.

Sources (paths relative to /root/reference):
  data/ohashi_csv/ohashi_OGTT.csv, ohashi_subjectinfo.csv   CC-BY-4.0, Ohashi et al. 2018 (see ATTRIBUTION.md)
      unit conversions as c-peptide/00-prepare-data.jl:30-31
  data/ohashi.jld2                            the reference's own prepared train / test split of the same subjects
      (written by c-peptide/00-prepare-data.jl): used to cross-check the unit conversions and for the split
  source_data/cude_neural_parameters.jld2     25 trained 2->4->4->1 weight vectors + their 57 training betas
      (written by c-peptide/02-conditional.jl:44-50)
  source_data/neural_network_parameters.jld2  legacy 2->6->6->1 weight vector (67 doubles)
  source_data/ude_neural_parameters.jld2      decoded content + SHA-256 of its 1455 bytes: target of the JLD2 writer test
  suppression/results/lambda=0.0.jld2         25 x 67 weights (4->3x5->1), group_data 3x8x37, losses, correlations
      (written by suppression/suppression.jl:76-91)
  suppression/results/lambda=0.001|0.01|0.010000000000000002|0.1.jld2   85 more trained 4->3x5->1 networks on the same
      data, their final objectives and validation objectives (round 4: soft pins for every stored network)
  source_data/advi/cude_result_{1..25}.jld2   25 more (network, 57 betas) pairs of the 2->4->4->1 c-peptide model (no
      producing script in the reference tree; same subjects and order as cude_neural_parameters.jld2, checked below)
  data/ohashi_production.csv                  dose-response table of the symbolic regression
The .jld2 files are decoded with the build's own reader (conditional-ude_amd/cude/jld2.py).
"""
import os
import sys

import numpy as np
import pandas as pd

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, os.path.join(ROOT, "conditional-ude_amd"))
from cude import jld2  # noqa: E402


def _save(name, **arrays):
    """np.savez_compressed, skipped when the committed file already holds exactly this content (zip members carry a
    time stamp, so a rewrite would change the bytes of an unchanged fixture)."""
    path = os.path.join(OUT, name)
    if os.path.exists(path):
        old = np.load(path)
        if set(old.files) == set(arrays) and all(
                np.asarray(arrays[k]).shape == old[k].shape and np.array_equal(np.asarray(arrays[k]), old[k], equal_nan=(
                    old[k].dtype.kind == "f")) for k in arrays):
            return
    np.savez_compressed(path, **arrays)


def main():
    os.makedirs(OUT, exist_ok=True)
    ogtt = pd.read_csv(os.path.join(REF, "data/ohashi_csv/ohashi_OGTT.csv"), sep=";").dropna()
    info = pd.read_csv(os.path.join(REF, "data/ohashi_csv/ohashi_subjectinfo.csv"), sep=";")
    info = info[info["No"].isin(ogtt["No"])]
    assert list(info["No"]) == list(ogtt["No"])
    glucose = ogtt.iloc[:, 1:6].to_numpy(dtype=np.float64) * 0.0551
    cpeptide = ogtt.iloc[:, 11:16].to_numpy(dtype=np.float64) * 0.3311
    ages = info["age"].to_numpy(dtype=np.float64)
    types = info["type"].to_numpy(dtype=str)
    subject_no = ogtt["No"].to_numpy()
    # the reference's prepared data set must be the same numbers (pins the unit conversions and the subject filter)
    prepared = jld2.load(os.path.join(REF, "data/ohashi.jld2"))
    for part in ("train", "test"):
        p = prepared[part]
        rows = np.array([np.flatnonzero(subject_no == s)[0] for s in p["subject_numbers"]])
        assert np.allclose(p["glucose"], glucose[rows], rtol=1e-12) and np.allclose(p["cpeptide"], cpeptide[rows], rtol=1e-12)
        assert np.array_equal(p["ages"], ages[rows]) and list(p["types"]) == list(types[rows])
    cude = jld2.load(os.path.join(REF, "source_data/cude_neural_parameters.jld2"))
    assert cude["width"] == 4 and cude["depth"] == 2
    nn = np.stack(cude["parameters"])
    betas = np.stack(cude["betas"])
    legacy = jld2.load(os.path.join(REF, "source_data/neural_network_parameters.jld2"))["parameters"]
    # covariate model (3 -> 4 -> 4 -> 1, inputs [dG, exp(beta), age]; c-peptide/07-covariate-inclusion.jl:59-65) and
    # the run of the 2 -> 4 -> 4 -> 1 model stored as cude_neural_parameters_sigma.jld2
    cov = jld2.load(os.path.join(REF, "source_data/cude_covariate_neural_parameters_2.jld2"))
    sig = jld2.load(os.path.join(REF, "source_data/cude_neural_parameters_sigma.jld2"))
    assert cov["width"] == 4 and cov["depth"] == 2 and sig["width"] == 4 and sig["depth"] == 2
    _save("ohashi_cude.npz", subject_no=subject_no, glucose=glucose,
                        cpeptide=cpeptide, ages=ages, t2dm=(types == "T2DM"), types=types,
                        timepoints=np.array([0.0, 30.0, 60.0, 90.0, 120.0]), nn_2x4x4x1=nn, betas_train=betas,
                        nn_2x6x6x1_legacy=legacy, train_subject_numbers=prepared["train"]["subject_numbers"],
                        test_subject_numbers=prepared["test"]["subject_numbers"],
                        best_model_index=np.int64(cude["best_model_index"]),          # 1-based, as stored
                        nn_3x4x4x1_cov=np.stack(cov["parameters"]), betas_train_cov=np.stack(cov["betas"]),
                        best_model_index_cov=np.int64(cov["best_model_index"]),
                        nn_2x4x4x1_sigma=np.stack(sig["parameters"]), betas_train_sigma=np.stack(sig["betas"]),
                        best_model_index_sigma=np.int64(sig["best_model_index"]))
    supp = jld2.load(os.path.join(REF, "suppression/results/lambda=0.0.jld2"))
    snn = np.stack(supp["neural_parameters"])
    group = supp["group_data"]
    _save("suppression_lambda0.npz", nn_4x3x5x1=snn, group_data=group,
                        correlations=supp["correlations"], losses=supp["losses"], timepoints=np.linspace(0.0, 30.0, 8),
                        gt_sup_param=supp["gt_sup_param"],
                        **{k: supp[k] for k in ("validation_data", "validation_data_nonoise", "gt_validation_param",
                                                "gt_validation_param_nonoise", "losses_valid", "losses_valid_nonoise",
                                                "correlations_valid", "correlations_valid_nonoise")})
    # the most regularised run (lambda = 1): the networks collapse to a constant output, the loss no longer depends on
    # the (unsaved) conditional parameters and the stored final losses become known answers of suppression_loss
    s1 = jld2.load(os.path.join(REF, "suppression/results/lambda=1.0.jld2"))
    assert np.array_equal(s1["group_data"], group) and s1["λ"] == 1.0
    # ... and so do the stored validation objectives (suppression_loss with lambda = 0 on the two validation sets,
    # suppression_model.jl:179-187), whose data sets are those of the lambda = 0 fixture
    assert all(np.array_equal(s1[k], supp[k]) for k in ("validation_data", "validation_data_nonoise"))
    _save("suppression_lambda1.npz", nn_4x3x5x1=np.stack(s1["neural_parameters"]),
                        losses=s1["losses"], losses_valid=s1["losses_valid"],
                        losses_valid_nonoise=s1["losses_valid_nonoise"])
    # the runs in between (round 4): live networks, the same three data sets -- soft pins (the per-subject global
    # minimum over theta of the data term must be <= and close to the stored objective)
    mids = {}
    for tag, fname in (("0.001", "lambda=0.001.jld2"), ("0.01", "lambda=0.01.jld2"),
                       ("0.01b", "lambda=0.010000000000000002.jld2"), ("0.1", "lambda=0.1.jld2")):
        sm = jld2.load(os.path.join(REF, "suppression/results", fname))
        assert np.array_equal(sm["group_data"], group)
        assert all(np.array_equal(sm[k], supp[k]) for k in ("validation_data", "validation_data_nonoise"))
        mids["lam_" + tag] = np.float64(sm["λ"])
        mids["nn_" + tag] = np.stack(sm["neural_parameters"])
        for k in ("losses", "losses_valid", "losses_valid_nonoise", "correlations", "correlations_valid",
                  "correlations_valid_nonoise"):
            mids[k + "_" + tag] = sm[k]
    _save("suppression_lambda_mid.npz", **mids)
    # 25 more stored (network, betas) pairs of the c-peptide model
    advi = [jld2.load(os.path.join(REF, "source_data/advi", f"cude_result_{k}.jld2")) for k in range(1, 26)]
    assert all(a["width"] == 4 and a["depth"] == 2 for a in advi)
    _save("advi_cude.npz", nn_2x4x4x1=np.stack([a["parameters"] for a in advi]),
          betas_train=np.stack([a["betas"] for a in advi]))
    # dose-response table the reference's symbolic regression was run on (30 exp(beta) x 30 dG values)
    prod = pd.read_csv(os.path.join(REF, "data/ohashi_production.csv"))
    _save("ohashi_production.npz", beta=prod["Beta"].to_numpy(dtype=np.float64),
                        glucose=prod["Glucose"].to_numpy(dtype=np.float64),
                        production=prod["Production"].to_numpy(dtype=np.float64))
    # external data set of c-peptide/04-symreg-external.jl (20 subjects, 14 irregular time points from -10 min)
    fuj = jld2.load(os.path.join(REF, "data/fujita.jld2"))
    _save("fujita.npz", glucose=fuj["glucose"], cpeptide=fuj["cpeptide"],
                        timepoints=fuj["timepoints"].astype(np.float64), ages=fuj["ages"].astype(np.float64))
    # a file JLD2.jl itself wrote, as decoded content + the digest of its bytes (the file itself is not copied): the
    # writer test regenerates the bytes from the content and must hit the digest
    import hashlib
    src = os.path.join(REF, "source_data/ude_neural_parameters.jld2")
    raw = open(src, "rb").read()
    f = jld2.JLD2File(src)
    _save("jld2_known_file.npz", width=np.int64(f["width"]),
                        depth=np.int64(f["depth"]), parameters=f["parameters"], julia_version=f.julia_version,
                        n_bytes=np.int64(len(raw)), sha256=hashlib.sha256(raw).hexdigest())
    # the training checkpoints (Vector{Vector{Float64}} entries): digests only -- their content is already in
    # ohashi_cude.npz (nn_2x4x4x1 / betas_train) or regenerated from a decode when the reference is present
    names, sizes, digests = [], [], []
    for name in ("cude_neural_parameters", "cude_neural_parameters_sigma", "cude_covariate_neural_parameters_2"):
        raw = open(os.path.join(REF, "source_data", name + ".jld2"), "rb").read()
        names.append(name), sizes.append(len(raw)), digests.append(hashlib.sha256(raw).hexdigest())
    _save("jld2_checkpoint_digests.npz", names=np.array(names), n_bytes=np.array(sizes, dtype=np.int64),
                        sha256=np.array(digests))
    print(glucose.shape, nn.shape, betas.shape, snn.shape, group.shape, "best model", cude["best_model_index"])
    print("losses", supp["losses"][:4], "group_data[:,0,0]", group[:, 0, 0])


if __name__ == "__main__":
    main()
