"""Extracts small DATA fixtures (inputs + stored results, no source text) from the reference checkout
into tests/golden/.  Run in the build container (needs /root/reference); the GPU box only sees the
committed .npz files.

Sources (paths relative to /root/reference):
  data/ohashi_csv/ohashi_OGTT.csv, ohashi_subjectinfo.csv   CC-BY-4.0, Ohashi et al. 2018 (see ATTRIBUTION.md)
      unit conversions as c-peptide/00-prepare-data.jl:30-31
  source_data/cude_neural_parameters.jld2     25 trained 2->4->4->1 weight vectors + their 57 training betas
      (written by c-peptide/02-conditional.jl:44-50)
  source_data/neural_network_parameters.jld2  legacy 2->6->6->1 weight vector (67 doubles)
  suppression/results/lambda=0.0.jld2         25 x 67 weights (4->3x5->1), group_data 3x8x37, losses, correlations
      (written by suppression/suppression.jl:76-91)
JLD2 stores these arrays uncompressed and contiguous; offsets were located by byte scan.
"""
import os
import numpy as np
import pandas as pd

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def f64(path, off, n):
    with open(os.path.join(REF, path), "rb") as fh:
        fh.seek(off)
        return np.frombuffer(fh.read(8 * n), dtype="<f8").copy()


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
    nn = np.stack([f64("source_data/cude_neural_parameters.jld2", 5688 + 400 * k, 37) for k in range(25)])
    betas = np.stack([f64("source_data/cude_neural_parameters.jld2", 16024 + 560 * k, 57) for k in range(25)])
    legacy = f64("source_data/neural_network_parameters.jld2", 816, 67)
    np.savez_compressed(os.path.join(OUT, "ohashi_cude.npz"), subject_no=ogtt["No"].to_numpy(), glucose=glucose,
                        cpeptide=cpeptide, ages=ages, t2dm=(types == "T2DM"), types=types,
                        timepoints=np.array([0.0, 30.0, 60.0, 90.0, 120.0]), nn_2x4x4x1=nn, betas_train=betas,
                        nn_2x6x6x1_legacy=legacy)
    sp = "suppression/results/lambda=0.0.jld2"
    snn = np.stack([f64(sp, 5536 + 640 * k, 67) for k in range(25)])
    group = f64(sp, 21552, 888).reshape(37, 8, 3).transpose(2, 1, 0)          # column-major 3 x 8 x 37
    np.savez_compressed(os.path.join(OUT, "suppression_lambda0.npz"), nn_4x3x5x1=snn, group_data=group,
                        correlations=f64(sp, 40520, 25), losses=f64(sp, 40824, 25),
                        timepoints=np.linspace(0.0, 30.0, 8))
    # dose-response table the reference's symbolic regression was run on (30 exp(beta) x 30 dG values)
    prod = pd.read_csv(os.path.join(REF, "data/ohashi_production.csv"))
    np.savez_compressed(os.path.join(OUT, "ohashi_production.npz"), beta=prod["Beta"].to_numpy(dtype=np.float64),
                        glucose=prod["Glucose"].to_numpy(dtype=np.float64),
                        production=prod["Production"].to_numpy(dtype=np.float64))
    print(glucose.shape, nn.shape, betas.shape, snn.shape, group.shape)
    print("losses", f64(sp, 40824, 25)[:4], "group_data[:,0,0]", group[:, 0, 0])


if __name__ == "__main__":
    main()
