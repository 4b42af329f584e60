"""Decodes the plotted DATA of the reference's vector figures into tests/golden/figure_traces.npz.

The reference has no tests and stores no c-peptide objective, but its committed figures are CairoMakie vector
graphics whose paths are the numbers its scripts computed, quantised by Cairo to 1/256 px:

  figures/revision/supplementary/model_fit_train_median.svg   (c-peptide/02-conditional.jl:444-524)
      panels a-c: for the median-error training subject of each glucose-tolerance type, the simulated plasma
      c-peptide on 0:0.1:120 min (solid line), its five measurements (markers) and two dotted simulations at the
      confidence bounds of beta; panel d: the fitted objective (sum of squared errors) of all 82 training-data
      subjects, drawn type by type in subject order
  figures/revision/supplementary/model_fit_test_all.svg       (c-peptide/02-conditional.jl:532-588)
      one panel per test subject (35, in subject order): model fit, confidence-bound simulations, measurements
  figures/revision/supplementary/model_fit_test_covariate_median.svg   (c-peptide/07-covariate-inclusion.jl)
      the test-median panels and the 35 test objectives of the covariate model (3 -> 4 -> 4 -> 1 network)
  figures/revision/figure_6/figure_6.svg                      (c-peptide/04-symreg-external.jl:72-170)
      the symbolic model (production 1.78 dG / (dG + k)) on the external data set (14 irregular time points from
      -10 min): quartile subjects' simulations on -10:0.1:240 min, their measurements, the 20 fitted objectives

  figures/revision/supplementary/likelihood_curves.svg        (c-peptide/02-conditional.jl:361-423)
      the likelihood profile of EVERY subject (82 training-data + 35 test subjects, drawn in that order): 1000 values
      of (SSE_i(beta_i + d) - SSE_i(beta_i)) / (2 sigma_i^2) on d = range(-10, 10, 1000) (src/likelihood-profiles.jl:
      4-17), clipped by Cairo to the axis limits 0 ... 10; the dashed line is the 7.16 threshold

Only pixel coordinates are stored, as integers in units of 1/256 px (exactly what Cairo wrote): poly-line vertices
and marker centres in drawing order.  Mapping them to data units is the test's job (tests/test_figure_pins.py: the
markers are the subject's own measurements, which calibrates every axis without reading a tick label).  Runs only
where /root/reference exists.
"""
import os
import re
import sys

import numpy as np

REF = "/root/reference/figures/revision"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden", "figure_traces.npz")
# stroke / fill colours of the three glucose-tolerance types (COLORS in c-peptide/02-conditional.jl:67-71)
TYPES = {"NGT": "rgb(0.392157%, 39.607844%, 61.56863%)", "IGT": "rgb(78.823531%, 30.588236%, 0%)",
         "T2DM": "rgb(0.392157%, 47.058824%, 31.37255%)"}
BLACK = "rgb(0%, 0%, 0%)"

PATH = re.compile(r'<path ([^>]*?)d="([^"]*)"([^>]*)/>')
NUM = re.compile(r"-?\d+\.?\d*")


def _q(v):
    """pixel coordinates -> integer multiples of 1/256 px (asserting that they are)."""
    k = np.rint(np.asarray(v) * 256.0)
    assert np.max(np.abs(k - np.asarray(v) * 256.0)) < 2e-3
    return k.astype(np.int32)


def primitives(svg_file):
    """Drawing-order list of ("line", attrs, (n,2)) and ("marker", attrs, centre(2,), radius) of one Cairo SVG."""
    text = open(svg_file).read()
    body = text[text.index("</defs>"):]
    out = []
    for before, d, after in PATH.findall(body):
        attrs = before + " " + after
        v = np.array([float(x) for x in NUM.findall(d)]).reshape(-1, 2)
        if 'fill="none"' in attrs and d.count("L") >= 10 and "C" not in d:
            out.append(("line", attrs, v))
        elif d.count("C") in (2, 4) and "L" not in d:
            # a Makie circle marker: one closed path of two or four Bezier arcs; centre = middle of the control-point box
            out.append(("marker", attrs, 0.5 * (v.min(axis=0) + v.max(axis=0)), 0.5 * float(np.ptp(v[:, 0]))))
    return out


def decode_type_panels(svg_file):
    """{type: fit, bounds[], markers(5,2), objectives(m,2)} of a figure with one coloured panel per type."""
    out = {t: dict(fit=None, bounds=[], markers=[], objectives=[]) for t in TYPES}
    for prim in primitives(svg_file):
        for t, colour in TYPES.items():
            if colour not in prim[1]:
                continue
            rec = out[t]
            if prim[0] == "line":
                if "stroke-dasharray" in prim[1]:
                    rec["bounds"].append(prim[2])
                elif rec["fit"] is None:                                 # solid, linewidth 2 or 1.5
                    rec["fit"] = prim[2]
            elif 'fill-opacity="1"' in prim[1] and prim[3] > 1.5:      # markersize 5 / 7: the measurements (+ legend)
                rec["markers"].append(prim[2])
            elif 'fill-opacity="0.8"' in prim[1]:                       # markersize 3: objective scatter
                rec["objectives"].append(prim[2])
    for t in TYPES:
        out[t]["markers"] = np.array(out[t]["markers"][:5])             # the legend's marker is drawn last
        out[t]["objectives"] = np.array(out[t]["objectives"]).reshape(-1, 2)
    return out


def decode_subject_panels(svg_file):
    """[dict(fit, bounds[], markers(5,2))] of a figure with one black panel per subject, in drawing order: per panel
    the dotted bound lines, then the solid fit, then five markers."""
    panels, cur = [], dict(fit=None, bounds=[], markers=[])
    for prim in primitives(svg_file):
        if BLACK not in prim[1]:
            continue
        if prim[0] == "line":
            if len(cur["markers"]) == 5:                                 # first line of the next panel
                panels.append(cur)
                cur = dict(fit=None, bounds=[], markers=[])
            if "stroke-dasharray" in prim[1]:
                cur["bounds"].append(prim[2])
            elif 'stroke-width="2"' in prim[1]:
                cur["fit"] = prim[2]
        elif prim[3] > 2.0 and cur["fit"] is not None and len(cur["markers"]) < 5:
            cur["markers"].append(prim[2])
    if cur["fit"] is not None and len(cur["markers"]) == 5:
        panels.append(cur)
    for p in panels:
        p["markers"] = np.array(p["markers"])
    return panels


def decode_quartile_panels(svg_file, n_obs=14):
    """figure_6 (c-peptide/04-symreg-external.jl:72-170): three panels, each a solid model line in the NGT colour,
    the subject's n_obs measurements as black markers and up to two dotted bound lines; then the objective scatter."""
    panels, objectives = [], []
    for prim in primitives(svg_file):
        if prim[0] == "line" and TYPES["NGT"] in prim[1]:
            if "stroke-dasharray" in prim[1]:
                panels[-1]["bounds"].append(prim[2])
            else:
                panels.append(dict(fit=prim[2], bounds=[], markers=[]))
        elif prim[0] == "marker" and BLACK in prim[1] and prim[3] > 1.5 and panels and len(panels[-1]["markers"]) < n_obs:
            panels[-1]["markers"].append(prim[2])
        elif prim[0] == "marker" and TYPES["NGT"] in prim[1] and prim[3] < 1.5:
            objectives.append(prim[2])
    for p in panels:
        p["markers"] = np.array(p["markers"])
    return panels, np.array(objectives)


def decode_small_markers_by_type(svg_file):
    """{type: (m,2)} centres of the markersize-3 scatter of a figure, per type colour, in drawing order."""
    out = {t: [] for t in TYPES}
    for prim in primitives(svg_file):
        if prim[0] == "marker" and prim[3] < 1.5:
            for t, colour in TYPES.items():
                if colour in prim[1]:
                    out[t].append(prim[2])
    return {t: np.array(v) for t, v in out.items()}


PROFILE_COLOURS = {"rgb(90.196079%, 62.352943%, 0%)": 0,        # identifiable
                   "rgb(0%, 44.705883%, 69.803923%)": 1,        # practically unidentifiable
                   "rgb(80.000001%, 47.450981%, 65.490198%)": 2}  # unidentifiable


def decode_likelihood_curves(svg_file):
    """The 117 profile curves in drawing (= subject) order.  Cairo cuts every poly-line at the clip rectangle of the
    axis, so a curve arrives as one or more left-to-right pieces (a profile that dips below 0 at its centre -- the
    reference's optimum is not exact -- or exceeds 10 is split); a piece that starts left of the end of the previous
    piece begins the next curve.  Returns (list of (class, [pieces (n, 2)]), threshold line (2, 2))."""
    text = open(svg_file).read()
    body = text[text.index("</defs>"):]
    curves, hline = [], None
    for before, d, after in PATH.findall(body):
        attrs = before + " " + after
        v = np.array([float(x) for x in NUM.findall(d)]).reshape(-1, 2)
        if "stroke-dasharray" in attrs and 'stroke-width="2.5"' in attrs and hline is None:
            hline = v
            continue
        m = re.search(r'stroke="([^"]*)"', attrs)
        if not m or m.group(1) not in PROFILE_COLOURS or 'stroke-width="1.5"' not in attrs or 'fill="none"' not in attrs:
            continue
        if v[:, 0].min() > 320:                                      # the legend's line samples
            continue
        cls = PROFILE_COLOURS[m.group(1)]
        if curves and curves[-1][0] == cls and v[0, 0] > curves[-1][1][-1][-1, 0] - 1e-9:
            curves[-1][1].append(v)
        else:
            curves.append((cls, [v]))
    return curves, hline


def main():
    arrays = {}
    # likelihood profiles of all 117 subjects: per curve its pieces' vertices, concatenated (CSR layout)
    curves, hline = decode_likelihood_curves(os.path.join(REF, "supplementary/likelihood_curves.svg"))
    assert len(curves) == 117 and hline is not None
    full = [c for c in curves if c[0] == 2][0][1]                     # an unidentifiable profile spans all of -10 ... 10
    arrays["profiles_xspan"] = _q(np.array([full[0][0, 0], full[-1][-1, 0]]))
    arrays["profiles_threshold_y"] = _q(hline[:1, 1])
    arrays["profiles_class"] = np.array([c[0] for c in curves], dtype=np.int32)
    arrays["profiles_vertices"] = _q(np.concatenate([np.concatenate(c[1]) for c in curves]))
    arrays["profiles_ptr"] = np.cumsum([0] + [sum(len(v) for v in c[1]) for c in curves]).astype(np.int64)
    print("likelihood_curves", len(curves), "curves,", int(arrays["profiles_ptr"][-1]), "vertices; classes",
          np.bincount(arrays["profiles_class"]))
    # figure_5 panel d (c-peptide/03-symreg.jl:100-112, :190-200): fitted objectives of the symbolic model for all 117
    # subjects ([train; test] order within each type)
    for t, v in decode_small_markers_by_type(os.path.join(REF, "figure_5/figure_5.svg")).items():
        arrays[f"symbolic_{t}_objectives"] = v
        print("figure_5 objectives", t, v.shape)
    panels, objectives = decode_quartile_panels(os.path.join(REF, "figure_6/figure_6.svg"))
    print("figure_6 panels", [(p["fit"].shape[0], p["markers"].shape, [b.shape[0] for b in p["bounds"]]) for p in panels],
          "objectives", objectives.shape)
    arrays["external_objectives"] = objectives
    for i, p in enumerate(panels):
        arrays[f"external_{i}_fit"] = _q(p["fit"])
        arrays[f"external_{i}_markers"] = p["markers"]
        for k, b in enumerate(p["bounds"]):
            arrays[f"external_{i}_bound{k}"] = _q(b)
    for tag, rel in (("train", "supplementary/model_fit_train_median.svg"),
                     ("covariate", "supplementary/model_fit_test_covariate_median.svg")):
        for t, rec in decode_type_panels(os.path.join(REF, rel)).items():
            arrays[f"{tag}_{t}_fit"] = _q(rec["fit"])
            arrays[f"{tag}_{t}_markers"] = rec["markers"]              # centres: half-units, kept as float64
            arrays[f"{tag}_{t}_objectives"] = rec["objectives"]
            for k, b in enumerate(rec["bounds"]):
                arrays[f"{tag}_{t}_bound{k}"] = _q(b)
            print(tag, t, "fit", rec["fit"].shape, "markers", rec["markers"].shape, "objectives",
                  rec["objectives"].shape, "bounds", [b.shape for b in rec["bounds"]])
    panels = decode_subject_panels(os.path.join(REF, "supplementary/model_fit_test_all.svg"))
    print("test_all panels", len(panels), "fit points", [p["fit"].shape[0] for p in panels][:8], "bounds",
          [len(p["bounds"]) for p in panels])
    for i, p in enumerate(panels):
        arrays[f"testall_{i}_fit"] = _q(p["fit"])
        arrays[f"testall_{i}_markers"] = p["markers"]
        for k, b in enumerate(p["bounds"]):
            arrays[f"testall_{i}_bound{k}"] = _q(b)
    if "--dry" not in sys.argv:
        np.savez_compressed(OUT, **arrays)
        print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
