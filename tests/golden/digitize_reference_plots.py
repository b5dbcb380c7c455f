#!/usr/bin/env python3
"""Objective extraction of the convergence histories the reference ships only as FIGURES (build container only: reads
/root/reference; the numbers it prints are what tests/golden/make_reference_fixture.py stores in the fixtures).

    examples/cylinder/newton/Re40_fixed_point/residual.png      (4500 x 2100, matplotlib, 300 dpi)
    examples/thermosyphon/baseflow/residual.png                 (same script)

Both were drawn by the plot_residuals.py next to them from a lightkrylov.log that is not shipped: left axes = Newton residual at
the start of every Newton step ('ko-', markersize 10), right axes = GMRES residuals per Newton step (step 0 = "init step", then the
inner steps; colours b, g, r, c, m, y, k and markers o, s, ^, D, v, <, > in that order).  What is measured here:

  * axes box        = the longest runs of spine-black pixels (rows / columns);
  * y calibration   = the major tick marks on the left spine (the long ticks protruding to the left of the box).  They are one
                      decade apart; ONE label is read by eye per axes (the top major tick: 1e-2 in the cylinder figure, 1e0 in the
                      thermosyphon one) -- everything else follows from pixel rows.  The straight-line fit through all major ticks
                      gives pixels per decade and its residual (< 0.5 px) is part of the error bar;
  * data points     = per series colour: mask -> binary erosion by a disc wider than the connecting line (kills the line, keeps the
                      marker cores) -> connected components -> centroids, ordered by x.  Circle, square and diamond markers are
                      centred on their data point.  Triangles are not: matplotlib's '^' has its data point at the centre of the
                      bounding box, erosion shrinks a triangle towards its incentre, which lies 0.382 half-heights below that
                      centre for the (0,1), (-1,-1), (1,-1) path; the half-height is measured from the un-eroded marker.
  * error bar       = one pixel of centroid uncertainty + the calibration residual, converted with the pixels-per-decade figure,
                      and, as an independent check, the difference between the two axes wherever the same number appears in both
                      (the "init" residual of GMRES in Newton step k is the Newton residual of step k).

Usage: python3 tests/golden/digitize_reference_plots.py            prints the tables
       digitize(path) -> dict                                       used by make_reference_fixture.py
"""
import sys

import numpy as np
from PIL import Image
from scipy import ndimage

SERIES = [("b", (0, 0, 255), "o"), ("g", (0, 128, 0), "s"), ("r", (255, 0, 0), "^"), ("c", (0, 191, 191), "D"), ("m", (191, 0, 191), "v"),
          ("y", (191, 191, 0), "<"), ("k", (0, 0, 0), ">")]


def _runs(mask1d):
    """(start, stop) of the runs of True in a 1-D mask"""
    d = np.diff(np.concatenate([[0], mask1d.astype(np.int8), [0]]))
    return list(zip(np.nonzero(d == 1)[0], np.nonzero(d == -1)[0]))


def _axes_boxes(black):
    """the two axes rectangles (x0, x1, y0, y1) from the spine pixels: columns / rows that are black over > 60 % of a box side"""
    H, W = black.shape
    cols = np.nonzero(black.sum(axis=0) > 0.6 * H)[0]
    xs = [int(np.mean(c)) for c in np.split(cols, np.nonzero(np.diff(cols) > 3)[0] + 1)]          # four vertical spines
    assert len(xs) == 4, xs
    boxes = []
    for x0, x1 in ((xs[0], xs[1]), (xs[2], xs[3])):
        rows = np.nonzero(black[:, x0:x1].sum(axis=1) > 0.9 * (x1 - x0))[0]
        ys = [int(np.mean(r)) for r in np.split(rows, np.nonzero(np.diff(rows) > 3)[0] + 1)]
        assert len(ys) == 2, ys
        boxes.append((x0, x1, ys[0], ys[1]))
    return boxes


def _major_ticks(black, box):
    """pixel rows of the major ticks on the left spine: black runs in the strip just left of the box; major ticks are the long ones"""
    x0, _, y0, y1 = box
    strip = black[:, x0 - 20:x0 - 2]                 # major ticks are 3.5 pt = 14.6 px long, minor ones 2 pt = 8.3 px
    length = strip.sum(axis=1)
    rows = []
    for a, b in _runs(length >= 12):
        if y0 - 3 <= a and b <= y1 + 4:
            w = length[a:b].astype(float)
            rows.append(float(np.sum(np.arange(a, b) * w) / np.sum(w)))
    return np.array(rows)


def _markers(rgb, box, colour, erode_px, tol=40):
    x0, x1, y0, y1 = box
    sub = rgb[y0 - 40:y1 + 40, x0 - 40:x1 + 40].astype(int)
    mask = np.all(np.abs(sub - np.array(colour)) <= tol, axis=2)
    # the dashed tolerance line (1.5 pt = 6 px, drawn last) cuts through the markers that sit on it: bridge vertical gaps of up to 10 px
    mask = ndimage.binary_closing(mask, structure=np.ones((11, 1), bool))
    yy, xx = np.mgrid[-erode_px:erode_px + 1, -erode_px:erode_px + 1]
    core = ndimage.binary_erosion(mask, structure=(xx * xx + yy * yy <= erode_px * erode_px))
    lab, n = ndimage.label(core)
    out = []
    for k in range(1, n + 1):
        if np.sum(lab == k) < 12:
            continue
        cy, cx = ndimage.center_of_mass(lab == k)
        # extent of the un-eroded marker around this core: the column through the core's centre (for the triangle correction)
        col = mask[:, int(round(cx))]
        a = b = int(round(cy))
        while a > 0 and col[a - 1]:
            a -= 1
        while b < len(col) - 1 and col[b + 1]:
            b += 1
        row = mask[int(round(cy))]
        c = d = int(round(cx))
        while c > 0 and row[c - 1]:
            c -= 1
        while d < len(row) - 1 and row[d + 1]:
            d += 1
        if d - c + 1 > 60 and b - a + 1 < 60:      # a legend patch: wider than any marker, and not a marker on a horizontal line
            continue
        out.append((cx + x0 - 40, cy + y0 - 40, 0.5 * (b - a + 1)))
    out.sort()
    return out


def digitize(path, top_decade_left, top_decade_right, nseries=None):
    """returns {"newton": values, "gmres": [values per Newton step], "rel_err": one-sigma relative error of a value, ...}"""
    img = np.array(Image.open(path).convert("RGB"))
    black = np.all(img < 60, axis=2)
    left, right = _axes_boxes(black)
    res = {"boxes": (left, right)}
    cal = []
    for box, top in ((left, top_decade_left), (right, top_decade_right)):
        rows = _major_ticks(black, box)
        dec = top - np.arange(len(rows))                       # successive decades downwards from the one label that was read
        b, a = np.polyfit(dec, rows, 1)                        # row = a + b * log10(value)
        fit_res = float(np.max(np.abs(a + b * dec - rows)))
        cal.append((a, b, fit_res, len(rows)))
    res["px_per_decade"] = [-c[1] for c in cal]
    res["calibration_residual_px"] = [c[2] for c in cal]
    res["major_ticks"] = [c[3] for c in cal]

    def value(row, k):
        return 10.0 ** ((row - cal[k][0]) / cal[k][1])

    # left axes: black circles, markersize 10 (41.7 px), line width 2 pt (8.3 px): erosion radius 9 px removes line, ticks and spines
    pts = [p for p in _markers(img, left, (0, 0, 0), 9) if left[0] + 5 < p[0] < left[1] - 5 and left[2] + 5 < p[1] < left[3] - 5]
    res["newton"] = np.array([value(p[1], 0) for p in pts])
    # right axes: markersize 6 (25 px), line width 1.5 pt (6.25 px): erosion radius 5 px
    res["gmres"] = []
    for name, colour, marker in SERIES[:nseries or len(SERIES)]:
        pts = [p for p in _markers(img, right, colour, 5) if right[0] - 30 < p[0] < right[1] + 30 and right[2] - 30 < p[1] < right[3] + 30]
        # (the legend patches -- rectangles of the same colours inside the axes -- are dropped in _markers by their width)
        if not pts:
            continue
        rows = np.array([p[1] for p in pts])
        if marker == "^":
            rows = rows - 0.382 * np.array([p[2] for p in pts])     # incentre -> bounding-box centre (the row axis points down)
        if marker == "v":
            rows = rows + 0.382 * np.array([p[2] for p in pts])
        res["gmres"].append(np.array([value(r, 1) for r in rows]))
    px = min(res["px_per_decade"])
    res["rel_err"] = float(np.log(10.0) * (1.0 + max(res["calibration_residual_px"])) / px)
    # cross-check between the two axes: init residual of GMRES in Newton step k = Newton residual of step k
    n = min(len(res["newton"]), len(res["gmres"]))
    res["cross_check"] = np.array([res["gmres"][k][0] / res["newton"][k] - 1.0 for k in range(n)])
    return res


def report(title, r):
    print(title)
    print("  axes boxes (x0, x1, y0, y1): %s %s;  major ticks %s, %.1f / %.1f px per decade, calibration residual %.2f / %.2f px"
          % (r["boxes"][0], r["boxes"][1], r["major_ticks"], r["px_per_decade"][0], r["px_per_decade"][1], *r["calibration_residual_px"]))
    print("  one-sigma relative error of a digitised value: %.2f %%;  GMRES init / Newton residual - 1 across the two axes: %s"
          % (100 * r["rel_err"], " ".join("%+.2f%%" % (100 * c) for c in r["cross_check"])))
    print("  Newton residuals :", " ".join("%.4e" % v for v in r["newton"]))
    for k, g in enumerate(r["gmres"]):
        print("  GMRES, Newton step %d (init, inner 1 ..): %s" % (k + 1, " ".join("%.4e" % v for v in g)))


def cylinder_re40():
    return digitize("/root/reference/examples/cylinder/newton/Re40_fixed_point/residual.png", -2, -2, nseries=3)


def thermosyphon():
    """top major tick of both axes: 1e-1 (read by eye).  Only the left axes (nine Newton residuals) is reliable here: the GMRES curves of
    the nine Newton steps lie on top of each other and their markers cannot be separated by colour and erosion."""
    r = digitize("/root/reference/examples/thermosyphon/baseflow/residual.png", -1, -1, nseries=1)
    r["gmres"] = [g[:1] for g in r["gmres"]]
    r["cross_check"] = r["cross_check"][:1]
    return r


if __name__ == "__main__":
    report("cylinder Re = 40 (examples/cylinder/newton/Re40_fixed_point/residual.png)", cylinder_re40())
    if "--all" in sys.argv:
        report("thermosyphon Ra = 510 (examples/thermosyphon/baseflow/residual.png)", thermosyphon())
