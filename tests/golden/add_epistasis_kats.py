#!/usr/bin/env python3
"""Adds the "epistasis_model" section to reference_kats.json: the inputs and expected outputs of the
reference's own unit tests of the MDR counting path, transcribed by hand (data only).  The reference pads
the affected and the unaffected group to multiples of 16 samples for its SSE loads; the arrays below keep
that padded shape (`padded`: [affected block | unaffected block]) exactly as the tests write them, and
tests/helpers strip it with the group sizes.

  test/test_epistasis_model.c:34-100    test_get_masks                     (order 4: the genotype masks of four SNPs)
  test/test_epistasis_model.c:116-194   test_get_counts                    (order 2 and 3)
  test/test_epistasis_model.c:196-289   test_get_counts_all_folds_order_2
  test/test_epistasis_model.c:291-366   test_get_counts_all_folds_order_3
  test/test_epistasis_model.c:369-431   test_get_confusion_matrix
  test/test_epistasis_model.c:434-519   test_get_confusion_matrix_excluding_samples
  test/test_epistasis_model.c:522-543   test_model_evaluation_formulas
  test/test_mdr.c:33-65                 test_get_high_risk_combinations{,2}
  test/test_cross_validation.c:36-283   test_get_k_folds (fold sizes; mask layout asserted structurally)
"""
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))
Z = [0] * 12

kat = {"_source": "transcribed from test/test_epistasis_model.c and test/test_mdr.c of opencb/hpg-variant (see add_epistasis_kats.py)"}

# ---- test_get_counts: 4 affected + 4 unaffected ------------------------------------------------
gt0 = [0, 0, 1, 0] + Z + [2, 1, 0, 2] + Z
gt1 = [0, 1, 1, 0] + Z + [0, 0, 1, 1] + Z
gt2 = [1, 2, 0, 1] + Z + [0, 2, 0, 0] + Z
o3_aff, o3_unaff = [0] * 27, [0] * 27
o3_aff[1], o3_aff[5], o3_aff[12] = 2, 1, 1
o3_unaff[3], o3_unaff[11], o3_unaff[18], o3_unaff[21] = 1, 1, 1, 1
kat["counts"] = {
    "num_affected": 4, "num_unaffected": 4, "padded_rows": [gt0, gt1, gt2],
    "order2": {"rows": [0, 1], "aff": [2, 1, 0, 0, 1, 0, 0, 0, 0], "unaff": [0, 1, 0, 1, 0, 0, 1, 1, 0]},
    "order3": {"rows": [0, 1, 2], "aff": o3_aff, "unaff": o3_unaff},
}

# ---- test_get_masks: the fourth SNP of the file's `genotypes` array and the expected 0 / 255 masks of all four SNPs, per
# SNP [genotype 0 | genotype 1 | genotype 2], 32 padded samples each (4 affected + 12 pad + 4 unaffected + 12 pad).  The
# order-4 cell counts follow from THESE masks by combination_counts' own arithmetic (AND the four masks of a cell, count
# the set bytes: model.c:76-124): tests derive them from the vectors below, nothing else is assumed.
gt3 = [0, 0, 0, 2] + Z + [1, 1, 0, 2] + Z
def _m(aff, unaff):
    return [255 * x for x in aff] + Z + [255 * x for x in unaff] + Z
kat["masks_order4"] = {
    "num_affected": 4, "num_unaffected": 4, "padded_rows": [gt0, gt1, gt2, gt3],
    "masks": [
        _m([1, 1, 0, 1], [0, 0, 1, 0]) + _m([0, 0, 1, 0], [0, 1, 0, 0]) + _m([0, 0, 0, 0], [1, 0, 0, 1]),
        _m([1, 0, 0, 1], [1, 1, 0, 0]) + _m([0, 1, 1, 0], [0, 0, 1, 1]) + _m([0, 0, 0, 0], [0, 0, 0, 0]),
        _m([0, 0, 1, 0], [1, 0, 1, 1]) + _m([1, 0, 0, 1], [0, 0, 0, 0]) + _m([0, 1, 0, 0], [0, 1, 0, 0]),
        _m([1, 1, 1, 0], [0, 0, 1, 0]) + _m([0, 0, 0, 0], [1, 1, 0, 0]) + _m([0, 0, 0, 1], [0, 0, 0, 1]),
    ],
}

# ---- test_get_counts_all_folds_order_{2,3}: 5 affected + 10 unaffected, 5 folds ------------------
f0 = [0, 0, 1, 0, 2] + [0] * 11 + [2, 1, 0, 2, 1, 0, 2, 1, 0, 2] + [0] * 6
f1 = [0, 1, 1, 0, 0] + [0] * 11 + [0, 0, 1, 1, 0, 0, 0, 0, 2, 2] + [0] * 6
f2 = [1, 2, 0, 1, 1] + [0] * 11 + [0, 2, 1, 0, 1, 1, 1, 2, 2, 0] + [0] * 6
masks = [
    [1, 0, 1, 1, 1] + [0] * 11 + [0, 0, 1, 1, 1, 1, 1, 1, 1, 1] + [0] * 6,
    [1, 1, 0, 1, 1] + [0] * 11 + [1, 1, 1, 1, 0, 0, 1, 1, 1, 1] + [0] * 6,
    [1, 1, 1, 0, 1] + [0] * 11 + [1, 1, 1, 1, 1, 1, 1, 1, 0, 0] + [0] * 6,
    [1, 1, 1, 1, 0] + [0] * 11 + [1, 1, 0, 0, 1, 1, 1, 1, 1, 1] + [0] * 6,
    [0, 1, 1, 1, 1] + [0] * 11 + [1, 1, 1, 1, 1, 1, 0, 0, 1, 1] + [0] * 6,
]
kat["counts_all_folds"] = {
    "num_affected": 5, "num_unaffected": 10, "num_folds": 5, "padded_rows": [f0, f1, f2], "padded_fold_masks": masks,
    "order2": {"rows": [0, 1],
               "aff": [[2, 0, 0, 0, 1, 0, 1, 0, 0], [2, 1, 0, 0, 0, 0, 1, 0, 0], [1, 1, 0, 0, 1, 0, 1, 0, 0],
                       [2, 1, 0, 0, 1, 0, 0, 0, 0], [1, 1, 0, 0, 1, 0, 1, 0, 0]],
               "unaff": [[1, 1, 1, 2, 0, 0, 1, 1, 1], [0, 1, 1, 2, 0, 0, 2, 1, 1], [1, 1, 0, 3, 0, 0, 2, 1, 0],
                         [1, 0, 1, 3, 0, 0, 2, 0, 1], [1, 1, 1, 2, 0, 0, 1, 1, 1]]},
    # the order-3 test checks some cells only: {fold: {cell: [aff, unaff]}}
    "order3": {"rows": [0, 1, 2], "some_cells": {
        "0": {"0": [0, 0], "1": [2, 1], "2": [0, 0], "4": [0, 1], "5": [0, 0], "8": [0, 1], "9": [0, 0], "11": [0, 1],
              "12": [1, 0], "15": [0, 0], "18": [0, 0], "19": [1, 1], "21": [0, 1], "24": [0, 1]},
        "1": {"0": [0, 0], "1": [2, 0], "2": [0, 0], "4": [0, 1], "5": [1, 0], "8": [0, 1], "9": [0, 0], "11": [0, 2],
              "12": [0, 0], "15": [0, 0], "18": [0, 1], "19": [1, 1], "21": [0, 1], "24": [0, 1]}}},
}

# ---- confusion matrices: expected {TP, FN, FP, TN} -------------------------------------------------
a0 = [1, 1, 0, 2, 2, 2, 1] + [0] * 9 + [0, 0, 0, 1, 2] + [0] * 11
a1 = [0, 0, 1, 1, 2, 2, 0] + [0] * 9 + [0, 1, 1, 2, 2] + [0] * 11
b0 = [1, 1, 0, 2] + Z + [2, 2, 1, 0, 0, 0, 1, 2] + [0] * 8
b1 = [0, 0, 1, 1] + Z + [2, 2, 0, 0, 1, 1, 2, 2] + [0] * 8
c0 = [1, 1, 0, 2, 2, 2] + [0] * 10 + [1, 0, 0, 0, 1, 2] + [0] * 10
c1 = [0, 0, 1, 1, 2, 2] + [0] * 10 + [0, 0, 1, 1, 2, 2] + [0] * 10
c2 = [1, 1, 1, 0, 1, 1] + [0] * 10 + [1, 0, 1, 1, 0, 0] + [0] * 10
risky2 = [[1, 0], [2, 1], [2, 2]]                       # cells 3, 7, 8 of the order-2 table
risky3 = [[0, 1, 1], [1, 0, 1], [2, 1, 0], [2, 2, 1]]   # cells 4, 10, 21, 25 of the order-3 table
kat["confusion"] = [
    {"line": 387, "num_affected": 7, "num_unaffected": 5, "padded_rows": [a0, a1], "risky": risky2, "subset": "TRAINING",
     "padded_fold_mask": [1] * 7 + [0] * 9 + [1] * 5 + [0] * 11, "training_size": [7, 5], "testing_size": [0, 0], "matrix": [6, 1, 1, 4]},
    {"line": 402, "num_affected": 4, "num_unaffected": 8, "padded_rows": [b0, b1], "risky": risky2, "subset": "TRAINING",
     "padded_fold_mask": [1] * 4 + Z + [1] * 8 + [0] * 8, "training_size": [4, 8], "testing_size": [0, 0], "matrix": [3, 1, 4, 4]},
    {"line": 426, "num_affected": 6, "num_unaffected": 6, "padded_rows": [c0, c1, c2], "risky": risky3, "subset": "TRAINING",
     "padded_fold_mask": [1] * 6 + [0] * 10 + [1] * 6 + [0] * 10, "training_size": [6, 6], "testing_size": [0, 0], "matrix": [6, 0, 3, 3]},
]
ma = [1, 1, 1, 1, 0, 0, 0] + [0] * 9 + [1, 1, 1, 0, 0] + [0] * 11
mb = [1, 0, 1, 0, 1, 0, 1] + [0] * 9 + [0, 1, 0, 1, 0] + [0] * 11
mc = [1, 1, 0, 1, 1, 1, 1] + [0] * 9 + [1, 1, 1, 0, 0] + [0] * 11
for line, mask, tr, te, sub, mat in [
        (455, ma, [4, 3], [3, 2], "TRAINING", [3, 1, 0, 3]), (463, ma, [4, 3], [3, 2], "TESTING", [3, 0, 1, 1]),
        (473, mb, [4, 2], [3, 3], "TRAINING", [3, 1, 0, 2]), (481, mb, [4, 2], [3, 3], "TESTING", [3, 0, 1, 2]),
        (491, mc, [6, 4], [1, 1], "TRAINING", [6, 0, 0, 4]), (499, mc, [6, 4], [1, 1], "TESTING", [0, 1, 1, 0])]:
    kat["confusion"].append({"line": line, "num_affected": 7, "num_unaffected": 5, "padded_rows": [a0, a1], "risky": risky2,
                             "subset": sub, "padded_fold_mask": mask, "training_size": tr, "testing_size": te, "matrix": mat})

# ---- evaluation formulas (the test checks value - expected <= 1e-6) --------------------------------
kat["evaluate"] = [
    {"matrix": [40, 2, 4, 10], "CA": 0.89285714, "BA": 0.83333333, "GAMMA": 0.96078431, "TAU_B": 0.70352647},
    {"matrix": [20, 10, 10, 20], "CA": 0.66666666, "BA": 0.66666666, "GAMMA": 0.6, "TAU_B": 0.33333333},
]

# ---- MDR high-risk cells -----------------------------------------------------------------------------
kat["mdr_high_risk"] = {
    "num_affected": 10, "num_unaffected": 80,
    # test_get_high_risk_combinations: counts come in (affected, unaffected) pairs
    "scalar": {"counts": [[8, 40], [4, 75], [9, 20], [8, 63]], "risky_indices": [0, 2, 3]},
    # test_get_high_risk_combinations2 (what the runner uses)
    "vector": {"aff": [8, 4, 9, 8, 4], "unaff": [40, 75, 20, 63, 40], "risky": [True, False, True, True, False]},
}

# ---- get_k_folds: fold sizes (samples, affected, unaffected), test/test_cross_validation.c:36-283 ------
def _sizes(*runs):
    out = []
    for n, triple in runs:
        out += [list(triple)] * n
    return out


kat["k_folds"] = [
    {"num_affected": 200, "num_unaffected": 200, "k": 10, "sizes": _sizes((10, (40, 20, 20)))},
    {"num_affected": 150, "num_unaffected": 250, "k": 4, "sizes": _sizes((2, (101, 38, 63)), (2, (99, 37, 62)))},
    {"num_affected": 150, "num_unaffected": 250, "k": 10, "sizes": _sizes((10, (40, 15, 25)))},
    {"num_affected": 50, "num_unaffected": 75, "k": 5, "sizes": _sizes((5, (25, 10, 15)))},
    {"num_affected": 50, "num_unaffected": 75, "k": 7, "sizes": _sizes((1, (19, 8, 11)), (4, (18, 7, 11)), (2, (17, 7, 10)))},
    {"num_affected": 50, "num_unaffected": 75, "k": 10, "sizes": _sizes((5, (13, 5, 8)), (5, (12, 5, 7)))},
    {"num_affected": 8, "num_unaffected": 12, "k": 8, "sizes": _sizes((4, (3, 1, 2)), (4, (2, 1, 1)))},
    # the test leaves fold 8 of this case unchecked (its loop starts at 9); None = not asserted by the reference
    {"num_affected": 8, "num_unaffected": 12, "k": 10, "sizes": _sizes((2, (3, 1, 2)), (6, (2, 1, 1))) + [None, [1, 0, 1]]},
    {"num_affected": 16, "num_unaffected": 4, "k": 5, "sizes": [[5, 4, 1], [4, 3, 1], [4, 3, 1], [4, 3, 1], [3, 3, 0]]},
    {"num_affected": 16, "num_unaffected": 4, "k": 10, "sizes": _sizes((4, (3, 2, 1)), (2, (2, 2, 0)), (4, (1, 1, 0)))},
]

path = os.path.join(HERE, "reference_kats.json")
d = json.load(open(path))
d["epistasis_model"] = kat
json.dump(d, open(path, "w"), indent=1)
print("epistasis_model section written")
