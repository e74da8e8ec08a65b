#!/usr/bin/env python3
"""Re-encode the golden vectors held by the reference's own C++ unit tests as .npz fixtures.

Run in the authoring container only (needs /root/reference):

    python tests/golden/make_golden_cpp.py

Sources (data only -- numeric initialisers, no code is carried over):
  /root/reference/src/ani/TestANISymmetryFunctions.h     18-atom water cluster, TorchANI-generated
      expected radial[72] / angular[216] for {non-periodic, cubic 9 A, triclinic}   (:63-252)
  /root/reference/src/schnet/TestCFConv.h                same cluster, SchNetPack-generated
      expected output[144] for {non-periodic ssp, cubic 5 A, triclinic, tanh}       (:81-247)

Outputs: tests/golden/ani_water18.npz, tests/golden/cfconv_water18.npz
"""
import os
import re

import numpy as np

REF = "/root/reference/src"
HERE = os.path.dirname(os.path.abspath(__file__))

_num = r"[-+]?(?:\d+\.\d*|\.\d+|\d+)(?:[eE][-+]?\d+)?"


def _function_body(text, name):
    start = text.index("void " + name + "(")
    brace = text.index("{", start)
    depth, i = 0, brace
    while True:
        if text[i] == "{":
            depth += 1
        elif text[i] == "}":
            depth -= 1
            if depth == 0:
                return text[brace:i + 1]
        i += 1


def _array(body, name):
    m = re.search(r"\b" + re.escape(name) + r"\s*(?:\[[^\]]*\])*\s*=\s*\{", body)
    if not m:
        raise KeyError(name)
    depth, i = 1, m.end()
    while depth:
        if body[i] == "{":
            depth += 1
        elif body[i] == "}":
            depth -= 1
        i += 1
    return np.array([float(x) for x in re.findall(_num, body[m.end():i - 1])], dtype=np.float64)


def ani():
    text = open(f"{REF}/ani/TestANISymmetryFunctions.h").read()
    common = _function_body(text, "testWater")
    out = {
        "positions": _array(common, "positions").reshape(18, 3).astype(np.float32),
        "species": _array(common, "species").astype(np.int32),
        "radial_functions": _array(common, "radialFunctions").reshape(-1, 2).astype(np.float32),
        "angular_functions": _array(common, "angularFunctions").reshape(-1, 4).astype(np.float32),
        "n_species": np.int32(2),
        "rc_radial": np.float32(4.5),
        "rc_angular": np.float32(3.5),
    }
    for tag, fn in (("nonperiodic", "testWaterNonperiodic"), ("periodic", "testWaterPeriodic"),
                    ("triclinic", "testWaterTriclinic")):
        body = _function_body(text, fn)
        out[f"{tag}_radial"] = _array(body, "expectedRadial").astype(np.float32).reshape(18, -1)
        out[f"{tag}_angular"] = _array(body, "expectedAngular").astype(np.float32).reshape(18, -1)
        if tag != "nonperiodic":
            out[f"{tag}_box"] = _array(body, "periodicVectors").astype(np.float32).reshape(3, 3)
    assert out["nonperiodic_radial"].shape == (18, 4) and out["nonperiodic_angular"].shape == (18, 12)
    np.savez(os.path.join(HERE, "ani_water18.npz"), **out)
    print("ani_water18.npz:", {k: np.shape(v) for k, v in out.items()})


def cfconv():
    text = open(f"{REF}/schnet/TestCFConv.h").read()
    common = _function_body(text, "testWater")
    out = {
        "positions": _array(common, "positions").reshape(18, 3).astype(np.float32),
        "w1": _array(common, "w1").reshape(8, 5).astype(np.float32),   # core layout [W][G]
        "w2": _array(common, "w2").reshape(8, 8).astype(np.float32),   # [out][in]
        "b1": np.arange(1, 9, dtype=np.float32),                       # TestCFConv.h:123
        "b2": (0.1 * np.arange(1, 9)).astype(np.float32),              # TestCFConv.h:124
        "x": (0.1 * np.arange(8 * 18)).astype(np.float32).reshape(18, 8),   # TestCFConv.h:126-127
        "width": np.int32(8), "n_gauss": np.int32(5),
        "cutoff": np.float32(2.0), "sigma": np.float32(0.5),
    }
    b1 = _array(common, "b1")
    b2 = _array(common, "b2")
    assert np.allclose(b1, out["b1"]) and np.allclose(b2, out["b2"])
    for tag, fn in (("nonperiodic_ssp", "testWaterNonperiodic"), ("periodic_ssp", "testWaterPeriodic"),
                    ("triclinic_ssp", "testWaterTriclinic"), ("nonperiodic_tanh", "testWaterTanh")):
        body = _function_body(text, fn)
        out[f"{tag}_output"] = _array(body, "expectedOutput").astype(np.float32).reshape(18, 8)
        if "periodicVectors" in body.split("testWater(")[0]:
            out[f"{tag}_box"] = _array(body, "periodicVectors").astype(np.float32).reshape(3, 3)
    np.savez(os.path.join(HERE, "cfconv_water18.npz"), **out)
    print("cfconv_water18.npz:", {k: np.shape(v) for k, v in out.items()})


if __name__ == "__main__":
    ani()
    cfconv()
