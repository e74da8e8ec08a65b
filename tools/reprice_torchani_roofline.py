"""Round 4, VERDICT r03 weak 2: the config-2 lines recorded under profiles/ priced the reference's DENSE network flops (all 1008
AEV columns) at the step's time and called the result `roofline.frac` -- 1.28 in r03i_torchani_graph.json, which no roofline
fraction can be.  This re-prices every such record in place: `frac` / `achieved` become the EXECUTED flops (issued / 3: the split-fp16
path issues three products per fp32 product over the live columns) and the old figure moves to `vs_reference_formulation`.
Numbers measured in round 3 are not changed, only which of them is called `frac`.  Idempotent."""
import glob
import json
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles")


def reprice(roof):
    if not isinstance(roof, dict) or roof.get("bound") != "mfma" or "vs_reference_formulation" in roof:
        return False
    issued = roof.get("issued")
    if not issued or "over the live AEV columns" not in issued.get("instruction", ""):
        return False
    dense, peak = roof["achieved"], roof["peak"]
    executed = issued["tflops"] / 3.0
    roof["vs_reference_formulation"] = {"tflops": dense, "ratio_to_fp32_matrix_peak": roof["frac"],
                                        "note": "the reference's dense product over all 1008 AEV columns priced at this step's time: a "
                                                "comparison of formulations, not a roofline fraction (re-labelled in round 4)"}
    roof["achieved"] = round(executed, 3)
    roof["frac"] = round(executed / peak, 5)
    roof["note"] = ("EXECUTED network flops (live AEV columns only, forward + input gradient, fp32-equivalent) / whole step time, against the "
                    "fp32 matrix peak; `issued` = the same x 3 against the dense fp16 peak.  Re-priced in round 4 from the round-3 "
                    "measurement (VERDICT r03 weak 2); the dense-formula figure is under vs_reference_formulation")
    return True


def walk(node):
    hit = False
    if isinstance(node, dict):
        if "roofline" in node and reprice(node["roofline"]):
            hit = True
        for v in node.values():
            hit = walk(v) or hit
    elif isinstance(node, list):
        for v in node:
            hit = walk(v) or hit
    return hit


for path in sorted(glob.glob(os.path.join(ROOT, "*.json"))):
    try:
        text = open(path).read()
        doc = json.loads(text)
    except Exception:
        continue
    if walk(doc):
        open(path, "w").write(json.dumps(doc) + "\n")
        print("re-priced", os.path.basename(path))
sys.exit(0)
