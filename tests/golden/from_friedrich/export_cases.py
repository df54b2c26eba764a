#!/usr/bin/env python3
"""Export the INPUTS of the golden cases (tests/golden/golden_v1.npz + manifest) and the SplitMix64 test vector as
tests/golden/from_friedrich/cases_v1.json, the file the Rust program reads.  Outputs are NOT exported: they are what the real
friedrich is asked for.      python tests/golden/from_friedrich/export_cases.py"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, ROOT)
from friedrich_amd import synth  # noqa: E402

G = np.load(os.path.join(HERE, "..", "golden_v1.npz"))
M = json.load(open(os.path.join(HERE, "..", "golden_v1.json")))
V2 = json.load(open(os.path.join(HERE, "..", "golden_v2.json")))["config0_default"]["config"]


def arr(name, key):
    k = f"{name}/{key}"
    return G[k].tolist() if k in G else []


def main():
    cases = {}
    for name, meta in M.items():
        if name == "readme_default":
            continue
        X = G[f"{name}/X"]
        n_add = G[f"{name}/Xadd"].shape[0] if f"{name}/Xadd" in G else 0
        yadd = []
        if n_add:
            # make_golden.py: the README set's additional outputs, sin(sum) elsewhere
            Xadd = G[f"{name}/Xadd"]
            yadd = [2.0, 3.0, -1.0, -2.0] if name == "readme_1d" else np.sin(Xadd.sum(axis=1)).tolist()
        cases[name] = {"kernel": meta["kernel"], "noise": meta["noise"], "eps": meta["eps"], "prior": meta["prior"],
                       "X": X.tolist(), "y": G[f"{name}/y"].tolist(), "Xq": arr(name, "Xq"), "Xadd": arr(name, "Xadd"), "yadd": yadd}
    seed = 0x5EED0000
    doc = {"what": "inputs of tests/golden/golden_v1 (rows of X / Xq / Xadd, raw outputs y, ConstantPrior value, kernel spec in the "
                   "grammar of oracle/oracle.py) for tests/golden/from_friedrich/src/main.rs",
           "splitmix_check": {"seed": seed, "values": synth.splitmix64_uniform(seed, 0, 8).tolist()},
           "config0_default": {"n": V2["n"], "d": V2["d"], "m": V2["m"], "cfg": V2["cfg"]},
           "cases": cases}
    with open(os.path.join(HERE, "cases_v1.json"), "w") as f:
        json.dump(doc, f)
    print("wrote cases_v1.json:", {k: (len(v["X"]), len(v["Xq"]), len(v["Xadd"])) for k, v in cases.items()})


if __name__ == "__main__":
    main()
