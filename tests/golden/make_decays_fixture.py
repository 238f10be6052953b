#!/usr/bin/env python3
"""Extract the reference's recorded shear-wave decay series into a JSON fixture.

Source: /root/reference/docs/decays.js -- five ``-omv`` max-|u| series the reference's authors
recorded with ``./main -w <omega> -nx 1000 -ny 1000 -of 500 -s 200000 -omv`` (shear-wave.py:15-21,60):
fp32, u0=0.1, rho0=1, omega in {0.2,0.4,0.6,0.8,1.0}, one sample every 500 steps (400 samples each,
printed with 6 significant digits).  The file is a JS literal holding only numbers; it is parsed
as data (never executed) and re-emitted as JSON.  Also stores the viscosity table of
docs/index.qmd:288-296.

Run in the authoring container (the reference is not present on the GPU box):
    python tests/golden/make_decays_fixture.py
"""
import ast
import json
import os

SRC = "/root/reference/docs/decays.js"
DST = os.path.join(os.path.dirname(os.path.abspath(__file__)), "decays.json")


def main():
    text = open(SRC).read()
    literal = text[text.index("["):].rstrip().rstrip(";")
    series = ast.literal_eval(literal)  # numbers, lists, dicts only
    out = {
        "source": "reference docs/decays.js (recorded -omv output, 1000x1000, fp32, u0=0.1, rho0=1, "
                  "sample every 500 steps)",
        "nx": 1000, "ny": 1000, "u0": 0.1, "rho0": 1.0, "every": 500,
        "series": [{"omega": s["omega"], "ts": s["ts"], "max_us": s["max_us"]} for s in series],
        # docs/index.qmd:288-296: omega, expected nu=(1/omega-1/2)/3, nu measured by the authors
        "viscosity_table": [
            {"omega": 0.2, "expected": 1.5, "measured": 1.50486},
            {"omega": 0.4, "expected": 0.6667, "measured": 0.666610},
            {"omega": 0.6, "expected": 0.3889, "measured": 0.388907},
            {"omega": 0.8, "expected": 0.25, "measured": 0.250267},
            {"omega": 1.0, "expected": 0.1667, "measured": 0.166934},
        ],
    }
    for s in out["series"]:
        assert len(s["ts"]) == len(s["max_us"]) == 400
    with open(DST, "w") as fh:
        json.dump(out, fh, separators=(",", ":"))
    print("wrote", DST, os.path.getsize(DST), "bytes")


if __name__ == "__main__":
    main()
