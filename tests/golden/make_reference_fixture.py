#!/usr/bin/env python3
"""Generates tests/golden/reference/fixture.json: what the CPU oracle measures on the reference
test-suite's own inputs (tests/refinputs.py).  The reference itself cannot run here (C#, no dotnet),
so these are the ORACLE's values on the reference's inputs -- both test tiers are held to them.

    python tests/golden/make_reference_fixture.py
"""
import json
import sys
from pathlib import Path

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE.parent))
sys.path.insert(0, str(HERE.parent.parent))

import numpy as np  # noqa: E402

import refinputs  # noqa: E402


def main():
    out = {"generator": "oracle (tests/golden/make_reference_fixture.py)", "seed": refinputs.SEED, "scenarios": {}}
    v = refinputs.random_vectors(128, 5000)
    out["inputs"] = {
        # the first values of new Random(65537).NextSingle(), as float32 bit patterns, and digests of the sets the suite uses
        "first8_bits": [int(x) for x in v[0, :8].view(np.uint32)],
        "random_vectors_128x1000": refinputs.sha(v[:1000]), "random_vectors_128x2000": refinputs.sha(v[:2000]),
        "random_vectors_128x5000": refinputs.sha(v), "normalized_128x2000": refinputs.sha(refinputs.normalize(v[:2000])),
    }
    for name, fn in refinputs.scenarios().items():
        r = fn(refinputs.OracleAdapter)
        refinputs.check(name, r)
        out["scenarios"][name] = r
        print(name, r, flush=True)
    (HERE / "reference" / "fixture.json").write_text(json.dumps(out, indent=1) + "\n")


if __name__ == "__main__":
    main()
