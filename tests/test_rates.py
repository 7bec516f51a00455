"""Host logic of the device Update_RCONST_x path, CPU: the rate table tools/extract_rates.py cuts out of the
generated Update_RCONST_g (gas.f:275-666), evaluated by a plain-Python restatement of the rate laws (oracle/rates_py.py),
reproduces what the COMPILED REFERENCE computes from the same inputs (tests/golden/rates_gas.npz, made by
tests/golden/make_rates_golden.py from oracle/_ref/libmistra_ref.so) — reaction by reaction, to the last bit where no
transcendental function is involved and to 4 ulp where exp / pow / log10 of two libm builds may differ."""
import json
import os
import sys

import numpy as np
import pytest

from conftest import REPO

sys.path.insert(0, os.path.join(REPO, "tools"))


def _load():
    from extract_rates import ENV
    table = json.load(open(os.path.join(REPO, "mistra_amd", "mech", "gas.rates.json")))
    slot = {n: i for i, n in enumerate(ENV["gas"])}
    g = np.load(os.path.join(REPO, "tests", "golden", "rates_gas.npz"))
    return table, slot, g["env"], g["rconst"]


def test_extracted_gas_rate_table_reproduces_the_reference():
    from oracle.rates_py import evaluate
    table, slot, env, want = _load()
    assert table["nreact"] == 331 and len(slot) == 74
    worst = 0.0
    for i in range(0, env.shape[0], 3):
        got = evaluate(table, slot, env[i])
        assert np.array_equal(got == 0.0, want[i] == 0.0)
        nz = want[i] != 0.0
        rel = np.abs(got[nz] - want[i][nz]) / np.abs(want[i][nz])
        worst = max(worst, rel.max())
    print("gas rate table vs compiled reference: max rel diff %.2e" % worst)
    assert worst <= 1e-15


def test_binary_table_matches_json():
    """mistra_amd/mech/gas.rates (what the library loads) holds the same programs as the JSON form."""
    from extract_rates import ENV, FUNC_ID, OP
    table, slot, _, _ = _load()
    raw = open(os.path.join(REPO, "mistra_amd", "mech", "gas.rates"), "rb").read()
    h = np.frombuffer(raw, np.int32, 6)
    assert h[0] == 0x5441524B and h[2] == 331 and h[3] == len(ENV["gas"])
    consts = np.frombuffer(raw, np.float64, h[4], 24)
    offs = np.frombuffer(raw, np.int32, 332, 24 + 8 * h[4])
    words = np.frombuffer(raw, np.int32, h[5], 24 + 8 * h[4] + 4 * 332)
    fid = {v[0]: k for k, v in FUNC_ID.items()}
    for r, prog in enumerate(table["programs"]):
        ws = words[offs[r]:offs[r + 1]]
        assert len(ws) == len(prog)
        for w, t in zip(ws, prog):
            op, arg = int(w) & 0xFF, int(w) >> 8
            if t[0] == "num":
                assert op == OP["const"] and consts[arg] == float(t[1])
            elif t[0] == "call":
                assert op == OP["call"] and fid[arg] == t[1]
            elif t[0] in ("var", "arr"):
                assert op == OP["env"] and 0 <= arg < 74
            else:
                assert op == OP[t[0]]
