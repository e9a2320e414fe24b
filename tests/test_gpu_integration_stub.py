"""GPU tier: the reference-side binding of INTEGRATION.md section 3, executed AS WRITTEN.

The boundary row of SURVEY.md 8(b) rests on the ctypes stub a maintainer of the reference would paste into
`simplestereo/_ssamd.py` (replacing `_passive.computeASW` / `computeGSW`, reference passive.py:88-90, 153-156).  This test
extracts that code block verbatim from INTEGRATION.md, runs it in a FRESH interpreter that never imports `simplestereo_amd`
(only numpy + ctypes + the in-tree libssamd.so, named through SSAMD_LIB as the stub itself provides), and compares its
outputs with maps the unmodified reference produced (tests/golden/cases.npz: G1 Tsukuba ASW, G2 consistent ASW, G4 GSW)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def stub_source():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    sec = text[text.index("## 3. The stub a maintainer of the reference would add"):]
    start = sec.index("```python\n") + len("```python\n")
    return sec[start:sec.index("\n```", start)] + "\n"


def test_stub_block_is_the_documented_one():
    src = stub_source()
    assert src.startswith("# simplestereo/_ssamd.py") and "def computeASW(" in src and "def computeGSW(" in src
    assert "simplestereo_amd" not in src                      # the stub binds the C ABI, not this package
    compile(src, "INTEGRATION.md#3", "exec")


DRIVER = r"""
import json, os, sys, types
import numpy as np
src = open(sys.argv[1]).read()
mod = types.ModuleType("_ssamd")
exec(compile(src, "INTEGRATION.md#3", "exec"), mod.__dict__)
assert "simplestereo_amd" not in sys.modules
G = sys.argv[2]
maps = np.load(os.path.join(G, "cases.npz"))
meta = json.load(open(os.path.join(G, "cases.json")))
z = np.load(os.path.join(G, "tsukuba_pair.npz"))
L, R = np.ascontiguousarray(z["left"]), np.ascontiguousarray(z["right"])
out = {}
for cid in ("G1", "G2", "G4"):
    m = meta[cid]; p = m["params"]
    assert m["input"] == "tsukuba"
    if p["algo"] == "asw":
        d = mod.computeASW(L, R, p["winSize"], p["maxDisparity"], p["minDisparity"], p["gammaC"], p["gammaP"], p["consistent"])
    else:
        d = mod.computeGSW(L, R, p["winSize"], p["maxDisparity"], p["minDisparity"], p["gamma"], p["fMax"], p["iterations"], p["bins"])
    out[cid] = {"dtype": str(d.dtype), "shape": list(d.shape), "differing": int(np.count_nonzero(d != maps[cid])), "checksum": int(d.astype(np.int64).sum())}
# the error path of the stub: a negative minDisparity is SSAMD_EINVAL -> ValueError with the library's message
try:
    mod.computeASW(L, R, 15, 16, -1, 5, 17.5)
    out["error"] = None
except ValueError as e:
    out["error"] = str(e)
print("RESULT " + json.dumps(out))
"""


def test_the_stub_as_written_reproduces_the_references_maps(tmp_path):
    stub = tmp_path / "_ssamd.py"
    stub.write_text(stub_source())
    drv = tmp_path / "driver.py"
    drv.write_text(DRIVER)
    env = dict(os.environ, SSAMD_LIB=os.path.join(ROOT, "simplestereo_amd", "libssamd.so"))
    env.pop("PYTHONPATH", None)
    r = subprocess.run([sys.executable, str(drv), str(stub), os.path.join(ROOT, "tests", "golden")], capture_output=True, text=True,
                       timeout=600, env=env, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr[-3000:]
    res = json.loads([l for l in r.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])
    meta = json.load(open(os.path.join(ROOT, "tests", "golden", "cases.json")))
    for cid in ("G1", "G2", "G4"):
        assert res[cid]["dtype"] == "int16" and res[cid]["shape"] == meta[cid]["shape"]
        assert res[cid]["differing"] == 0 and res[cid]["checksum"] == meta[cid]["checksum"], (cid, res[cid])
    assert res["error"] and "isparity" in res["error"]
