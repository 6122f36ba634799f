"""Runs the ctypes example of INTEGRATION.md §2 verbatim on the GPU and checks its claims."""
import os
import re

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_integration_md_ctypes_example_runs():
    import torch  # noqa: F401  (first: one HIP runtime per process)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    md = open(os.path.join(root, "INTEGRATION.md")).read()
    blocks = re.findall(r"```python\n(.*?)```", md, flags=re.S)
    code = [b for b in blocks if "ctypes.CDLL" in b][0]
    cwd = os.getcwd()
    os.chdir(root)
    try:
        ns = {}
        exec(compile(code, "INTEGRATION.md", "exec"), ns)
    finally:
        os.chdir(cwd)
    ns["torch"].cuda.synchronize()
    want = np.random.RandomState(0).randint(0, 4096, 256)
    np.testing.assert_array_equal(ns["idx"].cpu().numpy(), want)
    np.testing.assert_array_equal(ns["out"][0].cpu().numpy(), ns["o"].cpu().numpy()[want])
    assert np.isfinite(ns["losses"].cpu().numpy()).all()
