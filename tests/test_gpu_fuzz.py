"""Randomised parity sweep as a test: tools/fuzz_parity.py draws network shapes (both lineages, with
and without view directions, depths 2..8, any skip position, encoding lengths 1..10 / 1..4), sample
counts (2..96 coarse, 0..160 fine), ray counts (1, 2, 31, 33, ...), NDC or not, the render flags and
the saved-tensor format,
and compares the maps and both gradients of render_rays with the bf16-emulating oracle (maps 2e-2
absolute, whole-gradient relative L2 5e-2).  25 cases here; profiles hold two sweeps of 100."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_random_shapes_and_flags_match_the_oracle():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU and the HIP library; there is no CPU fallback")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_parity.py"), "25", "3"],
                       env=dict(os.environ, PYTHONPATH=ROOT), capture_output=True, text=True, timeout=280)
    lines = [l for l in r.stdout.splitlines() if l.startswith(("ok", "FAIL", "ERR"))]
    assert r.returncode == 0 and len(lines) == 25, "\n".join(l for l in r.stdout.splitlines() if not l.startswith("ok"))[-3000:] + r.stderr[-1500:]
