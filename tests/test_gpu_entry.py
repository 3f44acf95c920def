"""__graft_entry__ as the driver uses it: build() and smoke() in ONE fresh process on the GPU box (build() loads libxmh.so before anything
has imported torch -- the order that once left the process with two HIP runtimes and smoke() without a device)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_build_then_smoke_in_one_process():
    out = subprocess.run([sys.executable, "-c", "import __graft_entry__ as g; g.build(); g.smoke(); print('entry ok')"], cwd=ROOT, capture_output=True,
                         text=True, timeout=1200)
    assert out.returncode == 0 and "entry ok" in out.stdout, (out.stdout[-400:], out.stderr[-800:])
