"""The iteration kernels and the wave-per-query k-NN must not spill to scratch memory.

A spilled vector register costs a wait for every outstanding load wherever it is reloaded, and
a scratch allocation lengthens every launch; small source changes have tipped the register
allocation over more than once (the launch got 1.5x slower without any test failing).  This
compiles the device code the way the Makefile does and reads the compiler's resource remarks.
"""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "stereovision_amd", "csrc")


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not on PATH")
def test_iteration_kernels_use_no_scratch(tmp_path):
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off",
           "--cuda-device-only", "-c", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC,
           "-Rpass-analysis=kernel-resource-usage", "-o", str(tmp_path / "svicp.o"),
           os.path.join(CSRC, "svicp_api.hip")]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    name, seen = None, {}
    for line in out.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = m.group(1)
        m = re.search(r"ScratchSize \[bytes/lane\]: (\d+)", line)
        if m and name:
            seen[name] = int(m.group(1))
    # every variant of the single-pair kernel (five, each with the solve inside the launch and
    # with the solve in a launch of its own) and the batch kernel in its usual form
    # (searching every point every iteration in a batch is a debugging option)
    single = {k: v for k, v in seen.items() if "18icp_iterate_kernel" in k}
    batch = {k: v for k, v in seen.items() if "24icp_iterate_batch_kernelILb1ELb1" in k
             or "24icp_iterate_batch_kernelILb0ELb0" in k}
    knn = {k: v for k, v in seen.items() if "15knn_wave_kernel" in k}
    assert len(single) == 10 and len(batch) == 2 and len(knn) == 1, sorted(seen)
    spilling = {k: v for k, v in {**single, **batch, **knn}.items() if v != 0}
    assert not spilling, spilling
