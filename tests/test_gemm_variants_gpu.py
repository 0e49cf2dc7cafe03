"""GPU: every GEMM shape of tests/test_gemm_gpu.py through each large-problem kernel.

The library picks the kernel by problem size (256 x 256 tiles from half a tile per CU on, 128 x 256 tiles for groups
of few rows, the 128 x 128 kernel below that) and reads its
switches once per process, so the forced variants run in a child process:
  FI_GEMM_WS_MIN_TILES=0 FI_GEMM_BIG_MIN_TILES=0 -> every shape takes the 256 x 256 kernel (gemm_big.hip)
  FI_GEMM_WS_MIN_TILES=0 FI_GEMM_DMA_TM=256 / 128 -> every shape takes the persistent LDS-DMA kernel with
                                       256 x 128 / 128 x 256 tiles
  ... FI_GEMM_HW_SCALES=0 -> the 256 x 256 kernel without its hardware-scale path (power-of-two scales, which the
                                       reference's quantiser produces, otherwise ride the MFMA's E8M0 block scales)
(the default run of test_gemm_gpu.py covers the 128 x 128 kernel and the size-based choice)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("env", [{"FI_GEMM_WS_MIN_TILES": "0", "FI_GEMM_BIG_MIN_TILES": "0"},
                                 {"FI_GEMM_WS_MIN_TILES": "0", "FI_GEMM_DMA_TM": "256", "FI_GEMM_BIG": "0"},
                                 {"FI_GEMM_WS_MIN_TILES": "0", "FI_GEMM_DMA_TM": "128"},
                                 {"FI_GEMM_WS_MIN_TILES": "0", "FI_GEMM_BIG_MIN_TILES": "0", "FI_GEMM_HW_SCALES": "0"}],
                         ids=["dma-256x256", "dma-256x128", "dma-128x256", "dma-256x256-fold-only"])
def test_gemm_suite_through_forced_kernel(env):
    child_env = dict(os.environ)
    child_env.update(env)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gemm_gpu.py"), "-x", "-q",
                        "-p", "no:cacheprovider"], cwd=ROOT, env=child_env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert " passed" in r.stdout
