"""An input whose per-column lists do not fit the device fails LOUDLY (FSEQ_E_OOM with the sizes in the message), and the failure
stays with that call: found with C4's shape at mu = 1e-3 (profiles/r05_diversity_C4.txt: the lists would take 758 GB), where
the HIP runtime kept the failed hipMalloc as its last error and the next context of the process -- a different input -- failed
in the check behind its first kernel launch with "hipGetLastError(): out of memory"."""
import importlib

import numpy as np
import pytest

import fso
from test_gpu_parity import compare_long

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pkg():
    return importlib.import_module("founder-sequences_amd")


def test_lists_beyond_the_device_fail_with_oom_and_leave_no_error_behind(pkg):
    m, n, L = 20_000, 2_500_000, 100                                 # (a list holds at most m entries: the rows make the size)
    ctx = pkg.SegmentationContext(m, n, L, list_cap=m)               # 2,500,000 x 20,002 x 8 bytes = 400 GB of lists
    ctx.generate_synthetic(7, 8, 1000, 1e-4, 0)
    with pytest.raises(pkg.FseqError) as ei:
        ctx.run()
    assert ei.value.code == pkg.FSEQ_E_OOM
    assert "hipMalloc of" in str(ei.value) and "bytes free on the device" in str(ei.value)
    ctx.close()
    # the next context of the process (same thread) sees nothing of it
    msa = fso.synth_msa(fso.synth_spec(5, 6, 400, 1e-3, 0), 300, 2500)
    compare_long(pkg, msa, 20)
    # and neither does a context that was open while the other one failed
    a = pkg.SegmentationContext(300, 2500, 20)
    a.set_sequences(msa)
    b = pkg.SegmentationContext(m, n, L, list_cap=m)
    b.generate_synthetic(7, 8, 1000, 1e-4, 0)
    with pytest.raises(pkg.FseqError):
        b.run()
    b.close()
    a.run()
    ref = fso.segment_long(msa, 20, keep_dp=False, threads=2)
    assert a.result.max_segment_size == ref["max_segment_size"]
    assert np.array_equal(a.traceback()["lb"], ref["traceback"]["lb"])
    a.close()
