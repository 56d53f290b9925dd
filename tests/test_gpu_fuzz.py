"""The two shape fuzzers as collected GPU tests on fixed seeds (they were one-off scripts the driver never ran).
tests/fuzz_pipeline_shapes.py: 40 shapes -- 22 forced ones that cover every scan-kernel family (lean, lean multi-peak, general,
long-spectrum LDS-row, the two-pass kernel for 2048 < P <= 4096 with P % 64 != 0, the generic one), overlap, forward-backward
averaging and every eigen-stage path, then random ones -- each checked bit for bit (peak ports against the oracle's
find_local_max on the pipeline's own spectrum; stand-alone blocks and root_pipeline against the chains they stand for) and to
2e-4 dB / 1e-3 degrees against the fp64 evaluation of the reference's formulas.  tests/fuzz_find_local_max.py: 200 random vectors
(lengths 3..5000, NaN / inf sprinkled in) bit for bit against the oracle (lib/find_local_max_impl.cc:80-165)."""
import pytest

import fuzz_find_local_max
import fuzz_pipeline_shapes

pytestmark = pytest.mark.gpu


def test_fuzz_pipeline_shapes_fixed_seed():
    assert fuzz_pipeline_shapes.run(40, seed=20261005, verbose=True, forced=fuzz_pipeline_shapes.FORCED) == 0


def test_fuzz_find_local_max_fixed_seed():
    assert fuzz_find_local_max.run(200, seed=77, verbose=True) == 0
