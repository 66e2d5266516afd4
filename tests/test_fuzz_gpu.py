"""A bounded slice of the randomised parity sweep (tests/fuzz_parity.py) in the GPU suite: random
shapes through one E+M step against the fp64 restatement (1e-6 on v) and the scorer bit-exact."""
import pytest

from tests import fuzz_parity

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,seed,budget", [(150, 11, 60000), (30, 12, 1500000)], ids=["one_block", "many_blocks"])
def test_random_shapes_match_exact_arithmetic(n, seed, budget, gpu_ctx, orc, capsys):
    bad, kernels = fuzz_parity.run(n, seed, budget, ctx=gpu_ctx, orc=orc, verbose=False)
    assert bad == 0, capsys.readouterr().out[-4000:]
    assert len(kernels) >= 2                                  # grouped, per-column / sliced and mixed launches all occur
