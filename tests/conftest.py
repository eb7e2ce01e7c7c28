import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    # The CPU oracle (torch CPU ops) is what most of the -m gpu suite's wall time goes to.  torch sizes its thread pool by the
    # host's core count (256 on a GPU box), the box grants a one-GPU job about 16 cores: left alone, the pool is oversubscribed
    # 16-fold and the suite's time swings between 5 and 13 minutes with the host's load.  Same cap as bench.py's cpu_baseline.
    import torch
    torch.set_num_threads(min(16, os.cpu_count() or 1))


@pytest.fixture(scope='session')
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip('no GPU visible')
    return torch.device('cuda:0')


@pytest.fixture(params=['bf16x3', 'f32mfma'])
def conv_arith(request):
    """Runs a test once per conv arithmetic: the default (fp32 products from three bf16 pieces per operand, six bf16 MFMA
    products) and the v_mfma_f32_32x32x2_f32 kernels."""
    from bdvcil_amd import kernels as K
    prev = K.set_conv_arith(request.param)
    yield request.param
    K.set_conv_arith('bf16x3')
    K.FPROP_X3, K.DGRAD_X3, K.WGRAD_X3 = prev
