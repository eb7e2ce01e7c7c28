import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


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
