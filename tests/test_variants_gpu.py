"""GPU: the opt-in kernel variants (environment switches read once per process) keep passing the parity tests of the default
kernels: each runs the relevant tests of test_kernels_gpu.py in a child process with the switch set."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

VARIANTS = [
    ('RDFC_GLDS', '4', '(test_conv_fwd_dgrad_wgrad or conv_transpose) and bf16'),   # 4 consumer + 4 producer waves, pipelined fragment reads
    ('RDFC_GLDS', '5', '(test_conv_fwd_dgrad_wgrad or conv_transpose) and bf16'),   # 4 + 2
    ('RDFC_GLDS', '6', '(test_conv_fwd_dgrad_wgrad or conv_transpose) and bf16'),   # 64 co x 128 px wave tiles
    ('RDFC_GLDS', '3', '(test_conv_fwd_dgrad_wgrad or conv_transpose) and bf16'),   # 8-wave 128 co x 256 px workgroups
    ('RDFC_GLDS', '0', '(test_conv_fwd_dgrad_wgrad or conv_transpose) and bf16'),   # register-staged predecessors
    ('RDFC_WGRAD_GLDS_NINE', '1', 'test_conv_fwd_dgrad_wgrad and bf16'),            # all nine taps per workgroup
    ('RDFC_FUSED_STATS', '1', 'fused_statistics or stacked_batches'),               # conv-epilogue statistics (also under stacked-batch BatchNorm)
    ('RDFC_WGRAD_GLDS', '5', 'test_conv_fwd_dgrad_wgrad and bf16'),                 # 128 x 128 eight-wave weight-gradient tile
    ('RDFC_NORM_ONESTAGE', '1', 'test_norm_fwd_bwd or onestage'),                   # f64-atomic reductions
    ('RDFC_WGRAD_FLAT', '1', '(test_conv_fwd_dgrad_wgrad and not f32) or reflection_pad_folded'),   # flat pixel walk of the weight gradient on every shape
    ('RDFC_WGRAD_FLAT', '0', 'test_conv_fwd_dgrad_wgrad and bf16'),                  # ... and on none
]


@pytest.mark.parametrize('name,value,expr', VARIANTS, ids=[f'{n}={v}' for n, v, _ in VARIANTS])
def test_variant_passes_the_default_kernels_tests(name, value, expr):
    env = dict(os.environ, **{name: value})
    r = subprocess.run([sys.executable, '-m', 'pytest', os.path.join(ROOT, 'tests', 'test_kernels_gpu.py'), '-q', '-x', '-k', expr],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    tail = (r.stdout + r.stderr)[-2500:]
    assert r.returncode == 0, tail
    assert ' passed' in tail and ' skipped' not in tail.splitlines()[-1], tail
