"""GPU: the opt-in kernel variants (environment switches read once per process) keep passing the parity tests of the default
kernels: each runs the relevant tests of test_kernels_gpu.py in a child process with the switch set."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# what a forced tile height of the 3x3 kernel has to pass: every conv / statistics / reflection / addend test of test_kernels_gpu.py and the
# adjointness + value tests at the benchmark's launch sizes
JRK = '(conv or stat or reflect or addend or adjoint) and not f32 and not under_load and not scaling and not values_against and not tile_height'
VARIANTS = [
    ('RDFC_GLDS', '0', '(test_conv_fwd_dgrad_wgrad or conv_transpose) and bf16'),   # register-staged predecessors of the LDS-DMA 3x3 kernel
    ('RDFC_WGRAD_RING', '0', '(test_conv_fwd_dgrad_wgrad and not f32) or reflection_pad_folded'),   # segment walk + float atomics instead of the ring kernel
    ('RDFC_WGRAD_RING', '3', '((test_conv_fwd_dgrad_wgrad or conv_transpose) and not f32) or reflection_pad_folded'),   # the ring kernel also on launches the default leaves to the older kernels (small maps)
    ('RDFC_FUSED_STATS', '1', 'fused_statistics or stacked_batches'),               # conv-epilogue statistics (also under stacked-batch BatchNorm)
    ('RDFC_T3S2', '0', '(test_conv_fwd_dgrad_wgrad or conv_transpose) and not f32'),   # transposed 3x3 / stride 2 on igemm_glds' parity classes instead of the patch kernel
    ('RDFC_T3S2', '3', '(test_conv_fwd_dgrad_wgrad or conv_transpose or dgrad_with_addend) and not f32'),   # ... and the patch kernel also on the small launches the default leaves to igemm_glds (its one- and two-addend epilogues included)
    ('RDFC_GLDS_STACK', '0', '(test_conv_fwd_dgrad_wgrad or conv_transpose) and not f32'),   # per-image row tiles in the 3x3 kernel
    # round 5: every tile height of the 128-channel 3x3 / stride-1 kernel (the default picks one per launch: conv_glds.hip glds_plan) on every
    # conv test incl. fused statistics, reflection fold, addends, stacked batches and the full-size adjointness (that all heights give the same
    # bits at the benchmark's sizes is test_fullsize_gpu.py::test_every_tile_height_gives_the_same_bits, in one process)
    ('RDFC_GLDS_JR', '4', JRK, 'test_kernels_gpu.py', 'test_fullsize_gpu.py'),
    ('RDFC_GLDS_JR', '6', JRK, 'test_kernels_gpu.py', 'test_fullsize_gpu.py'),
    ('RDFC_GLDS_JR', '8', JRK, 'test_kernels_gpu.py', 'test_fullsize_gpu.py'),
    ('RDFC_GLDS_JR', '10', JRK, 'test_kernels_gpu.py', 'test_fullsize_gpu.py'),
    # round 6: epilogue addends as 8-byte pieces on the way into the staging tile (the path of rounds 4-5) instead of whole rows from the flush
    ('RDFC_RSF', '0', 'addend and not f32'),
    # round 6: the bias gradient behind a norm layer by its own pass over dy (rdfc_channel_sum_acc) instead of the apply pass's sums
    ('RDFC_BIAS_IN_APPLY', '0', 'test_g_b2a or test_train_step_against_upstream_trace', 'test_modules_gpu.py'),
    # round 6: the stems' backward as three passes (rdfc_act_bwd, bias sum, weight gradient) instead of rdfc_conv2d_wgrad_act
    ('RDFC_STEM_FOLD', '0', 'test_train_step_against_upstream_trace or test_g_a2b', 'test_modules_gpu.py'),
    # round 6: one kernel row per 256-thread workgroup in the ring weight gradient's 64-co tile (rounds 3-5) instead of three rows in lockstep
    ('RDFC_WGRAD_KG', '0', '(test_conv_fwd_dgrad_wgrad or adjoint or scaling_by_two) and not f32', 'test_kernels_gpu.py', 'test_fullsize_gpu.py'),
]


_RESULTS = {}


def _run_all():
    """every variant's child pytest, FOUR at a time (round 6: one after the other they were 216 s of a 720 s suite -- each is mostly process
    start-up, host-side references and small launches; the children share the GPU like any two processes do).  Started by the first of
    the parametrised tests below, which then only look up their own child's outcome."""
    if _RESULTS:
        return
    from concurrent.futures import ThreadPoolExecutor

    def one(variant):
        name, value, expr = variant[:3]
        test_files = variant[3:] if len(variant) > 3 else ('test_kernels_gpu.py',)
        # (each child computes its fp32 references on the host with torch's thread pool: three pools of cpu_count threads each oversubscribe
        #  the cores several-fold and everything crawls -- a quarter of the cores per child)
        nthr = str(max(4, (os.cpu_count() or 16) // 4))
        env = dict(os.environ, OMP_NUM_THREADS=nthr, MKL_NUM_THREADS=nthr, **{name: value})
        try:
            r = subprocess.run([sys.executable, '-m', 'pytest'] + [os.path.join(ROOT, 'tests', f) for f in test_files] + ['-q', '-x', '-k', expr, '-p', 'no:cacheprovider'],
                               cwd=ROOT, env=env, capture_output=True, text=True, timeout=1200)
            return r.returncode, (r.stdout + r.stderr)[-2500:]
        except subprocess.TimeoutExpired as e:
            return -1, f'timed out: {e}'
    with ThreadPoolExecutor(max_workers=4) as ex:
        for variant, res in zip(VARIANTS, ex.map(one, VARIANTS)):
            _RESULTS[variant[:2]] = res


@pytest.mark.parametrize('variant', VARIANTS, ids=[f'{v[0]}={v[1]}' for v in VARIANTS])
def test_variant_passes_the_default_kernels_tests(variant):
    _run_all()
    rc, tail = _RESULTS[variant[:2]]
    assert rc == 0, tail
    assert ' passed' in tail and ' skipped' not in tail.splitlines()[-1], tail
