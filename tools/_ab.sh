python -m pytest tests/test_checkpoint_gpu.py -x -q 2>&1 | tail -4
python - <<'PY'
import sys, numpy as np, torch
sys.path.insert(0,'tests'); sys.path.insert(0,'tests/golden')
from test_modules_gpu import *
from conftest import *
import test_modules_gpu as T
g = np.load('tests/golden/g_a2b_nlspn_small.npz')
from rdfc_gan_amd.generator import RDFGenerator
for dtype in (torch.bfloat16, torch.float16):
    setup_dtype(dtype)
    net = RDFGenerator(pretrained_on_imagenet=False, use_nlspn_refine=True, nlspn_configs=NLSPN_CFG)
    net.load_state_dict(det_state_dict(net.state_dict(), 0)); net.to(DEV)
    rgb, depth, normal = (t.to(DEV) for t in gen_inputs('g_a2b', 2, 60, 76))
    res = run(net, (rgb, depth, normal), 'g_a2b_nlspn_small')
    for k in g.files:
        if k.startswith(('eval.', 'train.')):
            d = res[k].detach().cpu().numpy().astype(np.float64) - g[k]
            a = np.abs(d)
            print(dtype, k, 'max', a.max(), 'rms', np.sqrt((d*d).mean()), 'q99', np.quantile(a, 0.99), 'q999', np.quantile(a, 0.999))
PY
