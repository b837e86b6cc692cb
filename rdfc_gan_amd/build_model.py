"""build_model(model_cfg) -- the registry upstream's train.py:84,103 / test.py:22,54 import from the
git-ignored lib/models/build_model.py.  `model_cfg` is the dict assembled at train.py:87-102:
type, device, distributed, args, is_train, [fixed_*_samples], num_classes, out_height, out_width,
label_wall, label_floor, label_ceiling."""
from .rdfc_gan import RDFCGAN

_MODELS = dict(RDFCGAN=RDFCGAN)


def build_model(model_cfg):
    cfg = dict(model_cfg)
    typ = cfg.pop('type')
    if typ not in _MODELS:
        raise KeyError(f'unknown model type {typ}; the HIP hot path provides {sorted(_MODELS)}')
    return _MODELS[typ](**cfg)
