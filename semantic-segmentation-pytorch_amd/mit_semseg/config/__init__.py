from .defaults import _C as cfg
from .node import CfgNode
from .presets import PRESETS, preset

__all__ = ['cfg', 'CfgNode', 'PRESETS', 'preset', 'load']


def load(path_or_preset, opts=None):
    """A fresh config: defaults <- YAML file (or `preset:NAME`) <- `opts` pairs"""
    c = cfg.clone()
    if path_or_preset:
        if str(path_or_preset).startswith('preset:'):
            c.merge_from_other_cfg(PRESETS[str(path_or_preset)[7:]])
        else:
            c.merge_from_file(path_or_preset)
    c.merge_from_list(opts or [])
    return c
