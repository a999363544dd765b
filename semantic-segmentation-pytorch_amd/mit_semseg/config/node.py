"""A small configuration tree with the surface of yacs' CfgNode that the reference's drivers use (train.py:208-262,
eval.py:150-193, test.py:130-200): attribute access to nested sections, `merge_from_file(yaml)`, `merge_from_list([key,
value, ...])` with dotted keys, `freeze()`, `clone()`, and `str(cfg)` = a YAML dump (train.py:238 writes it to DIR/config.yaml).
yacs is not a dependency of this build (it is absent from the MI355X image); PyYAML does the parsing.

Value coercion follows yacs: a string that is a Python literal ("(300, 375, 450)", "1e-4", "True") is evaluated, a value
replaces a default only if the types agree (tuple <-> list and int -> float are converted), unknown keys raise KeyError.
Sections accept NEW keys by attribute assignment while not frozen (train.py:255-259 adds TRAIN.batch_size / max_iters /
running_lr_*)."""
import ast
import copy
import io

import yaml


def _literal(v):
    if isinstance(v, str):
        try:
            return ast.literal_eval(v)
        except (ValueError, SyntaxError):
            return v
    return v


def _coerce(new, old, key):
    """`new` brought to the type of the default `old` (yacs _check_and_coerce_cfg_value_type)"""
    if old is None or type(new) is type(old):
        return new
    if isinstance(old, tuple) and isinstance(new, list):
        return tuple(new)
    if isinstance(old, list) and isinstance(new, tuple):
        return list(new)
    if isinstance(old, float) and isinstance(new, int) and not isinstance(new, bool):
        return float(new)
    if isinstance(old, str) and not isinstance(new, (dict, CfgNode)):
        return str(new)
    raise ValueError('config key %s: cannot replace %r (%s) by %r (%s)' % (key, old, type(old).__name__, new,
                                                                           type(new).__name__))


class CfgNode(dict):
    def __init__(self, init=None):
        super().__init__()
        object.__setattr__(self, '_frozen', False)
        for k, v in (init or {}).items():
            self[k] = CfgNode(v) if isinstance(v, dict) and not isinstance(v, CfgNode) else v

    # ---- attribute access -------------------------------------------------------------------------------------------
    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name)

    def __setattr__(self, name, value):
        if self._frozen:
            raise AttributeError('attempted to set %s on a frozen CfgNode' % name)
        self[name] = value

    # ---- merging ------------------------------------------------------------------------------------------------------
    def _merge(self, other, path):
        for k, v in other.items():
            full = '.'.join(path + [k])
            if k not in self:
                raise KeyError('non-existent config key: %s' % full)
            if isinstance(self[k], CfgNode):
                if not isinstance(v, dict):
                    raise ValueError('config key %s is a section' % full)
                self[k]._merge(v, path + [k])
            else:
                dict.__setitem__(self, k, _coerce(_literal(v), self[k], full))

    def merge_from_other_cfg(self, other):
        self._check_mutable()
        self._merge(other, [])

    def merge_from_file(self, path):
        self._check_mutable()
        with open(path, 'r') as f:
            loaded = yaml.safe_load(f) or {}
        self._merge(loaded, [])

    def merge_from_list(self, opts):
        """['TRAIN.lr_encoder', '0.01', 'DIR', 'ckpt/x', ...] (the `opts` remainder of the reference's command lines)"""
        self._check_mutable()
        opts = list(opts or [])
        if len(opts) % 2:
            raise ValueError('override list has odd length: %r; it must be a list of pairs' % (opts,))
        for key, value in zip(opts[0::2], opts[1::2]):
            node = self
            parts = key.split('.')
            for p in parts[:-1]:
                if p not in node or not isinstance(node[p], CfgNode):
                    raise KeyError('non-existent config key: %s' % key)
                node = node[p]
            if parts[-1] not in node:
                raise KeyError('non-existent config key: %s' % key)
            dict.__setitem__(node, parts[-1], _coerce(_literal(value), node[parts[-1]], key))

    # ---- state --------------------------------------------------------------------------------------------------------
    def _check_mutable(self):
        if self._frozen:
            raise AttributeError('the CfgNode is frozen')

    def _set_frozen(self, flag):
        object.__setattr__(self, '_frozen', flag)
        for v in self.values():
            if isinstance(v, CfgNode):
                v._set_frozen(flag)

    def freeze(self):
        self._set_frozen(True)

    def defrost(self):
        self._set_frozen(False)

    def is_frozen(self):
        return self._frozen

    def clone(self):
        return copy.deepcopy(self)

    def __deepcopy__(self, memo):
        out = CfgNode()
        for k, v in self.items():
            dict.__setitem__(out, k, copy.deepcopy(v, memo))
        return out

    # ---- dump ---------------------------------------------------------------------------------------------------------
    def to_dict(self):
        return {k: (v.to_dict() if isinstance(v, CfgNode) else (list(v) if isinstance(v, tuple) else v)) for k, v in self.items()}

    def dump(self, **kwargs):
        buf = io.StringIO()
        yaml.safe_dump(self.to_dict(), buf, default_flow_style=False, sort_keys=True, **kwargs)
        return buf.getvalue()

    def __str__(self):
        return self.dump()

    def __repr__(self):
        return 'CfgNode(%s)' % dict.__repr__(self)
