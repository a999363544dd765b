"""config_golden.json: the seven shipped YAML configurations of the reference (config/ade20k-*.yaml), parsed with PyYAML +
literal evaluation of string scalars (what yacs' merge_from_file does with them), for tests/test_config_cpu.py.  Build
container only:  python tests/golden/make_config_golden.py"""
import ast
import glob
import json
import os

import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get('SEMSEG_REFERENCE', '/root/reference')


def lit(v):
    if isinstance(v, dict):
        return {k: lit(x) for k, x in v.items()}
    if isinstance(v, str):
        try:
            v = ast.literal_eval(v)
        except (ValueError, SyntaxError):
            pass
    return list(v) if isinstance(v, tuple) else v


out = {}
for f in sorted(glob.glob(os.path.join(REF, 'config', '*.yaml'))):
    out[os.path.basename(f)[:-5]] = lit(yaml.safe_load(open(f)))
json.dump(out, open(os.path.join(HERE, 'config_golden.json'), 'w'), indent=1, sort_keys=True)
print(len(out), 'configs')
