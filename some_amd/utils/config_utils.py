"""Config reading / printing (reference utils/config_utils.py:11-52) without the lightning dependency."""
import os
import pathlib

import yaml

_loaded = {}


def override_dict(old_config: dict, new_config: dict):
    for k, v in new_config.items():
        if isinstance(v, dict) and isinstance(old_config.get(k), dict):
            override_dict(old_config[k], v)
        else:
            old_config[k] = v


def read_full_config(config_path) -> dict:
    """Recursive ``base_config`` inheritance with deep-dict override (utils/config_utils.py:19-41)."""
    config_path = pathlib.Path(config_path).resolve()
    key = config_path.as_posix()
    if key in _loaded:
        return _loaded[key]
    with open(config_path, 'r', encoding='utf8') as f:
        config = yaml.safe_load(f)
    bases = config.get('base_config')
    if bases is None:
        _loaded[key] = config
        return config
    if not isinstance(bases, list):
        bases = [bases]
    merged = {}
    for base in bases:
        override_dict(merged, read_full_config(pathlib.Path(base)))
    override_dict(merged, config)
    merged.pop('base_config')
    _loaded[key] = merged
    return merged


def print_config(config: dict):
    """Rank-zero only, same coloured 5-per-line format as the reference (utils/config_utils.py:44-52)."""
    if int(os.environ.get('RANK', '0')) != 0:
        return
    items = sorted(config.items())
    for i, (k, v) in enumerate(items):
        print(f"\033[0;33m{k}\033[0m: {v}", end='')
        if i < len(items) - 1:
            print(", ", end="")
        if i % 5 == 4:
            print()
    print()
