"""Config reading / printing (reference utils/config_utils.py:11-52) without the lightning dependency."""
import copy
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


def _open_config(path: pathlib.Path, included_from=None):
    """-> (resolved key, dict).  Relative ``base_config`` entries are resolved against the working directory as the reference does
    (utils/config_utils.py:35 ``pathlib.Path(base_config)``), then against the including file's directory and its parent (a config
    tree copied elsewhere); a ``configs/<name>.yaml`` that exists nowhere on disk is served from the built-in configs."""
    from .. import configs
    candidates = [path]
    if not path.is_absolute() and included_from is not None:
        candidates += [included_from.parent / path, included_from.parent.parent / path]
    for c in candidates:
        if c.exists():
            c = c.resolve()
            with open(c, 'r', encoding='utf8') as f:
                return c.as_posix(), (yaml.safe_load(f) or {})
    builtin = configs.builtin_yaml(path)
    if builtin is not None:
        return f'builtin:{path.as_posix()}', builtin
    raise FileNotFoundError(f"config file '{path}' not found" + (f" (base_config of '{included_from}')" if included_from else '') +
                            f"; built-in configs: {', '.join('configs/' + n + '.yaml' for n in ['base'] + configs.config_names())}")


def read_full_config(config_path, _included_from=None) -> dict:
    """Recursive ``base_config`` inheritance with deep-dict override (utils/config_utils.py:19-41): bases are squashed in list
    order, then the file's own keys override them; keys the built-in configs do not know are kept as they are."""
    key, config = _open_config(pathlib.Path(config_path), _included_from)
    if key in _loaded:
        return _loaded[key]
    bases = config.get('base_config')
    if bases is None:
        _loaded[key] = config
        return config
    if not isinstance(bases, list):
        bases = [bases]
    here = None if key.startswith('builtin:') else pathlib.Path(key)
    merged = {}
    for base in bases:
        override_dict(merged, copy.deepcopy(read_full_config(pathlib.Path(base), here)))
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
