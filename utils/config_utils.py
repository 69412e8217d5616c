from some_amd.utils.config_utils import read_full_config, print_config, override_dict  # noqa: F401  (drop-in shim)
