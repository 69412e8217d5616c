from some_amd.modules.rmvpe.spec import MelSpectrogram  # noqa: F401  (drop-in shim)
