from some_amd.modules.model.Gmidi_conform import midi_conforms  # noqa: F401  (drop-in shim for model_cls: modules.model.Gmidi_conform.midi_conforms)
