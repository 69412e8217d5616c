"""Training path on the HIP kernels (reference: train.py, training/me_task.py,
modules/losses/bound_loss.py, lr_scheduler/scheduler.py).  ``ops`` wraps the C-ABI training operators as autograd
functions (PyTorch keeps the tape; every tensor op on activations is a HIP kernel of libsome_amd.so), ``model`` is
the train-mode twin of ``midi_conforms``, ``task`` the loss / optimiser / step logic of ``MIDIExtractionTask``."""
