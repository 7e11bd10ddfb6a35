"""`restore_ckpt` with the reference's signature (src/utils_keras.py:125-235) for this package's model objects.

The reference restores a TF2 object-graph / TF1 name-based checkpoint into Keras variables; here the weight set
(reference variable names -> arrays) is resolved by `weights.resolve_weights` - "_" = "running test: do not load any
ckpt" (:142-144), `.npz`, or a TF2 checkpoint prefix / directory read by `ckpt_reader` - and handed to the model
object, which packs it for the HIP library."""
import os


def restore_ckpt(model, ckpt_path_or_file, ema_decay=0.9998, skip_mismatch=True, exclude_layers=None):
    if ckpt_path_or_file == "_":
        return                                    # test mode: keep the random-init weight set
    if exclude_layers:
        raise ValueError("exclude_layers is a fine-tuning option (training) and is not supported on the HIP path")
    if os.path.isdir(str(ckpt_path_or_file)):
        from . import ckpt_reader
        latest = ckpt_reader.latest_checkpoint(ckpt_path_or_file)
        if latest is None:
            raise FileNotFoundError("no checkpoint in %s" % ckpt_path_or_file)
        ckpt_path_or_file = latest
    model.load_weights(_resolve(model, ckpt_path_or_file, ema_decay, skip_mismatch))


def _resolve(model, path, ema_decay, skip_mismatch):
    path = str(path)
    if path.endswith(".npz"):
        from . import weights
        return weights.load_weights(path)
    from . import ckpt_reader
    return ckpt_reader.load_checkpoint(path, model.config.as_dict(), use_ema=ema_decay > 0, skip_mismatch=skip_mismatch)
