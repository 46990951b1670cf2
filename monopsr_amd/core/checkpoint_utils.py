"""Weight import for the instance path, honouring the reference's variable names
(core/checkpoint_utils.py:64-117 and MonoPSRModel.get_variable_restore_map, monopsr_model.py:1230-1266).

Checkpoints come in as a flat name -> array mapping: read straight from a TensorFlow checkpoint
(`load_checkpoint(prefix)` -> core/tf_checkpoint.py, no TensorFlow needed), from an .npz, or a dict.  What this
module reproduces is the reference's NAME LOGIC:

  * Object-Detection-API checkpoints hold ONE trunk under `FirstStageFeatureExtractor/resnet_v1_101/...`; the
    reference restores it into BOTH of its trunks by rewriting `FirstStageFeatureExtractor_crop/` and
    `FirstStageFeatureExtractor_full/` to that scope (checkpoint_utils.py:82-106);
  * only variables present in the checkpoint are restored, the rest keep their initial values
    (`get_variables_available_in_checkpoint`); shapes must match;
  * a MonoPSR checkpoint stores every variable under its own name (plus optimizer slots / EMA shadows, ignored).
"""
import numpy as np

from monopsr_amd.core import weights as W

OD_API_SCOPE = "FirstStageFeatureExtractor/"
TRUNK_SCOPES = ("FirstStageFeatureExtractor_crop/", "FirstStageFeatureExtractor_full/")
_IGNORED_SUFFIXES = ("/Adam", "/Adam_1", "/ExponentialMovingAverage", "/Momentum")


def expected_variables(scopes=(W.CROP_SCOPE, W.FULL_SCOPE), feat_elems=18432):
    """name -> shape of every variable of the instance path (TF layouts: conv HWIO, FC (in, out))."""
    return W.variable_shapes(scopes=scopes, feat_elems=feat_elems)


def restore_obj_detection_api_weights(weights, checkpoint, strict_shapes=True):
    """Fill both trunks of `weights` (in place) from an OD-API style mapping `checkpoint`
    (`FirstStageFeatureExtractor/resnet_v1_101/...`).  Returns the list of restored variable names."""
    restored = []
    for name in list(weights.keys()):
        for scope in TRUNK_SCOPES:
            if name.startswith(scope):
                src = OD_API_SCOPE + name[len(scope):]
                if src in checkpoint:
                    arr = np.asarray(checkpoint[src], dtype=np.float32)
                    if arr.shape != weights[name].shape:
                        if strict_shapes:
                            raise ValueError("shape mismatch for %s: checkpoint %s vs model %s" %
                                             (name, arr.shape, weights[name].shape))
                        continue
                    weights[name] = arr
                    restored.append(name)
    return restored


def restore_monopsr_weights(weights, checkpoint, strict_shapes=True):
    """Fill `weights` (in place) from a MonoPSR checkpoint mapping (variables under their own names)."""
    restored = []
    for name, arr in checkpoint.items():
        if name.endswith(_IGNORED_SUFFIXES) or name == "global_step":
            continue
        if name in weights:
            arr = np.asarray(arr, dtype=np.float32)
            if arr.shape != weights[name].shape:
                if strict_shapes:
                    raise ValueError("shape mismatch for %s: checkpoint %s vs model %s" %
                                     (name, arr.shape, weights[name].shape))
                continue
            weights[name] = arr
            restored.append(name)
    return restored


def load_checkpoint(path):
    """name -> array from a TensorFlow V2 checkpoint (prefix, .index file or directory with a `checkpoint` state
    file) or an .npz."""
    if str(path).endswith(".npz"):
        return load_npz(path)
    from monopsr_amd.core import tf_checkpoint
    return tf_checkpoint.read_checkpoint(path)


def save_checkpoint(prefix, weights, global_step=None):
    """Write `weights` (plus `global_step`) as a TensorFlow V2 checkpoint `<prefix>[-<global_step, 8 digits>]`, the
    naming tf.train.Saver(pad_step_number=True) uses (core/trainer.py:86, :190-193), and update the directory's
    `checkpoint` state file.  Returns the prefix written."""
    import os
    from monopsr_amd.core import tf_checkpoint
    tensors = dict(weights)
    if global_step is not None:
        tensors["global_step"] = np.asarray(global_step, dtype=np.int64)
        prefix = "%s-%08d" % (prefix, int(global_step))
    tf_checkpoint.write_checkpoint(prefix, tensors)
    tf_checkpoint.write_checkpoint_state(os.path.dirname(prefix) or ".", os.path.basename(prefix))
    return prefix


def load_npz(path):
    with np.load(path) as f:
        return {k: f[k] for k in f.files}


def save_npz(path, weights):
    np.savez(path, **weights)
