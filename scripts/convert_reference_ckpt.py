"""Reference ``.ckpt`` (HDF5, written by mimikit's ``CheckpointBank.save``, checkpoint.py:56-93) -> the flat ``.npz`` that
``mimikit_amd.checkpoint.load_network`` reads.

    python scripts/convert_reference_ckpt.py trainings/my_model/epoch=20.ckpt out/my_model/epoch=20.ckpt

Needs ``h5py`` (NOT installed in the build image of this repo, so this script is untested there - it only uses the file
layout the reference writes: group ``network`` with attribute ``config`` (YAML) and the sub-group ``state_dict`` whose
datasets are the tensors; file attributes ``dataset`` / ``training`` (YAML)).  h5mapper stores a TensorDict as one dataset
per key; nested groups, should a version produce them, are flattened by joining the path with '.'.
"""
import sys

import numpy as np


def main(src, dst):
    import h5py

    out = {}
    with h5py.File(src, "r") as f:
        net = f["network"]
        out["__network_config__"] = np.asarray(str(net.attrs["config"]))
        if "dataset" in f.attrs:
            out["__dataset_config__"] = np.asarray(str(f.attrs["dataset"]))
        if "training" in f.attrs:
            out["__training_config__"] = np.asarray(str(f.attrs["training"]))

        def walk(group, prefix):
            for name, item in group.items():
                key = f"{prefix}.{name}" if prefix else name
                if isinstance(item, h5py.Dataset):
                    out[key] = np.asarray(item[()])
                else:
                    walk(item, key)

        walk(net["state_dict"], "")
    with open(dst, "wb") as fh:
        np.savez(fh, **out)
    print(f"{dst}: {len(out) - 3} tensors")


if __name__ == "__main__":
    if len(sys.argv) != 3:
        raise SystemExit(__doc__)
    main(sys.argv[1], sys.argv[2])
