"""Student-initialisation checkpoint of the reference's /ckpt.py:1-23 (SURVEY 8(f) rank 4, 8(b) "State-dict contract").

The reference trains the radar student from `pillarnet_fullset_init.pth`, which `ckpt.py` derives from the trained LiDAR PillarNet
checkpoint: `epoch`, `it`, `optimizer_state`, `version` are carried over and every `model_state` entry appears twice, under its own
name (the frozen teacher) and under `'radar_' + name` (the student's starting point), teacher key first.  Entries whose shape
does not fit the student (the radar VFE's Linear is 15 -> 32, the LiDAR one 14 -> 32) are dropped later by `_load_state_dict`'s
shape test (detectors/detector3d_template.py), exactly as in the reference (detector3d_template.py:431-432).

    python -m radardistill_amd.ckpt ckpt/pillarnet_fullset_lidar.pth ckpt/pillarnet_fullset_init.pth
"""
from collections import OrderedDict

import torch


def student_init_state(lidar_ckpt):
    """Checkpoint dictionary of the trained LiDAR detector -> the distillation run's initial checkpoint dictionary."""
    new_state = {k: lidar_ckpt[k] for k in ('epoch', 'it', 'optimizer_state', 'version')}
    model_state = OrderedDict()
    for key, value in lidar_ckpt['model_state'].items():
        model_state[key] = value
        model_state['radar_' + key] = value
    new_state['model_state'] = model_state
    return new_state


def main(argv=None):
    import argparse
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("lidar_ckpt", help="trained LiDAR PillarNet checkpoint (ckpt/pillarnet_fullset_lidar.pth)")
    ap.add_argument("out", help="where to write the student-init checkpoint (ckpt/pillarnet_fullset_init.pth)")
    a = ap.parse_args(argv)
    # weights_only: nothing in the file is executed (tensors, numbers, strings and plain containers load; anything else is refused)
    ckpt = torch.load(a.lidar_ckpt, map_location="cpu", weights_only=True)
    torch.save(student_init_state(ckpt), a.out)


if __name__ == "__main__":
    main()
