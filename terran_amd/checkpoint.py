"""Checkpoint registry entries for the MI355X backends.

Mirrors the reference registry contract (terran/checkpoint.py:29-103, 213-245): an entry
maps (task, alias) to a dotted class path; the task facades resolve it with
`get_class_for_checkpoint` and instantiate `Class(device=...)`.  The `id`s are the
reference's, so the same `<id>.pth` weight files under `$TERRAN_HOME/checkpoints`
(terran/checkpoint.py:118-150) are used.
"""
import importlib
import os
from pathlib import Path

# `alias` is the reference's ('gpu-realtime', terran/checkpoint.py:40,64,89), so every reference-valid call such as
# Detection(checkpoint='gpu-realtime') keeps working; 'mi355x-realtime' is an additional name for the same entry.
CHECKPOINTS = [
    {'id': 'b5d77fff', 'name': 'RetinaFace', 'task': 'face-detection', 'class': 'terran_amd.retinaface.RetinaFace',
     'alias': 'gpu-realtime', 'aliases': ('mi355x-realtime',), 'default': True, 'kind': 'retinaface'},
    {'id': 'd206e4b0', 'name': 'ArcFace', 'task': 'face-recognition', 'class': 'terran_amd.arcface.ArcFace',
     'alias': 'gpu-realtime', 'aliases': ('mi355x-realtime',), 'default': True, 'kind': 'arcface'},
    {'id': '11a769ad', 'name': 'OpenPose', 'task': 'pose-estimation', 'class': 'terran_amd.openpose.OpenPose',
     'alias': 'gpu-realtime', 'aliases': ('mi355x-realtime',), 'default': True, 'kind': 'openpose'},
]


def get_terran_home():
    return Path(os.environ.get('TERRAN_HOME', '~/.terran')).expanduser()


def get_checkpoint(task_name, alias):
    for c in CHECKPOINTS:
        if c['task'] != task_name:
            continue
        if (alias is None and c['default']) or alias in (c['alias'], c['id']) + tuple(c.get('aliases', ())):
            return c
    return None


def get_class_for_checkpoint(task_name, alias):
    """Same contract as terran/checkpoint.py:213-245 (ValueError when unknown)."""
    c = get_checkpoint(task_name, alias)
    if not c:
        raise ValueError('Checkpoint not found.')
    module_path, class_name = c['class'].rsplit('.', maxsplit=1)
    return getattr(importlib.import_module(module_path), class_name)


def find_checkpoint_file(kind):
    for c in CHECKPOINTS:
        if c['kind'] == kind:
            p = get_terran_home() / 'checkpoints' / ('%s.pth' % c['id'])
            return p if p.exists() else None
    return None
