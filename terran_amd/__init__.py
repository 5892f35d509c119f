"""terran_amd -- MI355X-native (gfx950) detect + embed + pose path behind Terran's callables.

    from terran_amd import face_detection, extract_features, pose_estimation

mirror `terran.face_detection / extract_features / pose_estimation`
(terran/__init__.py:4-5): lazily-constructed `Detection`, `Recognition`, `Estimation`.
"""
from .facade import Detection, Recognition, Estimation          # noqa: F401
from .retinaface import RetinaFace                                # noqa: F401
from .arcface import ArcFace                                      # noqa: F401
from .openpose import OpenPose, PoseOverflow                      # noqa: F401

face_detection = Detection(lazy=True)
extract_features = Recognition(lazy=True)
pose_estimation = Estimation(lazy=True)
