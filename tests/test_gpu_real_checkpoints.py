"""-m gpu, skipped unless REAL Terran weights are present: the reference's only published numbers for this path
(docs/usage/quickstart.rst:156-159 face boxes / landmarks, 218-222 cosine distances, 261-270 poses) on the images they
were produced from.

Neither the weights (terran/checkpoint.py: `terran checkpoint download`, ~500 MB) nor a network exist in the build or GPU
containers, so this is a HOOK: it runs where `$TERRAN_HOME/checkpoints/{b5d77fff,d206e4b0,11a769ad}.pth` are the released
weights (not the seeded files the registry test writes) and `$TERRAN_AMD_KAT_ASSETS` (default: the reference tree, when
mounted) holds `many-faces-raw.jpg`, `rw-1.jpg`, `rw-2.jpg`, `th.jpg`, `many-poses-raw.jpg` from the reference's
`examples/readme/` and `docs/assets/`.  Images are decoded with Pillow (terran/io/image.py does the same).
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
IDS = {'retinaface': 'b5d77fff', 'arcface': 'd206e4b0', 'openpose': '11a769ad'}
ASSET_DIRS = [os.environ.get('TERRAN_AMD_KAT_ASSETS'), '/root/reference/examples/readme', '/root/reference/docs/assets']


def _asset(name):
    for d in ASSET_DIRS:
        if d and os.path.exists(os.path.join(d, name)):
            from PIL import Image
            return np.asarray(Image.open(os.path.join(d, name)).convert('RGB'))
    pytest.skip('image %s not found (set TERRAN_AMD_KAT_ASSETS)' % name)


def _real_weights(kind):
    """Path of the released checkpoint, or skip.  The released files are hundreds of MB with trained statistics; the
    seeded stand-ins tests write have unit BatchNorm variances everywhere."""
    from terran_amd import checkpoint, weights
    p = checkpoint.find_checkpoint_file(kind)
    if p is None:
        pytest.skip('no %s.pth under $TERRAN_HOME/checkpoints' % IDS[kind])
    sd = weights.load_state(p)
    var_keys = [k for k in sd if k.endswith('running_var')]
    if var_keys and all(np.allclose(np.asarray(sd[k]), 1.0) for k in var_keys[:8]):
        pytest.skip('%s holds seeded synthetic weights, not the released checkpoint' % p)
    return p


def test_quickstart_face_detection_numbers():
    from terran_amd import face_detection
    _real_weights('retinaface')
    faces = face_detection(_asset('many-faces-raw.jpg'))
    want = [((1326, 1048, 1475, 1229), [(1360, 1115), (1427, 1116), (1390, 1156), (1367, 1183), (1421, 1183)]),
            ((590, 539, 690, 667), [(604, 583), (647, 586), (615, 612), (608, 633), (642, 635)]),
            ((1711, 408, 1812, 530), [(1731, 451), (1775, 451), (1747, 477), (1735, 499), (1769, 499)])]
    got = {tuple(int(v) for v in f['bbox']): [tuple(int(v) for v in p) for p in f['landmarks']] for f in faces}
    for bbox, lms in want:                       # quickstart.rst:156-159 prints the first three
        near = [b for b in got if max(abs(a - c) for a, c in zip(b, bbox)) <= 1]
        assert near, 'no detection within a pixel of %s among %d' % (bbox, len(got))
        assert max(abs(a - c) for p, q in zip(got[near[0]], lms) for a, c in zip(p, q)) <= 1


def test_quickstart_cosine_distances():
    from terran_amd import extract_features, face_detection
    _real_weights('retinaface')
    _real_weights('arcface')
    feats = []
    for name in ('rw-1.jpg', 'rw-2.jpg', 'th.jpg'):
        im = _asset(name)
        feats.append(extract_features(im, faces_per_image=face_detection(im))[0])

    def cosine(u, v):
        return 1.0 - float(np.dot(u, v) / (np.linalg.norm(u) * np.linalg.norm(v)))
    # quickstart.rst:218-222 (float32 features from the reference's GPU run: 1e-3, north_star's embedding tolerance)
    assert abs(cosine(feats[0], feats[1]) - 0.5384056568145752) < 1e-3
    assert abs(cosine(feats[0], feats[2]) - 1.0747144743800163) < 1e-3
    assert abs(cosine(feats[1], feats[2]) - 1.06807991117239) < 1e-3


def test_quickstart_pose_numbers():
    from terran_amd import pose_estimation
    _real_weights('openpose')
    poses = pose_estimation(_asset('many-poses-raw.jpg'))
    assert len(poses) == 6                        # quickstart.rst:261-262
    kp = poses[0]['keypoints']                    # 263-270 prints rows 0, 1 and the last three
    assert kp.dtype == np.int32 and kp.shape == (18, 3)
    assert kp[0].tolist() == [0, 0, 0] and kp[1].tolist() == [714, 351, 1]
    assert kp[15].tolist() == [0, 0, 0] and kp[16].tolist() == [725, 286, 1] and kp[17].tolist() == [678, 292, 1]
