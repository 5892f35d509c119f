"""Generate the golden fixtures under tests/golden/ FROM THE REFERENCE ITSELF.

CONTAINER-ONLY: needs /root/reference (imported through tests/golden/ref_import.py
with stub parent packages).  The reference never travels; only the small .npz
files written here are committed.  Re-run:  python tests/golden/make_golden.py

Every fixture holds inputs (or the seeds that regenerate them) and the outputs the
REFERENCE code produced in this container (torch 2.10.0 CPU, numpy 2.2, Pillow 12.2)
with seeded random weights from terran_amd/weights.py.  Fixtures that cross one of
the three third-party shims (cv2.resize, torchvision nms, skimage Umeyama -- absent
here, restated in oracle/) say so in their `note` field.
"""
import importlib.util
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
os.environ.setdefault('TERRAN_HOME', '/tmp/terran_home')

from tests.golden import ref_import as R   # noqa: E402
from terran_amd import weights, synth      # noqa: E402


def _load(cls, st):
    m = cls()
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in st.items()})
    m.eval()
    return m


def save(name, **kw):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **kw)
    print('wrote %-28s %7.1f KB' % (name, os.path.getsize(path) / 1024))


def flat_dets(dets):
    counts = np.array([len(d) for d in dets], np.int64)
    bbox = np.array([o['bbox'] for d in dets for o in d]).reshape(-1, 4)
    lmk = np.array([o['landmarks'] for d in dets for o in d]).reshape(-1, 5, 2)
    score = np.array([o['score'] for d in dets for o in d]).reshape(-1)
    return counts, bbox, lmk, score


def flat_poses(poses):
    counts = np.array([len(p) for p in poses], np.int64)
    kp = np.array([o['keypoints'] for p in poses for o in p], np.int32).reshape(-1, 18, 3)
    sc = np.array([o['score'] for p in poses for o in p], np.float64).reshape(-1)
    return counts, kp, sc


def main():
    tdev = torch.device('cpu')
    # ------------------------------------------------------------------ nets
    RM = R.ref('terran.face.detection.retinaface.model')
    AM = R.ref('terran.face.recognition.arcface.model')
    PM = R.ref('terran.pose.openpose.model')
    st_r = weights.make_retinaface_state()
    st_a = weights.make_arcface_state()
    st_p = weights.make_openpose_state()

    img = synth.frames(21, 2, 64, 96)
    x = torch.from_numpy(img.astype(np.float32)).permute(0, 3, 1, 2).flip(1).contiguous()
    with torch.no_grad():
        outs = _load(RM.RetinaFace, st_r)(x)
    save('nets_retinaface.npz', images=img, note='RetinaFace module, seed %d, input = BGR 0..255' % weights.SEED_RETINAFACE,
         **{'out%d' % i: o.numpy() for i, o in enumerate(outs)})

    crops = np.random.default_rng(22).integers(0, 256, (2, 3, 112, 112), dtype=np.uint8)
    with torch.no_grad():
        emb = _load(AM.FaceResNet100, st_a)(torch.from_numpy(crops.astype(np.float32))).numpy()
    save('nets_arcface.npz', crops=crops, embeddings=emb, note='FaceResNet100 module, seed %d' % weights.SEED_ARCFACE)

    img = synth.frames(23, 1, 64, 96)
    xp = torch.from_numpy((np.transpose(img, (0, 3, 1, 2)).astype(np.float32) / 255.0 - 0.5))
    with torch.no_grad():
        paf, hm = _load(PM.BodyPoseModel, st_p)(xp)
    save('nets_openpose.npz', images=img, pafs=paf.numpy(), heatmaps=hm.numpy(),
         note='BodyPoseModel module, seed %d' % weights.SEED_OPENPOSE)

    # --------------------------------------------------------- RetinaFace.call
    RW = R.ref('terran.face.detection.retinaface.wrapper')

    def load_r():
        m = _load(RM.RetinaFace, st_r)
        m.register_forward_pre_hook(lambda mod, a: (a[0].contiguous(),))   # torch>=2 stride issue, model.py:285
        return m
    RW.load_model = load_r
    det = RW.RetinaFace(device=tdev)
    frames = synth.frames(0, 2, 208, 277)
    counts, bbox, lmk, score = flat_dets(det.call(frames))
    save('retinaface_call.npz', frames_seed=0, shape=np.array([2, 208, 277]), counts=counts, bbox=bbox,
         landmarks=lmk, score=score,
         note='reference RetinaFace.call(frames(0,2,208,277)); NMS = torchvision shim (oracle.retinaface_post.nms)')
    for s, ref in det.anchor_references.items():
        print('anchor ref', s, ref.numpy().tolist())
    save('retinaface_anchors.npz', **{'s%d' % s: ref.numpy() for s, ref in det.anchor_references.items()},
         plane_s16_3x4=R.ref('terran.face.detection.retinaface.anchors').anchors_plane(
             det.anchor_references[16], 3, 4, 16).numpy(),
         note='generate_anchor_reference + anchors_plane(ref16, 3, 4, 16)')

    # ------------------------------------------------------------ ArcFace.call
    AW = R.ref('terran.face.recognition.arcface.wrapper')
    AW.load_model = lambda: _load(AM.FaceResNet100, st_a)
    arc = AW.ArcFace(device=tdev)
    image = synth.frames(5, 1, 120, 160)[0]
    lms = synth.landmarks(6, 3, 120, 160)
    lms[2] += 60.0           # partly outside the image -> fill colour path
    crops = np.stack([AW.preprocess_face(image, l) for l in lms])
    feats = arc.call([image], [[{'landmarks': l} for l in lms]])[0]
    small = image[:100, :80]
    crop_nl = AW.preprocess_face_no_landmarks(small)
    feat_nl = arc.call([small], None)
    empty = arc.call([image], [[]])[0]
    save('arcface_call.npz', image=image, landmarks=lms, crops=crops, features=feats,
         crop_nolm=crop_nl, feature_nolm=feat_nl, empty_shape=np.array(empty.shape),
         empty_dtype=str(empty.dtype),
         note='reference preprocess_face (real PIL; Umeyama = skimage shim -> oracle.arcface_pre.umeyama), '
              'ArcFace.call with/without landmarks, empty case')

    # ----------------------------------------------------------- OpenPose.call
    PW = R.ref('terran.pose.openpose.wrapper')

    class Fake(torch.nn.Module):
        def __init__(self, p, h):
            super().__init__()
            self.p, self.h = p, h

        def forward(self, x):
            return self.p, self.h
    cases = [(10, 1, 23, 40), (11, 4, 23, 40), (12, 8, 46, 82), (13, 3, 16, 24), (14, 6, 30, 30),
             (15, 0, 10, 12), (16, 12, 46, 82)]
    out = {}
    for seed, P, h, w in cases:
        hmap, pafs = synth.pose_maps_batch(seed, 2, P, h, w)
        PW.load_model = lambda: Fake(torch.from_numpy(pafs), torch.from_numpy(hmap))
        m = PW.OpenPose(device=tdev, short_side=8 * h)
        c, kp, sc = flat_poses(m.call(np.zeros((2, 8 * h, 8 * w, 3), np.uint8)))
        out['c%d_counts' % seed], out['c%d_keypoints' % seed], out['c%d_scores' % seed] = c, kp, sc
        out['c%d_mapsum' % seed] = np.array([hmap.astype(np.float64).sum(), pafs.astype(np.float64).sum()])
    save('openpose_call.npz', cases=np.array(cases), **out,
         note='reference OpenPose.call with the model replaced by synth.pose_maps_batch(seed,2,P,h,w); '
              'input frames are (8h,8w) zeros so the cv2 shim is an identity')

    # adversarial maps (noise-free): exact plateaus, exact score ties, coincident peaks / zero-length limbs
    adv = {}
    adv_cases = [('plateau', 3, 20, 28), ('plateau', 4, 23, 40), ('twins', 3, 20, 28), ('twins', 5, 23, 40),
                 ('coincident', 3, 20, 28), ('coincident', 6, 23, 40)]
    for kind, seed, h, w in adv_cases:
        hmap, pafs = synth.pose_maps_adversarial(kind, seed, h, w)
        PW.load_model = lambda: Fake(torch.from_numpy(pafs[None]), torch.from_numpy(hmap[None]))
        m = PW.OpenPose(device=tdev, short_side=8 * h)
        c, kp, sc = flat_poses(m.call(np.zeros((1, 8 * h, 8 * w, 3), np.uint8)))
        key = '%s_%d' % (kind, seed)
        adv[key + '_counts'], adv[key + '_keypoints'], adv[key + '_scores'] = c, kp, sc
    save('openpose_adversarial.npz', cases=np.array([(k, str(s), str(h), str(w)) for k, s, h, w in adv_cases]), **adv,
         note='reference OpenPose.call with the model replaced by synth.pose_maps_adversarial(kind, seed, h, w)')

    # ---- end to end with the DECODER weights: frames that carry pose maps -> non-empty humans --------------
    st_d = weights.make_openpose_decoder_state()
    PW.load_model = lambda: _load(PM.BodyPoseModel, st_d)
    e2e = {}
    m = PW.OpenPose(device=tdev, short_side=96)
    coded = synth.pose_code_frames(60, 2, 96, 128, 3)
    e2e['a_counts'], e2e['a_keypoints'], e2e['a_scores'] = flat_poses(m.call(coded))             # scale 1
    big = synth.upscale_for_resize(coded, 288, 384)
    e2e['b_counts'], e2e['b_keypoints'], e2e['b_scores'] = flat_poses(m.call(big))                # scale 1/3 (cv2 shim)
    m184 = PW.OpenPose(device=tdev, short_side=184)
    hd = synth.upscale_for_resize(synth.pose_code_frames(61, 1, 184, 327, 4), 1080, 1920)
    e2e['c_counts'], e2e['c_keypoints'], e2e['c_scores'] = flat_poses(m184.call(hd))              # 1080p -> 184 x 327
    save('openpose_e2e.npz', **e2e,
         note='reference OpenPose.call, weights.make_openpose_decoder_state(): a = pose_code_frames(60,2,96,128,3) at '
              'short_side 96; b = the same upscaled to 288x384 (upscale_for_resize; crosses the cv2 shim); c = '
              'pose_code_frames(61,1,184,327,4) upscaled to 1080x1920 at short_side 184')

    # ------------------------------------------------------------------ bicubic
    m = np.random.default_rng(4).normal(0, 1, (1, 8, 6, 7)).astype(np.float32)
    up = torch.nn.functional.interpolate(torch.from_numpy(m), scale_factor=8, mode='bicubic',
                                         align_corners=False).numpy()
    save('bicubic.npz', maps=m, up=up, note='torch F.interpolate(scale_factor=8, bicubic, align_corners=False)')

    # ------------------------------------------------------------------ PIL pins
    from PIL import Image
    rng = np.random.default_rng(31)
    imgs, mats, warps = [], [], []
    for t in range(4):
        H, W = int(rng.integers(60, 140)), int(rng.integers(60, 140))
        im = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
        a = np.array([rng.uniform(0.4, 1.6), rng.uniform(-0.4, 0.4), rng.uniform(-20, 20),
                      rng.uniform(-0.4, 0.4), rng.uniform(0.4, 1.6), rng.uniform(-20, 20)])
        warp = np.array(Image.fromarray(im).transform(size=(112, 112), method=Image.AFFINE, data=a,
                                                      resample=Image.BILINEAR, fillcolor=0))
        imgs.append(im)
        mats.append(a)
        warps.append(warp)
    rs_in = rng.integers(0, 256, (97, 61, 3), dtype=np.uint8)
    rs_out = np.asarray(Image.fromarray(rs_in).resize((70, 112)))
    save('pil_pins.npz', **{'img%d' % i: im for i, im in enumerate(imgs)}, mats=np.array(mats),
         warps=np.array(warps), resize_in=rs_in, resize_out=rs_out,
         note='real Pillow: Image.transform(AFFINE,BILINEAR,fillcolor=0) and Image.resize default filter')

    # ------------------------------------------------------------------ facades
    def load_file(modname, rel):
        spec = importlib.util.spec_from_file_location(modname, os.path.join(R.REF_ROOT, 'terran', rel))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        return mod
    sys.modules['terran.face.detection.retinaface'].RetinaFace = RW.RetinaFace
    sys.modules['terran.face.recognition.arcface'].ArcFace = AW.ArcFace
    sys.modules['terran.pose.openpose'].OpenPose = PW.OpenPose
    FD = load_file('terran_ref_face_detection', 'face/detection/__init__.py')
    FR = load_file('terran_ref_face_recognition', 'face/recognition/__init__.py')
    FP = load_file('terran_ref_pose', 'pose/__init__.py')

    # C1: Detection(short_side=208) on one 640x480 frame (BASELINE.json configs[0])
    frame = synth.frames(0, 1, 480, 640)[0]
    d = FD.Detection(short_side=208, device=tdev)
    res = d(frame)
    c, bb, lm, sc = flat_dets([res])
    lst = d([frame[:400, :500], frame])           # list input -> per-image scales + pad merge
    c2, bb2, lm2, sc2 = flat_dets(lst)
    save('facade_detection.npz', frame_seed=0, counts=c, bbox=bb, landmarks=lm, score=sc,
         l_counts=c2, l_bbox=bb2, l_landmarks=lm2, l_score=sc2,
         note='reference Detection(short_side=208)(frames(0,1,480,640)[0]) and on the list '
              '[frame[:400,:500], frame]; crosses the cv2.resize shim (oracle.facade.cv2_resize_linear)')

    # pose facade, random-weight net (no person assembles: the empty path), list input, crosses the cv2 shim
    PW.load_model = lambda: _load(PM.BodyPoseModel, st_p)
    e = FP.Estimation(short_side=64, device=tdev)
    f2 = synth.frames(7, 1, 96, 128)[0]
    pr = e([f2[:80, :100], f2])
    c, kp, sc = flat_poses(pr)
    # pose facade with the decoder weights: non-empty results through pad-merge (odd pads: ceil top/left), un-pad and
    # the present == 0 reset (pose/__init__.py:94-122); single image; ndarray batch with a resize
    PW.load_model = lambda: _load(PM.BodyPoseModel, st_d)
    e = FP.Estimation(short_side=96, device=tdev)
    coded = synth.pose_code_frames(62, 2, 96, 128, 3)
    lst = e([coded[0][9:88, 14:115], coded[1]])             # 79 x 101 inside 96 x 128: pads (9, 8) x (14, 13)
    lc, lkp, lsc = flat_poses(lst)
    one = e(coded[1])
    oc, okp, osc = flat_poses([one])
    big = synth.upscale_for_resize(coded, 288, 384)
    bc, bkp, bsc = flat_poses(e(big))
    save('facade_pose.npz', frame_seed=7, counts=c, keypoints=kp, scores=sc,
         l_counts=lc, l_keypoints=lkp, l_scores=lsc, o_counts=oc, o_keypoints=okp, o_scores=osc,
         b_counts=bc, b_keypoints=bkp, b_scores=bsc,
         note='reference Estimation: (counts..) short_side=64 on list [frames(7,1,96,128)[0][:80,:100], same full], '
              'random-weight net; l_* = Estimation(short_side=96), decoder weights, list [pose_code_frames(62,2,96,128,3)'
              '[0][9:88,14:115], [1]]; o_* = single image [1]; b_* = the batch upscaled to 288x384 (cv2 shim)')

    # recognition facade rank handling
    rec = FR.Recognition(device=tdev)
    one = rec(image, {'landmarks': lms[0]})
    lst1 = rec(image, [{'landmarks': lms[0]}, {'landmarks': lms[1]}])
    many = rec([image, image], [[{'landmarks': lms[0]}], []])
    save('facade_recognition.npz', one=one, lst=lst1, many0=many[0], many1=many[1],
         many1_dtype=str(many[1].dtype), note='reference Recognition rank handling on arcface_call.npz image')

    # the reference's own tables, straight from its modules: state_dict keys + shapes of the three networks (default
    # constructors) and the limb tables of the pose wrapper.  terran_amd/arch.py (which the packer AND the oracle walk)
    # is compared with these, so a shared table error cannot hide behind oracle == product agreement.
    def keys_shapes(module):
        sd = module.state_dict()
        names = [k for k in sd if not k.endswith('num_batches_tracked')]
        return np.array(names), np.array([list(sd[k].shape) + [0] * (4 - sd[k].dim()) for k in names], np.int64)
    rk, rs = keys_shapes(RM.RetinaFace())
    ak, as_ = keys_shapes(AM.FaceResNet100())
    pk, ps = keys_shapes(PM.BodyPoseModel())
    save('arch_tables.npz', retinaface_keys=rk, retinaface_shapes=rs, arcface_keys=ak, arcface_shapes=as_,
         openpose_keys=pk, openpose_shapes=ps, map_idx=np.array(PW.map_idx, np.int64), limbseq=np.array(PW.limbseq, np.int64),
         note='reference modules (default constructors): state_dict keys / shapes (padded to 4 dims with 0) and '
              'pose/openpose/wrapper.py:12-24 map_idx / limbseq')


if __name__ == '__main__':
    main()
