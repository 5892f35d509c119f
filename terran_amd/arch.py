"""Architecture tables for the three networks on the hot path.

These tables are this repo's own statement of the graphs;
parameter *names* follow the reference's `state_dict` keys so that a real
Terran checkpoint (`torch.load(.pth)` -> dict of tensors) can be packed directly:

* RetinaFace-mnet     reference: terran/face/detection/retinaface/model.py:6-341
* ArcFace IR-ResNet100 reference: terran/face/recognition/arcface/model.py:4-97
* OpenPose body 2017   reference: terran/pose/openpose/model.py:27-141

Nothing here executes a network; `oracle/nets.py` (CPU) and `terran_amd/pack.py`
(HIP op-program builder) both walk these tables.
"""

# ----------------------------------------------------------------------------
# RetinaFace-mnet
# ----------------------------------------------------------------------------
# (name_prefix, cin, cout, stride_of_depthwise, return_both)
RETINA_SCALES = [
    [(8, 16, 2, False), (16, 32, 1, False), (32, 32, 2, False),
     (32, 64, 1, False), (64, 64, 2, True)],
    [(64, 128, 1, False), (128, 128, 1, False), (128, 128, 1, False),
     (128, 128, 1, False), (128, 128, 1, False), (128, 128, 2, True)],
]
RETINA_BASE_BN_EPS = 1e-5       # model.py:28,37,62,66,98
RETINA_REFINER_BN_EPS = 2e-5    # model.py:128-150,180-202
RETINA_STRIDES = (32, 16, 8)    # output / concat order, wrapper.py:100
RETINA_NUM_ANCHORS = 2          # model.py:255


def retinaface_param_specs():
    """Yield (key, shape, kind) for every tensor in the RetinaFace state_dict.

    kind in {'conv', 'dw', 'bias', 'bn'}; 'bn' expands to weight/bias/running_mean/
    running_var/num_batches_tracked under `key`.
    """
    specs = []

    def conv(key, cout, cin, k, bias=False, dw=False):
        specs.append((key + '.weight', (cout, 1 if dw else cin, k, k), 'dw' if dw else 'conv'))
        if bias:
            specs.append((key + '.bias', (cout,), 'bias'))

    def bn(key, c):
        specs.append((key, (c,), 'bn'))

    conv('base.first_conv_block.0', 8, 3, 3)
    bn('base.first_conv_block.1', 8)
    conv('base.first_conv_block.3', 8, 8, 3, dw=True)
    bn('base.first_conv_block.4', 8)
    for si, scale in enumerate(RETINA_SCALES):
        for bi, (cin, cout, stride, both) in enumerate(scale):
            p = 'base.scales.%d.%d' % (si, bi)
            conv(p + '.conv_block.0', cout, cin, 1)
            bn(p + '.conv_block.1', cout)
            conv(p + '.sep_block.0', cout, cout, 3, dw=True)
            bn(p + '.sep_block.1', cout)
    p = 'base.final_conv.0'
    conv(p + '.conv_block.0', 256, 128, 1)
    bn(p + '.conv_block.1', 256)
    conv(p + '.sep_block.0', 256, 256, 3, dw=True)
    bn(p + '.sep_block.1', 256)
    conv('base.final_conv.1', 256, 256, 1)
    bn('base.final_conv.2', 256)

    for s, cin in ((8, 64), (16, 128), (32, 256)):
        conv('refiner.conv_stride%d.0' % s, 64, cin, 1, bias=True)
        bn('refiner.conv_stride%d.1' % s, 64)
    for s in (8, 16):
        conv('refiner.aggr_stride%d.0' % s, 64, 64, 3, bias=True)
        bn('refiner.aggr_stride%d.1' % s, 64)
    for s in (8, 16, 32):
        p = 'refiner.context_stride%d' % s
        conv(p + '.context_3x3.0', 32, 64, 3, bias=True)
        bn(p + '.context_3x3.1', 32)
        conv(p + '.dimension_reducer.0', 16, 64, 3, bias=True)
        bn(p + '.dimension_reducer.1', 16)
        conv(p + '.context_5x5.0', 16, 16, 3, bias=True)
        bn(p + '.context_5x5.1', 16)
        conv(p + '.context_7x7.0', 16, 16, 3, bias=True)
        bn(p + '.context_7x7.1', 16)
        conv(p + '.context_7x7.3', 16, 16, 3, bias=True)
        bn(p + '.context_7x7.4', 16)
    A = RETINA_NUM_ANCHORS
    for head, per in (('cls', 2), ('bbox', 4), ('landmark', 10)):
        for s in (8, 16, 32):
            conv('outputs.%s_stride%d' % (head, s), per * A, 64, 1, bias=True)
    return specs


# ----------------------------------------------------------------------------
# ArcFace IR-ResNet100
# ----------------------------------------------------------------------------
ARC_UNITS = (3, 13, 30, 3)
ARC_CHANNELS = (64, 64, 128, 256, 512)
ARC_BN_EPS = 2e-5
ARC_MEAN = 127.5
ARC_STD = 0.0078125


def arcface_units():
    """Yield (stage, unit, cin, cout, stride, has_conv_shortcut)."""
    for st, n in enumerate(ARC_UNITS):
        cin, cout = ARC_CHANNELS[st], ARC_CHANNELS[st + 1]
        for u in range(n):
            if u == 0:
                yield st, u, cin, cout, 2, True      # model.py:9 (in==out but stride 2 => conv shortcut)
            else:
                yield st, u, cout, cout, 1, False


def arcface_param_specs():
    specs = []
    specs.append(('initial_layer.0.weight', (64, 3, 3, 3), 'conv'))
    specs.append(('initial_layer.1', (64,), 'bn'))
    specs.append(('initial_layer.2.weight', (64,), 'prelu'))
    for st, u, cin, cout, stride, sc in arcface_units():
        p = 'stages.%d.%d' % (st, u)
        specs.append((p + '.body.0', (cin,), 'bn'))
        specs.append((p + '.body.1.weight', (cout, cin, 3, 3), 'conv'))
        specs.append((p + '.body.2', (cout,), 'bn'))
        specs.append((p + '.body.3.weight', (cout,), 'prelu'))
        specs.append((p + '.body.4.weight', (cout, cout, 3, 3), 'conv'))
        specs.append((p + '.body.5', (cout,), 'bn_res'))
        if sc:
            specs.append((p + '.shortcut.0.weight', (cout, cin, 1, 1), 'conv'))
            specs.append((p + '.shortcut.1', (cout,), 'bn_res'))
    specs.append(('final_layer.0', (512,), 'bn'))
    specs.append(('final_layer.3.weight', (512, 7 * 7 * 512), 'linear'))
    specs.append(('final_layer.3.bias', (512,), 'bias'))
    specs.append(('final_layer.4', (512,), 'bn'))
    return specs


# ----------------------------------------------------------------------------
# OpenPose body (2017)
# ----------------------------------------------------------------------------
# model0: VGG19 front + CPM convs. ('name', cin, cout, k) or ('pool',)
OPENPOSE_MODEL0 = [
    ('conv1_1', 3, 64, 3), ('conv1_2', 64, 64, 3), ('pool',),
    ('conv2_1', 64, 128, 3), ('conv2_2', 128, 128, 3), ('pool',),
    ('conv3_1', 128, 256, 3), ('conv3_2', 256, 256, 3), ('conv3_3', 256, 256, 3),
    ('conv3_4', 256, 256, 3), ('pool',),
    ('conv4_1', 256, 512, 3), ('conv4_2', 512, 512, 3),
    ('conv4_3_CPM', 512, 256, 3), ('conv4_4_CPM', 256, 128, 3),
]
OPENPOSE_PAF_CH = 38
OPENPOSE_HM_CH = 19


def openpose_stage_layers(t, branch):
    """Layers of stage t (1..6), branch 1 (PAF, 38ch) or 2 (heatmap, 19ch).

    Returns list of (name, cin, cout, k, relu).  The ReLU flags reproduce the
    reference's `no_relu_layers` list *including its typo* (model.py:32-39):
    'Mconv7_stage6_L1' is listed twice and 'Mconv7_stage6_L2' is missing, so the
    final heat-map conv IS followed by a ReLU.
    """
    cout_final = OPENPOSE_PAF_CH if branch == 1 else OPENPOSE_HM_CH
    L = 'L%d' % branch
    if t == 1:
        return [
            ('conv5_1_CPM_' + L, 128, 128, 3, True),
            ('conv5_2_CPM_' + L, 128, 128, 3, True),
            ('conv5_3_CPM_' + L, 128, 128, 3, True),
            ('conv5_4_CPM_' + L, 128, 512, 1, True),
            ('conv5_5_CPM_' + L, 512, cout_final, 1, False),
        ]
    last_relu = (t == 6 and branch == 2)
    return [
        ('Mconv1_stage%d_%s' % (t, L), 185, 128, 7, True),
        ('Mconv2_stage%d_%s' % (t, L), 128, 128, 7, True),
        ('Mconv3_stage%d_%s' % (t, L), 128, 128, 7, True),
        ('Mconv4_stage%d_%s' % (t, L), 128, 128, 7, True),
        ('Mconv5_stage%d_%s' % (t, L), 128, 128, 7, True),
        ('Mconv6_stage%d_%s' % (t, L), 128, 128, 1, True),
        ('Mconv7_stage%d_%s' % (t, L), 128, cout_final, 1, last_relu),
    ]


def openpose_param_specs():
    specs = []
    for item in OPENPOSE_MODEL0:
        if item[0] == 'pool':
            continue
        name, cin, cout, k = item
        specs.append(('model0.%s.weight' % name, (cout, cin, k, k), 'conv'))
        specs.append(('model0.%s.bias' % name, (cout,), 'bias'))
    for t in range(1, 7):
        for b in (1, 2):
            for name, cin, cout, k, relu in openpose_stage_layers(t, b):
                kind = 'conv_out' if cout in (OPENPOSE_PAF_CH, OPENPOSE_HM_CH) else 'conv'
                specs.append(('model%d_%d.%s.weight' % (t, b, name), (cout, cin, k, k), kind))
                specs.append(('model%d_%d.%s.bias' % (t, b, name), (cout,), 'bias'))
    return specs


# Limb tables (pose/openpose/wrapper.py:12-24).  `MAP_IDX[l]` = the two
# channels of the 57-channel [heatmap19 | paf38] Caffe layout holding the
# (x, y) PAF components of limb l; PAF-tensor channel = value - 19.
MAP_IDX = [
    [31, 32], [39, 40], [33, 34], [35, 36], [41, 42], [43, 44],
    [19, 20], [21, 22], [23, 24], [25, 26], [27, 28], [29, 30],
    [47, 48], [49, 50], [53, 54], [51, 52], [55, 56], [37, 38],
    [45, 46],
]
LIMBSEQ = [
    [2, 3], [2, 6], [3, 4], [4, 5], [6, 7], [7, 8], [2, 9],
    [9, 10], [10, 11], [2, 12], [12, 13], [13, 14], [2, 1],
    [1, 15], [15, 17], [1, 16], [16, 18], [3, 17], [6, 18],
]
