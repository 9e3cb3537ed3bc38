"""Oracle: CRAFT detector forward (torch CPU, fp32).  TEST INFRASTRUCTURE ONLY.

Follows the Keras graph of the reference, ``keras_ocr/detection.py``:
  compute_input            :34-42
  make_vgg_block           :87-103   (conv 3x3 same + BN eps=1e-5 + ReLU [+ maxpool 2x2 valid])
  build_vgg_backbone       :312-335  (s1/s2/s3 = ReLU outputs .12/.19/.29, s4 = BN output .38)
  build_keras_model        :353-413  (slice5, concats, upconv, UpsampleLike, conv_cls)
  upconv                   :65-84
  UpsampleLike             :290-303  (resize_bilinear half_pixel_centers == torch
                                      interpolate(align_corners=False), detection.py:605-619)
Weights use the PyTorch state-dict naming that load_torch_weights consumes (:428-468).
"""
import numpy as np
import torch
import torch.nn.functional as F

MEAN = np.array([0.485, 0.456, 0.406])
VARIANCE = np.array([0.229, 0.224, 0.225])


def compute_input(image):
    """detection.py:34-42 — float32 image, in-place subtract / divide by float64 constants."""
    image = np.asarray(image).astype("float32")
    image -= MEAN * 255
    image /= VARIANCE * 255
    return image


def _t(w, name):
    return torch.from_numpy(np.ascontiguousarray(w[name]))


def _conv(w, name, x, padding=0, dilation=1):
    return F.conv2d(x, _t(w, name + ".weight"), _t(w, name + ".bias"), padding=padding, dilation=dilation)


def _bn(w, name, x, eps=1e-5):
    return F.batch_norm(
        x, _t(w, name + ".running_mean"), _t(w, name + ".running_var"), _t(w, name + ".weight"),
        _t(w, name + ".bias"), training=False, eps=eps)


def _vgg_block(w, prefix, n, x, pooling, relu=True):
    x = _conv(w, f"{prefix}.{n}", x, padding=1)
    x = _bn(w, f"{prefix}.{n + 1}", x)
    pre = x
    if relu:
        x = F.relu(x)
    if pooling:
        x = F.max_pool2d(x, 2, 2)
    return x, pre


def _upconv(w, n, x):
    x = F.relu(_bn(w, f"upconv{n}.conv.1", _conv(w, f"upconv{n}.conv.0", x)))
    x = F.relu(_bn(w, f"upconv{n}.conv.4", _conv(w, f"upconv{n}.conv.3", x, padding=1)))
    return x


def _upsample_like(src, tgt):
    return F.interpolate(src, size=tgt.shape[2:], mode="bilinear", align_corners=False)


@torch.no_grad()
def craft_forward(w, x_nhwc, return_intermediates=False):
    """x_nhwc: (N,H,W,3) float32, already normalised.  Returns (N,H//2,W//2,2) float32."""
    x = torch.from_numpy(np.ascontiguousarray(x_nhwc, dtype=np.float32)).permute(0, 3, 1, 2)
    inter = {}
    p = "basenet.slice1"
    x, _ = _vgg_block(w, p, 0, x, False)
    inter["basenet.slice1.2"] = x
    x, _ = _vgg_block(w, p, 3, x, True)
    x, _ = _vgg_block(w, p, 7, x, False)
    # slice1.10 block: s1 is the ReLU output (.12); pooling (.13) follows
    x = F.relu(_bn(w, p + ".11", _conv(w, p + ".10", x, padding=1)))
    s1 = x
    x = F.max_pool2d(x, 2, 2)
    x, _ = _vgg_block(w, "basenet.slice2", 14, x, False)
    x, _ = _vgg_block(w, "basenet.slice2", 17, x, False)
    s2 = x
    x, _ = _vgg_block(w, "basenet.slice3", 20, x, True)
    x, _ = _vgg_block(w, "basenet.slice3", 24, x, False)
    x, _ = _vgg_block(w, "basenet.slice3", 27, x, False)
    s3 = x
    x, _ = _vgg_block(w, "basenet.slice4", 30, x, True)
    x, _ = _vgg_block(w, "basenet.slice4", 34, x, False)
    _, s4 = _vgg_block(w, "basenet.slice4", 37, x, False)  # BN output, no ReLU (:333)
    # slice5 (:365-378): maxpool 3x3/s1/same (padding ignored), dilated conv, 1x1 conv
    s5 = F.max_pool2d(s4, 3, 1, 1)
    s5 = _conv(w, "basenet.slice5.1", s5, padding=6, dilation=6)
    s5 = _conv(w, "basenet.slice5.2", s5)
    inter.update(s1=s1, s2=s2, s3=s3, s4=s4, s5=s5)
    y = torch.cat([s5, s4], 1)
    y = _upconv(w, 1, y)
    y = torch.cat([_upsample_like(y, s3), s3], 1)
    y = _upconv(w, 2, y)
    y = torch.cat([_upsample_like(y, s2), s2], 1)
    y = _upconv(w, 3, y)
    y = torch.cat([_upsample_like(y, s1), s1], 1)
    feat = _upconv(w, 4, y)
    inter["features"] = feat
    y = F.relu(_conv(w, "conv_cls.0", feat, padding=1))
    y = F.relu(_conv(w, "conv_cls.2", y, padding=1))
    y = F.relu(_conv(w, "conv_cls.4", y, padding=1))
    y = F.relu(_conv(w, "conv_cls.6", y))
    y = _conv(w, "conv_cls.8", y)  # linear output for the vgg backbone (:411-412)
    out = y.permute(0, 2, 3, 1).contiguous().numpy()
    if return_intermediates:
        return out, {k: v.permute(0, 2, 3, 1).contiguous().numpy() for k, v in inter.items()}
    return out


def detector_predict(w, images_u8):
    """Detector.detect's device half (detection.py:777-779): compute_input + model.predict."""
    x = np.stack([compute_input(im) for im in images_u8])
    return craft_forward(w, x)
