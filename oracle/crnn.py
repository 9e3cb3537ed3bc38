"""Oracle: CRNN recogniser forward + CTC greedy decode (torch CPU, fp32).
TEST INFRASTRUCTURE ONLY.

Follows ``keras_ocr/recognition.py`` with Keras/TensorFlow layer semantics (SURVEY.md
Appendix C).  TensorFlow cannot be installed here; status: the STN sampler is **pinned** to the reference's own
``_transform`` (executed through a numpy stand-in of the TF ops it uses, tests/golden/make_golden.py); the LSTM, the
conv / BatchNorm / pooling stack and the CTC rule are **cross-checked by independent implementations** (torch.nn.LSTM /
Conv2d / BatchNorm2d modules with Keras-ordered weights, itertools.groupby: tests/test_thirdparty_crosscheck_cpu.py).
Every step cites the line it restates:

  Permute((2,1,3)) + flip axis 2                         :215-216
  conv_1..conv_7 3x3 same ReLU; BN *after* ReLU (Keras default eps=1e-3) at 3/5/7;
  MaxPooling2D(2) valid after bn_3, bn_5                 :217-242
  STN localisation net (5x5 conv16, 5x5 conv32, Flatten, Dense64 ReLU, Dense6) :268-278
  _transform bilinear sampler (scale by W/H, clipped corners) :73-166
  Reshape (W/4, H/4*512)                                 :282-288
  fc_9 Dense ReLU                                        :290
  lstm_10 / lstm_10_back (go_backwards, NOT re-reversed) -> Add :292-305
  lstm_11 / lstm_11_back -> Concatenate                  :306-319
  fc_12 Dense softmax; drop first rnn_steps_to_discard=2 :321-328
  CTCDecoder: keras.backend.ctc_decode greedy, -1 padding :169-184
  string assembly                                        :527-536

Keras LSTM (tf.keras >= 2.0 defaults): z = x@W + h@U + b, gate order [i, f, c~, o],
recurrent_activation = sigmoid, activation = tanh, zero initial state.
"""
import string

import numpy as np
import torch
import torch.nn.functional as F

DEFAULT_ALPHABET = string.digits + string.ascii_lowercase  # recognition.py:25
BN_EPS = 1e-3


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))


def _conv(w, name, x, relu=True):
    k = _t(w[name + "/kernel"]).permute(3, 2, 0, 1)  # HWIO -> OIHW
    pad = (k.shape[2] // 2, k.shape[3] // 2)
    y = F.conv2d(x, k, _t(w[name + "/bias"]), padding=pad)
    return F.relu(y) if relu else y


def _bn(w, name, x):
    return F.batch_norm(x, _t(w[name + "/moving_mean"]), _t(w[name + "/moving_variance"]), _t(w[name + "/gamma"]),
                        _t(w[name + "/beta"]), training=False, eps=BN_EPS)


def stn_transform(x_nhwc, theta):
    """recognition._transform (:73-166).  x_nhwc: (M,H,W,C) torch; theta: (M,6)."""
    M, H, W, C = x_nhwc.shape
    theta = theta.reshape(M, 2, 3)
    xs = torch.linspace(-1.0, 1.0, W)
    ys = torch.linspace(-1.0, 1.0, H)
    yy, xx = torch.meshgrid(ys, xs, indexing="ij")  # tf.meshgrid(x, y): rows = y
    xt = xx.reshape(-1)
    yt = yy.reshape(-1)
    x_s = (theta[:, 0, 0:1] * xt[None] + theta[:, 0, 1:2] * yt[None]) + theta[:, 0, 2:3]
    y_s = (theta[:, 1, 0:1] * xt[None] + theta[:, 1, 1:2] * yt[None]) + theta[:, 1, 2:3]
    x = 0.5 * (x_s + 1.0) * float(W)
    y = 0.5 * (y_s + 1.0) * float(H)
    x0 = torch.floor(x).to(torch.int64)
    x1 = x0 + 1
    y0 = torch.floor(y).to(torch.int64)
    y1 = y0 + 1
    x0 = x0.clamp(0, W - 1)
    x1 = x1.clamp(0, W - 1)
    y0 = y0.clamp(0, H - 1)
    y1 = y1.clamp(0, H - 1)
    flat = x_nhwc.reshape(M, H * W, C)

    def gather(yi, xi):
        idx = (yi * W + xi)[..., None].expand(-1, -1, C)
        return torch.gather(flat, 1, idx)

    pa, pb, pc, pd = gather(y0, x0), gather(y1, x0), gather(y0, x1), gather(y1, x1)
    x0f, x1f, y0f, y1f = x0.float(), x1.float(), y0.float(), y1.float()
    a = ((x1f - x) * (y1f - y))[..., None]
    b = ((x1f - x) * (y - y0f))[..., None]
    c = ((x - x0f) * (y1f - y))[..., None]
    d = ((x - x0f) * (y - y0f))[..., None]
    out = ((a * pa + b * pb) + c * pc) + d * pd
    return out.reshape(M, H, W, C)


def _lstm(w, name, x, go_backwards):
    """Keras LSTM, return_sequences=True.  x: (M,T,in).  Outputs in processing order."""
    W, U, b = _t(w[name + "/kernel"]), _t(w[name + "/recurrent_kernel"]), _t(w[name + "/bias"])
    M, T, _ = x.shape
    units = U.shape[0]
    h = torch.zeros(M, units)
    c = torch.zeros(M, units)
    outs = []
    steps = range(T - 1, -1, -1) if go_backwards else range(T)
    for t in steps:
        z = x[:, t] @ W + h @ U + b
        i, f, g, o = z.split(units, dim=1)
        c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(g)
        h = torch.sigmoid(o) * torch.tanh(c)
        outs.append(h)
    return torch.stack(outs, 1)


@torch.no_grad()
def crnn_forward(w, X, rnn_steps_to_discard=2, return_intermediates=False):
    """X: (M,31,200,1) float32 in [0,1].  Returns softmax probabilities (M,48,C+1)."""
    X = _t(X)
    if X.ndim == 3:
        X = X[..., None]
    inter = {}
    x = X.permute(0, 2, 1, 3)           # Permute((2,1,3)) -> (M,200,31,1)
    x = torch.flip(x, dims=[2])         # x[:, :, ::-1]
    x = x.permute(0, 3, 1, 2)           # NCHW with H=200, W=31
    x = _conv(w, "conv_1", x)
    x = _conv(w, "conv_2", x)
    x = _bn(w, "bn_3", _conv(w, "conv_3", x))
    x = F.max_pool2d(x, 2)
    x = _conv(w, "conv_4", x)
    x = _bn(w, "bn_5", _conv(w, "conv_5", x))
    x = F.max_pool2d(x, 2)
    x = _conv(w, "conv_6", x)
    x = _bn(w, "bn_7", _conv(w, "conv_7", x))
    inter["bn_7"] = x.permute(0, 2, 3, 1)
    # STN (recognition.py:243-281: only `if stn:`; a weight set without stn_* tensors = build_params["stn"] False)
    if "stn_conv_1/kernel" in w:
        loc = _conv(w, "stn_conv_1", x)
        loc = _conv(w, "stn_conv_2", loc)
        loc = loc.permute(0, 2, 3, 1).reshape(loc.shape[0], -1)  # Keras Flatten of NHWC
        loc = F.relu(loc @ _t(w["stn_dense_1/kernel"]) + _t(w["stn_dense_1/bias"]))
        theta = loc @ _t(w["stn_dense_2/kernel"]) + _t(w["stn_dense_2/bias"])
        inter["theta"] = theta
        x = stn_transform(x.permute(0, 2, 3, 1).contiguous(), theta)  # (M,50,7,512)
    else:
        x = x.permute(0, 2, 3, 1).contiguous()
    inter["stn"] = x
    M = x.shape[0]
    x = x.reshape(M, x.shape[1], -1)    # Reshape((W//4, (H//4)*512))
    x = F.relu(x @ _t(w["fc_9/kernel"]) + _t(w["fc_9/bias"]))
    inter["fc_9"] = x
    f1 = _lstm(w, "lstm_10", x, False)
    b1 = _lstm(w, "lstm_10_back", x, True)
    x = f1 + b1
    inter["rnn_1_add"] = x
    f2 = _lstm(w, "lstm_11", x, False)
    b2 = _lstm(w, "lstm_11_back", x, True)
    x = torch.cat([f2, b2], -1)
    inter["rnn_2"] = x
    logits = x @ _t(w["fc_12/kernel"]) + _t(w["fc_12/bias"])
    p = torch.softmax(logits, -1)[:, rnn_steps_to_discard:]
    inter["logits"] = logits[:, rnn_steps_to_discard:]
    if return_intermediates:
        return p.numpy(), {k: v.numpy() for k, v in inter.items()}
    return p.numpy()


def ctc_greedy_decode(probs):
    """keras.backend.ctc_decode(greedy=True) + the -1 re-padding of CTCDecoder (:169-184):
    per-step argmax (lowest index on ties), merge repeats, drop blank = last class."""
    probs = np.asarray(probs)
    M, T, C = probs.shape
    blank = C - 1
    out = np.full((M, T), -1, dtype=np.int64)
    best = np.log(probs + 1e-7).argmax(-1)
    for m in range(M):
        prev = -1
        k = 0
        for t in range(T):
            c = int(best[m, t])
            if c != prev and c != blank:
                out[m, k] = c
                k += 1
            prev = c
    return out


def decode_strings(labels, alphabet=DEFAULT_ALPHABET):
    """recognition.py:527-534."""
    blank = len(alphabet)
    return ["".join(alphabet[i] for i in row if i not in (blank, -1)) for row in labels]


def recognize_crops(w, X, alphabet=DEFAULT_ALPHABET):
    probs = crnn_forward(w, X)
    return decode_strings(ctc_greedy_decode(probs), alphabet), probs
