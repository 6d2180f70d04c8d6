"""Second, independent restatement of the reference graph on PyTorch CPU ops.

Test infrastructure only.  It deliberately shares no code with
``oracle/ju_oracle.py``: convolutions are ``F.conv2d`` / ``F.conv_transpose2d``
on NCHW tensors with BatchNorm applied by ``F.batch_norm``, the warp is
``F.grid_sample(bilinear, border, align_corners=False)`` with the grid
normalisation the reference's own ONNX surgery uses
(scripts/inference/onnx/replace_dense_warp.py:89-112), pooling is
``F.max_pool2d``, and depth/space shuffles are explicit ``view/permute`` in TF's
DCR order (scripts/training/keras_layers.py:129, 175).  Agreement between the
two restatements (<= 1e-5 on ``output_raw`` in float64) is the stand-in for the
unavailable TensorFlow oracle.
"""

import numpy as np
import torch
import torch.nn.functional as F

DT = torch.float64


def _t(a):
    return torch.from_numpy(np.asarray(a, dtype=np.float64))


def _conv(x, w, name, bias=False):
    k = _t(w[name + "/kernel"]).permute(3, 2, 0, 1)  # [kh,kw,cin,cout]->OIHW
    b = _t(w[name + "/bias"]) if bias else None
    return F.conv2d(x, k, b, padding=k.shape[-1] // 2)


def _bn(x, w, name, eps):
    return F.batch_norm(x, _t(w[name + "/moving_mean"]),
                        _t(w[name + "/moving_variance"]),
                        _t(w[name + "/gamma"]), _t(w[name + "/beta"]),
                        training=False, eps=eps)


def _act(cfg, part):
    """keras ReLU / LeakyReLU(negative_slope) (reference models.py:24-27) as torch ops."""
    name = getattr(cfg, part + "_activation", "relu")
    if name == "relu":
        return F.relu
    slope = getattr(cfg, part + "_negative_slope")
    return lambda x: F.leaky_relu(x, negative_slope=slope)


def _cba(x, w, c, b, eps, act=F.relu):
    return act(_bn(_conv(x, w, c), w, b, eps))


def _res(x, w, n, eps, act=F.relu):
    y = _cba(x, w, n + "/conv_1", n + "/bn_1", eps, act)
    y = _bn(_conv(y, w, n + "/conv_2"), w, n + "/bn_2", eps)
    return act(y + x)


# ---- 8-bit tower (the BUILD's scheme, joshupscale_amd/csrc/fp8.h), restated a second time ----
# e4m3 rounding is PyTorch's own float8_e4m3fn conversion (round to nearest even), not the
# oracle's frexp arithmetic; scales are derived with torch ops.
def _e4m3(x):
    """Nearest e4m3 value, ties to even, of float64 data through PyTorch's float8_e4m3fn cast.
    The cast takes float32, and float64 -> float32 -> e4m3 with two nearest roundings goes
    wrong for values within 2^-24 of an e4m3 tie (a few per 10^7: enough to flip a handful of
    activations per full-size tensor), so the first step rounds TO ODD (truncate, set the last
    bit when inexact), after which the second rounding is the correct one."""
    x = x.clamp(-448.0, 448.0)
    f = x.to(torch.float32)
    back = f.to(DT)
    f = torch.where(back.abs() > x.abs(), torch.nextafter(f, torch.zeros_like(f)), f)  # toward zero
    bits = f.view(torch.int32)
    f = torch.where(back != x, bits | 1, bits).view(torch.float32)
    return f.to(torch.float8_e4m3fn).to(DT)


def _fold32(w, conv, bn, eps):
    """BN folded into the kernel as the engine's loader stores it (csrc/model.cpp): float64
    arithmetic on the float32 variables, rounded once to float32.  OIHW kernel, bias."""
    k = torch.from_numpy(np.asarray(w[conv + "/kernel"], np.float32)).to(DT).permute(3, 2, 0, 1)
    g = torch.from_numpy(np.asarray(w[bn + "/gamma"], np.float32)).to(DT)
    var = torch.from_numpy(np.asarray(w[bn + "/moving_variance"], np.float32)).to(DT)
    scale = g / torch.sqrt(var + float(np.float32(eps)))
    bias = torch.from_numpy(np.asarray(w[bn + "/beta"], np.float32)).to(DT) - \
        torch.from_numpy(np.asarray(w[bn + "/moving_mean"], np.float32)).to(DT) * scale
    return (k * scale.view(-1, 1, 1, 1)).to(torch.float32), bias.to(torch.float32)


def _q_weights(k):
    """Per output channel 2^ew with max|w| 2^ew in (224, 448], then e4m3, scaled back."""
    amax = k.abs().amax(dim=(1, 2, 3)).to(DT)
    ew = torch.where(amax > 0, torch.floor(torch.log2(448.0 / amax.clamp(min=1e-300))), torch.zeros_like(amax))
    ew = ew.clamp(-32, 32).view(-1, 1, 1, 1)
    return _e4m3(k.to(DT) * torch.exp2(ew)) * torch.exp2(-ew)


def _q_act(x, e):
    return _e4m3(x * 2.0 ** e) * 2.0 ** -e      # (_e4m3 clamps to +-448)


def _act_exponent(amax):
    if not (amax > 0 and np.isfinite(amax)):
        return 0
    return int(min(max(np.floor(np.log2(224.0 / amax)), -16), 16))


def _res_fp8(x, w, n, eps, ex, et, act=F.relu):
    k1, b1 = _fold32(w, n + "/conv_1", n + "/bn_1", eps)
    k2, b2 = _fold32(w, n + "/conv_2", n + "/bn_2", eps)
    t = act(F.conv2d(_q_act(x, ex), _q_weights(k1), b1.to(DT), padding=1))
    y = F.conv2d(_q_act(t, et), _q_weights(k2), b2.to(DT), padding=1)
    return act(y + x)


def _up_tf1(x, s):
    # tf.compat.v1 resize_bilinear(align_corners=False, half_pixel_centers=False)
    n, c, h, w = x.shape
    ys = torch.arange(h * s, dtype=DT) / s
    xs = torch.arange(w * s, dtype=DT) / s
    y0 = ys.floor().long()
    x0 = xs.floor().long()
    y1 = (y0 + 1).clamp(max=h - 1)
    x1 = (x0 + 1).clamp(max=w - 1)
    fy = (ys - y0).view(1, 1, -1, 1)
    fx = (xs - x0).view(1, 1, 1, -1)
    r0 = x[:, :, y0]
    r1 = x[:, :, y1]
    top = r0[..., x0] + (r0[..., x1] - r0[..., x0]) * fx
    bot = r1[..., x0] + (r1[..., x1] - r1[..., x0]) * fx
    return top + (bot - top) * fy


def _d2s_dcr(x, bs):
    n, c, h, w = x.shape
    c2 = c // (bs * bs)
    x = x.view(n, bs, bs, c2, h, w).permute(0, 3, 4, 1, 5, 2)
    return x.reshape(n, c2, h * bs, w * bs)


def _s2d(x, bs):
    n, c, h, w = x.shape
    x = x.view(n, c, h // bs, bs, w // bs, bs).permute(0, 3, 5, 1, 2, 4)
    return x.reshape(n, bs * bs * c, h // bs, w // bs)


def _warp(img, flow):
    # img [1,3,H,W], flow [1,2,H,W] with channel 0 = dy, 1 = dx
    _, _, h, w = img.shape
    gy, gx = torch.meshgrid(torch.arange(h, dtype=DT), torch.arange(w, dtype=DT),
                            indexing="ij")
    qx = gx - flow[0, 1]
    qy = gy - flow[0, 0]
    # replace_dense_warp.py: grid = q / (size/2) + (-1 + 1/size)
    grid = torch.stack([qx / (w * 0.5) + (-1 + 1.0 / w),
                        qy / (h * 0.5) + (-1 + 1.0 / h)], dim=-1)[None]
    return F.grid_sample(img, grid, mode="bilinear", padding_mode="border",
                         align_corners=False)


class TorchSession:
    def __init__(self, weights, cfg):
        self.w = weights
        self.cfg = cfg
        h, w = cfg.frame_height, cfg.frame_width
        self.pre_gen = torch.zeros(1, 3, 4 * h, 4 * w, dtype=DT)
        self.last = [torch.zeros(1, 3, cfg.padded_height, cfg.padded_width, dtype=DT)
                     for _ in range(cfg.num_flow_inputs - 1)]
        self.output_raw = None

    def _flow(self, frames):
        w, cfg, eps = self.w, self.cfg, self.cfg.bn_eps
        act = _act(cfg, "flow")
        x = torch.cat(frames, dim=1)
        if cfg.flow_arch == "autoencoder":
            f = cfg.flow_filters
            nb = len(f) // 2
            for i in range(2 * nb):
                n = f"flow/block_{i + 1}"
                x = _cba(x, w, n + "/conv_1", n + "/bn_1", eps, act)
                x = _cba(x, w, n + "/conv_2", n + "/bn_2", eps, act)
                x = F.max_pool2d(x, 2) if i < nb else _up_tf1(x, 2)
            if len(f) % 2:
                x = _cba(x, w, "flow/conv_1", "flow/bn_1", eps, act)
        else:
            x = _cba(x, w, "flow/conv_1", "flow/bn_1", eps, act)
            for i in range(cfg.flow_res_blocks):
                x = _res(x, w, f"flow/block_{i + 1}", eps, act)
        x = _conv(x, w, "flow/conv_2", bias=True)
        return _d2s_dcr(x, 4)

    def run(self, frame_bgrx):
        w, cfg, eps = self.w, self.cfg, self.cfg.bn_eps
        h, wd = cfg.frame_height, cfg.frame_width
        cur = torch.from_numpy(frame_bgrx[..., :3].astype(np.float64))
        cur = (cur / 255 - 0.5).permute(2, 0, 1)[None]
        ph, pw = cfg.padded_height, cfg.padded_width
        pt, pl = (ph - h) // 2, (pw - wd) // 2
        cur_pad = F.pad(cur, (pl, pw - wd - pl, pt, ph - h - pt))
        flow = self._flow([cur_pad] + self.last)
        flow = flow[:, :, pt * 4:pt * 4 + 4 * h, pl * 4:pl * 4 + 4 * wd]
        pre_warp = _warp(self.pre_gen, flow)
        x = torch.cat([cur, _s2d(pre_warp, 4)], dim=1)
        act = _act(cfg, "gen")
        x = _cba(x, w, "generator/conv_1", "generator/bn_1", eps, act)
        fp8 = getattr(cfg, "fp8_tower", False)
        if fp8:
            amax = w.get("generator/fp8_amax")
            exps = [_act_exponent(7.0 if amax is None else float(np.float32(amax[j])))
                    for j in range(2 * cfg.gen_blocks)]
        for i in range(cfg.gen_blocks):
            if fp8:
                x = _res_fp8(x, w, f"generator/block_{i + 1}", eps, exps[2 * i], exps[2 * i + 1], act)
            else:
                x = _res(x, w, f"generator/block_{i + 1}", eps, act)
        k1 = _t(w["generator/conv_trans_1/kernel"]).permute(3, 2, 0, 1)
        x = act(_bn(F.conv_transpose2d(x, k1, stride=2), w, "generator/bn_2", eps))
        k2 = _t(w["generator/conv_trans_2/kernel"]).permute(3, 2, 0, 1)
        x = F.conv_transpose2d(x, k2, _t(w["generator/conv_trans_2/bias"]), stride=2)
        x = (torch.tanh(x) + _up_tf1(cur, 4)).clamp(-0.5, 0.5)
        self.output_raw = x
        self.pre_gen = x
        self.last = ([cur_pad] + self.last[:-1])[:cfg.num_flow_inputs - 1]  # (one flow input: no history)
        out = ((x + 0.5) * 255).to(torch.uint8)[0].permute(1, 2, 0).numpy()
        res = np.zeros(out.shape[:2] + (4,), np.uint8)
        res[..., :3] = out
        return res
