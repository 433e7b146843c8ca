"""Oracle for the `locatt_ops` local-window attention ops (TEST INFRASTRUCTURE).

Three interchangeable backends with the reference's Python signature
(`localAttention.h:11-40`; NCHW float32, weights (B,H,W,kH*kW)):

  * `CLocatt('port')`      - oracle/locatt_c.c, our C restatement (double accumulation)
  * `CLocatt('reference')` - oracle/_ref/liblocatt_ref.so, the reference's own
                             kernels.cuh compiled for the host (bit-for-bit pin)
  * `TorchLocatt`          - vectorised shifted-slice form in torch (fast, any
                             float dtype; used for the big shapes and as CPU baseline)
"""
import ctypes
import os

import torch
import torch.nn.functional as F

_HERE = os.path.dirname(os.path.abspath(__file__))
_fp = ctypes.POINTER(ctypes.c_float)


def _ptr(t):
    return ctypes.cast(t.data_ptr(), _fp)


class CLocatt:
    """ctypes view of the C restatement ('port') or the compiled reference kernels ('reference')."""

    def __init__(self, kind='port'):
        if kind == 'port':
            path, prefix = os.path.join(_HERE, 'liboracle_locatt.so'), 'oracle_'
        elif kind == 'reference':
            path, prefix = os.path.join(_HERE, '_ref', 'liblocatt_ref.so'), 'ref_'
        else:
            raise ValueError(kind)
        if not os.path.exists(path):
            raise FileNotFoundError(f'{path} not built: run `make -C oracle`')
        self.kind = kind
        self.lib = ctypes.CDLL(path)
        self.p = prefix

    @staticmethod
    def available(kind):
        name = 'liboracle_locatt.so' if kind == 'port' else os.path.join('_ref', 'liblocatt_ref.so')
        return os.path.exists(os.path.join(_HERE, name))

    def _call(self, name, a, b, B, C, H, W, kH, kW, out, extra=None):
        fn = getattr(self.lib, self.p + name)
        args = [_ptr(a), _ptr(b), B, C, H, W, kH, kW]
        if extra is not None:
            args.append(int(extra))
        args.append(_ptr(out))
        fn.restype = None
        fn(*args)
        return out

    @staticmethod
    def _c(t):
        assert t.dtype == torch.float32 and t.device.type == 'cpu'
        return t.contiguous()

    def similar_forward(self, x_ori, x_loc, kH, kW):
        x_ori, x_loc = self._c(x_ori), self._c(x_loc)
        B, C, H, W = x_ori.shape
        out = torch.empty(B, H, W, kH * kW)
        return self._call('similar_forward', x_ori, x_loc, B, C, H, W, kH, kW, out)

    def similar_backward(self, x, grad_out, kH, kW, is_ori):
        x, grad_out = self._c(x), self._c(grad_out)
        B, C, H, W = x.shape
        out = torch.empty(B, C, H, W)
        return self._call('similar_backward', x, grad_out, B, C, H, W, kH, kW, out, extra=bool(is_ori))

    def weighting_forward(self, x_ori, x_weight, kH, kW):
        x_ori, x_weight = self._c(x_ori), self._c(x_weight)
        B, C, H, W = x_ori.shape
        out = torch.empty(B, C, H, W)
        return self._call('weighting_forward', x_ori, x_weight, B, C, H, W, kH, kW, out)

    def weighting_backward_ori(self, x_weight, grad_out, kH, kW):
        x_weight, grad_out = self._c(x_weight), self._c(grad_out)
        B, C, H, W = grad_out.shape
        out = torch.empty(B, C, H, W)
        return self._call('weighting_backward_ori', x_weight, grad_out, B, C, H, W, kH, kW, out)

    def weighting_backward_weight(self, x_ori, grad_out, kH, kW):
        x_ori, grad_out = self._c(x_ori), self._c(grad_out)
        B, C, H, W = x_ori.shape
        out = torch.empty(B, H, W, kH * kW)
        return self._call('weighting_backward_weight', x_ori, grad_out, B, C, H, W, kH, kW, out)


class TorchLocatt:
    """Same five ops as shifted-slice loops over the kH*kW window offsets.

    Slot k <-> (dy,dx) = (k//kW - rH, k%kW - rW) (kernels.cuh:22-27); out-of-image
    slots give 0.  An F.unfold of (6,128,112,200) would be 5.6 GB, hence the loop."""

    @staticmethod
    def similar_forward(x_ori, x_loc, kH, kW):
        B, C, H, W = x_ori.shape
        rH, rW = kH // 2, kW // 2
        pad = F.pad(x_loc, (rW, rW, rH, rH))
        out = x_ori.new_empty(B, H, W, kH * kW)
        for k in range(kH * kW):
            dy, dx = k // kW, k % kW
            out[..., k] = (x_ori * pad[:, :, dy:dy + H, dx:dx + W]).sum(1)
        return out

    @staticmethod
    def weighting_forward(x_ori, x_weight, kH, kW):
        B, C, H, W = x_ori.shape
        rH, rW = kH // 2, kW // 2
        pad = F.pad(x_ori, (rW, rW, rH, rH))
        out = torch.zeros_like(x_ori)
        for k in range(kH * kW):
            dy, dx = k // kW, k % kW
            out += pad[:, :, dy:dy + H, dx:dx + W] * x_weight[..., k].unsqueeze(1)
        return out

    @staticmethod
    def similar_backward(x, grad_out, kH, kW, is_ori):
        if is_ori:      # grad wrt x_ori given x = x_loc: ck2c_ori(x_loc, grad)
            return TorchLocatt.weighting_forward(x, grad_out, kH, kW)
        return TorchLocatt._ck2c_loc(x, grad_out, kH, kW)

    @staticmethod
    def weighting_backward_ori(x_weight, grad_out, kH, kW):
        return TorchLocatt._ck2c_loc(grad_out, x_weight, kH, kW)

    @staticmethod
    def weighting_backward_weight(x_ori, grad_out, kH, kW):
        return TorchLocatt.similar_forward(grad_out, x_ori, kH, kW)

    @staticmethod
    def _ck2c_loc(x_ori, x_weight, kH, kW):
        """y[c,h,w] = sum_k x_ori[c,h-dy,w-dx] * weight[(h-dy,w-dx),k]  (kernels.cuh:99-118)."""
        B, C, H, W = x_ori.shape
        rH, rW = kH // 2, kW // 2
        out = torch.zeros(B, C, H + 2 * rH, W + 2 * rW, dtype=x_ori.dtype, device=x_ori.device)
        for k in range(kH * kW):
            dy, dx = k // kW, k % kW
            # source pixel (h',w') contributes to (h'+dy-rH, w'+dx-rW)
            out[:, :, dy:dy + H, dx:dx + W] += x_ori * x_weight[..., k].unsqueeze(1)
        return out[:, :, rH:rH + H, rW:rW + W].contiguous()


def local_attention(q, k, v, kH=9, kW=9, impl=TorchLocatt):
    """`similar -> softmax(./sqrt(C)) -> weighting` of LocalContextAttentionBlock.forward
    (encoder_utils.py:132-134)."""
    w = impl.similar_forward(q, k, kH, kW)
    w = torch.softmax(w / (k.shape[1] ** 0.5), dim=-1)
    return impl.weighting_forward(v, w, kH, kW)
