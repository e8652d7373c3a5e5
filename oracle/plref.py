"""ctypes wrapper over the CPU oracle (oracle/libplref.so) and, when present, the reference's
own scalar oracle (oracle/_ref/libref_naive.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product path (paddle-lite_amd/) never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ACT_NONE, ACT_RELU, ACT_RELU6, ACT_LEAKY = 0, 1, 2, 4


class ConvShape(C.Structure):
    _fields_ = [("n", C.c_int), ("cin", C.c_int), ("h", C.c_int), ("w", C.c_int),
                ("cout", C.c_int), ("kh", C.c_int), ("kw", C.c_int),
                ("pad", C.c_int * 4), ("stride", C.c_int * 2), ("dil", C.c_int * 2),
                ("groups", C.c_int)]


def build(force=False):
    """Compile oracle/libplref.so (and oracle/_ref when /root/reference is present)."""
    so = os.path.join(_HERE, "libplref.so")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(os.path.join(_HERE, "plref.c")):
        subprocess.check_call(["make", "-s", "-C", _HERE, "libplref.so"])
    if os.path.isdir("/root/reference/lite") and (force or not os.path.exists(os.path.join(_HERE, "_ref", "libref_naive.so"))):
        subprocess.check_call(["make", "-s", "-C", _HERE, "ref"])


_lib = None
_ref = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(os.path.join(_HERE, "libplref.so"))
        _lib.plref_fold_scales.restype = C.c_float
        _lib.plref_epilogue_f32.restype = C.c_float
        _lib.plref_epilogue_i8.restype = C.c_int8
        _lib.plref_round_sat_i8.restype = C.c_int8
    return _lib


def ref_lib():
    """The reference's naive_math_impl.h compiled in place; None when not built (GPU box w/o _ref)."""
    global _ref
    if _ref is None:
        p = os.path.join(_HERE, "_ref", "libref_naive.so")
        if not os.path.exists(p):
            return None
        _ref = C.CDLL(p)
    return _ref


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def shape(n, cin, h, w, cout, kh, kw, pad, stride, dil, groups):
    s = ConvShape()
    s.n, s.cin, s.h, s.w, s.cout, s.kh, s.kw, s.groups = n, cin, h, w, cout, kh, kw, groups
    if len(pad) == 2:
        pad = (pad[0], pad[0], pad[1], pad[1])
    s.pad[:] = list(pad)
    s.stride[:] = list(stride)
    s.dil[:] = list(dil)
    return s


def out_dims(s):
    oh, ow = C.c_int(), C.c_int()
    lib().plref_conv_out_dims(C.byref(s), C.byref(oh), C.byref(ow))
    return oh.value, ow.value


def conv2d_acc(s, x, w, via_gemm=False):
    x = np.ascontiguousarray(x, np.int8)
    w = np.ascontiguousarray(w, np.int8)
    oh, ow = out_dims(s)
    acc = np.empty((s.n, s.cout, oh, ow), np.int32)
    if via_gemm:
        k = (s.cin // s.groups) * s.kh * s.kw
        ws = np.empty(max(1, k * oh * ow), np.int8)
        lib().plref_conv2d_i8_acc_im2col_gemm(C.byref(s), _p(x, C.c_int8), _p(w, C.c_int8),
                                              _p(acc, C.c_int32), _p(ws, C.c_int8))
    else:
        lib().plref_conv2d_i8_acc(C.byref(s), _p(x, C.c_int8), _p(w, C.c_int8), _p(acc, C.c_int32))
    return acc


def fold_scales(int8_out, in_scale, w_scale, out_scale, bias, cout, act, alpha):
    w_scale = np.ascontiguousarray(w_scale, np.float32)
    so = np.empty(cout, np.float32)
    bo = np.empty(cout, np.float32)
    bp = None
    if bias is not None:
        bias = np.ascontiguousarray(bias, np.float32)
        bp = _p(bias, C.c_float)
    a = lib().plref_fold_scales(int(int8_out), C.c_float(in_scale), _p(w_scale, C.c_float),
                                int(w_scale.size), C.c_float(out_scale), bp, cout, act,
                                C.c_float(alpha), _p(so, C.c_float), _p(bo, C.c_float))
    return so, bo, float(np.float32(a))


def epilogue(acc, scale, bias, act, alpha, int8_out):
    acc = np.ascontiguousarray(acc, np.int32)
    n, cout = acc.shape[0], acc.shape[1]
    spatial = int(np.prod(acc.shape[2:])) if acc.ndim > 2 else 1
    scale = np.ascontiguousarray(scale, np.float32)
    bias = np.ascontiguousarray(bias, np.float32)
    if int8_out:
        y = np.empty(acc.shape, np.int8)
        lib().plref_apply_epilogue_i8(_p(acc, C.c_int32), n, cout, spatial, _p(scale, C.c_float),
                                      _p(bias, C.c_float), act, C.c_float(alpha), _p(y, C.c_int8))
    else:
        y = np.empty(acc.shape, np.float32)
        lib().plref_apply_epilogue_f32(_p(acc, C.c_int32), n, cout, spatial, _p(scale, C.c_float),
                                       _p(bias, C.c_float), act, C.c_float(alpha), _p(y, C.c_float))
    return y


def conv2d(s, x, w, bias, in_scale, w_scale, out_scale, act, alpha, int8_out, via_gemm=False):
    """Full reference-semantics int8 conv: accumulator + folded-scale epilogue."""
    acc = conv2d_acc(s, x, w, via_gemm)
    sc, bi, al = fold_scales(int8_out, in_scale, w_scale, out_scale, bias, s.cout, act, alpha)
    return epilogue(acc, sc, bi, act, al, int8_out), acc


def im2col(x, kh, kw, pad, stride, dil):
    x = np.ascontiguousarray(x, np.int8)
    cin, h, w = x.shape
    s = shape(1, cin, h, w, 1, kh, kw, pad, stride, dil, 1)
    oh, ow = out_dims(s)
    col = np.empty((cin * kh * kw, oh * ow), np.int8)
    lib().plref_im2col_i8(_p(x, C.c_int8), cin, h, w, kh, kw, s.pad, s.stride, s.dil, oh, ow,
                          _p(col, C.c_int8))
    return col


def gemm_acc(a, b):
    a = np.ascontiguousarray(a, np.int8)
    b = np.ascontiguousarray(b, np.int8)
    m, k = a.shape
    n = b.shape[1]
    c = np.empty((m, n), np.int32)
    lib().plref_gemm_i8_acc(m, n, k, _p(a, C.c_int8), _p(b, C.c_int8), _p(c, C.c_int32))
    return c


def fc_route(m, n_weight_scale):
    """The reference's dispatch (fc_compute.cc:66-71): gemm_s8 + fill_bias_fc (two roundings) iff m > 1 and a single
    weight scale; gemv_int8 per row (one fused multiply-add) otherwise."""
    return 1 if (m > 1 and n_weight_scale == 1) else 0


def fc(x, w, bias, scale, relu, int8_out, route=0):
    """x [m,k] int8, w [k,n] int8, scale [n] (already folded), bias [n] or None.  route: see fc_route()."""
    x = np.ascontiguousarray(x, np.int8)
    w = np.ascontiguousarray(w, np.int8)
    m, k = x.shape
    n = w.shape[1]
    acc = np.empty((m, n), np.int32)
    lib().plref_fc_i8_acc(m, n, k, _p(x, C.c_int8), _p(w, C.c_int8), _p(acc, C.c_int32))
    scale = np.ascontiguousarray(scale, np.float32)
    bp = None
    if bias is not None:
        bias = np.ascontiguousarray(bias, np.float32)
        bp = _p(bias, C.c_float)
    if int8_out:
        y = np.empty((m, n), np.int8)
        lib().plref_fc_epilogue_i8(_p(acc, C.c_int32), m, n, _p(scale, C.c_float), bp, int(relu), _p(y, C.c_int8))
    else:
        y = np.empty((m, n), np.float32)
        f = lib().plref_fc_epilogue_f32_two_roundings if route == 1 else lib().plref_fc_epilogue_f32
        f(_p(acc, C.c_int32), m, n, _p(scale, C.c_float), bp, int(relu), _p(y, C.c_float))
    return y, acc


def calib_f32_to_i8(x, scale):
    x = np.ascontiguousarray(x, np.float32)
    y = np.empty(x.shape, np.int8)
    lib().plref_calib_f32_to_i8(_p(x, C.c_float), _p(y, C.c_int8), C.c_float(scale), C.c_int64(x.size))
    return y


def calib_i8_to_f32(x, scale):
    x = np.ascontiguousarray(x, np.int8)
    y = np.empty(x.shape, np.float32)
    lib().plref_calib_i8_to_f32(_p(x, C.c_int8), _p(y, C.c_float), C.c_float(scale), C.c_int64(x.size))
    return y


def global_avg_pool(x):
    x = np.ascontiguousarray(x, np.float32)
    n, c = x.shape[:2]
    y = np.empty((n, c, 1, 1), np.float32)
    lib().plref_global_avg_pool_f32(_p(x, C.c_float), n * c, int(np.prod(x.shape[2:])), _p(y, C.c_float))
    return y


def pool2d(x, pooling_type, ksize, strides, pads, exclusive=True, ceil_mode=False):
    """x [n,c,h,w] fp32; pads {top, bottom, left, right}."""
    x = np.ascontiguousarray(x, np.float32)
    n, c, h, w = x.shape
    oh = lib().plref_pool_out_size(h, ksize[0], pads[0], pads[1], strides[0], int(ceil_mode))
    ow = lib().plref_pool_out_size(w, ksize[1], pads[2], pads[3], strides[1], int(ceil_mode))
    y = np.empty((n, c, oh, ow), np.float32)
    pad = (C.c_int * 4)(*pads)
    lib().plref_pool2d_f32(_p(x, C.c_float), n * c, h, w, oh, ow, ksize[0], ksize[1], strides[0], strides[1], pad,
                           int(pooling_type == "max"), int(exclusive), _p(y, C.c_float))
    return y


def elementwise_add(x, y, relu=False):
    x = np.ascontiguousarray(x, np.float32)
    y = np.ascontiguousarray(y, np.float32)
    assert x.shape == y.shape
    o = np.empty(x.shape, np.float32)
    lib().plref_elementwise_add_f32(_p(x, C.c_float), _p(y, C.c_float), _p(o, C.c_float), C.c_int64(x.size), int(relu))
    return o


def softmax(x):
    x = np.ascontiguousarray(x, np.float32)
    rows, cols = int(np.prod(x.shape[:-1])), x.shape[-1]
    y = np.empty(x.shape, np.float32)
    lib().plref_softmax_f32(_p(x, C.c_float), rows, cols, _p(y, C.c_float))
    return y


# ---- the reference itself (oracle/_ref), for pinning the restatement and minting goldens ----
def ref_conv_acc(s, x, w):
    """conv_basic<int8_t,int> from the reference header (naive_math_impl.h:351-453), raw accumulator."""
    r = ref_lib()
    assert r is not None, "oracle/_ref/libref_naive.so not built"
    x = np.ascontiguousarray(x, np.int8)
    w = np.ascontiguousarray(w, np.int8)
    oh, ow = out_dims(s)
    out = np.zeros((s.n, s.cout, oh, ow), np.int32)
    r.ref_conv_basic_i8(_p(x, C.c_int8), _p(out, C.c_int32), s.n, s.cout, oh, ow, s.cin, s.h, s.w,
                        _p(w, C.c_int8), None, s.groups, s.kw, s.kh, s.stride[1], s.stride[0],
                        s.dil[1], s.dil[0], s.pad[2], s.pad[0], 0, 0)
    return out


def ref_gemm_acc(a, b):
    r = ref_lib()
    assert r is not None
    a = np.ascontiguousarray(a, np.int8)
    b = np.ascontiguousarray(b, np.int8)
    m, k = a.shape
    n = b.shape[1]
    c = np.zeros((m, n), np.int32)
    r.ref_basic_gemm_i8(0, 0, m, n, k, _p(a, C.c_int8), k, _p(b, C.c_int8), n, _p(c, C.c_int32), n, None, 0, 0)
    return c


def ref_conv_f32(s, x, w, bias, act, six, alpha):
    """conv_basic<float,float>: the float baseline of conv_int8_compute_test.cc:298-321."""
    r = ref_lib()
    assert r is not None
    x = np.ascontiguousarray(x, np.float32)
    w = np.ascontiguousarray(w, np.float32)
    oh, ow = out_dims(s)
    out = np.zeros((s.n, s.cout, oh, ow), np.float32)
    bp = None
    if bias is not None:
        bias = np.ascontiguousarray(bias, np.float32)
        bp = _p(bias, C.c_float)
    r.ref_conv_basic_f32(_p(x, C.c_float), _p(out, C.c_float), s.n, s.cout, oh, ow, s.cin, s.h, s.w,
                         _p(w, C.c_float), bp, s.groups, s.kw, s.kh, s.stride[1], s.stride[0],
                         s.dil[1], s.dil[0], s.pad[2], s.pad[0], int(bias is not None), act,
                         C.c_float(six), C.c_float(alpha))
    return out
