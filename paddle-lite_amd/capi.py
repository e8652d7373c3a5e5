"""ctypes binding of libplhip.so (include/plhip.h) — the only way Python reaches the device code.

There is deliberately NO fallback: if the HIP library is missing or a call fails this raises.
Used by tests/ (parity checks through the C ABI), bench.py and __graft_entry__.py.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PLHIP_LIB_PATH") or os.path.join(_HERE, "libplhip.so")  # override: diagnostic builds only (tools/slp_hazard_variants.py)

OUT_I32, OUT_F32, OUT_I8 = 0, 1, 2
OUT_F32_GAP = 3  # plhip_dwpw_fused_int8 only: the fp32 output averaged over each plane, [n][cout]
ACT_NONE, ACT_RELU, ACT_RELU6, ACT_LEAKY = 0, 1, 2, 4
_OUT_DTYPE = {OUT_I32: np.int32, OUT_F32: np.float32, OUT_I8: np.int8}

# every symbol include/plhip.h declares (tests check that the library exports all of them)
EXPORTS = [
    "plhip_device_count", "plhip_ctx_create", "plhip_ctx_create_on_stream", "plhip_ctx_destroy",
    "plhip_ctx_stream", "plhip_last_error", "plhip_malloc", "plhip_free", "plhip_memcpy_h2d",
    "plhip_memcpy_d2h", "plhip_memcpy_d2d", "plhip_memset", "plhip_stream_sync", "plhip_event_create",
    "plhip_event_record", "plhip_event_elapsed_ms", "plhip_event_destroy",
    "plhip_conv_packed_weight_bytes", "plhip_pack_conv_weights", "plhip_conv_workspace_bytes",
    "plhip_conv2d_int8", "plhip_conv2d_int8_fused", "plhip_conv_impl_name", "plhip_depthwise_conv_int8", "plhip_dwpw_fused_int8", "plhip_dwpw_fused_supported", "plhip_graph_begin", "plhip_graph_end", "plhip_graph_launch", "plhip_graph_destroy",
    "plhip_fc_packed_weight_bytes", "plhip_pack_fc_weights", "plhip_fc_int8",
    "plhip_calib_f32_to_i8", "plhip_calib_i8_to_f32", "plhip_global_avg_pool_f32", "plhip_softmax_f32",
    "plhip_pool2d_f32", "plhip_pool2d_max_i8", "plhip_elementwise_add_f32", "plhip_selftest",
    "plhip_debug_set", "plhip_debug_read_fw_stamps", "plhip_debug_read_fs_stamps", "plhip_debug_read_f7_stamps", "plhip_conv2d_calib_supported", "plhip_conv2d_calib_int8",
]


class ConvDesc(C.Structure):
    _fields_ = [("n", C.c_int), ("cin", C.c_int), ("h", C.c_int), ("w", C.c_int),
                ("cout", C.c_int), ("kh", C.c_int), ("kw", C.c_int),
                ("pad", C.c_int * 4), ("stride", C.c_int * 2), ("dil", C.c_int * 2),
                ("groups", C.c_int), ("act", C.c_int), ("act_alpha", C.c_float)]


class PoolDesc(C.Structure):
    _fields_ = [("planes", C.c_int), ("h", C.c_int), ("w", C.c_int), ("oh", C.c_int), ("ow", C.c_int),
                ("kh", C.c_int), ("kw", C.c_int), ("pad", C.c_int * 4), ("stride", C.c_int * 2),
                ("is_max", C.c_int), ("exclusive", C.c_int)]


def conv_desc(n, cin, h, w, cout, kh, kw, pad=(0, 0, 0, 0), stride=(1, 1), dil=(1, 1), groups=1,
              act=ACT_NONE, alpha=0.0):
    d = ConvDesc()
    d.n, d.cin, d.h, d.w, d.cout, d.kh, d.kw = n, cin, h, w, cout, kh, kw
    if len(pad) == 2:
        pad = (pad[0], pad[0], pad[1], pad[1])
    d.pad[:] = [int(p) for p in pad]
    d.stride[:] = [int(s) for s in stride]
    d.dil[:] = [int(s) for s in dil]
    d.groups, d.act, d.act_alpha = groups, act, alpha
    return d


def out_hw(d):
    keh = d.dil[0] * (d.kh - 1) + 1
    kew = d.dil[1] * (d.kw - 1) + 1
    return (int((d.h + d.pad[0] + d.pad[1] - keh) / d.stride[0]) + 1,
            int((d.w + d.pad[2] + d.pad[3] - kew) / d.stride[1]) + 1)


class PlhipError(RuntimeError):
    pass


_lib = None
# Diagnostic knobs this PROCESS set through plhip_debug_set (the library itself never reads the environment).  The binding
# forwards PLHIP_<KNOB>=<int> variables of the A/B scripts (tools/*.sh) explicitly and records them here;
# bench.py prints the dict in its JSON line, so a measurement taken with a knob says so.
KNOBS = ("STEM_MFMA", "CONV_PATCH", "CONV_PATCH_S2", "PATCH_DEBUG", "PATCH_DELAY", "STEM7", "DW_STAGE", "DW_STAGE_NP2", "DW_FASTV",
         "DW5_DIRECT", "DW_RS1", "DW_RS2", "GEMM_VARIANT", "GEMM_AREG", "GEMM_MA", "GEMM_DEBUG", "SUBSAMPLE_1X1", "GEMM_TR", "TR_DELAY",
         "TR_CFG", "GEMM_WIDE", "WIDE_NTT", "FC_MFMA", "IMPLICIT_GEMM", "FUSED_STREAM", "FUSED_SMALL")
KNOBS_SET = {}


def load():
    """dlopen libplhip.so and declare prototypes.  Raises if the library is absent (no CPU fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PlhipError("%s is missing: build it with __graft_entry__.build() (hipcc --offload-arch=gfx950)" % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    vp, i32, f32, sz = C.c_void_p, C.c_int, C.c_float, C.c_size_t
    L.plhip_device_count.restype = i32
    L.plhip_debug_set.argtypes = [C.c_char_p, i32]
    L.plhip_debug_set.restype = i32
    for k in KNOBS:
        v = os.environ.get("PLHIP_" + k)
        if v is not None:
            if L.plhip_debug_set(k.encode(), int(v)) != 0:
                raise PlhipError("unknown diagnostic knob %s" % k)
            KNOBS_SET[k] = int(v)
    if KNOBS_SET:
        import sys
        sys.stderr.write("capi: diagnostic knobs set from the environment: %s\n" % KNOBS_SET)
    L.plhip_ctx_create.argtypes = [i32, C.POINTER(vp)]
    L.plhip_ctx_create_on_stream.argtypes = [i32, vp, C.POINTER(vp)]
    L.plhip_ctx_destroy.argtypes = [vp]
    L.plhip_ctx_destroy.restype = None
    L.plhip_ctx_stream.argtypes = [vp]
    L.plhip_ctx_stream.restype = vp
    L.plhip_last_error.argtypes = [vp]
    L.plhip_last_error.restype = C.c_char_p
    L.plhip_malloc.argtypes = [vp, sz, C.POINTER(vp)]
    L.plhip_free.argtypes = [vp, vp]
    L.plhip_memcpy_h2d.argtypes = [vp, vp, vp, sz]
    L.plhip_memcpy_d2h.argtypes = [vp, vp, vp, sz]
    L.plhip_memcpy_d2d.argtypes = [vp, vp, vp, sz]
    L.plhip_memset.argtypes = [vp, vp, i32, sz]
    L.plhip_stream_sync.argtypes = [vp]
    L.plhip_event_create.argtypes = [vp, C.POINTER(vp)]
    L.plhip_event_record.argtypes = [vp, vp]
    L.plhip_event_elapsed_ms.argtypes = [vp, vp, vp, C.POINTER(f32)]
    L.plhip_event_destroy.argtypes = [vp, vp]
    L.plhip_conv_packed_weight_bytes.argtypes = [C.POINTER(ConvDesc)]
    L.plhip_conv_packed_weight_bytes.restype = sz
    L.plhip_pack_conv_weights.argtypes = [vp, C.POINTER(ConvDesc), vp, vp]
    L.plhip_conv_workspace_bytes.argtypes = [C.POINTER(ConvDesc)]
    L.plhip_conv_workspace_bytes.restype = sz
    L.plhip_conv2d_int8.argtypes = [vp, C.POINTER(ConvDesc), vp, vp, vp, vp, vp, i32, vp, sz]
    L.plhip_conv2d_int8_fused.argtypes = [vp, C.POINTER(ConvDesc), vp, vp, vp, vp, vp, vp, i32, vp, f32, vp, sz]
    L.plhip_conv2d_calib_supported.argtypes = [C.POINTER(ConvDesc)]
    L.plhip_conv2d_calib_int8.argtypes = [vp, C.POINTER(ConvDesc), vp, f32, vp, vp, vp, vp, i32]
    L.plhip_conv_impl_name.argtypes = [C.POINTER(ConvDesc)]
    L.plhip_conv_impl_name.restype = C.c_char_p
    L.plhip_depthwise_conv_int8.argtypes = [vp, C.POINTER(ConvDesc), vp, vp, vp, vp, vp, i32]
    L.plhip_dwpw_fused_int8.argtypes = [vp, C.POINTER(ConvDesc), vp, vp, vp, vp, i32, vp, vp, vp, i32, f32, vp, i32]
    L.plhip_dwpw_fused_supported.argtypes = [C.POINTER(ConvDesc), i32, i32]
    L.plhip_dwpw_fused_supported.restype = i32
    L.plhip_fc_packed_weight_bytes.argtypes = [i32, i32]
    L.plhip_fc_packed_weight_bytes.restype = sz
    L.plhip_pack_fc_weights.argtypes = [vp, i32, i32, vp, vp]
    L.plhip_fc_int8.argtypes = [vp, i32, i32, i32, vp, vp, vp, vp, i32, vp, i32]
    L.plhip_calib_f32_to_i8.argtypes = [vp, vp, vp, f32, C.c_int64]
    L.plhip_calib_i8_to_f32.argtypes = [vp, vp, vp, f32, C.c_int64]
    L.plhip_global_avg_pool_f32.argtypes = [vp, vp, i32, i32, vp]
    L.plhip_softmax_f32.argtypes = [vp, vp, i32, i32, vp]
    L.plhip_pool2d_f32.argtypes = [vp, C.POINTER(PoolDesc), vp, vp]
    L.plhip_pool2d_max_i8.argtypes = [vp, C.POINTER(PoolDesc), vp, vp]
    L.plhip_elementwise_add_f32.argtypes = [vp, vp, vp, vp, C.c_int64, i32]
    L.plhip_selftest.argtypes = [vp]
    _lib = L
    return L


class Context:
    """One plhip_ctx (device + stream).  Thin, explicit device-memory helpers for tests/bench."""

    def __init__(self, device=0, stream=None):
        self.L = load()
        h = C.c_void_p()
        if stream is None:
            st = self.L.plhip_ctx_create(device, C.byref(h))
        else:
            st = self.L.plhip_ctx_create_on_stream(device, C.c_void_p(stream), C.byref(h))
        if st != 0:
            raise PlhipError("plhip_ctx_create failed (%d): %s" % (st, self.L.plhip_last_error(None).decode()))
        self.h = h
        self._allocs = []

    def check(self, st, what=""):
        if st != 0:
            raise PlhipError("%s failed (%d): %s" % (what, st, self.L.plhip_last_error(self.h).decode()))

    def close(self):
        if self.h:
            for p in self._allocs:
                self.L.plhip_free(self.h, p)
            self._allocs = []
            self.L.plhip_ctx_destroy(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # ---- memory ----
    def malloc(self, nbytes):
        p = C.c_void_p()
        self.check(self.L.plhip_malloc(self.h, nbytes, C.byref(p)), "plhip_malloc")
        self._allocs.append(p)
        return p

    def free(self, p):
        self._allocs = [q for q in self._allocs if q.value != p.value]
        self.check(self.L.plhip_free(self.h, p), "plhip_free")

    def to_device(self, arr):
        arr = np.ascontiguousarray(arr)
        p = self.malloc(max(1, arr.nbytes))
        if arr.nbytes:
            self.check(self.L.plhip_memcpy_h2d(self.h, p, arr.ctypes.data_as(C.c_void_p), arr.nbytes), "h2d")
        return p

    def to_host(self, p, shape, dtype):
        out = np.empty(shape, dtype)
        if out.nbytes:
            self.check(self.L.plhip_memcpy_d2h(self.h, out.ctypes.data_as(C.c_void_p), p, out.nbytes), "d2h")
        return out

    def sync(self):
        self.check(self.L.plhip_stream_sync(self.h), "sync")

    def selftest(self):
        self.check(self.L.plhip_selftest(self.h), "plhip_selftest")

    # ---- whole-op helpers on host arrays (upload, run through the C ABI, download) ----
    def conv2d(self, d, x, w, scale, bias, out_kind, depthwise=False):
        oh, ow = out_hw(d)
        dx = self.to_device(np.ascontiguousarray(x, np.int8))
        dw = self.to_device(np.ascontiguousarray(w, np.int8))
        ds = self.to_device(np.ascontiguousarray(scale, np.float32)) if scale is not None else C.c_void_p()
        db = self.to_device(np.ascontiguousarray(bias, np.float32)) if bias is not None else C.c_void_p()
        esz = 1 if out_kind == OUT_I8 else 4
        dy = self.malloc(d.n * d.cout * oh * ow * esz)
        tmp = [dx, dw, dy]
        if depthwise:
            self.check(self.L.plhip_depthwise_conv_int8(self.h, C.byref(d), dx, dw, ds, db, dy, out_kind), "depthwise")
        else:
            dwp = self.malloc(self.L.plhip_conv_packed_weight_bytes(C.byref(d)))
            self.check(self.L.plhip_pack_conv_weights(self.h, C.byref(d), dw, dwp), "pack")
            wsb = self.L.plhip_conv_workspace_bytes(C.byref(d))
            dws = self.malloc(wsb) if wsb else C.c_void_p()
            self.check(self.L.plhip_conv2d_int8(self.h, C.byref(d), dx, dwp, ds, db, dy, out_kind, dws, wsb), "conv2d")
            tmp += [dwp] + ([dws] if wsb else [])
        y = self.to_host(dy, (d.n, d.cout, oh, ow), _OUT_DTYPE[out_kind])
        for p in tmp + ([ds] if scale is not None else []) + ([db] if bias is not None else []):
            self.free(p)
        return y

    def conv2d_calib(self, d, x_f32, calib_scale, w, scale, bias, out_kind):
        """plhip_conv2d_calib_int8 on host arrays: calib[fp32_to_int8](calib_scale) + conv2d in one launch."""
        oh, ow = out_hw(d)
        dx = self.to_device(np.ascontiguousarray(x_f32, np.float32))
        dw = self.to_device(np.ascontiguousarray(w, np.int8))
        ds = self.to_device(np.ascontiguousarray(scale, np.float32)) if scale is not None else C.c_void_p()
        db = self.to_device(np.ascontiguousarray(bias, np.float32)) if bias is not None else C.c_void_p()
        dy = self.malloc(d.n * d.cout * oh * ow * (1 if out_kind == OUT_I8 else 4))
        dwp = self.malloc(self.L.plhip_conv_packed_weight_bytes(C.byref(d)))
        self.check(self.L.plhip_pack_conv_weights(self.h, C.byref(d), dw, dwp), "pack")
        self.check(self.L.plhip_conv2d_calib_int8(self.h, C.byref(d), dx, calib_scale, dwp, ds, db, dy, out_kind), "conv2d_calib")
        y = self.to_host(dy, (d.n, d.cout, oh, ow), _OUT_DTYPE[out_kind])
        for p in [dx, dw, dy, dwp] + ([ds] if scale is not None else []) + ([db] if bias is not None else []):
            self.free(p)
        return y

    def conv2d_fused(self, d, x, w, scale, bias, residual, residual_relu, calib_scale, want_f32=True):
        """plhip_conv2d_int8_fused on host arrays: returns (y_f32 or None, y_i8 or None)."""
        oh, ow = out_hw(d)
        shape = (d.n, d.cout, oh, ow)
        dx = self.to_device(np.ascontiguousarray(x, np.int8))
        dw = self.to_device(np.ascontiguousarray(w, np.int8))
        ds = self.to_device(np.ascontiguousarray(scale, np.float32))
        db = self.to_device(np.ascontiguousarray(bias, np.float32)) if bias is not None else C.c_void_p()
        dr = self.to_device(np.ascontiguousarray(residual, np.float32)) if residual is not None else C.c_void_p()
        n_out = int(np.prod(shape))
        dyf = self.malloc(n_out * 4) if want_f32 else C.c_void_p()
        dyq = self.malloc(n_out) if calib_scale is not None else C.c_void_p()
        dwp = self.malloc(self.L.plhip_conv_packed_weight_bytes(C.byref(d)))
        self.check(self.L.plhip_pack_conv_weights(self.h, C.byref(d), dw, dwp), "pack")
        wsb = self.L.plhip_conv_workspace_bytes(C.byref(d))
        dws = self.malloc(wsb) if wsb else C.c_void_p()
        self.check(self.L.plhip_conv2d_int8_fused(self.h, C.byref(d), dx, dwp, ds, db, dyf, dr, int(residual_relu), dyq,
                                                  float(calib_scale) if calib_scale is not None else 0.0, dws, wsb), "conv2d_fused")
        yf = self.to_host(dyf, shape, np.float32) if want_f32 else None
        yq = self.to_host(dyq, shape, np.int8) if calib_scale is not None else None
        for p_ in [dx, dw, ds, dwp] + ([db] if bias is not None else []) + ([dr] if residual is not None else []) + \
                ([dyf] if want_f32 else []) + ([dyq] if calib_scale is not None else []) + ([dws] if wsb else []):
            self.free(p_)
        return yf, yq

    def dwpw_fused(self, d_dw, x, w_dw, s_dw, b_dw, w_pw, s_pw, b_pw, pw_act, pw_alpha, out_kind):
        """Fused depthwise -> pointwise through the C ABI (host arrays in, host array out)."""
        oh, ow = out_hw(d_dw)
        cout = w_pw.shape[0]
        d_pw = conv_desc(d_dw.n, d_dw.cin, oh, ow, cout, 1, 1, act=pw_act, alpha=pw_alpha)
        dx = self.to_device(np.ascontiguousarray(x, np.int8))
        dwd = self.to_device(np.ascontiguousarray(w_dw, np.int8))
        dsd = self.to_device(np.ascontiguousarray(s_dw, np.float32))
        dbd = self.to_device(np.ascontiguousarray(b_dw, np.float32)) if b_dw is not None else C.c_void_p()
        dwp_raw = self.to_device(np.ascontiguousarray(w_pw, np.int8))
        dwp = self.malloc(self.L.plhip_conv_packed_weight_bytes(C.byref(d_pw)))
        self.check(self.L.plhip_pack_conv_weights(self.h, C.byref(d_pw), dwp_raw, dwp), "pack")
        dsp = self.to_device(np.ascontiguousarray(s_pw, np.float32)) if s_pw is not None else C.c_void_p()
        dbp = self.to_device(np.ascontiguousarray(b_pw, np.float32)) if b_pw is not None else C.c_void_p()
        esz = 1 if out_kind == OUT_I8 else 4
        dy = self.malloc(d_dw.n * cout * oh * ow * esz)
        self.check(self.L.plhip_dwpw_fused_int8(self.h, C.byref(d_dw), dx, dwd, dsd, dbd, cout, dwp, dsp, dbp, pw_act, pw_alpha,
                                                dy, out_kind), "dwpw_fused")
        y = self.to_host(dy, (d_dw.n, cout, 1, 1), np.float32) if out_kind == OUT_F32_GAP else self.to_host(dy, (d_dw.n, cout, oh, ow), _OUT_DTYPE[out_kind])
        for p in [dx, dwd, dsd, dwp_raw, dwp, dy] + ([dbd] if b_dw is not None else []) + ([dsp] if s_pw is not None else []) + \
                ([dbp] if b_pw is not None else []):
            self.free(p)
        return y

    def fc(self, x, w, scale, bias, relu, out_kind):
        x = np.ascontiguousarray(x, np.int8)
        w = np.ascontiguousarray(w, np.int8)
        m, k = x.shape
        n = w.shape[1]
        dx, dw = self.to_device(x), self.to_device(w)
        dwp = self.malloc(self.L.plhip_fc_packed_weight_bytes(k, n))
        self.check(self.L.plhip_pack_fc_weights(self.h, k, n, dw, dwp), "pack_fc")
        ds = self.to_device(np.ascontiguousarray(scale, np.float32)) if scale is not None else C.c_void_p()
        db = self.to_device(np.ascontiguousarray(bias, np.float32)) if bias is not None else C.c_void_p()
        esz = 1 if out_kind == OUT_I8 else 4
        dy = self.malloc(m * n * esz)
        self.check(self.L.plhip_fc_int8(self.h, m, k, n, dx, dwp, ds, db, int(relu), dy, out_kind), "fc")
        y = self.to_host(dy, (m, n), _OUT_DTYPE[out_kind])
        for p in [dx, dw, dwp, dy] + ([ds] if scale is not None else []) + ([db] if bias is not None else []):
            self.free(p)
        return y

    def calib_f32_to_i8(self, x, scale):
        x = np.ascontiguousarray(x, np.float32)
        dx = self.to_device(x)
        dy = self.malloc(max(1, x.size))
        self.check(self.L.plhip_calib_f32_to_i8(self.h, dx, dy, scale, x.size), "calib_f32_to_i8")
        y = self.to_host(dy, x.shape, np.int8)
        self.free(dx), self.free(dy)
        return y

    def calib_i8_to_f32(self, x, scale):
        x = np.ascontiguousarray(x, np.int8)
        dx = self.to_device(x)
        dy = self.malloc(max(4, x.size * 4))
        self.check(self.L.plhip_calib_i8_to_f32(self.h, dx, dy, scale, x.size), "calib_i8_to_f32")
        y = self.to_host(dy, x.shape, np.float32)
        self.free(dx), self.free(dy)
        return y

    def global_avg_pool(self, x):
        x = np.ascontiguousarray(x, np.float32)
        n, c = x.shape[:2]
        sp = int(np.prod(x.shape[2:]))
        dx = self.to_device(x)
        dy = self.malloc(n * c * 4)
        self.check(self.L.plhip_global_avg_pool_f32(self.h, dx, n * c, sp, dy), "pool")
        y = self.to_host(dy, (n, c, 1, 1), np.float32)
        self.free(dx), self.free(dy)
        return y

    def softmax(self, x):
        x = np.ascontiguousarray(x, np.float32)
        rows, cols = int(np.prod(x.shape[:-1])), x.shape[-1]
        dx = self.to_device(x)
        dy = self.malloc(x.size * 4)
        self.check(self.L.plhip_softmax_f32(self.h, dx, rows, cols, dy), "softmax")
        y = self.to_host(dy, x.shape, np.float32)
        self.free(dx), self.free(dy)
        return y

    def pool2d(self, x, pooling_type, ksize, strides, pads, exclusive=True, ceil_mode=False):
        """x [n,c,h,w] fp32; pads {top, bottom, left, right}; output dims by PoolOutputSize (pool_op.cc:44-61)."""
        i8 = np.asarray(x).dtype == np.int8
        x = np.ascontiguousarray(x, np.int8 if i8 else np.float32)
        n, c, h, w = x.shape

        def osz(i, k, p0, p1, s):
            return (i - k + p0 + p1 + (s - 1 if ceil_mode else 0)) // s + 1
        d = PoolDesc()
        d.planes, d.h, d.w = n * c, h, w
        d.oh, d.ow = osz(h, ksize[0], pads[0], pads[1], strides[0]), osz(w, ksize[1], pads[2], pads[3], strides[1])
        d.kh, d.kw = ksize
        d.pad[:] = list(pads)
        d.stride[:] = list(strides)
        d.is_max, d.exclusive = int(pooling_type == "max"), int(exclusive)
        dx = self.to_device(x)
        dy = self.malloc(n * c * d.oh * d.ow * (1 if i8 else 4))
        fn = self.L.plhip_pool2d_max_i8 if i8 else self.L.plhip_pool2d_f32
        self.check(fn(self.h, C.byref(d), dx, dy), "pool2d")
        y = self.to_host(dy, (n, c, d.oh, d.ow), np.int8 if i8 else np.float32)
        self.free(dx), self.free(dy)
        return y

    def elementwise_add(self, x, y, relu=False):
        x = np.ascontiguousarray(x, np.float32)
        y = np.ascontiguousarray(y, np.float32)
        dx, dy = self.to_device(x), self.to_device(y)
        do = self.malloc(max(4, x.size * 4))
        self.check(self.L.plhip_elementwise_add_f32(self.h, dx, dy, do, x.size, int(relu)), "elementwise_add")
        o = self.to_host(do, x.shape, np.float32)
        self.free(dx), self.free(dy), self.free(do)
        return o
