"""ctypes binding of libpaddle_lite_hip.so (lite/api/lite_capi.h): drives the C++ KernelLite classes and the mini
predictor.  No fallback: a missing library or a failing call raises."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libpaddle_lite_hip.so")
PREC_FLOAT, PREC_INT8, PREC_ANY = 1, 2, 4
LAYOUT_NCHW, LAYOUT_ANY = 1, 2


class LiteError(RuntimeError):
    pass


_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise LiteError("%s is missing: run __graft_entry__.build()" % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    vp, i32, f32, i64, cs = C.c_void_p, C.c_int, C.c_float, C.c_int64, C.c_char_p
    L.pllite_last_error.restype = cs
    L.pllite_registered_kernels.argtypes = [cs, i32, i32]
    L.pllite_adopt_stream.argtypes = [i32, vp]
    L.pllite_packed_weight_cache_stats.argtypes = [C.POINTER(C.c_long), C.POINTER(C.c_long)]
    L.pllite_packed_weight_cache_stats.restype = None
    L.pllite_predictor_create.argtypes = [i32]
    L.pllite_predictor_create.restype = vp
    L.pllite_predictor_destroy.argtypes = [vp]
    L.pllite_predictor_destroy.restype = None
    L.pllite_add_feed.argtypes = [vp, cs, C.POINTER(i64), i32, i32]
    L.pllite_add_io_copy.argtypes = [vp, cs, cs, i32]
    L.pllite_add_calib.argtypes = [vp, cs, cs, f32, i32]
    L.pllite_add_conv.argtypes = [vp, cs, cs, cs, vp, C.POINTER(i64), vp, C.POINTER(i32), C.POINTER(i32), i32,
                                  C.POINTER(i32), i32, i32, f32, f32, vp, i32, f32, i32, cs]
    L.pllite_add_fc.argtypes = [vp, cs, cs, vp, i32, i32, vp, f32, vp, i32, f32, i32, i32]
    L.pllite_add_global_avg_pool.argtypes = [vp, cs, cs]
    L.pllite_add_softmax.argtypes = [vp, cs, cs]
    L.pllite_add_pool.argtypes = [vp, cs, cs, cs, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32), i32, i32, i32]
    L.pllite_add_elementwise_add.argtypes = [vp, cs, cs, cs, cs]
    L.pllite_predictor_create_planner.restype = vp
    L.pllite_graph_feed.argtypes = [vp, cs, C.POINTER(i64), i32, i32]
    L.pllite_graph_conv.argtypes = [vp, cs, cs, cs, vp, C.POINTER(i64), vp, C.POINTER(i32), C.POINTER(i32), i32,
                                    C.POINTER(i32), i32, i32, f32, f32, vp, i32, cs]
    L.pllite_graph_fc.argtypes = [vp, cs, cs, vp, i32, i32, vp, f32, vp, i32, i32]
    L.pllite_graph_pool.argtypes = [vp, cs, cs, cs, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32), i32, i32, i32]
    L.pllite_graph_elementwise_add.argtypes = [vp, cs, cs, cs, cs]
    L.pllite_graph_softmax.argtypes = [vp, cs, cs]
    L.pllite_graph_fetch.argtypes = [vp, cs]
    L.pllite_graph_set_fuse.argtypes = [vp, i32]
    L.pllite_graph_set_fuse_dwpw.argtypes = [vp, i32]
    L.pllite_graph_plan.argtypes = [vp, cs, i32]
    L.pllite_graph_lower.argtypes = [vp, cs, i32]
    L.pllite_load_model.argtypes = [vp, vp, i64, i32]
    L.pllite_graph_num_ops.argtypes = [vp]
    L.pllite_graph_op_params.argtypes = [vp, i32, cs, i32, vp, C.POINTER(i64), vp, C.POINTER(i32), vp, C.POINTER(i32),
                                         C.POINTER(f32), C.POINTER(i32), C.POINTER(f32)]
    L.pllite_set_input.argtypes = [vp, cs, vp, i64]
    L.pllite_run.argtypes = [vp, i32]
    L.pllite_run_graph.argtypes = [vp]
    L.pllite_sync.argtypes = [vp]
    L.pllite_num_instructions.argtypes = [vp]
    L.pllite_run_instruction.argtypes = [vp, i32]
    L.pllite_get_var.argtypes = [vp, cs, vp, i64, C.POINTER(i64), C.POINTER(i64), C.POINTER(i32)]
    L.pllite_var_device_ptr.argtypes = [vp, cs]
    L.pllite_var_device_ptr.restype = vp
    L.pllite_kernel_names.argtypes = [vp, cs, i32]
    L.pllite_time_instruction.argtypes = [vp, i32, i32, C.POINTER(f32), C.POINTER(f32), cs, i32]
    L.pllite_copy_var_to_device.argtypes = [vp, cs, vp, i64]
    _lib = L
    return L


def _ia(vals, t=C.c_int):
    return (t * len(vals))(*[int(v) for v in vals])


class Predictor:
    """Mini CxxPredictor on TARGET(kHIP) (lite/api/hip_predictor.h)."""

    def __init__(self, device=0, stream=None, planner=False):
        """planner=True: an object that can only build and plan a graph (no device touched) — CPU tests of the
        lowering rules."""
        self.L = load()
        if planner:
            self.h = self.L.pllite_predictor_create_planner()
        else:
            if stream is not None:
                self._ck(self.L.pllite_adopt_stream(device, C.c_void_p(stream)))
            self.h = self.L.pllite_predictor_create(device)
        if not self.h:
            raise LiteError("pllite_predictor_create: " + self.L.pllite_last_error().decode())
        self._keep = []

    def _ck(self, rc):
        if rc != 0:
            raise LiteError(self.L.pllite_last_error().decode())

    def close(self):
        if self.h:
            self.L.pllite_predictor_destroy(self.h)
            self.h = None

    def add_feed(self, name, dims, precision=PREC_FLOAT):
        self._ck(self.L.pllite_add_feed(self.h, name.encode(), _ia(dims, C.c_int64), len(dims), precision))

    def add_io_copy(self, src, dst, host_to_device=True):
        self._ck(self.L.pllite_add_io_copy(self.h, src.encode(), dst.encode(), int(host_to_device)))

    def add_calib(self, src, dst, scale, fp32_to_int8=True):
        self._ck(self.L.pllite_add_calib(self.h, src.encode(), dst.encode(), scale, int(fp32_to_int8)))

    def add_conv(self, op_type, src, dst, w, bias, strides, paddings, dilations, groups, act, act_coef, input_scale,
                 weight_scale, output_scale, int8_out, padding_algorithm=""):
        w = np.ascontiguousarray(w, np.int8)
        ws = np.ascontiguousarray(weight_scale, np.float32)
        bp = None
        if bias is not None:
            bias = np.ascontiguousarray(bias, np.float32)
            bp = bias.ctypes.data_as(C.c_void_p)
        self._ck(self.L.pllite_add_conv(self.h, op_type.encode(), src.encode(), dst.encode(), w.ctypes.data_as(C.c_void_p),
                                        _ia(w.shape, C.c_int64), bp, _ia(strides), _ia(paddings), len(paddings),
                                        _ia(dilations), groups, act, act_coef, input_scale, ws.ctypes.data_as(C.c_void_p),
                                        ws.size, output_scale, int(int8_out), padding_algorithm.encode()))

    def add_fc(self, src, dst, w, bias, input_scale, weight_scale, output_scale, int8_out, relu):
        w = np.ascontiguousarray(w, np.int8)
        ws = np.ascontiguousarray(weight_scale, np.float32)
        bp = None
        if bias is not None:
            bias = np.ascontiguousarray(bias, np.float32)
            bp = bias.ctypes.data_as(C.c_void_p)
        self._ck(self.L.pllite_add_fc(self.h, src.encode(), dst.encode(), w.ctypes.data_as(C.c_void_p), w.shape[0],
                                      w.shape[1], bp, input_scale, ws.ctypes.data_as(C.c_void_p), ws.size, output_scale,
                                      int(int8_out), int(relu)))

    def add_global_avg_pool(self, src, dst):
        self._ck(self.L.pllite_add_global_avg_pool(self.h, src.encode(), dst.encode()))

    def add_softmax(self, src, dst):
        self._ck(self.L.pllite_add_softmax(self.h, src.encode(), dst.encode()))

    def add_pool(self, src, dst, pooling_type, ksize, strides, paddings, global_pooling=False, exclusive=True,
                 ceil_mode=False):
        self._ck(self.L.pllite_add_pool(self.h, src.encode(), dst.encode(), pooling_type.encode(), _ia(ksize), _ia(strides),
                                        _ia(paddings), int(global_pooling), int(exclusive), int(ceil_mode)))

    def add_elementwise_add(self, x, y, dst, act_type=""):
        self._ck(self.L.pllite_add_elementwise_add(self.h, x.encode(), y.encode(), dst.encode(), act_type.encode()))

    # ---- graph mode: ops as the optimiser sees them; graph_lower() applies the reference's kernel-pick / cast rules
    def graph_feed(self, name, dims, precision=PREC_FLOAT):
        self._ck(self.L.pllite_graph_feed(self.h, name.encode(), _ia(dims, C.c_int64), len(dims), precision))

    def graph_conv(self, op_type, src, dst, w, bias, strides, paddings, dilations, groups, act, act_coef, input_scale,
                   weight_scale, padding_algorithm=""):
        w = np.ascontiguousarray(w, np.int8)
        ws = np.ascontiguousarray(weight_scale, np.float32)
        bp = None
        if bias is not None:
            bias = np.ascontiguousarray(bias, np.float32)
            bp = bias.ctypes.data_as(C.c_void_p)
        self._ck(self.L.pllite_graph_conv(self.h, op_type.encode(), src.encode(), dst.encode(), w.ctypes.data_as(C.c_void_p),
                                          _ia(w.shape, C.c_int64), bp, _ia(strides), _ia(paddings), len(paddings),
                                          _ia(dilations), groups, act, act_coef, input_scale, ws.ctypes.data_as(C.c_void_p),
                                          ws.size, padding_algorithm.encode()))

    def graph_fc(self, src, dst, w, bias, input_scale, weight_scale, relu=False):
        w = np.ascontiguousarray(w, np.int8)
        ws = np.ascontiguousarray(weight_scale, np.float32)
        bp = None
        if bias is not None:
            bias = np.ascontiguousarray(bias, np.float32)
            bp = bias.ctypes.data_as(C.c_void_p)
        self._ck(self.L.pllite_graph_fc(self.h, src.encode(), dst.encode(), w.ctypes.data_as(C.c_void_p), w.shape[0],
                                        w.shape[1], bp, input_scale, ws.ctypes.data_as(C.c_void_p), ws.size, int(relu)))

    def graph_pool(self, src, dst, pooling_type, ksize, strides, paddings, global_pooling=False, exclusive=True,
                   ceil_mode=False):
        self._ck(self.L.pllite_graph_pool(self.h, src.encode(), dst.encode(), pooling_type.encode(), _ia(ksize),
                                          _ia(strides), _ia(paddings), int(global_pooling), int(exclusive), int(ceil_mode)))

    def graph_elementwise_add(self, x, y, dst, act_type=""):
        self._ck(self.L.pllite_graph_elementwise_add(self.h, x.encode(), y.encode(), dst.encode(), act_type.encode()))

    def graph_softmax(self, src, dst):
        self._ck(self.L.pllite_graph_softmax(self.h, src.encode(), dst.encode()))

    def graph_set_fuse(self, on):
        self._ck(self.L.pllite_graph_set_fuse(self.h, int(on)))

    def graph_set_fuse_dwpw(self, on):
        """Opt-in: a depthwise conv [int8_out] takes its sole 1x1 consumer over (one instruction, one launch where the
        shape fits the fused kernel)."""
        self._ck(self.L.pllite_graph_set_fuse_dwpw(self.h, int(on)))

    def graph_fetch(self, name):
        self._ck(self.L.pllite_graph_fetch(self.h, name.encode()))

    def graph_plan(self):
        buf = C.create_string_buffer(1 << 18)
        self._ck(self.L.pllite_graph_plan(self.h, buf, len(buf)))
        return [s for s in buf.value.decode().split("\n") if s]

    def graph_lower(self):
        buf = C.create_string_buffer(1 << 14)
        self._ck(self.L.pllite_graph_lower(self.h, buf, len(buf)))
        return [s for s in buf.value.decode().split("\n") if s]

    def load_model(self, blob, batch):
        """Parse a PLHIPM01 container (bytes) into the predictor's graph (lite/model_parser/hip_model.h)."""
        buf = (C.c_char * len(blob)).from_buffer_copy(blob)
        self._ck(self.L.pllite_load_model(self.h, buf, len(blob), batch))

    def graph_ops(self):
        """[(type, w int8 flat, bias or None, weight_scale, input_scale, act, act_coef)] of the graph's ops."""
        res = []
        for i in range(self.L.pllite_graph_num_ops(self.h)):
            t = C.create_string_buffer(64)
            nw, nb, ns = C.c_int64(), C.c_int(), C.c_int()
            isc, act, coef = C.c_float(), C.c_int(), C.c_float()
            self._ck(self.L.pllite_graph_op_params(self.h, i, t, 64, None, C.byref(nw), None, C.byref(nb), None, C.byref(ns),
                                                   C.byref(isc), C.byref(act), C.byref(coef)))
            w = np.empty(nw.value, np.int8)
            b = np.empty(nb.value, np.float32)
            s_ = np.empty(ns.value, np.float32)
            self._ck(self.L.pllite_graph_op_params(self.h, i, None, 0, w.ctypes.data_as(C.c_void_p), None,
                                                   b.ctypes.data_as(C.c_void_p), None, s_.ctypes.data_as(C.c_void_p), None, None, None, None))
            res.append((t.value.decode(), w, b if nb.value else None, s_, isc.value, act.value, coef.value))
        return res

    def set_input(self, name, arr):
        arr = np.ascontiguousarray(arr)
        self._ck(self.L.pllite_set_input(self.h, name.encode(), arr.ctypes.data_as(C.c_void_p), arr.nbytes))

    def run(self, skip_io_copy=False):
        self._ck(self.L.pllite_run(self.h, int(skip_io_copy)))

    def run_graph(self):
        """Device part of the program as ONE recorded launch graph (records on the first call, after a plain run())."""
        self._ck(self.L.pllite_run_graph(self.h))

    def sync(self):
        self._ck(self.L.pllite_sync(self.h))

    def num_instructions(self):
        return self.L.pllite_num_instructions(self.h)

    def run_instruction(self, i):
        self._ck(self.L.pllite_run_instruction(self.h, i))

    def get_var(self, name, dtype, max_bytes=1 << 30):
        nb, nd = C.c_int64(), C.c_int()
        dims = (C.c_int64 * 4)()
        # first query the size with a tiny probe: capacity check raises, so allocate generously via dims
        buf = np.empty(max_bytes if max_bytes < (1 << 24) else (1 << 24), np.uint8)
        rc = self.L.pllite_get_var(self.h, name.encode(), buf.ctypes.data_as(C.c_void_p), buf.nbytes, C.byref(nb), dims, C.byref(nd))
        if rc != 0 and "too small" in self.L.pllite_last_error().decode():
            buf = np.empty(max_bytes, np.uint8)
            rc = self.L.pllite_get_var(self.h, name.encode(), buf.ctypes.data_as(C.c_void_p), buf.nbytes, C.byref(nb), dims, C.byref(nd))
        self._ck(rc)
        shape = tuple(dims[i] for i in range(nd.value))
        return buf[:nb.value].view(dtype).reshape(shape).copy()

    def device_ptr(self, name):
        return self.L.pllite_var_device_ptr(self.h, name.encode())

    def copy_var_to_device(self, name, dst_ptr, nbytes):
        self._ck(self.L.pllite_copy_var_to_device(self.h, name.encode(), C.c_void_p(dst_ptr), nbytes))

    def time_instruction(self, index, reps=10):
        """(avg_ms, min_ms, kernel_func_name) of one instruction, timed with DeviceTimer<kHIP> (lite/core/profile/timer.h)."""
        a, m = C.c_float(), C.c_float()
        buf = C.create_string_buffer(256)
        self._ck(self.L.pllite_time_instruction(self.h, index, reps, C.byref(a), C.byref(m), buf, len(buf)))
        return a.value, m.value, buf.value.decode()

    def kernel_names(self):
        buf = C.create_string_buffer(1 << 16)
        self._ck(self.L.pllite_kernel_names(self.h, buf, len(buf)))
        return [s for s in buf.value.decode().split("\n") if s]
